export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for G in 256; do
cd /tmp && GRAPHLILY_BFS_DECIDE_GRID=$G rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace_emu -- python $R/scripts/r03_emulate_trace.py orkut 0/8 pull_push 2>&1 | grep "CALL 4" | cut -c1-60
cd $R; echo "== decide grid $G"; python scripts/r03_emulate_trace_summary.py gpurun_out/trace_emu | grep "decide\|total"; rm -rf gpurun_out/trace_emu
done
