cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python benchmarks/bench_graphs.py --graphs hollywood --out gpurun_out/r02_six_graphs_hollywood_rerun.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-160
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^CALL" /tmp/bfs_trace.log
python scripts/r02_bfs_trace_summary.py /tmp/bfs_trace /tmp/bfs_trace.log | tee gpurun_out/r02_bfs_pull_push_trace.txt
