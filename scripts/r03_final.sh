# round-3 measurement batch (one box): GPU suite with full-size margins, the bench line (+ emulated ranks), rocprofv3
# passes, six graphs, SpMSpV sweep, the reference's own benchmark drivers + their C-ABI call timelines, the 2- and 8-rank
# same-GPU legs of bench.py, the BFS timelines (one GPU and one emulated rank of 8), SSSP loops
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_margins.jsonl
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r03_gputests_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_gputests_final.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r03_gputests_final.log | tail -14
cp gpurun_out/fullsize_margins.jsonl gpurun_out/r03_fullsize_margins.jsonl
timeout 900 python bench.py --emulate-rank 0/8,3/8,7/8,1/4,0/2 > gpurun_out/r03_bench_orkut_n1.json 2> gpurun_out/r03_bench_final.err
echo "bench rc=$?"; cut -c1-1200 gpurun_out/r03_bench_orkut_n1.json
# the same emulated ranks with three launches per slot (push step, pull step, gl_bfs_bits_decide), and the one-GPU BFS with
# two launches per slot: same-box A/B of the one-launch slot
GRAPHLILY_BFS_SHARD_STEP=0 GRAPHLILY_BFS_ONE_LAUNCH=0 timeout 900 python bench.py --emulate-rank 0/8,1/4,0/2 --no-cpu-baseline --no-pattern > gpurun_out/r03_bench_orkut_n1_three_launch_slots.json 2>> gpurun_out/r03_bench_final.err
python - <<'PY'
import json
for f in ("gpurun_out/r03_bench_orkut_n1.json", "gpurun_out/r03_bench_orkut_n1_three_launch_slots.json"):
    for l in open(f):
        if l.startswith('{"metric'):
            d = json.loads(l)
            print(f, {k: d["bfs"][k]["ms"] for k in ("pull_push", "pull")}, [(e["rank"], e["world"], e["pull_push"]["schedule_ms"], e["pull"]["schedule_ms"]) for e in d.get("bfs_emulated_ranks", [])])
PY
timeout 300 python scripts/r03_shard_shape_sweep.py orkut 0/8 0x0,32x8,64x4,128x2,256x1 2>&1 | grep shape | tee gpurun_out/r03_shard_shape_sweep.txt
mkdir -p build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench_sync.hip -o build/ubench_sync && build/ubench_sync | tee gpurun_out/r03_ubench_sync.txt
for c in "hollywood 0.999" "ogbn_products 0.9995" "googleplus 0.999" "pokec 0.9995"; do set -- $c; for h in 1 0; do echo "== $1 $2 work hint $h"; GRAPHLILY_SPMSPV_WORK_HINT=$h python scripts/r03_spmspv_call_trace.py $1 $2 2>&1 | grep "blocking\|enqueued"; done; done | tee gpurun_out/r03_spmspv_call_hint_ab.txt
timeout 1500 bash scripts/profile_bench.sh > gpurun_out/r03_profile.log 2>&1; tail -3 gpurun_out/r03_profile.log
timeout 900 python benchmarks/bench_graphs.py --out gpurun_out/r03_six_graphs.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-260
timeout 600 python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank,sssp 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_reference_benches_on_hip.txt
bash scripts/r03_refbench_timeline.sh orkut > gpurun_out/r03_refbench_timeline.log 2>&1
cp gpurun_out/api_bfs.txt gpurun_out/r03_api_timeline_bench_bfs.txt; cp gpurun_out/api_pagerank.txt gpurun_out/r03_api_timeline_bench_pagerank.txt
timeout 600 python bench.py --gpus 2 --backend gloo --same-gpu --steps 20 --no-cpu-baseline > gpurun_out/r03_bench_2ranks_same_gpu.json 2> gpurun_out/r03_bench_2ranks.err; echo "2-rank rc=$?"; cut -c1-400 gpurun_out/r03_bench_2ranks_same_gpu.json
timeout 900 python bench.py --gpus 8 --backend gloo --same-gpu --steps 20 --no-cpu-baseline --no-pattern > gpurun_out/r03_bench_8ranks_same_gpu.json 2> gpurun_out/r03_bench_8ranks.err; echo "8-rank rc=$?"; cut -c1-400 gpurun_out/r03_bench_8ranks_same_gpu.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^CALL" /tmp/bfs_trace.log | tail -2
python scripts/r02_timeline.py /tmp/bfs_trace > gpurun_out/r03_bfs_bits_timeline.txt; tail -3 gpurun_out/r03_bfs_bits_timeline.txt
for spec in "0/8 pull_push" "3/8 pull_push" "0/8 pull" "1/4 pull_push" "0/2 pull_push"; do set -- $spec
  cd /tmp && rm -rf /tmp/emu_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/emu_trace -- python $GRAFT_REPO_ROOT/scripts/r03_emulate_trace.py orkut $1 $2 > /tmp/emu_trace.log 2>&1
  cd $GRAFT_REPO_ROOT; n=$(echo $1 | tr / of); { grep "^CALL 4" /tmp/emu_trace.log; python scripts/r03_emulate_trace_summary.py /tmp/emu_trace; } > gpurun_out/r03_emulated_rank${n}_$2_timeline.txt; tail -1 gpurun_out/r03_emulated_rank${n}_$2_timeline.txt
done
timeout 600 python scripts/r03_sssp_loops.py orkut pokec 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_sssp_loops.txt
timeout 2400 python benchmarks/bench_spmspv.py --out gpurun_out/r03_spmspv_sweep.jsonl > gpurun_out/r03_spmspv_sweep.log 2>&1; tail -2 gpurun_out/r03_spmspv_sweep.log | cut -c1-200
