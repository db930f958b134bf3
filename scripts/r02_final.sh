# round-2 final measurement batch (one box): GPU suite, bench line, rocprofv3 passes, six graphs, SpMSpV sweep,
# the reference's benchmark drivers, the two-rank leg of bench.py, the BFS timelines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_margins.jsonl gpurun_out/format_seconds.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02_gputests_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_gputests_final.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r02_gputests_final.log | tail -14
timeout 600 python bench.py --steps 100 --warmup 5 > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err
echo "bench rc=$?"; cut -c1-900 gpurun_out/r02_bench_final.json
timeout 1200 bash scripts/profile_bench.sh > gpurun_out/r02_profile.log 2>&1; tail -3 gpurun_out/r02_profile.log
timeout 900 python benchmarks/bench_graphs.py --out gpurun_out/r02_six_graphs.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-260
timeout 900 python benchmarks/bench_spmspv.py --semirings Arithmetic,Logical --out gpurun_out/r02_spmspv_sweep.jsonl > gpurun_out/r02_spmspv_sweep.log 2>&1; tail -2 gpurun_out/r02_spmspv_sweep.log | cut -c1-200
timeout 600 python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank,sssp 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_reference_benches_on_hip.txt
timeout 600 python bench.py --gpus 2 --backend gloo --same-gpu --steps 20 --no-cpu-baseline > gpurun_out/r02_bench_2ranks_same_gpu.json 2> gpurun_out/r02_bench_2ranks.err; echo "2-rank rc=$?"; cut -c1-600 gpurun_out/r02_bench_2ranks_same_gpu.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^CALL" /tmp/bfs_trace.log | tail -2
python scripts/r02_timeline.py /tmp/bfs_trace > gpurun_out/r02_bfs_bits_timeline.txt; tail -3 gpurun_out/r02_bfs_bits_timeline.txt
python scripts/r02_bfs_trace_summary.py /tmp/bfs_trace /tmp/bfs_trace.log | tee gpurun_out/r02_bfs_pull_push_trace.txt | head -8
for g in orkut pokec googleplus hollywood ogbl_ppa ogbn_products; do python scripts/r02_bfs_loop.py $g 2>&1 | grep "device_loop\|pull bits" | head -6 | cut -c1-75,200- ; done | tee gpurun_out/r02_bfs_schedules.txt
