# round-2 final measurement batch (one box): GPU suite, bench line, rocprofv3 passes, six graphs, SpMSpV sweep,
# the reference's benchmark drivers, the two-rank leg of bench.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_margins.jsonl gpurun_out/format_seconds.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02_gputests_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_gputests_final.log
tail -14 gpurun_out/r02_gputests_final.log
timeout 600 python bench.py --steps 100 --warmup 5 > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err
echo "bench rc=$?"; cut -c1-900 gpurun_out/r02_bench_final.json
timeout 1200 bash scripts/profile_bench.sh > gpurun_out/r02_profile.log 2>&1; tail -3 gpurun_out/r02_profile.log
timeout 900 python benchmarks/bench_graphs.py --out gpurun_out/r02_six_graphs.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-260
timeout 900 python benchmarks/bench_spmspv.py --semirings Arithmetic,Logical --out gpurun_out/r02_spmspv_sweep.jsonl > gpurun_out/r02_spmspv_sweep.log 2>&1; tail -2 gpurun_out/r02_spmspv_sweep.log | cut -c1-200
timeout 600 python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank,sssp 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_reference_benches_on_hip.txt
timeout 600 python bench.py --gpus 2 --backend gloo --same-gpu --steps 20 --no-cpu-baseline > gpurun_out/r02_bench_2ranks_same_gpu.json 2> gpurun_out/r02_bench_2ranks.err; echo "2-rank rc=$?"; cut -c1-600 gpurun_out/r02_bench_2ranks_same_gpu.json
