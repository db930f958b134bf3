cd /root/repo
for it in 100 400 1600; do for lib in "" scripts/_variants/r6base.so; do
echo -n "orkut pattern iters=$it lib=${lib:-cur}: "; GRAPHLILY_HIP_LIB=$lib timeout 600 python scripts/probe_spmv.py --graph orkut --flags 0 --no-copy --iters $it 2>&1 | grep "^op 0 mask 0"
done; done 2>&1 | tee gpurun_out/r06_sustained_probe.txt
timeout 600 python benchmarks/bench_graphs.py --graphs orkut --out gpurun_out/r06_bench_graphs_orkut.jsonl 2>&1 | grep -v amdgpu | tail -3
