cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputests_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_gputests_final.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r02_gputests_final.log | tail -6
timeout 900 python benchmarks/bench_spmspv.py --semirings Arithmetic,Logical --out gpurun_out/r02_spmspv_sweep.jsonl > gpurun_out/r02_spmspv_sweep.log 2>&1; tail -2 gpurun_out/r02_spmspv_sweep.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 100 --warmup 5 > gpurun_out/r02_bench_last.json 2> gpurun_out/r02_bench_last.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_last.json')); print(d['value'], d['roofline']['frac'], d['bfs'])"
