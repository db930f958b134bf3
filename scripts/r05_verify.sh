#!/bin/bash
# round 5: last verification on the last code -- the full GPU suite, smoke, the driver's bench command, the reference's drivers
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_gputests.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r05_gputests.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_verify.json 2> gpurun_out/r05_bench_verify.err; echo "bench rc=$?"
python -c "
import json
d=[json.loads(l) for l in open('gpurun_out/r05_bench_verify.json') if l.startswith('{')][0]
print(json.dumps(d['headline'])); print(d['roofline']['traffic'], d['six_graphs']['pokec']['spmv']['frac_hbm_peak'], d['six_graphs']['orkut']['pagerank'])"
timeout 1500 python benchmarks/run_reference_benches.py --graph orkut --apps spmv,spmv_verify,bfs,pagerank,sssp 2>&1 | grep "average_time\|iteration:\|passed"
