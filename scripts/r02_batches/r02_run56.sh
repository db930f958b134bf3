cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 100 --warmup 5 > gpurun_out/r02_bench_try.json 2> gpurun_out/r02_bench_try.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_try.json')); print(d['value'], d['roofline']['frac'], d['bfs'], d['selfcheck_ok'])"
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
