cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_margins.jsonl
timeout 1700 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r02_gputests_s8.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_gputests_s8.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r02_gputests_s8.log | tail -14
timeout 600 python bench.py --steps 100 --warmup 5 > gpurun_out/r02_bench_s8.json 2> gpurun_out/r02_bench_s8.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/r02_bench_s8.json; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_s8.json')); print(json.dumps(d['bfs'])); print(json.dumps(d['roofline']))"
