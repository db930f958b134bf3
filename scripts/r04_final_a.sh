# round-4 final measurement batch, part A (one box): GPU suite, the bench line (+ emulated ranks incl. PageRank / SSSP pull),
# rocprofv3 kernel stats + PMC passes for orkut and ogbn-products
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out build
rm -f gpurun_out/fullsize_margins.jsonl
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r04_gputests_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_gputests_final.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04_gputests_final.log | tail -14
cp gpurun_out/fullsize_margins.jsonl gpurun_out/r04_fullsize_margins.jsonl 2>/dev/null
( time timeout 900 python bench.py > gpurun_out/r04_bench_orkut_n1.json 2> gpurun_out/r04_bench_final.err ) 2>&1 | grep real
echo "bench rc=$?"; cut -c1-700 gpurun_out/r04_bench_orkut_n1.json
timeout 1500 python bench.py --emulate-rank 0/8,3/8,7/8,1/4,0/2 --no-cpu-baseline --no-six-graphs --no-spmspv --no-pattern > gpurun_out/r04_bench_orkut_emulated_ranks.json 2>> gpurun_out/r04_bench_final.err
python - <<'PY'
import json
for f in ("gpurun_out/r04_bench_orkut_n1.json", "gpurun_out/r04_bench_orkut_emulated_ranks.json"):
    for l in open(f):
        if l.startswith('{"metric'):
            d = json.loads(l)
            print(f, d["value"], d["roofline"]["frac"], {k: d["bfs"][k]["ms"] for k in ("pull_push", "pull")}, d["bfs"].get("host_unpack_ms"))
            for e in d.get("bfs_emulated_ranks", []):
                print("   rank %d/%d" % (e["rank"], e["world"]), e["pull_push"]["schedule_ms"], e["pull"]["schedule_ms"], e.get("pagerank"), e.get("sssp_pull"))
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o build/ubench_gather scripts/ubench_gather.hip 2>/dev/null
timeout 1500 bash scripts/profile_bench.sh > gpurun_out/r04_profile.log 2>&1; tail -2 gpurun_out/r04_profile.log
GRAPH=ogbn_products timeout 1200 bash scripts/profile_bench.sh > gpurun_out/r04_profile_products.log 2>&1; tail -2 gpurun_out/r04_profile_products.log
