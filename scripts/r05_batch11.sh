# round 5, batch 11: full GPU suite on the delta-coded build, the driver's bench command + the default line, rocprofv3 stats + PMC
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/fullsize_margins.jsonl
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r05_gputests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_gputests.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r05_gputests.log | tail -14
cp gpurun_out/fullsize_margins.jsonl gpurun_out/r05_fullsize_margins.jsonl 2>/dev/null
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_orkut_n1_steps20_warmup5.json 2> gpurun_out/r05_bench.err; echo "bench rc=$?"
timeout 900 python bench.py > gpurun_out/r05_bench_orkut_n1.json 2>> gpurun_out/r05_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("r05_bench_orkut_n1_steps20_warmup5", "r05_bench_orkut_n1"):
    d = [json.loads(l) for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][0]
    print(f, json.dumps(d["headline"]))
    print("   pattern", d["pattern_plan"].get("ms_per_step"), d["pattern_plan"].get("frac_hbm_peak"), d["pattern_plan"].get("bytes_per_nnz"), "spmspv", [(c["blocking_call_ms"], c["gbps"]) for c in d["spmspv"]["cases"]])
    for k, v in d["six_graphs"].items():
        if not k.startswith("_"): print("   ", k, v["spmv"]["ms"], v["spmv"]["frac_hbm_peak"], v["spmv"]["kernel_frac_hbm_peak"], "pat", v["spmv_pattern"]["ms"], v.get("bfs", {}).get("pull_push_ms"), v.get("bfs", {}).get("pull_ms"), v.get("pagerank", {}).get("ms_per_iter"), v.get("sssp", {}).get("pull_push_ms"))
    print("   tail has bfs:", '"bfs": {"pull_push"' in json.dumps(d)[-3000:], "line bytes", len(json.dumps(d)))
PY
bash scripts/profile_bench.sh > gpurun_out/r05_profile_bench.log 2>&1; tail -3 gpurun_out/r05_profile_bench.log
python scripts/pmc_summary.py r05 2>&1 | tail -40
