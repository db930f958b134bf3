#!/bin/bash
# round 5, the final measurement batch on the last code: full GPU suite, the driver's bench command + the default line, rocprofv3
# stats + PMC passes (orkut, ogbn-products), standalone six-graphs leg, the SpMSpV sweep through the C++ module layer, the
# reference's own bench drivers over both app layers
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
PY
bash scripts/profile_bench.sh > gpurun_out/r05_profile_bench.log 2>&1; tail -3 gpurun_out/r05_profile_bench.log
GRAPH=ogbn_products bash scripts/profile_bench.sh > gpurun_out/r05_profile_bench_products.log 2>&1; tail -2 gpurun_out/r05_profile_bench_products.log
timeout 900 python benchmarks/bench_graphs.py --out gpurun_out/r05_six_graphs.jsonl 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python benchmarks/run_spmspv_cpp.py --out gpurun_out/r05_spmspv_sweep_cpp.jsonl 2>&1 | tail -1
timeout 1500 python benchmarks/run_reference_benches.py --graph orkut --apps spmv,spmv_verify,bfs,pagerank,sssp 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_reference_benches_on_hip.txt
timeout 1500 python benchmarks/run_reference_benches.py --graph orkut --apps bfs_refapps,pagerank_refapps,sssp_refapps 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_reference_benches_on_hip_refapps.txt
grep "average_time\|iteration\|passed" gpurun_out/r05_reference_benches_on_hip.txt gpurun_out/r05_reference_benches_on_hip_refapps.txt
