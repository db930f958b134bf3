# round-4 final batch, part D: the GPU suite and the bench line once more on the last code (after the SpMSpV fold / one-launch
# kernel changes), and the driver's exact command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r04_gputests_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_gputests_final.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04_gputests_final.log | tail -9
timeout 900 python bench.py > gpurun_out/r04_bench_orkut_n1.json 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_orkut_n1_steps20_warmup5.json 2>> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("r04_bench_orkut_n1", "r04_bench_orkut_n1_steps20_warmup5"):
    d = [json.loads(l) for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][0]
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: d["bfs"][k]["ms"] for k in ("pull_push", "pull")}, d["bfs"].get("host_unpack_ms"), [c["blocking_call_ms"] for c in d["spmspv"]["cases"]])
    for k, v in d["six_graphs"].items():
        if not k.startswith("_"): print("   ", k, v["spmv"]["ms"], v["spmv"]["frac_hbm_peak"], v["spmv"]["kernel_frac_hbm_peak"], v["bfs"]["pull_push_ms"], v["pagerank"]["ms_per_iter"], v["sssp"]["pull_push_ms"])
PY
