# round 5, batch 7: SpMSpV rendezvous with block-wide polling (one round of loads + one block scan instead of up to eight
# dependent rounds) -- parity tests, then same-box A/B of blocking calls against round 4's build, then the C++ sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py tests/test_cpp_layer.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12
for rep in 1 2; do for c in "hollywood 0.99" "hollywood 0.9" "ogbn_products 0.99" "orkut 0.99" "pokec 0.95" "hollywood 0.9995"; do for v in r04 cur; do
lib=""; [ "$v" != cur ] && lib=scripts/_variants/$v.so
echo -n "$c $v: "; GRAPHLILY_HIP_LIB=$lib timeout 300 python scripts/spmspv_call_trace.py $c 2>&1 | grep "blocking\|enqueued" | tr '\n' ' '; echo
done; done; done | tee gpurun_out/r05_ab_spmspv_rendezvous.txt
timeout 900 python benchmarks/run_spmspv_cpp.py --out gpurun_out/r05_spmspv_sweep_cpp.jsonl 2>&1 | tail -2
python - <<'PY'
import json
new = {(r["graph"], r["semiring"], r["vector_sparsity"]): r for r in map(json.loads, open("gpurun_out/r05_spmspv_sweep_cpp.jsonl"))}
old = {(r["graph"], r["semiring"], r["vector_sparsity"]): r for r in map(json.loads, open("profiles/r04_spmspv_sweep_cpp.jsonl"))}
print("verified", sum(1 for r in new.values() if r["verified"]), "of", len(new))
for k in sorted(new):
    if k[1] == "Arithmetic" and k in old:
        print("%-30s %.4f  r04 %.4f ms -> %.4f ms  (x%.2f)  %7.1f GB/s" % (k[0], k[2], old[k]["ms"], new[k]["ms"], old[k]["ms"] / new[k]["ms"], new[k]["gbps"]))
PY
