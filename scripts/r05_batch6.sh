# round 5, batch 6: SpMSpV pattern records (4-byte bins for column-constant matrices) -- parity tests, the C++ apps test, the
# bench-line SpMSpV leg and the sweep through the C++ module layer; dist pre-flight tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py tests/test_gpu_typed.py tests/test_cpp_layer.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25
timeout 900 python benchmarks/run_spmspv_cpp.py --out gpurun_out/r05_spmspv_sweep_cpp.jsonl 2>&1 | tail -5
python - <<'PY'
import json
new = {(r["graph"], r["semiring"], r["vector_sparsity"]): r for r in map(json.loads, open("gpurun_out/r05_spmspv_sweep_cpp.jsonl"))}
old = {(r["graph"], r["semiring"], r["vector_sparsity"]): r for r in map(json.loads, open("profiles/r04_spmspv_sweep_cpp.jsonl"))}
print("verified", sum(1 for r in new.values() if r["verified"]), "of", len(new))
for k in sorted(new):
    if k[1] == "Arithmetic" and k in old:
        print("%-30s %.4f  r04 %.4f ms -> %.4f ms  (x%.2f)  %7.1f GB/s" % (k[0], k[2], old[k]["ms"], new[k]["ms"], old[k]["ms"] / new[k]["ms"], new[k]["gbps"]))
PY
