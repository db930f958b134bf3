#!/bin/bash
# round 5, batch 19: SpMSpV bin kernel, mid-size vectors cut by ENTRIES when the matrix has no long column (no rendezvous) --
# parity tests, then the C++ sweep twice on the same box: the knob that forces the old cut, then the default
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
GRAPHLILY_DEBUG=spmspv_by_entries_maxcol=0 timeout 900 python benchmarks/run_spmspv_cpp.py --out gpurun_out/r05_spmspv_sweep_cpp_by_products.jsonl 2>&1 | tail -2
timeout 900 python benchmarks/run_spmspv_cpp.py --out gpurun_out/r05_spmspv_sweep_cpp_by_entries.jsonl 2>&1 | tail -2
python - <<'PY'
import json
new = {(r["graph"], r["semiring"], r["vector_sparsity"]): r for r in map(json.loads, open("gpurun_out/r05_spmspv_sweep_cpp_by_entries.jsonl"))}
old = {(r["graph"], r["semiring"], r["vector_sparsity"]): r for r in map(json.loads, open("gpurun_out/r05_spmspv_sweep_cpp_by_products.jsonl"))}
print("verified", sum(1 for r in new.values() if r["verified"]), "of", len(new))
for k in sorted(new):
    if k[1] == "Arithmetic" and k in old:
        print("%-30s %.4f  %7d entries  by products %.4f ms -> by entries %.4f ms  (x%.2f)" % (k[0], k[2], new[k]["vector_nnz"], old[k]["ms"], new[k]["ms"], old[k]["ms"] / new[k]["ms"]))
PY
