#!/bin/bash
# round 5, batch 25: SpMSpV ranges cut by products + col_cost x columns -- parity tests, then the column weight swept on one box
# (0 = the cut by products), blocking calls
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
for rep in 1 2; do for c in "orkut 0.9" "orkut 0.95" "hollywood 0.9" "hollywood 0.95" "ogbn_products 0.9" "pokec 0.9" "orkut 0.99" "hollywood 0.99" "hollywood 0.9995"; do for k in 0 2 4 8 16; do
echo -n "$c col_cost=$k: "; GRAPHLILY_DEBUG=spmspv_col_cost=$k timeout 300 python scripts/spmspv_call_trace.py $c 2>&1 | grep "blocking\|enqueued" | tr '\n' ' '; echo
done; done; done | tee gpurun_out/r05_spmspv_col_cost_sweep.txt
