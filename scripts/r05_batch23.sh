#!/bin/bash
# round 5, batch 23: bin kernel, the next batch's stream loads issued before the current batch is sorted (pf1; pf2 = cur: + tile and
# rank packed into one register) -- parity tests, then same-box A/B of blocking calls
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
for rep in 1 2; do for c in "hollywood 0.9" "hollywood 0.95" "orkut 0.9" "orkut 0.95" "ogbn_products 0.9" "pokec 0.9" "hollywood 0.99" "orkut 0.99"; do for v in r05c pf1 cur; do
lib=""; [ "$v" != cur ] && lib=scripts/_variants/$v.so
echo -n "$c $v: "; GRAPHLILY_HIP_LIB=$lib timeout 300 python scripts/spmspv_call_trace.py $c 2>&1 | grep "blocking\|enqueued" | tr '\n' ' '; echo
done; done; done | tee gpurun_out/r05_ab_spmspv_prefetch.txt
