#!/bin/bash
# round 5, batch 16: the bin kernel's no-rendezvous path widened from 4 to 16 windows (c consecutive entries per thread, one block
# scan) -- parity tests, then same-box A/B of blocking calls against the build before it
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
for rep in 1 2; do for c in "hollywood 0.99" "hollywood 0.995" "hollywood 0.998" "ogbn_products 0.995" "orkut 0.995" "orkut 0.998" "pokec 0.99" "hollywood 0.9995" "hollywood 0.9"; do for v in r05c cur; do
lib=""; [ "$v" != cur ] && lib=scripts/_variants/$v.so
echo -n "$c $v: "; GRAPHLILY_HIP_LIB=$lib timeout 300 python scripts/spmspv_call_trace.py $c 2>&1 | grep "blocking\|enqueued" | tr '\n' ' '; echo
done; done; done | tee gpurun_out/r05_ab_spmspv_local_windows.txt
