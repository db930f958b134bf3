#!/bin/bash
# round 5, batch 27: bin kernel with 512-thread workgroups, two per CU (-DGL_BIN_THREADS=512): parity tests on that build, then
# same-box A/B of blocking calls against the 1024-thread build
cd /root/repo; mkdir -p gpurun_out
GRAPHLILY_HIP_LIB=$PWD/scripts/_variants/bt512.so timeout 1800 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
for rep in 1 2; do for c in "orkut 0.9" "orkut 0.95" "hollywood 0.9" "hollywood 0.95" "ogbn_products 0.9" "pokec 0.9" "orkut 0.99" "hollywood 0.99" "ogbn_products 0.99" "hollywood 0.995" "hollywood 0.9995" "googleplus 0.999"; do for v in cur bt512; do
lib=""; [ "$v" != cur ] && lib=scripts/_variants/$v.so
echo -n "$c $v: "; GRAPHLILY_HIP_LIB=$lib timeout 300 python scripts/spmspv_call_trace.py $c 2>&1 | grep "blocking\|enqueued" | tr '\n' ' '; echo
done; done; done | tee gpurun_out/r05_ab_spmspv_512_threads.txt
