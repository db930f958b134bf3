#!/bin/bash
# round 5, batch 17: does it matter that the wavefront that polls is the one that published?  (-DGL_POLL_WAVE=1: wavefront 1 polls)
cd /root/repo; mkdir -p gpurun_out
for v in stamps0 stamps1; do echo "#### $v"; GRAPHLILY_HIP_LIB=$PWD/scripts/_variants/$v.so timeout 600 python scripts/spmspv_stamps.py --graph hollywood --sparsity 0.99 0.9995 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_spmspv_stamps_poll_wave.txt
for rep in 1 2; do for c in "hollywood 0.99" "orkut 0.99" "ogbn_products 0.99" "hollywood 0.9995" "googleplus 0.999" "pokec 0.999"; do for v in r05c poll1; do
echo -n "$c $v: "; GRAPHLILY_HIP_LIB=scripts/_variants/$v.so timeout 300 python scripts/spmspv_call_trace.py $c 2>&1 | grep "blocking\|enqueued" | tr '\n' ' '; echo
done; done; done | tee gpurun_out/r05_ab_spmspv_poll_wave.txt
