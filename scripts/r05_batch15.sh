#!/bin/bash
# round 5, batch 15: phase stamps of the SpMSpV kernels in blocking calls, every workgroup (scripts/spmspv_stamps.py)
cd /root/repo; mkdir -p gpurun_out
export GRAPHLILY_HIP_LIB=$PWD/scripts/_variants/stamps.so
for g in hollywood orkut; do
  timeout 600 python scripts/spmspv_stamps.py --graph $g --sparsity 0.9 0.99 0.995 0.9995 >> gpurun_out/r05_spmspv_stamps.txt 2>&1
done
tail -5 gpurun_out/r05_spmspv_stamps.txt
