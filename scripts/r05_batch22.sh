#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
GRAPHLILY_HIP_LIB=$PWD/scripts/_variants/stamps0.so timeout 600 python scripts/spmspv_stamps.py --graph hollywood --sparsity 0.9 0.95 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_spmspv_stamps_batches.txt
GRAPHLILY_HIP_LIB=$PWD/scripts/_variants/stamps0.so timeout 600 python scripts/spmspv_stamps.py --graph orkut --sparsity 0.9 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_spmspv_stamps_batches.txt
