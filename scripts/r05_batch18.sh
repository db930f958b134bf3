#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
GRAPHLILY_HIP_LIB=$PWD/scripts/_variants/stamps0.so timeout 600 python scripts/spmspv_stamps.py --graph hollywood --sparsity 0.99 0.9995 2>&1 | grep -v amdgpu.ids | tail -3
