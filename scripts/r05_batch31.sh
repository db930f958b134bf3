#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for a in "ogbn_products 4" "orkut 0" "hollywood 4"; do GRAPHLILY_HIP_LIB=$PWD/scripts/_variants/clocks.so timeout 600 python scripts/unit_clocks.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05_unit_clocks_hubs.txt
