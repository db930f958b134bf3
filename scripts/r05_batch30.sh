#!/bin/bash
# round 5, batch 30: per-unit durations of the main SpMV launch over 24 runs -- how much of the tail repeats (what a re-cut from
# measured times could take back)
cd /root/repo; mkdir -p gpurun_out
for g in orkut pokec ogbn_products hollywood; do GRAPHLILY_HIP_LIB=$PWD/scripts/_variants/clocks.so timeout 600 python scripts/unit_clocks.py $g 4 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05_unit_clocks.txt
GRAPHLILY_HIP_LIB=$PWD/scripts/_variants/clocks.so timeout 600 python scripts/unit_clocks.py orkut 0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_unit_clocks.txt
