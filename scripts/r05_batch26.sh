#!/bin/bash
# round 5, batch 26: the reference's bench_spmspv.cpp, unmodified, on stand-ins written where it has its dataset directory compiled in
cd /root/repo; mkdir -p gpurun_out
df -h / /tmp 2>/dev/null | tail -2; free -g | head -2
timeout 2400 python benchmarks/run_reference_benches.py --apps spmspv 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_reference_bench_spmspv.txt
tail -70 gpurun_out/r05_reference_bench_spmspv.txt
