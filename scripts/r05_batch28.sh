#!/bin/bash
# round 5, batch 28: chained SpMV runs (gl_spmv_run_chained) -- parity tests, then PageRank / SSSP pull with and without
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_apps.py tests/test_gpu_format.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
for g in orkut ogbn_products pokec hollywood; do timeout 600 python scripts/probe_pagerank.py $g 2>&1 | grep -v amdgpu.ids | tail -2; done | tee gpurun_out/r05_pagerank_chained.txt
