#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for g in orkut ogbn_products pokec; do timeout 600 python scripts/probe_pagerank.py $g 2>&1 | grep -v amdgpu.ids | tail -1; done | tee gpurun_out/r05_pagerank_readback.txt
timeout 900 python -m pytest tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -2
