#!/bin/bash
# round 5, batch 32: row blocks re-cut from measured unit durations at plan creation -- parity tests, then same-box A/B (knob off / on)
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_format.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
GRAPHS="orkut ogbn_products hollywood pokec ogbl_ppa orkut_community_shuffled" bash scripts/ab_variants.sh cur=GRAPHLILY_DEBUG=spmv_tune=0 cur=GRAPHLILY_DEBUG=spmv_tune_verbose=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ab_tuned_blocks.txt
for g in orkut pokec; do GRAPHLILY_DEBUG=spmv_tune_verbose=1 timeout 300 python scripts/probe_spmv.py --graph $g --flags 4 --no-copy --iters 100 2>&1 | grep "graphlily\|plan create" | cut -c1-250; done | tee -a gpurun_out/r05_ab_tuned_blocks.txt
