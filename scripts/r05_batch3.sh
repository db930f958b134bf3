# round 5, batch 3: run-coded hot stream + branch-free accumulates (padding entries name a dummy slot) -- format / SpMV / SpMSpV
# tests, smoke; same-box A/B against round 4's build on six graphs, general (flags=4) and pattern (flags=0) layouts; mixes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_format.py tests/test_gpu_spmv.py tests/test_gpu_spmspv.py tests/test_gpu_typed.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -15
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
GRAPHS="orkut ogbn_products pokec hollywood ogbl_ppa googleplus" bash scripts/ab_variants.sh r04 cur 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_ab_runcoded_branchfree.txt
for mix in 1 2 3 4; do for g in orkut ogbn_products; do for f in 4 0; do echo -n "$g flags=$f mix=$mix: "; GRAPHLILY_DEBUG=spmv_mix=$mix python scripts/probe_spmv.py --graph $g --flags $f --no-copy --iters 100 2>&1 | grep "^op 0 mask 0"; done; done; done | tee gpurun_out/r05_mix_sweep.txt
