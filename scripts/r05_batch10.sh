# round 5, batch 10: cold : hot elements per step on the large graphs with the delta-coded cold stream (elements of 4 / 8 groups)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for g in orkut ogbn_products hollywood orkut_community_shuffled; do for f in 4 0; do for mix in 1 2 3 4; do
echo -n "$g flags=$f mix=$mix: "; GRAPHLILY_DEBUG=spmv_mix=$mix timeout 300 python scripts/probe_spmv.py --graph $g --flags $f --no-copy --iters 100 2>&1 | grep "^op 0 mask 0"
done; done; done | tee gpurun_out/r05_mix_sweep_delta_cold.txt
GRAPHS="pokec ogbl_ppa" bash scripts/ab_variants.sh r04 r05a cur 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_ab_delta_cold_small_graphs.txt
