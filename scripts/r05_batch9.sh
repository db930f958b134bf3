# round 5, batch 9: hot-table size sweep on the graphs whose table is capped (nnz < 64 M: 2048 columns) and on the thin-headed
# community stand-ins -- hot entries cost 6.19 / 2.19 B now, cold ones 7 / 3 B + a gather
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for g in pokec ogbl_ppa googleplus orkut_community orkut; do for f in 4 0; do for h in 1 2048 4096 8192 16384 32768; do
echo -n "$g flags=$f spmv_hot=$h: "; GRAPHLILY_DEBUG=spmv_hot=$h timeout 300 python scripts/probe_spmv.py --graph $g --flags $f --no-copy --iters 100 2>&1 | grep "^op 0 mask 0\|^plan create" | sed 's/plan create.*hot_columns.: \([0-9]*\).*hot_nnz.: \([0-9]*\).*/[hot cols \1 nnz \2]/' | tr '\n' ' '; echo
done; done; done | tee gpurun_out/r05_hot_table_sweep.txt
for g in pokec ogbl_ppa; do for f in 4 0; do for mix in 1 2 3; do
echo -n "$g flags=$f mix=$mix: "; GRAPHLILY_DEBUG=spmv_mix=$mix timeout 300 python scripts/probe_spmv.py --graph $g --flags $f --no-copy --iters 100 2>&1 | grep "^op 0 mask 0"
done; done; done | tee gpurun_out/r05_mix_sweep_small_graphs.txt
