# round 5, batch 1: the reference's own bench_spmv.cpp (unmodified) + its verify on the orkut and ogbn-products stand-ins;
# hot-column shares of the stand-ins (input to the run-coded hot layout)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python scripts/hot_share.py googleplus ogbl_ppa hollywood pokec ogbn_products orkut community community_shuffled > gpurun_out/r05_hot_share.txt 2>&1
cat gpurun_out/r05_hot_share.txt | grep -v "amdgpu.ids"
for g in orkut ogbn_products; do
  timeout 1500 python benchmarks/run_reference_benches.py --graph $g --apps spmv,spmv_verify 2>&1 | grep -v "amdgpu.ids"
done > gpurun_out/r05_reference_bench_spmv.txt
cat gpurun_out/r05_reference_bench_spmv.txt
