cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python benchmarks/bench_spmspv.py --graphs googleplus,pokec,hollywood --semirings Arithmetic,Logical --out gpurun_out/r02_spmspv_sweep_a.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-420
