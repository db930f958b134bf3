cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python scripts/r02_spmv_ab.py --graphs hollywood --variants "base;BALANCE=0" --reps 3 2>&1 | grep -v amdgpu.ids | cut -c1-140; done
timeout 600 python benchmarks/bench_graphs.py --graphs hollywood 2>&1 | grep -v amdgpu.ids | cut -c1-200
