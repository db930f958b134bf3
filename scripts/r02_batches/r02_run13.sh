cd $GRAFT_REPO_ROOT
export GRAPHLILY_HIP_LIB=scripts/_variants/dbg.so
for D in 0 1 2 3 7; do
echo "== dbg=$D (bit0 no gather, bit1 no accumulate, bit2 no hot lookup) general"
GRAPHLILY_SPMV_DBG=$D timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,orkut --variants "base" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-100
echo "== dbg=$D pattern"
GRAPHLILY_SPMV_DBG=$D timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,orkut --flags 0 --variants "base" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-100
done
