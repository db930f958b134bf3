cd $GRAFT_REPO_ROOT
for g in orkut ogbn_products hollywood; do timeout 300 python scripts/r02_clocks.py $g 2>&1 | grep -v amdgpu.ids; done
