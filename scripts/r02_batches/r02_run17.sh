cd $GRAFT_REPO_ROOT
timeout 600 python scripts/r02_spmv_ab.py --graphs googleplus --variants "base;MIX=2;MIX=8;MIX=1;MIX=6;MIX=9;HOT_FLOOR=1;HOT_FLOOR=1,MIX=2;HOT_FLOOR=1,MIX=8;HOT_FLOOR=2,MIX=2" --reps 3 2>&1 | grep -v amdgpu.ids | cut -c1-330
timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,ogbl_ppa --variants "base;MIX=2;MIX=5;MIX=9" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-330
