cd $GRAFT_REPO_ROOT
timeout 600 python scripts/r02_spmv_ab.py --graphs googleplus --variants "base;BLOCKS=512,SEGMENTS=1;BLOCKS=512,SEGMENTS=1,HOT_FLOOR=2;BLOCKS=384,SEGMENTS=1" --reps 3 2>&1 | grep -v amdgpu.ids | cut -c1-260
timeout 600 python scripts/r02_spmv_ab.py --graphs ogbl_ppa,pokec --variants "base;BLOCKS=512,SEGMENTS=1;BLOCKS=512,SEGMENTS=1,HOT=8192;BLOCKS=256,SEGMENTS=2" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-260
