cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GRAPHLILY_SPMV_FUSE=0
timeout 900 python scripts/r02_spmv_ab.py --graphs pokec --variants "base;BLOCKS=256,SEGMENTS=1;HOT_FLOOR=1,BLOCKS=256,SEGMENTS=1;HOT_FLOOR=2,BLOCKS=256,SEGMENTS=1;COMPACT=0;COMPACT=0,BLOCKS=256,SEGMENTS=1;HOT_FLOOR=1,BLOCKS=256,SEGMENTS=1,MIX=9;HOT_FLOOR=1,BLOCKS=256,SEGMENTS=1,MIX=6" --out gpurun_out/r02_ab_pokec.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-400
timeout 900 python scripts/r02_spmv_ab.py --graphs googleplus,ogbl_ppa --variants "base;COMPACT=0;HOT_FLOOR=1;HOT_FLOOR=1,COMPACT=0;HOT=0,COMPACT=0" --out gpurun_out/r02_ab_gplus.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-400
for g in googleplus pokec; do timeout 300 python scripts/r02_clocks.py $g 2>&1 | grep -v amdgpu.ids; done
timeout 300 python scripts/r02_clocks.py pokec HOT_FLOOR=1,BLOCKS=256,SEGMENTS=1 2>&1 | grep -v amdgpu.ids
