cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/r02_spmv_ab.py --graphs googleplus,pokec,orkut --variants "FUSE=0;FUSE=1;FUSE=2;FUSE=3;FUSE=0,BLOCKS=256,SEGMENTS=1;FUSE=2,BLOCKS=256,SEGMENTS=1;FUSE=3,BLOCKS=256,SEGMENTS=1" --out gpurun_out/r02_ab_fuse.jsonl 2>&1 | grep -v amdgpu.ids
