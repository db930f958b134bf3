cd $GRAFT_REPO_ROOT
for g in pokec ogbl_ppa googleplus; do timeout 300 python scripts/r02_clocks.py $g base gpurun_out/clocks_$g.txt 2>&1 | grep -v amdgpu.ids | grep -v "by xcc\|by se\|by cu\|corr("; done
