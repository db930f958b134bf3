# round 5, batch 12: would column segments help the thin-headed / short-row graphs?  blocks x segments forced through the planner
# knobs (general layout, flags=4; pattern, flags=0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for g in orkut_community_shuffled orkut_community pokec orkut; do for f in 4 0; do for shape in "0 0" "256 2" "128 2" "128 4" "512 1"; do set -- $shape
echo -n "$g flags=$f blocks=$1 segments=$2: "; GRAPHLILY_DEBUG=spmv_blocks=$1,spmv_segments=$2 timeout 300 python scripts/probe_spmv.py --graph $g --flags $f --no-copy --iters 100 2>&1 | grep "^op 0 mask 0\|^plan create" | sed "s/plan create.*'num_units': \([0-9]*\), 'blocks': \([0-9]*\), 'segments': \([0-9]*\).*/[units \1 blocks \2 seg \3]/" | tr '\n' ' '; echo
done; done; done | tee gpurun_out/r05_shape_sweep.txt
# ... and where a step's time goes: scratch builds with the gather / the accumulates / the prefix scans switched off (results are
# wrong by construction; only the times matter)
GRAPHS="orkut pokec hollywood" bash scripts/ab_variants.sh cur abl_gather abl_acc abl_scan abl_acc_gather 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_ablation.txt
