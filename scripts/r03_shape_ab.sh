# same-box A/B of the planner's shape model (round 3 re-fit of stream_rate): the library before the change vs the tree,
# whole stand-ins and the row shards of orkut, general (flags 4) and pattern (flags 0) layouts
export GRAPHS="googleplus ogbl_ppa hollywood pokec ogbn_products orkut"
bash scripts/ab_variants.sh r03_old_shape cur 2>&1 | grep -v amdgpu
for sh in 0/2 1/4 3/8; do GRAPHS=orkut SHARD=$sh bash scripts/ab_variants.sh r03_old_shape cur 2>&1 | grep -v amdgpu; done
