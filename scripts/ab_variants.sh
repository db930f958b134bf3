#!/bin/bash
# scratch: A/B several builds of the library on ONE box -- the general kernel's time differs by 4-9 % between boxes of
# the pool, more than most kernel changes.  Variants are scripts/_variants/<name>.so (scripts/build_variant.sh) or "cur"
# (the in-tree build); an optional =VAR=VALUE sets an environment variable for that run.
# usage (on the GPU box): bash scripts/ab_variants.sh head cur cur=GRAPHLILY_SPMV_COMPACT=0
#   GRAPHS="orkut ogbn_products" SHARD=0/1 FLAGS="4 0" (4 = general layout with streamed values, 0 = pattern) override the defaults
for rep in 1 2; do for g in ${GRAPHS:-orkut ogbn_products}; do for f in ${FLAGS:-4 0}; do
for v in "$@"; do name=${v%%=*}; kv=""; [ "$v" != "$name" ] && kv=${v#*=}
lib=""; [ "$name" != cur ] && lib=scripts/_variants/$name.so
echo -n "$g ${SHARD:-0/1} flags=$f $v: "
env GRAPHLILY_HIP_LIB=$lib $kv python scripts/probe_spmv.py --graph $g --shard ${SHARD:-0/1} --flags $f --no-copy --iters 100 2>&1 | grep "^op 0 mask 0"
done; done; done; done
