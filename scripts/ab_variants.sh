# scratch: A/B several builds of the library on ONE box (box-to-box variation exceeds most kernel changes)
# usage: bash scripts/ab_variants.sh head:1 cur:0 cur:1   (name = scripts/_variants/<name>.so, cur = the in-tree build; :0/:1 = tickets)
for rep in 1 2; do for g in ${GRAPHS:-orkut ogbn_products}; do for f in 4 0; do
for v in "$@"; do name=${v%%:*}; t=${v##*:}
lib=scripts/_variants/$name.so; [ $name = cur ] && lib=graphlily_amd/lib/libgraphlily_hip.so
echo -n "$g flags=$f $name tickets=$t: "; GRAPHLILY_SPMV_TICKETS=$t python scripts/probe_spmv.py --lib $lib --graph $g --flags $f --no-copy --iters 100 2>&1 | grep "^op 0 mask 0"; done; done; done; done
