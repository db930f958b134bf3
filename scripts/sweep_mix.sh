# scratch: cold:hot mix sweep of the general (wide) and pattern (quad) kernels on one box
for g in ${GRAPHS:-orkut ogbn_products hollywood}; do
for m in ${MIXES:-5 6 7 3 9 10 11}; do echo -n "$g general mix=$m: "; GRAPHLILY_SPMV_MIX=$m python scripts/probe_spmv.py --graph $g --flags 4 --no-copy --iters 100 2>&1 | grep "^op 0 mask 0"; done
for m in ${PMIXES:-2 1 6 9}; do echo -n "$g pattern mix=$m: "; GRAPHLILY_SPMV_MIX=$m python scripts/probe_spmv.py --graph $g --flags 0 --no-copy --iters 100 2>&1 | grep "^op 0 mask 0"; done
done
