#!/bin/bash
# scratch: sweep unroll for the SpMV kernel on one stand-in
G=${1:-ogbn_products}
for U in 2 4 8; do
  echo "== unroll $U"
  GRAPHLILY_SPMV_UNROLL=$U timeout 300 python scripts/probe_spmv.py --graph $G --ops 0,2 --iters 30 2>&1 | grep -E "op . mask 0|plan create"
done
