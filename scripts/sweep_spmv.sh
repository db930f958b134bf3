#!/bin/bash
# scratch: sweep the (row blocks x segments) decomposition of the SpMV plan on one stand-in
G=${1:-ogbn_products}
for SHAPE in "0 0" "256 1" "512 1" "160 2" "150 4" "150 5" "150 7" "128 2" "128 4" "1024 1"; do
  set -- $SHAPE
  echo "== blocks $1 segments $2 (0 = planner)"
  GRAPHLILY_SPMV_BLOCKS=$1 GRAPHLILY_SPMV_SEGMENTS=$2 timeout 300 python scripts/probe_spmv.py --graph $G --ops 0 --iters 30 2>&1 | grep -E "op 0 mask|plan create"
done
