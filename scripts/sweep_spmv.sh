#!/bin/bash
# scratch: sweep tile size x unroll for the SpMV kernel on one stand-in
G=${1:-ogbn_products}
for U in 2 4 8; do
  for T in 512 2048 8192 32768; do
    echo "== unroll $U tile $T"
    GRAPHLILY_SPMV_UNROLL=$U GRAPHLILY_SPMV_TILE_NNZ=$T timeout 300 python scripts/probe_spmv.py --graph $G --ops 0 --iters 30 2>&1 | grep -E "op 0 mask 0|plan create"
  done
done
