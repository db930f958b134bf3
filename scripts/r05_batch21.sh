#!/bin/bash
# round 5, batch 21: the no-helper-launch mode (self-hot: no packed vector, the prologue gathers the hot table from x) on graphs
# above its 16 M nnz bound, general layout
cd /root/repo; mkdir -p gpurun_out
GRAPHS="pokec ogbl_ppa hollywood ogbn_products" FLAGS="4" bash scripts/ab_variants.sh cur cur=GRAPHLILY_DEBUG=spmv_helper=2,spmv_hot=4096 cur=GRAPHLILY_DEBUG=spmv_helper=2,spmv_hot=2048 cur=GRAPHLILY_DEBUG=spmv_helper=2,spmv_hot=1024 2>&1 | tee gpurun_out/r05_self_hot_sweep.txt
