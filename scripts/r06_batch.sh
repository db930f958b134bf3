#!/bin/bash
# round 6: the round's gpurun batches, one case per experiment (EXPERIMENTS.md R6.x).  Outputs land in gpurun_out/r06_<case>.txt;
# the ones that are cited get copied to profiles/.
# usage (through gpurun): bash scripts/r06_batch.sh <case> [args]
cd /root/repo; mkdir -p gpurun_out
CASE=${1:-help}; shift || true
probe() {   # probe <graph> <flags> <GRAPHLILY_DEBUG value> [lib]: one line "graph flags knobs: <op 0 mask 0 line> | plan info"
  local g=$1 f=$2 k=$3 lib=${4:-}
  echo -n "$g flags=$f [$k] ${lib:+lib=$lib }: "
  env GRAPHLILY_DEBUG="$k" GRAPHLILY_HIP_LIB=${lib:+scripts/_variants/$lib.so} timeout 600 python scripts/probe_spmv.py --graph $g --flags $f --no-copy --iters 100 2>&1 \
    | grep -E "^op 0 mask 0|^plan create" | sed -e 's/^plan create [0-9.]*s //' | tr '\n' ' '; echo
}
case $CASE in
two_wg)   # R6.1: two 1024-thread workgroups per CU on half-height tiles (LDS <= 80 KB each), by planner knobs alone
  for rep in 1 2; do for g in ${GRAPHS:-orkut ogbn_products pokec}; do for f in 0 4; do
    probe $g $f ""
    probe $g $f "spmv_blocks=512"
    probe $g $f "spmv_blocks=512,spmv_hot=5120"
    probe $g $f "spmv_blocks=512,spmv_hot=3072"
    probe $g $f "spmv_blocks=768,spmv_hot=5120"
  done; done; done 2>&1 | tee gpurun_out/r06_two_wg.txt
  ;;
pmc_pattern)   # R6.2: where the pattern kernel's wave cycles go (SQ / TA / TCP / LDS counters, separate passes)
  bash scripts/pmc_probe.sh ${1:-orkut} 2>&1 | tee gpurun_out/r06_pmc_pattern_${1:-orkut}.txt
  ;;
lds_atomic)    # R6.3: what an LDS atomic / read instruction costs by op, active lanes and address pattern (scripts/ubench_lds_atomic.hip)
  timeout 300 build/ubench_lds_atomic 2>&1 | tee gpurun_out/r06_ubench_lds_atomic.txt
  ;;
ref_tests)     # the reference's own acceptance suites, unmodified, float + ufixed
  timeout 1500 python benchmarks/run_reference_benches.py --apps tests --write-reference-dataset-dir 2>&1 | tee gpurun_out/r06_reference_test_suites.txt
  ;;
pytest)        # pytest -m gpu on the files / -k expression given
  timeout 3000 python -m pytest -m gpu -x -q "$@" 2>&1 | tail -15 | tee gpurun_out/r06_pytest_last.txt
  ;;
rare)          # R6.6 what-if (needs the `spmv_ablate_rare` knob of commit "what-if: rare columns", since removed): the pattern kernel
               # without the entries of the rarest columns (wrong results: how much do their lines cost?)
  for rep in 1 2; do for g in ${GRAPHS:-orkut ogbn_products pokec}; do for d in 0 1 2 3 5 8; do
    probe $g 0 "spmv_ablate_rare=$d"
  done; done; done 2>&1 | tee gpurun_out/r06_whatif_rare_columns.txt
  ;;
hot_floor)     # R6.7: degree floor of the hot table (entries per row block a column must average), pattern layout with the row-packed stream
  for rep in 1 2; do for g in ${GRAPHS:-pokec ogbl_ppa googleplus hollywood ogbn_products orkut}; do for d in 4 2 1; do
    probe $g 0 "spmv_hot_floor=$d"
  done; done; done 2>&1 | tee gpurun_out/r06_hot_floor_sweep.txt
  ;;
spmspv_priv)   # R6.10: private per-workgroup regions behind the bins (no global reservations for large runs): parity with the path
               # forced on small matrices, then blocking / back-to-back call times against the round's first build and with the knob off
  GRAPHLILY_DEBUG="spmspv_priv_min=1,spmspv_priv_min_nnz=0" timeout 1500 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -3
  timeout 1500 python -m pytest tests/test_gpu_spmspv.py -m gpu -x -q 2>&1 | tail -2
  for rep in 1 2; do for g in ${GRAPHS:-orkut hollywood ogbn_products}; do for sp in 0.9 0.95 0.99; do
    for v in "r6base:" "cur:spmspv_priv=0" "cur:"; do lib=${v%%:*}; k=${v#*:}; [ "$lib" = cur ] && lib="" || lib=scripts/_variants/$lib.so
      echo -n "$g $sp [$v] "; GRAPHLILY_HIP_LIB=$lib GRAPHLILY_DEBUG="$k" timeout 600 python scripts/spmspv_call_trace.py $g $sp 2>&1 | grep "blocking run\|back to back" | tr '\n' ' '; echo
    done; done; done; done 2>&1 | tee gpurun_out/r06_ab_spmspv_private_regions.txt
  ;;
bool3)         # R6.11: the boolean layout's stream delta-coded to 3 bytes per entry (knob bool_compress: 0 off, 1 on, 2 on + report)
  timeout 2400 python -m pytest -m gpu -x -q tests/test_gpu_format.py tests/test_gpu_bfs_sharded.py tests/test_gpu_apps.py tests/test_gpu_configs.py tests/test_gpu_spmv.py -k "bool or bfs or Logical or format or config" 2>&1 | tail -5
  for g in ${GRAPHS:-orkut ogbn_products pokec hollywood ogbl_ppa googleplus}; do
    GRAPHLILY_DEBUG="bool_compress=2" timeout 600 python scripts/probe_spmv.py --graph $g --flags 2 --ops 1 --no-copy --iters 5 2>&1 | grep "bool_plan_compress" | sed "s/^/$g: /"
  done
  for rep in 1 2; do for g in ${GRAPHS:-orkut ogbn_products pokec hollywood ogbl_ppa googleplus}; do for k in 0 1; do for d in 0.5 0.02; do
    echo -n "$g bool_compress=$k density $d: "
    GRAPHLILY_DEBUG="bool_compress=$k" timeout 600 python scripts/probe_spmv.py --graph $g --flags 2 --ops 1 --no-copy --iters 100 --density $d 2>&1 | grep -E "^op 1 mask" | tr '\n' ' '; echo
  done; done; done; done
  for rep in 1 2; do for k in 0 1; do
    GRAPHLILY_DEBUG="bool_compress=$k" timeout 900 python benchmarks/bench_graphs.py --graphs ${BGRAPHS:-orkut,pokec} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('bool_compress=$k', r['graph'], {k: v for k, v in r.items() if 'bfs' in k})"
  done; done
  ;;
bool3b)        # R6.11: ring depth / 10-bit decoder on the delta-coded stream: variants = lib[:knobs] (cur = in-tree build)
  timeout 1200 python -m pytest -m gpu -x -q tests/test_gpu_format.py -k "boolean" 2>&1 | tail -3
  for g in ${GRAPHS:-orkut ogbn_products pokec hollywood}; do
    GRAPHLILY_DEBUG="bool_compress=2" timeout 600 python scripts/probe_spmv.py --graph $g --flags 2 --ops 1 --no-copy --iters 5 2>&1 | grep "bool_plan_compress" | sed "s/^/$g: /"
  done
  for rep in 1 2; do for g in ${GRAPHS:-orkut ogbn_products pokec hollywood}; do for v in "$@"; do for d in 0.5 0.02; do
    lib=${v%%:*}; k=""; [ "$v" != "$lib" ] && k=${v#*:}; [ "$lib" = cur ] && lib="" || lib=scripts/_variants/$lib.so
    echo -n "$g [$v] density $d: "
    GRAPHLILY_HIP_LIB=$lib GRAPHLILY_DEBUG="$k" timeout 600 python scripts/probe_spmv.py --graph $g --flags 2 --ops 1 --no-copy --iters 100 --density $d 2>&1 | grep -E "^op 1 mask" | sed -e 's/GTEPS.*//' | tr '\n' ' '; echo
  done; done; done; done
  ;;
ab)            # same-box A/B: GRAPHS / FLAGS as in scripts/ab_variants.sh; arguments = variants (scripts/_variants/<name>.so, or cur[=KNOBS])
  bash scripts/ab_variants.sh "$@" 2>&1 | tee gpurun_out/r06_ab_${AB_NAME:-last}.txt
  ;;
*) echo "cases: two_wg pmc_pattern lds_atomic ref_tests pytest rare hot_floor spmspv_priv bool3 bool3b ab"; exit 1;;
esac
