#!/bin/bash
# round 6, the measurement batch on ONE box (usage: bash scripts/r06_final.sh [legs], default all):
#   tests    full GPU suite
#   bench    the driver's bench command + the auto-sized line
#   prof     rocprofv3 --kernel-trace --stats + the FETCH_SIZE / WRITE_SIZE passes (orkut, ogbn-products) -> gpurun_out/prof*
#   six      standalone six-graphs leg
#   spmspv   the SpMSpV sweep through the C++ module layer and through Python
#   ref      the reference's unmodified bench drivers (both app layers), bench_spmspv.cpp, and its own test suites
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
LEGS=${*:-tests bench prof six spmspv ref}
for leg in $LEGS; do case $leg in
tests)
  rm -f gpurun_out/fullsize_margins.jsonl
  timeout 3000 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r06_gputests.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/r06_gputests.log
  grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r06_gputests.log | tail -14
  cp gpurun_out/fullsize_margins.jsonl gpurun_out/r06_fullsize_margins.jsonl 2>/dev/null
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
bench)
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_orkut_n1_steps20_warmup5.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
  timeout 900 python bench.py > gpurun_out/r06_bench_orkut_n1.json 2>> gpurun_out/r06_bench.err; echo "bench rc=$?"
  python - <<'PY'
import json
for f in ("r06_bench_orkut_n1_steps20_warmup5", "r06_bench_orkut_n1"):
    d = [json.loads(l) for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][0]
    print(f, json.dumps(d["headline"]))
PY
  ;;
prof)
  bash scripts/profile_bench.sh > gpurun_out/r06_profile_bench.log 2>&1; tail -3 gpurun_out/r06_profile_bench.log
  GRAPH=ogbn_products bash scripts/profile_bench.sh > gpurun_out/r06_profile_bench_products.log 2>&1; tail -2 gpurun_out/r06_profile_bench_products.log ;;
six)
  timeout 900 python benchmarks/bench_graphs.py --out gpurun_out/r06_six_graphs.jsonl 2>&1 | grep -v amdgpu.ids | tail -3 ;;
spmspv)
  timeout 1500 python benchmarks/run_spmspv_cpp.py --out gpurun_out/r06_spmspv_sweep_cpp.jsonl 2>&1 | tail -1
  timeout 1500 python benchmarks/bench_spmspv.py --out gpurun_out/r06_spmspv_sweep.jsonl 2>&1 | tail -1 ;;
ref)
  timeout 1500 python benchmarks/run_reference_benches.py --graph orkut --apps spmv,spmv_verify,bfs,pagerank,sssp 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r06_reference_benches_on_hip.txt
  timeout 1500 python benchmarks/run_reference_benches.py --graph orkut --apps bfs_refapps,pagerank_refapps,sssp_refapps 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r06_reference_benches_on_hip_refapps.txt
  grep "average_time\|iteration\|passed" gpurun_out/r06_reference_benches_on_hip.txt gpurun_out/r06_reference_benches_on_hip_refapps.txt
  timeout 1800 python benchmarks/run_reference_benches.py --apps spmspv --write-reference-dataset-dir 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r06_reference_bench_spmspv.txt; tail -3 gpurun_out/r06_reference_bench_spmspv.txt
  timeout 1500 python benchmarks/run_reference_benches.py --apps tests --write-reference-dataset-dir 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r06_reference_test_suites.txt; tail -1 gpurun_out/r06_reference_test_suites.txt ;;
esac; done
