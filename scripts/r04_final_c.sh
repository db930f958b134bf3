# round-4 final batch, part C: the SpMSpV sweeps again (Python mirror and C++ module layer) on the last code -- after the fold's
# ballot-scan trim and the one-launch kernel's bucket sort -- and the sweep under rocprofv3 for the per-kernel averages
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py tests/test_cpp_layer.py -x -q 2>&1 | tail -2
timeout 2400 python benchmarks/bench_spmspv.py --out gpurun_out/r04_spmspv_sweep.jsonl > gpurun_out/r04_spmspv_sweep.log 2>&1; tail -1 gpurun_out/r04_spmspv_sweep.log | cut -c1-200
timeout 1500 python benchmarks/run_spmspv_cpp.py --out gpurun_out/r04_spmspv_sweep_cpp.jsonl 2>&1 | grep "^#"
python - <<'PY'
import json
for f in ("r04_spmspv_sweep", "r04_spmspv_sweep_cpp"):
    rows = [json.loads(l) for l in open("gpurun_out/%s.jsonl" % f)]
    small = sorted(r["ms"] for r in rows if r["vector_sparsity"] >= 0.999)
    print(f, len(rows), "cases, verified", sum(r["verified"] for r in rows), ">= 99.9 %%: median %.4f ms, min %.4f, max %.4f" % (small[len(small) // 2], small[0], small[-1]))
    for r in rows:
        if r["vector_sparsity"] in (0.99,) and r["semiring"] == "Arithmetic": print("   ", r["graph"], r["vector_sparsity"], r["ms"])
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sweep_trace
timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sweep_trace -- python $GRAFT_REPO_ROOT/benchmarks/bench_spmspv.py --out $GRAFT_REPO_ROOT/gpurun_out/r04_spmspv_sweep_under_rocprof.jsonl > /tmp/sweep_trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/sweep_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 "$f" > gpurun_out/r04_spmspv_sweep_kernel_stats.csv; grep "gl::" "$f" >> gpurun_out/r04_spmspv_sweep_kernel_stats.csv; }
grep "spmspv_" gpurun_out/r04_spmspv_sweep_kernel_stats.csv | cut -c1-160
