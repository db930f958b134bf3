# round 4, third batch: rocprofv3 passes (orkut bench line; ogbn-products SpMV legs), the SpMSpV sweep with its kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o build/ubench_gather scripts/ubench_gather.hip 2>/dev/null
timeout 1500 bash scripts/profile_bench.sh > gpurun_out/r04_profile.log 2>&1; tail -2 gpurun_out/r04_profile.log
GRAPH=ogbn_products timeout 1200 bash scripts/profile_bench.sh > gpurun_out/r04_profile_products.log 2>&1; tail -2 gpurun_out/r04_profile_products.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sweep_trace
timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sweep_trace -- python $GRAFT_REPO_ROOT/benchmarks/bench_spmspv.py --out $GRAFT_REPO_ROOT/gpurun_out/r04_spmspv_sweep_under_rocprof.jsonl > /tmp/sweep_trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/sweep_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 "$f" > gpurun_out/r04_spmspv_sweep_kernel_stats.csv; grep "gl::" "$f" >> gpurun_out/r04_spmspv_sweep_kernel_stats.csv; }
timeout 2400 python benchmarks/bench_spmspv.py --out gpurun_out/r04_spmspv_sweep.jsonl > gpurun_out/r04_spmspv_sweep.log 2>&1; tail -2 gpurun_out/r04_spmspv_sweep.log | cut -c1-200
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r04_spmspv_sweep.jsonl")]
print(len(rows), "cases, verified", sum(r["verified"] for r in rows))
small = sorted(r["ms"] for r in rows if r["vector_sparsity"] >= 0.999)
print(">= 99.9 %%: median %.4f ms, min %.4f, max %.4f" % (small[len(small) // 2], small[0], small[-1]))
for r in rows:
    if r["vector_sparsity"] in (0.9, 0.99) and r["semiring"] == "Arithmetic": print(r["graph"], r["vector_sparsity"], r["ms"], r["gbps"], r["direction"])
PY
