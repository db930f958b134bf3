# round-4 final measurement batch, part B: SpMSpV sweep (plain and under rocprofv3), per-kernel breakdown of blocking calls,
# the reference's own benchmark drivers + their C-ABI call timelines, one RCCL rank with the exchange inside the graph,
# two ranks sharing the GPU over gloo, the standalone six-graph file
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
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
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sweep_trace
timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sweep_trace -- python $GRAFT_REPO_ROOT/benchmarks/bench_spmspv.py --out $GRAFT_REPO_ROOT/gpurun_out/r04_spmspv_sweep_under_rocprof.jsonl > /tmp/sweep_trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/sweep_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 "$f" > gpurun_out/r04_spmspv_sweep_kernel_stats.csv; grep "gl::" "$f" >> gpurun_out/r04_spmspv_sweep_kernel_stats.csv; }
bash scripts/r04_spmspv_trace.sh "hollywood 0.99" "hollywood 0.9995" "ogbn_products 0.99" "googleplus 0.9999" > gpurun_out/r04_spmspv_call_breakdown.txt 2>&1; tail -12 gpurun_out/r04_spmspv_call_breakdown.txt
timeout 600 python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank,sssp 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_reference_benches_on_hip.txt
GRAPHLILY_MODULE_FUSION=0 timeout 600 python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank,sssp 2>&1 | grep -v amdgpu.ids | sed 's/^/[GRAPHLILY_MODULE_FUSION=0] /' | tee -a gpurun_out/r04_reference_benches_on_hip.txt
bash scripts/r03_refbench_timeline.sh orkut > gpurun_out/r04_refbench_timeline.log 2>&1
cp gpurun_out/api_bfs.txt gpurun_out/r04_api_timeline_bench_bfs.txt; cp gpurun_out/api_pagerank.txt gpurun_out/r04_api_timeline_bench_pagerank.txt
bash scripts/r04_refbench_sssp_timeline.sh orkut > gpurun_out/r04_refbench_sssp_timeline.log 2>&1; cp gpurun_out/api_sssp.txt gpurun_out/r04_api_timeline_bench_sssp.txt
timeout 600 python bench.py --gpus 1 --force-dist --cabi-comm --no-cpu-baseline --no-six-graphs --no-spmspv --no-pattern > gpurun_out/r04_bench_one_rccl_rank_cabi.json 2> gpurun_out/r04_bench_cabi.err; echo "cabi rc=$?"
timeout 600 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-six-graphs --no-spmspv --no-pattern > gpurun_out/r04_bench_one_rccl_rank_torch.json 2>> gpurun_out/r04_bench_cabi.err; echo "torch comm rc=$?"
timeout 600 python bench.py --gpus 2 --backend gloo --same-gpu --steps 20 --no-cpu-baseline --no-six-graphs --no-spmspv > gpurun_out/r04_bench_2ranks_same_gpu.json 2> gpurun_out/r04_bench_2ranks.err; echo "2-rank rc=$?"
python - <<'PY'
import json
for f in ("r04_bench_one_rccl_rank_cabi", "r04_bench_one_rccl_rank_torch", "r04_bench_2ranks_same_gpu"):
    try:
        for l in open("gpurun_out/%s.json" % f):
            if l.startswith('{"metric'):
                d = json.loads(l); b = d.get("bfs", {})
                print(f, d["value"], {k: b[k]["ms"] for k in ("pull_push", "pull") if k in b}, b.get("schedule"), b.get("exchange"), b.get("error"))
    except Exception as e: print(f, e)
PY
timeout 900 python benchmarks/bench_graphs.py --out gpurun_out/r04_six_graphs.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-260
