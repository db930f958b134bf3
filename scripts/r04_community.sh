# round 4: the layout on a stand-in WITH planted communities (vertices numbered by community) and on the same graph relabelled
# at random; general / pattern SpMV and BFS via bench.py; the offline row-clustering estimate runs on the CPU (scripts/r03_row_clustering_offline.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for g in orkut_community orkut_community_shuffled orkut; do
  timeout 600 python bench.py --graph $g --no-six-graphs --no-spmspv --no-cpu-baseline > gpurun_out/r04_bench_$g.json 2> gpurun_out/r04_bench_$g.err
  python - "$g" <<'PY'
import json, sys
g = sys.argv[1]
for l in open("gpurun_out/r04_bench_%s.json" % g):
    if l.startswith('{"metric'):
        d = json.loads(l)
        print(g, "n", d["config"]["n"], "nnz", d["config"]["nnz"], "general ms", d["ms_per_step"], "frac", d["frac_hbm_peak"], "kernel frac", d["roofline"]["frac"],
              "pattern ms", d.get("pattern_plan", {}).get("ms_per_step"), "pattern frac", d.get("pattern_plan", {}).get("frac_hbm_peak"),
              "bfs pull_push", d.get("bfs", {}).get("pull_push", {}).get("ms"), "pull", d.get("bfs", {}).get("pull", {}).get("ms"), "ok", d["selfcheck_ok"])
PY
done
