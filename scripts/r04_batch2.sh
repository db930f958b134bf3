# round 4, second batch: GPU suite, bench line with the new spmspv / six_graphs objects, same-box A/B of the schedules
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
( time timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04_bench_try.json 2> gpurun_out/r04_bench_try.err ) 2>&1 | grep real
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_try.json"):
    if l.startswith('{"metric'):
        d = json.loads(l)
        print({k: d[k] for k in ("value", "ms_per_step", "frac_hbm_peak")}, d["roofline"]["frac"])
        print("spmspv", json.dumps(d.get("spmspv"))[:900])
        sg = d.get("six_graphs", {})
        print("six_graphs seconds", sg.get("_seconds"), "skipped", sg.get("_skipped_for_time"), sg.get("error"))
        for k, v in sg.items():
            if k.startswith("_"): continue
            print(k, v.get("seconds"), "spmv", v["spmv"]["ms"], v["spmv"]["frac_hbm_peak"], v["spmv"]["kernel_frac_hbm_peak"], v["spmv"]["ok"], "pat", v["spmv_pattern"]["ms"],
                  "bfs", v.get("bfs", {}).get("pull_ms"), v.get("bfs", {}).get("pull_push_ms"), v.get("bfs", {}).get("ok"),
                  "pr", v.get("pagerank", {}).get("ms_per_iter"), "sssp", v.get("sssp", {}).get("pull_ms"), v.get("sssp", {}).get("pull_push_ms"), v.get("sssp", {}).get("ok"))
PY
tail -3 gpurun_out/r04_bench_try.err
timeout 1200 python scripts/r04_ab_schedules.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_ab_schedules.txt
python - <<'PY'
import json
for l in open("gpurun_out/r04_ab_schedules.txt"):
    try: r = json.loads(l)
    except Exception: print(l[:300]); continue
    print(r["graph"], {k: (v.get("pull_push_ms", v.get("ms")), v.get("schedule_gpu_ms")) for k, v in r["bfs"].items()})
    print("   sssp", {k: (v.get("pull_push_ms", v.get("ms")), v.get("pushes")) for k, v in r["sssp"].items()})
PY
