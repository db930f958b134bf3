# round 3, one-launch shard slots: emulated ranks (bench + rocprofv3 timelines), A/B against the three-launch slot
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --emulate-rank 0/8,3/8,7/8,1/4,0/2 --no-cpu-baseline > gpurun_out/r03b_bench_emulated.json 2> gpurun_out/r03b_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
for l in open('gpurun_out/r03b_bench_emulated.json'):
    if l.startswith('{"metric'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['frac'], d.get('bfs'))
        for e in d.get('bfs_emulated_ranks', []):
            print(e)
PY
GRAPHLILY_BFS_SHARD_STEP=0 timeout 900 python bench.py --emulate-rank 0/8,1/4,0/2 --no-cpu-baseline --no-pattern > gpurun_out/r03b_bench_emulated_three_launch.json 2>> gpurun_out/r03b_bench.err
python - <<'PY'
import json
for l in open('gpurun_out/r03b_bench_emulated_three_launch.json'):
    if l.startswith('{"metric'):
        d = json.loads(l)
        for e in d.get('bfs_emulated_ranks', []):
            print('three-launch', e)
PY
for spec in "0/8 pull_push" "3/8 pull_push" "0/8 pull" "1/4 pull_push" "0/2 pull_push"; do set -- $spec
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/emu_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/emu_trace -- python $GRAFT_REPO_ROOT/scripts/r03_emulate_trace.py orkut $1 $2 > /tmp/emu_trace.log 2>&1
  cd $GRAFT_REPO_ROOT; n=$(echo $1 | tr / of); { grep "^CALL 4" /tmp/emu_trace.log; python scripts/r03_emulate_trace_summary.py /tmp/emu_trace; } > gpurun_out/r03b_emulated_rank${n}_$2_timeline.txt; tail -1 gpurun_out/r03b_emulated_rank${n}_$2_timeline.txt
done
timeout 900 python -m pytest tests/test_cpp_layer.py tests/test_gpu_apps.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -5
