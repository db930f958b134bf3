# full GPU suite + bench line (round 2, after re-entry)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_margins.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02_gputests_4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_gputests_4.log
tail -30 gpurun_out/r02_gputests_4.log
timeout 600 python bench.py --steps 100 --warmup 5 > gpurun_out/r02_bench_4.json 2> gpurun_out/r02_bench_4.err
echo "bench rc=$?"
tail -c 2500 gpurun_out/r02_bench_4.json
tail -5 gpurun_out/r02_bench_4.err
