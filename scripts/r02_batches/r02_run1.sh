set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_margins.jsonl
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02_gputests_1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_gputests_1.log
tail -40 gpurun_out/r02_gputests_1.log
python benchmarks/run_reference_benches.py --graph orkut > gpurun_out/r02_reference_benches_orkut.txt 2>&1
tail -30 gpurun_out/r02_reference_benches_orkut.txt
python bench.py --steps 100 --warmup 5 > gpurun_out/r02_bench_1.json 2> gpurun_out/r02_bench_1.err
tail -c 3000 gpurun_out/r02_bench_1.json
