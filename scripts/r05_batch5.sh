# round 5, batch 5: the C++ app drivers of include/graphlily/app (device-resident BFS from C++, world-of-one gl_dist_*) against
# the oracle; the reference's UNMODIFIED bench drivers on the orkut stand-in, with this repo's app headers and with the checkout's
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_cpp_layer.py tests/test_gpu_configs.py -m gpu -x -q -k "cpp_app or reference_benchmark or reference_app or own_bar or module_layer_parity" 2>&1 | grep -v "amdgpu.ids" | tail -25
cp gpurun_out/fullsize_margins.jsonl gpurun_out/r05_fullsize_margins_own_bar.jsonl 2>/dev/null
timeout 1500 python benchmarks/run_reference_benches.py --graph orkut --apps spmv,spmv_verify,bfs,pagerank,sssp 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_reference_benches_on_hip.txt
cat gpurun_out/r05_reference_benches_on_hip.txt
timeout 1500 python benchmarks/run_reference_benches.py --graph orkut --apps bfs_refapps,pagerank_refapps,sssp_refapps 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_reference_benches_on_hip_refapps.txt
cat gpurun_out/r05_reference_benches_on_hip_refapps.txt
