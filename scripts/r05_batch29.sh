#!/bin/bash
# round 5, batch 29: how long should the device idle between plan creation and the warm-up steps of the driver's short command?
cd /root/repo; mkdir -p gpurun_out
for rep in 1 2; do for s in 0.5 0.0 0.02 0.1 2.0; do
echo -n "settle=$s steps20: "; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --settle $s --no-six-graphs --no-bfs --no-spmspv --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][0]
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done; done | tee gpurun_out/r05_settle_sweep.txt
