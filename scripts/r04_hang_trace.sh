#!/bin/bash
# where does `bench.py --force-dist --cabi-comm` stop with the one-GPU RCCL probe on?  (python stack after 90 s)
export GRAPHLILY_DEBUG=dist_self_probe=1
timeout 150 python -X faulthandler -c "
import faulthandler, sys, runpy
faulthandler.dump_traceback_later(90, exit=True)
sys.argv = ['bench.py', '--gpus', '1', '--force-dist', '--cabi-comm', '--no-six-graphs', '--no-spmspv', '--steps', '10', '--warmup', '2', '--scale', '0.1', '--bfs-runs', '1', '--no-cpu-baseline', '--no-pattern']
runpy.run_path('bench.py', run_name='__main__')
" 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -40
