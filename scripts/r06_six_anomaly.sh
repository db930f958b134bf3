#!/bin/bash
# round 6 scratch: why did bench.py's six_graphs leg read the orkut pattern plan at 0.243 ms when every other leg reads 0.18-0.19?
cd /root/repo
for variant in "" "--no-bfs --no-spmspv --no-pattern" "--no-spmspv" "--no-bfs"; do
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $variant > /tmp/b.json 2>/dev/null
  python3 - "$variant" <<'PY'
import json, sys
d=[json.loads(l) for l in open('/tmp/b.json') if l.startswith('{')][0]
s=d.get('six_graphs',{})
print("variant [%s]: pattern_plan %s | six orkut general %s pattern %s pagerank %s | products pattern %s" % (sys.argv[1], d.get('pattern_plan',{}).get('ms_per_step'),
      s.get('orkut',{}).get('spmv',{}).get('ms'), s.get('orkut',{}).get('spmv_pattern',{}).get('ms'), s.get('orkut',{}).get('pagerank',{}).get('ms_per_iter'),
      s.get('ogbn_products',{}).get('spmv_pattern',{}).get('ms')))
PY
done 2>&1 | tee gpurun_out/r06_six_anomaly.txt
