#!/bin/bash
# scratch: hardware-counter passes over the SpMV kernel (separate rocprofv3 --pmc runs, no tracing)
R=${GRAFT_REPO_ROOT:-/root/repo}
G=${1:-ogbn_products}
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 240 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_$i -- python $R/scripts/probe_spmv.py --graph $G --ops 0 --iters 3 > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "spmv_rbcs_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    v = v[1:] if len(v) > 2 else v
    print("%-40s %16.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
done
