#!/bin/bash
# SQ / TCP / TCC counters for the MSDA kernels (micro-benchmark, encoder shape).  bash tools/pmc_fwd.sh <outdir> [MSDA_GATHER]
OUT=$(realpath -m "${1:-gpurun_out/pmc_fwd}"); ROOT=$(pwd); export MSDA_GATHER=${2:-2}
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/set$i" -- python "$ROOT/tools/msda_kernel_bench.py" --iters 2 --warmup 1 --kinds enc > "$OUT/set$i.log" 2>&1 || echo "set $i failed: $(tail -2 $OUT/set$i.log | cut -c1-200)"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "set*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "msda" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print("   %-36s %.4g" % (c, sum(v) / len(v)))
PY
