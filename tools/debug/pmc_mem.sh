#!/bin/bash
# usage: tools/debug/pmc_mem.sh <tag>: vector-memory path counters (TA / TCP / TCC) of the MSDA kernels on the encoder-shape micro-benchmark
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm_$tag
mkdir -p $OUT
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  echo "pass $i: $set" >> $OUT/progress.log
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- python3 $GRAFT_REPO_ROOT/tools/msda_fused_bench.py --kinds enc --iters 3 --warmup 1 > $OUT/s$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/s*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "msda" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print("$tag", k[:60])
    for c, v in sorted(cs.items()):
        print("    %-44s %.4g" % (c, sum(v) / len(v)))
PY
