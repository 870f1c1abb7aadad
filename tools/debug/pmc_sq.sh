#!/bin/bash
# usage: tools/debug/pmc_sq.sh <tag> [ENV=val ...]: SQ counters of the MSDA kernels on the encoder-shape micro-benchmark
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/msda_fused_bench.py --kinds enc --iters 3 --warmup 1 --offsets ${MSDA_BENCH_OFFSETS:-init} > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/msda_fused_bench.py --kinds enc --iters 3 --warmup 1 --offsets ${MSDA_BENCH_OFFSETS:-init} > $OUT/b.log 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
for sub in ("a", "b"):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % sub)
    if not fs:
        print("no counters for", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "msda" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print("$tag", k[:60], " ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
rm -rf $OUT/a $OUT/b
