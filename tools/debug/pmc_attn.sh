#!/bin/bash
# usage: tools/debug/pmc_attn.sh <tag>: SQ counters of the attention kernels at the train step's shapes (two rocprofv3 --pmc passes)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/debug/attn_bwd_time.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/debug/attn_bwd_time.py > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % sub)
    if not fs:
        print("no counters for", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "attn" in k: acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        print("$tag", k[0][:40], "grid", k[1], " ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
