#!/bin/bash
# SQ counters of the small-linear weight-gradient kernel (stand-alone, tools/debug/wgrad_time.py)
OUT=$(realpath -m ${1:-gpurun_out/r05_wgrad_pmc}); ROOT=$(pwd)
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
[ -n "$2" ] && export MONOSOWA_POINTWISE_LIB=$ROOT/tools/debug/variants/pw_$2.so
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/debug/wgrad_time.py 8800 256 256 > $OUT/p$i.log 2>&1
done
python3 - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "p*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "linear_wgrad" not in k and "Cijk_Ailk_Bljk" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print("   %-34s %14.0f per launch" % (c, acc[k][c] / max(1, n[k][c])))
PY
rm -rf $OUT/p[0-9]
cat $OUT/summary.txt
