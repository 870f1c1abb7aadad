#!/bin/bash
# per-kernel times of the encoder-shape operator pair for the named builds under tools/debug/variants/ (rocprofv3 kernel trace),
# on train-step-like offsets: bash tools/debug/r05_variant_kernels.sh "prio0 prio1 prio2" [offsets] [reps]
LIBS=${1:-"prio0 prio1"}
OFF=${2:-init:0.05}
REPS=${3:-2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05_variants
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for tag in $LIBS; do
    export MONOSOWA_MSDA_LIB=$ROOT/tools/debug/variants/$tag.so
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$tag -- python3 $ROOT/tools/msda_fused_bench.py --kinds enc --iters 30 --warmup 5 --offsets $OFF > $OUT/$tag.log 2>&1
    python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t_$tag/*/*kernel_stats.csv")[0]
row = []
for r in csv.DictReader(open(f)):
    for k in ("scatter_rows", "gather_win_kernel<true", "gather_win_kernel<false", "plan_fused"):
        if k in r["Name"]: row.append("%s %.1f" % (k.split("_kernel")[0], float(r["AverageNs"]) / 1e3))
print("$tag rep $rep:", "  ".join(sorted(row)), "us")
PY
    rm -rf $OUT/t_$tag
  done
done
