#!/bin/bash
# Phase budget of the cell scatter (-DMSDA_ROWS_SKIP builds, msda_scatter_rows.hip) on offsets like the train step's (init:0.05) and on
# the micro-benchmark's (init = init:0.3): which part of the kernel follows the candidates scanned and which the points delivered.
# Build first (CPU):  for v in 0 1 3 19 35 64 128; do tools/debug/build_variant.sh skip$v -DMSDA_ROWS_SKIP=$v; done
OUT=$(realpath -m ${1:-gpurun_out/r05_rows_phases})
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for off in init:0.05 init; do
  for f in $ROOT/tools/debug/variants/skip*.so; do
    tag=$(basename $f .so)
    export MONOSOWA_MSDA_LIB=$f
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$tag -- python3 $ROOT/tools/msda_fused_bench.py --kinds enc --iters 20 --warmup 5 --offsets $off > $OUT/$tag.log 2>&1
    python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t_$tag/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "scatter_rows" in r["Name"]: print("$off %-8s scatter_rows avg %8.1f us" % ("$tag", float(r["AverageNs"]) / 1e3))
PY
    rm -rf $OUT/t_$tag
  done
done
