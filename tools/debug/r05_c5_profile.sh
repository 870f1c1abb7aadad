#!/bin/bash
# config 5 (1920x1280, B = 4, S = 51000): per-kernel times of the encoder-shape operator pair and the scatter's scan census
OUT=$(realpath -m ${1:-gpurun_out/r05_c5})
ROOT=$(pwd)
mkdir -p $OUT
MONOSOWA_MSDA_LIB=tools/debug/variants/count.so python tools/debug/scan_census.py --micro init --resolution 1920x1280 --batch 4 > $OUT/census_c5.log 2>&1
MONOSOWA_MSDA_LIB=tools/debug/variants/count.so python tools/debug/scan_census.py --micro init > $OUT/census_kitti.log 2>&1
grep "per launch" $OUT/census_c5.log $OUT/census_kitti.log
cd /tmp && export TMPDIR=/tmp
for cfg in "c5 --resolution 1920x1280 --batch 4" "kitti --batch 16"; do
  set -- $cfg; tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $ROOT/tools/msda_fused_bench.py --kinds enc --iters 20 --warmup 5 "$@" > $OUT/fused_$tag.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    if "msda" in r["Name"]: print("$tag %-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/trace_$tag
done
