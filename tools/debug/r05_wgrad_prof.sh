#!/bin/bash
# kernel-level durations of the small-linear weight-gradient kernels per build variant (rocprofv3 kernel trace of tools/debug/wgrad_time.py)
OUT=$(realpath -m ${1:-gpurun_out/r05_wgrad_prof}); ROOT=$(pwd)
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in ${2:?names of builds under tools/debug/variants (pw_<name>.so, hipcc -DMONO_WGRAD_WORKGROUPS=...)}; do
  export MONOSOWA_POINTWISE_LIB=$ROOT/tools/debug/variants/pw_$v.so
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 $ROOT/tools/debug/wgrad_time.py ${3:-8800 256 256} > $OUT/$v.log 2>&1
  f=$(find $OUT/$v -name '*kernel_stats.csv' | head -1)
  echo "== $v" >> $OUT/summary.txt
  python3 - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("linear_wgrad", "Cijk", "colsum", "partial_sum")):
        print("  %-60s calls %6s  avg %8.2f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/$v
done
cat $OUT/summary.txt
