#!/bin/bash
# per-kernel times of the fused operator pair (encoder shape) for every tools/debug/variants/*.so, under rocprofv3
cd /tmp && export TMPDIR=/tmp
for f in $GRAFT_REPO_ROOT/tools/debug/variants/*.so; do
  tag=$(basename $f .so)
  export MONOSOWA_MSDA_LIB=$f
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pv_$tag -- python3 $GRAFT_REPO_ROOT/tools/msda_fused_bench.py --kinds enc --iters 30 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/pv_$tag.log 2>&1
  grep "msda_" $GRAFT_REPO_ROOT/gpurun_out/pv_$tag.log | sed "s/^/$tag /"
  python3 - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pv_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    if "msda" in r["Name"]: print("$tag %-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
