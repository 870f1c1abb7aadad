#!/bin/bash
# usage: tools/debug/prof_enc_kernels.sh <out dir> [ENV=val ...]: per-kernel durations of the encoder-shape fused pair (rocprofv3 --kernel-trace --stats)
out=$(realpath -m $1); shift
for kv in "$@"; do export "$kv"; done
mkdir -p $out
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $ROOT/tools/msda_fused_bench.py --kinds enc --iters 20 --warmup 5 > $out/run.log 2>&1
f=$(ls $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "msda" in r["Name"]:
        print("%-70s calls %4s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
