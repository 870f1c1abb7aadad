#!/bin/bash
# usage: tools/debug/sweep_env.sh <out dir> <ENV NAME> <v1> <v2> ...   -- encoder-shape micro-benchmark under each value of one MSDA_* variable,
# then a rocprofv3 kernel trace (per-kernel durations) for every value
out=$1; name=$2; shift 2
mkdir -p $out
for v in "$@"; do
  env $name=$v python tools/msda_fused_bench.py --kinds enc --iters 40 > $out/${name}_$v.log 2>&1
  echo "$name=$v: $(grep -E 'msda_(fwd|bwd)' $out/${name}_$v.log | sed 's/offsets=init: //' | tr '\n' ' ')"
done
