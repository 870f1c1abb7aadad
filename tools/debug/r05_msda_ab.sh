#!/bin/bash
# round 5: variants of the MSDA library against the round-4 build, same box, alternating.
#   base    = HEAD of round 4 (tools/debug/variants/base.so)
#   new     = the tree's library (monosowa_amd/lib)
#   others  = tools/debug/variants/<name>.so (tools/debug/build_variant.sh <name> -D...)
# usage (GPU box, repo root): bash tools/debug/r05_msda_ab.sh gpurun_out/<dir> "base new pb1 ..." [steps|nosteps] [sweep spec] [c5]
OUT=${1:-gpurun_out/r05_msda_ab}
LIBS=${2:-"base new"}
STEPS=${3:-steps}
SWEEP=${4:-init,normal:2}
mkdir -p $OUT
V=tools/debug/variants
setlib() { if [ $1 = new ]; then unset MONOSOWA_MSDA_LIB; else export MONOSOWA_MSDA_LIB=$V/$1.so; fi; }
for rep in 1 2; do
  for lib in $LIBS; do
    setlib $lib
    python tools/msda_fused_bench.py --sweep $SWEEP --iters 30 --out $OUT/sweep_${lib}_$rep.json > $OUT/sweep_${lib}_$rep.log 2>&1
    sed "s/^/$lib $rep: /" $OUT/sweep_${lib}_$rep.log | grep fwd
    if [ "$5" = c5 ]; then
      python tools/msda_fused_bench.py --kinds enc --resolution 1920x1280 --batch 4 --iters 30 > $OUT/c5_${lib}_$rep.log 2>&1
      sed "s/^/$lib $rep c5: /" $OUT/c5_${lib}_$rep.log | grep fused
    fi
  done
done
[ $STEPS = steps ] || exit 0
for rep in 1 2; do
  for lib in $LIBS; do
    setlib $lib
    python bench.py --steps 20 --no-cpu-baseline --no-inference-leg --no-dataloader-leg --no-step-roofline --no-offsets-probe > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err
    python - <<PY
import json
d = json.load(open("$OUT/bench_${lib}_$rep.json"))
r = d["roofline"]
print("$lib $rep: %.2f ms/step  msda_bwd %.4f ms  fwd %.4f ms  all msda %.3f ms/step" % (d["ms_per_step"], r["avg_launch_ms"], r["forward_same_shape"]["avg_launch_ms"], r["all_msda_aggregate"]["ms_per_step"]))
PY
  done
done
