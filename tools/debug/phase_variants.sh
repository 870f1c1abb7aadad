#!/bin/bash
# times the fused forward at the encoder shape with each phase-skip build of the library (tools/debug/variants/msda_skip*.so)
cd $GRAFT_REPO_ROOT
echo "full: $(python tools/msda_fused_bench.py --kinds enc --iters 40 2>&1 | grep msda_fwd)"
for f in tools/debug/variants/msda_skip*.so; do
  echo "$(basename $f): $(MONOSOWA_MSDA_LIB=$PWD/$f timeout -k 5 100 python tools/msda_fused_bench.py --kinds enc --iters 40 2>&1 | grep 'msda_fwd')"
done
