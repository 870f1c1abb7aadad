#!/bin/bash
# times the fused operator pair at the decoder shape (Lq = 550) with each measurement build of the library (tools/debug/variants/*.so)
cd $GRAFT_REPO_ROOT
echo "full: $(python tools/msda_fused_bench.py --kinds 550 --iters 40 2>&1 | grep 'msda_bwd' | tr '\n' ' ')"
for f in tools/debug/variants/*.so; do
  echo "$(basename $f): $(MONOSOWA_MSDA_LIB=$PWD/$f timeout -k 5 100 python tools/msda_fused_bench.py --kinds 550 --iters 40 2>&1 | grep 'msda_bwd' | tr '\n' ' ')"
done
