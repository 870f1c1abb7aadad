#!/bin/bash
# small-linear weight-gradient kernel: parity tests, stand-alone timing of the build variants, alternating-step A/B in the train step
OUT=${1:-gpurun_out/r05_wgrad}
mkdir -p $OUT
python -m pytest tests/test_linear_wgrad_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
python tools/debug/wgrad_time.py > $OUT/time_default.log 2>&1; cat $OUT/time_default.log
python tools/ab_step.py monosowa_amd.token_linear.SMALL_WGRAD_KERNEL 0 1 --steps 80 > $OUT/ab.log 2>&1; tail -8 $OUT/ab.log
