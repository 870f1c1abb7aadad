#!/bin/bash
# round 5: the epilogue-GEMM bottleneck paths against the MIOpen + pointwise-pass paths, alternating steps in one process
for v in 1 3; do
  echo "== CONV1X1_EPILOGUE 0 vs $v"
  python tools/ab_step.py monosowa_amd.monodetr.backbone.CONV1X1_EPILOGUE 0 $v --steps 60 2>&1 | tail -3
done
