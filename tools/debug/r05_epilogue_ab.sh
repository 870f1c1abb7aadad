#!/bin/bash
# round 5: A/B of the epilogue-GEMM bottleneck paths, alternating steps in one process (tools/ab_step.py)
for pair in "monosowa_amd.monodetr.backbone.CONV1X1_EPILOGUE 0 1" "monosowa_amd.monodetr.backbone.CONV1X1_EPILOGUE 0 3"; do
  echo "== $pair"
  python tools/ab_step.py $pair --steps 60 2>&1 | tail -3
done
