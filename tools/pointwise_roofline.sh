#!/bin/bash
# Per-kernel HBM roofline of this repo's pointwise / norm / reduction library (mono::*) inside the train step:
#   bytes moved (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 on gfx950 as MI355X_MICROARCH.md
#   prescribes for wide streams) / average duration (the kernel trace of the same runs) / 8 TB/s, 15 largest kernels by time.
# Run on the GPU box from the repo root:   bash tools/pointwise_roofline.sh gpurun_out/pw_roofline
set -e
OUT=$(realpath -m "${1:-gpurun_out/pw_roofline}")
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  echo "pass $C" >> "$OUT/progress.log"
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preheat-seconds 0 --no-cpu-baseline --no-inference-leg --no-dataloader-leg --no-step-roofline --no-offsets-probe > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"
done
python3 "$ROOT/tools/pointwise_roofline.py" "$OUT" > "$OUT/pointwise_roofline.txt"
cat "$OUT/pointwise_roofline.txt"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
