#!/bin/bash
# HBM traffic of the MSDA kernels from PMC counters, as MI355X_MICROARCH.md prescribes: separate --pmc passes (FETCH_SIZE
# takes 3 TCC slots, WRITE_SIZE 2), no tracing domains besides the kernel trace, plus the calibration kernels of
# tools/ubench/pmc_calib.hip.  Collected on the FUSED operator the train step runs (tools/msda_fused_bench.py, encoder
# and 550-query decoder shapes at B = 16, the module's initial offset pattern).  Run on the GPU box from the repo root:
#     bash tools/collect_pmc.sh gpurun_out/pmc
set -e
OUT=$(realpath -m "${1:-gpurun_out/pmc}")
ROOT=$(pwd)
mkdir -p "$OUT"
[ -x "$ROOT/tools/ubench/pmc_calib" ] || hipcc -O2 --offload-arch=gfx950 -o "$ROOT/tools/ubench/pmc_calib" "$ROOT/tools/ubench/pmc_calib.hip"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/calib_$C" -- "$ROOT/tools/ubench/pmc_calib" > "$OUT/calib_$C.log" 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/msda_$C" -- python3 "$ROOT/tools/msda_fused_bench.py" --iters 3 --warmup 1 --kinds enc,550 > "$OUT/msda_$C.log" 2>&1
  # the other BASELINE geometries (encoder shape only): config 4 = 1408x376 at B = 16 (S = 11044), config 5 = 1920x1280 at B = 4 (S = 51000)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/c4_$C" -- python3 "$ROOT/tools/msda_fused_bench.py" --iters 3 --warmup 1 --kinds enc --resolution 1408x376 > "$OUT/c4_$C.log" 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/c5_$C" -- python3 "$ROOT/tools/msda_fused_bench.py" --iters 3 --warmup 1 --kinds enc --resolution 1920x1280 --batch 4 > "$OUT/c5_$C.log" 2>&1
done
python3 "$ROOT/tools/pmc_summarize.py" "$OUT"
# the raw counter tables are tens of MB each (gpurun copies at most 64 MiB back): keep the summary, the logs and the JSON
for d in "$OUT"/calib_* "$OUT"/msda_* "$OUT"/c4_* "$OUT"/c5_*; do [ -d "$d" ] && rm -rf "$d"; done
