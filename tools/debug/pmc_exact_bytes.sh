#!/bin/bash
# HBM-side read / write BYTES of the MSDA kernels from the size-resolved L2 <-> fabric request counters of gfx950
# (TCC_EA0_RDREQ_DRAM_32B: "1 64-byte request will be counted to 2, 128-byte as 4" -- no request-size assumption, unlike FETCH_SIZE =
# requests x 64 B), next to the request-size histogram, with the calibration kernels of tools/ubench/pmc_calib.hip (known byte counts)
# measured the same way.  A cross-check of tools/collect_pmc.sh's corrected FETCH_SIZE, separate --pmc passes, kernel trace only.
#     bash tools/debug/pmc_exact_bytes.sh gpurun_out/pmc_exact
set -e
OUT=$(realpath -m "${1:-gpurun_out/pmc_exact}")
ROOT=$(pwd)
mkdir -p "$OUT"
[ -x "$ROOT/tools/ubench/pmc_calib" ] || hipcc -O2 --offload-arch=gfx950 -o "$ROOT/tools/ubench/pmc_calib" "$ROOT/tools/ubench/pmc_calib.hip"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
         "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum TCC_EA0_WRREQ_sum"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/calib_$i" -- "$ROOT/tools/ubench/pmc_calib" > "$OUT/calib_$i.log" 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/msda_$i" -- python3 "$ROOT/tools/msda_fused_bench.py" --iters 3 --warmup 1 --kinds enc,550 > "$OUT/msda_$i.log" 2>&1
done
python3 "$ROOT/tools/debug/pmc_exact_summarize.py" "$OUT"
