#!/bin/bash
# MFMA utilisation of the dense kernels (GEMM / conv) in a train step: rocprofv3 --pmc on a short bench run,
# after a warm run (MIOpen find-db).  bash tools/pmc_mfma.sh gpurun_out/pmc_mfma
OUT=$(realpath -m "${1:-gpurun_out/pmc_mfma}"); ROOT=$(pwd); mkdir -p "$OUT"
python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-inference-leg --no-dataloader-leg > /dev/null 2> "$OUT/warm.err"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/run" -- python "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-inference-leg --no-dataloader-leg > "$OUT/run.log" 2>&1 || tail -3 "$OUT/run.log"
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "run", "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
rows = []
for k, c in acc.items():
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0: continue
    # SQ_VALU_MFMA_BUSY_CYCLES sums busy cycles over SIMDs (4 per CU, 256 CUs); GRBM_GUI_ACTIVE sums the 8 XCDs
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    rows.append((c["GRBM_GUI_ACTIVE"], util, n[k], k))
tot = sum(r[0] for r in rows)
print("kernels with MFMA activity, by GPU-active cycles; MfmaUtil = MFMA busy cycles / (active cycles x 1024 SIMDs)")
for g, u, cnt, k in sorted(rows, reverse=True)[:14]:
    print("  %5.1f%% of MFMA-kernel time  MfmaUtil %5.1f%%  x%-4d %s" % (100 * g / tot, 100 * u, cnt, k))
w = sum(r[0] * r[1] for r in rows) / tot
print("time-weighted MfmaUtil over these kernels: %.1f%%" % (100 * w))
PY
rm -rf "$OUT/run"
