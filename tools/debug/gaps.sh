#!/bin/bash
# GPU idle time inside a train step: gaps between consecutive kernels of the last traced step, by position in the step
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --preheat-seconds 1 --no-cpu-baseline --no-inference-leg --no-dataloader-leg > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# steps end with the AdamW kernel
ends = [i for i, n in enumerate(names) if "adamw_kernel" in n]
# two adamw launches per step (two param groups): take pairs
step_ends = ends[1::2] if len(ends) >= 4 else ends
a, b = step_ends[-2] + 1, step_ends[-1] + 1
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("last step: %d kernels, span %.2f ms, busy %.2f ms, idle %.2f ms (under the profiler)" % (len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
# idle by tenth of the step
bins = [0.0] * 10
big = []
for p, q in zip(seg, seg[1:]):
    gap = int(q["Start_Timestamp"]) - int(p["End_Timestamp"])
    if gap > 0:
        pos = (int(p["End_Timestamp"]) - t0) / (t1 - t0)
        bins[min(9, int(pos * 10))] += gap
        if gap > 30000: big.append((gap / 1e3, pos, p["Kernel_Name"][:50], q["Kernel_Name"][:50]))
print("idle ms by tenth of the step:", " ".join("%.2f" % (x / 1e6) for x in bins))
for g in sorted(big, reverse=True)[:12]:
    print("  gap %7.1f us at %.2f of the step: after %s -> before %s" % g)
PY
