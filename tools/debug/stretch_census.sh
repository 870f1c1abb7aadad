#!/bin/bash
# which kernels run between the start of the backward (criterion backward) and the first encoder-layer backward kernel --
# the stretch of the step that is bound by the host's enqueue rate
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stretch
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --preheat-seconds 1 --no-cpu-baseline > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/t/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last step: from the last matched_bwd / focal_bwd kernel back to ... forward to the first scatter_rows
ends = [i for i, n in enumerate(names) if "scatter_rows_kernel" in n]
starts = [i for i, n in enumerate(names) if "focal_bwd_kernel" in n or "matched_bwd_kernel" in n]
s0 = [s for s in starts if s < ends[-1]][-1]
while s0 - 1 in starts: s0 -= 1
e0 = [e for e in ends if e > s0][0]
seg = rows[s0:e0]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("stretch: %d launches, span %.2f ms, kernel time %.2f ms (under the profiler)" % (len(seg), (t1 - t0) / 1e6, busy / 1e6))
acc = collections.Counter(); tm = collections.Counter()
for r in seg:
    k = r["Kernel_Name"].split("(")[0][:80]
    acc[k] += 1; tm[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, n in acc.most_common(40):
    print("%4d x %8.1f us  %s" % (n, tm[k] / 1e3, k))
PY
