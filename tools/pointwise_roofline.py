"""Summarises the two rocprofv3 --pmc passes of tools/pointwise_roofline.sh: for every mono:: kernel (this repo's pointwise /
norm / reduction library) bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; the factor 2 is the gfx950 correction for
wide coalesced reads, MI355X_MICROARCH.md "HBM"), average duration from the kernel traces of the same runs, fraction of 8 TB/s."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
PEAK = 8.0e12


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "mono::"):
        n = n.replace(p, "")
    return n[:56]


pmc = {}
dur = defaultdict(list)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(d, C, "*", "*counter_collection.csv"))
    acc = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "mono::" in r["Kernel_Name"] and r["Counter_Name"] == C:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    pmc[C] = acc
    t = glob.glob(os.path.join(d, C, "*", "*kernel_trace.csv"))
    for r in csv.DictReader(open(t[0])):
        if "mono::" in r["Kernel_Name"]:
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
rows = []
for k, ds in dur.items():
    n = len(ds) / 2.0                       # two passes
    avg = sum(ds) / len(ds)
    fetch = 2.0 * sum(pmc["FETCH_SIZE"].get(k, [0])) / max(1, len(pmc["FETCH_SIZE"].get(k, [0])))
    write = sum(pmc["WRITE_SIZE"].get(k, [0])) / max(1, len(pmc["WRITE_SIZE"].get(k, [0])))
    rows.append((avg * n, k, n, avg, fetch, write))
rows.sort(reverse=True)
steps = 4.0                                 # 1 warm-up + 3 timed steps per pass (plus the first, untimed one: launches / 5)
print("# mono:: kernels inside the train step (bench.py, B = 16): bytes per launch from PMC counters (2 x FETCH_SIZE + WRITE_SIZE),")
print("# average duration from the kernel trace of the same runs; profiled runs clock ~3 % lower than unprofiled ones.")
print("%-58s %9s %9s %10s %10s %8s %7s" % ("kernel", "launches", "avg us", "fetch MB", "write MB", "GB/s", "of 8TB/s"))
tot_t = 0.0
for total, k, n, avg, fetch, write in rows[:24]:
    gbps = (fetch + write) / avg / 1e9
    print("%-58s %9d %9.1f %10.2f %10.2f %8.0f %6.1f%%" % (k, n, avg * 1e6, fetch / 1e6, write / 1e6, gbps, 100 * (fetch + write) / avg / PEAK))
    tot_t += total
# the hand-written DENSE kernels among them (frozen layer1, conv1x1_fused.hip): also against the f32 matrix peak.  Pixel counts of the
# default bench workload (B = 16, 1280 x 384: layer1 at 96 x 320)
M_PIX = 16 * 96 * 320
DENSE = {"conv1x1_head_kernel<256>": 2.0 * M_PIX * 256 * 64, "conv1x1_head_kernel<64>": 2.0 * M_PIX * 64 * 64,
         "conv1x1_tail_kernel": 2.0 * M_PIX * 64 * 256, "conv1x1_tail_ds_kernel": 2.0 * M_PIX * 128 * 256,
         # the small linears' weight gradients (small_wgrad.hip): 53 launches per step of different shapes (49 x [8800, 256] x [8800, 256],
         # 4 x [30720, 512 | 256] x [30720, 256 | 512]); the step's total as bench.py's roofline.step counts it / 53
         "linear_wgrad_partial_kernel": 101953044480.0 / 53}
print("# dense kernels of this library: FLOP per launch (default bench shape) / average duration / 157.3 TFLOP/s (f32 matrix peak)")
for total, k, n, avg, fetch, write in rows:
    if k in DENSE:
        print("%-58s %9d %9.1f us %8.1f GFLOP %7.1f TFLOP/s %6.1f%% of the f32 matrix peak; %5.1f%% of 8 TB/s"
              % (k, n, avg * 1e6, DENSE[k] / 1e9, DENSE[k] / avg / 1e12, 100 * DENSE[k] / avg / 157.3e12, 100 * (fetch + write) / avg / PEAK))
print("# the 24 kernels above: %.2f ms of kernel time over the profiled steps; all mono:: kernels: %.2f ms over %d launches"
      % (tot_t * 1e3, sum(r[0] for r in rows) * 1e3, sum(r[2] for r in rows)))
