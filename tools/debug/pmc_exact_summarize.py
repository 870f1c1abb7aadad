"""Per-launch averages of the counters tools/debug/pmc_exact_bytes.sh collected, per kernel (and grid), calibration kernels first."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "msda::" not in k and "calib" not in k:
            continue
        acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for (k, grid), cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rd = m.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0.0) * 32
    wr = (m.get("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 0.0) + m.get("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 0.0)) * 32
    hist = "rd requests %.3g (32 B %.3g, 64 B %.3g, 128 B %.3g)" % (m.get("TCC_EA0_RDREQ_sum", 0), m.get("TCC_EA0_RDREQ_32B_sum", 0),
                                                                   m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0))
    line = "%-62s grid=%-9s read %8.1f MB  write %8.1f MB (atomic part %.1f MB)  %s" % (
        k[-62:], grid, rd / 1e6, wr / 1e6, m.get("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 0.0) * 32 / 1e6, hist)
    print(line)
    out.append(line)
open(os.path.join(d, "summary.txt"), "w").write("\n".join(out) + "\n")
