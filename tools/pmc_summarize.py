"""Turns the rocprofv3 --pmc CSVs of tools/collect_pmc.sh into per-launch HBM bytes for the MSDA kernels,
corrected with the calibration kernels (known byte counts), and writes <dir>/msda_traffic.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def counters(pattern):
    out = defaultdict(list)
    files = sorted(glob.glob(os.path.join(d, pattern, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:          # newest run only (gpurun merges successive runs into the same directory)
        for r in csv.DictReader(open(f)):
            out[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r))
    return out


calib = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for (k, c), v in counters("calib_" + C).items():
        calib[(k, c)] = sum(x[1] for x in v) / len(v)
known = {"calib_stream_f4": {"FETCH_SIZE": (64 << 20) * 16 + (64 << 20) * 4, "WRITE_SIZE": (64 << 20) * 4},
         "calib_rowgather": {"FETCH_SIZE": (2 << 20) * 128 + (2 << 20) * 4},
         "calib_seg32": {"FETCH_SIZE": (2 << 20) * 32 + (2 << 20) * 4}}
factor = {}
for k, cs in known.items():
    for c, nbytes in cs.items():
        raw = calib.get((k, c))
        if raw:
            factor[(k, c)] = nbytes / (raw * 1024.0)     # counter unit: KiB
print("calibration: true bytes / (counter * 1024):", {"%s/%s" % k: round(v, 3) for k, v in factor.items()})

res = {"calibration": {"%s/%s" % k: v for k, v in factor.items()}, "unit": "bytes per launch", "kernels": {}}
msda = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for (k, c), v in counters("msda_" + C).items():
        if "msda::" not in k:
            continue
        grids = defaultdict(list)
        for _, val, r in v:
            grids[(r["Grid_Size"], r.get("LDS_Block_Size", ""))].append(val)
        for g, vals in grids.items():
            msda.setdefault((k, g[0]), {})[c] = sum(vals) / len(vals) * 1024.0
f_gather = factor.get(("calib_rowgather", "FETCH_SIZE"), 1.0)
f_write = factor.get(("calib_stream_f4", "WRITE_SIZE"), 1.0)
for (k, grid), cs in sorted(msda.items()):
    fetch = cs.get("FETCH_SIZE", 0.0) * f_gather
    write = cs.get("WRITE_SIZE", 0.0) * f_write
    res["kernels"]["%s grid=%s" % (k, grid)] = {"fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write,
                                                "raw_fetch_counter_bytes": cs.get("FETCH_SIZE", 0.0),
                                                "raw_write_counter_bytes": cs.get("WRITE_SIZE", 0.0)}
    print("%-70s fetch %8.1f MB  write %8.1f MB" % (k + " grid=" + grid, fetch / 1e6, write / 1e6))
# op-level sums in the keys bench.py looks up.  The fused micro-benchmark runs the encoder shape (Lq = S: tile-window gather
# + cell scatter behind the device-side directional plan) and the 550-query decoder shape (record gather + row-band scatter); a kernel template that serves both
# shapes is told apart by its grid (the encoder launch is the larger one).
by_kernel = defaultdict(list)
for (k, grid), cs in msda.items():
    by_kernel[k].append((int(grid), k, grid))
ops = {
    "msda_fwd_Lq10200_B16": [("gather_win_kernel<false", -1)],
    "msda_bwd_Lq10200_B16": [("dir_stats_kernel", -1), ("dir_plan_kernel", -1), ("row_candidates_kernel", -1), ("scatter_rows_kernel", -1),
                             ("gather_win_kernel<true", -1)],
    "msda_fwd_Lq550_B16": [("gather_rec_kernel<false", 0)],
    "msda_bwd_Lq550_B16": [("bwd_prep_kernel", 0), ("bwd_scatter_bands_kernel", 0), ("gather_rec_kernel<true", 0)],
}
res["_source"] = {"command": "bash tools/collect_pmc.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on "
                             "tools/msda_fused_bench.py --kinds enc,550: the fused strided operator of the train step, B = 16)",
                  "corrections": "FETCH_SIZE x %.3f (128-byte row gathers, tools/ubench/pmc_calib.hip), WRITE_SIZE x %.3f" % (f_gather, f_write),
                  "kernels": {}}
for op, parts in ops.items():
    tot, used = 0.0, []
    for name, which in parts:
        for k, lst in by_kernel.items():
            if name in k:
                _, kk, grid = sorted(lst)[which]
                tot += res["kernels"]["%s grid=%s" % (kk, grid)]["hbm_bytes"]
                used.append("%s grid=%s" % (kk, grid))
    res[op] = tot
    res["_source"]["kernels"][op] = used
    print("%-28s %8.1f MB per launch (fetch x%.2f + write x%.2f corrected) <- %s" % (op, tot / 1e6, f_gather, f_write, ", ".join(u.split("(")[0] for u in used)))
# the other BASELINE geometries (collect_pmc.sh: c4_* / c5_* passes, encoder shape only): every msda:: kernel of the pass but the
# one-off forward that produces the saved tensors is summed per direction
for tag, key_f, key_b in (("c4", "msda_fwd_Lq11044_B16", "msda_bwd_Lq11044_B16"), ("c5", "msda_fwd_Lq51000_B4", "msda_bwd_Lq51000_B4")):
    per = defaultdict(dict)
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for (k, c), v in counters(tag + "_" + C).items():
            if "msda::" in k:
                per[k][c] = sum(x[1] for x in v) / len(v) * 1024.0
    if not per:
        continue
    fwd = bwd = 0.0
    used_f, used_b = [], []
    for k, cs in per.items():
        nbytes = cs.get("FETCH_SIZE", 0.0) * f_gather + cs.get("WRITE_SIZE", 0.0) * f_write
        if "gather_win_kernel<false" in k:
            fwd += nbytes; used_f.append(k)
        else:
            bwd += nbytes; used_b.append(k)
    res[key_f], res[key_b] = fwd, bwd
    res["_source"]["kernels"][key_f], res["_source"]["kernels"][key_b] = used_f, used_b
    print("%-28s %8.1f MB   %-28s %8.1f MB per launch" % (key_f, fwd / 1e6, key_b, bwd / 1e6))
# which code the counters were collected on (monosowa_amd/build.py leaves the commit next to the libraries: the GPU box has no .git)
_bc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "monosowa_amd", "lib", "BUILD_COMMIT")
res["_collected_at"] = open(_bc).read().strip() if os.path.exists(_bc) else None
json.dump(res, open(os.path.join(d, "msda_traffic.json"), "w"), indent=1)
