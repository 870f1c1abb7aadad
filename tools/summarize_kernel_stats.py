"""Category summary of a rocprofv3 run of bench.py:
    python tools/summarize_kernel_stats.py <..._kernel_stats.csv> <steps in the run>
    python tools/summarize_kernel_stats.py <..._kernel_trace.csv> <last K steps to summarise> [kernels to list, default 25]"""
import csv
import sys


def cat(n):
    if "msda::" in n:
        return "msda (this repo)"
    if n.startswith("Cijk"):
        return "gemm (hipBLASLt)"
    if "batched_transpose" in n:
        return "MIOpen layout transpose"
    if "attn::" in n:
        return "attention (this repo)"
    if "mono::" in n:
        return "pointwise / norms (this repo)"
    if any(k in n for k in ("igemm", "Conv", "conv", "miopen", "gtc", "Sp3Asm", "xdlops")):
        return "conv (MIOpen/CK)"
    if "attn_fwd" in n or "bwd_kernel" in n:
        return "attention (aotriton)"
    if "layer_norm" in n or "GammaBeta" in n or "GradInput" in n or "roupNorm" in n or "RowwiseMoments" in n or "group_norm" in n:
        return "norms"
    if "elementwise" in n or "Elementwise" in n:
        return "elementwise"
    if "reduce" in n or "Reduce" in n:
        return "reduce"
    if "copyBuffer" in n or "fillBuffer" in n:
        return "copy/fill"
    if "multi_tensor" in n or "foreach" in n:
        return "optimizer (foreach)"
    if "cdist" in n:
        return "cdist"
    if "upsample" in n:
        return "upsample"
    if "index" in n or "gather" in n or "scatter" in n:
        return "index/gather/scatter"
    return "other"


def rows_from_trace(path, steps):
    """Per-kernel totals over the LAST `steps` train steps of a rocprofv3 *_kernel_trace.csv (steps are delimited by
    the backbone's single stem-pooling launch), i.e. without warm-up, MIOpen's find phase and JIT effects."""
    trace = list(csv.DictReader(open(path)))
    trace.sort(key=lambda r: int(r["Start_Timestamp"]))
    # (one launch per train step: the stem's pooling -- the one-pass stem kernel since round 4, torch's max-pool before)
    marks = [i for i, r in enumerate(trace) if "bias_relu_maxpool_kernel" in r["Kernel_Name"] or "max_pool_forward" in r["Kernel_Name"]]
    if len(marks) < steps:
        raise SystemExit("only %d steps in the trace" % len(marks))
    first = marks[-int(steps)]
    acc = {}
    for r in trace[first:]:
        d = acc.setdefault(r["Kernel_Name"], [0, 0])
        d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[1] += 1
    tot = sum(v[0] for v in acc.values())
    rows = [{"Name": k, "TotalDurationNs": v[0], "Calls": v[1], "AverageNs": v[0] / v[1], "Percentage": 100.0 * v[0] / tot}
            for k, v in acc.items()]
    rows.sort(key=lambda r: -r["TotalDurationNs"])
    span = int(trace[-1]["End_Timestamp"]) - int(trace[first]["Start_Timestamp"])
    print("window: last %d steps, %.1f ms/step wall on the GPU timeline" % (steps, span / 1e6 / steps))
    return rows


def main():
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    if sys.argv[1].endswith("kernel_trace.csv"):
        rows = rows_from_trace(sys.argv[1], steps)
    else:
        rows = list(csv.DictReader(open(sys.argv[1])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("kernel time %.1f ms/step over %.0f launches/step" % (tot / 1e6 / steps, sum(int(r["Calls"]) for r in rows) / steps))
    cats = {}
    for r in rows:
        d = cats.setdefault(cat(r["Name"]), [0.0, 0])
        d[0] += float(r["TotalDurationNs"])
        d[1] += int(r["Calls"])
    for c, (t, n) in sorted(cats.items(), key=lambda kv: -kv[1][0]):
        print("%-26s %7.2f ms/step %5.1f%%  %6.0f launches/step" % (c, t / 1e6 / steps, 100 * t / tot, n / steps))
    print()
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25         # third argument: how many kernels to list
    for r in rows[:top]:
        print("%5.2f%% %7.2f ms/step %5.0f x %8.1f us  %s" % (float(r["Percentage"]), float(r["TotalDurationNs"]) / 1e6 / steps,
                                                           int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))


if __name__ == "__main__":
    main()
