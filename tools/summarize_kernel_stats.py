"""Category summary of a rocprofv3 *_kernel_stats.csv of a bench.py run:  python tools/summarize_kernel_stats.py <csv> <steps>"""
import csv
import sys


def cat(n):
    if "msda::" in n:
        return "msda (this repo)"
    if n.startswith("Cijk"):
        return "gemm (hipBLASLt)"
    if "batched_transpose" in n:
        return "MIOpen layout transpose"
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


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
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
    for r in rows[:25]:
        print("%5.2f%% %7.2f ms/step %5.0f x %8.1f us  %s" % (float(r["Percentage"]), float(r["TotalDurationNs"]) / 1e6 / steps,
                                                           int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))


if __name__ == "__main__":
    main()
