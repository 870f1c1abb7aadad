"""GPU time of the ways to sum a [rows, C] gradient over its rows (bias gradient of a linear layer)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from monosowa_amd.pointwise import colsum
from torch.profiler import profile, ProfilerActivity
for rows, C in ((8800, 256), (30720, 81), (8800, 516), (30720, 512), (8800, 6), (8800, 3)):
    g = torch.randn(rows, C, device="cuda")
    ones = torch.ones(rows, device="cuda")
    ones_row = torch.ones(1, rows, device="cuda")
    for name, f in (("sum(0)", lambda: g.sum(0)), ("mv(g.t(), ones)", lambda: torch.mv(g.t(), ones)), ("ones @ g", lambda: ones_row @ g),
                    ("colsum", lambda: colsum(g))):
        try:
            for _ in range(3): f()
        except Exception as e:
            print(rows, C, name, "n/a"); continue
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(30): f()
            torch.cuda.synchronize()
        ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        print("%6d x %4d  %-16s %6.1f us per call (%d launches)" % (rows, C, name, sum(k.device_time for k in ks) / 30, len(ks) // 30))
