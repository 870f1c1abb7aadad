"""The frozen bottleneck tail at layer1's size ([16, 96, 320]): the one-pass kernel against the three passes it replaces."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from monosowa_amd import miopen_tuning
miopen_tuning.use_shipped_db(0)
from monosowa_amd import pointwise as PW


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


cl = lambda t: t.contiguous(memory_format=torch.channels_last)
x = cl(torch.randn(16, 64, 96, 320, device="cuda"))
res = cl(torch.randn(16, 256, 96, 320, device="cuda"))
w = cl(torch.randn(256, 64, 1, 1, device="cuda") / 8)
b_in, b_out = torch.randn(64, device="cuda"), torch.randn(256, device="cuda")
w_kn = w.view(256, 64).t().contiguous()


def three_passes():
    h = PW.bias_act(x.clone(), b_in, None, True)
    y = F.conv2d(h, w)
    return PW.bias_act(y, b_out, res, True)


with torch.no_grad():
    t_clone = timeit(lambda: x.clone())
    t3 = timeit(three_passes) - t_clone
    t1 = timeit(lambda: PW.conv1x1_tail(x, b_in, w_kn, b_out, res))
    err = (PW.conv1x1_tail(x, b_in, w_kn, b_out, res) - three_passes()).abs().max().item()
print("three passes %.1f us, one pass %.1f us (%.2f TB/s of 1.13 GB), max difference %.2e" % (t3, t1, 1.13e9 / t1 / 1e6, err))
