import torch
from torch.profiler import profile, ProfilerActivity
x = torch.randn(8800, 256, device="cuda", requires_grad=True)
w = torch.randn(256, 256, device="cuda", requires_grad=True)
b = torch.randn(256, device="cuda", requires_grad=True)
def a():
    return torch.relu(torch.nn.functional.linear(x, w, b))
def c():
    return torch._addmm_activation(b, x, w.t())
ya, yc = a(), c()
print("max diff", (ya - yc).abs().max().item(), "bitwise", bool((ya == yc).all()))
for f in (a, c):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        y = f(); torch.cuda.synchronize()
    ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(f.__name__, len(ks), [k.name[:40] for k in ks])
