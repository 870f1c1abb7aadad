"""host cost of the Python idioms around every ctypes kernel launch"""
import time, torch
dev = torch.device("cuda", 0)
x = torch.empty(16, device=dev)
def t(fn, n=20000):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("torch.cuda.current_stream().cuda_stream   %.2f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("torch._C._cuda_getCurrentRawStream(0)     %.2f us" % t(lambda: torch._C._cuda_getCurrentRawStream(0)))
def ctx():
    with torch.cuda.device(dev):
        pass
print("with torch.cuda.device(dev): pass         %.2f us" % t(ctx))
print("torch.cuda.current_device()               %.2f us" % t(lambda: torch.cuda.current_device()))
print("torch.empty(1024, device)                 %.2f us" % t(lambda: torch.empty(1024, dtype=torch.float32, device=dev)))
print("x.data_ptr()                              %.2f us" % t(lambda: x.data_ptr()))
print("x.is_contiguous()                         %.2f us" % t(lambda: x.is_contiguous()))
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a): return a.view_as(a)
    @staticmethod
    def backward(ctx, g): return g
y = torch.zeros(4, device=dev, requires_grad=True)
print("custom Function.apply (trivial)           %.2f us" % t(lambda: F.apply(y), 5000))
