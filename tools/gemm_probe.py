import torch, time, os
x = torch.randn(163200, 256, device="cuda"); w = torch.randn(256, 256, device="cuda"); b = torch.randn(256, device="cuda")
def t(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time()-t0)/n*1e3
print("linear 163200x256x256 fwd ms", t(lambda: torch.nn.functional.linear(x, w, b)), "backend pref hipblaslt:", os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT"))
print("matmul no bias ms", t(lambda: x @ w.t()))
g = torch.randn(163200, 256, device="cuda")
print("wgrad (g^T x) ms", t(lambda: g.t() @ x))
print("dgrad (g w) ms", t(lambda: g @ w))
w2 = torch.randn(128, 256, device="cuda")
print("linear N=128 ms", t(lambda: torch.nn.functional.linear(x, w2)))
for S in (16, 32, 64, 128):
    if 163200 % S: continue
    gs, xs = g.view(S, 163200 // S, 256), x.view(S, 163200 // S, 256)
    print("split-K wgrad via bmm, %3d slices: ms" % S, t(lambda: torch.bmm(gs.transpose(1, 2), xs).sum(0)))
    ref = g.t() @ x; got = torch.bmm(gs.transpose(1, 2), xs).sum(0)
print("max rel diff", ((ref - got).abs().max() / ref.abs().max()).item())
print("bias grad g.sum(0) ms", t(lambda: g.sum(0)))
