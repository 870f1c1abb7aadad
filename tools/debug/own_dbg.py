"""A/B of the saved self-attention backward: owner_grads 1 (GRADS scatter + finish) vs 0 (scatter + window gather)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import test_msda_gpu as T
from monosowa_amd import _lib
from monosowa_amd.ms_deform_attn_func import MSDeformAttnFusedMergedFunction
MSDA = T._msda()
levels = [(24, 40), (12, 20), (6, 10), (3, 5)]
B, M, D, L, P = 2, 8, 32, 4, 4
shapes = torch.tensor(levels, dtype=torch.long, device="cuda")
lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
MSDA.attach_host_geometry(shapes, lsi, levels, lsi.tolist())
S = int(shapes.prod(1).sum()); Lq = S
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w,
                                            indexing="ij")[::-1], -1).reshape(-1, 2) for h, w in levels])
ref = ref[None, :, None, :].expand(B, Lq, L, 2).contiguous()
for masked in (False, True):
    for off_scale in (0.5, 3.0):
        torch.manual_seed(31 + masked)
        value = torch.randn(B, S, M, D, device="cuda")
        mask = (torch.rand(B, S, device="cuda") < 0.25) if masked else None
        proj0 = torch.cat([torch.randn(B, Lq, M * 32, device="cuda") * off_scale, torch.randn(B, Lq, M * 16, device="cuda")], -1)
        go = torch.randn(B, Lq, M * D, device="cuda")
        res = {}
        for mode in (0, 1):
            _lib.set_option("owner_grads", mode)
            v = value.clone().requires_grad_(True)
            proj = proj0.clone().requires_grad_(True)
            out = MSDeformAttnFusedMergedFunction.apply(v, shapes, lsi, proj, ref, mask)
            out.backward(go)
            torch.cuda.synchronize()
            res[mode] = (v.grad.view(B, S, M, D).clone(), proj.grad[:, :, :M * 32].clone().view(B, Lq, M, L, P, 2), proj.grad[:, :, M * 32:].clone().view(B, Lq, M, L, P))
        for k, name in enumerate(("grad_value", "d_offsets", "d_logits")):
            a, b = res[0][k], res[1][k]
            err = (a - b).abs()
            tol = 1e-4 * a.abs().max()
            print("masked", masked, "scale", off_scale, name, "max diff %.3g" % err.max().item(), "ref max %.3g" % a.abs().max().item(),
                  "n bad", int((err > tol).sum()), "of", a.numel())
            if k == 0:
                for l, (h, w) in enumerate(levels):
                    e = err[:, int(lsi[l]):int(lsi[l]) + h * w]
                    print("    level", l, "max diff %.3g" % e.max().item(), "bad rows", int((e.amax(-1) > tol).sum()))
            elif int((err > tol).sum()):
                idx = (err > tol).nonzero()
                print("    first bad", idx[:6].tolist())

# ---- which rows differ, and is the difference a whole point contribution (missing / doubled)? ---------------------------------
for directional in (1, 0):
    _lib.set_option("directional", directional)
    torch.manual_seed(31)
    value = torch.randn(B, S, M, D, device="cuda")
    proj0 = torch.cat([torch.randn(B, Lq, M * 32, device="cuda") * 3.0, torch.randn(B, Lq, M * 16, device="cuda")], -1)
    go = torch.randn(B, Lq, M * D, device="cuda")
    res = {}
    for mode in (0, 1):
        _lib.set_option("owner_grads", mode)
        v = value.clone().requires_grad_(True)
        proj = proj0.clone().requires_grad_(True)
        out = MSDeformAttnFusedMergedFunction.apply(v, shapes, lsi, proj, ref, None)
        out.backward(go)
        torch.cuda.synchronize()
        res[mode] = v.grad.view(B, S, M, D).clone()
    d = res[1] - res[0]
    bad = (d.abs().amax(-1) > 1e-4).nonzero()
    print("directional", directional, "bad rows (b, token, m):", bad.tolist()[:40])
    for b_, t_, m_ in bad.tolist()[:8]:
        l_ = max(l for l in range(4) if t_ >= int(lsi[l]))
        r = t_ - int(lsi[l_]); print("   level", l_, "y", r // levels[l_][1], "x", r % levels[l_][1], "old |row| %.4f new |row| %.4f diff |.| %.4f" % (
            res[0][b_, t_, m_].norm().item(), res[1][b_, t_, m_].norm().item(), d[b_, t_, m_].norm().item()))
