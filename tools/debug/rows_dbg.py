import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import test_msda_gpu as T
from monosowa_amd import _lib
MSDA = T._msda()
levels = [(24, 40), (12, 20), (6, 10), (3, 5)]
value, shapes, lsi, loc, w, go = T._random_case(1 * 131 + 32, 1, 8, 32, 1275, levels, 4, np.float32)
v, s, i, lc, ww, g = map(T._dev, (value, shapes, lsi, loc, w, go))
MSDA.attach_host_geometry(s, i, shapes.tolist(), lsi.tolist())
res = {}
for mode in (0, 1):
    _lib.set_option("scatter_rows", mode)
    gv, gl, gw = MSDA.ms_deform_attn_backward(v, s, i, lc, ww, g, 64)
    torch.cuda.synchronize()
    res[mode] = (gv.cpu().numpy(), gl.cpu().numpy(), gw.cpu().numpy())
for k, name in enumerate(("grad_value", "grad_loc", "grad_attw")):
    a, b = res[0][k], res[1][k]
    err = np.abs(a - b)
    print(name, "max diff", err.max(), "ref max", np.abs(a).max(), "n bad", (err > 1e-4 * np.abs(a).max()).sum(), "of", a.size)
    if k == 1:
        idx = np.argwhere(err > 1e-4 * np.abs(a).max())
        print(idx[:10])
        for ix in idx[:5]:
            print("   old", a[tuple(ix)], "new", b[tuple(ix)])
