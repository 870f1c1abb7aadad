import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from oracle import msda_oracle as O
import test_msda_gpu as T
MSDA = T._msda()
B = 16
shapes, lsi, ref, offsets, logits, value, go = T._kitti_encoder_inputs(B, 41)
loc = (ref[None, :, None, None, None, :] + offsets / shapes[None, None, None, :, None, ::-1]).astype(np.float32)
e = np.exp(logits - logits.max(-1, keepdims=True))
aw = (e / e.sum(-1, keepdims=True)).reshape(B, -1, 8, 4, 4).astype(np.float32)
s, i = T._dev(shapes), T._dev(lsi)
MSDA.attach_host_geometry(s, i, shapes.tolist(), lsi.tolist())
v, lc, w, g = T._dev(value), T._dev(loc), T._dev(aw), T._dev(go)
gv, gl, gw = MSDA.ms_deform_attn_backward(v, s, i, lc, w, g, 64)
torch.cuda.synchronize()
for b in (0, 9):
    want = O.backward(value[b:b + 1], shapes, lsi, loc[b:b + 1], aw[b:b + 1], go[b:b + 1])
    got = gl[b:b+1].cpu().numpy()
    err = np.abs(got - want[1])
    scale = np.abs(want[1]).max()
    idx = np.argwhere(err > 1e-4 * scale)
    print("sample", b, "bad elements", len(idx), "max", err.max() / scale)
    for ix in idx[:12]:
        _, q, m, l, p, xy = ix
        H, W = shapes[l]
        x, y = loc[b, q, m, l, p]
        print("  q %d m %d l %d p %d xy %d: got %.6f want %.6f | w_im %.7f h_im %.7f" % (q, m, l, p, xy, got[tuple(ix)], want[1][tuple(ix)],
              np.float32(x) * np.float32(W) - np.float32(0.5), np.float32(y) * np.float32(H) - np.float32(0.5)))
