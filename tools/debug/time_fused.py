import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import test_msda_gpu as T
from monosowa_amd import _lib
MSDA = T._msda()
B = 16
shapes, lsi, ref, offsets, logits, value, go = T._kitti_encoder_inputs(B, 3, 4.0)
S, M = value.shape[1], value.shape[2]
s, i = T._dev(shapes), T._dev(lsi)
MSDA.attach_host_geometry(s, i, shapes.tolist(), lsi.tolist())
proj = torch.cat([T._dev(offsets).reshape(B, S, M * 32), T._dev(logits).reshape(B, S, M * 16)], -1).contiguous()
refp = T._dev(np.broadcast_to(ref[None, :, None, :], (B, S, 4, 2)).copy())
v, g = T._dev(value), T._dev(go)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for w in (0, 3):
    _lib.set_option("window", w)
    tf = timeit(lambda: MSDA.ms_deform_attn_fused_forward_merged(v, s, i, proj, refp))
    tb = timeit(lambda: MSDA.ms_deform_attn_fused_backward_merged(v, s, i, proj, refp, g))
    print("window %d: fused fwd %.3f ms, fused bwd (K1+K2+K3) %.3f ms" % (w, tf, tb))
