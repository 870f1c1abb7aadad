"""Phase clocks of the window gather kernels from a -DMSDA_WIN_STAMP=1 build of the library (MONOSOWA_MSDA_LIB=<that .so>):
shader-clock cycles per wave and launch, by phase.  stdout."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from monosowa_amd import MultiScaleDeformableAttention as MSDA, _lib  # noqa: E402
import msda_fused_bench as fb  # noqa: E402

NAMES = ["loop head", "taps", "fill wait", "barrier after the fill", "prefetch + saves", "row loops", "fix-up + reduce + stores", "wait for the unit's inputs",
         "geometry scalars", "barrier before the fill", "fill issue", "-"]


def main():
    iters = 10
    dev = torch.device("cuda:0")
    value, shapes, lsi, proj, ref, go = fb.make(16, "enc", "init", dev)
    _, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref)
    fwd = lambda: MSDA.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref)
    bwd = lambda: MSDA.ms_deform_attn_fused_backward_merged_saved(value, shapes, lsi, loc, attw, ref, go)
    lib = _lib.load()
    buf = (ctypes.c_ulonglong * 24)()
    for name, fn in (("forward", fwd), ("backward", bwd)) if hasattr(lib, "msda_debug_stamps") else ():
        fn()
        torch.cuda.synchronize()
        lib.msda_debug_stamps(buf)
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        lib.msda_debug_stamps(buf)
        vals = list(buf)[0:12] if name == "forward" else list(buf)[12:24]
        waves = 256 * (16 if name == "forward" else 12)
        tot = sum(vals)
        print("%s: %.0f k cycles per wave and launch" % (name, tot / iters / waves / 1e3))
        for n, v in zip(NAMES, vals):
            if v:
                print("   %-28s %8.1f k cycles  %5.1f %%" % (n, v / iters / waves / 1e3, 100.0 * v / tot))
    if hasattr(lib, "msda_debug_rows_stamps"):
        rb = (ctypes.c_ulonglong * 8)()
        lib.msda_debug_rows_stamps(rb)
        for _ in range(iters):
            bwd()
        torch.cuda.synchronize()
        lib.msda_debug_rows_stamps(rb)
        names = ["loop head", "taps + go staging + next fetch", "slot reservations + entries", "barrier 1", "bucket walk", "barrier 2", "tail", "wait for the batch's inputs"]
        tot = sum(rb)
        print("row scatter: %.3g cycles summed over waves and %d launches" % (tot, iters))
        for n, v in zip(names, list(rb)):
            print("   %-32s %5.1f %%" % (n, 100.0 * v / tot))


if __name__ == "__main__":
    main()
