"""Time of the small-linear weight + bias gradient (csrc/small_wgrad.hip) against torch.mm + colsum, back to back on L2-warm operands and
behind a producer GEMM (the situation in the step: dY has just been written).  python tools/debug/wgrad_time.py [R M N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from monosowa_amd.pointwise import linear_wgrad, colsum


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    shapes = [(8800, 256, 256), (30720, 512, 256), (30720, 256, 512), (8800, 256, 1024), (8800, 1024, 256)]
    if len(sys.argv) == 4:
        shapes = [tuple(int(v) for v in sys.argv[1:4])]
    for R, M, N in shapes:
        gy = torch.randn(R, M, device="cuda"); x = torch.randn(R, N, device="cuda")
        if os.environ.get("WGRAD_ZEROS"):                    # all-zero operands: what the data costs the matrix pipe (power management)
            gy.zero_(); x.zero_()
        w = torch.randn(M, M, device="cuda")
        ours = timed(lambda: linear_wgrad(gy, x))
        lib = timed(lambda: (torch.mm(gy.t(), x), colsum(gy)))
        prod = timed(lambda: torch.mm(gy, w))
        ours_p = timed(lambda: (torch.mm(gy, w, out=gy2), linear_wgrad(gy2, x))) if (gy2 := torch.empty_like(gy)) is not None else 0
        lib_p = timed(lambda: (torch.mm(gy, w, out=gy2), torch.mm(gy2.t(), x), colsum(gy2)))
        print("R=%d M=%d N=%d: ours %.1f us  torch.mm + colsum %.1f us | behind a producer GEMM (%.1f us): ours %.1f  library %.1f"
              % (R, M, N, ours, lib, prod, ours_p - prod, lib_p - prod), flush=True)


if __name__ == "__main__":
    main()
