"""Library GEMMs with epilogues through the C-ABI shim over hipBLASLt (include/monosowa_gemm.h, csrc/gemm_lt.cpp).

What the PyTorch front end cannot ask hipBLASLt for: a per-channel scale + shift + residual + ReLU in the GEMM's epilogue (a ResNet
bottleneck's ``relu(bn(conv1x1(x)) (+ identity))`` as ONE launch over the channels-last pixel matrix) and the bias gradient as a
by-product of the weight-gradient GEMM.  Row-major float32 GPU matrices; no CPU path."""
import ctypes
import os

import torch

from ._lib import on_device, raw_stream

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmonosowa_gemm.so")
SYMBOLS = ("mono_gemm_nt_epilogue_f32", "mono_gemm_tn_bgrad_f32", "mono_gemm_nn_f32", "mono_gemm_set_autotune", "mono_gemm_cache_size")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("HIP extension %s is missing: run `python -m monosowa_amd.build`" % _PATH)
        lib = ctypes.CDLL(_PATH)
        P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float
        lib.mono_gemm_nt_epilogue_f32.restype = I
        lib.mono_gemm_nt_epilogue_f32.argtypes = [P, L, P, L, P, L, P, L, I, I, I, P, F, P, I, P]
        lib.mono_gemm_tn_bgrad_f32.restype = I
        lib.mono_gemm_tn_bgrad_f32.argtypes = [P, L, P, L, P, L, P, I, I, I, P]
        lib.mono_gemm_nn_f32.restype = I
        lib.mono_gemm_nn_f32.argtypes = [P, L, P, L, P, L, I, I, I, P]
        lib.mono_gemm_set_autotune.restype = I
        lib.mono_gemm_set_autotune.argtypes = [I]
        lib.mono_gemm_cache_size.restype = I
        n = os.environ.get("MONOSOWA_GEMM_AUTOTUNE")              # candidates timed per problem key (default 32; 1: the library's first choice)
        if n is not None:
            lib.mono_gemm_set_autotune(int(n))
        _lib = lib
    return _lib


def _ok(t):
    return t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0


def supported(*mats):
    return all(m is None or _ok(m) for m in mats)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def gemm_nt(a, w, scale=None, bias=None, residual=None, relu=False, out=None):
    """``act(scale * (a @ w.T) + residual + bias)`` in one launch; a [M, K], w [N, K], scale / bias [N], residual [M, N]."""
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and supported(a, w, residual)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with on_device(a.device):
        code = load().mono_gemm_nt_epilogue_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _ptr(residual),
                                                residual.stride(0) if residual is not None else N, out.data_ptr(), out.stride(0), M, N, K,
                                                _ptr(scale), 1.0 if residual is not None else 0.0, _ptr(bias), int(bool(relu)), raw_stream())
    if code:
        raise RuntimeError("mono_gemm_nt_epilogue_f32 failed with code %d" % code)
    return out


def gemm_tn_bgrad(gy, x, with_bias=True):
    """``(gy.T @ x, gy.sum(0))`` from one launch: the weight and bias gradients of ``y = x w.T + b``; gy [M, N], x [M, K]."""
    M, N = gy.shape
    K = x.shape[1]
    assert x.shape[0] == M and supported(gy, x)
    gw = torch.empty((N, K), dtype=torch.float32, device=gy.device)
    gb = torch.empty((N,), dtype=torch.float32, device=gy.device) if with_bias else None
    with on_device(gy.device):
        code = load().mono_gemm_tn_bgrad_f32(gy.data_ptr(), gy.stride(0), x.data_ptr(), x.stride(0), gw.data_ptr(), K, _ptr(gb), M, N, K,
                                             raw_stream())
    if code:
        raise RuntimeError("mono_gemm_tn_bgrad_f32 failed with code %d" % code)
    return gw, gb


def gemm_nn(gy, w, out=None):
    """``gy @ w``; gy [M, N], w [N, K]."""
    M, N = gy.shape
    K = w.shape[1]
    assert w.shape[0] == N and supported(gy, w)
    gx = torch.empty((M, K), dtype=torch.float32, device=gy.device) if out is None else out
    with on_device(gy.device):
        code = load().mono_gemm_nn_f32(gy.data_ptr(), gy.stride(0), w.data_ptr(), w.stride(0), gx.data_ptr(), gx.stride(0), M, N, K, raw_stream())
    if code:
        raise RuntimeError("mono_gemm_nn_f32 failed with code %d" % code)
    return gx
