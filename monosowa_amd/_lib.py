"""ctypes loader for the C-ABI library (include/monosowa_msda.h).  No fallback: a missing
library is an error, never a silent CPU path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmonosowa_msda.so")

# every symbol include/monosowa_msda.h declares
SYMBOLS = ("msda_abi_version", "msda_strerror", "msda_set_option", "msda_options_stamp", "msda_debug_counter", "msda_backward_workspace_bytes",
           "msda_forward_f32", "msda_forward_f64", "msda_backward_f32", "msda_backward_f64",
           "msda_fused_forward_f32", "msda_fused_backward_f32", "msda_fused_forward_strided_f32",
           "msda_fused_backward_strided_f32", "msda_fused_save_supported", "msda_fused_save_supported_view", "msda_fused_forward_save_f32", "msda_fused_forward_view_f32", "msda_fused_backward_view_f32", "msda_saved_plan_f32", "msda_fused_backward_view_planned_f32",
           "msda_fused_backward_saved_f32")

_lib = None
ABI_VERSION = 10


class MSDALibraryError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MONOSOWA_MSDA_LIB", LIB_PATH)        # another BUILD of the same library (A/B measurements)
    if not os.path.exists(path):
        raise MSDALibraryError(
            "HIP extension %s is missing: build it with `python -m monosowa_amd.build` "
            "(needs hipcc; there is no CPU fallback)" % path)
    lib = ctypes.CDLL(path)
    P, I, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.msda_abi_version.restype = I
    lib.msda_strerror.restype = ctypes.c_char_p
    lib.msda_strerror.argtypes = [I]
    lib.msda_set_option.restype = I
    lib.msda_set_option.argtypes = [ctypes.c_char_p, I]
    lib.msda_options_stamp.restype = I
    lib.msda_options_stamp.argtypes = []
    lib.msda_debug_counter.restype = I
    lib.msda_debug_counter.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong)]
    lib.msda_backward_workspace_bytes.restype = Z
    lib.msda_backward_workspace_bytes.argtypes = [I] * 8
    for suf in ("f32", "f64"):
        f = getattr(lib, "msda_forward_" + suf)
        f.restype = I
        f.argtypes = [P] * 6 + [I] * 7 + [P, P, P]
        b = getattr(lib, "msda_backward_" + suf)
        b.restype = I
        b.argtypes = [P] * 9 + [I] * 7 + [P, P, P, Z, P]
    lib.msda_fused_forward_f32.restype = I
    lib.msda_fused_forward_f32.argtypes = [P] * 6 + [I, P] + [I] * 7 + [P, P, P]
    lib.msda_fused_backward_f32.restype = I
    lib.msda_fused_backward_f32.argtypes = [P] * 6 + [I] + [P] * 4 + [I] * 7 + [P, P, P, Z, P]
    lib.msda_fused_forward_strided_f32.restype = I
    lib.msda_fused_forward_strided_f32.argtypes = [P] * 6 + [I, P] + [I] * 9 + [P, P, P]
    lib.msda_fused_backward_strided_f32.restype = I
    lib.msda_fused_backward_strided_f32.argtypes = [P] * 6 + [I] + [P] * 4 + [I] * 9 + [P, P, P, Z, P]
    lib.msda_fused_save_supported.restype = I
    lib.msda_fused_save_supported.argtypes = [I] * 7 + [P, P]
    lib.msda_fused_save_supported_view.restype = I
    lib.msda_fused_save_supported_view.argtypes = [I] * 10 + [P, P]
    lib.msda_fused_forward_save_f32.restype = I
    lib.msda_fused_forward_save_f32.argtypes = [P] * 6 + [I, P, P, P] + [I] * 9 + [P, P, P]
    lib.msda_fused_backward_saved_f32.restype = I
    lib.msda_fused_backward_saved_f32.argtypes = [P] * 6 + [I] + [P] * 4 + [I] * 9 + [P, P, P, Z, P]
    # (value, token stride, mask, shapes, lsi, offsets, logits, ref, ref_dim, out, loc_save, attn_save, B..P, strides, hosts, stream)
    lib.msda_fused_forward_view_f32.restype = I
    lib.msda_fused_forward_view_f32.argtypes = [P, I, P, P, P, P, P, P, I, P, P, P] + [I] * 9 + [P, P, P]
    # (value, token stride, mask, shapes, lsi, offsets|loc, logits|attn, saved, ref, ref_dim, grad_out, gv, goff, glog, B..P, ...)
    lib.msda_fused_backward_view_f32.restype = I
    lib.msda_fused_backward_view_f32.argtypes = [P, I, P, P, P, P, P, I, P, I, P, P, P, P] + [I] * 9 + [P, P, P, Z, P]
    lib.msda_saved_plan_f32.restype = I
    lib.msda_saved_plan_f32.argtypes = [P, P, P, I] + [I] * 7 + [I, I, P, P, P, Z, P]
    lib.msda_fused_backward_view_planned_f32.restype = I
    lib.msda_fused_backward_view_planned_f32.argtypes = [P, I, P, P, P, P, P, P, I, P, P, P, P] + [I] * 9 + [P, P, P, Z, P]
    if lib.msda_abi_version() != ABI_VERSION:
        raise MSDALibraryError("ABI version mismatch in %s" % LIB_PATH)
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s (code %d)" % (what, load().msda_strerror(code).decode(), code))


def set_option(name, value):
    check(load().msda_set_option(name.encode(), int(value)), "msda_set_option(%s, %s)" % (name, value))


def debug_counter(name):
    """Reads and resets a device-side diagnostic counter (include/monosowa_msda.h: msda_debug_counter); synchronises the device."""
    out = ctypes.c_ulonglong(0)
    check(load().msda_debug_counter(name.encode(), ctypes.byref(out)), "msda_debug_counter(%s)" % name)
    return int(out.value)


def raw_stream():
    """The current HIP stream handle of the current device as an integer (what the C-ABI entry points take): the raw accessor
    costs 0.1 us where ``torch.cuda.current_stream().cuda_stream`` builds a Stream object for 2.7 us -- on every launch of the
    host-bound stretch of a train step."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


import contextlib

_SAME_DEVICE = contextlib.nullcontext()


def on_device(device):
    """``with on_device(t.device):`` -- makes ``device`` current for a launch.  A no-op object when it already is (one process per
    GPU: always): ``torch.cuda.device(...)`` costs 5-8 us of host time per use, and the launches of the decoder's backward sit
    right behind the matcher's synchronisation where the GPU waits for the host."""
    import torch
    if device.index is None or device.index == torch.cuda.current_device():
        return _SAME_DEVICE
    return torch.cuda.device(device)
