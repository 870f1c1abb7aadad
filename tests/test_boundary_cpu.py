"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, and the Python mirror keeps the reference's preconditions
(ops/src/ms_deform_attn.h:38,60; ops/src/cuda/ms_deform_attn_cuda.cu:28-52)."""
import ctypes
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "monosowa_msda.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msda_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from monosowa_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python -m monosowa_amd.build` first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SYMBOLS) == names
    assert _lib.load().msda_abi_version() == _lib.ABI_VERSION
    assert b"NULL" in _lib.load().msda_strerror(-1)


def test_cpu_tensors_are_rejected_like_the_reference():
    from monosowa_amd import MultiScaleDeformableAttention as MSDA
    shapes = torch.tensor([[2, 2]], dtype=torch.long)
    lsi = torch.zeros(1, dtype=torch.long)
    v = torch.zeros(1, 4, 1, 4)
    loc = torch.zeros(1, 1, 1, 1, 1, 2)
    w = torch.zeros(1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, w, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, w, torch.zeros(1, 1, 4), 64)


def test_install_registers_reference_module_name():
    from monosowa_amd import MultiScaleDeformableAttention as MSDA
    saved = sys.modules.pop("MultiScaleDeformableAttention", None)
    try:
        MSDA.install()
        import MultiScaleDeformableAttention as again
        assert again.ms_deform_attn_forward is MSDA.ms_deform_attn_forward
        assert again.ms_deform_attn_backward is MSDA.ms_deform_attn_backward
    finally:
        sys.modules.pop("MultiScaleDeformableAttention", None)
        if saved is not None:
            sys.modules["MultiScaleDeformableAttention"] = saved


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "monosowa_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
                assert "libmsda_oracle" not in src


def test_pointwise_library_exports_declared_symbols():
    from monosowa_amd import pointwise
    text = open(os.path.join(ROOT, "include", "monosowa_pointwise.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(mono_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(pointwise.SYMBOLS)
    lib = ctypes.CDLL(pointwise._PATH)
    for n in names:
        assert hasattr(lib, n)
    # CPU tensors take the plain PyTorch formulation
    y, b, r = torch.randn(2, 8, 3, 5), torch.randn(8), torch.randn(2, 8, 3, 5)
    assert torch.allclose(pointwise.bias_act(y.clone(), b, r), torch.relu(y + b.view(1, -1, 1, 1) + r))


def test_attention_library_exports_declared_symbols():
    from monosowa_amd import flash_attn
    text = open(os.path.join(ROOT, "include", "monosowa_attn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(mono_attn_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(flash_attn.SYMBOLS)
    lib = ctypes.CDLL(flash_attn._PATH)
    for n in names:
        assert hasattr(lib, n)
    # the product path refuses CPU tensors instead of falling back
    q = torch.randn(1, 2, 5, 32)
    assert not flash_attn.supported(q, q, q)
    with pytest.raises(RuntimeError):
        flash_attn.attention(q, q, q)
    mha = torch.nn.MultiheadAttention(64, 2)
    assert not flash_attn.mha_supported(mha, torch.randn(5, 1, 64), torch.randn(5, 1, 64), torch.randn(5, 1, 64))
    # key padding masks the kernels read: [B, Lk] bool / uint8 on the queries' device; an additive float mask (0 / -inf), another
    # shape or another device keeps the module path
    dev = torch.device("cpu")
    assert flash_attn.mask_supported(None, 2, 7, dev)
    assert flash_attn.mask_supported(torch.zeros(2, 7, dtype=torch.bool), 2, 7, dev)
    assert flash_attn.mask_supported(torch.zeros(2, 7, dtype=torch.uint8), 2, 7, dev)
    assert not flash_attn.mask_supported(torch.zeros(2, 7), 2, 7, dev)
    assert not flash_attn.mask_supported(torch.zeros(7, 2, dtype=torch.bool), 2, 7, dev)
    assert not flash_attn.mask_supported(torch.zeros(2, 7, dtype=torch.bool, device="meta"), 2, 7, dev)


def test_gemm_library_exports_declared_symbols():
    from monosowa_amd import gemm_lt
    text = open(os.path.join(ROOT, "include", "monosowa_gemm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(mono_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(gemm_lt.SYMBOLS)
    lib = ctypes.CDLL(gemm_lt._PATH)
    for n in names:
        assert hasattr(lib, n)
    # argument checks come before any device work
    assert gemm_lt.load().mono_gemm_nn_f32(None, 0, None, 0, None, 0, 1, 1, 1, None) == -1
    assert not gemm_lt.supported(torch.randn(4, 4))


def test_kitti_library_exports_declared_symbols():
    from monosowa_amd import kitti_eval
    text = open(os.path.join(ROOT, "include", "monosowa_kitti.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(mono_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(kitti_eval.SYMBOLS)
    lib = ctypes.CDLL(kitti_eval._PATH)
    for n in names:
        assert hasattr(lib, n)
    assert lib.mono_extract_dets_f32(None, None, None, None, None, None, 1, 1, 1, 1, None) == -1
