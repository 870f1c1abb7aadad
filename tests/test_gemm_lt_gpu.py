"""The hipBLASLt shim (include/monosowa_gemm.h): epilogue GEMMs against float64 PyTorch evaluations of the same expressions."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(7680, 512, 128), (1000, 64, 256), (30720, 256, 1024), (333, 132, 68)])
@pytest.mark.parametrize("scale,bias,residual,relu", [(True, True, True, True), (False, True, False, True), (True, False, False, False),
                                                      (False, False, True, False), (False, True, False, False)])
def test_nt_epilogue_matches_float64(M, N, K, scale, bias, residual, relu):
    """relu(scale * (A W^T) + C + bias): the bottleneck's conv1x1 + frozen-BN + identity + ReLU as one library launch."""
    from monosowa_amd import gemm_lt
    torch.manual_seed(M + N + K)
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5
    s = torch.rand(N, device="cuda") + 0.5 if scale else None
    b = torch.randn(N, device="cuda") if bias else None
    c = torch.randn(M, N, device="cuda") if residual else None
    got = gemm_lt.gemm_nt(a, w, s, b, c, relu)
    ref = a.double() @ w.double().t()
    if scale:
        ref = ref * s.double()
    if residual:
        ref = ref + c.double()
    if bias:
        ref = ref + b.double()
    if relu:
        ref = ref.clamp_min(0)
    assert _rel(got, ref) <= 2e-6


@pytest.mark.parametrize("M,N,K", [(8800, 256, 256), (163200, 256, 256), (30720, 128, 512), (999, 36, 260)])
def test_tn_bgrad_gives_weight_and_bias_gradient(M, N, K):
    from monosowa_amd import gemm_lt
    torch.manual_seed(M + N)
    gy, x = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
    gw, gb = gemm_lt.gemm_tn_bgrad(gy, x)
    assert _rel(gw, gy.double().t() @ x.double()) <= 2e-5
    assert _rel(gb, gy.double().sum(0)) <= 2e-5
    gw2, none = gemm_lt.gemm_tn_bgrad(gy, x, with_bias=False)
    assert none is None and _rel(gw2, gy.double().t() @ x.double()) <= 2e-5


def test_nn_and_strided_views():
    """dX = dY W; and operands that are column blocks of wider buffers (leading dimension > width)."""
    from monosowa_amd import gemm_lt
    torch.manual_seed(5)
    gy, w = torch.randn(4000, 192, device="cuda"), torch.randn(192, 320, device="cuda")
    assert _rel(gemm_lt.gemm_nn(gy, w), gy.double() @ w.double()) <= 2e-5
    big = torch.randn(4000, 512, device="cuda")
    a = big[:, 128:384]                                   # [4000, 256] view, row stride 512
    wt = torch.randn(64, 256, device="cuda")
    b = torch.randn(64, device="cuda")
    assert _rel(gemm_lt.gemm_nt(a, wt, None, b, None, True), (a.double() @ wt.double().t() + b.double()).clamp_min(0)) <= 2e-6


def test_kernel_selection_is_cached_per_problem_key_and_can_skip_the_timing():
    """One selection per (shape, epilogue) key: a repeated call adds nothing to the cache, another epilogue or shape adds one;
    mono_gemm_set_autotune(1) takes the library's first choice without timing (no host synchronisation) and computes the same product."""
    from monosowa_amd import gemm_lt
    lib = gemm_lt.load()
    torch.manual_seed(2)
    a, w = torch.randn(640, 96, device="cuda"), torch.randn(48, 96, device="cuda")
    b = torch.randn(48, device="cuda")
    n0 = lib.mono_gemm_cache_size()
    y1 = gemm_lt.gemm_nt(a, w, None, b, None, True)
    n1 = lib.mono_gemm_cache_size()
    y2 = gemm_lt.gemm_nt(a, w, None, b, None, True)
    assert lib.mono_gemm_cache_size() == n1 == n0 + 1 and torch.equal(y1, y2)
    gemm_lt.gemm_nt(a, w, None, b, None, False)                       # another epilogue: another key
    assert lib.mono_gemm_cache_size() == n1 + 1
    prev = lib.mono_gemm_set_autotune(1)
    try:
        a2 = torch.randn(704, 96, device="cuda")                       # a new shape, selected without timing
        y3 = gemm_lt.gemm_nt(a2, w, None, b, None, True)
        assert lib.mono_gemm_cache_size() == n1 + 2
        assert _rel(y3, (a2.double() @ w.double().t() + b.double()).clamp_min(0)) <= 2e-6
    finally:
        assert lib.mono_gemm_set_autotune(prev) == 1
