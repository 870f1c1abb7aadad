"""`mono_linear_wgrad_f32` (csrc/small_wgrad.hip): dW = dY^T X and db = colsum(dY) of a linear over a few thousand tokens -- autograd's
AddmmBackward for the nn.Linear layers of the reference decoder (depthaware_transformer.py:339-354,440-515) -- against float64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(R, M, N, ldy=None, ldx=None, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    gy = torch.randn(R, ldy or M, generator=g, device="cuda")[:, :M]
    x = torch.randn(R, ldx or N, generator=g, device="cuda")[:, :N]
    return gy, x


@pytest.mark.parametrize("R,M,N", [(8800, 256, 256), (8800, 64, 64), (2049, 128, 192), (30720, 512, 256), (30720, 256, 512),
                                   (64, 64, 64), (65, 64, 128), (4097, 1024, 256), (8800, 256, 1024)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_weight_and_bias_gradient_equal_float64(R, M, N, with_bias):
    from monosowa_amd.pointwise import linear_wgrad
    gy, x = _case(R, M, N)
    gw, gb = linear_wgrad(gy, x, with_bias)
    ref_w = gy.double().t() @ x.double()
    scale = ref_w.abs().max().item()
    assert gw.shape == (M, N) and gw.is_contiguous()
    assert (gw.double() - ref_w).abs().max().item() <= 2e-6 * scale + 1e-5
    if with_bias:
        ref_b = gy.double().sum(0)
        assert gb.shape == (M,)
        assert (gb.double() - ref_b).abs().max().item() <= 2e-6 * ref_b.abs().max().item() + 1e-4
    else:
        assert gb is None


def test_strided_rows_and_determinism():
    from monosowa_amd.pointwise import linear_wgrad
    gy, x = _case(5000, 128, 64, ldy=160, ldx=96, seed=3)          # views of wider matrices: leading dimensions > widths
    assert not gy.is_contiguous() and not x.is_contiguous()
    gw, gb = linear_wgrad(gy, x)
    assert (gw.double() - gy.double().t() @ x.double()).abs().max().item() < 2e-3
    assert (gb.double() - gy.double().sum(0)).abs().max().item() < 2e-3
    gw2, gb2 = linear_wgrad(gy, x)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)          # fixed summation order: bit-identical


def test_unsupported_shapes_are_refused():
    from monosowa_amd.pointwise import linear_wgrad, linear_wgrad_applies
    gy, x = _case(4096, 96, 64)
    assert not linear_wgrad_applies(gy, x)
    with pytest.raises(ValueError):
        linear_wgrad(gy, x)
    gy, x = _case(32, 64, 64)
    assert not linear_wgrad_applies(gy, x)


def test_token_linear_backward_uses_it_and_matches_autograd():
    from monosowa_amd import token_linear as tl
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 256).cuda()
    x = torch.randn(16, 550, 256, device="cuda", requires_grad=True)
    gy = torch.randn(16, 550, 256, device="cuda")
    y = tl.token_linear(x, lin)
    y.backward(gy)
    got = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None; lin.weight.grad = None; lin.bias.grad = None
    torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double()).backward(gy.double())
    # (float64 autograd on the same parameters: .grad is float32 there, so compare against explicit float64 products)
    ref_w = gy.double().reshape(-1, 256).t() @ x.detach().double().reshape(-1, 256)
    ref_b = gy.double().reshape(-1, 256).sum(0)
    assert (got[1].double() - ref_w).abs().max().item() < 1e-3
    assert (got[2].double() - ref_b).abs().max().item() < 1e-3
    assert (got[0].double() - gy.double() @ lin.weight.detach().double()).abs().max().item() < 1e-4
