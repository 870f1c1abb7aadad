"""Row a10 on the GPU: the whole backbone path of this package -- ResNet-50 / ResNet-101 body with frozen batch-norm FOLDED
into the convolutions, channels-last activations, the fused bias + ReLU (+ residual, + byte mask) kernels, the three-handle
stage outputs, the 1x1 / 3x3 input projections with the convolution bias inside the NHWC GroupNorm kernels, and the sine
position encoding -- against an independent float64 CPU evaluation of the SAME state dict written with nothing but
F.conv2d, the frozen-BN affine map, F.max_pool2d and F.group_norm.

Structure restated from the reference: lib/models/monodetr/backbone.py:28-135 (FrozenBatchNorm2d, BackboneBase with
layer2..4 -> "0", "1", "2", Joiner), monodetr.py:84-105 (input_proj: Conv2d 1x1 + GroupNorm(32, 256) per level, one extra
3x3 stride-2 level from C5), :165-184 (srcs / masks / pos per level), position_encoding.py:36-56.  torchvision (the
reference's ResNet source) is absent from this image, so reference OUTPUTS do not exist: this test
pins the optimised GPU path to plain arithmetic on the published ResNet-v1.5 structure."""
import math
import os

import pytest
import torch
import torch.nn.functional as F
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEPTHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


def _model(name):
    from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "monodetr.yaml")))
    torch.manual_seed(5)
    model, _ = build_model(dict(cfg["model"], backbone=name, device="cuda", depth_map_size=(20, 6)))
    # frozen batch-norm buffers away from the identity map (a checkpoint's are), input projections with a bias
    gen = torch.Generator().manual_seed(11)
    for n, b in model.backbone.named_buffers():
        if n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=gen) * 1.5 + 0.5)
        elif n.endswith("weight"):
            b.copy_(torch.rand(b.shape, generator=gen) * 0.6 + 0.5)
        else:
            b.copy_(torch.randn(b.shape, generator=gen) * 0.2)
    for proj in model.input_proj:
        proj[0].bias.data.copy_(torch.randn(proj[0].bias.shape, generator=gen) * 0.3)
        proj[1].weight.data.copy_(torch.rand(proj[1].weight.shape, generator=gen) + 0.5)
        proj[1].bias.data.copy_(torch.randn(proj[1].bias.shape, generator=gen) * 0.3)
    sd = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    return to_mi355x_layout(model.cuda()), sd


def _reference(sd, name, x):
    """float64, plain PyTorch functions only."""
    def conv_bn(x, conv, bn, stride, padding):
        y = F.conv2d(x, sd[conv + ".weight"], None, stride, padding)
        scale = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-5)             # FrozenBatchNorm2d, backbone.py:52-65
        shift = sd[bn + ".bias"] - sd[bn + ".running_mean"] * scale
        return y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)

    p = "backbone.0.body."
    x = F.relu(conv_bn(x, p + "conv1", p + "bn1", 2, 3))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, blocks in enumerate(DEPTHS[name], start=1):
        for b in range(blocks):
            q = "%slayer%d.%d." % (p, li, b)
            stride = 2 if (b == 0 and li > 1) else 1                                         # v1.5: the 3x3 carries the stride
            out = F.relu(conv_bn(x, q + "conv1", q + "bn1", 1, 0))
            out = F.relu(conv_bn(out, q + "conv2", q + "bn2", stride, 1))
            out = conv_bn(out, q + "conv3", q + "bn3", 1, 0)
            idt = conv_bn(x, q + "downsample.0", q + "downsample.1", stride, 0) if (q + "downsample.0.weight") in sd else x
            x = F.relu(out + idt)
        if li >= 2:
            feats.append(x)
    srcs = []
    for l in range(4):
        src_in = feats[l] if l < 3 else feats[2]
        k = "input_proj.%d." % l
        y = F.conv2d(src_in, sd[k + "0.weight"], sd[k + "0.bias"], 1 if l < 3 else 2, 0 if l < 3 else 1)
        srcs.append(F.group_norm(y, 32, sd[k + "1.weight"], sd[k + "1.bias"], 1e-5))
    return feats, srcs


def _sine_reference(B, H, W, num_pos_feats=128, temperature=10000.0):
    """All-valid mask: the counts are 1..H / 1..W, normalised by the last one + 1e-6 and scaled by 2 pi."""
    y = torch.arange(1, H + 1, dtype=torch.float64) / (H + 1e-6) * 2 * math.pi
    x = torch.arange(1, W + 1, dtype=torch.float64) / (W + 1e-6) * 2 * math.pi
    c = torch.arange(num_pos_feats, dtype=torch.float64)
    period = temperature ** (2 * torch.floor(c / 2) / num_pos_feats)
    even = (torch.arange(num_pos_feats) % 2 == 0)
    ay, ax = y[:, None] / period, x[:, None] / period
    ey = torch.where(even, ay.sin(), ay.cos())                      # [H, F]
    ex = torch.where(even, ax.sin(), ax.cos())                      # [W, F]
    pos = torch.cat([ey[:, None, :].expand(H, W, -1), ex[None, :, :].expand(H, W, -1)], -1)
    return pos.permute(2, 0, 1)[None].expand(B, -1, -1, -1)


def _rel(got, want):
    return float((got.double().cpu() - want).abs().max() / want.abs().max())


@pytest.mark.parametrize("name", ["resnet50", "resnet101"])
@pytest.mark.parametrize("train", [False, True])
def test_backbone_joiner_and_projections_equal_a_plain_float64_evaluation(name, train):
    model, sd = _model(name)
    model.train(train)
    gen = torch.Generator().manual_seed(3)
    images = torch.randn(2, 3, 96, 320, generator=gen)
    want_feats, want_srcs = _reference(sd, name, images.double())
    x = images.cuda().contiguous(memory_format=torch.channels_last)
    with torch.set_grad_enabled(train):            # train: the trainable stages run the affine-in-kernel / forked-ReLU nodes
        features, pos = model.backbone(x)
        srcs, masks, pos = model.project_features(features, pos)
    assert [tuple(f.tensors.shape) for f in features] == [tuple(w.shape) for w in want_feats]
    for l, (f, w) in enumerate(zip(features, want_feats)):
        assert not f.mask.any()
        assert _rel(f.tensors, w) <= 1e-4, ("C%d" % (l + 3), _rel(f.tensors, w))
    assert len(srcs) == 4 and len(pos) == 4 and len(masks) == 4
    for l, (s, w) in enumerate(zip(srcs, want_srcs)):
        assert tuple(s.shape) == tuple(w.shape)
        assert _rel(s, w) <= 1e-4, ("input_proj level %d" % l, _rel(s, w))
        want_pos = _sine_reference(2, s.shape[2], s.shape[3])
        assert tuple(pos[l].shape) == tuple(want_pos.shape)
        assert _rel(pos[l], want_pos) <= 1e-5, ("sine position encoding level %d" % l, _rel(pos[l], want_pos))
    if train:
        # the fused nodes carry a gradient to the trainable stages and none to the frozen stem / layer1 (backbone.py:72-74)
        sum(s.square().mean() for s in srcs).backward()
        body = model.backbone[0].body
        assert body.layer2[0].conv1.weight.grad is not None and torch.isfinite(body.layer4[-1].conv3.weight.grad).all()
        assert body.conv1.weight.grad is None and body.layer1[0].conv1.weight.grad is None


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W", [(2, 64, 192, 640), (1, 8, 7, 9), (3, 64, 1, 1), (2, 12, 10, 5)])
def test_frozen_stem_pass_equals_bias_relu_maxpool(N, C, H, W):
    """mono_bias_relu_maxpool_nhwc_f32 (the frozen stem: BN shift + ReLU + 3x3 / 2 max-pool in one pass) against the three PyTorch
    ops, bit for bit (max and add commute exactly here), odd extents and borders included."""
    from monosowa_amd import pointwise as PW
    torch.manual_seed(N + C + H + W)
    y = (torch.randn(N, C, H, W, device="cuda") * 3).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(C, device="cuda")
    assert PW.bias_relu_maxpool_supported(y, bias)
    got = PW.bias_relu_maxpool(y, bias)
    want = torch.nn.functional.max_pool2d(torch.relu(y + bias.view(1, -1, 1, 1)), kernel_size=3, stride=2, padding=1)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)
    y.requires_grad_(True)
    assert not PW.bias_relu_maxpool_supported(y, bias), "a stem that is trained keeps its autograd path"


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(2, 96, 320), (1, 5, 7), (3, 1, 1), (1, 33, 31)])
def test_frozen_bottleneck_tail_equals_the_three_passes(N, H, W):
    """mono_conv1x1_tail_f32 (bias + ReLU on the 3x3 convolution's output, the 1x1 convolution on the exact-f32 matrix cores, shift +
    identity + ReLU: one pass) against the PyTorch ops evaluated in float64; pixel counts that are not a multiple of 32 included."""
    from monosowa_amd import pointwise as PW
    torch.manual_seed(N * 1000 + H + W)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    x = cl(torch.randn(N, 64, H, W, device="cuda"))
    res = cl(torch.randn(N, 256, H, W, device="cuda"))
    w = torch.randn(256, 64, 1, 1, device="cuda") / 8
    b_in, b_out = torch.randn(64, device="cuda"), torch.randn(256, device="cuda")
    w_kn = w.view(256, 64).t().contiguous()
    assert PW.conv1x1_tail_supported(x, w_kn, res)
    got = PW.conv1x1_tail(x, b_in, w_kn, b_out, res)
    h = torch.relu(x.double() + b_in.double().view(1, -1, 1, 1))
    want = torch.relu(F.conv2d(h, w.double()) + b_out.double().view(1, -1, 1, 1) + res.double())
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (got.double() - want).abs().max() <= 1e-5 * want.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("K,N,H,W", [(256, 2, 96, 320), (64, 2, 96, 320), (256, 1, 5, 7), (64, 3, 1, 1)])
def test_frozen_bottleneck_head_equals_conv_bias_relu(K, N, H, W):
    """mono_conv1x1_head_f32 (conv1 + bn1 shift + ReLU in one pass, K = 64 / 256 -> 64 channels) against float64."""
    from monosowa_amd import pointwise as PW
    torch.manual_seed(K + N + H + W)
    x = torch.randn(N, K, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, K, 1, 1, device="cuda") / K ** 0.5
    b = torch.randn(64, device="cuda")
    w_kn = w.view(64, K).t().contiguous()
    assert PW.conv1x1_head_supported(x, w_kn)
    got = PW.conv1x1_head(x, w_kn, b)
    want = torch.relu(F.conv2d(x.double(), w.double()) + b.double().view(1, -1, 1, 1))
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (got.double() - want).abs().max() <= 1e-5 * want.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(2, 96, 320), (1, 5, 7), (3, 1, 1)])
def test_frozen_first_bottleneck_tail_with_downsample_equals_the_passes(N, H, W):
    """mono_conv1x1_tail_ds_f32: conv3 of relu(x + b) and the downsample convolution of the block's input in one accumulator,
    shift + ReLU -- against float64."""
    from monosowa_amd import pointwise as PW
    torch.manual_seed(N * 77 + H + W)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    x, x0 = cl(torch.randn(N, 64, H, W, device="cuda")), cl(torch.randn(N, 64, H, W, device="cuda"))
    w, wd = torch.randn(256, 64, 1, 1, device="cuda") / 8, torch.randn(256, 64, 1, 1, device="cuda") / 8
    b_in, b_out = torch.randn(64, device="cuda"), torch.randn(256, device="cuda")
    w_kn, wd_kn = w.view(256, 64).t().contiguous(), wd.view(256, 64).t().contiguous()
    assert PW.conv1x1_tail_ds_supported(x, w_kn, x0, wd_kn)
    got = PW.conv1x1_tail_ds(x, b_in, w_kn, x0, wd_kn, b_out)
    h = torch.relu(x.double() + b_in.double().view(1, -1, 1, 1))
    want = torch.relu(F.conv2d(h, w.double()) + F.conv2d(x0.double(), wd.double()) + b_out.double().view(1, -1, 1, 1))
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (got.double() - want).abs().max() <= 1e-5 * want.abs().max()


@pytest.mark.parametrize("downsample,n_out", [(False, 2), (True, 2), (False, 3)], ids=["identity", "downsample", "three-consumers"])
def test_trainable_bottleneck_through_epilogue_gemms_equals_float64(downsample, n_out, monkeypatch):
    """A TRAINABLE bottleneck (layer2-4 of the reference's backbone.py:72-74 freeze rule) with its two 1 x 1 convolutions run as
    library GEMMs whose epilogue applies the frozen norm's scale + shift, the identity and the ReLU (backbone.CONV1X1_EPILOGUE = 3:
    forward and input gradient): output and every gradient (input pair, three weights) against a float64 evaluation with F.conv2d,
    the affine map and F.relu; ragged pixel count, frozen-BN buffers away from the identity."""
    from monosowa_amd.monodetr import backbone as BB
    torch.manual_seed(13)
    inplanes, planes, stride = (256, 128, 2) if downsample else (512, 128, 1)
    ds = None
    if downsample:
        ds = torch.nn.Sequential(torch.nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), BB.FrozenBatchNorm2d(planes * 4))
    blk = BB.Bottleneck(inplanes, planes, stride, ds).cuda()
    gen = torch.Generator().manual_seed(3)
    for n, b in blk.named_buffers():
        b.copy_((torch.rand(b.shape, generator=gen) * 1.5 + 0.5) if ("var" in n or n.endswith("weight")) else torch.randn(b.shape, generator=gen) * 0.2)
    blk = blk.to(memory_format=torch.channels_last)
    blk.n_out = n_out                                    # 3: the last block of a stage (next stage, its identity branch, an input projection)
    x = torch.randn(3, inplanes, 13, 22, device="cuda").contiguous(memory_format=torch.channels_last)

    def run(flag):
        monkeypatch.setattr(BB, "CONV1X1_EPILOGUE", flag)
        blk.zero_grad(set_to_none=True)
        xa = x.clone().requires_grad_(True)
        ys = blk(xa)                                      # one tensor object per consumer
        assert len(ys) == n_out
        torch.manual_seed(1)
        gs = [torch.randn(ys[0].shape, device="cuda") for _ in ys]
        torch.autograd.backward(list(ys), gs)
        return [ys[0].detach(), xa.grad] + [p.grad for p in blk.parameters()], sum(gs)

    got, gsum = run(3)
    old, _ = run(0)
    # float64 reference
    sd = {k: v.detach().double().cpu() for k, v in blk.state_dict().items()}
    xd = x.double().cpu().requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.endswith("weight") and "conv" in k or k == "downsample.0.weight"}

    def bn(y, p):
        scale = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + 1e-5)
        return y * scale.view(1, -1, 1, 1) + (sd[p + ".bias"] - sd[p + ".running_mean"] * scale).view(1, -1, 1, 1)
    o = F.relu(bn(F.conv2d(xd, params["conv1.weight"]), "bn1"))
    o = F.relu(bn(F.conv2d(o, params["conv2.weight"], None, stride, 1), "bn2"))
    o = bn(F.conv2d(o, params["conv3.weight"]), "bn3")
    idt = bn(F.conv2d(xd, params["downsample.0.weight"], None, stride), "downsample.1") if downsample else xd
    yd = F.relu(o + idt)
    yd.backward(gsum.double().cpu())
    names = [n for n, _ in blk.named_parameters()]
    want = [yd.detach(), xd.grad] + [params[n].grad for n in names]
    for name, a, b, c in zip(["out", "grad_x"] + names, got, want, old):
        assert _rel(a, b) <= 2e-5, (name, _rel(a, b))
        assert _rel(c, b) <= 2e-5, ("module path " + name, _rel(c, b))


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W", [(2, 128, 48, 160), (1, 4, 3, 5), (3, 512, 1, 1)])
def test_relu_backward_with_the_frozen_norms_scale_in_the_same_pass(N, C, H, W):
    """pointwise.relu_grad_from_output(scale=...) (mono_relu_grad_scale_f32): scale[c] * g * (y > 0) on channels-last tensors -- the ReLU
    backward of a trainable 1 x 1 convolution + frozen BN without an identity branch (backbone.py:72-115 -> torchvision Bottleneck
    conv1 / bn1 / relu), exact against the two-step evaluation; and the flag that switches the node's backward to it changes no result
    beyond rounding."""
    from monosowa_amd.pointwise import relu_grad_from_output
    torch.manual_seed(C)
    g = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    scale = torch.rand(C, device="cuda") + 0.5
    got = relu_grad_from_output([g], y, scale)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, torch.where(y > 0, g * scale.view(1, -1, 1, 1), torch.zeros_like(g)))
    with pytest.raises(ValueError):
        relu_grad_from_output([g, g], y, scale)


@pytest.mark.gpu
def test_scaled_gradient_path_of_the_epilogue_convolution_equals_the_unscaled_one():
    from monosowa_amd.monodetr import backbone as bb
    from monosowa_amd.monodetr.backbone import FrozenBatchNorm2d
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(256, 64, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    bn = FrozenBatchNorm2d(64).cuda()
    bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    x0 = torch.randn(2, 256, 24, 80, device="cuda").contiguous(memory_format=torch.channels_last)
    go = torch.randn(2, 64, 24, 80, device="cuda").contiguous(memory_format=torch.channels_last)
    res = {}
    for flag in (1, 0):
        bb.CONV1X1_SCALED_GRAD = flag
        try:
            x = x0.clone().requires_grad_(True)
            conv.weight.grad = None
            y = bb.conv_bn(x, conv, bn, None)
            y.backward(go)
            res[flag] = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone())
        finally:
            bb.CONV1X1_SCALED_GRAD = 1
    for a, b, name in zip(res[1], res[0], ("y", "dx", "dw")):
        assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1.0), name
    # and against float64
    xd = x0.double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    scale, shift = bn.scale_shift()
    yd = torch.relu(torch.nn.functional.conv2d(xd, wd) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    yd.backward(go.double())
    assert (res[1][1].double() - xd.grad).abs().max().item() <= 2e-5 * xd.grad.abs().max().item()
    assert (res[1][2].double() - wd.grad).abs().max().item() <= 2e-5 * wd.grad.abs().max().item()
