"""The "bf16x3" convolution math mode (buctd_amd/csrc/conv3x3.hip): 3x3/s1/p1 convolutions on the bf16 matrix cores
with split-fp32 operands.  Bars: a single convolution within 5e-5 of the fp32 result relative to the output scale
(product error ~2^-16); whole networks within the north_star bar 1e-3 * max(1, max|ref|) with identical arg-max."""
import math
import os
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.fixture
def bf16x3():
    from buctd_amd import ops
    old = ops.get_conv_math()
    ops.set_conv_math("bf16x3")
    yield
    ops.set_conv_math(old)


SHAPES = [(2, 24, 18, 48, 48), (3, 12, 9, 384, 384), (2, 17, 13, 96, 96), (2, 20, 14, 64, 64), (4, 6, 5, 192, 192),
          (2, 10, 8, 16, 16), (2, 9, 7, 32, 128), (2, 13, 11, 48, 96), (8, 96, 72, 48, 48), (2, 11, 10, 64, 256)]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_bf16x3_fwd_dgrad(dev, bf16x3, shape):
    from buctd_amd import ops
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Ci, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).requires_grad_(True)
    b = torch.randn(Co, generator=g)
    y_ref = F.conv2d(x, w, b, 1, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    assert ops._bf16x3_ok(ops.conv_desc((N, H, W, Ci), (Co, Ci, 3, 3), 1, 1)), "shape should take the bf16x3 path"
    xd = nhwc(x.detach()).to(dev)
    wd = w.detach().contiguous(memory_format=torch.channels_last).to(dev)
    y = ops.conv_fwd(xd, wd, b.to(dev), 1, 1)
    sc = y_ref.abs().max().item()
    err = (nchw(y).cpu() - y_ref).abs().max().item()
    assert err <= 5e-5 * sc, f"fwd {shape}: {err:.3e} vs scale {sc:.2f}"
    dx = ops.conv_dgrad(nhwc(dy).to(dev), wd, tuple(xd.shape), 1, 1)
    sc = x.grad.abs().max().item()
    err = (nchw(dx).cpu() - x.grad).abs().max().item()
    assert err <= 5e-5 * sc, f"dgrad {shape}: {err:.3e} vs scale {sc:.2f}"


def test_conv3x3_bf16x3_epilogues(dev, bf16x3):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(9)
    N, H, W, Ci, Co = 3, 13, 11, 48, 96
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    res = torch.randn(N, Co, H, W, generator=g)
    scale, shift = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    zr = F.conv2d(x, w, None, 1, 1)
    ref = F.relu(zr * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    xd, wd = nhwc(x).to(dev), w.contiguous(memory_format=torch.channels_last).to(dev)
    y = ops.conv_fwd(xd, wd, None, 1, 1, scale=scale.to(dev), shift=shift.to(dev), residual=nhwc(res).to(dev), relu=True)
    assert (nchw(y).cpu() - ref).abs().max().item() <= 5e-5 * ref.abs().max().item()
    z, part, info = ops.conv_fwd(xd, wd, None, 1, 1, stats=True)
    assert len(info) == 3 and int(info[2].sum()) == N * H * W, "valid-row counts must add up to N*H*W"
    rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    mean, invstd = ops.bn_finalize(part, info, N * H * W, Co, 1e-5, 0.1, rm, rv)
    assert (mean.cpu() - zr.mean((0, 2, 3))).abs().max().item() <= 2e-5
    ref_is = 1.0 / torch.sqrt(zr.var((0, 2, 3), unbiased=False) + 1e-5)
    assert ((invstd.cpu() - ref_is).abs() / ref_is).max().item() <= 5e-5
    rv_ref = 0.9 + 0.1 * zr.var((0, 2, 3), unbiased=True)
    assert ((rv.cpu() - rv_ref).abs() / rv_ref).max().item() <= 5e-5


@pytest.mark.parametrize("name", ["prenet_w16_96x64", "coam_w16_96x64_colored", "coam_w48_384x288"])
def test_networks_in_bf16x3_mode(dev, bf16x3, name):
    from oracle import recipes
    from buctd_amd import models
    path = os.path.join(GOLD, f"model_{name}.npz")
    cfg, omodel, x, joints = recipes.build(name)
    if os.path.isfile(path):
        ref = np.load(path)["out"]
    else:
        with torch.no_grad():
            ref = omodel(x).numpy()
    m = getattr(models, cfg.MODEL.NAME).get_pose_net(cfg, is_train=False)
    m.load_state_dict(omodel.state_dict(), strict=True)
    m = m.to(dev).eval()
    with torch.no_grad():
        y = m(x.to(dev)).cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(y - ref).max()
    print(f"{name} [bf16x3]: |hip - reference| = {err:.3e} (scale {scale:.1f}, {err / scale:.2e} relative)")
    assert err <= 1e-3 * scale
    assert np.array_equal(y.reshape(y.shape[0], y.shape[1], -1).argmax(2), ref.reshape(ref.shape[0], ref.shape[1], -1).argmax(2))
    if name.endswith("96x64") or name.endswith("colored"):
        # one train step: loss within 1e-3 relative of the reference's value
        from buctd_amd.core.loss import JointsMSELoss
        gold = np.load(path)
        tgt, wt = recipes.make_targets(cfg, joints, 77)
        m.train()
        recipes.set_dropout(m, 0.0)
        loss = JointsMSELoss(True)(m(x.to(dev)), tgt.to(dev), wt.to(dev))
        loss.backward()
        assert abs(loss.item() - float(gold["loss"])) <= 1e-3 * abs(float(gold["loss"]))
        names = [str(s) for s in gold["grad_names"]]
        params = dict(m.named_parameters())
        gmax = float(gold["grad_norms"].max())
        bad = [k for k, gn in zip(names, gold["grad_norms"])
               if gn > 1e-6 * gmax and abs(params[k].grad.norm().item() - gn) > 5e-2 * gn]  # skip mathematically-zero grads
        assert not bad, f"gradient norms off: {bad[:5]}"


def test_conv3x3_speed_report(dev):
    """Not a pass/fail bar: prints the stage-4 branch-0 conv timing in both math modes (HIP events)."""
    from buctd_amd import ops
    N, H, W, C = 32, 96, 72, 48
    x = torch.randn(N, H, W, C, device=dev)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    flops = 2.0 * N * H * W * C * C * 9
    for mode in ("fp32", "bf16x3"):
        ops.set_conv_math(mode)
        for _ in range(3):
            ops.conv_fwd(x, w, None, 1, 1, stats=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv_fwd(x, w, None, 1, 1, stats=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"conv 3x3 48->48 @96x72 N=32 [{mode}]: {us:.1f} us/launch, {flops / us / 1e6:.1f} TFLOP/s-equivalent, "
              f"{85.0e6 / us / 1e3:.0f} GB/s algorithmic")
    ops.set_conv_math("fp32")
