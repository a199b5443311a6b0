"""The gathered bf16x6 convolutions (buctd_amd/csrc/conv_gather_x6.hip): 1x1 and stride-2 3x3, forward (with the fused
epilogue and BatchNorm statistics) and data gradient (four output parities of a stride-2 convolution in one launch),
against fp64 evaluations of torch's conv2d - the 2e-6 bar of the 3x3 stride-1 bf16x6 kernel - and against the exact-fp32
kernels they replace (reference layers: lib/models/pose_hrnet.py:60-108, 187-245, 338-372)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-6


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


# (N, H, W, Ci, Co, k, stride, pad)
SHAPES = [(2, 24, 18, 48, 96, 3, 2, 1), (2, 12, 10, 96, 192, 3, 2, 1), (3, 8, 6, 192, 384, 3, 2, 1), (2, 16, 12, 48, 48, 3, 2, 1),
          (2, 20, 14, 64, 64, 3, 2, 1), (1, 6, 4, 256, 96, 3, 2, 1), (2, 2, 2, 48, 192, 3, 2, 1), (4, 96, 72, 48, 96, 3, 2, 1),
          (2, 24, 18, 64, 256, 1, 1, 0), (2, 24, 18, 256, 64, 1, 1, 0), (3, 12, 9, 96, 48, 1, 1, 0), (2, 7, 5, 192, 96, 1, 1, 0),
          (2, 6, 5, 384, 192, 1, 1, 0), (1, 1, 1, 48, 48, 1, 1, 0), (2, 48, 36, 96, 48, 1, 1, 0),
          # >= 65536 rows with 64 / 128 / 256 output channels: the row-streaming 1x1 kernel (conv1x1_rows_x6_kernel), incl. a
          # ragged last 32-row block
          (16, 64, 64, 64, 256, 1, 1, 0), (17, 64, 62, 256, 64, 1, 1, 0), (8, 96, 96, 128, 128, 1, 1, 0), (16, 65, 65, 64, 64, 1, 1, 0)]


@pytest.mark.parametrize("shape", SHAPES)
def test_gconv_fwd_dgrad_vs_fp64(dev, shape):
    from buctd_amd import ops
    N, H, W, Ci, Co, k, st, pad = shape
    assert ops.get_conv_math() == "bf16x6"
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Ci, H, W, generator=g).double().requires_grad_(True)
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).double().requires_grad_(True)
    b = torch.randn(Co, generator=g).double()
    y_ref = F.conv2d(x, w, b, st, pad)
    dy = torch.randn(y_ref.shape, generator=g).double()
    y_ref.backward(dy)
    d = ops.conv_desc((N, H, W, Ci), (Co, Ci, k, k), st, pad)
    assert ops._gconv_ok(d, 0) and ops._gconv_ok(d, 1), "shape should take the gathered bf16x6 kernel"
    xd = nhwc(x.detach().float()).to(dev)
    wd = w.detach().float().contiguous(memory_format=torch.channels_last).to(dev)
    y = ops.conv_fwd(xd, wd, b.float().to(dev), st, pad)
    sc = y_ref.abs().max().item()
    err = (nchw(y).cpu().double() - y_ref).abs().max().item()
    assert err <= TOL * sc, f"fwd {shape}: {err:.3e} vs scale {sc:.2f}"
    dyd = nhwc(dy.float()).to(dev)
    dx = ops.conv_dgrad(dyd, wd, tuple(xd.shape), st, pad)
    sc = x.grad.abs().max().item()
    err = (nchw(dx).cpu().double() - x.grad).abs().max().item()
    assert err <= TOL * sc, f"dgrad {shape}: {err:.3e} vs scale {sc:.2f}"
    res = torch.randn(xd.shape, generator=g).to(dev)
    dx2 = ops.conv_dgrad(dyd, wd, tuple(xd.shape), st, pad, residual=res)
    assert (dx2 - dx - res).abs().max().item() <= 1e-6 * max(1.0, sc)
    # against the exact-fp32 kernels these replace
    ops.set_conv_math("fp32")
    try:
        y32 = ops.conv_fwd(xd, wd, b.float().to(dev), st, pad)
        dx32 = ops.conv_dgrad(dyd, wd, tuple(xd.shape), st, pad)
    finally:
        ops.set_conv_math("bf16x6")
    assert (y32 - y).abs().max().item() <= 1e-5 * y_ref.abs().max().item()
    assert (dx32 - dx).abs().max().item() <= 1e-5 * sc


@pytest.mark.parametrize("shape", [(2, 24, 18, 48, 96, 3, 2, 1), (3, 12, 10, 192, 48, 1, 1, 0), (2, 10, 8, 64, 256, 1, 1, 0)])
def test_gconv_epilogue_and_statistics(dev, shape):
    """eval-BN scale/shift + residual + ReLU in the epilogue, and the train-mode BatchNorm statistics (Welford partials +
    valid-row counts of the padded position tiles) finalised by buctd_bn_finalize."""
    from buctd_amd import ops
    N, H, W, Ci, Co, k, st, pad = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(N, H, W, Ci, generator=g).to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).contiguous(memory_format=torch.channels_last).to(dev)
    y0 = ops.conv_fwd(x, w, None, st, pad)
    scale = (torch.rand(Co, generator=g) + 0.5).to(dev)
    shift = torch.randn(Co, generator=g).to(dev)
    res = torch.randn(y0.shape, generator=g).to(dev)
    y1 = ops.conv_fwd(x, w, None, st, pad, scale=scale, shift=shift, residual=res, relu=True)
    assert (y1 - torch.relu(y0 * scale + shift + res)).abs().max().item() <= 2e-6 * max(1.0, y0.abs().max().item())
    y2, part, info = ops.conv_fwd(x, w, None, st, pad, stats=True)
    assert torch.equal(y2, y0) and len(info) == 3
    rows = y0.numel() // Co
    assert int(info[2].sum().item()) == rows
    rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    mean, invstd = ops.bn_finalize(part, info, rows, Co, 1e-5, 0.1, rm, rv)
    z = y0.reshape(rows, Co).double()
    assert (mean.double() - z.mean(0)).abs().max().item() <= 1e-6
    assert (invstd.double() - 1.0 / torch.sqrt(z.var(0, unbiased=False) + 1e-5)).abs().max().item() <= 1e-5


def test_gconv_dispatch_boundaries(dev):
    """odd input sizes of a stride-2 convolution, thin channels and the fp32 math mode stay on the exact-fp32 kernels"""
    from buctd_amd import ops
    assert not ops._gconv_ok(ops.conv_desc((2, 9, 7, 48), (96, 48, 3, 3), 2, 1), 0)
    assert not ops._gconv_ok(ops.conv_desc((2, 384, 288, 3), (64, 3, 3, 3), 2, 1), 0)
    assert not ops._gconv_ok(ops.conv_desc((2, 8, 8, 48), (32, 48, 1, 1), 1, 0), 0)       # 32 output channels
    assert ops._gconv_ok(ops.conv_desc((2, 8, 8, 32), (48, 32, 1, 1), 1, 0), 0)
    ops.set_conv_math("fp32")
    try:
        assert not ops._gconv_ok(ops.conv_desc((2, 8, 8, 48), (96, 48, 3, 3), 2, 1), 0)
    finally:
        ops.set_conv_math("bf16x6")
    # data-gradient image of a stride-2 filter: four parity classes of 1 + 2 + 2 + 4 taps = the nine taps once
    lib = ops.lib()
    per_tap = 96 // 16 * 48 * 192 // 2           # (Co / 16 half-steps) x Ci rows x 192 B / 2 half-steps per step
    assert lib.buctd_gconv_x6_prep_bytes(2, 48, 96, 1) == 9 * per_tap
    assert lib.buctd_gconv_x6_prep_bytes(2, 48, 96, 0) == (9 * 3 + 1) // 2 * 96 * 192
    assert lib.buctd_gconv_x6_prep_bytes(3, 48, 96, 0) == 0 and lib.buctd_gconv_x6_prep_bytes(1, 40, 96, 0) == 0


def test_prepared_images_follow_in_place_weight_updates(dev):
    """An optimizer kernel rewrites filters through raw pointers (no autograd version bump): weights_updated() +
    refresh_prepared() rebuild every registered image - the gathered kernels' and the 3x3 stride-1 kernels' - with one launch
    each, and the next convolutions see the new filters (forward and data gradient)."""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 12, 8, 48, generator=g).to(dev)
    ws = [(torch.randn(96, 48, 3, 3, generator=g) / 20).contiguous(memory_format=torch.channels_last).to(dev),
          (torch.randn(48, 48, 1, 1, generator=g) / 7).contiguous(memory_format=torch.channels_last).to(dev),
          (torch.randn(48, 48, 3, 3, generator=g) / 20).contiguous(memory_format=torch.channels_last).to(dev)]
    cfgs = [(2, 1), (1, 0), (1, 1)]

    def run():
        outs = []
        for w, (st, pad) in zip(ws, cfgs):
            y = ops.conv_fwd(x, w, None, st, pad)
            outs += [y, ops.conv_dgrad(torch.ones_like(y), w, tuple(x.shape), st, pad)]
        return outs
    before = run()
    for w in ws:
        ptr_before, ver = w.data_ptr(), w._version
        w.data.mul_(1.5).add_(0.01)      # in place, like the fused optimizer kernel (through .data: no version bump)
        assert w.data_ptr() == ptr_before and w._version == ver
    stale = run()
    assert all(torch.equal(a, b) for a, b in zip(before, stale))          # images are cached on (pointer, version, epoch)
    ops.weights_updated()
    ops.refresh_prepared(dev)
    fresh = run()
    ref = []
    for w, (st, pad) in zip(ws, cfgs):
        w2 = w.detach().clone().contiguous(memory_format=torch.channels_last)
        y = ops.conv_fwd(x, w2, None, st, pad)
        ref += [y, ops.conv_dgrad(torch.ones_like(y), w2, tuple(x.shape), st, pad)]
    assert all(torch.equal(a, b) for a, b in zip(fresh, ref))
    assert not torch.equal(fresh[0], before[0])


# (N, H, W, Ci, Co, k): the weight gradients of the same layers on the bf16x6 kernel (csrc/conv_gather_wgrad.hip)
WG_SHAPES = [(2, 24, 18, 48, 96, 3), (2, 12, 10, 96, 192, 3), (3, 8, 6, 192, 384, 3), (2, 16, 12, 48, 48, 3), (2, 20, 14, 64, 64, 3),
             (1, 6, 4, 256, 96, 3), (2, 4, 4, 48, 192, 3), (4, 96, 72, 48, 96, 3), (2, 24, 18, 64, 256, 1), (2, 24, 18, 256, 64, 1),
             (3, 12, 9, 96, 48, 1), (2, 7, 5, 192, 96, 1), (2, 6, 5, 384, 192, 1), (1, 1, 2, 48, 48, 1), (2, 48, 36, 96, 48, 1),
             (1, 5, 3, 32, 64, 1), (32, 24, 18, 192, 48, 1), (16, 64, 64, 64, 128, 1), (17, 64, 62, 256, 64, 1)]


@pytest.mark.parametrize("shape", WG_SHAPES)
def test_gconv_weight_gradient_vs_fp64(dev, shape):
    """buctd_gconv_wgrad_x6 (every wave stages, splits and multiplies its own 32-pixel k-steps; partial slabs added in a fixed
    order) against torch's fp64 conv2d_weight: <= 2e-6 of the largest element; accumulation into an existing gradient; run-to-run
    bit-reproducible."""
    from buctd_amd import ops, _C
    N, H, W, Ci, Co, k = shape
    stride, pad = (1, 0) if k == 1 else (2, 1)
    kind = 1 if k == 1 else 2
    assert _C.lib().buctd_gconv_wgrad_x6_supported(kind, N, H, W, Ci, Co) == 1
    g = torch.Generator().manual_seed(sum(shape))
    Ho, Wo = (H, W) if k == 1 else (H // 2, W // 2)
    x = torch.randn(N, H, W, Ci, generator=g).to(dev)
    dy = torch.randn(N, Ho, Wo, Co, generator=g).to(dev)
    w_like = torch.empty(Co, Ci, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    dw = ops.conv_wgrad(x, dy, w_like, stride, pad)
    dw2 = ops.conv_wgrad(x, dy, w_like, stride, pad)
    base = torch.randn(Co, Ci, k, k, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    acc = base.clone(memory_format=torch.preserve_format)
    ops.conv_wgrad(x, dy, w_like, stride, pad, out=acc, accumulate=1)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2).cpu(), (Co, Ci, k, k), dy.double().permute(0, 3, 1, 2).cpu(),
                                      stride=stride, padding=pad)
    sc = ref.abs().max().item()
    assert torch.equal(dw, dw2), "not run-to-run reproducible"
    assert (dw.cpu().double() - ref).abs().max().item() <= TOL * sc
    assert (acc.cpu().double() - base.cpu().double() - ref).abs().max().item() <= 2 * TOL * max(sc, base.abs().max().item())


def test_gconv_weight_gradient_dispatch(dev):
    """channel counts without a bf16x6 tile (not multiples of 32 or 48) and odd stride-2 inputs stay on the exact-fp32 kernel"""
    from buctd_amd import _C
    lib = _C.lib()
    assert lib.buctd_gconv_wgrad_x6_supported(1, 2, 8, 8, 48, 16) == 0
    assert lib.buctd_gconv_wgrad_x6_supported(2, 2, 9, 8, 48, 48) == 0
    assert lib.buctd_gconv_wgrad_x6_supported(2, 2, 8, 8, 64, 256) == 1
    assert lib.buctd_gconv_wgrad_x6_supported(3, 2, 8, 8, 48, 48) == 0


@pytest.mark.parametrize("shape", [(16, 64, 65, 64, 256), (9, 96, 80, 256, 64)])
def test_rows_kernel_statistics_accumulator(dev, shape):
    """conv1x1_rows_x6_kernel (1x1 convolutions on >= 65536 rows): the BatchNorm statistics it adds to the accumulator, through
    the consumer that decodes them (bn_apply_acc), against an fp64 BatchNorm of the fp64 convolution; ragged last row block."""
    import torch.nn as tnn
    from buctd_amd import ops
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, H, W, Ci, generator=g).to(dev) + 0.3
    w = (torch.randn(Co, Ci, 1, 1, generator=g) / math.sqrt(Ci)).contiguous(memory_format=torch.channels_last).to(dev)
    bn = tnn.BatchNorm2d(Co).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    z, acc, info = ops.conv_fwd(x, w, None, 1, 0, stats="acc")
    assert info[0] == "acc"
    rows = N * H * W
    bnin = ops.BnAccInput(acc, rows, bn, True, relu=False)
    y = ops.bn_apply_acc(z, bnin, None, False)
    torch.cuda.synchronize()
    zr = F.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().cpu())
    yr = F.batch_norm(zr, None, None, bn.weight.double().cpu(), bn.bias.double().cpu(), True, 0.1, bn.eps).permute(0, 2, 3, 1)
    assert (z.cpu().double() - zr.permute(0, 2, 3, 1)).abs().max().item() <= TOL * zr.abs().max().item()
    assert (y.cpu().double() - yr).abs().max().item() <= 2e-5 * max(1.0, yr.abs().max().item())
    zz = zr.permute(0, 2, 3, 1).reshape(rows, Co)
    assert (bnin.mean.cpu().double() - zz.mean(0)).abs().max().item() <= 1e-6
    assert (bn.running_var.cpu().double() - (0.9 + 0.1 * zz.var(0, unbiased=True))).abs().max().item() <= 1e-5
