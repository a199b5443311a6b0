"""x6 planes (buctd_amd/csrc/x6p.h): pre-split, zero-padded activations for the bf16x6 3x3 kernels.
  * the planes round-trip is exact and pads / guards are zero;
  * the forward / data-gradient kernel fed from planes (buctd_conv3x3_bf16x6_p) is BIT-IDENTICAL to the fp32-input kernel
    wherever both use the same tile, and within 2e-6 of fp64 everywhere;
  * the planes weight gradient (buctd_conv3x3_wgrad_bf16x6_p, LDS-DMA staging) is within 3e-6 of an fp64 evaluation of
    torch autograd (reference: nn.Conv2d of lib/models/pose_hrnet.py:28-57)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 24, 18, 48, 48), (3, 12, 9, 384, 384), (2, 17, 13, 96, 96), (4, 6, 5, 192, 192), (2, 13, 11, 48, 96),
          (8, 96, 72, 48, 48), (32, 12, 9, 96, 48), (1, 2, 2, 48, 48), (2, 3, 73, 48, 48), (2, 20, 14, 64, 64),
          (20, 96, 72, 48, 48)]


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("shape", [(2, 5, 7, 48), (1, 1, 1, 16), (3, 12, 9, 384), (2, 96, 72, 48)])
def test_planes_round_trip_and_padding(dev, shape):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * torch.exp2(torch.randint(-20, 20, shape, generator=g).float())).to(dev)
    pl = ops.to_planes(x)
    assert torch.equal(ops.from_planes(pl), x)          # h + m + l == x exactly
    # pads and guards: everything that is not a pixel is zero
    N, H, W, Cn = shape
    SW, IB = W + 2, (H + 1) * (W + 2)
    rows = pl.buf.numel() // (6 * Cn)
    raw = pl.buf.view(rows, 6 * Cn)
    real = torch.zeros(rows, dtype=torch.bool, device=dev)
    n = torch.arange(N, device=dev).view(N, 1, 1)
    y = torch.arange(H, device=dev).view(1, H, 1)
    xx = torch.arange(W, device=dev).view(1, 1, W)
    real[(128 + n * IB + (y + 1) * SW + xx + 1).flatten()] = True
    assert raw[~real].abs().max().item() == 0
    # fused producer BatchNorm + ReLU: same value as bn_apply
    mean, invstd = torch.randn(Cn, device=dev), torch.rand(Cn, device=dev) + 0.5
    gamma, beta = torch.randn(Cn, device=dev), torch.randn(Cn, device=dev)
    pl2 = ops.to_planes(x, bn=(mean, invstd, gamma, beta, True))
    assert torch.equal(ops.from_planes(pl2), ops.bn_apply(x, mean, invstd, gamma, beta, None, True))


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_from_planes(dev, shape):
    from buctd_amd import ops
    assert ops.get_conv_math() == "bf16x6"
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)
    b = torch.randn(Co, generator=g)
    y_ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    xd = nhwc(x).to(dev)
    wd = w.contiguous(memory_format=torch.channels_last).to(dev)
    xp = ops.to_planes(xd)
    y, part, info = ops.conv3x3_planes(xp, wd, 0, Co, bias=b.to(dev), stats=True)
    sc = y_ref.abs().max().item()
    err = (nchw(y).cpu().double() - y_ref).abs().max().item()
    assert err <= 2e-6 * sc, f"fwd {shape}: {err:.3e} vs scale {sc:.2f}"
    y0, part0, info0 = ops.conv_fwd(xd, wd, b.to(dev), 1, 1, stats=True)
    assert torch.equal(y, y0)                            # same MFMA order, same split: bit-identical
    # BatchNorm statistics of the epilogue: the finalized moments agree with the fp32-input kernel's
    m1, i1 = ops.bn_finalize(part, info, N * H * W, Co, 1e-5, 0.1, None, None)
    m0, i0 = ops.bn_finalize(part0, info0, N * H * W, Co, 1e-5, 0.1, None, None)
    assert (m1 - m0).abs().max().item() <= 1e-6 * max(1.0, m0.abs().max().item())
    assert ((i1 - i0) / i0).abs().max().item() <= 1e-5
    # data gradient (flip) with the skip gradient joining in the epilogue
    dy = torch.randn(N, Co, H, W, generator=g)
    dyd = nhwc(dy).to(dev)
    res = torch.randn(N, H, W, Ci, generator=g).to(dev)
    dx = ops.conv3x3_planes(ops.to_planes(dyd), wd, 1, Ci, residual=res)
    dx0 = ops.conv_dgrad(dyd, wd, tuple(xd.shape), 1, 1, residual=res)
    assert torch.equal(dx, dx0)


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[3] % 48 == 0 and s[4] % 48 == 0])
def test_conv3x3_wgrad_from_planes(dev, shape):
    from buctd_amd import ops
    N, H, W, Ci, Co = shape
    assert ops.wgrad_planes_ok(N, H, W, Ci, Co)
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    xd, dyd = nhwc(x).to(dev), nhwc(dy).to(dev)
    wd = w.detach().float().contiguous(memory_format=torch.channels_last).to(dev)
    xp, dyp = ops.to_planes(xd), ops.to_planes(dyd)
    dw = ops.conv_wgrad_planes(xp, dyp, torch.empty_like(wd))
    sc = w.grad.abs().max().item()
    err = (dw.cpu().double() - w.grad).abs().max().item()
    assert err <= 3e-6 * sc, f"wgrad {shape}: {err:.3e} vs scale {sc:.2f}"
    dw2 = ops.conv_wgrad_planes(xp, dyp, dw.clone(), accumulate=1)
    assert (dw2.cpu().double() - 2 * w.grad).abs().max().item() <= 7e-6 * sc
    # deterministic: fixed summation order
    assert torch.equal(ops.conv_wgrad_planes(xp, dyp, torch.empty_like(wd)), dw)
