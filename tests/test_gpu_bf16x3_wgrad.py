"""bf16x3 weight gradient of the 3x3/s1/p1 convolution (buctd_amd/csrc/conv3x3_wgrad.hip) against torch autograd on
the CPU; bar 5e-5 of the gradient scale (the reduction runs over up to 2e5 positions)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 24, 18, 48, 48), (3, 12, 9, 384, 384), (2, 17, 13, 96, 96), (2, 20, 14, 64, 64), (4, 6, 5, 192, 192),
          (2, 9, 7, 32, 128), (2, 13, 11, 48, 96), (8, 96, 72, 48, 48), (2, 11, 10, 64, 256), (32, 12, 9, 96, 48)]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_wgrad_bf16x3(dev, shape):
    from buctd_amd import ops
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    old = ops.get_conv_math()
    ops.set_conv_math("bf16x3")
    try:
        d = ops.conv_desc((N, H, W, Ci), (Co, Ci, 3, 3), 1, 1)
        assert ops.lib().buctd_conv3x3_wgrad_bf16x3_supported(d.N, d.H, d.W, d.Ci, d.Co) == 1
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        wd = w.detach().contiguous(memory_format=torch.channels_last).to(dev)
        dw = ops.conv_wgrad(xd, dyd, wd, 1, 1)
        sc = w.grad.abs().max().item()
        err = (dw.cpu() - w.grad).abs().max().item()
        assert err <= 5e-5 * sc, f"wgrad {shape}: {err:.3e} vs scale {sc:.2f}"
        dw2 = ops.conv_wgrad(xd, dyd, wd, 1, 1, out=dw.clone(), accumulate=1)
        assert (dw2.cpu() - 2 * w.grad).abs().max().item() <= 1e-4 * sc
    finally:
        ops.set_conv_math(old)
