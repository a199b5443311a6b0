"""Weight gradient of the 3x3/s1/p1 convolution on the bf16 matrix cores (buctd_amd/csrc/conv3x3_wgrad.hip) against an
fp64 evaluation of torch autograd on the CPU.  Bars relative to the gradient scale: bf16x6 (fp32-class, the default)
and the exact fp32 kernel 3e-6 (the reduction runs over up to 2e5 positions in fp32), bf16x3 5e-5."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 24, 18, 48, 48), (3, 12, 9, 384, 384), (2, 17, 13, 96, 96), (2, 20, 14, 64, 64), (4, 6, 5, 192, 192),
          (2, 9, 7, 32, 128), (2, 13, 11, 48, 96), (8, 96, 72, 48, 48), (2, 11, 10, 64, 256), (32, 12, 9, 96, 48),
          (1, 2, 2, 48, 48), (2, 3, 73, 32, 32),
          # ragged last 48-channel chunk of the input (the 256 -> 48 transition of HRNet-W48 takes this path)
          (2, 10, 9, 256, 48), (2, 7, 6, 64, 48), (1, 5, 4, 80, 96)]
TOL = {"bf16x6": 3e-6, "fp32": 1e-5, "bf16x3": 5e-5}


@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3", "fp32"])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_wgrad(dev, mode, shape):
    from buctd_amd import ops
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    old = ops.get_conv_math()
    ops.set_conv_math(mode)
    try:
        d = ops.conv_desc((N, H, W, Ci), (Co, Ci, 3, 3), 1, 1)
        if mode != "fp32":
            assert getattr(ops.lib(), f"buctd_conv3x3_wgrad_{mode}_supported")(d.N, d.H, d.W, d.Ci, d.Co) == 1
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        wd = w.detach().float().contiguous(memory_format=torch.channels_last).to(dev)
        dw = ops.conv_wgrad(xd, dyd, wd, 1, 1)
        sc = w.grad.abs().max().item()
        err = (dw.cpu().double() - w.grad).abs().max().item()
        assert err <= TOL[mode] * sc, f"wgrad {shape} [{mode}]: {err:.3e} vs scale {sc:.2f}"
        dw2 = ops.conv_wgrad(xd, dyd, wd, 1, 1, out=dw.clone(), accumulate=1)
        assert (dw2.cpu().double() - 2 * w.grad).abs().max().item() <= 2 * TOL[mode] * sc + 1e-6 * sc
    finally:
        ops.set_conv_math(old)
