"""Weight gradient of the preNet's full-resolution convolutions (pose_hrnet.py:431-442: 3 -> 64 3x3, 64 -> 3 7x7, 3 -> 3 7x7,
stride 1, 'same' padding): the thin-channel kernel of conv.hip against an fp64 autograd evaluation."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Ci,Co,k,N,H,W", [(3, 64, 3, 2, 192, 256), (64, 3, 7, 2, 192, 256), (3, 3, 7, 3, 160, 144),
                                             (4, 48, 3, 2, 181, 203), (32, 2, 7, 1, 300, 250)])
def test_thin_wgrad_vs_fp64(Ci, Co, k, N, H, W):
    from buctd_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Ci * 100 + Co + k)
    x = torch.randn(N, Ci, H, W, generator=g, dtype=torch.float64)
    dy = torch.randn(N, Co, H, W, generator=g, dtype=torch.float64)
    w = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, 1, k // 2).backward(dy)
    ref = w.grad                                                       # [Co][Ci][k][k]
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)           # NHWC
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
    wd = torch.zeros(Co, Ci, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    out = torch.empty_like(wd)
    ops.conv_wgrad(xd, dyd, wd, 1, k // 2, out=out, accumulate=0)
    err = (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-5, f"thin wgrad {Ci}->{Co} k{k}: rel err {err:.2e}"
    # accumulate on top of an existing gradient
    base = torch.randn(Co, Ci, k, k, generator=g).contiguous(memory_format=torch.channels_last).to(dev)
    acc = base.clone()
    ops.conv_wgrad(xd, dyd, wd, 1, k // 2, out=acc, accumulate=1)
    err = (acc.cpu().double() - (base.cpu().double() + ref)).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-5
