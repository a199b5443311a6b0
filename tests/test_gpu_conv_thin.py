"""Weight gradient of the preNet's full-resolution convolutions (pose_hrnet.py:431-442: 3 -> 64 3x3, 64 -> 3 7x7, 3 -> 3 7x7,
stride 1, 'same' padding): the thin-channel kernel of conv.hip against an fp64 autograd evaluation."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Ci,Co,k,N,H,W", [(3, 64, 3, 2, 192, 256), (64, 3, 7, 2, 192, 256), (3, 3, 7, 3, 160, 144),
                                             (4, 48, 3, 2, 181, 203), (32, 2, 7, 1, 300, 250),
                                             # 3x3 from <= 4 channels to 64, W a multiple of 16: lane = output channel
                                             (4, 64, 3, 1, 70, 144), (1, 64, 3, 2, 33, 128), (2, 64, 3, 1, 50, 80),
                                             (3, 64, 3, 40, 40, 64),
                                             # 64 -> <= 3 channels, 7x7: the rolling-row kernel (ragged strips, short images, segments)
                                             (64, 2, 7, 1, 70, 90), (64, 3, 7, 1, 33, 200), (64, 1, 7, 2, 50, 64), (64, 3, 7, 3, 17, 100)])
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


@pytest.mark.parametrize("Ci,Co,N,H,W", [(64, 3, 2, 192, 256), (3, 3, 3, 160, 144), (32, 2, 1, 300, 250), (17, 4, 2, 181, 203),
                                         (8, 1, 2, 70, 45), (64, 3, 2, 65, 63), (16, 3, 3, 15, 130),
                                         # thin on both sides: the four-pixel kernel with 4-channel pixels
                                         (3, 2, 2, 70, 90), (2, 3, 1, 95, 77), (4, 1, 2, 64, 80), (1, 3, 1, 66, 70)])
def test_thin_forward_vs_fp64(Ci, Co, N, H, W):
    """7x7 'same' convolutions with <= 4 output channels (preNet): forward incl. bias and the BatchNorm partials"""
    from buctd_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Ci * 10 + Co)
    x = torch.randn(N, Ci, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, Ci, 7, 7, generator=g, dtype=torch.float64) * (49 * Ci) ** -0.5
    b = torch.randn(Co, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, b, 1, 3).permute(0, 2, 3, 1)
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.float().contiguous(memory_format=torch.channels_last).to(dev)
    bd = b.float().to(dev)
    d = ops.conv_desc(xd.shape, ops._wshape(wd), 1, 3)
    import ctypes as C
    from buctd_amd._C import lib
    assert lib().buctd_conv2d_fwd_thin(C.byref(d)) == 1
    y = ops.conv_fwd(xd, wd, bd, 1, 3)
    err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-6, f"thin forward {Ci}->{Co}: rel err {err:.2e}"
    y2, part, info = ops.conv_fwd(xd, wd, bd, 1, 3, stats=True)
    assert torch.equal(y, y2)
    rows = N * H * W
    mean, invstd = ops.bn_finalize(part, info, rows, Co, 1e-5, 0.1, None, None)
    rm = ref.reshape(-1, Co).mean(0)
    rv = ref.reshape(-1, Co).var(0, unbiased=False)
    assert (mean.cpu().double() - rm).abs().max().item() <= 1e-5
    assert ((invstd.cpu().double() - (rv + 1e-5).rsqrt()).abs() / (rv + 1e-5).rsqrt()).max().item() <= 1e-5


@pytest.mark.parametrize("Ci,Co,N,H,W", [(64, 3, 2, 192, 256), (3, 3, 3, 160, 144), (32, 2, 1, 300, 250), (17, 4, 2, 181, 203),
                                         (2, 4, 2, 70, 90), (3, 2, 1, 95, 77), (1, 3, 2, 64, 80), (3, 1, 1, 66, 70)])
def test_thin_dgrad_vs_fp64(Ci, Co, N, H, W):
    from buctd_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Ci * 7 + Co)
    dy = torch.randn(N, Co, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, Ci, 7, 7, generator=g, dtype=torch.float64) * (49 * Co) ** -0.5
    x = torch.zeros(N, Ci, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, 1, 3).backward(dy)
    ref = x.grad.permute(0, 2, 3, 1)
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.float().contiguous(memory_format=torch.channels_last).to(dev)
    dx = ops.conv_dgrad(dyd, wd, (N, H, W, Ci), 1, 3)
    err = (dx.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-6, f"thin dgrad {Ci}->{Co}: rel err {err:.2e}"


@pytest.mark.parametrize("Ci,Co,N,H,W", [(3, 64, 2, 192, 256), (3, 64, 1, 97, 131), (4, 32, 2, 64, 80), (2, 128, 1, 120, 90)])
def test_thin_stride2_dgrad_vs_fp64(Ci, Co, N, H, W):
    """data gradient of a stride-2 3x3 convolution with <= 4 input channels (the stem conv1 behind a preNet), odd sizes too"""
    from buctd_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Ci * 3 + Co)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    dy = torch.randn(N, Co, Ho, Wo, generator=g, dtype=torch.float64)
    w = torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64) * (9 * Co) ** -0.5
    x = torch.zeros(N, Ci, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, 2, 1).backward(dy)
    ref = x.grad.permute(0, 2, 3, 1)
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.float().contiguous(memory_format=torch.channels_last).to(dev)
    dx = ops.conv_dgrad(dyd, wd, (N, H, W, Ci), 2, 1)
    err = (dx.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-6, f"thin stride-2 dgrad {Ci}->{Co} {H}x{W}: rel err {err:.2e}"


@pytest.mark.parametrize("Ci,N,H,W", [(3, 2, 192, 256), (4, 1, 70, 144), (1, 2, 33, 128), (2, 3, 41, 197), (3, 1, 64, 64), (3, 2, 50, 80)])
def test_thin_input_3x3_forward_vs_fp64(Ci, N, H, W):
    """3x3 'same' convolutions from <= 4 input channels to 64 (first preNet convolution): forward incl. bias and the
    BatchNorm partials formed in the same launch (W < 128 takes the general kernel)"""
    from buctd_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Ci * 31 + W)
    x = torch.randn(N, Ci, H, W, generator=g, dtype=torch.float64) + 0.3
    w = torch.randn(64, Ci, 3, 3, generator=g, dtype=torch.float64) * (9 * Ci) ** -0.5
    b = torch.randn(64, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, b, 1, 1).permute(0, 2, 3, 1)
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.float().contiguous(memory_format=torch.channels_last).to(dev)
    bd = b.float().to(dev)
    y = ops.conv_fwd(xd, wd, bd, 1, 1)
    err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-6, f"thin-input forward {Ci}->64: rel err {err:.2e}"
    y2, part, info = ops.conv_fwd(xd, wd, bd, 1, 1, stats=True)
    assert torch.equal(y, y2)
    rows = N * H * W
    mean, invstd = ops.bn_finalize(part, info, rows, 64, 1e-5, 0.1, None, None)
    rm = ref.reshape(-1, 64).mean(0)
    rv = ref.reshape(-1, 64).var(0, unbiased=False)
    assert (mean.cpu().double() - rm).abs().max().item() <= 2e-6 * ref.abs().max().item()
    assert ((invstd.cpu().double() - (rv + 1e-5).rsqrt()).abs() / (rv + 1e-5).rsqrt()).max().item() <= 1e-5


@pytest.mark.parametrize("Ci,N,H,W", [(3, 2, 192, 256), (4, 1, 140, 288), (1, 2, 66, 128), (3, 1, 64, 64), (2, 1, 130, 160)])
def test_thin_input_3x3_stride2_vs_fp64(Ci, N, H, W):
    """the stem's first convolution (<= 4 channels -> 64, 3x3, stride 2, pad 1): forward with bias and BatchNorm partials, and
    the weight gradient, on the lane-per-output-channel kernels where the output width is a multiple of 16 (else the general path)"""
    from buctd_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Ci * 17 + W)
    x = torch.randn(N, Ci, H, W, generator=g, dtype=torch.float64) - 0.2
    w = (torch.randn(64, Ci, 3, 3, generator=g, dtype=torch.float64) * (9 * Ci) ** -0.5).requires_grad_(True)
    b = torch.randn(64, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, b, 2, 1)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    refy = ref.detach().permute(0, 2, 3, 1)
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.detach().float().contiguous(memory_format=torch.channels_last).to(dev)
    y, part, info = ops.conv_fwd(xd, wd, b.float().to(dev), 2, 1, stats=True)
    err = (y.cpu().double() - refy).abs().max().item() / refy.abs().max().item()
    assert err <= 2e-6, f"thin-input stride-2 forward {Ci}->64: rel err {err:.2e}"
    rows = refy.numel() // 64
    mean, invstd = ops.bn_finalize(part, info, rows, 64, 1e-5, 0.1, None, None)
    rm, rv = refy.reshape(-1, 64).mean(0), refy.reshape(-1, 64).var(0, unbiased=False)
    assert (mean.cpu().double() - rm).abs().max().item() <= 2e-6 * refy.abs().max().item()
    assert ((invstd.cpu().double() - (rv + 1e-5).rsqrt()).abs() / (rv + 1e-5).rsqrt()).max().item() <= 1e-5
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty_like(wd)
    ops.conv_wgrad(xd, dyd, wd, 2, 1, out=out, accumulate=0)
    err = (out.cpu().double() - w.grad).abs().max().item() / w.grad.abs().max().item()
    assert err <= 2e-5, f"thin-input stride-2 wgrad {Ci}->64: rel err {err:.2e}"
