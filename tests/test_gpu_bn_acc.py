"""BatchNorm statistics without finalize launches (csrc/bn_acc.h): the producing kernel adds exact fixed-point sums into an
accumulator, the consuming kernel decodes them.  Parity against an fp64 computation of nn.BatchNorm2d in train mode
(reference lib/models/pose_hrnet.py:41-57), bit-determinism, the out-of-range / NaN poison, the zeroed pool."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def wcl(w):
    return w.contiguous(memory_format=torch.channels_last)


class _Bn:
    """the attributes ops.BnAccInput reads from a BatchNorm module"""

    def __init__(self, Cn, dev, eps=1e-5, momentum=0.1):
        g = torch.Generator().manual_seed(Cn)
        self.weight = (1.0 + 0.3 * torch.randn(Cn, generator=g)).to(dev)
        self.bias = (0.2 * torch.randn(Cn, generator=g)).to(dev)
        self.running_mean = torch.zeros(Cn, device=dev)
        self.running_var = torch.ones(Cn, device=dev)
        self.eps, self.momentum, self.track_running_stats = eps, momentum, True


# N, H, W, Ci, Co, kernel, stride: the 3x3 tile variants (48 / 96 / 32- / 64-column tiles, ragged position tiles, the
# 448-position tile of the full-resolution branch) and the gathered 1x1 / stride-2 kernels
CASES = [
    (2, 24, 18, 48, 48, 3, 1),
    (3, 17, 13, 96, 96, 3, 1),
    (2, 12, 9, 384, 384, 3, 1),
    (2, 14, 10, 64, 64, 3, 1),
    (2, 14, 10, 128, 128, 3, 1),
    (2, 14, 10, 32, 32, 3, 1),
    (8, 96, 72, 48, 48, 3, 1),
    (32, 96, 72, 48, 48, 3, 1),
    (2, 20, 20, 64, 256, 1, 1),
    (2, 24, 18, 48, 96, 3, 2),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("scale", [1.0, 3e3, 2e-4])
def test_forward_statistics_from_accumulator(dev, case, scale):
    """conv (statistics -> accumulator) -> bn_apply_acc: output, mean / invstd and running statistics vs fp64"""
    from buctd_amd import ops
    N, H, W, Ci, Co, k, stride = case
    if N >= 32 and scale != 1.0:
        pytest.skip("one scale is enough at the bench shape")
    g = torch.Generator().manual_seed(7 * Ci + Co + H)
    x = (torch.randn(N, Ci, H, W, generator=g) + 0.5) * scale
    w = torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (Ci * k * k) ** 0.5)
    pad = 1 if k == 3 else 0
    z, acc, info = ops.conv_fwd(nhwc(x).to(dev), wcl(w).to(dev), None, stride, pad, stats="acc")
    assert info[0] == "acc", "this shape should take the accumulator form"
    bn = _Bn(Co, dev)
    res = torch.randn(z.shape, generator=g).to(dev) * scale
    bnin = ops.BnAccInput(acc, z.numel() // Co, bn, True)
    y = ops.bn_apply_acc(z, bnin, res, True)
    # fp64 reference on the device's own z (the convolution has its own tests)
    zd = z.double().cpu().reshape(-1, Co)
    mu = zd.mean(0)
    var = zd.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + bn.eps)
    yr = torch.relu((zd - mu) * invstd * bn.weight.double().cpu() + bn.bias.double().cpu() + res.double().cpu().reshape(-1, Co))
    tol = 2e-6
    assert (bnin.mean.double().cpu() - mu).abs().max() <= tol * max(1.0, mu.abs().max()) * 1.0
    assert ((bnin.invstd.double().cpu() - invstd).abs() / invstd).max() <= 3e-7
    assert (y.double().cpu().reshape(-1, Co) - yr).abs().max() <= 2e-5 * max(1.0, yr.abs().max().item())
    rows = zd.shape[0]
    rm = 0.1 * mu
    rv = 0.9 + 0.1 * var * rows / (rows - 1)
    assert (bn.running_mean.double().cpu() - rm).abs().max() <= 1e-6 * max(1.0, rm.abs().max().item())
    assert ((bn.running_var.double().cpu() - rv).abs() / rv).max() <= 1e-6


@pytest.mark.parametrize("shape", [(2, 24, 18, 48), (3, 17, 13, 96), (2, 12, 9, 384), (8, 96, 72, 48)])
@pytest.mark.parametrize("relu,with_y", [(True, True), (True, False), (False, False)])
def test_backward_from_accumulator(dev, shape, relu, with_y):
    """bn_bwd on the accumulator path (streaming reduction -> accumulator -> apply, dgamma / dbeta by the apply kernel) vs
    autograd in fp64"""
    from buctd_amd import ops
    N, H, W, Cn = shape
    g = torch.Generator().manual_seed(N * H + Cn)
    z = torch.randn(N, H, W, Cn, generator=g) * 1.7 + 0.3
    dy = torch.randn(N, H, W, Cn, generator=g)
    res = torch.randn(N, H, W, Cn, generator=g) if with_y else None
    gamma = 1.0 + 0.3 * torch.randn(Cn, generator=g)
    beta = 0.2 * torch.randn(Cn, generator=g)
    zd = z.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rd = res.double().requires_grad_(True) if with_y else None
    zz = zd.reshape(-1, Cn)
    mu, var = zz.mean(0), zz.var(0, unbiased=False)
    out = ((zd - mu) / torch.sqrt(var + 1e-5)) * gd + bd
    if with_y:
        out = out + rd
    if relu:
        out = torch.relu(out)
    out.backward(dy.double())
    mean = mu.detach().float().to(dev)
    invstd = (1.0 / torch.sqrt(var + 1e-5)).detach().float().to(dev)
    dgamma = torch.zeros(Cn, device=dev)
    dbeta = torch.zeros(Cn, device=dev)
    dz, dres = ops.bn_bwd(dy.to(dev), out.detach().float().to(dev) if with_y else None, z.to(dev), mean, invstd, gamma.to(dev),
                          relu, with_y and relu, dgamma, dbeta, 0, beta=beta.to(dev))
    sc = max(1.0, zd.grad.abs().max().item())
    assert (dz.double().cpu() - zd.grad).abs().max() <= 1e-5 * sc
    assert (dgamma.double().cpu() - gd.grad).abs().max() <= 2e-6 * max(1.0, gd.grad.abs().max().item()) * 10
    assert (dbeta.double().cpu() - bd.grad).abs().max() <= 2e-6 * max(1.0, bd.grad.abs().max().item()) * 10
    if dres is not None:
        assert (dres.double().cpu() - rd.grad).abs().max() <= 1e-6 * sc


def test_accumulated_statistics_are_bit_deterministic(dev):
    """integer atomics: the same launch sequence gives the same bits every time, whatever the arrival order"""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(3)
    x = nhwc(torch.randn(16, 48, 96, 72, generator=g)).to(dev)
    w = wcl(torch.randn(48, 48, 3, 3, generator=g) * 0.05).to(dev)
    outs = []
    for _ in range(6):
        bn = _Bn(48, dev)
        z, acc, info = ops.conv_fwd(x, w, None, 1, 1, stats="acc")
        bnin = ops.BnAccInput(acc, z.numel() // 48, bn, True)
        y = ops.bn_apply_acc(z, bnin, None, True)
        outs.append((bnin.mean.clone(), bnin.invstd.clone(), y.clone(), bn.running_var.clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


@pytest.mark.parametrize("bad", [float("nan"), 3e22])
def test_broken_activations_poison_the_statistics(dev, bad):
    """a NaN (or a sum beyond the fixed-point range) must come out as NaN statistics, not as a plausible number"""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(5)
    x = nhwc(torch.randn(2, 48, 24, 18, generator=g))
    x[1, 7, 5, 11] = bad
    w = wcl(torch.randn(48, 48, 3, 3, generator=g) * 0.05).to(dev)
    bn = _Bn(48, dev)
    z, acc, info = ops.conv_fwd(x.to(dev), w, None, 1, 1, stats="acc")
    bnin = ops.BnAccInput(acc, z.numel() // 48, bn, True)
    y = ops.bn_apply_acc(z, bnin, None, False)
    torch.cuda.synchronize()
    assert torch.isnan(bnin.mean).any() and torch.isnan(y).any()


def test_pool_hands_out_zeroed_slices_and_rewinds(dev):
    from buctd_amd import ops
    pool = ops._AccPool()
    pool.CHUNK = 1 << 16
    p0 = pool.take(1000, dev)
    p1 = pool.take(1000, dev)
    assert p1 - p0 == 1024
    st = pool.state[dev.index]
    st["buf"][:2048].fill_(7)                       # a producer dirtied its slices
    pool.reset(dev)
    assert pool.take(1000, dev) == p0 and int(st["buf"][:2048].sum()) == 0
    big = pool.take(1 << 17, dev)                   # larger than a chunk: a fresh zeroed chunk of that size
    assert pool.state[dev.index]["buf"].numel() >= (1 << 17) and big == pool.state[dev.index]["buf"].data_ptr()
    other = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(other):
        pool.take(256, dev)                         # a second stream is ordered behind the fill once
    assert other.cuda_stream in pool.state[dev.index]["seen"]
    torch.cuda.synchronize()


@pytest.mark.parametrize("ratio", [1e2, 1e3])
@pytest.mark.parametrize("case", [(4, 24, 18, 48, 48, 3), (32, 96, 72, 48, 48, 3), (2, 20, 20, 64, 256, 1), (16, 96, 72, 64, 256, 1)])
def test_statistics_hold_at_large_mean_over_sigma(dev, ratio, case):
    """|mean| / sigma of the convolution OUTPUT = 1e2 and 1e3 (a bias-like offset on every channel): the sums travel around a
    per-wave pivot (3x3 kernels) / as fp64 terms and decode as s2 - s1 * mu in fp64, so the variance keeps its digits where a
    raw fp32 sum of squares would lose |mean|^2 / var of them.  Reference: fp64 statistics of the device's own output."""
    from buctd_amd import ops
    N, H, W, Ci, Co, k = case
    g = torch.Generator().manual_seed(Ci + Co + int(ratio))
    # an input whose channel 0 is constant makes every output channel carry the offset w[:, 0].sum() * const
    x = torch.randn(N, Ci, H, W, generator=g) * 0.05
    x[:, 0] = 1.0
    w = torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (Ci * k * k) ** 0.5)
    sig = float(torch.nn.functional.conv2d(x[:1], w, padding=k // 2).std())
    w[:, 0] = 0.0
    w[:, 0, k // 2, k // 2] = ratio * sig * (1.0 + 0.1 * torch.randn(Co, generator=g))     # offset = ratio * sigma on every channel
    z, acc, info = ops.conv_fwd(nhwc(x).to(dev), wcl(w).to(dev), None, 1, k // 2, stats="acc")
    assert info[0] == "acc"
    bn = _Bn(Co, dev)
    bnin = ops.BnAccInput(acc, z.numel() // Co, bn, True)
    ops.bn_apply_acc(z, bnin, None, False)
    zd = z.double().cpu().reshape(-1, Co)
    mu, var = zd.mean(0), zd.var(0, unbiased=False)
    got_ratio = float((mu.abs() / var.sqrt()).median())
    assert got_ratio > 0.5 * ratio, f"the case is meant to have |mean| / sigma ~ {ratio:g}, has {got_ratio:.1f}"
    invstd = 1.0 / torch.sqrt(var + bn.eps)
    assert ((bnin.mean.double().cpu() - mu).abs() / mu.abs()).max() <= 3e-7
    # (a raw fp32 sum of squares would be off by ~ratio^2 * 2^-24: 6e-4 at 1e2, 6e-2 at 1e3)
    assert ((bnin.invstd.double().cpu() - invstd).abs() / invstd).max() <= 2e-5 * max(1.0, ratio / 1e2)


def test_nan_in_every_shard_still_decodes_as_nan(dev):
    """a diverged activation poisons the accumulator copies of ALL XCDs: 4 x 2^61 wraps negative and 8 x 2^61 to zero, so the
    poison mark must be checked per copy before the copies are added (bn_acc.h: bnacc_gather_lds) - every consumer, not only
    the convolution prologue, must decode NaN"""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(11)
    x = nhwc(torch.randn(8, 48, 96, 72, generator=g))
    x[:, ::7, ::5, 3] = float("nan")                 # NaNs in every tile of every image: all eight shards see one
    w = wcl(torch.randn(48, 48, 3, 3, generator=g) * 0.05).to(dev)
    bn = _Bn(48, dev)
    z, acc, info = ops.conv_fwd(x.to(dev), w, None, 1, 1, stats="acc")
    bnin = ops.BnAccInput(acc, z.numel() // 48, bn, True)
    y = ops.bn_apply_acc(z, bnin, None, False)
    torch.cuda.synchronize()
    assert torch.isnan(bnin.mean).any() and torch.isnan(bnin.invstd).any() and torch.isnan(y).any()
    assert not torch.isfinite(bn.running_mean).all()


def test_wide_batchnorm_takes_the_accumulator_path(dev):
    """Co = 2048 (ResNet-50 layer4): 90 KB of dynamic LDS in bn_apply_acc - above the 64 KB default limit"""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(2048)
    Ci, Co = 64, 2048
    x = torch.randn(4, Ci, 8, 6, generator=g)
    w = torch.randn(Co, Ci, 1, 1, generator=g) * Ci ** -0.5
    z, acc, info = ops.conv_fwd(nhwc(x).to(dev), wcl(w).to(dev), None, 1, 0, stats="acc")
    if info[0] != "acc":
        pytest.skip("this shape does not take the accumulator form")
    bn = _Bn(Co, dev)
    bnin = ops.BnAccInput(acc, z.numel() // Co, bn, True)
    y = ops.bn_apply_acc(z, bnin, None, True)
    zd = z.double().cpu().reshape(-1, Co)
    yr = torch.relu((zd - zd.mean(0)) / torch.sqrt(zd.var(0, unbiased=False) + bn.eps) * bn.weight.double().cpu() + bn.bias.double().cpu())
    assert (y.double().cpu().reshape(-1, Co) - yr).abs().max() <= 2e-5 * max(1.0, yr.abs().max().item())


def test_stale_accumulator_reference_is_refused(dev):
    from buctd_amd import ops, _C
    a = ops.AccRef(48, dev)
    _ = a.ptr
    ops.step_boundary(dev)
    with pytest.raises(_C.BuctdHipError):
        _ = a.ptr
