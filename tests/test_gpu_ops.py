"""Per-kernel parity: every C-ABI entry point against a plain PyTorch fp32 CPU computation of the
same op on the same seeded inputs (tolerances stated per test).  Run with -m gpu on the MI355X."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def wcl(w):
    return w.contiguous(memory_format=torch.channels_last)


def close(a, b, tol, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.3e}) > tol {tol}"


CONV_CASES = [
    # N, H, W, Ci, Co, k, stride, pad, bias
    (2, 24, 18, 48, 48, 3, 1, 1, False),
    (2, 12, 9, 384, 384, 3, 1, 1, False),
    (3, 17, 13, 96, 192, 3, 2, 1, False),
    (2, 20, 20, 64, 256, 1, 1, 0, False),
    (2, 9, 7, 256, 64, 1, 1, 0, False),
    (1, 33, 29, 3, 64, 3, 2, 1, False),
    (2, 16, 12, 48, 14, 1, 1, 0, True),
    (2, 16, 12, 3, 3, 7, 1, 3, True),
    (2, 11, 10, 17, 48, 3, 1, 1, True),
    (2, 10, 8, 64, 3, 7, 1, 3, True),
    (2, 14, 10, 32, 32, 3, 1, 1, False),
    (2, 14, 10, 128, 128, 3, 1, 1, False),
    (8, 96, 72, 48, 48, 3, 1, 1, False),   # BM=128 tile path
    (4, 48, 36, 96, 96, 3, 1, 1, False),
    (2, 8, 6, 256, 256, 4, 2, 1, False),   # deconv geometry
    (2, 24, 18, 48, 96, 3, 2, 1, False),   # stride-2 data gradient: parity-class decomposition, even sizes
    (2, 16, 12, 48, 48, 3, 2, 1, True),
    (2, 13, 8, 192, 384, 3, 2, 1, False),  # odd height, even width
    (1, 6, 5, 64, 64, 3, 2, 1, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, case):
    from buctd_amd import ops
    N, H, W, Ci, Co, k, st, pad, has_b = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)
    b = torch.randn(Co, generator=g) if has_b else None
    x.requires_grad_(True)
    w.requires_grad_(True)
    y_ref = F.conv2d(x, w, b, stride=st, padding=pad)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)

    xd, wd = nhwc(x.detach()).to(dev), wcl(w.detach()).to(dev)
    bd = b.to(dev) if has_b else None
    y = ops.conv_fwd(xd, wd, bd, st, pad)
    close(nchw(y), y_ref, 2e-5, "conv fwd")
    dyd = nhwc(dy).to(dev)
    dx = ops.conv_dgrad(dyd, wd, tuple(xd.shape), st, pad)
    close(nchw(dx), x.grad, 2e-5, "conv dgrad")
    dw = ops.conv_wgrad(xd, dyd, wd, st, pad)
    close(dw, w.grad, 5e-5, "conv wgrad")
    # accumulate flag
    dw2 = ops.conv_wgrad(xd, dyd, wd, st, pad, out=dw.clone(), accumulate=1)
    close(dw2, 2 * w.grad, 5e-5, "conv wgrad accumulate")


def test_conv_eval_epilogue_and_stats(dev):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(5)
    N, H, W, Ci, Co = 3, 13, 11, 48, 96
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    res = torch.randn(N, Co, H, W, generator=g)
    scale, shift = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(x, w, None, 1, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    y = ops.conv_fwd(nhwc(x).to(dev), wcl(w).to(dev), None, 1, 1, scale=scale.to(dev), shift=shift.to(dev),
                     residual=nhwc(res).to(dev), relu=True)
    close(nchw(y), ref, 2e-5, "conv eval epilogue")
    # Welford partials -> mean / invstd / running stats
    z, part, info = ops.conv_fwd(nhwc(x).to(dev), wcl(w).to(dev), None, 1, 1, stats=True)
    zr = F.conv2d(x, w, None, 1, 1)
    rm, rv = torch.zeros(Co), torch.ones(Co)
    bn_ref = F.batch_norm(zr, rm, rv, None, None, True, 0.1, 1e-5)
    rmd, rvd = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    mean, invstd = ops.bn_finalize(part, info, z.numel() // Co, Co, 1e-5, 0.1, rmd, rvd)
    close(mean, zr.mean((0, 2, 3)), 1e-5, "bn mean")
    close(invstd, 1.0 / torch.sqrt(zr.var((0, 2, 3), unbiased=False) + 1e-5), 1e-5, "bn invstd")
    close(rmd, rm, 1e-5, "running mean")
    close(rvd, rv, 1e-5, "running var")
    one, zero = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    close(nchw(ops.bn_apply(z, mean, invstd, one, zero)), bn_ref, 2e-5, "bn apply")


@pytest.mark.parametrize("shape", [(4, 24, 18, 48), (2, 12, 9, 384), (3, 7, 5, 3), (2, 48, 36, 96)])
@pytest.mark.parametrize("relu,res", [(True, True), (True, False), (False, False), (False, True)])
def test_bn_train_fwd_bwd(dev, shape, relu, res):
    from buctd_amd import ops
    N, H, W, Cn = shape
    g = torch.Generator().manual_seed(11)
    z = (torch.randn(N, Cn, H, W, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(Cn, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(Cn, generator=g).requires_grad_(True)
    r = torch.randn(N, Cn, H, W, generator=g).requires_grad_(True) if res else None
    out = F.batch_norm(z, None, None, gamma, beta, True, 0.1, 1e-5)
    if res:
        out = out + r
    if relu:
        out = F.relu(out)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)

    zd = nhwc(z.detach()).to(dev)
    part, info = ops.bn_stats(zd)
    mean, invstd = ops.bn_finalize(part, info, zd.numel() // Cn, Cn, 1e-5, 0.1, None, None)
    rd = nhwc(r.detach()).to(dev) if res else None
    y = ops.bn_apply(zd, mean, invstd, gamma.detach().to(dev), beta.detach().to(dev), rd, relu)
    close(nchw(y), out, 2e-5, "bn fwd")
    dgamma = torch.empty(Cn, device=dev)
    dbeta = torch.empty(Cn, device=dev)
    dz, dres = ops.bn_bwd(nhwc(dy).to(dev), y, zd, mean, invstd, gamma.detach().to(dev), relu, res and relu, dgamma,
                          dbeta, 0)
    close(nchw(dz), z.grad, 5e-5, "bn dz")
    close(dgamma, gamma.grad, 5e-5, "bn dgamma")
    close(dbeta, beta.grad, 5e-5, "bn dbeta")
    if res and relu:
        close(nchw(dres), r.grad, 1e-6, "bn dres")


def _mm(dev, A, B, Cshape, **kw):
    from buctd_amd import ops
    out = torch.full(Cshape, float("nan"), device=dev)
    ops.matmul(A.to(dev), B.to(dev), out, **kw)
    return out


def test_matmul_layouts(dev):
    g = torch.Generator().manual_seed(3)
    Bn, M, N, K = 3, 150, 100, 72
    A = torch.randn(Bn, M, K, generator=g)
    Bm = torch.randn(Bn, N, K, generator=g)
    ref = torch.einsum("bmk,bnk->bmn", A, Bm)
    # rows x rows
    out = _mm(dev, A, Bm, (Bn, M, N), batch=Bn, M=M, N=N, K=K, a_layout=0, b_layout=0, lda=K, ldb=K, ldc=N,
              stride_a=M * K, stride_b=N * K, stride_c=M * N, alpha=0.5)
    close(out, 0.5 * ref, 2e-5, "mm rows/rows")
    # rows x cols
    Bt = Bm.transpose(1, 2).contiguous()  # [B,K,N]
    out = _mm(dev, A, Bt, (Bn, M, N), batch=Bn, M=M, N=N, K=K, a_layout=0, b_layout=1, lda=K, ldb=N, ldc=N,
              stride_a=M * K, stride_b=N * K, stride_c=M * N)
    close(out, ref, 2e-5, "mm rows/cols")
    # cols x cols
    At = A.transpose(1, 2).contiguous()  # [B,K,M]
    out = _mm(dev, At, Bt, (Bn, M, N), batch=Bn, M=M, N=N, K=K, a_layout=1, b_layout=1, lda=M, ldb=N, ldc=N,
              stride_a=M * K, stride_b=N * K, stride_c=M * N)
    close(out, ref, 2e-5, "mm cols/cols")
    # cols x rows
    out = _mm(dev, At, Bm, (Bn, M, N), batch=Bn, M=M, N=N, K=K, a_layout=1, b_layout=0, lda=M, ldb=K, ldc=N,
              stride_a=M * K, stride_b=N * K, stride_c=M * N)
    close(out, ref, 2e-5, "mm cols/rows")
    # odd sizes -> scalar path, bias per column
    M2, N2, K2 = 37, 19, 13
    A2, B2, bias = torch.randn(M2, K2, generator=g), torch.randn(N2, K2, generator=g), torch.randn(N2, generator=g)
    out = _mm(dev, A2, B2, (M2, N2), batch=1, M=M2, N=N2, K=K2, a_layout=0, b_layout=0, lda=K2, ldb=K2, ldc=N2,
              bias=bias.to(dev))
    close(out, A2 @ B2.t() + bias, 2e-5, "mm scalar path")


def test_matmul_splitk_and_groups(dev):
    g = torch.Generator().manual_seed(4)
    # channel-attention logits: tiny output, long reduction -> split-K
    Bn, T, Cn = 4, 1728, 48
    q, y = torch.randn(Bn, T, Cn, generator=g), torch.randn(Bn, T, Cn, generator=g)
    ref = torch.einsum("btc,btd->bcd", q, y)
    out = _mm(dev, q, y, (Bn, Cn, Cn), batch=Bn, M=Cn, N=Cn, K=T, a_layout=1, b_layout=1, lda=Cn, ldb=Cn, ldc=Cn,
              stride_a=T * Cn, stride_b=T * Cn, stride_c=Cn * Cn)
    close(out, ref, 5e-5, "mm split-K")
    # fc_o forward: out[b][t'][c] = sum_t W[t'][t] on[b][t][c] + bias[t']
    T2 = 432
    Wt, on, bias = torch.randn(T2, T2, generator=g) * 0.05, torch.randn(Bn, T2, Cn, generator=g), torch.randn(T2, generator=g)
    ref = torch.einsum("st,btc->bsc", Wt, on) + bias.view(1, -1, 1)
    out = _mm(dev, Wt, on, (Bn, T2, Cn), batch=1, M=T2, N=Bn * Cn, K=T2, a_layout=0, b_layout=1, lda=T2, ldb=Cn,
              ldc=Cn, Nc=Cn, gsbn=T2 * Cn, gsc=T2 * Cn, bias=bias.to(dev), bias_axis=1)
    close(out, ref, 2e-5, "mm n-groups (fc_o fwd)")
    # fc_o weight gradient: dW[t'][t] = sum_{b,c} d[b][t'][c] on[b][t][c]
    d = torch.randn(Bn, T2, Cn, generator=g)
    ref = torch.einsum("bsc,btc->st", d, on)
    out = _mm(dev, d, on, (T2, T2), batch=1, M=T2, N=T2, K=Bn * Cn, a_layout=0, b_layout=0, lda=Cn, ldb=Cn, ldc=T2,
              Kc=Cn, gsa=T2 * Cn, gsbk=T2 * Cn)
    close(out, ref, 5e-5, "mm k-groups (fc_o wgrad)")
    # fc_o data gradient: don[b][t][c] = sum_t' W[t'][t] d[b][t'][c]
    ref = torch.einsum("st,bsc->btc", Wt, d)
    out = _mm(dev, Wt, d, (Bn, T2, Cn), batch=1, M=T2, N=Bn * Cn, K=T2, a_layout=1, b_layout=1, lda=T2, ldb=Cn,
              ldc=Cn, Nc=Cn, gsbn=T2 * Cn, gsc=T2 * Cn)
    close(out, ref, 2e-5, "mm cols + n-groups (fc_o dgrad)")
    # bias gradient through a ones vector (N = 1)
    ones = torch.ones(Bn * Cn)
    out = _mm(dev, d, ones, (T2,), batch=1, M=T2, N=1, K=Bn * Cn, a_layout=0, b_layout=0, lda=Cn, ldb=Bn * Cn, ldc=1,
              Kc=Cn, gsa=T2 * Cn, gsbk=Cn)
    close(out, d.sum((0, 2)), 5e-5, "mm bias grad")


@pytest.mark.parametrize("rows,L", [(64, 48), (10, 432), (6, 1728), (3, 6912), (5, 100)])
def test_softmax_fwd_bwd(dev, rows, L):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(7)
    s = (torch.randn(rows, L, generator=g) * 3).requires_grad_(True)
    scale = 0.37
    p_ref = torch.softmax(s * scale, -1)
    dp = torch.randn(rows, L, generator=g)
    p_ref.backward(dp)
    p, pd = ops.softmax_dropout_fwd(s.detach().to(dev), L, scale, 0.0, 123, inplace=False)
    assert pd is p
    close(p, p_ref, 2e-6, "softmax fwd")
    ds = ops.softmax_dropout_bwd(dp.to(dev), p, L, scale, 0.0, 123, inplace=False)
    close(ds, s.grad, 2e-6, "softmax bwd")


def test_softmax_dropout_mask_consistency(dev):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(8)
    rows, L, pdrop = 512, 432, 0.1
    s = torch.randn(rows, L, generator=g).to(dev)
    p, pd = ops.softmax_dropout_fwd(s, L, 1.0, pdrop, 999, inplace=False)
    keep = (pd != 0)
    frac = keep.float().mean().item()
    assert abs(frac - 0.9) < 0.01, f"keep fraction {frac}"
    close(pd[keep], (p / 0.9)[keep], 1e-6, "inverted dropout scale")
    # backward regenerates the same mask: ds = p * (g - sum g p) with g = dpd * keep / 0.9
    dpd = torch.randn(rows, L, generator=g).to(dev)
    ds = ops.softmax_dropout_bwd(dpd, p, L, 1.0, pdrop, 999, inplace=False)
    gk = (dpd * keep / 0.9).cpu()
    pc = p.cpu()
    ref = pc * (gk - (gk * pc).sum(-1, keepdim=True))
    close(ds, ref, 2e-6, "dropout softmax bwd")
    # different seed -> different mask
    _, pd2 = ops.softmax_dropout_fwd(s, L, 1.0, pdrop, 1000, inplace=False)
    assert ((pd2 != 0) != keep).any()


def test_layout_fuse_resize_misc(dev):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 6, 20, 14, generator=g)
    close(ops.nchw_to_nhwc(x.to(dev), 0, 3), nhwc(x[:, :3]), 0, "nchw->nhwc slice")
    close(ops.nchw_to_nhwc(x.to(dev), 3, 3), nhwc(x[:, 3:]), 0, "nchw->nhwc slice2")
    xh = torch.randn(2, 37, 5, 50, generator=g)
    close(ops.nhwc_to_nchw(xh.to(dev)), nchw(xh), 0, "nhwc->nchw")
    # fuse: relu(a + up2(b) + up4(c))
    a = torch.randn(2, 48, 16, 12, generator=g, requires_grad=True)
    b = torch.randn(2, 48, 8, 6, generator=g, requires_grad=True)
    c = torch.randn(2, 48, 4, 3, generator=g, requires_grad=True)
    ref = F.relu(a + F.interpolate(b, scale_factor=2, mode="nearest") + F.interpolate(c, scale_factor=4, mode="nearest"))
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    out = ops.fuse_sum([nhwc(a.detach()).to(dev), nhwc(b.detach()).to(dev), nhwc(c.detach()).to(dev)], [0, 1, 2], True)
    close(nchw(out), ref, 1e-6, "fuse fwd")
    dyd = nhwc(dy).to(dev)
    close(nchw(ops.fuse_sum_bwd(dyd, out, 0)), a.grad, 1e-6, "fuse bwd s0")
    close(nchw(ops.fuse_sum_bwd(dyd, out, 1)), b.grad, 1e-5, "fuse bwd s1")
    close(nchw(ops.fuse_sum_bwd(dyd, out, 2)), c.grad, 1e-5, "fuse bwd s2")
    # bilinear (no antialias, align_corners False) from an NCHW slice
    cond = torch.rand(2, 6, 96, 72, generator=g) * 255
    for (ho, wo) in [(24, 18), (12, 9), (6, 5)]:
        ref = F.interpolate(cond[:, 3:], size=(ho, wo), mode="bilinear", align_corners=False)
        close(nchw(ops.resize_bilinear_from_nchw(cond.to(dev), 3, 3, ho, wo)), ref, 1e-6, "resize")
    # colsum / add / relu_bwd / scale
    t = torch.randn(1000, 48, generator=g)
    out = torch.empty(48, device=dev)
    close(ops.colsum(t.to(dev), 48, out, 0), t.sum(0), 2e-6, "colsum")
    close(ops.colsum(t.to(dev), 48, out, 1), 2 * t.sum(0), 2e-6, "colsum accumulate")
    u = torch.randn(1000, 48, generator=g)
    close(ops.add(t.to(dev), u.to(dev), relu=True), F.relu(t + u), 0, "add relu")
    close(ops.relu_bwd(t.to(dev), u.to(dev)), t * (u > 0), 0, "relu bwd")
    s = torch.tensor([0.25], device=dev)
    close(ops.scale(t.to(dev), s, 2.0), t * 0.5, 0, "scale")
    # maxpool
    xm = torch.randn(2, 16, 15, 13, generator=g, requires_grad=True)
    ref = F.max_pool2d(xm, 3, 2, 1)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    ym, idx = ops.maxpool3x3s2_fwd(nhwc(xm.detach()).to(dev))
    close(nchw(ym), ref, 0, "maxpool fwd")
    close(nchw(ops.maxpool3x3s2_bwd(nhwc(dy).to(dev), idx, tuple(nhwc(xm.detach()).shape))), xm.grad, 1e-6, "maxpool bwd")


def test_layernorm_dropout(dev):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(10)
    x = torch.randn(300, 112, generator=g, requires_grad=True)
    gamma = (torch.rand(112, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(112, generator=g).requires_grad_(True)
    ref = F.layer_norm(x, (112,), gamma, beta, 1e-5)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    y, mean, invstd = ops.layernorm_fwd(x.detach().to(dev), gamma.detach().to(dev), beta.detach().to(dev), 1e-5)
    close(y, ref, 5e-6, "ln fwd")
    dg, db = torch.empty(112, device=dev), torch.empty(112, device=dev)
    dx = ops.layernorm_bwd(dy.to(dev), x.detach().to(dev), mean, invstd, gamma.detach().to(dev), dg, db, 0)
    close(dx, x.grad, 2e-5, "ln dx")
    close(dg, gamma.grad, 2e-5, "ln dgamma")
    close(db, beta.grad, 2e-5, "ln dbeta")
    # LayerNorm(a + b) in one pass == add, then LayerNorm: same bits; the sum is returned only when the backward needs it
    a, b = x.detach().to(dev), torch.randn(300, 112, generator=g).to(dev)
    y2, m2, i2, s2 = ops.add_layernorm_fwd(a, b, gamma.detach().to(dev), beta.detach().to(dev), 1e-5, keep_sum=True)
    y3, m3, i3 = ops.layernorm_fwd(a + b, gamma.detach().to(dev), beta.detach().to(dev), 1e-5)
    assert torch.equal(s2, a + b) and torch.equal(y2, y3) and torch.equal(m2, m3) and torch.equal(i2, i3)
    y4, _, _, s4 = ops.add_layernorm_fwd(a, b, gamma.detach().to(dev), beta.detach().to(dev), 1e-5, keep_sum=False)
    assert s4 is None and torch.equal(y4, y3)
    from buctd_amd import ops_seq
    ln = torch.nn.LayerNorm(112).to(dev)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out = ops_seq.AddLayerNorm.apply(ar, br, ln)
    out.backward(dy.to(dev))
    a64, b64 = a.cpu().double().requires_grad_(True), b.cpu().double().requires_grad_(True)
    F.layer_norm(a64 + b64, (112,), ln.weight.detach().cpu().double(), ln.bias.detach().cpu().double(), 1e-5).backward(dy.double())
    close(ar.grad, a64.grad.float(), 2e-5, "add+ln da")
    close(br.grad, b64.grad.float(), 2e-5, "add+ln db")
    with torch.no_grad():
        assert torch.equal(ops_seq.AddLayerNorm.apply(a, b, ln), out.detach())
    z = torch.ones(100000, device=dev)
    d1 = ops.dropout(z, 0.1, 42)
    assert abs((d1 != 0).float().mean().item() - 0.9) < 0.01
    assert torch.equal(d1, ops.dropout(z, 0.1, 42))


def test_loss_decode_target_adam(dev):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(12)
    N, K, H, W = 5, 14, 24, 18
    pred = torch.randn(N, K, H, W, generator=g, requires_grad=True)
    gt = torch.rand(N, K, H, W, generator=g)
    w = (torch.rand(N, K, 1, generator=g) > 0.3).float() * torch.rand(N, K, 1, generator=g)
    # reference JointsMSELoss (core/loss.py:23-41)
    loss_ref = 0
    for k in range(K):
        hp = pred.reshape(N, K, -1)[:, k] * w[:, k]
        hg = gt.reshape(N, K, -1)[:, k] * w[:, k]
        loss_ref = loss_ref + 0.5 * F.mse_loss(hp, hg)
    loss_ref = loss_ref / K
    loss_ref.backward()
    loss, grad = ops.joints_mse(pred.detach().to(dev), gt.to(dev), w.to(dev), True)
    close(loss, loss_ref, 1e-5, "mse loss")
    close(grad, pred.grad, 1e-5, "mse grad")
    # argmax with exact ties and non-positive rows
    hm = torch.randn(N, K, H, W, generator=g)
    hm[0, 0] = 0.5
    hm[0, 0, 3, 4] = 2.0
    hm[0, 0, 10, 2] = 2.0  # later tie must lose
    hm[1, 1] = -1.0        # max <= 0 -> preds zeroed
    flat = hm.reshape(N, K, -1).numpy()
    idx_ref = flat.argmax(2)
    mv_ref = flat.max(2)
    preds, maxvals, idx = ops.argmax_decode(hm.to(dev))
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    close(maxvals.reshape(N, K), torch.from_numpy(mv_ref), 0, "maxvals")
    pr = np.stack([idx_ref % W, idx_ref // W], -1).astype(np.float32) * (mv_ref > 0)[..., None]
    close(preds, torch.from_numpy(pr), 0, "preds")
    # flip-back + shift + average (core/function.py:226-236)
    a, bfl = torch.randn(N, K, H, W, generator=g), torch.randn(N, K, H, W, generator=g)
    pairs = [[0, 1], [2, 3], [4, 5]]
    fb = bfl.numpy()[:, :, :, ::-1].copy()
    for p0, p1 in pairs:
        tmp = fb[:, p0].copy()
        fb[:, p0] = fb[:, p1]
        fb[:, p1] = tmp
    fbt = torch.from_numpy(fb.copy())
    fbs = fbt.clone()
    fbs[:, :, :, 1:] = fbt[:, :, :, 0:-1]
    perm = list(range(K))
    for p0, p1 in pairs:
        perm[p0], perm[p1] = p1, p0
    permd = torch.tensor(perm, dtype=torch.int32, device=dev)
    close(ops.flipback_avg(a.to(dev), bfl.to(dev), permd, True), (a + fbs) * 0.5, 1e-7, "flipback shift")
    close(ops.flipback_avg(a.to(dev), bfl.to(dev), permd, False), (a + fbt) * 0.5, 1e-7, "flipback")
    # Adam against torch.optim.Adam over 3 steps
    p = torch.randn(1003, generator=g)
    pr_ = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr_], lr=1e-3)
    pd_, m, v = p.to(dev), torch.zeros(1003, device=dev), torch.zeros(1003, device=dev)
    for step in range(1, 4):
        gr = torch.randn(1003, generator=g)
        pr_.grad = gr.clone()
        opt.step()
        ops.adam_step(pd_, gr.to(dev), m, v, 1e-3, 0.9, 0.999, 1e-8, step)
    close(pd_, pr_, 1e-6, "adam")


def test_target_and_condition_render_vs_oracle(dev):
    """Device versions of generate_target (JointsDataset.py:397-453) and the condition renderers (500-543)
    against the oracle restatements (which are pinned to the reference / hand-derived vectors on the CPU side)."""
    from buctd_amd import ops
    from oracle import core as oc
    g = torch.Generator().manual_seed(21)
    B, K, W, H = 4, 14, 288, 384
    joints = torch.rand(B, K, 3, generator=g) * torch.tensor([W * 1.3, H * 1.3, 0.0]) - torch.tensor([W * 0.15, H * 0.15, 0.0])
    joints[0, 0, :2] = torch.tensor([-40.0, 10.0])
    joints[0, 1, :2] = torch.tensor([W + 60.0, 5.0])
    joints[0, 2, :2] = torch.tensor([-3.0, -3.0])
    vis = (torch.rand(B, K, generator=g) > 0.2).float()
    target, weight = ops.gaussian_target(joints.to(dev), vis.to(dev), (72, 96), (288, 384), 3)
    for b in range(B):
        v3 = vis[b].view(K, 1).repeat(1, 3).numpy()
        t, w = oc.generate_target(joints[b].numpy(), v3, K, (72, 96), (288, 384), 3)
        assert np.abs(target[b].cpu().numpy() - t).max() <= 2e-7, "gaussian target"
        assert np.array_equal(weight[b].cpu().numpy(), w)
    # colored condition (3 channels, CrowdPose palette) incl. key points on the border and duplicates
    cj = torch.rand(B, K, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    cj[1, 0] = torch.tensor([3.0, 4.0])        # reflect-101 corner
    cj[1, 1] = torch.tensor([W - 2.0, H - 2.0])
    cj[1, 2] = cj[1, 3]                        # same pixel: later key point overwrites the earlier colour
    cj[2, 0] = torch.tensor([0.4, 10.0])       # x truncates to 0 -> rejected
    colors = torch.tensor(oc.CROWDPOSE_KPT_COLORS, dtype=torch.float32)
    cond = ops.cond_render(cj.to(dev), colors.to(dev), H, W).cpu().numpy()
    for b in range(B):
        ref = oc.get_condition_image_colored(cj[b].numpy(), (H, W, 3), oc.CROWDPOSE_KPT_COLORS).transpose(2, 0, 1)
        assert np.abs(cond[b] - ref).max() <= 2e-3, f"colored condition image {b}: {np.abs(cond[b] - ref).max()}"
        assert abs(cond[b].max() - 255.0) < 1e-3
    mono = ops.cond_render(cj.to(dev), None, H, W, truncate=True).cpu().numpy()
    for b in range(B):
        ref = oc.get_condition_image(cj[b].numpy(), (H, W))[0:1].astype(np.float64)
        d = np.abs(mono[b] - ref)
        assert d.max() <= 1.0 and (d > 0).mean() < 1e-3, "mono condition: only integer-boundary flips allowed"


@pytest.mark.parametrize("momentum,nesterov,wd", [(0.9, False, 1e-4), (0.9, True, 0.0), (0.0, False, 1e-2)])
def test_fused_sgd_matches_torch_sgd(dev, momentum, nesterov, wd):
    """engine.FusedSGD (the 'sgd' branch of the reference's get_optimizer, lib/utils/utils.py:260-267) against
    torch.optim.SGD on the CPU: three steps, parameters and momentum buffers."""
    import copy
    from buctd_amd import engine
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
    twin = copy.deepcopy(net).to(dev)
    ref = torch.optim.SGD(net.parameters(), lr=0.05, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    fused = engine.FusedSGD(engine.FlatParams(twin), lr=0.05, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    for it in range(3):
        grads = [torch.randn_like(p) for p in net.parameters()]
        for p, q, g in zip(net.parameters(), twin.parameters(), grads):
            p.grad = g.clone()
            q.grad = g.to(dev)
        ref.step()
        fused.step()
    for p, q in zip(net.parameters(), twin.parameters()):
        assert (p.detach() - q.detach().cpu()).abs().max().item() <= 1e-6 * max(1.0, p.abs().max().item())
    sd = fused.state_dict()
    if momentum:
        back = torch.optim.SGD(copy.deepcopy(net).parameters(), lr=1.0, momentum=momentum)
        back.load_state_dict(sd)        # torch accepts what FusedSGD writes
        rs = ref.state_dict()["state"]
        for i in rs:
            assert (rs[i]["momentum_buffer"] - sd["state"][i]["momentum_buffer"].cpu()).abs().max().item() <= 1e-6
        again = engine.FusedSGD(engine.FlatParams(copy.deepcopy(twin)), lr=0.01, momentum=momentum)
        again.load_state_dict(ref.state_dict())
        assert again.param_groups[0]["lr"] == 0.05 and again.step_count == 1


def test_fused_sgd_skips_parameters_without_gradient(dev):
    """torch.optim.SGD leaves a parameter whose .grad is None alone - no weight decay, no momentum: TransPose's frozen
    pos_embedding (transpose_h.py:129, requires_grad=False) and modules that are constructed but never called.  The
    flat-arena kernel must not decay them either (wd > 0, momentum > 0, three steps; one parameter joins at step 2)."""
    import copy
    from buctd_amd import engine
    torch.manual_seed(5)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Conv2d(3, 8, 3)
            self.frozen = torch.nn.Parameter(torch.randn(7, 5), requires_grad=False)
            self.unused = torch.nn.Linear(6, 3)
            self.late = torch.nn.Parameter(torch.randn(9))
            self.b = torch.nn.Conv2d(8, 4, 1)

    net = Net()
    twin = copy.deepcopy(net).to(dev)
    ref = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-2)
    fused = engine.FusedSGD(engine.FlatParams(twin), lr=0.05, momentum=0.9, weight_decay=1e-2)
    frozen0, unused0 = net.frozen.detach().clone(), net.unused.weight.detach().clone()
    for it in range(3):
        ref.zero_grad()
        fused.zero_grad()
        for (n, p), q in zip(net.named_parameters(), twin.parameters()):
            if n.startswith(("frozen", "unused")) or (n == "late" and it < 1):
                continue
            g = torch.randn_like(p)
            p.grad = g.clone()
            q.grad = g.to(dev)
        ref.step()
        fused.step()
    for (n, p), q in zip(net.named_parameters(), twin.parameters()):
        assert (p.detach() - q.detach().cpu()).abs().max().item() <= 1e-6 * max(1.0, p.abs().max().item()), n
    assert torch.equal(twin.frozen.detach().cpu(), frozen0) and torch.equal(twin.unused.weight.detach().cpu(), unused0)
