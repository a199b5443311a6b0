"""CoAM attention cores: ops.PositionAttention / ops.ChannelAttention (batched MFMA matmuls + fused softmax) against
the formulas of reference lib/models/self_attention.py:74-87 and 146-159 evaluated in fp64 on the CPU.
Bar: forward and all gradients within 2e-5 relative (fp32 round-off through a softmax)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _e(a, b):
    return ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B,T,C,h", [(2, 96, 16, 1), (3, 6, 128, 1), (2, 24, 64, 2), (2, 384, 16, 2), (1, 1728, 96, 1)])
def test_position_attention(dev, B, T, C, h):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    q = torch.randn(B, T, h * C, generator=g, dtype=torch.float64).requires_grad_(True)
    k = torch.randn(B, T, h * C, generator=g, dtype=torch.float64).requires_grad_(True)
    v = torch.randn(B, T, h * C, generator=g, dtype=torch.float64).requires_grad_(True)
    qh = q.view(B, T, h, C).permute(0, 2, 1, 3)
    kh = k.view(B, T, h, C).permute(0, 2, 3, 1)
    vh = v.view(B, T, h, C).permute(0, 2, 1, 3)
    att = torch.softmax(torch.matmul(qh, kh) / math.sqrt(C), -1)
    out = torch.matmul(att, vh).permute(0, 2, 1, 3).contiguous().view(B, T, h * C)
    dout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dout)
    qd, kd, vd = (t.detach().float().to(dev).requires_grad_(True) for t in (q, k, v))
    o = ops.PositionAttention.apply(qd, kd, vd, h, 0.1, False)
    o.backward(dout.float().to(dev))
    for name, a, b in (("out", o.detach(), out.detach()), ("dq", qd.grad, q.grad), ("dk", kd.grad, k.grad),
                       ("dv", vd.grad, v.grad)):
        err = _e(a, b)
        assert err <= 2e-5, f"position attention {name} (B{B} T{T} C{C} h{h}): rel err {err:.2e}"


@pytest.mark.parametrize("B,T,C,h", [(2, 96, 16, 1), (3, 6, 128, 1), (2, 24, 64, 2), (2, 384, 32, 2), (2, 1728, 48, 1),
                                     (4, 1728, 48, 1), (2, 576, 96, 1)])   # the last two: fc_o on the bf16x6 GEMM
def test_channel_attention(dev, B, T, C, h):
    from buctd_amd import ops
    g = torch.Generator().manual_seed(B * 77 + T)
    # token-major inputs [B,T,C]; the reference sees them channel-major [B,C,T]
    qn = torch.randn(B, T, C, generator=g, dtype=torch.float64).requires_grad_(True)
    yn = torch.randn(B, T, C, generator=g, dtype=torch.float64).requires_grad_(True)
    W = (torch.randn(T, T, generator=g, dtype=torch.float64) / math.sqrt(T)).requires_grad_(True)
    bias = torch.randn(T, generator=g, dtype=torch.float64).requires_grad_(True)
    dk = T // h
    qc, yc = qn.permute(0, 2, 1), yn.permute(0, 2, 1)                      # [B,C,T]
    qh = qc.reshape(B, C, h, dk).permute(0, 2, 1, 3)                        # [B,h,C,dk]
    kh = yc.reshape(B, C, h, dk).permute(0, 2, 3, 1)                        # [B,h,dk,C]
    vh = yc.reshape(B, C, h, dk).permute(0, 2, 1, 3)
    att = torch.softmax(torch.matmul(qh, kh) / math.sqrt(dk), -1)           # [B,h,C,C]
    out = torch.matmul(att, vh).permute(0, 2, 1, 3).contiguous().view(B, C, h * dk)
    out = torch.nn.functional.linear(out, W, bias)                           # [B,C,T]
    out_tok = out.permute(0, 2, 1)                                           # [B,T,C]
    dout = torch.randn(out_tok.shape, generator=g, dtype=torch.float64)
    out_tok.backward(dout)
    qd, yd = (t.detach().float().to(dev).requires_grad_(True) for t in (qn, yn))
    Wd = torch.nn.Parameter(W.detach().float().to(dev))
    bd = torch.nn.Parameter(bias.detach().float().to(dev))
    o = ops.ChannelAttention.apply(qd, yd, Wd, bd, h, 0.1, False)
    o.backward(dout.float().to(dev))
    for name, a, b in (("out", o.detach(), out_tok.detach()), ("dq", qd.grad, qn.grad), ("dy", yd.grad, yn.grad),
                       ("dW", Wd.grad, W.grad), ("db", bd.grad, bias.grad)):
        err = _e(a, b)
        assert err <= 2e-5, f"channel attention {name} (B{B} T{T} C{C} h{h}): rel err {err:.2e}"
    # second backward accumulates into the existing parameter gradients
    o2 = ops.ChannelAttention.apply(qd, yd, Wd, bd, h, 0.1, False)
    o2.backward(dout.float().to(dev))
    assert _e(Wd.grad, 2 * W.grad) <= 2e-5 and _e(bd.grad, 2 * bias.grad) <= 2e-5


@pytest.mark.parametrize("mode", ["bf16x6", "fp32"])
@pytest.mark.parametrize("B,T,d", [(2, 384, 48), (1, 3072, 112), (3, 128, 16), (2, 256, 128), (2, 256, 96)])
def test_fused_mha_forward_vs_fp64(dev, B, T, d, mode):
    """attn_mha.hip (TransPose encoder self-attention, eval): softmax(q k^T / sqrt(d)) v against an fp64 evaluation and
    against the materialised HIP path; logits spread wide enough to exercise the online rescaling."""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(B * T + d)
    qk = torch.randn(B, T, 2 * d, generator=g) * 1.5
    qk[:, ::7, :d] *= 4.0                       # a few peaked rows
    v = torch.randn(B, T, d, generator=g)
    q64, k64, v64 = qk[..., :d].double(), qk[..., d:].double(), v.double()
    ref = torch.softmax(q64 @ k64.transpose(1, 2) / math.sqrt(d), dim=-1) @ v64
    assert ops.mha_fused_ok(T, d)
    old = ops.get_conv_math()
    ops.set_conv_math(mode)        # bf16x6 (default): split-operand bf16 MFMA kernel; fp32: exact fp32 MFMA kernel
    try:
        out = ops.mha_fwd(qk.to(dev), v.to(dev))
    finally:
        ops.set_conv_math(old)
    err = (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-5, f"fused MHA[{mode}] (B{B} T{T} d{d}): rel err {err:.2e}"
    mat = ops.PositionAttention.apply(qk.to(dev), None, v.to(dev), 1, 0.0, False)
    assert (mat - out).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert not ops.mha_fused_ok(100, d) and not ops.mha_fused_ok(T, 20)


@pytest.mark.parametrize("B,T,d", [(8, 384, 112), (16, 128, 48), (3, 256, 16), (1, 3072, 112), (2, 256, 128)])
def test_fused_mha_presplit_matches_the_in_kernel_split(dev, B, T, d, monkeypatch):
    """buctd_mha_fwd_bf16x6_ws (keys / values split once into the workspace, DMA-staged tiles, 32 queries per wavefront,
    the key range split over the two wave quartets and merged, one image per XCD - B a multiple of 8 takes the XCD-aware
    workgroup map, other B the plain one) against buctd_mha_fwd_bf16x6: the same exact pieces and products, a different
    association of the online soft-max (32-key tiles, two halves merged) -> equal to fp32 round-off, and as close to fp64.
    The workspace is dirtied first: pad bytes of the image rows are never read."""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(B * T + d + 5)
    qk = (torch.randn(B, T, 2 * d, generator=g) * 1.5)
    qk[:, ::5, :d] *= 4.0
    v = torch.randn(B, T, d, generator=g)
    ref64 = torch.softmax(qk[..., :d].double() @ qk[..., d:].double().transpose(1, 2) / math.sqrt(d), dim=-1) @ v.double()
    qk, v = qk.to(dev), v.to(dev)
    monkeypatch.setattr(ops, "_MHA_PRESPLIT", False)
    ref = ops.mha_fwd(qk, v)
    monkeypatch.setattr(ops, "_MHA_PRESPLIT", True)
    ops.workspace(ops.lib().buctd_mha_fwd_bf16x6_workspace(B, T, d), dev).fill_(0xFF)
    out = ops.mha_fwd(qk, v)
    scale = ref64.abs().max().item()
    assert (out - ref).abs().max().item() <= 4e-6 * scale
    e_new = (out.cpu().double() - ref64).abs().max().item() / scale
    e_old = (ref.cpu().double() - ref64).abs().max().item() / scale
    assert e_new <= max(2e-6, 2 * e_old), f"rel err {e_new:.2e} (in-kernel split: {e_old:.2e})"
    # the optional log-sum-exp output (merged over the two key halves)
    lse = torch.empty(B, T, device=dev)
    ws = ops.workspace(ops.lib().buctd_mha_fwd_bf16x6_workspace(B, T, d), dev)
    out2 = torch.empty_like(out)
    ops.check(ops.lib().buctd_mha_fwd_bf16x6_ws(B, T, d, ops.ptr(qk), qk.data_ptr() + 4 * d, ops.ptr(v), 2 * d, d,
                                                1.0 / math.sqrt(d), ops.ptr(out2), ops.ptr(lse), ops.ptr(ws), ws.numel(),
                                                ops.stream_ptr()), "mha_fwd_ws")
    l64 = torch.logsumexp(qk[..., :d].double() @ qk[..., d:].double().transpose(1, 2) / math.sqrt(d), dim=-1)
    assert torch.equal(out2, out) and (lse.double() - l64).abs().max().item() <= 1e-4
    assert ops.lib().buctd_mha_fwd_bf16x6_workspace(B, 100, d) == 0


@pytest.mark.parametrize("B,T,d", [(2, 384, 48), (1, 3072, 112), (3, 128, 16), (2, 256, 128)])
def test_fused_mha_training_forward_backward_vs_fp64(dev, B, T, d):
    """attn_mha_train.hip (TransPose encoder self-attention, training; reference transpose_h.py:168-213): fused forward +
    flash-style backward against torch autograd in fp64, no dropout.  Nothing T x T is allocated."""
    from buctd_amd import ops
    g = torch.Generator().manual_seed(B * T + d + 1)
    qk = (torch.randn(B, T, 2 * d, generator=g) * 1.2).double().requires_grad_(True)
    v = torch.randn(B, T, d, generator=g).double().requires_grad_(True)
    with torch.no_grad():
        qk[:, ::7, :d] *= 3.0
    q64, k64 = qk[..., :d], qk[..., d:]
    ref = torch.softmax(q64 @ k64.transpose(1, 2) / math.sqrt(d), dim=-1) @ v
    dout = torch.randn(ref.shape, generator=g).double()
    ref.backward(dout)
    assert ops.mha_train_ok(T, d)
    qkd = qk.detach().float().to(dev).requires_grad_(True)
    vd = v.detach().float().to(dev).requires_grad_(True)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out = ops.FusedMHA.apply(qkd, vd, 0.1, False)
    out.backward(dout.float().to(dev))
    peak = torch.cuda.max_memory_allocated() - base
    if T >= 2048:          # (small T: the grow-only 1 MB workspace and the outputs themselves exceed T x T)
        assert peak < 4 * B * T * T // 2, "the fused path must not allocate anything T x T"
    for name, a, b in (("out", out.detach(), ref.detach()), ("dqk", qkd.grad, qk.grad), ("dv", vd.grad, v.grad)):
        err = _e(a, b)
        assert err <= 2e-5, f"fused MHA training {name} (B{B} T{T} d{d}): rel err {err:.2e}"


def test_fused_mha_training_dropout_matches_materialised_path(dev, monkeypatch):
    """With attention dropout the fused kernels draw the SAME mask as the materialised path for the same seed (counter hash
    keyed by (seed, (b T + q) T + key)): outputs and gradients agree to round-off."""
    from buctd_amd import ops
    B, T, d = 2, 256, 48
    g = torch.Generator().manual_seed(11)
    qk = torch.randn(B, T, 2 * d, generator=g)
    v = torch.randn(B, T, d, generator=g)
    dout = torch.randn(B, T, d, generator=g).to(dev)
    monkeypatch.setattr(ops, "next_seed", lambda: 0x1234567890ABCDEF)
    res = []
    for fused in (True, False):
        qkd, vd = qk.to(dev).requires_grad_(True), v.to(dev).requires_grad_(True)
        out = ops.FusedMHA.apply(qkd, vd, 0.3, True) if fused else ops.PositionAttention.apply(qkd, None, vd, 1, 0.3, True)
        out.backward(dout)
        res.append((out.detach(), qkd.grad, vd.grad))
    nodrop = ops.FusedMHA.apply(qk.to(dev), v.to(dev), 0.3, False)
    assert (res[0][0] - nodrop).abs().max().item() > 1e-2, "dropout must change the output"
    for name, a, b in zip(("out", "dqk", "dv"), res[0], res[1]):
        err = ((a - b).norm() / b.norm()).item()
        assert err <= 2e-5, f"fused vs materialised with dropout: {name} rel err {err:.2e}"
