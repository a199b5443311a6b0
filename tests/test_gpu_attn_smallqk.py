"""Fused narrow-contraction position attention (buctd_amd/csrc/attn_smallqk.hip, ops.SmallQKAttention) against the
formulas of reference lib/models/self_attention.py:74-86 evaluated in fp64 on the CPU, against the materialised
T x T path (ops.PositionAttention) at the full CoAM-W48 size, and with dropout through an extracted mask.
Bar: forward and all gradients within 2e-5 relative (fp32 round-off through a softmax)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _e(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _inputs(B, T, d, C, seed, wscale=1.0):
    g = torch.Generator().manual_seed(seed)
    yq = torch.randn(B, T, d, generator=g, dtype=torch.float64)
    wq = torch.randn(C, d, generator=g, dtype=torch.float64) * wscale
    bq = torch.randn(C, generator=g, dtype=torch.float64) * wscale
    k = torch.randn(B, T, C, generator=g, dtype=torch.float64)
    v = torch.randn(B, T, C, generator=g, dtype=torch.float64)
    dout = torch.randn(B, T, C, generator=g, dtype=torch.float64)
    return yq, wq, bq, k, v, dout


def _reference(yq, wq, bq, k, v, dout, mask=None, p=0.0):
    ts = [t.clone().requires_grad_(True) for t in (yq, wq, bq, k, v)]
    yq, wq, bq, k, v = ts
    q = torch.nn.functional.linear(yq, wq, bq)
    att = torch.softmax(torch.matmul(q, k.transpose(1, 2)) / math.sqrt(k.shape[2]), -1)
    if mask is not None:
        att = att * mask / (1.0 - p)
    out = torch.matmul(att, v)
    out.backward(dout)
    return out.detach(), [t.grad for t in ts]


def _run(dev, yq, wq, bq, k, v, dout, p, training):
    from buctd_amd import ops
    yd, kd, vd = (t.float().to(dev).requires_grad_(True) for t in (yq, k, v))
    wd = torch.nn.Parameter(wq.float().to(dev))
    bd = torch.nn.Parameter(bq.float().to(dev))
    o = ops.SmallQKAttention.apply(yd, wd, bd, kd, vd, p, training)
    o.backward(dout.float().to(dev))
    return o.detach(), [yd.grad, wd.grad, bd.grad, kd.grad, vd.grad]


SHAPES = [(2, 128, 3, 48), (2, 384, 1, 16), (1, 1728, 3, 48), (2, 192, 7, 32), (1, 64, 17, 96), (2, 64, 5, 64),
          (1, 128, 14, 128), (1, 64, 3, 192)]


@pytest.mark.parametrize("mode", ["bf16x6", "fp32"])
@pytest.mark.parametrize("B,T,d,C", SHAPES)
def test_smallqk_vs_fp64(dev, B, T, d, C, mode):
    """default mode (bf16x6: three bf16 pieces per operand, six MFMAs per product - the narrow-query shapes; wider ones
    and C = 192 run the fp32 kernels) and the exact fp32 MFMA kernels: the same fp32-class bar"""
    from buctd_amd import ops
    assert ops.attn_smallqk_ok(T, d, C)
    old = ops.get_conv_math()
    ops.set_conv_math(mode)
    try:
        args = _inputs(B, T, d, C, B * 1000 + T + d)
        ref_o, ref_g = _reference(*args)
        o, grads = _run(dev, *args, 0.1, False)
        names = ["out", "dyq", "dwq", "dbq", "dk", "dv"]
        for name, a, b in zip(names, [o] + grads, [ref_o] + ref_g):
            err = _e(a, b)
            assert err <= 2e-5, f"smallqk[{mode}] {name} (B{B} T{T} d{d} C{C}): rel err {err:.2e}"
    finally:
        ops.set_conv_math(old)


@pytest.mark.parametrize("mode,bar", [("bf16x3", 5e-5), ("bf16x6", 2e-5)])
@pytest.mark.parametrize("B,T,d,C", [(2, 128, 3, 48), (2, 384, 1, 16), (1, 1728, 3, 96), (2, 192, 7, 32), (2, 64, 5, 64),
                                     (1, 64, 3, 192), (1, 128, 2, 128)])
def test_smallqk_bf16x3_vs_fp64(dev, B, T, d, C, mode, bar):
    """split-operand kernels: T- and C-contractions on the bf16 matrix cores; two pieces per operand (bf16x3 mode, bar
    5e-5 relative) or three (bf16x6, fp32 class); incl. the dropout mask agreement of forward and backward."""
    from buctd_amd import ops
    old = ops.get_conv_math()
    ops.set_conv_math(mode)
    try:
        args = _inputs(B, T, d, C, B * 1000 + T + d)
        ref_o, ref_g = _reference(*args)
        o, grads = _run(dev, *args, 0.1, False)
        for name, a, b in zip(["out", "dyq", "dwq", "dbq", "dk", "dv"], [o] + grads, [ref_o] + ref_g):
            err = _e(a, b)
            assert err <= bar, f"smallqk[{mode}] {name} (B{B} T{T} d{d} C{C}): rel err {err:.2e}"
        # dropout: forward and backward must still agree on the mask (v = I exposes it)
        if C == 64 and T == 64:
            yq, wq, bq, k, v, dout = args
            eye = torch.eye(T, dtype=torch.float64).expand(B, T, T).contiguous()
            ops.manual_seed(77)
            pd, _ = _run(dev, yq, wq, bq, k, eye, dout, 0.3, True)
            mask = (pd.cpu() != 0).double()
            ops.manual_seed(77)
            o, grads = _run(dev, yq, wq, bq, k, v, dout, 0.3, True)
            ref_o, ref_g = _reference(yq, wq, bq, k, v, dout, mask, 0.3)
            for name, a, b in zip(["out", "dyq", "dwq", "dbq", "dk", "dv"], [o] + grads, [ref_o] + ref_g):
                assert _e(a, b) <= bar, f"smallqk[{mode}]+dropout {name}: rel err {_e(a, b):.2e}"
    finally:
        ops.set_conv_math(old)


def test_smallqk_unsupported_shapes_fall_back():
    from buctd_amd import ops
    assert not ops.attn_smallqk_ok(432, 3, 48)       # T not a multiple of 64
    assert not ops.attn_smallqk_ok(128, 3, 40)       # channel count without a kernel instance
    assert not ops.attn_smallqk_ok(128, 20, 48)      # contraction too wide
    assert not ops.attn_smallqk_ok(128, 3, 48, h=2)  # multi-head


def test_smallqk_dropout_mask_consistency(dev):
    """v = I exposes the dropped attention matrix; the same seed must drive forward and backward."""
    from buctd_amd import ops
    B, T, d, C, p = 2, 64, 3, 64, 0.3
    yq, wq, bq, k, v, dout = _inputs(B, T, d, C, 5)
    eye = torch.eye(T, dtype=torch.float64).expand(B, T, T).contiguous()
    ops.manual_seed(1234)
    pd, _ = _run(dev, yq, wq, bq, k, eye, dout, p, True)
    mask = (pd.cpu() != 0).double()
    frac = 1.0 - mask.mean().item()
    assert abs(frac - p) < 0.03, f"dropped fraction {frac:.3f} vs p {p}"
    # ... and it is the documented function of the launch seed (tests/helpers/dropout_mask.py; its statistics: test_host.py)
    from tests.helpers.dropout_mask import keep_mask
    ops.manual_seed(1234)
    want = torch.from_numpy(keep_mask(ops.next_seed(), B, T, p))
    att_pos = (torch.softmax(torch.matmul(torch.nn.functional.linear(yq, wq, bq), k.transpose(1, 2)) / math.sqrt(C), -1) > 1e-30)
    assert torch.equal(mask.bool() & att_pos, want & att_pos), "the kernel's mask differs from its host mirror"
    q = torch.nn.functional.linear(yq, wq, bq)
    att = torch.softmax(torch.matmul(q, k.transpose(1, 2)) / math.sqrt(C), -1)
    assert _e(pd, att * mask / (1 - p)) <= 2e-5
    ops.manual_seed(1234)
    o, grads = _run(dev, yq, wq, bq, k, v, dout, p, True)
    ref_o, ref_g = _reference(yq, wq, bq, k, v, dout, mask, p)
    for name, a, b in zip(["out", "dyq", "dwq", "dbq", "dk", "dv"], [o] + grads, [ref_o] + ref_g):
        err = _e(a, b)
        assert err <= 2e-5, f"smallqk+dropout {name}: rel err {err:.2e}"
    # masks differ between images and between seeds
    assert not torch.equal(mask[0], mask[1])
    ops.manual_seed(99)
    pd2, _ = _run(dev, yq, wq, bq, k, eye, dout, p, True)
    assert not torch.equal(pd2.cpu() != 0, pd.cpu() != 0)


def test_smallqk_matches_materialised_path_full_size(dev):
    """CoAM-W48 stage-2 size (T = 96*72, C = 48, colored condition d = 3): fused vs the T x T path, fwd + bwd, with
    the weight scale the reference initialises fc_q with (std 1e-3) and with O(1) logits."""
    from buctd_amd import ops, nn
    B, T, d, C = 2, 6912, 3, 48
    for wscale in (1e-3, 0.5):
        yq, wq, bq, k, v, dout = (t.float().to(dev) for t in _inputs(B, T, d, C, 11, wscale))
        res = []
        for fused in (True, False):
            yd, kd, vd = (t.clone().requires_grad_(True) for t in (yq, k, v))
            wd, bd = torch.nn.Parameter(wq.clone()), torch.nn.Parameter(bq.clone())
            if fused:
                o = ops.SmallQKAttention.apply(yd, wd, bd, kd, vd, 0.1, False)
            else:
                lin = nn.Linear(d, C).to(dev)
                lin.weight, lin.bias = wd, bd
                o = ops.PositionAttention.apply(lin(yd), kd, vd, 1, 0.1, False)
            o.backward(dout)
            res.append([o.detach(), yd.grad, wd.grad, bd.grad, kd.grad, vd.grad])
        for name, a, b in zip(["out", "dyq", "dwq", "dbq", "dk", "dv"], *res):
            err = _e(a, b)
            assert err <= 5e-5, f"fused vs materialised {name} (wscale {wscale}): rel err {err:.2e}"


def test_position_attention_module_uses_fused_path(dev):
    """PositionAttentionModule forward/backward: fused path == materialised path on the same module (eval dropout)."""
    from buctd_amd.models.pose_hrnet_coam import PositionAttentionModule
    torch.manual_seed(3)
    m = PositionAttentionModule(d_model=48, d_cond=3, kernel_size=3, H=16, W=24, n_heads=1).to(dev)
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.2)
    m.eval()
    x = torch.randn(2, 16, 24, 48, device=dev)
    cond = torch.randn(2, 16, 24, 3, device=dev)
    outs = []
    for fused in (True, False):
        m.pa.fused = fused
        m.zero_grad(set_to_none=True)
        xd, cd = x.clone().requires_grad_(True), cond.clone().requires_grad_(True)
        y = m(xd, cd)
        y.backward(torch.ones_like(y) * 0.01 + y.detach() * 0.1)
        outs.append([y.detach(), xd.grad, cd.grad] + [p.grad.clone() for p in m.parameters()])
    names = ["out", "dx", "dcond"] + [n for n, _ in m.named_parameters()]
    for name, a, b in zip(names, *outs):
        if name == "pa.fc_k.bias":
            continue  # mathematically zero (a common shift of all keys leaves every softmax row unchanged)
        assert _e(a, b) <= 5e-5, f"{name}: fused vs materialised rel err {_e(a, b):.2e}"


@pytest.mark.slow
def test_smallqk_speed_report(dev):
    """Not a pass/fail bar: prints fused vs materialised timings at the CoAM-W48 size, batch 32."""
    from buctd_amd import ops
    B, T, d, C = 32, 6912, 3, 48
    g = torch.Generator().manual_seed(0)
    yq = torch.randn(B, T, d, generator=g).to(dev).requires_grad_(True)
    k = torch.randn(B, T, C, generator=g).to(dev).requires_grad_(True)
    v = torch.randn(B, T, C, generator=g).to(dev).requires_grad_(True)
    wq = torch.nn.Parameter((torch.randn(C, d, generator=g) * 0.1).to(dev))
    bq = torch.nn.Parameter(torch.zeros(C).to(dev))
    dout = torch.randn(B, T, C, generator=g).to(dev)
    for it in range(3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        o = ops.SmallQKAttention.apply(yq, wq, bq, k, v, 0.1, True)
        e[1].record()
        o.backward(dout)
        e[2].record()
        torch.cuda.synchronize()
        if it == 2:
            print(f"smallqk attention B{B} T{T} C{C}: fwd {e[0].elapsed_time(e[1]):.2f} ms, bwd {e[1].elapsed_time(e[2]):.2f} ms")
