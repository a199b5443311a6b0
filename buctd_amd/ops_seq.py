"""Autograd Functions of the TransPose encoder and the ResNet stem on top of buctd_amd.ops:
LayerNorm, element-wise dropout, position-embedding add, channel concat, max-pool, and the
nn.MultiheadAttention input projection (reference lib/models/transpose_h.py:168-243, 668-681;
lib/models/pose_resnet.py:122).  Every numeric step is a libbuctd_hip.so kernel."""
import ctypes as C

import torch

from . import ops
from ._C import check, lib, ptr, stream_ptr


def copy_channels(src, cs0, dst, cd0, cc):
    rows = src.numel() // src.shape[-1]
    check(lib().buctd_copy_channels(ptr(src), rows, src.shape[-1], cs0, ptr(dst), dst.shape[-1], cd0, cc, stream_ptr()),
          "copy_channels")


def add_bcast(a, v):
    out = torch.empty_like(a)
    n = v.numel()
    check(lib().buctd_add_bcast(ptr(a), ptr(v), ptr(out), a.numel() // n, n, stream_ptr()), "add_bcast")
    return out


class ConcatChannels(torch.autograd.Function):
    """torch.cat((a, b), dim=channel) on NHWC tensors."""

    @staticmethod
    def forward(ctx, a, b):
        ca, cb = a.shape[-1], b.shape[-1]
        out = torch.empty(a.shape[:-1] + (ca + cb,), dtype=torch.float32, device=a.device)
        copy_channels(a, 0, out, 0, ca)
        copy_channels(b, 0, out, ca, cb)
        ctx.dims = (tuple(a.shape), tuple(b.shape))
        return out

    @staticmethod
    def backward(ctx, dy):
        sa, sb = ctx.dims
        dy = dy.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty(sa, dtype=torch.float32, device=dy.device)
            copy_channels(dy, 0, da, 0, sa[-1])
        if ctx.needs_input_grad[1]:
            db = torch.empty(sb, dtype=torch.float32, device=dy.device)
            copy_channels(dy, sa[-1], db, 0, sb[-1])
        return da, db


class AddPos(torch.autograd.Function):
    """tokens [B,T,d] + pos [T,d] (sine embedding: a constant, no gradient - transpose_h.py:507-511)."""

    @staticmethod
    def forward(ctx, x, pos):
        return add_bcast(x, pos)

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("learnable position embeddings are not used by any BUCTD recipe")
        return dy, None


class Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, training):
        if not training or p == 0.0:
            ctx.seed = None
            return x
        ctx.seed, ctx.p = ops.next_seed(), p
        return ops.dropout(x.contiguous(), p, ctx.seed)

    @staticmethod
    def backward(ctx, dy):
        if ctx.seed is None:
            return dy, None, None
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed), None, None  # same mask, same 1/(1-p) scale


class AddLayerNorm(torch.autograd.Function):
    """LayerNorm(a + b) with affine parameters (post-norm residual, transpose_h.py:204-209)."""

    @staticmethod
    def forward(ctx, a, b, ln):
        keep = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])     # backward() runs only then (it also owns ln's gradients)
        y, mean, invstd, s = ops.add_layernorm_fwd(a.contiguous(), b.contiguous(), ln.weight, ln.bias, ln.eps, keep_sum=keep)
        ctx.ln = ln
        if keep:
            ctx.save_for_backward(s, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        s, mean, invstd = ctx.saved_tensors
        ln = ctx.ln
        dg, acc = ops.grad_target(ln.weight)
        db, acc2 = ops.grad_target(ln.bias)
        assert acc == acc2
        ds = ops.layernorm_bwd(dy.contiguous(), s, mean, invstd, ln.weight, dg, db, acc)
        ops.grad_done(ln.weight, ln.bias)
        return ds, ds, None


class InProjection(torch.autograd.Function):
    """nn.MultiheadAttention input projection with q = k = src + pos and v = src:
    qk = (src+pos) W[:2d]^T + b[:2d],  v = src W[2d:]^T + b[2d:]  (transpose_h.py:192-197)."""

    @staticmethod
    def forward(ctx, qin, src, w, b):
        d = w.shape[1]
        B, T, _ = src.shape
        qk = ops.conv_fwd(qin.view(B, 1, T, d), w[: 2 * d], b[: 2 * d], 1, 0)
        v = ops.conv_fwd(src.view(B, 1, T, d), w[2 * d:], b[2 * d:], 1, 0)
        ctx.pw, ctx.pb = w, b
        ctx.save_for_backward(qin, src)
        return qk.view(B, T, 2 * d), v.view(B, T, d)

    @staticmethod
    def backward(ctx, dqk, dv):
        qin, src = ctx.saved_tensors
        w, b = ctx.pw, ctx.pb
        d = w.shape[1]
        B, T, _ = src.shape
        dqk = dqk.contiguous().view(B, 1, T, 2 * d)
        dv = dv.contiguous().view(B, 1, T, d)
        x4 = (B, 1, T, d)
        dqin = ops.conv_dgrad(dqk, w[: 2 * d], x4, 1, 0).view(B, T, d)
        dsrc = ops.conv_dgrad(dv, w[2 * d:], x4, 1, 0).view(B, T, d)
        gw, acc = ops.grad_target(w)
        gb, accb = ops.grad_target(b)
        ops.conv_wgrad(qin.view(x4), dqk, w[: 2 * d], 1, 0, out=gw[: 2 * d], accumulate=acc)
        ops.conv_wgrad(src.view(x4), dv, w[2 * d:], 1, 0, out=gw[2 * d:], accumulate=acc)
        ops.colsum(dqk, 2 * d, gb[: 2 * d], accb)
        ops.colsum(dv, d, gb[2 * d:], accb)
        ops.grad_done(w, b)
        return dqin, dsrc, None, None


class MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, idx = ops.maxpool3x3s2_fwd(x)
        ctx.save_for_backward(idx)
        ctx.xshape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool3x3s2_bwd(dy.contiguous(), idx, ctx.xshape)
