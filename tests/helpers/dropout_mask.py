"""Host mirror of the dropout mask of the fused position attention (buctd_amd/csrc/attn_smallqk.hip: rowkey / colkey / keepf;
the reference applies nn.Dropout to the attention matrix, lib/models/self_attention.py:84): the mask as a function of the
launch seed, for tests - keep(b, i, j) = fin(rowkey(s0, b*T + i) + colkey(s1, j)) >= p * 2^32."""
import numpy as np

_M = np.uint64(0xFFFFFFFF)


def _mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M
    x ^= x >> np.uint64(16)
    return x


def rowkey(s0, rows):
    return _mix32(np.uint64(s0) ^ ((rows.astype(np.uint64) * np.uint64(0x9E3779B1)) & _M))


def colkey(s1, cols):
    return _mix32((np.uint64(s1) + ((cols.astype(np.uint64) * np.uint64(0x85EBCA77)) & _M)) & _M)


def finish(x):
    """rotate-xor, then the low 32 bits of the 24-bit product (v_mul_u32_u24)"""
    x = x ^ (((x << np.uint64(11)) | (x >> np.uint64(21))) & _M)
    return ((x & np.uint64(0xFFFFFF)) * np.uint64(0x9E3779)) & _M


def keep_mask(seed, B, T, p, rows=None, cols=None):
    """[B, T, T] booleans (or the [rows] x [cols] block of every image): True = kept"""
    s0, s1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    th = p * 4294967296.0
    thr = np.uint64(0xFFFFFFFF if th >= 4294967295.0 else int(th))
    rows = np.arange(T) if rows is None else np.asarray(rows)
    cols = np.arange(T) if cols is None else np.asarray(cols)
    out = []
    for b in range(B):
        x = (rowkey(s0, b * T + rows)[:, None] + colkey(s1, cols)[None, :]) & _M
        out.append(finish(x) >= thr)
    return np.stack(out)
