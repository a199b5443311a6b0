"""mask quality of the attention-dropout finisher candidates (numpy): keep rate, row/column count variance vs binomial,
correlation between neighbouring rows / columns, joint 2x2 pattern chi-square"""
import numpy as np
M32 = np.uint64(0xFFFFFFFF)
def mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & M32; x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & M32; x ^= x >> np.uint64(16)
    return x
def rot(x, r): return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & M32
def fin_old(x):
    x = x ^ rot(x, 11); x = (x * np.uint64(0x9E3779B1)) & M32; x ^= x >> np.uint64(15); return x
def fin_new(x):
    y = ((x & np.uint64(0xFFFFFF)) * np.uint64(0x9E3779)) & M32          # v_mul_u32_u24
    return ((rot(x, 24) ^ y) + np.uint64(0x85EBCA77)) & M32              # v_alignbit + v_xad_u32
def fin_none(x): return x
T = 2048
for seed in (1, 12345):
    s0, s1 = np.uint64(seed * 2654435761 & 0xFFFFFFFF), np.uint64((seed * 40503 + 77) & 0xFFFFFFFF)
    rows = np.arange(T, dtype=np.uint64); cols = np.arange(T, dtype=np.uint64)
    rk = mix32(s0 ^ ((rows * np.uint64(0x9E3779B1)) & M32)); ck = mix32((s1 + cols * np.uint64(0x85EBCA77)) & M32)
    x = (rk[:, None] + ck[None, :]) & M32
    thr = np.uint64(int(0.1 * 2 ** 32))
    for name, fin in (("old", fin_old), ("new", fin_new), ("none", fin_none)):
        keep = (fin(x) >= thr).astype(np.float64)
        p = keep.mean(); q = 1 - p
        rv = keep.sum(1).var() / (T * p * q); cv = keep.sum(0).var() / (T * p * q)
        d = 1 - keep
        def corr(a, b): return ((a - q) * (b - q)).mean() / (p * q)
        c_r = corr(d[:-1], d[1:]); c_c = corr(d[:, :-1], d[:, 1:]); c_d = corr(d[:-1, :-1], d[1:, 1:])
        # rectangle test: E[d(i,j) d(i,j') d(i',j) d(i',j')] vs q^4 (the additive structure shows up here)
        i = np.arange(0, T - 1, 2); a = d[i][:, i] * d[i][:, i + 1] * d[i + 1][:, i] * d[i + 1][:, i + 1]
        print(f"seed {seed} {name:5s}: keep {p:.5f}  row-var/binom {rv:.3f}  col-var/binom {cv:.3f}  corr row {c_r:+.4f} col {c_c:+.4f} diag {c_d:+.4f}  rect {a.mean() / q ** 4:.3f}")
