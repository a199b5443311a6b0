"""The BatchNorm streaming launches of the HRNet-W48 branches at batch N, timed alone (HIP events, 60 launches after a warm-up):
the group applies of the train step - forward apply (+ residual, ReLU), bn2 backward apply (dy, y, z -> dz, dres), bn1 backward
apply (dy, z -> dz; mask rebuilt) - on the accumulator forms, with the bytes they move and the rate, and a plain copy of the
same number of bytes for scale.
    python scratch/time_bn_stream.py [N] [sets, e.g. 01 23 0 0123]      (scratch/run_alt.py for another build)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import _C
lib = _C.lib()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)]
main = torch.cuda.current_stream()
sp = C.c_void_p(main.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
ctx = []
for (H, W, Cn) in shapes:
    rows = N * H * W
    c = dict(rows=rows, C=Cn)
    for k in ("dy", "y", "z", "x"):
        c[k] = torch.randn(rows, Cn, device=dev)
    for k in ("dz", "dres", "out"):
        c[k] = torch.empty(rows, Cn, device=dev)
    for k in ("mean", "beta", "dg", "db", "rm"):
        c[k] = torch.zeros(Cn, device=dev)
    for k in ("invstd", "gamma", "rv"):
        c[k] = torch.ones(Cn, device=dev)
    nacc = int(lib.buctd_bn_acc_bytes(Cn)) // 8
    # forward accumulator: the sums of z in the kernel's fixed point (shard 0 only; rows * 1.0 for sum z^2 so that var > 0)
    c["acc_b"] = torch.zeros(nacc, dtype=torch.int64, device=dev)
    c["acc_f"] = torch.zeros(nacc, dtype=torch.int64, device=dev)
    c["acc_f"].view(8, 4, Cn)[0, 3] = rows          # hi word of sum z^2 = rows (mean 0, var 1)
    c["mo"], c["io"] = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
    ctx.append(c)


def bwd(which, second):
    it = (_C.BnBwdItem * len(which))()
    for k, i in enumerate(which):
        c, d = ctx[i], it[k]
        d.dy, d.z, d.mean, d.invstd, d.gamma = p(c["dy"]), p(c["z"]), p(c["mean"]), p(c["invstd"]), p(c["gamma"])
        d.y = p(c["y"]) if second else None
        d.beta = None if second else p(c["beta"])
        d.relu, d.rows, d.C = 1, c["rows"], c["C"]
        d.dz, d.dres = p(c["dz"]), (p(c["dres"]) if second else None)
        d.dgamma, d.dbeta, d.accumulate = p(c["dg"]), p(c["db"]), 0
        d.acc, d.acc_ready = p(c["acc_b"]), 1
    _C.check(lib.buctd_bn_bwd_acc_group(len(which), it, sp), "bwd")


def fwd(which):
    it = (_C.BnApplyItem * len(which))()
    for k, i in enumerate(which):
        c, d = ctx[i], it[k]
        d.z, d.gamma, d.beta, d.residual, d.relu, d.y = p(c["z"]), p(c["gamma"]), p(c["beta"]), p(c["x"]), 1, p(c["out"])
        d.rows, d.C = c["rows"], c["C"]
        d.st.acc, d.st.rows, d.st.eps, d.st.momentum = p(c["acc_f"]), c["rows"], 1e-5, 0.1
        d.st.mean_out, d.st.invstd_out, d.st.running_mean, d.st.running_var = p(c["mo"]), p(c["io"]), p(c["rm"]), p(c["rv"])
    _C.check(lib.buctd_bn_apply_acc_group(len(which), it, sp), "fwd")


def run(fn, reps=60, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(reps):
        fn()
    b.record(main)
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


sets = [[0, 1], [2, 3], [0], [0, 1, 2, 3]]
if len(sys.argv) > 2:
    sets = [[int(ch) for ch in a] for a in sys.argv[2:]]
for which in sets:
    el = sum(ctx[i]["rows"] * ctx[i]["C"] for i in which)
    src = torch.empty(el * 5, device=dev)
    dst = torch.empty(el * 5, device=dev)
    out = []
    for name, fn, bpe in (("forward apply (z, res -> y)", lambda: fwd(which), 12),
                          ("bn2 backward apply (dy, y, z -> dz, dres)", lambda: bwd(which, True), 20),
                          ("bn1 backward apply (dy, z -> dz)", lambda: bwd(which, False), 12)):
        t = run(fn)
        n = el * bpe // 8            # a copy moving the same bytes: n floats read + n floats written
        tc = run(lambda: dst[:n].copy_(src[:n]))
        out.append(f"   {name:44s} {t:7.1f} us = {el * bpe / t / 1e6:5.2f} TB/s   (copy of the same bytes {tc:6.1f} us = {el * bpe / tc / 1e6:5.2f} TB/s)")
    print(f"members {which}, {el / 1e6:.2f} M elements:\n" + "\n".join(out), flush=True)
