"""solo timings: partial-statistics launches (+ finalize) against the accumulator forms.  python scratch/time_bn_acc.py"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from buctd_amd import ops, _C
from buctd_amd._C import lib, ptr, check
dev = torch.device("cuda:0")
def timeit(fn, n=60, warm=80):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
class Bn:
    def __init__(s, Cn):
        s.weight = torch.ones(Cn, device=dev); s.bias = torch.zeros(Cn, device=dev); s.running_mean = torch.zeros(Cn, device=dev)
        s.running_var = torch.ones(Cn, device=dev); s.eps = 1e-5; s.momentum = 0.1; s.track_running_stats = True
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (N, H, W, Cn) in ((32, 96, 72, 48), (32, 48, 36, 96), (32, 24, 18, 192), (32, 12, 9, 384)):
    x = torch.randn(N, H, W, Cn, device=dev); w = (torch.randn(Cn, Cn, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    bn = Bn(Cn); rows = N * H * W
    z, part, info = ops.conv_fwd(x, w, None, 1, 1, stats=True)
    mean, invstd = ops.bn_finalize(part, info, rows, Cn, 1e-5, 0.1, bn.running_mean, bn.running_var)
    res = {}
    res["fwd plain"] = timeit(lambda: ops.conv_fwd(x, w, None, 1, 1))
    res["fwd partials"] = timeit(lambda: ops.conv_fwd(x, w, None, 1, 1, stats=True))
    res["fwd acc"] = timeit(lambda: ops.conv_fwd(x, w, None, 1, 1, stats="acc"))
    res["finalize"] = timeit(lambda: ops.bn_finalize(part, info, rows, Cn, 1e-5, 0.1, bn.running_mean, bn.running_var))
    res["fwd bnin arrays + partials"] = timeit(lambda: ops.conv_fwd(z, w, None, 1, 1, stats=True, in_bn=(mean, invstd, bn.weight, bn.bias, True)))
    z1, acc1, _ = ops.conv_fwd(x, w, None, 1, 1, stats="acc")
    bi = ops.BnAccInput(acc1, rows, bn, True)
    res["fwd bnin acc + acc"] = timeit(lambda: ops.conv_fwd(z, w, None, 1, 1, stats="acc", in_bn=bi))
    res["fwd bnin acc, no stats"] = timeit(lambda: ops.conv_fwd(z, w, None, 1, 1, in_bn=bi))
    res["bn_apply(+res,relu)"] = timeit(lambda: ops.bn_apply(z, mean, invstd, bn.weight, bn.bias, x, True))
    res["bn_apply_acc(+res,relu)"] = timeit(lambda: ops.bn_apply_acc(z, bi, x, True))
    dg, db = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
    ws = torch.empty(int(lib().buctd_bn_bwd_workspace(rows, Cn)), dtype=torch.uint8, device=dev)
    dz, dres = torch.empty_like(z), torch.empty_like(z)
    y = ops.bn_apply(z, mean, invstd, bn.weight, bn.bias, x, True)
    def old_bwd():
        check(lib().buctd_bn_bwd(ptr(x), ptr(y), ptr(z), ptr(mean), ptr(invstd), ptr(bn.weight), None, 1, rows, Cn, ptr(dz), ptr(dres), ptr(dg), ptr(db), 0, ptr(ws), ws.numel(), sp), "b")
    res["bn_bwd old (reduce+finalize+apply)"] = timeit(old_bwd)
    res["bn_bwd acc (reduce+apply)"] = timeit(lambda: ops.bn_bwd(x, y, z, mean, invstd, bn.weight, True, True, dg, db, 0))
    a = ops.AccRef(Cn, dev)
    def acc_ready():
        check(lib().buctd_bn_bwd_acc(ptr(x), ptr(y), ptr(z), ptr(mean), ptr(invstd), ptr(bn.weight), None, 1, rows, Cn, ptr(dz), ptr(dres), ptr(dg), ptr(db), 0, C.c_void_p(a.ptr), 1, sp), "b")
    res["bn_bwd acc apply only"] = timeit(acc_ready)
    print(f"--- {N}x{H}x{W}x{Cn}")
    for k, v in res.items(): print(f"  {k:40s} {v:8.1f} us")
