"""3x3 forward launches of the train step, timed alone on the GPU (HIP events, 60 launches after a warm-up): the branch
convolutions of HRNet-W48 / W32 at batch N as single launches and as the group launches block.hip issues, WITH the BatchNorm
statistics accumulator (the train-mode kernels of conv3x3_lean.hip) and without (the general kernel).
    python scratch/time_c3.py [w48|w32] [N] [sets, e.g. 0 01 23 0000]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops, _C
lib = _C.lib()
dev = torch.device("cuda:0")
fam = sys.argv[1] if len(sys.argv) > 1 else "w48"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
shapes = {"w48": [(96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)],
          "w32": [(64, 48, 32), (32, 24, 64), (16, 12, 128), (8, 6, 256)]}[fam]
accb = lambda Cn: int(lib.buctd_bn_acc_bytes(Cn))
ctx = []
for (H, W, Cn) in shapes:
    c = dict(H=H, W=W, C=Cn)
    c["w"] = ops._conv3x3_prepared((torch.randn(Cn, Cn, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last), 0)
    c["x"] = [torch.randn(N, H, W, Cn, device=dev) for _ in range(4)]
    c["y"] = [torch.empty(N, H, W, Cn, device=dev) for _ in range(4)]
    c["acc"] = [torch.zeros(accb(Cn) // 8, dtype=torch.int64, device=dev) for _ in range(4)]
    ctx.append(c)
main = torch.cuda.current_stream()


def group(which, stats):
    arr = (_C.C3Conv * len(which))()
    seen = {}
    for k, i in enumerate(which):
        j = seen.get(i, 0)
        seen[i] = j + 1
        c, d = ctx[i], arr[k]
        d.N, d.H, d.W, d.Ci, d.Co = N, c["H"], c["W"], c["C"], c["C"]
        d.x, d.wprep, d.y = c["x"][j].data_ptr(), c["w"].data_ptr(), c["y"][j].data_ptr()
        d.stats_acc = c["acc"][j].data_ptr() if stats else None
    _C.check(lib.buctd_conv3x3_bf16x6_group(len(which), arr, main.cuda_stream), "group")


def run(which, stats, reps=60, warm=120):
    for _ in range(warm):
        group(which, stats)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(reps):
        group(which, stats)
    b.record(main)
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def check(which):
    """train-mode launch vs the general kernel: outputs bit for bit, accumulators vs fp64 sums of the output"""
    for c in ctx:
        for a_ in c["acc"]:
            a_.zero_()
    group(which, False)
    torch.cuda.synchronize()
    ref = {}
    seen = {}
    for i in which:
        j = seen.get(i, 0)
        seen[i] = j + 1
        ref[(i, j)] = ctx[i]["y"][j].clone()
        ctx[i]["y"][j].zero_()
    group(which, True)
    torch.cuda.synchronize()
    msg = []
    for (i, j), r in ref.items():
        y = ctx[i]["y"][j]
        Cn = ctx[i]["C"]
        w = ctx[i]["acc"][j].view(8, 4, Cn).sum(0).double()
        s1 = w[1] + w[0] * 2.0 ** -48
        s2 = w[3] + w[2] * 2.0 ** -48
        yd = y.double().reshape(-1, Cn)
        e1 = ((s1 - yd.sum(0)).abs() / yd.abs().sum(0)).max().item()
        e2 = ((s2 - (yd * yd).sum(0)).abs() / (yd * yd).sum(0)).max().item()
        msg.append(f"{i}.{j}: out equal {torch.equal(y, r)} s1 {e1:.1e} s2 {e2:.1e}")
    print("   check " + "; ".join(msg), flush=True)


sets = [[0], [0, 1], [2, 3], [0, 0], [0, 0, 0, 0], [1], [1, 1, 1, 1], [1, 2], [0, 1, 2, 3]]
if len(sys.argv) > 3:
    sets = [[int(ch) for ch in a] for a in sys.argv[3:]]
for which in sets:
    check(which)
    flops = sum(2.0 * N * ctx[i]["H"] * ctx[i]["W"] * ctx[i]["C"] ** 2 * 9 for i in which)
    t1, t0 = run(which, True), run(which, False)
    print(f"members {which}: train-mode kernel {t1:7.1f} us = {flops / t1 / 1e6 / 416.7:.3f} of the bf16x6 roof | "
          f"general kernel (no statistics) {t0:7.1f} us", flush=True)
