"""The four HRNet branch convolutions of one layer-step: four launches one after another / on four streams / as ONE group
launch (buctd_conv3x3_bf16x6_group).  Also checks the group launch bit for bit against the single launches (output and
statistics accumulators).   python scratch/time_group_conv.py [w48|w32] [N]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops, _C
lib = _C.lib()
dev = torch.device("cuda:0")
fam = sys.argv[1] if len(sys.argv) > 1 else "w48"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
shapes = {"w48": [(96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)],
          "w32": [(64, 48, 32), (32, 24, 64), (16, 12, 128), (8, 6, 256)]}[fam]
ctx = []
for (H, W, Cn) in shapes:
    x = torch.randn(N, H, W, Cn, device=dev)
    w = (torch.randn(Cn, Cn, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    wf = ops._conv3x3_prepared(w, 0)
    res = torch.randn(N, H, W, Cn, device=dev)
    ctx.append(dict(H=H, W=W, C=Cn, x=x, wf=wf, res=res))
accb = lambda Cn: int(lib.buctd_bn_acc_bytes(Cn))


def single(c, out, acc, st, res=False):
    lib.buctd_conv3x3_bf16x6_acc(N, c["H"], c["W"], c["C"], c["C"], c["x"].data_ptr(), c["wf"].data_ptr(),
                                 c["res"].data_ptr() if res else None, 0, out.data_ptr(), acc.data_ptr() if acc is not None else None,
                                 None, None, None, 0, st.cuda_stream)


def group(which, outs, accs, st, res=False):
    arr = (_C.C3Conv * len(which))()
    for k, i in enumerate(which):
        c, d = ctx[i], arr[k]
        d.N, d.H, d.W, d.Ci, d.Co = N, c["H"], c["W"], c["C"], c["C"]
        d.x, d.wprep, d.y = c["x"].data_ptr(), c["wf"].data_ptr(), outs[k].data_ptr()
        d.residual = c["res"].data_ptr() if res else None
        d.stats_acc = accs[k].data_ptr() if accs is not None else None
    _C.check(lib.buctd_conv3x3_bf16x6_group(len(which), arr, st.cuda_stream), "group")


main = torch.cuda.current_stream()
# ---- bit identity
which = list(range(len(shapes)))
o1 = [torch.empty_like(ctx[i]["x"]) for i in which]
o2 = [torch.empty_like(ctx[i]["x"]) for i in which]
a1 = [torch.zeros(accb(ctx[i]["C"]) // 8, dtype=torch.int64, device=dev) for i in which]
a2 = [torch.zeros(accb(ctx[i]["C"]) // 8, dtype=torch.int64, device=dev) for i in which]
for k, i in enumerate(which):
    single(ctx[i], o1[k], a1[k], main, res=True)
group(which, o2, a2, main, res=True)
torch.cuda.synchronize()
for k in range(len(which)):
    Cn = ctx[which[k]]["C"]
    s1 = a1[k].view(8, 4, Cn).sum(0)
    s2 = a2[k].view(8, 4, Cn).sum(0)
    print(f"conv {k}: outputs equal {torch.equal(o1[k], o2[k])}, accumulators equal {torch.equal(s1, s2)}", flush=True)

streams = [torch.cuda.Stream() for _ in shapes]


def run(mode, which, reps=40):
    outs = [o1[i] if len(set(which)) == len(which) else o1[i + 4 * k] for k, i in enumerate(which)]
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    if mode == "streams":
        for st in streams:
            st.wait_stream(main)
        for _ in range(reps):
            for k, i in enumerate(which):
                single(ctx[i], outs[k], None, streams[i])
        for i in which:
            main.wait_stream(streams[i])
    elif mode == "serial":
        for _ in range(reps):
            for k, i in enumerate(which):
                single(ctx[i], outs[k], None, main)
    else:
        for _ in range(reps):
            group(which, outs, None, main)
    b.record(main)
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


sets = ([0, 1, 2, 3], [0, 1, 2], [0, 1], [2, 3], [0], [1], [2], [3])
if len(sys.argv) > 3:
    sets = [[int(c) for c in a] for a in sys.argv[3:]]
    o1 = o1 * 4
for which in sets:
    for _ in range(2):
        for m in ("serial", "streams", "group"):
            run(m, which)
    r = {m: run(m, which) for m in ("serial", "streams", "group")}
    print(f"branches {which}: serial {r['serial']:.1f} us, streams {r['streams']:.1f} us, group {r['group']:.1f} us "
          f"({100 * (1 - r['group'] / r['serial']):.0f} % vs serial)", flush=True)
