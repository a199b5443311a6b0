"""How much does running the four HRNet-W48 branch convolutions on separate HIP streams buy over running them one after
another?  (each stream: a chain of back-to-back 3x3 convolutions of its branch shape)   python scratch/time_branch_concurrency.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops, _C
lib = _C.lib()
dev = torch.device("cuda:0")
shapes = [(96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
N = 32
ctx = []
for (H, W, Cn) in shapes:
    x = torch.randn(N, H, W, Cn, device=dev); out = torch.empty_like(x)
    w = (torch.randn(Cn, Cn, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    wf = ops._conv3x3_prepared(w, 0)
    ng, rpg = C.c_int(), C.c_int()
    lib.buctd_conv3x3_bf16x6_stats_groups(N, H, W, Cn, Cn, C.byref(ng), C.byref(rpg))
    part = torch.empty(ng.value * Cn * 2, device=dev); cnt = torch.empty(ng.value, dtype=torch.int32, device=dev)
    ctx.append((H, W, Cn, x, out, wf, part, cnt))
def conv(c, st):
    H, W, Cn, x, out, wf, part, cnt = c
    lib.buctd_conv3x3_bf16x6(N, H, W, Cn, Cn, P(x), P(wf), None, None, None, None, 0, P(out), P(part), P(cnt), C.c_void_p(st.cuda_stream))
main = torch.cuda.current_stream()
streams = [torch.cuda.Stream() for _ in shapes]
def run(par, which, reps=40):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    if par:
        for st in streams: st.wait_stream(main)
        for _ in range(reps):
            for i in which: conv(ctx[i], streams[i])
        for i in which: main.wait_stream(streams[i])
    else:
        for _ in range(reps):
            for i in which: conv(ctx[i], main)
    b.record(main); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for which in ([0, 1, 2, 3], [0, 1], [1, 2], [0, 0], [1, 1]):
    for _ in range(3): run(False, which); run(True, which)
    s = run(False, which); p = run(True, which)
    print(f"branches {which}: serial {s:.1f} us per round, on separate streams {p:.1f} us ({100 * (1 - p / s):.0f} % saved)", flush=True)
