"""3x3 weight gradient (six-MFMA mode): position-major tiles + transpose reads (variant 0) against channel-major tiles +
unaligned 16-byte reads (variant 1): solo launches and the branch groups.   python scratch/time_wgrad_cm.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import _C
lib = _C.lib()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
main = torch.cuda.current_stream()
fams = {"w48": [(96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)], "w32": [(64, 48, 32), (32, 24, 64), (16, 12, 128), (8, 6, 256)]}


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(reps):
        fn()
    b.record(main)
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


res = {}
for fam, shapes in fams.items():
    ctx = [dict(H=H, W=W, C=Cn, x=torch.randn(N, H, W, Cn, device=dev), dy=torch.randn(N, H, W, Cn, device=dev)) for (H, W, Cn) in shapes]
    for var in (0, 1):
        lib.buctd_wg3_variant_(var)
        for i, c in enumerate(ctx):
            ws = torch.empty(lib.buctd_conv3x3_wgrad_bf16x6_workspace(N, c["H"], c["W"], c["C"], c["C"]), dtype=torch.uint8, device=dev)
            dw = torch.zeros(c["C"], 3, 3, c["C"], device=dev)
            f = lambda: _C.check(lib.buctd_conv3x3_wgrad_bf16x6(N, c["H"], c["W"], c["C"], c["C"], c["x"].data_ptr(), c["dy"].data_ptr(), dw.data_ptr(), 0,
                                                                ws.data_ptr(), ws.numel(), main.cuda_stream), "wgrad")
            res[(fam, i, var)] = (timed(f), dw)
        for which in ([0, 1, 2, 3], [0, 1], [1, 2], [2, 3]):
            arr = (_C.Wg3Conv * len(which))()
            keep = []
            for k, i in enumerate(which):
                c, it = ctx[i], arr[k]
                ws = torch.empty(lib.buctd_conv3x3_wgrad_bf16x6_group_workspace(len(which), N, c["H"], c["W"], c["C"], c["C"]), dtype=torch.uint8, device=dev)
                dw = torch.zeros(c["C"], 3, 3, c["C"], device=dev)
                keep += [ws, dw]
                it.N, it.H, it.W, it.Ci, it.Co = N, c["H"], c["W"], c["C"], c["C"]
                it.x, it.dy, it.dw, it.accumulate = c["x"].data_ptr(), c["dy"].data_ptr(), dw.data_ptr(), 0
                it.workspace, it.workspace_bytes = ws.data_ptr(), ws.numel()
            f = lambda: _C.check(lib.buctd_conv3x3_wgrad_bf16x6_group(len(which), arr, main.cuda_stream), "group")
            res[(fam, tuple(which), var)] = (timed(f), keep[1::2])
    for i, c in enumerate(ctx):
        (t0, d0), (t1, d1) = res[(fam, i, 0)], res[(fam, i, 1)]
        rel = ((d0 - d1).abs().max() / d0.abs().max()).item()
        print(f"{fam} branch {i} {c['C']:3d} ch @{c['H']}x{c['W']}: position-major {t0:6.1f} us, channel-major {t1:6.1f} us ({100 * (t1 / t0 - 1):+.0f} %), max rel diff {rel:.1e}", flush=True)
    for which in ([0, 1, 2, 3], [0, 1], [1, 2], [2, 3]):
        (t0, d0), (t1, d1) = res[(fam, tuple(which), 0)], res[(fam, tuple(which), 1)]
        rel = max(((a - b).abs().max() / a.abs().max()).item() for a, b in zip(d0, d1))
        solo = max(((res[(fam, i, 0)][1] - b).abs().max() / b.abs().max()).item() for i, b in zip(which, d1))
        print(f"{fam} group {which}: position-major {t0:6.1f} us, channel-major {t1:6.1f} us ({100 * (t1 / t0 - 1):+.0f} %), max rel diff {rel:.1e} (vs solo {solo:.1e})", flush=True)
