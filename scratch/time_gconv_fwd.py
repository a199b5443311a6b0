"""Forward (with statistics accumulator) and 1x1 data gradient of the fuse-layer / transition convolutions at the CoAM-W48 shapes
(run once per library build: python scratch/run_alt.py <lib> scratch/time_gconv_fwd.py).   python scratch/time_gconv_fwd.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
N = 32


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    ops.acc_pool.reset(dev)
    return a.elapsed_time(b) / reps * 1e3


tot = [0.0, 0.0]
for (H, W, Ci, Co, k) in [(48, 36, 96, 48, 1), (24, 18, 192, 48, 1), (12, 9, 384, 48, 1), (24, 18, 192, 96, 1), (12, 9, 384, 96, 1), (12, 9, 384, 192, 1),
                          (96, 72, 48, 96, 3), (96, 72, 48, 48, 3), (48, 36, 48, 192, 3), (48, 36, 96, 192, 3), (48, 36, 48, 48, 3), (24, 18, 48, 384, 3),
                          (48, 36, 96, 96, 3), (24, 18, 96, 384, 3), (24, 18, 192, 384, 3), (96, 72, 256, 96, 3), (192, 144, 64, 64, 3)]:
    stride, pad = (1, 0) if k == 1 else (2, 1)
    x = torch.randn(N, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    tf = timeit(lambda: ops.conv_fwd(x, w, None, stride, pad, stats="acc"))
    msg = f"{k}x{k} {Ci:3d}->{Co:3d} @{H}x{W}: fwd+stats {tf:6.1f} us"
    tot[0] += tf
    Ho, Wo = (H, W) if k == 1 else (H // 2, W // 2)
    dy = torch.randn(N, Ho, Wo, Co, device=dev)
    td = timeit(lambda: ops.conv_dgrad(dy, w, tuple(x.shape), stride, pad))
    msg += f", dgrad {td:6.1f} us"
    tot[1] += td
    print(msg, flush=True)
print(f"sum: fwd {tot[0]:.0f} us, dgrad {tot[1]:.0f} us")
