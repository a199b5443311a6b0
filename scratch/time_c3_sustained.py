"""sustained solo timing of the bf16x6 3x3 kernels on the four HRNet-W48 branch shapes (forward with statistics, data gradient,
weight gradient): python scratch/time_c3_sustained.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
for (H, W, Cn) in ((96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)):
    N = 32
    x = torch.randn(N, H, W, Cn, device=dev); dy = torch.randn(N, H, W, Cn, device=dev)
    w = (torch.randn(Cn, Cn, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    gw = torch.empty_like(w)
    out = []
    for name, fn in (("fwd", lambda: ops.conv_fwd(x, w, None, 1, 1, stats=True)), ("dgrad", lambda: ops.conv_dgrad(dy, w, tuple(x.shape), 1, 1)),
                     ("wgrad", lambda: ops.conv_wgrad(x, dy, w, 1, 1, out=gw, accumulate=0))):
        for _ in range(150): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(300): fn()
        b.record(); b.synchronize()
        out.append(f"{name} {a.elapsed_time(b) / 300 * 1e3:.1f} us")
    print(f"{H}x{W} C{Cn}: " + ", ".join(out), flush=True)
