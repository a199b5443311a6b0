"""Solo timings (HIP events, median of 30 after 5 warm-ups) of the 3x3 conv family with fp32 inputs vs x6 planes inputs,
for the four HRNet-W48 stage-4 branch shapes at batch 32:   python scratch/time_planes.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
SHAPES = [(32, 96, 72, 48), (32, 48, 36, 96), (32, 24, 18, 192), (32, 12, 9, 384)]


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for (N, H, W, Cn) in SHAPES:
    x = torch.randn(N, H, W, Cn, device=dev)
    dy = torch.randn(N, H, W, Cn, device=dev)
    w = (torch.randn(Cn, Cn, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    gw = torch.empty_like(w)
    xp, dyp = ops.to_planes(x), ops.to_planes(dy)
    mean, invstd = torch.randn(Cn, device=dev), torch.rand(Cn, device=dev) + 0.5
    gflop = 2.0 * N * H * W * Cn * Cn * 9 / 1e9
    rows = [
        ("fwd+stats fp32-in", lambda: ops.conv_fwd(x, w, None, 1, 1, stats=True)),
        ("fwd+stats planes", lambda: ops.conv3x3_planes(xp, w, 0, Cn, stats=True)),
        ("dgrad fp32-in", lambda: ops.conv_dgrad(dy, w, tuple(x.shape), 1, 1)),
        ("dgrad planes", lambda: ops.conv3x3_planes(dyp, w, 1, Cn)),
        ("wgrad fp32-in", lambda: ops.conv_wgrad(x, dy, w, 1, 1, out=gw, accumulate=0)),
        ("wgrad planes", lambda: ops.conv_wgrad_planes(xp, dyp, gw)),
        ("to_planes", lambda: ops.to_planes(x, out=xp)),
        ("to_planes+bn", lambda: ops.to_planes(x, bn=(mean, invstd, mean, mean, True), out=xp)),
        ("bn_apply", lambda: ops.bn_apply(x, mean, invstd, mean, mean, dy, True)),
    ]
    print(f"== {N}x{H}x{W}x{Cn}  {gflop:.2f} GFLOP  (22.0 us at the 417 TF-eq roof)")
    for name, fn in rows:
        med, mn = timeit(fn)
        tf = gflop / med * 1e3 if "planes" in name and "to_" not in name or "fp32-in" in name else 0
        print(f"  {name:20s} median {med:8.1f} us  min {mn:8.1f} us" + (f"   {tf:6.1f} TF-eq  frac {tf / 416.7:.3f}" if tf else ""))
