"""Solo per-kernel timing of the BatchNorm kernels on the four HRNet-W48 branch shapes (run under rocprofv3 --kernel-trace
--stats, or alone for HIP-event totals):  python scratch/time_bn2.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (N, H, W, Cn) in [(32, 96, 72, 48), (32, 48, 36, 96), (32, 24, 18, 192), (32, 12, 9, 384)]:
    z = torch.randn(N, H, W, Cn, device=dev)
    dy = torch.randn(N, H, W, Cn, device=dev)
    y = torch.randn(N, H, W, Cn, device=dev)
    mean, invstd = torch.randn(Cn, device=dev), torch.rand(Cn, device=dev) + 0.5
    gamma, beta = torch.randn(Cn, device=dev), torch.randn(Cn, device=dev)
    dg, db = torch.empty(Cn, device=dev), torch.empty(Cn, device=dev)
    el = N * H * W * Cn

    def t(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) * 1e3 / reps

    t_bwd_y = t(lambda: ops.bn_bwd(dy, y, z, mean, invstd, gamma, True, True, dg, db, 0))
    t_bwd_r = t(lambda: ops.bn_bwd(dy, None, z, mean, invstd, gamma, True, False, dg, db, 0, beta=beta))
    t_app = t(lambda: ops.bn_apply(z, mean, invstd, gamma, beta, y, True))
    print(f"{N}x{H}x{W}x{Cn}: bn_bwd(y,dres) {t_bwd_y:6.1f} us [{el * 32 / t_bwd_y / 1e6:5.2f} TB/s of 32 B/elem]  "
          f"bn_bwd(rebuild) {t_bwd_r:6.1f} us [{el * 20 / t_bwd_r / 1e6:5.2f} TB/s of 20 B/elem]  "
          f"bn_apply+res {t_app:6.1f} us [{el * 12 / t_app / 1e6:5.2f} TB/s]")
