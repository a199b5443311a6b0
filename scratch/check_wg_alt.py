"""correctness of the 3x3 weight gradient of an alternative library build against fp64 (a few shapes): python scratch/run_alt.py <lib> scratch/test_wg_alt.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
worst = 0.0
for (N, H, W, Cn) in ((2, 24, 18, 48), (3, 12, 9, 96), (32, 96, 72, 48), (32, 48, 36, 96), (32, 24, 18, 192), (32, 12, 9, 384), (5, 17, 13, 64)):
    g = torch.Generator().manual_seed(H * W + Cn)
    x = torch.randn(N, Cn, H, W, generator=g); dy = torch.randn(N, Cn, H, W, generator=g)
    w = torch.zeros(Cn, Cn, 3, 3)
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=1, padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev); dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.contiguous(memory_format=torch.channels_last).to(dev)
    dw = ops.conv_wgrad(xd, dyd, wd, 1, 1)
    err = (dw.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    dw2 = ops.conv_wgrad(xd, dyd, wd, 1, 1, out=dw.clone(), accumulate=1)
    err2 = (dw2.cpu().double() - 2 * ref).abs().max().item() / ref.abs().max().item()
    worst = max(worst, err, err2)
    print(f"N{N} {H}x{W} C{Cn}: rel err {err:.2e} (accumulate {err2:.2e})", flush=True)
print("WORST", worst, "OK" if worst < 3e-6 else "FAIL")
