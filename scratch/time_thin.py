"""solo times of the preNet's full-resolution convolutions (forward / data gradient / weight gradient): python scratch/time_thin.py [H W]"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 192)
N = 32
def tm(fn, n=10):
    for _ in range(3): fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n * 1e3
for (Ci, Co, k) in ((3, 64, 3), (64, 3, 7), (3, 3, 7)):
    x = torch.randn(N, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, k, k, device=dev) / math.sqrt(Ci * k * k)).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd(x, w, None, 1, k // 2)
    dy = torch.randn_like(y)
    gw = torch.empty_like(w)
    f = tm(lambda: ops.conv_fwd(x, w, None, 1, k // 2, stats=True))
    d = tm(lambda: ops.conv_dgrad(dy, w, tuple(x.shape), 1, k // 2))
    g = tm(lambda: ops.conv_wgrad(x, dy, w, 1, k // 2, out=gw, accumulate=0))
    fl = 2.0 * N * H * W * Ci * Co * k * k / 1e6
    print(f"{H}x{W} {Ci}->{Co} k{k}: fwd+stats {f:7.1f} us ({fl / f:5.1f} TF)  dgrad {d:7.1f} us ({fl / d:5.1f})  wgrad {g:7.1f} us ({fl / g:5.1f})")
