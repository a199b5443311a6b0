import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
def tm(fn, n=10):
    for _ in range(3): fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n * 1e3
for H, W in ((384, 288), (256, 192)):
    x = torch.randn(32, H, W, 3, device=dev)
    w = (torch.randn(64, 3, 3, 3, device=dev) / 5).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd(x, w, None, 2, 1); dy = torch.randn_like(y); gw = torch.empty_like(w)
    print(f"stem conv1 {H}x{W}: fwd+stats {tm(lambda: ops.conv_fwd(x, w, None, 2, 1, stats=True)):.1f} us  wgrad {tm(lambda: ops.conv_wgrad(x, dy, w, 2, 1, out=gw, accumulate=0)):.1f} us")
