import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
def tm(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (H, W) in ((384, 288), (256, 192)):
    for Ci, Co in ((64, 3), (3, 3)):
        x = torch.randn(32, H, W, Ci, device=dev)
        w = (torch.randn(Co, Ci, 7, 7, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
        t = tm(lambda: ops.conv_fwd(x, w, None, 1, 3, stats=True))
        print(f"{H}x{W} {Ci}->{Co} k7 fwd+stats: {t:.0f} us", flush=True)
