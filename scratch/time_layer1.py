import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
def tm(fn, n=20):
    for _ in range(5): fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n * 1e3
for (H, W, Ci, Co) in ((96, 72, 64, 256), (96, 72, 256, 64), (64, 48, 64, 256), (64, 48, 256, 64), (32, 24, 64, 128), (16, 12, 128, 256)):
    x = torch.randn(32, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, 1, 1, device=dev) / math.sqrt(Ci)).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd(x, w, None, 1, 0); dy = torch.randn_like(y)
    y64 = torch.nn.functional.conv2d(x[:2].permute(0, 3, 1, 2).double(), w.double()).permute(0, 2, 3, 1)
    err = (y[:2].double() - y64).abs().max().item() / y64.abs().max().item()
    print(f"1x1 {Ci}->{Co} @{H}x{W}: fwd+stats {tm(lambda: ops.conv_fwd(x, w, None, 1, 0, stats=True)):6.1f} us  dgrad {tm(lambda: ops.conv_dgrad(dy, w, tuple(x.shape), 1, 0)):6.1f} us  (fwd err {err:.1e})")
