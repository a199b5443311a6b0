"""sustained timing of one weight gradient: python scratch/time_wgrad_one.py N H W Ci Co k stride pad"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
N, H, W, Ci, Co, k, st, pad = (int(v) for v in sys.argv[1:9])
x = torch.randn(N, H, W, Ci, device=dev)
w = (torch.randn(Co, Ci, k, k, device=dev) / math.sqrt(Ci * k * k)).contiguous(memory_format=torch.channels_last)
y = ops.conv_fwd(x, w, None, st, pad)
dy = torch.randn_like(y)
gw = torch.empty_like(w)
fn = lambda: ops.conv_wgrad(x, dy, w, st, pad, out=gw, accumulate=0)
for _ in range(50): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): fn()
b.record(); b.synchronize()
us = a.elapsed_time(b) / 100 * 1e3
fl = 2.0 * y.numel() * Ci * k * k
print(f"wgrad {sys.argv[1:9]}: {us:.1f} us, {fl / us / 1e6:.1f} TFLOP/s-eq")
