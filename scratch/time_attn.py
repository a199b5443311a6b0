"""CoAM position attention at the C4 shape (B 32, T 6912, C 48, d 3): forward and backward wall time, HIP events.
   python scratch/time_attn.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
B, T, d, Cc = 32, 6912, 3, 48
yq = torch.randn(B, T, d, device=dev, requires_grad=True)
k = torch.randn(B, T, Cc, device=dev, requires_grad=True)
v = torch.randn(B, T, Cc, device=dev, requires_grad=True)
wq = torch.nn.Parameter(torch.randn(Cc, d, device=dev) * 0.1); bq = torch.nn.Parameter(torch.zeros(Cc, device=dev))
dout = torch.randn(B, T, Cc, device=dev)
def timed(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
with torch.no_grad():
    tf = timed(lambda: ops.SmallQKAttention.apply(yq, wq, bq, k, v, 0.1, True))
tfb = timed(lambda: ops.SmallQKAttention.apply(yq, wq, bq, k, v, 0.1, True).backward(dout))
print(f"position attention forward {tf:.3f} ms, forward + backward {tfb:.3f} ms (backward {tfb - tf:.3f})")
