"""sustained timing of one gathered convolution: python scratch/time_gconv.py N H W Ci Co k stride pad [dgrad]"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
N, H, W, Ci, Co, k, st, pad = (int(v) for v in sys.argv[1:9])
dg = len(sys.argv) > 9
x = torch.randn(N, H, W, Ci, device=dev)
w = (torch.randn(Co, Ci, k, k, device=dev) / math.sqrt(Ci * k * k)).contiguous(memory_format=torch.channels_last)
y = ops.conv_fwd(x, w, None, st, pad)
dy = torch.randn_like(y)
fn = (lambda: ops.conv_dgrad(dy, w, tuple(x.shape), st, pad)) if dg else (lambda: ops.conv_fwd(x, w, None, st, pad, stats=True))
NW, NT = (3, 6) if os.environ.get('PMC_SHORT') else (200, 300)
for _ in range(NW): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(NT): fn()
b.record(); b.synchronize()
us = a.elapsed_time(b) / NT * 1e3
fl = 2.0 * y.numel() * Ci * k * k
print(f"{os.environ.get('BUCTD_LIB_ALT', 'product')}: {'dgrad' if dg else 'fwd'} {sys.argv[1:9]}: {us:.1f} us, {fl / us / 1e6:.1f} TFLOP/s-eq")
