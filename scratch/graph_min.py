import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
a = torch.randn(1024, device=dev); b = torch.empty_like(a)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    b = a * 2 + 1
g.replay(); torch.cuda.synchronize(); print("torch-only graph ok", float(b[0]), float(a[0] * 2 + 1)); sys.stdout.flush()
from buctd_amd import ops
x = torch.randn(2, 16, 12, 48, device=dev); w = (torch.randn(48, 48, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
y0 = ops.conv_fwd(x, w, None, 1, 1)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    y = ops.conv_fwd(x, w, None, 1, 1)
g2.replay(); torch.cuda.synchronize(); print("custom-kernel graph ok", float((y - y0).abs().max())); sys.stdout.flush()
