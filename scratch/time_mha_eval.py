"""eval attention kernels, one TransPose-A6 encoder layer shape: python scratch/time_mha_eval.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, d = 3072, 112
qk = torch.randn(B, T, 2 * d, device=dev)
v = torch.randn(B, T, d, device=dev)
fl = 4.0 * B * T * T * d
for name, pre in (("in-kernel split", False), ("pre-split + DMA", True)):
    ops._MHA_PRESPLIT = pre
    for _ in range(3): ops.mha_fwd(qk, v)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.mha_fwd(qk, v)
    b.record(); b.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"{name}: {us:.1f} us per layer call (B={B}), {fl / us / 1e6:.1f} TFLOP/s-eq = {fl / us / 1e6 / 416.7:.3f} of 416.7")
