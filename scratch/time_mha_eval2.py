"""per-call durations of the eval attention (clock behaviour under sustained MFMA load): python scratch/time_mha_eval2.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
B, T, d = 32, 3072, 112
qk = torch.randn(B, T, 2 * d, device=dev)
v = torch.randn(B, T, d, device=dev)
for pre in (False, True):
    ops._MHA_PRESPLIT = pre
    ops.mha_fwd(qk, v); torch.cuda.synchronize()
    time.sleep(1.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for i in range(40):
        ops.mha_fwd(qk, v); ev[i + 1].record()
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(40)]
    print("presplit" if pre else "in-kernel", " ".join(f"{u:.0f}" for u in us))
    # spaced calls: 50 ms idle between launches
    sp = []
    for i in range(8):
        time.sleep(0.05)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.mha_fwd(qk, v); b.record(); b.synchronize()
        sp.append(a.elapsed_time(b) * 1e3)
    print("   spaced:", " ".join(f"{u:.0f}" for u in sp))
