"""long sustained run: where do the per-call durations settle?  python scratch/time_mha_eval3.py [q16]"""
import os, sys, time, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
B, T, d = 32, 3072, 112
qk = torch.randn(B, T, 2 * d, device=dev)
v = torch.randn(B, T, d, device=dev)
for pre in (True, False):
    ops._MHA_PRESPLIT = pre
    ops.mha_fwd(qk, v); torch.cuda.synchronize()
    n = 1500
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n // 50 + 1)]
    ev[0].record()
    for i in range(n):
        ops.mha_fwd(qk, v)
        if (i + 1) % 50 == 0: ev[(i + 1) // 50].record()
        if i == n - 200:
            p = subprocess.run("rocm-smi --showclocks --showpower | grep -E 'sclk|Power'", shell=True, capture_output=True, text=True)
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 / 50 for i in range(n // 50)]
    print("presplit" if pre else "in-kernel", " ".join(f"{u:.0f}" for u in us))
    print(p.stdout)
