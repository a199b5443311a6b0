"""A few launches of one roofline op for rocprofv3 (--pmc / --kernel-trace):
    python scratch/one_op.py <mode: bf16x6|bf16x3|fp32> <kind: fwd|dgrad|wgrad> [N H W Ci Co] [launches]
fwd = forward with the BN-statistics epilogue, as in the train step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops  # noqa: E402

mode, kind = sys.argv[1], sys.argv[2]
N, H, W, Ci, Co = [int(v) for v in sys.argv[3:8]] if len(sys.argv) > 7 else (32, 96, 72, 48, 48)
launches = int(sys.argv[8]) if len(sys.argv) > 8 else 10
ops.set_conv_math(mode)
dev = torch.device("cuda:0")
x = torch.randn(N, H, W, Ci, device=dev)
dy = torch.randn(N, H, W, Co, device=dev)
w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
gw = torch.empty_like(w)
fn = {"fwd": lambda: ops.conv_fwd(x, w, None, 1, 1, stats=True),
      "dgrad": lambda: ops.conv_dgrad(dy, w, tuple(x.shape), 1, 1),
      "wgrad": lambda: ops.conv_wgrad(x, dy, w, 1, 1, out=gw, accumulate=0)}[kind]
fn()
torch.cuda.synchronize()
for _ in range(launches):
    fn()
torch.cuda.synchronize()
