"""A few launches of one roofline op for rocprofv3 (--pmc / --kernel-trace):
    python scratch/one_op.py <mode: bf16x6|bf16x3|fp32> <kind: fwd|dgrad|wgrad> [N H W Ci Co] [launches]
fwd = forward with the BN-statistics epilogue, as in the train step."""
import os
import sys

import torch


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops  # noqa: E402

mode, kind = sys.argv[1], sys.argv[2]
if kind in ("gemm", "mha", "attn"):
    # python scratch/one_op.py bf16x6 gemm|mha|attn [launches]: fc_o forward product (T = 6912, N = 32 x 48), fused MHA
    # (T = 3072, d = 112, N = 32), CoAM position attention forward + backward (T = 6912, C = 48, N = 32)
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    ops.set_conv_math(mode)
    dev = torch.device("cuda:0")
    if kind == "gemm":
        Bn, T, Cn = 32, 6912, 48
        W = torch.randn(T, T, device=dev) * T ** -0.5
        on = torch.randn(Bn, T, Cn, device=dev)
        out = torch.empty_like(on)
        w_img = ops.x6_image(W, T, T, 0, vs=T, ks=1)
        on_img = ops.x6_image(on, Bn * Cn, T, 1, vg=Cn, vgs=T * Cn, vs=1, ks=Cn)
        fn = lambda: ops.x6_gemm(w_img, on_img, out, T, Bn * Cn, T, ldc=Cn, Nc=Cn, gsc=T * Cn)
    elif kind == "mha":
        qk = torch.randn(32, 3072, 224, device=dev)
        v = torch.randn(32, 3072, 112, device=dev)
        fn = lambda: ops.mha_fwd(qk, v)
    else:
        B, T, d, Cc = 32, 6912, 3, 48
        yq = torch.randn(B, T, d, device=dev, requires_grad=True)
        k = torch.randn(B, T, Cc, device=dev, requires_grad=True)
        v = torch.randn(B, T, Cc, device=dev, requires_grad=True)
        wq = torch.nn.Parameter(torch.randn(Cc, d, device=dev) * 0.1); bq = torch.nn.Parameter(torch.zeros(Cc, device=dev))
        dout = torch.randn(B, T, Cc, device=dev)
        fn = lambda: ops.SmallQKAttention.apply(yq, wq, bq, k, v, 0.1, True).backward(dout)
    fn()
    torch.cuda.synchronize()
    for _ in range(launches):
        fn()
    torch.cuda.synchronize()
    sys.exit(0)
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
