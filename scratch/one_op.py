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
if kind in ("fwd_group", "wgrad_group"):
    # python scratch/one_op.py bf16x6 fwd_group|wgrad_group [launches]: the two-member group launches of the train step (branches 0
    # and 1 of HRNet-W48 at batch 32: 48 -> 48 @96x72 + 96 -> 96 @48x36), forward with the statistics accumulators / weight
    # gradients + slab reduction - the launches bench.py's `roofline` / `roofline_wgrad` describe
    from buctd_amd import _C
    lib = _C.lib()
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    ops.set_conv_math(mode)
    dev = torch.device("cuda:0")
    Nb = 32
    shapes = [(48, 96, 72), (96, 48, 36)]
    st = torch.cuda.current_stream().cuda_stream
    arr, wg, keep = (_C.C3Conv * 2)(), (_C.Wg3Conv * 2)(), []
    for d, it, (c_, h_, w_) in zip(arr, wg, shapes):
        wt = (torch.randn(c_, c_, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        x_, y_ = torch.randn(Nb, h_, w_, c_, device=dev), torch.randn(Nb, h_, w_, c_, device=dev)
        acc = torch.zeros(int(lib.buctd_bn_acc_bytes(c_)) // 8, dtype=torch.int64, device=dev)
        wp, dw = ops._conv3x3_prepared(wt, 0), torch.empty(c_, 3, 3, c_, device=dev)
        ws = torch.empty(int(lib.buctd_conv3x3_wgrad_bf16x6_group_workspace(2, Nb, h_, w_, c_, c_)), dtype=torch.uint8, device=dev)
        keep += [wt, x_, y_, acc, wp, dw, ws]
        d.N, d.H, d.W, d.Ci, d.Co = Nb, h_, w_, c_, c_
        d.x, d.wprep, d.y, d.stats_acc = x_.data_ptr(), wp.data_ptr(), y_.data_ptr(), acc.data_ptr()
        it.N, it.H, it.W, it.Ci, it.Co = Nb, h_, w_, c_, c_
        it.x, it.dy, it.dw, it.accumulate = x_.data_ptr(), y_.data_ptr(), dw.data_ptr(), 0
        it.workspace, it.workspace_bytes = ws.data_ptr(), ws.numel()
    if kind == "fwd_group":
        fn = lambda: _C.check(lib.buctd_conv3x3_bf16x6_group(2, arr, st), "group")
    else:
        fn = lambda: _C.check(lib.buctd_conv3x3_wgrad_bf16x6_group(2, wg, st), "wgrad group")
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
