"""TransPose linear layers (tokens [32*3072, Cin] x [Cin, Cout]): 1x1-conv path vs the matmul kernel"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
def tm(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
B, T = 32, 3072
for Ci, Co in ((112, 224), (112, 112), (112, 192), (192, 112)):
    x = torch.randn(B, 1, T, Ci, device=dev)
    w = torch.randn(Co, Ci, device=dev) * Ci ** -0.5
    b = torch.randn(Co, device=dev)
    y = torch.empty(B * T, Co, device=dev)
    t1 = tm(lambda: ops.conv_fwd(x, w, b, 1, 0))
    t2 = tm(lambda: ops.matmul(x, w, y, batch=1, M=B * T, N=Co, K=Ci, a_layout=0, b_layout=0, lda=Ci, ldb=Ci, ldc=Co, bias=b, bias_axis=0))
    by = 4.0 * B * T * (Ci + Co)
    print(f"{Ci}->{Co}: conv path {t1:.1f} us, matmul {t2:.1f} us; HBM floor {by / 5e6:.1f} us at 5 TB/s, {2.0*B*T*Ci*Co/1e6/ t1:.0f} / {2.0*B*T*Ci*Co/1e6/t2:.0f} TFLOP/s")
