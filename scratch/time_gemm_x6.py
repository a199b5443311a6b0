"""fc_o GEMMs of CoAM-W48 (T = 6912, N = 32 x 48) in bf16x6: image preparation and product timings vs the fp32 kernel"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import buctd_amd._C as _C_sel
if os.environ.get('BUCTD_TUNING_LIB', '1') == '1' and os.path.isfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libbuctd_hip_trace.so')):
    _C_sel.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libbuctd_hip_trace.so')   # experiment switches live in the tuning build
from buctd_amd import ops
dev = torch.device("cuda:0")
Bn, T, Cn = 32, int(os.environ.get("T", "6912")), 48
W = torch.randn(T, T, device=dev) * T ** -0.5
on = torch.randn(Bn, T, Cn, device=dev)
out = torch.empty_like(on)
def tm(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
w_img = ops.x6_image(W, T, T, 0, vs=T, ks=1)
on_img = ops.x6_image(on, Bn * Cn, T, 1, vg=Cn, vgs=T * Cn, vs=1, ks=Cn)
fl = 2.0 * T * T * Bn * Cn
t = tm(lambda: ops.x6_image(W, T, T, 0, vs=T, ks=1, out=w_img)); print(f"image of W (row-major): {t:.0f} us")
wt_img = ops.x6_image(W, T, T, 0, vs=1, ks=T)
t = tm(lambda: ops.x6_image(W, T, T, 0, vs=1, ks=T, out=wt_img)); print(f"image of W^T: {t:.0f} us")
t = tm(lambda: ops.x6_image(on, Bn * Cn, T, 1, vg=Cn, vgs=T * Cn, vs=1, ks=Cn, out=on_img)); print(f"image of on (B operand): {t:.0f} us")
t = tm(lambda: ops.x6_gemm(w_img, on_img, out, T, Bn * Cn, T, ldc=Cn, Nc=Cn, gsc=T * Cn)); print(f"fwd product: {t:.0f} us = {fl/t/1e6:.0f} TFLOP/s-eq")
a_img = ops.x6_image(out, T, Bn * Cn, 0, vs=Cn, ks=1, kg=Cn, kgs=T * Cn)
b_img = ops.x6_image(on, T, Bn * Cn, 1, vs=Cn, ks=1, kg=Cn, kgs=T * Cn)
t = tm(lambda: ops.x6_image(out, T, Bn * Cn, 0, vs=Cn, ks=1, kg=Cn, kgs=T * Cn, out=a_img)); print(f"image of dout (A operand, wgrad): {t:.0f} us")
dW = torch.empty(T, T, device=dev)
t = tm(lambda: ops.x6_gemm(a_img, b_img, dW, T, T, Bn * Cn, ldc=T)); print(f"wgrad product: {t:.0f} us = {fl/t/1e6:.0f} TFLOP/s-eq")
t = tm(lambda: ops.matmul(W, on, out, batch=1, M=T, N=Bn * Cn, K=T, a_layout=0, b_layout=1, lda=T, ldb=Cn, ldc=Cn, Nc=Cn, gsbn=T * Cn, gsc=T * Cn))
print(f"fp32 kernel fwd: {t:.0f} us = {fl/t/1e6:.0f} TFLOP/s")
