"""preNet weight gradients: thin kernel vs the implicit-GEMM kernel (BUCTD_WGRAD_THIN=0) at the C2 / C3 input sizes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import buctd_amd._C as _C_sel
if os.environ.get('BUCTD_TUNING_LIB', '1') == '1' and os.path.isfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libbuctd_hip_trace.so')):
    _C_sel.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libbuctd_hip_trace.so')   # experiment switches live in the tuning build
from buctd_amd import ops
dev = torch.device("cuda:0")
def tm(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (H, W) in ((384, 288), (256, 192)):
    for Ci, Co, k in ((3, 64, 3), (64, 3, 7), (3, 3, 7)):
        x = torch.randn(32, H, W, Ci, device=dev); dy = torch.randn(32, H, W, Co, device=dev)
        w = torch.zeros(Co, Ci, k, k, device=dev).contiguous(memory_format=torch.channels_last); out = torch.empty_like(w)
        t = tm(lambda: ops.conv_wgrad(x, dy, w, 1, k // 2, out=out, accumulate=0))
        print(f"{H}x{W} {Ci}->{Co} k{k}: {t:.0f} us", flush=True)
