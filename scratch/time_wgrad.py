"""Time the bf16x3 3x3 weight-gradient kernel on the HRNet-W48 branch shapes."""
import sys, ctypes as C, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
from buctd_amd._C import lib, ptr, stream_ptr
dev = torch.device('cuda:0')
def run(N, H, W, Ci, Co, iters=30):
    x = torch.randn(N, H, W, Ci, device=dev); dy = torch.randn(N, H, W, Co, device=dev)
    dw = torch.empty(Co, 3, 3, Ci, device=dev)
    need = lib().buctd_conv3x3_wgrad_bf16x3_workspace(N, H, W, Ci, Co)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    s = stream_ptr()
    fn = lambda: lib().buctd_conv3x3_wgrad_bf16x3(N, H, W, Ci, Co, ptr(x), ptr(dy), ptr(dw), 0, ptr(ws), need, s)
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * N * H * W * Ci * Co * 9
    print(f"wgrad N{N} {H}x{W} {Ci}->{Co}: {us:.1f} us incl. reduce ({fl/us/1e6:.0f} TF/s-eq)")
shapes = [(32,96,72,48,48),(32,48,36,96,96),(32,24,18,192,192),(32,12,9,384,384)]
for shp in shapes[:int(os.environ.get("SHAPES", "4"))]:
    run(*shp)
