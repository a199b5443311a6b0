"""Time the bf16x3 3x3 conv (and the fp32 engine) on the HRNet-W48 branch shapes. usage: time_conv.py [fp32]"""
import sys, ctypes as C, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops, _C
from buctd_amd._C import lib, ptr, stream_ptr
dev = torch.device('cuda:0')
def run(N, H, W, Ci, Co, iters=50, fp32=False):
    x = torch.randn(N, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty(N, H, W, Co, device=dev)
    ng, rpg = C.c_int(), C.c_int()
    lib().buctd_conv3x3_bf16x3_stats_groups(N, H, W, Ci, Co, C.byref(ng), C.byref(rpg))
    part = torch.empty(ng.value, Co, 2, device=dev); cnt = torch.empty(ng.value, dtype=torch.int32, device=dev)
    wp = ops._conv3x3_prepared(w, 0)
    s = stream_ptr()
    fns = [("bf16x3", lambda: lib().buctd_conv3x3_bf16x3(N, H, W, Ci, Co, ptr(x), ptr(wp), None, None, None, None, 0, ptr(y), ptr(part), ptr(cnt), s))]
    if fp32:
        d = ops.conv_desc((N, H, W, Ci), (Co, Ci, 3, 3), 1, 1)
        ng2, rpg2 = C.c_int(), C.c_int()
        lib().buctd_conv2d_stats_groups(C.byref(d), 0, C.byref(ng2), C.byref(rpg2))
        part2 = torch.empty(ng2.value, Co, 2, device=dev)
        fns.append(("fp32", lambda: lib().buctd_conv2d_fwd(C.byref(d), ptr(x), ptr(w), None, None, None, None, 0, ptr(y), ptr(part2), s)))
    fl = 2.0 * N * H * W * Ci * Co * 9
    by = 4.0 * N * H * W * (Ci + Co) + 36.0 * Ci * Co
    out = f"N{N} {H}x{W} {Ci}->{Co}:"
    for name, fn in fns:
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out += f" {name} {us:.1f} us ({fl/us/1e6:.0f} TF/s-eq, {by/us/1e3:.0f} GB/s)"
    print(out)
shapes = [(32,96,72,48,48),(32,48,36,96,96),(32,24,18,192,192),(32,12,9,384,384),(32,96,72,64,64)]
if os.environ.get("SHAPES"):
    shapes = shapes[:int(os.environ["SHAPES"])]
for shp in shapes:
    run(*shp, fp32="fp32" in sys.argv)
