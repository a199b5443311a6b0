import sys, ctypes as C, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops, _C
from buctd_amd._C import lib, ptr, stream_ptr
dev = torch.device('cuda:0')
def run(N, H, W, Ci, Co, flip=0, iters=50):
    x = torch.randn(N, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty(N, H, W, Co, device=dev)
    ng, rpg = C.c_int(), C.c_int()
    lib().buctd_conv3x3_bf16x3_stats_groups(N, H, W, Ci, Co, C.byref(ng), C.byref(rpg))
    part = torch.empty(ng.value, Co, 2, device=dev); cnt = torch.empty(ng.value, dtype=torch.int32, device=dev)
    d = ops.conv_desc((N, H, W, Ci), (Co, Ci, 3, 3), 1, 1)
    ng2, rpg2 = C.c_int(), C.c_int()
    lib().buctd_conv2d_stats_groups(C.byref(d), 0, C.byref(ng2), C.byref(rpg2))
    part2 = torch.empty(ng2.value, Co, 2, device=dev)
    s = stream_ptr()
    res = {}
    for name, fn in (("bf16x3", lambda: lib().buctd_conv3x3_bf16x3(N, H, W, Ci, Co, ptr(x), ptr(w), flip, None, None, None, None, 0, ptr(y), ptr(part), ptr(cnt), s)),
                     ("fp32", lambda: lib().buctd_conv2d_fwd(C.byref(d), ptr(x), ptr(w), None, None, None, None, 0, ptr(y), ptr(part2), s))):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * N * H * W * Ci * Co * 9
    by = 4.0 * N * H * W * (Ci + Co) + 36.0 * Ci * Co
    print(f"N{N} {H}x{W} {Ci}->{Co}: bf16x3 {res['bf16x3']:.1f} us ({fl/res['bf16x3']/1e6:.0f} TF/s-eq, {by/res['bf16x3']/1e3:.0f} GB/s) | fp32 {res['fp32']:.1f} us ({fl/res['fp32']/1e6:.0f} TF/s)")
for shp in [(32,96,72,48,48),(32,48,36,96,96),(32,24,18,192,192),(32,12,9,384,384),(32,96,72,64,64)]:
    run(*shp)
