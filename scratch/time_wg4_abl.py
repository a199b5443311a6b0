"""times scratch/wg4_abl_<k>.so (scratch/wg4_abl.sh) on the roofline shape: python scratch/time_wg4_abl.py 0 1 2 3"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N, H, W, Cn = (int(v) for v in os.environ.get("SHAPE", "32,96,72,48").split(","))
x = torch.randn(N, H, W, Cn, device=dev)
dy = torch.randn(N, H, W, Cn, device=dev)
xp, dyp = ops.to_planes(x), ops.to_planes(dy)
gw = torch.empty(Cn, 3, 3, Cn, device=dev)
ws = torch.empty(256 * Cn * 9 * Cn * 4, dtype=torch.uint8, device=dev)
here = os.path.dirname(os.path.abspath(__file__))
for k in sys.argv[1:]:
    lib = C.CDLL(os.path.join(here, f"wg4_abl_{k}.so"))
    fn = lib.buctd_conv3x3_wgrad_bf16x6_p
    fn.argtypes = [C.c_int] * 5 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: fn(N, H, W, Cn, Cn, xp.buf.data_ptr(), dyp.buf.data_ptr(), gw.data_ptr(), 0, ws.data_ptr(), ws.numel(), st)
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); call(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"abl {k}: median {ts[15]:.1f} us  min {ts[0]:.1f} us (kernel + slab reduce)")
