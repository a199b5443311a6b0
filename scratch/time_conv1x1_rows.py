"""1x1 convolutions of layer1 (forward with statistics accumulator, data gradient with residual): time and check against fp64.
   python scratch/time_conv1x1_rows.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops, _C
import torch.nn.functional as F
lib = _C.lib()
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (N, H, W, Ci, Co) in [(32, 96, 72, 64, 256), (32, 96, 72, 256, 64), (32, 96, 72, 64, 64), (32, 64, 48, 64, 256), (32, 64, 48, 256, 64), (5, 120, 110, 128, 128)]:
    x = torch.randn(N, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, 1, 1, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, H, W, Co, device=dev)
    res = torch.randn(N, H, W, Ci, device=dev)
    z, acc, info = ops.conv_fwd(x, w, None, 1, 0, stats="acc")
    dx = ops.conv_dgrad(dy, w, tuple(x.shape), 1, 0, residual=res)
    torch.cuda.synchronize()
    sl = slice(0, 2)
    zr = F.conv2d(x[sl].double().permute(0, 3, 1, 2).cpu(), w.double().cpu()).permute(0, 2, 3, 1)
    e1 = (z[sl].cpu().double() - zr).abs().max().item() / zr.abs().max().item()
    dxr = F.conv_transpose2d(dy[sl].double().permute(0, 3, 1, 2).cpu(), w.double().cpu()).permute(0, 2, 3, 1) + res[sl].double().cpu()
    e2 = (dx[sl].cpu().double() - dxr).abs().max().item() / dxr.abs().max().item()
    words = torch.empty(lib.buctd_bn_acc_bytes(Co) // 8, dtype=torch.int64, device=dev)
    import ctypes as C
    C.memmove  # (no-op: keep ctypes imported)
    hipacc = torch.frombuffer(bytearray(0), dtype=torch.int64) if False else None
    raw = torch.empty(0)
    # accumulator words -> sums
    buf = torch.empty(lib.buctd_bn_acc_bytes(Co) // 8, dtype=torch.int64, device=dev)
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(acc.ptr), buf.numel() * 8, 3)
    ws = buf.view(8, 4, Co).sum(0).cpu().double()
    s1 = ws[1] + ws[0] * 2.0 ** -48
    s2 = ws[3] + ws[2] * 2.0 ** -48
    zz = z.double().view(-1, Co)
    e3 = ((s1 - zz.sum(0).cpu()).abs().max() / zz.sum(0).abs().max().cpu()).item()
    e4 = ((s2 - (zz * zz).sum(0).cpu()).abs().max() / (zz * zz).sum(0).max().cpu()).item()
    tf = timeit(lambda: ops.conv_fwd(x, w, None, 1, 0, stats="acc"))
    tb = timeit(lambda: ops.conv_dgrad(dy, w, tuple(x.shape), 1, 0, residual=res))
    print(f"1x1 {Ci:3d}->{Co:3d} @{H}x{W} N={N}: fwd+stats {tf:6.1f} us (err {e1:.1e}, sums {e3:.1e} / {e4:.1e}), dgrad+res {tb:6.1f} us (err {e2:.1e})", flush=True)
    ops.acc_pool.reset(dev)
