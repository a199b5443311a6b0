"""1x1 / stride-2 3x3 weight gradients: the bf16x6 kernel (conv_gather_wgrad.hip) against the exact-fp32 MFMA kernel - error
against fp64 at small shapes and time at the CoAM-W48 shapes.   python scratch/time_gconv_wgrad.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")


def run(N, H, W, Ci, Co, k, reps=20, check=False):
    stride, pad = (1, 0) if k == 1 else (2, 1)
    Ho, Wo = (H, W) if k == 1 else (H // 2, W // 2)
    x = torch.randn(N, H, W, Ci, device=dev)
    dy = torch.randn(N, Ho, Wo, Co, device=dev)
    w = torch.empty(Co, Ci, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for on in (True, False):
        ops._GCONV_WGRAD["on"] = on
        for _ in range(3):
            out = ops.conv_wgrad(x, dy, w, stride, pad)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            out = ops.conv_wgrad(x, dy, w, stride, pad)
        b.record(); b.synchronize()
        res[on] = (a.elapsed_time(b) / reps * 1e3, out.clone())
    ops._GCONV_WGRAD["on"] = True
    msg = f"{k}x{k} {Ci:3d}->{Co:3d} @{H}x{W} N={N}: bf16x6 {res[True][0]:7.1f} us, fp32 MFMA {res[False][0]:7.1f} us"
    if check:
        ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2).cpu(), (Co, Ci, k, k), dy.double().permute(0, 3, 1, 2).cpu(),
                                          stride=stride, padding=pad)
        sc = ref.abs().max().item()
        e1 = (res[True][1].cpu().double() - ref).abs().max().item() / sc
        e0 = (res[False][1].cpu().double() - ref).abs().max().item() / sc
        msg += f"; max err / max |dw|: bf16x6 {e1:.1e}, fp32 {e0:.1e}"
    print(msg, flush=True)


for (N, H, W, Ci, Co, k) in [(2, 12, 10, 48, 96, 1), (3, 8, 6, 64, 256, 1), (2, 12, 10, 48, 48, 3), (3, 8, 6, 96, 192, 3), (2, 12, 8, 64, 64, 3),
                             (2, 10, 6, 256, 96, 3), (1, 6, 4, 32, 64, 1)]:
    run(N, H, W, Ci, Co, k, reps=2, check=True)
for (H, W, Ci, Co, k) in [(96, 72, 64, 256, 1), (96, 72, 256, 64, 1), (96, 72, 64, 64, 1), (48, 36, 96, 48, 1), (24, 18, 192, 48, 1), (12, 9, 384, 48, 1),
                          (24, 18, 192, 96, 1), (12, 9, 384, 192, 1), (96, 72, 48, 96, 3), (96, 72, 48, 48, 3), (48, 36, 96, 192, 3),
                          (48, 36, 48, 48, 3), (24, 18, 192, 384, 3), (24, 18, 96, 384, 3), (96, 72, 256, 96, 3), (192, 144, 64, 64, 3)]:
    run(32, H, W, Ci, Co, k)
