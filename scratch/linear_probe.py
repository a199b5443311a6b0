import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
from buctd_amd._C import lib
dev = torch.device("cuda:0")
ops.set_conv_math("bf16x6")
B, T = 32, 3072
for Ci, Co in ((112, 336), (112, 112), (112, 192), (192, 112), (128, 384)):
    print(Ci, Co, "gconv supported:", lib().buctd_gconv_x6_supported(1, B, 1, T, Ci, Co, 0))
    x = torch.randn(B, 1, T, Ci, device=dev); x1 = x.view(1, B, T, Ci); w = (torch.randn(Co, Ci, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last); b = torch.randn(Co, device=dev)
    d = ops.conv_desc(x.shape, ops._wshape(w), 1, 0)
    print("   _gconv_ok:", ops._gconv_ok(d, 0))
    for name, fn in (("conv_fwd as B images of one row", lambda: ops.conv_fwd(x, w, b, 1, 0)), ("conv_fwd as one image of B rows", lambda: ops.conv_fwd(x1, w, b, 1, 0))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"   {name}: {e0.elapsed_time(e1) * 50:.1f} us")
