import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
from buctd_amd.models.pose_hrnet_coam import PositionAttentionModule, ChannelAttentionModule
dev = torch.device("cuda:0")
ops.set_conv_math("bf16x3")
def tm(fn, n=5):
    for _ in range(2): fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n
for (H, W, C) in [(96, 72, 48), (48, 36, 96), (24, 18, 192)]:
    for cls in (ChannelAttentionModule, PositionAttentionModule):
        m = cls(d_model=C, d_cond=3, kernel_size=3, H=H, W=W, n_heads=1).to(dev).train()
        x = torch.randn(32, H, W, C, device=dev, requires_grad=True)
        cond = torch.randn(32, H, W, 3, device=dev)
        out = [None]
        def f(): out[0] = m(x, cond)
        tf = tm(f)
        g = torch.randn_like(out[0])
        def fb():
            m.zero_grad(set_to_none=True); x.grad = None
            y = m(x, cond); y.backward(g)
        tfb = tm(fb)
        print(f"{cls.__name__:24s} {H}x{W} C{C}: fwd {tf:6.2f} ms, fwd+bwd {tfb:6.2f} ms -> bwd {tfb - tf:6.2f} ms")
