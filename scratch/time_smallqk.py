import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
B, T, d, C = 32, 6912, 3, 48
g = torch.Generator().manual_seed(0)
yq = torch.randn(B, T, d, generator=g).to(dev).requires_grad_(True)
k = torch.randn(B, T, C, generator=g).to(dev).requires_grad_(True)
v = torch.randn(B, T, C, generator=g).to(dev).requires_grad_(True)
wq = torch.nn.Parameter((torch.randn(C, d, generator=g) * 0.1).to(dev)); bq = torch.nn.Parameter(torch.zeros(C).to(dev))
dout = torch.randn(B, T, C, generator=g).to(dev)
ops.set_conv_math(os.environ.get("MATH", "bf16x3"))
for training in (True, False):
    for it in range(3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); o = ops.SmallQKAttention.apply(yq, wq, bq, k, v, 0.1, training); e[1].record(); o.backward(dout); e[2].record()
        torch.cuda.synchronize()
    print(f"dropout={'on' if training else 'off'}: fwd {e[0].elapsed_time(e[1]):.2f} ms, bwd {e[1].elapsed_time(e[2]):.2f} ms")
