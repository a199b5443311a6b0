"""fused vs materialised training attention, one TransPose-A6 encoder layer shape: python scratch/time_mha_train.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, d = 3072, 112
qk = torch.randn(B, T, 2 * d, device=dev, requires_grad=True)
v = torch.randn(B, T, d, device=dev, requires_grad=True)
dout = torch.randn(B, T, d, device=dev)
def run(fused):
    out = ops.FusedMHA.apply(qk, v, 0.1, True) if fused else ops.PositionAttention.apply(qk, None, v, 1, 0.1, True)
    out.backward(dout)
for fused in (True, False):
    for _ in range(2): run(fused)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): run(fused)
    b.record(); b.synchronize()
    ms = a.elapsed_time(b) / 5
    fl = 2.0 * B * T * T * d * (2 + 5)      # forward 2 products, backward 5 (S and dPd computed twice: 7 launches' worth)
    print(f"{'fused' if fused else 'materialised'}: fwd+bwd {ms:.2f} ms per layer (B={B}), peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
    torch.cuda.reset_peak_memory_stats()
