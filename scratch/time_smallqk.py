"""solo times of the three CoAM position-attention kernels at the C4 shape (B 32, T 6912, R4 4, C 48, p 0.1, bf16x6):
python scratch/time_smallqk.py [lib.so]   (an alternative library path is loaded by scratch/run_alt.py conventions)"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
from buctd_amd._C import lib, ptr, check, stream_ptr
dev = torch.device("cuda:0")
B, T, R4, C, p = int(os.environ.get("B", 32)), 6912, 4, 48, float(os.environ.get("P", 0.1))
g = torch.Generator().manual_seed(3)
qp = torch.randn(B, T, R4, generator=g).to(dev); kp = torch.randn(B, T, R4, generator=g).to(dev)
v = torch.randn(B, T, C, generator=g).to(dev); dout = torch.randn(B, T, C, generator=g).to(dev)
out = torch.empty_like(v); m = torch.empty(B, T, device=dev); linv = torch.empty(B, T, device=dev)
dqp = torch.empty_like(qp); dkp = torch.empty_like(kp); dv = torch.empty_like(v); dvec = torch.empty(B, T, device=dev)
scale = 1 / math.sqrt(C)
def fwd(): check(lib().buctd_attn_smallqk_fwd(B, T, R4, C, ptr(qp), ptr(kp), ptr(v), scale, p, 77, 2, ptr(out), ptr(m), ptr(linv), stream_ptr()), "f")
def bwd(): check(lib().buctd_attn_smallqk_bwd(B, T, R4, C, ptr(qp), ptr(kp), ptr(v), ptr(out), ptr(dout), ptr(m), ptr(linv), scale, p, 77, 2, ptr(dqp), ptr(dkp), ptr(dv), ptr(dvec), stream_ptr()), "b")
def tm(fn, n=10):
    for _ in range(3): fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n
print(f"fwd {tm(fwd)*1e3:.0f} us   bwd (q + kv) {tm(bwd)*1e3:.0f} us   checksum {out.double().sum().item():.6f} {dv.double().sum().item():.6f} {dqp.double().sum().item():.6f}")
