import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
for rows, C in ((32 * 384 * 288, 3), (32 * 384 * 288, 64), (32 * 96 * 72, 48), (700001, 5)):
    x = torch.randn(rows, C, device=dev); out = torch.zeros(C, device=dev)
    for _ in range(3): ops.colsum(x, C, out, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.colsum(x, C, out, 0)
    e1.record(); torch.cuda.synchronize()
    err = (out.cpu().double() - x.double().sum(0).cpu()).abs().max().item() / max(1.0, x.double().sum(0).abs().max().item())
    print(f"colsum {rows} x {C}: {e0.elapsed_time(e1) * 100:.1f} us, rel err {err:.1e}")
