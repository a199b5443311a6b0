import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
dev = torch.device("cuda:0")
def tm(fn, n=20):
    for _ in range(3): fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n * 1e3
for (N, H, W, C) in [(32, 96, 72, 48), (32, 48, 36, 96), (32, 24, 18, 192), (32, 12, 9, 384), (32, 96, 72, 256)]:
    z = torch.randn(N, H, W, C, device=dev); dy = torch.randn_like(z); res = torch.randn_like(z)
    g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
    part, info = ops.bn_stats(z)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd = ops.bn_finalize(part, info, z.numel() // C, C, 1e-5, 0.1, rm, rv)
    y = ops.bn_apply(z, mean, invstd, g, b, res, True)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    mb = z.numel() * 4 / 1e6
    t_fin = tm(lambda: ops.bn_finalize(part, info, z.numel() // C, C, 1e-5, 0.1, rm, rv))
    t_app = tm(lambda: ops.bn_apply(z, mean, invstd, g, b, None, True))
    t_appr = tm(lambda: ops.bn_apply(z, mean, invstd, g, b, res, True))
    t_bwd = tm(lambda: ops.bn_bwd(dy, None, z, mean, invstd, g, True, False, dg, db, 0, beta=b))
    t_bwdr = tm(lambda: ops.bn_bwd(dy, y, z, mean, invstd, g, True, True, dg, db, 0))
    print(f"{H}x{W} C{C} ({mb:.0f} MB): finalize {t_fin:.1f} us | apply {t_app:.1f} us ({2*mb/t_app:.0f} GB/s... {2*mb/t_app/1e3:.2f} TB/s) | apply+res {t_appr:.1f} ({3*mb/t_appr/1e3:.2f} TB/s) | bwd(no res) {t_bwd:.1f} us ({5*mb/t_bwd/1e3:.2f} TB/s) | bwd(res) {t_bwdr:.1f} us ({8*mb/t_bwdr/1e3:.2f} TB/s)")
