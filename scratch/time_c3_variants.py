"""sustained solo timing of the variants of the bf16x6 3x3 forward / data-gradient launch a BasicBlock uses:
python scratch/time_c3_variants.py [H W C]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops, _C
lib = _C.lib()
dev = torch.device("cuda:0")
shapes = [(96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)]
if len(sys.argv) >= 4:
    shapes = [tuple(int(v) for v in sys.argv[1:4])]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for (H, W, Cn) in shapes:
    N = 32
    x = torch.randn(N, H, W, Cn, device=dev); dy = torch.randn(N, H, W, Cn, device=dev)
    z = torch.randn(N, H, W, Cn, device=dev); yb = torch.randn(N, H, W, Cn, device=dev)
    out = torch.empty_like(x)
    w = (torch.randn(Cn, Cn, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    wf, wb = ops._conv3x3_prepared(w, 0), ops._conv3x3_prepared(w, 1)
    ng, rpg = C.c_int(), C.c_int()
    lib.buctd_conv3x3_bf16x6_stats_groups(N, H, W, Cn, Cn, C.byref(ng), C.byref(rpg))
    part = torch.empty(ng.value * Cn * 2, device=dev); cnt = torch.empty(ng.value, dtype=torch.int32, device=dev)
    mean = torch.zeros(Cn, device=dev); inv = torch.ones(Cn, device=dev); ga = torch.ones(Cn, device=dev); be = torch.zeros(Cn, device=dev)
    st = ops.stream_ptr()
    cases = {
        "fwd+stats": lambda: lib.buctd_conv3x3_bf16x6(N, H, W, Cn, Cn, P(x), P(wf), None, None, None, None, 0, P(out), P(part), P(cnt), st),
        "fwd bnin+stats": lambda: lib.buctd_conv3x3_bf16x6_bnin(N, H, W, Cn, Cn, P(x), P(wf), None, None, None, None, 0, P(out), P(part), P(cnt), P(mean), P(inv), P(ga), P(be), 1, st),
        "dgrad": lambda: lib.buctd_conv3x3_bf16x6(N, H, W, Cn, Cn, P(dy), P(wb), None, None, None, None, 0, P(out), None, None, st),
        "dgrad+res": lambda: lib.buctd_conv3x3_bf16x6(N, H, W, Cn, Cn, P(dy), P(wb), None, None, None, P(x), 0, P(out), None, None, st),
        "dgrad bnstat(z)": lambda: lib.buctd_conv3x3_bf16x6_bnstat(N, H, W, Cn, Cn, P(dy), P(wb), None, P(out), P(z), None, P(mean), P(inv), P(ga), P(be), P(part), st),
        "dgrad+res bnstat(z,y)": lambda: lib.buctd_conv3x3_bf16x6_bnstat(N, H, W, Cn, Cn, P(dy), P(wb), P(x), P(out), P(z), P(yb), P(mean), P(inv), P(ga), None, P(part), st),
    }
    res = []
    for name, fn in cases.items():
        for _ in range(150): assert fn() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(300): fn()
        b.record(); b.synchronize()
        res.append(f"{name} {a.elapsed_time(b) / 300 * 1e3:.1f}")
    print(f"{H}x{W} C{Cn}: " + " | ".join(res), flush=True)
