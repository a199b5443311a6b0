"""The 3x3 weight gradients of the HRNet branches: one launch pair (kernel + slab reduction) per branch / one GROUP launch
pair for all (buctd_conv3x3_wgrad_bf16x6_group).   python scratch/time_group_wgrad.py [w48|w32] [N] [sets ...]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops, _C
lib = _C.lib()
dev = torch.device("cuda:0")
fam = sys.argv[1] if len(sys.argv) > 1 else "w48"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
shapes = {"w48": [(96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)],
          "w32": [(64, 48, 32), (32, 24, 64), (16, 12, 128), (8, 6, 256)]}[fam]
ctx = []
for (H, W, Cn) in shapes:
    x = torch.randn(N, H, W, Cn, device=dev)
    dy = torch.randn(N, H, W, Cn, device=dev)
    dw = torch.empty(Cn, 3, 3, Cn, device=dev)
    ws1 = torch.empty(lib.buctd_conv3x3_wgrad_bf16x6_workspace(N, H, W, Cn, Cn), dtype=torch.uint8, device=dev)
    ctx.append(dict(H=H, W=W, C=Cn, x=x, dy=dy, dw=dw, ws1=ws1, dw2=torch.empty_like(dw)))
main = torch.cuda.current_stream()


def single(c):
    lib.buctd_conv3x3_wgrad_bf16x6(N, c["H"], c["W"], c["C"], c["C"], c["x"].data_ptr(), c["dy"].data_ptr(), c["dw"].data_ptr(), 0,
                                   c["ws1"].data_ptr(), c["ws1"].numel(), main.cuda_stream)


def group(which, wss):
    arr = (_C.Wg3Conv * len(which))()
    for k, i in enumerate(which):
        c, it = ctx[i], arr[k]
        it.N, it.H, it.W, it.Ci, it.Co = N, c["H"], c["W"], c["C"], c["C"]
        it.x, it.dy, it.dw, it.accumulate = c["x"].data_ptr(), c["dy"].data_ptr(), c["dw2"].data_ptr(), 0
        it.workspace, it.workspace_bytes = wss[k].data_ptr(), wss[k].numel()
    _C.check(lib.buctd_conv3x3_wgrad_bf16x6_group(len(which), arr, main.cuda_stream), "group")


def run(mode, which, wss, reps=30):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(reps):
        if mode == "serial":
            for i in which:
                single(ctx[i])
        else:
            group(which, wss)
    b.record(main)
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


sets = [[int(c) for c in a] for a in sys.argv[3:]] or [[0, 1, 2, 3], [0, 1, 2], [0, 1], [0], [1], [2], [3]]
for which in sets:
    wss = []
    for i in which:
        c = ctx[i]
        wss.append(torch.empty(lib.buctd_conv3x3_wgrad_bf16x6_group_workspace(len(which), N, c["H"], c["W"], c["C"], c["C"]),
                               dtype=torch.uint8, device=dev))
    for _ in range(2):
        run("serial", which, wss); run("group", which, wss)
    s, g = run("serial", which, wss), run("group", which, wss)
    err = max(((ctx[i]["dw"] - ctx[i]["dw2"]).abs().max() / ctx[i]["dw"].abs().max()).item() for i in which)
    print(f"branches {which}: serial {s:.1f} us, group {g:.1f} us ({100 * (1 - g / s):.0f} % saved), max rel diff {err:.1e}", flush=True)
