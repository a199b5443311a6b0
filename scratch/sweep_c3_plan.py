"""Sweep the (MF, WN) tile plan of the bf16x6 3x3 conv on the HRNet-W48 branch shapes (BUCTD_C3_FORCE)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import buctd_amd._C as _C_sel
if os.environ.get('BUCTD_TUNING_LIB', '1') == '1' and os.path.isfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libbuctd_hip_trace.so')):
    _C_sel.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libbuctd_hip_trace.so')   # experiment switches live in the tuning build
from buctd_amd import ops
from buctd_amd._C import lib, ptr, stream_ptr
dev = torch.device("cuda:0")
def run(N, H, W, Ci, Co, force, iters=40):
    if force: os.environ["BUCTD_C3_FORCE"] = force
    else: os.environ.pop("BUCTD_C3_FORCE", None)
    torch.manual_seed(0); x = torch.randn(N, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty(N, H, W, Co, device=dev)
    ng, rpg = C.c_int(), C.c_int()
    if lib().buctd_conv3x3_bf16x6_stats_groups(N, H, W, Ci, Co, C.byref(ng), C.byref(rpg)) != 0:
        return None
    part = torch.empty(ng.value, Co, 2, device=dev); cnt = torch.empty(ng.value, dtype=torch.int32, device=dev)
    wp = torch.empty(lib().buctd_conv3x3_bf16x6_prep_bytes(Ci, Co, 0), dtype=torch.uint8, device=dev)
    s = stream_ptr()
    lib().buctd_conv3x3_bf16x6_prep(Ci, Co, ptr(w), 0, ptr(wp), s)
    fn = lambda: lib().buctd_conv3x3_bf16x6(N, H, W, Ci, Co, ptr(x), ptr(wp), None, None, None, None, 0, ptr(y), ptr(part), ptr(cnt), s)
    for _ in range(5):
        if fn() != 0: return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters, y
shapes = [(32,96,72,48,48),(32,48,36,96,96),(32,24,18,192,192),(32,12,9,384,384)]
for shp in shapes:
    base = run(*shp, None)
    line = f"{shp}: default {base[0]:.1f}"
    for force in ("2,3,1","4,3,1","2,3,2","4,3,2","4,2,1","2,4,1","2,2,1","1,4,1"):
        r = run(*shp, force)
        if r is None: line += f" | {force}: n/a"; continue
        ok = torch.equal(r[1], base[1]) or (r[1]-base[1]).abs().max().item() < 1e-4
        line += f" | {force}: {r[0]:.1f}{'' if ok else ' WRONG'}"
    print(line, flush=True)
