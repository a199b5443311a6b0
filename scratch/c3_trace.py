"""Cycle stamps (s_memtime) of wave 0 of eight workgroups of the bf16x6 3x3 forward kernel.
usage: BUCTD_LIB_TRACE=1 python scratch/c3_trace.py N H W Ci Co   (needs scratch/build_trace_lib.sh)"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import buctd_amd._C as _C
_C.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbuctd_hip_trace.so")
from buctd_amd import ops
from buctd_amd._C import lib, ptr, stream_ptr
N, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 96, 72, 48, 48)
dev = torch.device("cuda:0")
x = torch.randn(N, H, W, Ci, device=dev)
w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
y = torch.empty(N, H, W, Co, device=dev)
ng, rpg = C.c_int(), C.c_int()
lib().buctd_conv3x3_bf16x6_stats_groups(N, H, W, Ci, Co, C.byref(ng), C.byref(rpg))
part = torch.empty(ng.value, Co, 2, device=dev); cnt = torch.empty(ng.value, dtype=torch.int32, device=dev)
wp = torch.empty(lib().buctd_conv3x3_bf16x6_prep_bytes(Ci, Co, 0), dtype=torch.uint8, device=dev)
s = stream_ptr()
lib().buctd_conv3x3_bf16x6_prep(Ci, Co, ptr(w), 0, ptr(wp), s)
fn = lambda: lib().buctd_conv3x3_bf16x6(N, H, W, Ci, Co, ptr(x), ptr(wp), None, None, None, None, 0, ptr(y), ptr(part), ptr(cnt), s)
for _ in range(int(os.environ.get("WARM", "5"))): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print(f"shape N{N} {H}x{W} {Ci}->{Co}: one launch {e0.elapsed_time(e1)*1e3:.1f} us")
buf = np.zeros((8, 64), dtype=np.uint64)
raw = C.CDLL(_C.LIB_PATH)
raw.buctd_debug_c3_trace(buf.ctypes.data_as(C.c_void_p))
nch = Ci // 16
t00 = min(int(buf[s_, 0]) for s_ in range(8) if buf[s_, 0])
for s_ in range(8):
    b = buf[s_].astype(np.int64)
    if b[0] == 0: continue
    rel = lambda k: int(b[k] - b[0])
    chunks = [(rel(2 + 2 * c), rel(3 + 2 * c)) for c in range(min(nch, 24))]
    steps = [int(b[50 + k] - b[50]) for k in range(5)] if nch > 1 else []
    wall = int(b[63] - b[62])
    print(f"   core clock = {rel(61)} cycles / {wall} ticks of 100 MHz = {rel(61) / max(wall, 1) * 100:.0f} MHz")
    print(f"   step-2 detail (rel. to step 2 start): store_a begins {int(b[56]-b[52])}, ends {int(b[57]-b[52])}, B+A loads issued {int(b[58]-b[52])}, step 3 starts {int(b[53]-b[52])}")
    print(f"wg slot {s_}: start +{int(b[0]) - t00}, prologue {rel(1)}, chunks(start,end) {chunks[:4]}{'...' if nch > 4 else ''} last {chunks[-1]}, "
          f"epi start {rel(60)}, end {rel(61)}; chunk-1 steps at {steps}")

wall = np.zeros((4096, 2), dtype=np.uint64)
raw.buctd_debug_c3_wall(wall.ctypes.data_as(C.c_void_p))
wl = wall[wall[:, 0] > 0].astype(np.int64)
t0 = wl[:, 0].min()
st, en = (wl[:, 0] - t0) / 100.0, (wl[:, 1] - t0) / 100.0      # us
print(f"{len(wl)} workgroups; start times us: min {st.min():.2f} p25 {np.percentile(st,25):.2f} p50 {np.percentile(st,50):.2f} "
      f"p75 {np.percentile(st,75):.2f} max {st.max():.2f}; end: min {en.min():.2f} p50 {np.percentile(en,50):.2f} max {en.max():.2f}; "
      f"lifetime mean {np.mean(en-st):.2f} us")
order = np.argsort(st)
print("start time by dispatch index (every 32nd):", [f"{i}:{st[i]:.1f}" for i in range(0, len(st), 32)])
life = en - st
hist, edges = np.histogram(life, bins=12)
print("lifetime histogram:", [f"{edges[i]:.1f}-{edges[i+1]:.1f}:{hist[i]}" for i in range(len(hist))])
idx = np.nonzero(wall[:, 0] > 0)[0]
for x in range(8):
    m = idx % 8 == x
    print(f"  xcd {x}: n {m.sum()} lifetime mean {life[m].mean():.1f} max {life[m].max():.1f} end max {en[m].max():.1f}")
slow = np.argsort(-life)[:24]
print("slowest dispatch indices:", sorted(int(idx[i]) for i in slow))
fast = np.argsort(life)[:24]
print("fastest dispatch indices:", sorted(int(idx[i]) for i in fast))
