"""Run one bf16x6 conv shape back to back for a few seconds and sample the shader clock / power with rocm-smi;
also derive the clock from in-kernel cycle stamps (trace library) against HIP-event time."""
import os, sys, subprocess, time, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
from buctd_amd._C import lib, ptr, stream_ptr
N, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 48, 36, 96, 96)
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
def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    return " | ".join(l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l)
print("idle:", smi())
t_end = time.time() + 6
k = 0
while time.time() < t_end:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2000): fn()
    e1.record()
    if k % 2 == 0: print("busy:", smi())
    torch.cuda.synchronize()
    print(f"  2000 launches: {e0.elapsed_time(e1) / 2000 * 1e3:.1f} us each")
    k += 1
