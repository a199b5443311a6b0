import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
from buctd_amd._C import lib, ptr, stream_ptr
dev = torch.device('cuda:0')
N, H, W, Ci, Co = 32, 96, 72, 48, 48
x = torch.randn(N, H, W, Ci, device=dev)
w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
y = torch.empty(N, H, W, Co, device=dev)
ng, rpg = C.c_int(), C.c_int()
lib().buctd_conv3x3_bf16x3_stats_groups(N, H, W, Ci, Co, C.byref(ng), C.byref(rpg))
part = torch.empty(ng.value, Co, 2, device=dev); cnt = torch.empty(ng.value, dtype=torch.int32, device=dev)
wp = ops._conv3x3_prepared(w, 0)
def launch():
    lib().buctd_conv3x3_bf16x3(N, H, W, Ci, Co, ptr(x), ptr(wp), None, None, None, None, 0, ptr(y), ptr(part), ptr(cnt), stream_ptr())
for _ in range(20): launch()
torch.cuda.synchronize()
# (a) one pair around 100 launches
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): launch()
e1.record(); torch.cuda.synchronize()
print("amortised: %.1f us/launch" % (e0.elapsed_time(e1) * 10))
# (b) a pair around every launch, CPU running ahead
pairs = []
for _ in range(100):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); launch(); b.record(); pairs.append((a, b))
torch.cuda.synchronize()
print("per-launch torch events: %.1f us" % (sum(a.elapsed_time(b) for a, b in pairs) * 10))
# (c) library dispatch events
lib().buctd_conv3x3_bf16x3_timing_begin(N, H, W, Ci, Co)
for _ in range(100): launch()
tot, n = C.c_double(), C.c_int()
lib().buctd_conv3x3_bf16x3_timing_end(C.byref(tot), C.byref(n))
print("dispatch events (hipExtLaunchKernel): %.1f us over %d" % (tot.value / max(n.value, 1), n.value))
