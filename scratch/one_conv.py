"""10 launches of the roofline kernel (bf16x3 3x3 conv 48->48 @96x72, N=32, BN-stat epilogue) for rocprofv3 --pmc."""
import sys, ctypes as C, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops
from buctd_amd._C import lib, ptr, stream_ptr
dev = torch.device('cuda:0')
N, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 96, 72, 48, 48)
x = torch.randn(N, H, W, Ci, device=dev)
w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
y = torch.empty(N, H, W, Co, device=dev)
ng, rpg = C.c_int(), C.c_int()
lib().buctd_conv3x3_bf16x3_stats_groups(N, H, W, Ci, Co, C.byref(ng), C.byref(rpg))
part = torch.empty(ng.value, Co, 2, device=dev); cnt = torch.empty(ng.value, dtype=torch.int32, device=dev)
wp = ops._conv3x3_prepared(w, 0)
for _ in range(10):
    lib().buctd_conv3x3_bf16x3(N, H, W, Ci, Co, ptr(x), ptr(wp), None, None, None, None, 0, ptr(y), ptr(part), ptr(cnt), stream_ptr())
torch.cuda.synchronize()
