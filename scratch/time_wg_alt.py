"""solo time of the bf16x6 3x3 weight gradient for several library builds: python scratch/time_wg_alt.py lib1.so lib2.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from buctd_amd import _C
_C.LIB_PATH = sys.argv[1]
from buctd_amd import ops
dev = torch.device("cuda:0")
out = []
for (H, W, Cn) in ((96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)):
    N = 32
    x = torch.randn(N, H, W, Cn, device=dev); dy = torch.randn(N, H, W, Cn, device=dev)
    w = (torch.randn(Cn, Cn, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    gw = torch.empty_like(w)
    fn = lambda: ops.conv_wgrad(x, dy, w, 1, 1, out=gw, accumulate=0)
    for _ in range(150): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300): fn()
    b.record(); b.synchronize()
    out.append("C%%d %%.1f" %% (Cn, a.elapsed_time(b) / 300 * 1e3))
print(os.path.basename(sys.argv[1]), " | ".join(out))
''' % ROOT
for lib in sys.argv[1:]:
    if not os.path.isabs(lib): lib = os.path.join(ROOT, lib)
    subprocess.run([sys.executable, "-c", code, lib])
