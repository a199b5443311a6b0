"""times scratch/mha_abl_<k>.so (scratch/mha_abl.sh), sustained: python scratch/time_mha_abl.py 0 1 2 ..."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import ops  # noqa
dev = torch.device("cuda:0")
B, T, d = 32, 3072, 112
qk = torch.randn(B, T, 2 * d, device=dev); v = torch.randn(B, T, d, device=dev)
out = torch.empty(B, T, d, device=dev)
ws = torch.empty(ops.lib().buctd_mha_fwd_bf16x6_workspace(B, T, d), dtype=torch.uint8, device=dev)
here = os.path.dirname(os.path.abspath(__file__))
for k in sys.argv[1:]:
    lib = C.CDLL(os.path.join(here, f"mha_abl_{k}.so"))
    fn = lib.abl_mha
    fn.argtypes = [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: fn(B, T, d, qk.data_ptr(), qk.data_ptr() + 4 * d, v.data_ptr(), 2 * d, d, 0.0945, out.data_ptr(), ws.data_ptr(), ws.numel(), st)
    for _ in range(300):
        assert call() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300): call()
    b.record(); b.synchronize()
    print(f"abl {k}: {a.elapsed_time(b) / 300 * 1e3:.1f} us per call (split kernel + attention), sustained")
