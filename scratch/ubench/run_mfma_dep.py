import os, ctypes as C, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_dep.so"))
dev = torch.device("cuda:0")
out = torch.empty(256 * 256, device=dev); cyc = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for nacc in (1, 2, 3, 4, 6):
    for _ in range(2):
        assert lib.ub_dep(nacc, 500, 256, C.c_void_p(out.data_ptr()), C.c_void_p(cyc.data_ptr()), st) == 0
    torch.cuda.synchronize()
    print(f"accumulator reused every {nacc} MFMA(s): {cyc.float().mean().item() / (500 * 24):.1f} cycles per MFMA (one wave per SIMD)")
