import os, ctypes as C, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_mem.so"))
dev = torch.device("cuda:0")
out = torch.empty(256 * 256, device=dev); cyc = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
g = torch.ones(4 * 65536 * 4, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(kind, L, inter, iters=400):
    for _ in range(2):
        rc = lib.ub_mem(kind, L, inter, iters, 256, C.c_void_p(g.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(cyc.data_ptr()), st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    return cyc.float().mean().item() / iters
print(f"16 MFMAs alone: {run(0, 0, 0):.0f} cycles per iteration (one wave per SIMD, 256 CUs)")
for kind, name in ((1, "ds_read_b128"), (2, "global_load_dwordx4 (L2-resident)")):
    for L in (4, 8, 12):
        print(f"  + {L:2d} {name}: bunched in front {run(kind, L, 0):.0f} | interleaved one per MFMA {run(kind, L, 1):.0f}")
