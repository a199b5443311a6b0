import os, ctypes as C, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_valu2.so"))
dev = torch.device("cuda:0")
out = torch.empty(256 * 512, device=dev); cyc = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(big, V, mode, wps, iters=500, blocks=256):
    cyc.zero_()
    for _ in range(2):
        rc = lib.ub_run(big, V, mode, wps, iters, blocks, C.c_void_p(out.data_ptr()), C.c_void_p(cyc.data_ptr()), st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    c = cyc.cpu().numpy().reshape(blocks, 8)[:, :4 * wps]
    if mode == 3:
        return "mfma-wave %.1f / valu-wave %.1f" % (c[:, :4].mean() / (iters * 16), c[:, 4:].mean() / (iters * 16))
    return "%.1f" % (c.mean() / (iters * 16))
for big in (0, 1):
    print("==== MFMA", "32x32x16 (32 cycles at peak)" if big else "16x16x32 (16 cycles at peak)")
    for wps in (1, 2):
        print(f"-- {wps} wave(s) per SIMD")
        print(f"   MFMA only: {run(big, 0, 0, wps)}")
        for V in (1, 2, 3, 4, 6, 8, 12, 16):
            print(f"   V={V}: VALU only {run(big, V, 1, wps)} | same wave MFMA+VALU {run(big, V, 2, wps)}" +
                  (f" | split waves (one MFMA, one VALU) {run(big, V, 3, wps)}" if wps == 2 else ""))
