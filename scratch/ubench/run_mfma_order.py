import os, ctypes as C, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_order.so"))
dev = torch.device("cuda:0")
out = torch.empty(256 * 256, device=dev); cyc = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for mf in (1, 4):
    for order in (0, 1):
        for _ in range(2):
            assert lib.ub_order(order, mf, 400, 256, C.c_void_p(out.data_ptr()), C.c_void_p(cyc.data_ptr()), st) == 0
        torch.cuda.synchronize()
        print(f"MF={mf} order {order} ({'accumulator-major' if order else 'product-major'}): {cyc.float().mean().item() / (400 * 18 * mf):.2f} cycles per MFMA")
