"""fresh box, first process: is it the GPU or the host that is slow, and for how long?  Every half second for a minute:
GPU time of a fixed GEMM batch (events), wall time of 300 tiny launches (launch path), wall time of a fixed pure-Python loop.
python scratch/first_run_probe2.py"""
import time, torch
t00 = time.time()
dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
big = torch.empty(1 << 28, device=dev)           # 1 GB stream
small = torch.zeros(64, device=dev)
torch.cuda.synchronize()
print(f"init {time.time() - t00:.2f} s", flush=True)
def sample():
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(10): c = a @ b
    e1.record()
    for _ in range(4): big.add_(1.0)
    e2.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): small.add_(1.0)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_sync = time.perf_counter() - t0
    t0 = time.perf_counter(); s = 0
    for i in range(200000): s += i * i
    t_py = time.perf_counter() - t0
    return e0.elapsed_time(e1), e1.elapsed_time(e2), t_enq * 1e3, t_sync * 1e3, t_py * 1e3
end = time.time() + float(__import__("os").environ.get("SECS", "45"))
while time.time() < end:
    g, m, q, sy, py = sample()
    print(f"t={time.time() - t00:6.1f}s  gemm x10 {g:7.2f} ms  1GB add x4 {m:6.2f} ms  300 launches: enqueue {q:6.2f} ms, done {sy:6.2f} ms  python loop {py:6.2f} ms", flush=True)
    time.sleep(0.4)
