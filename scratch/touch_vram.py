"""touch GB gigabytes of device memory once (fresh-box experiments): python scratch/touch_vram.py [GB]"""
import sys, time, torch
gb = int(sys.argv[1]) if len(sys.argv) > 1 else 160
t0 = time.time()
bufs = []
for i in range(gb):
    bufs.append(torch.empty(1 << 28, dtype=torch.float32, device="cuda:0").fill_(1.0))
torch.cuda.synchronize()
print(f"touched {gb} GB in {time.time() - t0:.1f} s")
