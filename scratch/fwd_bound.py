"""Is the forward pass host-bound?  Forward-only wall time (train mode, no backward) at several batch sizes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import models, ops
dev = torch.device("cuda:0")
for B in (32, 16, 8, 2):
    cfg = bench.coam_w48_cfg(B)
    net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=True).to(dev).train()
    x, tgt, wt = bench.synthetic_batch(cfg, B, dev, 1)
    with torch.no_grad():
        for _ in range(3): net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): net(x)
        t_host = (time.perf_counter() - t0) / 10       # time to ENQUEUE (no sync yet)
        torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 10
    print(f"batch {B:2d}: forward {t_all*1e3:.1f} ms per pass (enqueue alone {t_host*1e3:.1f} ms)", flush=True)
    del net
