"""Host enqueue time vs wall time of one TransPose-H-A6 eval forward (C5) and one preNet-W32 train step (C2)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import models, engine
from buctd_amd.core.loss import JointsMSELoss
dev = torch.device("cuda:0")
cfg = bench.transpose_a6_cfg(32)
net = models.transpose_h.get_pose_net(cfg, is_train=False).to(dev).eval()
x = torch.randn(32, 6, 256, 192, device=dev)
with torch.no_grad():
    for _ in range(3): net(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net(x)
    th = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / 10
print(f"C5 eval forward: wall {ta*1e3:.1f} ms, host enqueue {th*1e3:.1f} ms")
del net
cfg = bench.prenet_cfg(32, 32, (192, 256))
net = models.pose_hrnet.get_pose_net(cfg, is_train=True).to(dev)
model = engine.DataParallel(net); opt = engine.get_optimizer(cfg, model); model.flatten()
xb, tgt, wt = bench.synthetic_batch(cfg, 32, dev, 1)
crit = JointsMSELoss(True)
def step():
    loss = crit(model(xb), tgt, wt); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
th = (time.perf_counter() - t0) / 10
torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / 10
print(f"C2 train step: wall {ta*1e3:.1f} ms, host enqueue {th*1e3:.1f} ms")
