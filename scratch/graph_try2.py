import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
ops.set_conv_math("bf16x3")
dev = torch.device("cuda:0")
B = 4
cfg = bench.coam_w48_cfg(B)
net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, B, dev, 1)
crit = JointsMSELoss(True)
mode = sys.argv[1]
def step():
    if mode == "fwd":
        with torch.no_grad():
            return crit(model(x), tgt, wt)
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward()
    if mode == "full": opt.step()
    return loss
for _ in range(3): l = step()
torch.cuda.synchronize(); print("eager ok", l.item()); sys.stdout.flush()
g = torch.cuda.CUDAGraph()
opt.zero_grad()
print("begin capture"); sys.stdout.flush()
with torch.cuda.graph(g):
    print(" in capture"); sys.stdout.flush()
    lg = step()
    print(" step enqueued"); sys.stdout.flush()
print("captured"); sys.stdout.flush()
g.replay(); torch.cuda.synchronize(); print("replayed", lg.item())
