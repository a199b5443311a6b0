"""Feasibility: capture one whole train step (fwd + loss + bwd + Adam) in a HIP graph and replay it."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
ops.set_conv_math("bf16x3")
dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "32"))
cfg = bench.coam_w48_cfg(B)
net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, B, dev, 1)
crit = JointsMSELoss(True)
def step():
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward(); opt.step(); return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): l = step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager loss", l.item())
t0 = time.time()
for _ in range(5): step()
torch.cuda.synchronize(); print("eager ms/step", (time.time() - t0) / 5 * 1e3)
g = torch.cuda.CUDAGraph()
opt.zero_grad()
with torch.cuda.graph(g):
    lg = step()
torch.cuda.synchronize()
print("captured")
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print("graph ms/step", (time.time() - t0) / 10 * 1e3, "loss", lg.item())
