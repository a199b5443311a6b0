import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
ops.set_conv_math("bf16x3")
dev = torch.device("cuda:0")
cfg = bench.coam_w48_cfg(4)
net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, 4, dev, 1)
crit = JointsMSELoss(True)
def step():
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward(); opt.step(); return loss
print("default stream:", step().item()); sys.stdout.flush()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    l = step()
    print("enqueued on side stream"); sys.stdout.flush()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("side stream:", l.item())
