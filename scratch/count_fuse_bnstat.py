"""How many fuse-row backward launches of a C4 step carry BatchNorm sums, and which terms stay untagged."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
dev = torch.device("cuda:0")
cfg = bench.coam_w48_cfg(2)
net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net); opt = engine.get_optimizer(cfg, model)
x, t, w = bench.synthetic_batch(cfg, 2, dev, 1)
n = collections.Counter()
raw_a, raw_b, raw_f = ops.fuse_sum_bwd_bnstat, ops.fuse_sum_bwd, ops.FuseSum.forward
def a(dy, y, s, bns): n["bnstat launches"] += 1; n["bnstat terms"] += len(bns); return raw_a(dy, y, s, bns)
def b(dy, y, s): n["plain launches"] += 1; return raw_b(dy, y, s)
ops.fuse_sum_bwd_bnstat, ops.fuse_sum_bwd = a, b
raw_bn = ops.bn_bwd
def bn(*args, **kw): n["bn_bwd acc_ready" if kw.get("acc_ready") else "bn_bwd with reduction"] += 1; return raw_bn(*args, **kw)
ops.bn_bwd = bn
for _ in range(2):
    n.clear()
    loss = JointsMSELoss(True)(model(x), t, w); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
print(dict(n))
