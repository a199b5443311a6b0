"""Who issues the device-to-device copies in a train step? (torch profiler, grouped by python stack)"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
ops.set_conv_math("bf16x3")
dev = torch.device("cuda:0")
cfg = bench.coam_w48_cfg(8)
net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, 8, dev, 1)
crit = JointsMSELoss(True)
def step():
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter(); stacks = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::zeros", "aten::zero_", "aten::fill_", "aten::cat", "aten::mul", "aten::sum"):
        cnt[ev.name] += 1
        st = [s for s in ev.stack if "buctd_amd" in s or "bench.py" in s][:2]
        stacks[ev.name][" <- ".join(s.split("/")[-1] for s in st)] += 1
for k, v in cnt.most_common():
    print(k, v)
    for s, n in stacks[k].most_common(6):
        print("     ", n, s)
