"""torch.profiler (CPU side) of one train step at batch 2: where the host time of forward and backward goes.
python scratch/host_profile_torch.py [workload]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
WL = sys.argv[1] if len(sys.argv) > 1 else "train_c2"
mk, module = bench.TRAIN_WORKLOADS[WL][0], bench.TRAIN_WORKLOADS[WL][1]
cfg = mk(2)
net = getattr(models, module).get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, 2, dev, 1)
crit = JointsMSELoss(True)
def step():
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
