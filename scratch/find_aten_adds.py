"""which aten adds run in one CoAM-W48 train step (shapes + where autograd issues them): python scratch/find_aten_adds.py"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
WL = sys.argv[1] if len(sys.argv) > 1 else "train_c4"
mk, module = bench.TRAIN_WORKLOADS[WL][0], bench.TRAIN_WORKLOADS[WL][1]
cfg = mk(32)
net = getattr(models, module).get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, 32, dev, 1)
crit = JointsMSELoss(True)
def step():
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name.split("::")[1] in ("add", "add_", "mul", "mul_", "copy_", "clone", "contiguous", "sum", "cat", "zeros", "zero_", "fill_", "sub", "div", "to", "_to_copy", "index", "slice", "select", "empty_like", "zeros_like"):
        st = [s for s in (e.stack or []) if "buctd_amd" in s or "bench" in s]
        agg[(e.name, str(e.input_shapes)[:90], st[0][-70:] if st else "(autograd engine)")] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    if os.environ.get("ONLY") is None or k[0].split("::")[1] in os.environ["ONLY"].split(","): print(v, k)
