"""live device memory per train step, with and without gc.collect(): python scratch/leak_probe.py [workload]"""
import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
dev = torch.device("cuda:0")
WL = sys.argv[1] if len(sys.argv) > 1 else "train_c4"
B = int(os.environ.get("B", "8"))
cfg = bench.TRAIN_WORKLOADS[WL][0](B)
net = getattr(models, bench.TRAIN_WORKLOADS[WL][1]).get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net); opt = engine.get_optimizer(cfg, model); model.flatten()
x, tgt, wt = bench.synthetic_batch(cfg, B, dev, 1)
crit = JointsMSELoss(True)
def step():
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward(); opt.step()
def live(): torch.cuda.synchronize(); return torch.cuda.memory_allocated() / 2**20
for i in range(8):
    step(); a = live()
    n = gc.collect(); b = live()
    print(f"step {i}: live {a:9.1f} MB, after gc.collect() ({n} objects) {b:9.1f} MB", flush=True)
gc.disable()
for i in range(4):
    step(); print(f"gc disabled, step {i}: live {live():9.1f} MB", flush=True)
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
import collections
c = collections.Counter(type(o).__name__ for o in gc.garbage)
print("garbage types:", c.most_common(12))
ts = [o for o in gc.garbage if torch.is_tensor(o) and o.is_cuda]
print("cuda tensors in garbage:", len(ts), sum(t.numel() * 4 for t in ts) / 2**20, "MB")
fn = [o for o in gc.garbage if "Backward" in type(o).__name__ or "Ctx" in type(o).__name__][:10]
print([type(o).__name__ for o in fn])
