"""Host time of the Python backward functions of one train step (they run on autograd's device thread, which cProfile does
not see): perf_counter around every autograd.Function.backward of buctd_amd.ops, batch 2 (GPU work negligible).
   python scratch/host_backward_split.py [train_c2|train_c4]"""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
dev = torch.device("cuda:0")
WL = sys.argv[1] if len(sys.argv) > 1 else "train_c2"
mk, module = bench.TRAIN_WORKLOADS[WL][0], bench.TRAIN_WORKLOADS[WL][1]
cfg = mk(2)
net = getattr(models, module).get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, 2, dev, 1)
crit = JointsMSELoss(True)
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
on = {"v": False}
for name in dir(ops):
    cls = getattr(ops, name)
    if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
        for which in ("forward", "backward"):
            raw = getattr(cls, which)
            def wrap(raw=raw, key=(name, which)):
                def f(*a, **k):
                    t0 = time.perf_counter()
                    r = raw(*a, **k)
                    if on["v"]:
                        e = acc[key]; e[0] += 1; e[1] += time.perf_counter() - t0
                    return r
                return staticmethod(f)
            setattr(cls, which, wrap())
def step():
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
on["v"] = True
n = 10
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"{WL} batch 2: {(time.perf_counter() - t0) / n * 1e3:.1f} ms/step")
for (name, which), (c, t, _) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {name + '.' + which:34s} {c / n:6.1f} calls/step {t / n * 1e3:7.2f} ms/step {t / c * 1e6:7.1f} us/call")
