"""cProfile of the host side of a train step, main thread + the autograd device thread (via threading.setprofile), batch 2."""
import cProfile, pstats, os, sys, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
ops.set_conv_math("bf16x6")
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
# the backward functions run on autograd's device thread: wrap them so that they are profiled there
prof_bw = cProfile.Profile()
for name in dir(ops):
    cls = getattr(ops, name)
    if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
        raw = cls.backward
        def wrap(raw=raw):
            def f(*a, **k):
                prof_bw.enable()
                try:
                    return raw(*a, **k)
                finally:
                    prof_bw.disable()
            return staticmethod(f)
        cls.backward = wrap()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
print("=== main thread (forward, optimizer) ===")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
print("=== backward functions (device thread) ===")
pstats.Stats(prof_bw).sort_stats("tottime").print_stats(28)
