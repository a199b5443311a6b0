import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
ops.set_conv_math("bf16x6")
dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "2"))
cfg = bench.coam_w48_cfg(B)
net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, B, dev, 1)
crit = JointsMSELoss(True)
acc = [0.0] * 5
def step(rec):
    t0 = time.perf_counter(); out = model(x); t1 = time.perf_counter()
    loss = crit(out, tgt, wt); opt.zero_grad(); t2 = time.perf_counter()
    loss.backward(); t3 = time.perf_counter()
    opt.step(); t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[i] += d
for _ in range(3): step(False)
n = 10
for _ in range(n): step(True)
print("batch", B, "host ms: forward %.1f | loss+zero_grad %.1f | backward %.1f | adam %.1f | final sync wait %.1f" % tuple(a / n * 1e3 for a in acc))
