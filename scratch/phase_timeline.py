"""Forward/backward phase timeline of one CoAM-W48 train step (HIP events on the main stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
ops.set_conv_math("bf16x6")
dev = torch.device("cuda:0")
cfg = bench.coam_w48_cfg(32)
net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, 32, dev, 1)
crit = JointsMSELoss(True)
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
class Tap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, name):
        ctx.name = name; mark("F:" + name); return t.view_as(t)
    @staticmethod
    def backward(ctx, g):
        mark("B:" + ctx.name); return g, None
def tap(lst, name):
    return [Tap.apply(t, name) if i == 0 else t for i, t in enumerate(lst)]
def fwd(self, xin):
    xx = models.hrnet_common.to_device_input(xin)
    mark("F:start")
    feat = self.stem(ops.nchw_to_nhwc(xx, 0, 3))
    feat = Tap.apply(feat, "stem+layer1")
    xl = self.enter_stage(2, feat, first=True); xl = tap(xl, "trans1")
    yl = self.stage2(xl); yl = tap(yl, "stage2")
    xl = self.enter_stage(3, yl); xl = tap(xl, "trans2")
    xl = self.stage2_att(xl, xx); xl = tap(xl, "att")
    yl = self.stage3(xl); yl = tap(yl, "stage3")
    xl = self.enter_stage(4, yl); xl = tap(xl, "trans3")
    yl = self.stage4(xl); yl = tap(yl, "stage4")
    out = ops.ToNCHW.apply(self.final_layer(yl[0]))
    return Tap.apply(out, "final")
def step():
    marks.clear()
    out = fwd(net, x)
    loss = crit(out, tgt, wt); mark("F:loss")
    opt.zero_grad(); loss.backward(); mark("B:end"); opt.step(); mark("adam")
for _ in range(3): step()
torch.cuda.synchronize()
step(); torch.cuda.synchronize()
t0 = marks[0][1]
prev = 0.0
for name, e in marks:
    t = t0.elapsed_time(e)
    print(f"{name:16s} t={t:7.2f} ms  (+{t - prev:6.2f})")
    prev = t
