"""Histogram of convolution calls outside the native blocks (shape, direction, math path) in one train step: python scratch/conv_census.py [train_c4|train_c3|train_c2]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
ops.set_conv_math("bf16x6")
import buctd_amd.ops as _o; _o._side["on"] = False; _o._branch["on"] = False   # serial streams: event brackets = kernel time
dev = torch.device("cuda:0")
WL = sys.argv[1] if len(sys.argv) > 1 else "train_c4"
cfg = bench.TRAIN_WORKLOADS[WL][0](32)
model = getattr(models, bench.TRAIN_WORKLOADS[WL][1]).get_pose_net(cfg, is_train=True).to(dev).train()
x, tgt, wt = bench.synthetic_batch(cfg, 32, dev, 1)
from buctd_amd.core.loss import JointsMSELoss
crit = JointsMSELoss(True)
hist = collections.Counter()
def wrap(name, fn, shape_of):
    def f(*a, **k):
        d = shape_of(*a, **k)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(); out = fn(*a, **k); ev[1].record()
        times.append((name, d, ev))
        return out
    return f
times = []
def d_fwd(x, w, bias=None, stride=1, pad=0, **k):
    d = ops.conv_desc(x.shape, ops._wshape(w), stride, pad); return (d.H, d.W, d.Ci, d.Co, d.R, d.stride, "b3" if ops._bf16x3_ok(d) else "f32")
def d_dg(dy, w, x_shape, stride=1, pad=0, **k):
    d = ops.conv_desc(x_shape, ops._wshape(w), stride, pad); return (d.H, d.W, d.Ci, d.Co, d.R, d.stride, "b3" if ops._bf16x3_ok(d) else "f32")
def d_wg(x, dy, w, stride=1, pad=0, **k):
    d = ops.conv_desc(x.shape, ops._wshape(w), stride, pad); return (d.H, d.W, d.Ci, d.Co, d.R, d.stride, "b3" if ops._bf16x3_ok(d) else "f32")
ops.conv_fwd = wrap("fwd", ops.conv_fwd, d_fwd)
ops.conv_dgrad = wrap("dgrad", ops.conv_dgrad, d_dg)
ops.conv_wgrad = wrap("wgrad", ops.conv_wgrad, d_wg)
for it in range(2):
    times.clear()
    loss = crit(model(x), tgt, wt); loss.backward()
    model.zero_grad(set_to_none=True)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, d, ev in times:
    a = agg[(name,) + d]; a[0] += 1; a[1] += ev[0].elapsed_time(ev[1]) * 1e3
tot = sum(v[1] for v in agg.values())
print(f"total conv time (event-bracketed, includes launch gaps) {tot/1e3:.1f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{k[0]:6s} {k[1]:3d}x{k[2]:<3d} {k[3]:4d}->{k[4]:<4d} k{k[5]} s{k[6]} {k[7]:4s} n={n:3d} total {t/1e3:7.2f} ms avg {t/n:7.1f} us")
