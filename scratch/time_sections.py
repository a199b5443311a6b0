"""CoAM-W48 train forward / backward by network section (CUDA events at the section boundaries; backward boundaries through
gradient hooks on the boundary tensors).   python scratch/time_sections.py [c4|c3|c2]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from buctd_amd import ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
if which == "c4":
    from buctd_amd.models.pose_hrnet_coam import get_pose_net
    cfg = bench.coam_w48_cfg(32); x = torch.randn(32, 6, 384, 288, device=dev)
else:
    from buctd_amd.models.pose_hrnet import get_pose_net
    cfg = bench.prenet_cfg(32, 48, (288, 384)) if which == "c3" else bench.prenet_cfg(32, 32, (192, 256))
    x = torch.randn(32, 6, cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0], device=dev)
model = get_pose_net(cfg, is_train=True).to(dev).train()
fw, bw = [], []


def mark(name, t):
    e = torch.cuda.Event(enable_timing=True); e.record(); fw.append((name, e))
    t0 = t[0] if isinstance(t, (list, tuple)) else t
    if torch.is_tensor(t0) and t0.requires_grad:
        def hook(g, name=name):
            ev = torch.cuda.Event(enable_timing=True); ev.record(); bw.append((name, ev)); return g
        t0.register_hook(hook)
    return t


def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        return mark(name, f(*a, **k))
    setattr(obj, attr, g)


wrap(model, "stem", "stem+layer1")
orig_enter = model.enter_stage
model.enter_stage = lambda s, prev, first=False: mark("transition%d" % (s - 1), orig_enter(s, prev, first))
for s in (2, 3, 4):
    st = getattr(model, "stage%d" % s)
    f = st.forward
    st.forward = (lambda f, s: lambda *a, **k: mark("stage%d" % s, f(*a, **k)))(f, s)
for i in (1, 2, 3, 4):
    att = getattr(model, "stage%d_att" % i, None)
    if att is not None:
        f = att.forward
        att.forward = (lambda f, i: lambda *a, **k: mark("att%d" % i, f(*a, **k)))(f, i)
for it in range(3):
    fw.clear(); bw.clear()
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    out = model(x)
    s1 = torch.cuda.Event(enable_timing=True); s1.record()
    out.sum().backward()
    ops.wait_side_stream()
    s2 = torch.cuda.Event(enable_timing=True); s2.record()
    torch.cuda.synchronize()
    for p in model.parameters():
        p.grad = None
    ops.acc_pool.reset(dev)
print(f"forward {s0.elapsed_time(s1):.2f} ms, backward {s1.elapsed_time(s2):.2f} ms")
prev = s0
for name, e in fw:
    print(f"  fwd {name:14s} {prev.elapsed_time(e):7.2f} ms"); prev = e
print(f"  fwd {'final':14s} {prev.elapsed_time(s1):7.2f} ms")
prev = s1
for name, e in bw:
    print(f"  bwd down to the output of {name:14s} {prev.elapsed_time(e):7.2f} ms"); prev = e
print(f"  bwd {'rest (stem+layer1, side-stream join)':14s} {prev.elapsed_time(s2):7.2f} ms")
