import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import recipes
from buctd_amd import models, ops
from buctd_amd.models.hrnet_common import HighResolutionModule
dev = torch.device("cuda:0")
cfg, omodel, x, _ = recipes.build("coam_w48_384x288")
m = getattr(models, cfg.MODEL.NAME).get_pose_net(cfg, is_train=False)
m.load_state_dict(omodel.state_dict(), strict=True)
m = m.to(dev).eval()
rec = {}
def hook(name):
    def f(mod, inp, out):
        rec.setdefault(name, []).append([o.detach().clone() for o in out])
    return f
for n, mod in m.named_modules():
    if isinstance(mod, HighResolutionModule):
        mod.register_forward_hook(hook(n))
xd = x.to(dev)
for on in (True, False, True):
    ops._branch["on"] = on
    with torch.no_grad():
        y = m(xd)
    torch.cuda.synchronize()
    print("fork", on, "out absmax", float(y.abs().max()))
for n, (a, b, c) in rec.items():
    d = [float((u - v).abs().max()) for u, v in zip(a, b)] + [float((u - v).abs().max()) for u, v in zip(c, b)]
    print(n, ["%.2e" % v for v in d])
