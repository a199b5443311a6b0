"""which module's backward first differs between two math variants?  python scratch/diag_bwd.py <recipe> maskA maskB"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import recipes
from buctd_amd import ops
from buctd_amd.core.loss import JointsMSELoss
import test_gpu_models as T
name = sys.argv[1]
ma, mb = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
cfg, omodel, x, joints = recipes.build(name)
tgt, wt = recipes.make_targets(cfg, joints, 77)
def run(mask):
    ops._GCONV_MASK = mask
    m = T.product_model(cfg, omodel, dev).train()
    recipes.set_dropout(m, 0.0)
    rec, order = {}, []
    for n, mod in m.named_modules():
        def fh(mod, inp, out, n=n):
            if torch.is_tensor(out):
                order.append(n)
                out.register_hook(lambda g, n=n: rec.__setitem__(n, g.detach().clone()))
        mod.register_forward_hook(fh)
    y = m(x.to(dev))
    JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev)).backward()
    torch.cuda.synchronize()
    return rec, order, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
ra, order, pa = run(ma)
rb, _, pb = run(mb)
print("gradient w.r.t. each module OUTPUT, in forward order (relative difference between the variants):")
for n in order:
    if n in ra and n in rb and ra[n].shape == rb[n].shape:
        d = (ra[n] - rb[n]).norm().item() / max(rb[n].norm().item(), 1e-30)
        if d > 1e-4: print(f"  {d:.2e}  {n}")
print("parameter gradients with relative difference > 1e-3:")
for k in pa:
    d = (pa[k] - pb[k]).norm().item() / max(pb[k].norm().item(), 1e-30)
    if d > 1e-3: print(f"  {d:.2e}  {k}")
