"""conditioning of a small recipe's train step per recipe seed: fp32 CPU oracle and the HIP path (gathered convs on / off) vs
the fp64 oracle.  python scratch/seed_scan.py <recipe> seed [seed ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import recipes
from buctd_amd import ops
from buctd_amd.core.loss import JointsMSELoss
import test_gpu_models as T
name = sys.argv[1]
dev = torch.device("cuda:0")
torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
for seed in [int(v) for v in sys.argv[2:]]:
    cfg, omodel, x, joints = recipes.build(name, seed=seed)
    tgt, wt = recipes.make_targets(cfg, joints, 77)
    g64, g32 = T._oracle_grads(omodel, x, tgt, wt, torch.float64), T._oracle_grads(omodel, x, tgt, wt, torch.float32)
    gmax = max(v.norm().item() for v in g64.values())
    keys = [k for k in g64 if g64[k].norm().item() > 1e-6 * gmax]
    e32 = [(g32[k].double() - g64[k]).norm().item() / g64[k].norm().item() for k in keys]
    out = [f"seed {seed}: cpu32 med {np.median(e32):.1e} max {max(e32):.1e}"]
    for mask in (15, 0):
        ops._GCONV_MASK = mask
        m = T.product_model(cfg, omodel, dev).train()
        recipes.set_dropout(m, 0.0)
        y = m(x.to(dev))
        JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev)).backward()
        p = dict(m.named_parameters())
        eh = [(p[k].grad.detach().cpu().double() - g64[k]).norm().item() / g64[k].norm().item() for k in keys]
        out.append(f"hip[{mask}] med {np.median(eh):.1e} max {max(eh):.1e}")
    print(" | ".join(out), flush=True)
