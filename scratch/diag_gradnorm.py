"""which gradient tensors of a small golden model deviate in norm from the reference: python scratch/diag_gradnorm.py <recipe> [BUCTD_GCONV_X6=0 to compare]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import recipes
from buctd_amd.core.loss import JointsMSELoss
import test_gpu_models as T
name = sys.argv[1]
dev = torch.device("cuda:0")
gold = np.load(os.path.join(T.GOLD, f"model_{name}.npz"), allow_pickle=False)
cfg, omodel, x, joints = recipes.build(name)
tgt, wt = recipes.make_targets(cfg, joints, 77)
m = T.product_model(cfg, omodel, dev).train()
recipes.set_dropout(m, 0.0)
y = m(x.to(dev))
loss = JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev))
loss.backward()
print("loss", loss.item(), float(gold["loss"]), "fwd err", np.abs(y.detach().cpu().numpy() - gold["train_out"]).max())
g64 = T._oracle_grads(omodel, x, tgt, wt, torch.float64)
g32 = T._oracle_grads(omodel, x, tgt, wt, torch.float32)
params = dict(m.named_parameters())
rows = []
for k, gn in zip([str(s) for s in gold["grad_names"]], gold["grad_norms"]):
    g = params[k].grad
    den = g64[k].norm().item()
    if den <= 1e-6 * max(v.norm().item() for v in g64.values()): continue
    eh = (g.detach().cpu().double() - g64[k]).norm().item() / den
    ec = (g32[k].double() - g64[k]).norm().item() / den
    rows.append((abs(g.norm().item() - gn) / max(gn, 1e-6), eh, ec, k, gn))
rows.sort(reverse=True)
for r in rows[:14]:
    print(f"norm dev {r[0]:.3e}  err hip {r[1]:.2e} cpu32 {r[2]:.2e}  ref norm {r[4]:.3e}  {r[3]}")
print("median hip", np.median([r[1] for r in rows]), "cpu32", np.median([r[2] for r in rows]))
