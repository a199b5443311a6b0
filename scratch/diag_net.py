import sys, copy, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import recipes, core as ocore
from test_gpu_models import product_model, _oracle_grads
from buctd_amd.core.loss import JointsMSELoss
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else "prenet_w16_96x64"
cfg, omodel, x, joints = recipes.build(name)
tgt, wt = recipes.make_targets(cfg, joints, 77)
m = product_model(cfg, omodel, dev).train(); recipes.set_dropout(m, 0.0)
y = m(x.to(dev)); loss = JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev)); loss.backward()
g64 = _oracle_grads(omodel, x, tgt, wt, torch.float64); g32 = _oracle_grads(omodel, x, tgt, wt, torch.float32)
rows = []
for k, p in m.named_parameters():
    r = g64[k]; den = r.norm().item()
    rows.append(((p.grad.cpu().double() - r).norm().item() / max(den, 1e-30), (g32[k].double() - r).norm().item() / max(den, 1e-30), den, k))
rows.sort(reverse=True)
for r in rows[:14]: print(f"hip {r[0]:.2e} cpu32 {r[1]:.2e} |g| {r[2]:.3e} {r[3]}")
import statistics
print("median hip", statistics.median(r[0] for r in rows), "median cpu32", statistics.median(r[1] for r in rows))
# order of appearance: error vs depth
order = [k for k, _ in m.named_parameters()]
d = {r[3]: r for r in rows}
for k in order[::max(1, len(order)//40)]: print(f"  {d[k][0]:.2e} {d[k][1]:.2e} {k}")
