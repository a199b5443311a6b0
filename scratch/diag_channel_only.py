import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_models as tm
from oracle import recipes
from buctd_amd import ops
from buctd_amd.core.loss import JointsMSELoss
dev = torch.device("cuda:0")
for name in ("coam_w16_96x64_channel_only", "coam_w16_96x64_colored"):
  for seed in (1234, 1, 2):
    for mode in ("bf16x6", "fp32"):
        ops.set_conv_math(mode)
        cfg, omodel, x, joints = recipes.build(name, seed=seed)
        tgt, wt = recipes.make_targets(cfg, joints, 77)
        m = tm.product_model(cfg, omodel, dev).train(); recipes.set_dropout(m, 0.0)
        y = m(x.to(dev)); loss = JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev)); loss.backward()
        g64, g32 = tm._oracle_grads(omodel, x, tgt, wt, torch.float64), tm._oracle_grads(omodel, x, tgt, wt, torch.float32)
        params = dict(m.named_parameters()); gmax = max(v.norm().item() for v in g64.values())
        eh, ec = [], []
        for k in g64:
            den = g64[k].norm().item()
            if den <= 1e-6 * gmax or params[k].grad is None: continue
            eh.append((params[k].grad.detach().cpu().double() - g64[k]).norm().item() / den); ec.append((g32[k].double() - g64[k]).norm().item() / den)
        print(f"{name} seed {seed} {mode}: median hip {np.median(eh):.2e} cpu {np.median(ec):.2e}; max hip {max(eh):.2e} cpu {max(ec):.2e}", flush=True)

# which parameters carry the worst relative error (last configuration evaluated above is re-run for seed 1, bf16x6)
ops.set_conv_math("bf16x6")
name = "coam_w16_96x64_channel_only"
cfg, omodel, x, joints = recipes.build(name)
tgt, wt = recipes.make_targets(cfg, joints, 77)
m = tm.product_model(cfg, omodel, dev).train(); recipes.set_dropout(m, 0.0)
y = m(x.to(dev)); loss = JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev)); loss.backward()
g64, g32 = tm._oracle_grads(omodel, x, tgt, wt, torch.float64), tm._oracle_grads(omodel, x, tgt, wt, torch.float32)
params = dict(m.named_parameters()); gmax = max(v.norm().item() for v in g64.values())
rows = []
for k in g64:
    den = g64[k].norm().item()
    if den <= 1e-6 * gmax or params[k].grad is None: continue
    rows.append(((params[k].grad.detach().cpu().double() - g64[k]).norm().item() / den, (g32[k].double() - g64[k]).norm().item() / den, den / gmax, k))
for r in sorted(rows, reverse=True)[:8]:
    print(f"  hip {r[0]:.2e} cpu {r[1]:.2e} |g|/gmax {r[2]:.2e} {r[3]}")
