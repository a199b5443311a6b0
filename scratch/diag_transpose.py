"""Per-parameter gradient error of the TransPose-H train step (HIP vs fp64 oracle, fp32 CPU oracle beside it), listed in
backward order, to localise a lossy kernel.  gpurun -- python scratch/diag_transpose.py [recipe]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import recipes, core as ocore          # noqa: E402
from buctd_amd import models                        # noqa: E402
from buctd_amd.core.loss import JointsMSELoss       # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "transpose_w16_96x64"
dev = torch.device("cuda:0")
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
cfg, omodel, x, joints = recipes.build(name, seed=seed)
quiet = len(sys.argv) > 3
tgt, wt = recipes.make_targets(cfg, joints, 77)


def ograds(dtype):
    m = copy.deepcopy(omodel).to(dtype).train()
    recipes.set_dropout(m, 0.0)
    acts = {}
    loss = ocore.JointsMSELoss(True)(m(x.to(dtype)), tgt.to(dtype), wt.to(dtype))
    loss.backward()
    return {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None}, loss.item()


g64, l64 = ograds(torch.float64)
g32, l32 = ograds(torch.float32)
m = getattr(models, cfg.MODEL.NAME).get_pose_net(cfg, is_train=False)
m.load_state_dict(omodel.state_dict(), strict=True)
m = m.to(dev).train()
recipes.set_dropout(m, 0.0)
loss = JointsMSELoss(True)(m(x.to(dev)), tgt.to(dev), wt.to(dev))
loss.backward()
print(f"loss hip {loss.item():.8f} cpu32 {l32:.8f} fp64 {l64:.8f}")
params = dict(m.named_parameters())
order = [k for k, _ in omodel.named_parameters() if k in g64][::-1]     # roughly backward order
gmax = max(v.norm().item() for v in g64.values())
for k in order:
    den = g64[k].norm().item()
    if den <= 1e-6 * gmax or params[k].grad is None:
        continue
    eh = (params[k].grad.detach().cpu().double() - g64[k]).norm().item() / den
    ec = (g32[k].double() - g64[k]).norm().item() / den
    flag = " <<<" if eh > 10 * max(ec, 1e-6) else ""
    worst = max(locals().get("worst", 0.0), eh / max(ec, 1e-7))
    if not quiet or flag:
        print(f"{k:70s} hip {eh:.2e}  cpu32 {ec:.2e}{flag}")

print(f"seed {seed}: worst hip/cpu32 error ratio {worst:.1f}")
