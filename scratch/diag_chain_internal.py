"""internal pre-activations of the BasicBlock chains (not module outputs) under two math variants: flipped ReLUs?
python scratch/diag_chain_internal.py <recipe> maskA maskB"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import recipes
from buctd_amd import ops
import test_gpu_models as T
name, ma, mb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
cfg, omodel, x, joints = recipes.build(name)
from buctd_amd.core.loss import JointsMSELoss
tgt, wt = recipes.make_targets(cfg, joints, 77)
orig = ops.BasicChainFn.backward
def run(mask):
    ops._GCONV_MASK = mask
    rec = []
    def bwd(ctx, dy):
        xs, act, stat = ctx.saved_tensors
        blocks = ctx.blocks
        rec.append((tuple(xs.shape), act.clone(), stat.clone(), [(b[1].weight.detach().clone(), b[1].bias.detach().clone(),
                                                                   b[3].weight.detach().clone(), b[3].bias.detach().clone()) for b in blocks], xs.detach().clone()))
        return orig(ctx, dy)
    ops.BasicChainFn.backward = staticmethod(bwd)
    m = T.product_model(cfg, omodel, dev).train()
    recipes.set_dropout(m, 0.0)
    y = m(x.to(dev))
    JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev)).backward()
    torch.cuda.synchronize()
    ops.BasicChainFn.backward = staticmethod(orig)
    return rec
ra, rb = run(ma), run(mb)
for ci, (A, B) in enumerate(zip(ra, rb)):
    shape, actA, statA, gb, xA = A
    _, actB, statB, _, xB = B
    for k in range(actA.shape[0]):
        for which, zi, mi, gi in (("bn1", 0, 0, 0), ("bn2", 1, 2, 2)):
            pa = (actA[k, zi] - statA[k, mi]) * statA[k, mi + 1] * gb[k][gi] + gb[k][gi + 1]
            pb = (actB[k, zi] - statB[k, mi]) * statB[k, mi + 1] * gb[k][gi] + gb[k][gi + 1]
            if which == "bn2":
                xa = xA if k == 0 else actA[k - 1, 2]; xb_ = xB if k == 0 else actB[k - 1, 2]
                pa = pa + xa; pb = pb + xb_
            flips = int(((pa > 0) != (pb > 0)).sum().item())
            if flips or statA[k, mi + 1].max().item() > 100:
                near = torch.minimum(pa.abs(), pb.abs())[(pa > 0) != (pb > 0)]
                print(f"chain {ci} shape {shape} block {k} {which}: {flips} flipped pre-activations "
                      f"(|value| <= {near.max().item() if flips else 0:.2e}), max invstd {statA[k, mi + 1].max().item():.1f}, "
                      f"pre-activation max diff {(pa - pb).abs().max().item():.2e}")
