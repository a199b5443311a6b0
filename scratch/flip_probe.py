"""Is a gradient mismatch at ONE input a flipped ReLU or an arithmetic defect?  A pre-activation within round-off of zero lands
on either side of its ReLU depending on summation order; an arithmetic defect does not care about 1e-6 of noise on the input.
For the recipe input (seed 1234) of a small net and perturbed copies of it: the gradient errors of the HIP path against the
fp64 oracle at the input itself and at K copies with relative noise 1e-6 - a flip shows as a BIMODAL worst error (copies on the
oracle's side of the ReLU sit at the fp32 level), a defect as the same error on every copy.
    python scratch/flip_probe.py [net] [alts, e.g. 0 1 2 3 4]   (tests/ helpers; needs oracle/ - a test tool)"""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_models as T
from oracle import recipes
from buctd_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else "transpose_w16_96x64"
alts = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 3, 4]
dev = torch.device("cuda:0")
cfg, omodel, x, joints = recipes.build(name, seed=1234)
tgt, wt = recipes.make_targets(cfg, joints, 77)
m = T.product_model(cfg, omodel, dev).train()
recipes.set_dropout(m, 0.0)
rgb = (torch.arange(x.shape[1]).view(1, -1, 1, 1) < 3)
for alt in alts:
    xa = x
    if alt:
        ga = torch.Generator().manual_seed(9100 + alt)
        xa = x + 0.25 * torch.randn(x.shape, generator=ga) * rgb
    for math in ("bf16x6", "fp32"):
        ops.set_conv_math(math)
        row = []
        for k in range(4):
            xk = xa
            if k:
                gk = torch.Generator().manual_seed(77000 + 10 * alt + k)
                xk = xa * (1 + 1e-6 * torch.randn(xa.shape, generator=gk))
            mh, mc, wh, wc, wk = T._grad_errors(m, omodel, xk, tgt, wt, dev)
            on = T._grad_errors.onset
            row.append(f"[{'x' if not k else 'x+1e-6 #%d' % k}: median {mh:.1e}/{mc:.1e} worst {wh:.1e}/{wc:.1e} at {wk}"
                       f"{'' if on is None else ' onset ' + on[1] + ' after ' + str(on[0]) + ' clean'}]")
        print(f"{name} alt {alt} {math}:\n   " + "\n   ".join(row), flush=True)
ops.set_conv_math("bf16x6")
