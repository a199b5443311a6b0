"""do two math variants of the same train-mode forward differ by flipped ReLUs?  python scratch/diag_flip.py <recipe> [maskA maskB]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import recipes
from buctd_amd import ops
import test_gpu_models as T
name = sys.argv[1]
ma, mb = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (15, 0)
dev = torch.device("cuda:0")
cfg, omodel, x, joints = recipes.build(name)
outs = {}
def run(mask):
    ops._GCONV_MASK = mask
    m = T.product_model(cfg, omodel, dev).train()
    recipes.set_dropout(m, 0.0)
    rec = {}
    hs = []
    for n, mod in m.named_modules():
        def hook(mod, inp, out, n=n):
            if torch.is_tensor(out): rec[n] = out.detach().clone()
        hs.append(mod.register_forward_hook(hook))
    m(x.to(dev))
    for h in hs: h.remove()
    return rec
a, b = run(ma), run(mb)
tot = 0
for n in a:
    if n not in b or a[n].shape != b[n].shape: continue
    za, zb = a[n], b[n]
    flips = ((za > 0) != (zb > 0)) & ((za == 0) | (zb == 0))       # an exact zero on one side: a ReLU output
    nf = int(flips.sum().item())
    if nf:
        mag = torch.maximum(za.abs(), zb.abs())[flips].max().item()
        tot += nf
        print(f"{n}: {nf} flipped ReLU outputs of {za.numel()} (largest value involved {mag:.2e}); max |diff| anywhere {((za - zb).abs().max().item()):.2e}")
print("total flips", tot)
print("module outputs whose values differ by more than 1e-4 (relative to their max):")
for n in a:
    if n in b and a[n].shape == b[n].shape:
        d = (a[n] - b[n]).abs().max().item() / max(b[n].abs().max().item(), 1e-30)
        if d > 1e-4: print(f"  {d:.2e}  {n}  shape {tuple(a[n].shape)}")
