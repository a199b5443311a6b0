import sys, torch, torch.nn.functional as F, numpy as np
sys.path.insert(0,'.')
from oracle import recipes
torch.set_num_threads(8)
name = sys.argv[1] if len(sys.argv)>1 else "coam_w16_96x64_colored"
cfg, m, x, j = recipes.build(name)
with torch.no_grad(): y32 = m(x)
orig = F.conv2d
def split(t):
    hi = t.bfloat16().float(); lo = (t - hi).bfloat16().float(); return hi, lo
def conv_x3(inp, w, b=None, *a, **k):
    ih, il = split(inp); wh, wl = split(w)
    return orig(ih, wh, b, *a, **k) + orig(ih, wl, None, *a, **k) + orig(il, wh, None, *a, **k)
def conv_bf16(inp, w, b=None, *a, **k):
    return orig(inp.bfloat16().float(), w.bfloat16().float(), b, *a, **k)
for tag, fn in (("bf16x3", conv_x3), ("plain bf16 inputs", conv_bf16)):
    F.conv2d = fn; torch.nn.functional.conv2d = fn
    with torch.no_grad(): y = m(x)
    F.conv2d = orig
    scale = y32.abs().max().item()
    print(f"{name} {tag}: max abs diff {(y-y32).abs().max().item():.3e} (scale {scale:.1f}, rel {(y-y32).abs().max().item()/scale:.2e}), argmax equal: {bool((y.flatten(2).argmax(2)==y32.flatten(2).argmax(2)).all())}")
