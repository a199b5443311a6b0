"""log the gathered-convolution forward calls of one small model (arguments) : python scratch/diag_gcalls.py <recipe>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import recipes
from buctd_amd import ops
import test_gpu_models as T
dev = torch.device("cuda:0")
cfg, omodel, x, joints = recipes.build(sys.argv[1])
ops._GCONV_MASK = 5
orig = ops._gconv_fwd
seen = {}
def logged(x, w, d, bias, scale, shift, residual, relu, stats):
    key = (d.N, d.H, d.W, d.Ci, d.Co, d.R, d.stride, bias is not None, scale is not None, residual is not None, bool(relu), bool(stats),
           x.is_contiguous(), tuple(x.stride()))
    seen[key] = seen.get(key, 0) + 1
    return orig(x, w, d, bias, scale, shift, residual, relu, stats)
ops._gconv_fwd = logged
m = T.product_model(cfg, omodel, dev).train()
m(x.to(dev))
for k, v in sorted(seen.items()):
    print(v, "x N,H,W,Ci,Co,R,stride", k[:7], "bias", k[7], "scale", k[8], "res", k[9], "relu", k[10], "stats", k[11], "contig", k[12], k[13])
