import sys, copy, torch, numpy as np
sys.path.insert(0, '.')
from oracle import models as om
from buctd_amd.models import hrnet_common as hc
from buctd_amd import nn as bnn, ops
dev = torch.device('cuda:0')
torch.manual_seed(0)

def run(name, omod, pmod, x, nhwc_in=True):
    pmod.load_state_dict(omod.state_dict(), strict=True)
    pmod = pmod.to(dev).train()
    res = {}
    for dt in (torch.float64, torch.float32):
        m = copy.deepcopy(omod).to(dt).train()
        xi = x.to(dt).clone().requires_grad_(True)
        y = m(xi)
        y = y[0] if isinstance(y, (list, tuple)) else y
        g = torch.Generator().manual_seed(1)
        dy = torch.randn(y.shape, generator=g).to(dt)
        y.backward(dy)
        res[dt] = (y.detach(), xi.grad, {k: p.grad for k, p in m.named_parameters()})
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    yd = pmod(xd)
    yd = yd[0] if isinstance(yd, (list, tuple)) else yd
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(res[torch.float64][0].shape, generator=g)
    yd.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev))
    y64, gx64, gp64 = res[torch.float64]
    y32, gx32, gp32 = res[torch.float32]
    def e(a, b): return ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()
    print(f"[{name}] fwd: hip {e(yd.detach().permute(0,3,1,2), y64):.2e} cpu32 {e(y32, y64):.2e} | dx: hip {e(xd.grad.permute(0,3,1,2), gx64):.2e} cpu32 {e(gx32, gx64):.2e}")
    worst = max(((e(p.grad, gp64[k]), e(gp32[k], gp64[k]), k) for k, p in pmod.named_parameters()), key=lambda t: t[0])
    print(f"     worst param grad: hip {worst[0]:.2e} cpu32 {worst[1]:.2e} ({worst[2]})")

class OCB(torch.nn.Module):
    def __init__(s, ci, co, k=3, st=1, relu=True):
        super().__init__(); s.c = om.cbr(ci, co, k, st, relu)
    def forward(s, x): return s.c(x)
class PCB(torch.nn.Module):
    def __init__(s, ci, co, k=3, st=1, relu=True):
        super().__init__(); s.c = bnn.ConvBN(bnn.Conv2d(ci, co, k, st, (k-1)//2, bias=False), bnn.BatchNorm2d(co), bnn.ReLU(True) if relu else None)
    def forward(s, x): return s.c(x)

for (N,H,W,ci,co,k,st,relu) in [(3,24,16,16,16,3,1,True),(3,3,2,128,128,3,1,True),(3,6,4,64,128,3,2,False),(3,12,8,64,16,1,1,False),(32,12,9,384,384,3,1,True)]:
    x = torch.randn(N, ci, H, W)
    run(f"convbn {N}x{H}x{W} {ci}->{co} k{k}s{st} relu={relu}", OCB(ci,co,k,st,relu), PCB(ci,co,k,st,relu), x)
x = torch.randn(3, 16, 24, 16)
run("basicblock", om.BasicBlock(16,16), hc.BasicBlock(16,16), x)
x = torch.randn(3, 128, 3, 2)
run("basicblock lowres", om.BasicBlock(128,128), hc.BasicBlock(128,128), x)
o4, _ = om.make_layer(om.BasicBlock, 32, 32, 4); p4, _ = hc.make_residual_layer(hc.BasicBlock, 32, 32, 4)
run("4 basicblocks", o4, p4, torch.randn(3, 32, 12, 8))

print("---- HR modules")
def run_hr(nb, mso=True):
    ch = [16 * 2 ** i for i in range(nb)]
    o = om.HighResolutionModule(nb, om.BasicBlock, [1]*nb, list(ch), list(ch), 'SUM', mso)
    p = hc.HighResolutionModule(nb, hc.BasicBlock, [1]*nb, list(ch), list(ch), 'SUM', mso)
    p.load_state_dict(o.state_dict(), strict=True)
    p = p.to(dev).train()
    xs = [torch.randn(3, ch[i], 24 >> i, 16 >> i) for i in range(nb)]
    out = {}
    for dt in (torch.float64, torch.float32):
        m = copy.deepcopy(o).to(dt).train()
        xi = [x.to(dt).clone().requires_grad_(True) for x in xs]
        ys = m(list(xi))
        loss = sum((y * torch.linspace(-1, 1, y.numel(), dtype=dt).view(y.shape)).sum() for y in ys)
        loss.backward()
        out[dt] = ([y.detach() for y in ys], [x.grad for x in xi], {k: q.grad for k, q in m.named_parameters()})
    xd = [x.permute(0,2,3,1).contiguous().to(dev).requires_grad_(True) for x in xs]
    yd = p(list(xd))
    for y, y64 in zip(yd, out[torch.float64][0]):
        wgt = torch.linspace(-1, 1, y64.numel()).view(y64.shape).permute(0,2,3,1).contiguous().to(dev)
        y.backward(wgt, retain_graph=True)
    def e(a, b): return ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()
    for i in range(len(yd)):
        print(f"[hr{nb} out{i}] fwd hip {e(yd[i].detach().permute(0,3,1,2), out[torch.float64][0][i]):.2e} cpu32 {e(out[torch.float32][0][i], out[torch.float64][0][i]):.2e}")
    for i in range(nb):
        print(f"[hr{nb} dx{i}] hip {e(xd[i].grad.permute(0,3,1,2), out[torch.float64][1][i]):.2e} cpu32 {e(out[torch.float32][1][i], out[torch.float64][1][i]):.2e}")
    errs = sorted(((e(q.grad, out[torch.float64][2][k]), e(out[torch.float32][2][k], out[torch.float64][2][k]), k) for k, q in p.named_parameters()), reverse=True)[:4]
    for t in errs: print(f"     param grad: hip {t[0]:.2e} cpu32 {t[1]:.2e} ({t[2]})")
run_hr(2); run_hr(4); run_hr(4, False)
