import sys, os, torch, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from buctd_amd import ops
dev = torch.device('cuda:0')
ops.set_conv_math("bf16x3")
N, H, W, Ci, Co = 1, 6, 5, 32, 32
for case in range(3):
    x = torch.zeros(N, Ci, H, W); dy = torch.zeros(N, Co, H, W)
    if case == 0:
        x[0, 3, 2, 2] = 1.0; dy[0, 5, 2, 2] = 1.0
    elif case == 1:
        x[0, 3, 1, 4] = 2.0; dy[0, 5, 2, 3] = 1.0; dy[0, 17, 0, 4] = 3.0
    else:
        x = torch.randn(N, Ci, H, W); dy = torch.randn(N, Co, H, W)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    dw = ops.conv_wgrad(x.permute(0,2,3,1).contiguous().to(dev), dy.permute(0,2,3,1).contiguous().to(dev),
                        w.detach().contiguous(memory_format=torch.channels_last).to(dev), 1, 1).cpu()
    ref = w.grad
    print("case", case, "max err", (dw - ref).abs().max().item(), "ref nnz", int((ref != 0).sum()), "got nnz", int((dw.abs() > 1e-6).sum()))
    if case < 2:
        print(" ref:", [(tuple(i.tolist()), round(ref[tuple(i)].item(), 3)) for i in (ref != 0).nonzero()][:8])
        print(" got:", [(tuple(i.tolist()), round(dw[tuple(i)].item(), 3)) for i in (dw.abs() > 1e-6).nonzero()][:12])
