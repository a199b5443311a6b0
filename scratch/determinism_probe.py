"""Is the eval forward bit-reproducible (a) twice in one process, (b) when the weights were first used at another batch?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buctd_amd import models, ops
from oracle import recipes
dev = torch.device("cuda:0")
cfg, omodel, x, joints = recipes.build("coam_w16_96x64_colored")
def fresh():
    net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)
    net.load_state_dict(omodel.state_dict(), strict=True)
    return net.to(dev).eval()
with torch.no_grad():
    n1 = fresh(); a = n1(x.to(dev)); b = n1(x.to(dev))
    print("same net twice:", torch.equal(a, b), (a - b).abs().max().item())
    n2 = fresh(); _ = n2((x * 1.01).to(dev)); c = n2(x.to(dev))
    print("other first batch:", torch.equal(a, c), (a - c).abs().max().item())
    for st in ("0", "1"):
        ops._branch["on"] = st == "1"
        n3 = fresh(); d = n3(x.to(dev))
        print("branch streams", st, torch.equal(a, d), (a - d).abs().max().item())
