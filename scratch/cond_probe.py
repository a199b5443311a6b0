import sys, copy, torch, statistics
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import recipes, cfg as ocfg, models as om, core as ocore
from test_gpu_models import _oracle_grads
torch.set_num_threads(8)
def probe(size, batch, mods):
    c = ocfg.hrnet_cfg(16, 17, size, "pose_hrnet", use_pre_net=True, stage_modules=mods)
    torch.manual_seed(1234); m = om.get_pose_net(c, False); recipes.randomize(m, 1235)
    x, j = recipes.make_inputs(c, batch, 1236, 3); recipes.calibrate_bn(m, x)
    tgt, wt = recipes.make_targets(c, j, 77)
    g64 = _oracle_grads(m, x, tgt, wt, torch.float64); g32 = _oracle_grads(m, x, tgt, wt, torch.float32)
    mx = max(v.norm().item() for v in g64.values())
    e = [((g32[k].double()-g64[k]).norm()/g64[k].norm()).item() for k in g64 if g64[k].norm().item() > 1e-6*mx]
    print(size, batch, mods, "median %.2e max %.2e" % (statistics.median(e), max(e)))
probe((64,96),3,(1,2,2)); probe((96,128),4,(1,2,2)); probe((128,192),4,(1,1,1)); probe((128,192),8,(1,2,2))
