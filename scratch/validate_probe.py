"""validate() throughput with the flip test: two forwards against one paired forward, eager and on a ForwardGraph, by TEST batch."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core import function
from buctd_amd.core.loss import JointsMSELoss
dev = torch.device("cuda:0")
ops.set_conv_math("bf16x6")
WL = sys.argv[1] if len(sys.argv) > 1 else "train_c2"
mk, module = bench.TRAIN_WORKLOADS[WL][0], bench.TRAIN_WORKLOADS[WL][1]


class Dataset:
    def __init__(self, n, k, size):
        self.n, self.image_size = n, size
        self.flip_pairs = [[a, a + 1] for a in range(1, k - 1, 2)]
        self.kpt_colors = [[(37 * j) % 256, (91 * j) % 256, (53 * j) % 256] for j in range(k)]
    def __len__(self): return self.n
    def evaluate(self, *a, **k): return {"AP": 0.0}, 0.0


for b in (1, 4, 16, 32):
    cfg = mk(b).clone(); cfg.defrost(); cfg.TEST.FLIP_TEST = True; cfg.TEST.POST_PROCESS = True; cfg.PRINT_FREQ = 10 ** 6; cfg.freeze()
    torch.manual_seed(0)
    net = getattr(models, module).get_pose_net(cfg, is_train=False).to(dev).eval()
    w, h = cfg.MODEL.IMAGE_SIZE; k = cfg.MODEL.NUM_JOINTS; hw, hh = cfg.MODEL.HEATMAP_SIZE
    nb = 12
    g = torch.Generator().manual_seed(1)
    loader = []
    for i in range(nb):
        meta = {"center": torch.rand(b, 2, generator=g) * 100 + 50, "scale": torch.rand(b, 2, generator=g) + 0.5, "score": torch.rand(b, generator=g),
                "annotation_id": torch.arange(b) + b * i, "image": ["x.jpg"] * b,
                "cond_joints": torch.cat([torch.rand(b, k, 2, generator=g) * min(w, h), torch.zeros(b, k, 1)], 2), "cond_joints_vis": torch.ones(b, k, 3)}
        loader.append((torch.randn(b, 6, h, w, generator=g), torch.rand(b, k, hh, hw, generator=g), torch.ones(b, k, 1), meta))
    ds = Dataset(nb * b, k, cfg.MODEL.IMAGE_SIZE)
    row = []
    for paired in (False, True):
        for wrap in (False, True):
            function.PAIRED_FLIP_FORWARD = paired
            model = engine.ForwardGraph(net, warmup=1) if wrap else net
            function.validate(cfg, loader[:4], ds, model, JointsMSELoss(True), "/tmp", "/tmp", None)      # settle
            torch.cuda.synchronize(); t0 = time.perf_counter()
            function.validate(cfg, loader, ds, model, JointsMSELoss(True), "/tmp", "/tmp", None)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            row.append(f"{'paired' if paired else 'two fwd'}{' + graph' if wrap else ''}: {nb * b / dt:.0f}")
    print(f"{WL} validate() with flip test, TEST batch {b}, persons/s: " + " | ".join(row), flush=True)
