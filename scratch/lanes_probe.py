"""Serving throughput at a few persons per call: eager forward, ForwardGraph (one lane), ForwardGraph.submit (two lanes)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
dev = torch.device("cuda:0")
ops.set_conv_math("bf16x6")
for wl in ("train_c2", "train_c4", "infer_c5"):
    if wl == "infer_c5":
        cfg, module = bench.transpose_a6_cfg(1), "transpose_h"
    else:
        cfg, module = bench.TRAIN_WORKLOADS[wl][0](1), bench.TRAIN_WORKLOADS[wl][1]
    torch.manual_seed(1)
    net = getattr(models, module).get_pose_net(cfg, is_train=False).to(dev).eval()
    fg = engine.ForwardGraph(net, warmup=2, autoselect=False)
    w, h = cfg.MODEL.IMAGE_SIZE
    with torch.no_grad():
        for b in (1, 2, 4, 8):
            xs = [torch.randn(b, 6, h, w, device=dev) for _ in range(4)]
            for x in xs: net(x); fg(x); fg.submit(x).result()
            n = 60
            def run_eager():
                for i in range(n): net(xs[i % 4])
            def run_one():
                for i in range(n): fg(xs[i % 4])
            def run_two():
                hs = []
                for i in range(n):
                    hs.append(fg.submit(xs[i % 4]))
                    if len(hs) == 3: hs.pop(0).result()
                for hd in hs: hd.result()
            res = []
            for fn in (run_eager, run_one, run_two):
                fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
                res.append(n * b / (time.perf_counter() - t0))
            print(f"{wl} net, {b} per call, persons/s: eager {res[0]:.0f} | graph {res[1]:.0f} | graph, two lanes {res[2]:.0f}", flush=True)
