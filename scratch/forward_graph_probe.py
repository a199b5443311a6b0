"""Eval-mode forward: eager engine against engine.ForwardGraph by batch size (persons per call), for the networks of bench.py."""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="train_c2")
    ap.add_argument("--batches", default="1,2,4,8,16,32")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.set_conv_math("bf16x6")
    if args.workload == "infer_c5":
        cfg, module = bench.transpose_a6_cfg(1), "transpose_h"
    else:
        mk, module = bench.TRAIN_WORKLOADS[args.workload][0], bench.TRAIN_WORKLOADS[args.workload][1]
        cfg = mk(1)
    torch.manual_seed(1)
    net = getattr(models, module).get_pose_net(cfg, is_train=False).to(dev).eval()
    fg = engine.ForwardGraph(net, warmup=2, static_output=True, autoselect=False)
    w, h = cfg.MODEL.IMAGE_SIZE
    with torch.no_grad():
        for b in [int(v) for v in args.batches.split(",")]:
            x = torch.randn(b, 6, h, w, device=dev)
            for _ in range(4):
                net(x); fg(x)
            res = []
            for fn in (net, fg):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    fn(x)
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / 30 * 1e3)
            same = torch.equal(net(x) if not isinstance(net(x), list) else net(x)[-1], fg(x) if not isinstance(fg(x), list) else fg(x)[-1])
            print(f"{args.workload} eval forward, batch {b}: eager {res[0]:.2f} ms ({b / res[0] * 1e3:.0f} persons/s), graph {res[1]:.2f} ms "
                  f"({b / res[1] * 1e3:.0f} persons/s), x{res[0] / res[1]:.2f}, identical {same}", flush=True)


if __name__ == "__main__":
    main()
