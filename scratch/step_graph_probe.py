"""Eager engine against engine.StepGraph on one box: ms per step and host enqueue time per step, for a train workload of bench.py.
C4 (train-mode dropout) can only be TIMED under a graph (--repeat-masks: every replay repeats one mask - not a training mode)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="train_c2")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeat-masks", action="store_true")
    ap.add_argument("--streams", default="single")
    args = ap.parse_args()
    from buctd_amd import engine, models, ops
    from buctd_amd.core.loss import JointsMSELoss
    rank, world, device = engine.init_distributed()
    ops.set_conv_math("bf16x6")
    bench.first_touch(device)
    make_cfg, module, metric, describe, rshape = bench.TRAIN_WORKLOADS[args.workload]
    cfg = make_cfg(args.batch)
    torch.manual_seed(1234)
    ops.manual_seed(1234)
    net = getattr(models, module).get_pose_net(cfg, is_train=True).to(device)
    model = engine.DataParallel(net)
    optimizer = engine.get_optimizer(cfg, model)
    model.flatten()
    criterion = JointsMSELoss(cfg.LOSS.USE_TARGET_WEIGHT)
    x, target, weight = bench.synthetic_batch(cfg, args.batch, device, seed=100)
    model.train()

    def eager():
        out = model(x)
        loss = criterion(out, target, weight)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        return loss

    gs = engine.StepGraph(model, criterion, optimizer, warmup=0, streams=args.streams, allow_repeated_dropout_masks=args.repeat_masks)

    def graphed():
        return gs(x, target, weight)[1]

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = 0.0
        for _ in range(n):
            a = time.perf_counter()
            loss = fn()
            host += time.perf_counter() - a
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return 1e3 * dt / n, 1e3 * host / n, float(loss)

    for _ in range(5):
        eager()
    rows = []
    for rnd in range(2):
        rows.append(("eager",) + timed(eager, args.steps))
        t0 = time.perf_counter()
        graphed()
        torch.cuda.synchronize()
        if rnd == 0:
            print(f"capture + first replay: {time.perf_counter() - t0:.2f} s", flush=True)
        rows.append(("graph",) + timed(graphed, args.steps))
    for name, ms, host, loss in rows:
        print(f"{args.workload} batch {args.batch} {name}: {ms:.2f} ms/step = {args.batch / ms * 1e3:.1f} img/s, host {host:.2f} ms/step, "
              f"loss {loss:.6f}", flush=True)


if __name__ == "__main__":
    main()
