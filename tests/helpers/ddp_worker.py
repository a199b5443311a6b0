"""Worker for tests/test_gpu_ddp.py: N ranks (all on GPU 0, gloo) or one plain process train the same small model on
the same batch for two steps and dump a checksum of the parameters.  usage: ddp_worker.py OUT.npz"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from buctd_amd import engine, models, ops  # noqa: E402
from buctd_amd.core.loss import JointsMSELoss  # noqa: E402
from oracle import recipes  # noqa: E402


def main():
    rank, world, dev = engine.init_distributed()
    one_rank_rccl = os.environ.get("BUCTD_DDP_MODE") == "rccl1"
    if one_rank_rccl:
        # a process group of ONE rank over the production backend ("nccl" = RCCL): the exchange machinery runs for real
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        engine.reserve_streams(dev, data_parallel=True)      # as engine.init_distributed does in front of a real group
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    ops.set_conv_math(os.environ.get("BUCTD_CONV_MATH", "fp32"))
    cfg, omodel, x, joints = recipes.build("coam_w16_96x64_colored")
    net = getattr(models, cfg.MODEL.NAME).get_pose_net(cfg, is_train=True)
    net.load_state_dict(omodel.state_dict(), strict=True)
    net = net.to(dev).train()
    recipes.set_dropout(net, 0.0)
    model = engine.DataParallel(net, bucket_bytes=1 << 16,   # many small buckets: exercises the overlap machinery
                                exchange_in_world_of_one=one_rank_rccl)
    opt = engine.get_optimizer(cfg, model)
    tgt, wt = recipes.make_targets(cfg, joints, 77)
    crit = JointsMSELoss(True)
    if os.environ.get("BUCTD_DDP_MODE") == "shards":
        # every rank trains on its OWN shard (seeded by the rank); rank 0 dumps the exchanged, averaged gradient of every
        # parameter - the test holds it against the oracle's mean over the two replicas (per-replica BatchNorm)
        xs, js = recipes.make_inputs(cfg, x.shape[0], 900 + rank, 3)
        ts, ws = recipes.make_targets(cfg, js, 950 + rank)
        loss = crit(model(xs.to(dev)), ts.to(dev), ws.to(dev))
        opt.zero_grad()
        loss.backward()
        model.flat.collect()
        scale = model.sync_gradients()
        torch.cuda.synchronize()
        if rank == 0:
            out = {"loss": np.array(loss.item()), "world": np.array(world), "scale": np.array(scale)}
            for name, prm in model.module.named_parameters():
                if prm.grad is not None:
                    out["g::" + name] = (prm.grad * scale).detach().cpu().numpy()
            np.savez(sys.argv[1], **out)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    xd, td, wd = x.to(dev), tgt.to(dev), wt.to(dev)
    losses = []
    if one_rank_rccl:
        model.bucket_trace = []
    for _ in range(2):
        loss = crit(model(xd), td, wd)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    flat = model.flat.flat.detach().cpu().numpy()
    if one_rank_rccl:
        import torch.distributed as dist
        assert model._comm_stream is not None and len(model.bucket_trace) >= 2 * len(model.buckets.buckets) > 2
        np.savez(sys.argv[1], flat=flat, losses=np.array(losses), world=world, exchanges=len(model.bucket_trace),
                 hip_streams=1 + len(ops.compute_streams(dev)) + 1)
        dist.destroy_process_group()
        return
    if rank == 0:
        np.savez(sys.argv[1], flat=flat, losses=np.array(losses), world=world)
    if world > 1:
        import torch.distributed as dist
        # every rank must hold the same parameters
        ref = model.flat.flat.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, model.flat.flat), "ranks diverged"
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
