"""CPU suite (no GPU): host-side logic of buctd_amd - C-ABI export table, configuration node, state_dict contract of
the model mirror, flat parameter arena, and the N > 1 data-parallel path on gloo (world size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from buctd_amd import _C
    header = open(os.path.join(ROOT, "include", "buctd_hip.h")).read()
    declared = set(re.findall(r"\b(buctd_[a-z0-9_]+)\s*\(", header))
    declared -= {"buctd_conv_desc", "buctd_matmul_desc"}
    assert declared == set(_C.SIGNATURES), (declared ^ set(_C.SIGNATURES))
    lib = _C.lib()  # resolves each symbol or raises; no compute call without a GPU
    assert lib.buctd_version() >= 100
    out = subprocess.run(["nm", "-D", "--defined-only", _C.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (buctd_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported


def test_ops_refuse_cpu_tensors():
    from buctd_amd import ops, _C
    x = torch.zeros(1, 4, 4, 16)
    w = torch.zeros(16, 16, 3, 3).contiguous(memory_format=torch.channels_last)
    with pytest.raises(_C.BuctdHipError):
        ops.conv_fwd(x, w, None, 1, 1)


def test_config_node_matches_reference_yaml_workflow(tmp_path):
    from buctd_amd.config import cfg, CfgNode
    c = cfg.clone()
    c.defrost()
    y = tmp_path / "exp.yaml"
    y.write_text("MODEL:\n  NAME: pose_hrnet\n  NUM_JOINTS: 14\n  IMAGE_SIZE:\n  - 288\n  - 384\n  EXTRA:\n"
                 "    FINAL_CONV_KERNEL: 1\n    USE_ATTENTION: false\n    STAGE2:\n      NUM_CHANNELS:\n      - 48\n"
                 "      - 96\nTRAIN:\n  LR: 0.001\nGPUS: (0,1,2,3)\n")
    c.merge_from_file(str(y))
    c.merge_from_list(["MODEL.NAME", "pose_hrnet_coam", "MODEL.EXTRA.USE_ATTENTION", "True", "MODEL.ATT_MODULES",
                       "[False, True, False, False]", "TRAIN.LR", "0.002", "GPUS", "(0,)", "DATASET.COLORED", "True"])
    assert c.MODEL.NAME == "pose_hrnet_coam" and c["MODEL"]["EXTRA"]["USE_ATTENTION"] is True
    assert c.MODEL.ATT_MODULES == [False, True, False, False] and c.GPUS == (0,) and c.TRAIN.LR == 0.002
    assert c.MODEL.EXTRA.STAGE2.NUM_CHANNELS == [48, 96] and c.DATASET.COLORED is True
    with pytest.raises(KeyError):
        c.merge_from_list(["MODEL.NOPE", "1"])
    c.freeze()
    with pytest.raises(AttributeError):
        c.MODEL.NAME = "x"
    assert isinstance(c.clone(), CfgNode)


@pytest.mark.parametrize("name", ["prenet_w16_96x64", "coam_w16_96x64_colored", "coam_w16_96x64_stacked_2heads", "coam_w16_96x64_selfatt",
                                  "transpose_w16_96x64", "resnet18_96x64"])
def test_state_dict_contract_equals_oracle(name):
    from oracle import recipes
    from buctd_amd import models
    cfg, omodel, _, _ = recipes.build(name)
    m = getattr(models, cfg.MODEL.NAME).get_pose_net(cfg, is_train=False)
    a, b = m.state_dict(), omodel.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    m.load_state_dict(b, strict=True)
    # is_train=True applies the reference init: conv / linear weights ~ N(0, 0.001), BN 1/0
    mt = getattr(models, cfg.MODEL.NAME).get_pose_net(cfg, is_train=True)
    assert float(mt.conv1.weight.detach().std()) < 2e-3 and float(mt.bn1.weight.detach().min()) == 1.0
    if cfg.MODEL.NAME != "pose_resnet":
        with pytest.raises(ValueError):
            mt.init_weights("/nonexistent/hrnet.pth")


def test_flat_params_arena_keeps_layout_and_values():
    from buctd_amd import engine, nn as bnn
    net = torch.nn.Sequential(bnn.Conv2d(8, 16, 3, 1, 1), bnn.BatchNorm2d(16), bnn.Linear(5, 7))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    flat = engine.FlatParams(net)
    after = net.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
    w = net[0].weight
    assert w.is_contiguous(memory_format=torch.channels_last)
    s, e = flat.span(w)
    assert torch.equal(flat.flat[s:e].view(16, 3, 3, 8), w.permute(0, 2, 3, 1))  # memory is [Co][R][S][Ci]
    g = flat(w)
    assert g.shape == w.shape and g.stride() == w.stride() and g.data_ptr() == flat.grad[s:].data_ptr()
    assert all(flat.offsets[id(p)] % 4 == 0 for p in flat.params)
    w.grad = torch.ones_like(w)                   # foreign gradient gets adopted by collect()
    flat.collect()
    assert flat.grad_is_arena(w) and float(flat.grad[s:e].sum()) == w.numel()


def _dp_worker(rank, world, port, q, one_rank=False):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from buctd_amd import engine, nn as bnn
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                  # replicas start different: rank 0 must win
    net = torch.nn.Sequential(bnn.Conv2d(4, 8, 3, 1, 1), bnn.BatchNorm2d(8), bnn.Linear(6, 3))
    if one_rank:
        assert engine.reserve_streams(torch.device("cpu"), data_parallel=True) is None
        plain = engine.DataParallel(torch.nn.Sequential(bnn.Linear(6, 3)))
        assert not plain._exchange and plain.sync_gradients() == 1.0          # a group of one is ignored by default
    dp = engine.DataParallel(net, bucket_bytes=64, overlap=False, exchange_in_world_of_one=one_rank)
    assert dp._exchange
    flat = dp.flatten()
    ref = flat.flat.clone()
    gathered = [torch.empty_like(ref) for _ in range(world)]
    dist.all_gather(gathered, ref)
    same_params = all(torch.equal(gathered[0], t) for t in gathered)
    rm = net[1].running_mean.clone()
    # fake backward: each rank writes rank-dependent gradients straight into the arena
    dp._start_step()
    for p in flat.params:
        g = flat(p)
        g.fill_(float(rank + 1))
        p.grad = g
        dp._grad_ready(p)
    scale = dp.sync_gradients()
    expect = sum(range(1, world + 1))
    # every element of every parameter's gradient (the arena's alignment gaps between tensors are not exchanged-for)
    ok = all(bool(torch.all(p.grad * scale == expect / world)) for p in flat.params)
    vals = set(float(v) for p in flat.params for v in (p.grad * scale).flatten()[:1])
    q.put((rank, same_params and ok, float(rm.abs().sum()), len(dp.buckets.buckets), vals, scale))
    dist.destroy_process_group()


def test_data_parallel_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, rm, nb, vals, scale in res:
        assert same, "parameters were not broadcast from rank 0"
        assert nb >= 2, "expected several gradient buckets"
        assert scale == 0.5 and vals == {1.5}, (vals, scale)   # (1 + 2) / 2: mean over the two replicas


def test_exchange_in_a_world_of_one_and_stream_reservation_on_cpu():
    """DataParallel(exchange_in_world_of_one=True): a process group of ONE rank (gloo here, RCCL in test_gpu_ddp.py) runs the
    whole exchange - buckets, collectives, scale 1/1 - and leaves the gradients as they were; without the flag the same
    group is ignored.  engine.reserve_streams is a no-op on a CPU device."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_dp_worker, args=(0, 1, 29650 + (os.getpid() + 7) % 200, q, True))
    p.start()
    rank, same, rm, nb, vals, scale = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same and nb >= 2 and scale == 1.0 and vals == {1.0}, (nb, vals, scale)


def test_host_mirror_numpy_paths_match_reference_golden():
    """The product's host-side (numpy) entry points of core.inference / core.evaluate / utils.transforms against the
    outputs the REFERENCE produced for the same inputs (tests/golden/core.npz)."""
    from buctd_amd.core import evaluate, inference
    from buctd_amd.utils import transforms
    from oracle import core as oc
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "core.npz"))
    preds, maxvals = inference.get_max_preds(g["hm"])
    assert np.array_equal(preds, g["preds"]) and np.array_equal(maxvals, g["maxvals"])

    class Cfg:
        class TEST:
            POST_PROCESS = True
    fp, mv = inference.get_final_preds(Cfg, g["hm"].copy(), g["center"], g["scale"])
    assert np.abs(fp - g["final_preds"]).max() <= 1e-4 and np.array_equal(mv, g["maxvals"])
    acc, avg, cnt, _ = evaluate.accuracy(g["hm"], g["gt"])
    assert np.allclose(acc, g["acc"]) and abs(avg - float(g["avg_acc"])) <= 1e-12 and cnt == int(g["cnt"])
    assert np.array_equal(transforms.flip_back(g["hm"].copy(), oc.CROWDPOSE_FLIP_PAIRS), g["flip_back"])
    # fliplr_joints and the affine against the oracle's restatement (pinned on the reference by make_golden.py)
    rng = np.random.RandomState(3)
    j, v = rng.rand(14, 3) * 200, (rng.rand(14, 1) > 0.3).astype(np.float64).repeat(3, 1)
    a, av = transforms.fliplr_joints(j.copy(), v.copy(), 288, oc.CROWDPOSE_FLIP_PAIRS)
    b, bv = oc.fliplr_joints(j.copy(), v.copy(), 288, oc.CROWDPOSE_FLIP_PAIRS)
    assert np.array_equal(a, b) and np.array_equal(av, bv)
    for rot, inv in ((0, 0), (0, 1), (30, 0), (-40, 1)):
        c, s = np.array([123.4, 77.7], np.float32), np.array([1.3, 1.7], np.float32)
        t = transforms.get_affine_transform(c, s, rot, [72, 96], inv=inv)
        assert np.abs(t - oc.get_affine_transform(c, s, rot, [72, 96], inv=inv)).max() <= 1e-9


def test_validate_shard_merge_gloo_world2(tmp_path):
    """gather_validation_shards (core/function.py): two ranks fill disjoint batches, both end with the full tables."""
    import subprocess
    import sys
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, numpy as np, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from buctd_amd.core.function import gather_validation_shards\n"
        "dist.init_process_group('gloo')\n"
        "r = dist.get_rank()\n"
        "n, k = 10, 3\n"
        "full_p = np.arange(n * k * 3, dtype=np.float32).reshape(n, k, 3) + 0.25\n"
        "full_b = np.arange(n * 7, dtype=np.float64).reshape(n, 7) * 1.5\n"
        "mine = np.array([(i // 2) % 2 == r for i in range(n)])\n"
        "p = np.where(mine[:, None, None], full_p, 0).astype(np.float32)\n"
        "b = np.where(mine[:, None], full_b, 0)\n"
        "paths = [f'img{i}.jpg' if mine[i] else None for i in range(n)]\n"
        "P, B, S = gather_validation_shards(p, b, paths, mine)\n"
        "assert np.array_equal(P, full_p) and np.array_equal(B, full_b) and S == [f'img{i}.jpg' for i in range(n)]\n"
        "print('OK', r)\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and out.stdout.count("OK") == 2, out.stdout + out.stderr


def test_fused_adam_checkpoint_is_torch_adam_format():
    """engine.FusedAdam reads and writes torch.optim.Adam state_dicts (what the reference keeps in
    checkpoint['optimizer'], tools/train.py:243-266), so optimizer state moves between the two in both directions."""
    import copy
    from buctd_amd import engine
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
    twin = copy.deepcopy(net)
    ref = torch.optim.Adam(net.parameters(), lr=1e-3)
    for _ in range(2):
        ref.zero_grad()
        net(torch.randn(2, 3, 8, 8)).square().mean().backward()
        ref.step()
    sd = ref.state_dict()
    fused = engine.FusedAdam(engine.FlatParams(twin), lr=5e-4)
    fused.load_state_dict(copy.deepcopy(sd))
    assert fused.step_count == 2 and fused.param_groups[0]["lr"] == 1e-3
    for i, p in enumerate(fused.flat.params):
        o, e = fused.flat.span(p)
        assert torch.equal(torch.as_strided(fused.exp_avg, p.shape, p.stride(), o), sd["state"][i]["exp_avg"])
        assert torch.equal(torch.as_strided(fused.exp_avg_sq, p.shape, p.stride(), o), sd["state"][i]["exp_avg_sq"])
    back = torch.optim.Adam(copy.deepcopy(net).parameters(), lr=1.0)
    back.load_state_dict(fused.state_dict())            # the reverse direction: torch accepts what FusedAdam writes
    st = back.state_dict()["state"]
    assert all(torch.equal(st[i]["exp_avg"], sd["state"][i]["exp_avg"]) and float(st[i]["step"]) == 2.0 for i in st)
    assert back.param_groups[0]["lr"] == 1e-3
    bad = copy.deepcopy(sd)
    bad["param_groups"][0]["weight_decay"] = 0.1
    with pytest.raises(ValueError):
        fused.load_state_dict(bad)


def test_get_optimizer_branches_like_the_reference():
    """lib/utils/utils.py:258-274: 'sgd' -> SGD(LR, MOMENTUM, WD, NESTEROV), 'adam' -> Adam(LR), anything else -> None."""
    from buctd_amd import engine
    from buctd_amd.config import cfg
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    c = cfg.clone()
    c.defrost()
    c.TRAIN.OPTIMIZER, c.TRAIN.LR = "sgd", 0.01
    opt = engine.get_optimizer(c, net)
    g = opt.param_groups[0]
    assert isinstance(opt, engine.FusedSGD) and (g["lr"], g["momentum"], g["weight_decay"], g["nesterov"]) == (0.01, 0.9, 1e-4, False)
    sd = opt.state_dict()
    assert sd["state"] == {} and sd["param_groups"][0]["dampening"] == 0
    c.TRAIN.OPTIMIZER = "adam"
    assert isinstance(engine.get_optimizer(c, torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3))), engine.FusedAdam)
    c.TRAIN.OPTIMIZER = "rmsprop"
    assert engine.get_optimizer(c, torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3))) is None


def test_position_attention_dropout_mask_statistics():
    """The dropout mask of the fused position attention (tests/helpers/dropout_mask.py mirrors the kernel's counter hash;
    test_gpu_attn_smallqk.py holds the kernel to the mirror bit for bit): kept fraction, correlations between neighbours, between
    whole rows / columns, and the four-point statistic over rectangles (i, j), (i, j'), (i', j), (i', j') - the one the additive
    key sum fails without its finisher (0.07 at p = 0.1, 0.33 at p = 0.5) - all at the level of independent draws."""
    from tests.helpers.dropout_mask import keep_mask, rowkey, colkey
    T = 2048
    for seed, p in ((0x0123456789ABCDEF, 0.1), (0xFEDCBA9876543210, 0.5)):
        k = keep_mask(seed, 1, T, p)[0].astype(np.float64)
        m = k.mean()
        assert abs(m - (1 - p)) < 4 * np.sqrt(p * (1 - p)) / T, (p, m)
        kc, v = k - m, m * (1 - m)
        sig = 1.0 / T                     # std of a correlation estimated from T * T independent pairs
        pairs = [(kc[:, :-1], kc[:, 1:]), (kc[:-1], kc[1:]), (kc[:-1, :-1], kc[1:, 1:]), (kc[:-1, 1:], kc[1:, :-1]),
                 (kc[:, :-2], kc[:, 2:]), (kc[:-2], kc[2:])]
        for a, b in pairs:
            assert abs((a * b).mean() / v) < 5 * sig
        rs = np.random.RandomState(1).randint(0, T, (300, 2))
        rr = np.array([(kc[a] * kc[b]).mean() / v for a, b in rs if a != b])
        cc = np.array([(kc[:, a] * kc[:, b]).mean() / v for a, b in rs if a != b])
        for c in (rr, cc):                # a correlation of two rows: std 1 / sqrt(T)
            assert 0.8 / np.sqrt(T) < c.std() < 1.25 / np.sqrt(T) and np.abs(c).max() < 5 / np.sqrt(T)
        for arr in (k.mean(1), k.mean(0)):  # row / column means: binomial
            assert 0.85 < arr.std() / np.sqrt(v / T) < 1.15
        for di, dj in ((1, 1), (1, 7), (5, 3), (100, 200)):
            q = (kc[:-di, :-dj] * kc[:-di, dj:] * kc[di:, :-dj] * kc[di:, dj:]).mean() / v ** 2
            assert abs(q) < 5 * sig, (p, di, dj, q)
    # the statistic is sensitive: the unfinished key sum fails it by two orders of magnitude
    x = (rowkey(1, np.arange(T))[:, None] + colkey(2, np.arange(T))[None, :]) & np.uint64(0xFFFFFFFF)
    kc = (x >= np.uint64(2 ** 31)).astype(np.float64)
    kc -= kc.mean()
    assert (kc[:-1, :-1] * kc[:-1, 1:] * kc[1:, :-1] * kc[1:, 1:]).mean() / 0.25 ** 2 > 0.2
