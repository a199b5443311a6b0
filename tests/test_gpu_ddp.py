"""engine.DataParallel with real device tensors: two ranks (both on GPU 0, gloo - RCCL refuses two ranks per device)
train the same batch for two steps; the averaged gradient of two identical replicas is the single-process gradient,
so the parameters must come out bit-identical to a plain one-process run.  Exercises bucketed all-reduce launched from
the backward callbacks, the communication stream, the weight-gradient / branch streams and FusedAdam's gradient sync."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "ddp_worker.py")


def _env(**kw):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


@pytest.mark.parametrize("math", ["bf16x6", "fp32"])
def test_two_ranks_match_one_process(tmp_path, math):
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    r = subprocess.run([sys.executable, WORKER, one], env=_env(BUCTD_CONV_MATH=math), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", WORKER, two],
                       env=_env(BUCTD_CONV_MATH=math, BUCTD_DIST_BACKEND="gloo", BUCTD_SINGLE_DEVICE="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert int(b["world"]) == 2
    assert np.array_equal(a["losses"], b["losses"])
    assert np.array_equal(a["flat"], b["flat"]), f"max diff {np.abs(a['flat'] - b['flat']).max():.3e}"


def test_two_ranks_on_different_shards_average_like_two_oracle_replicas(tmp_path):
    """Rank-DEPENDENT data (the identical-batch test above also passes if the exchange is a no-op with a compensating
    scale): each rank runs its own shard, and the gradient rank 0 holds after the exchange must be the mean of the two
    replicas' gradients as the oracle computes them one replica at a time (per-replica BatchNorm statistics, like
    nn.DataParallel in tools/train.py:147) - and must NOT be either replica's own gradient."""
    import copy
    import torch
    from oracle import recipes, core as oc
    two = str(tmp_path / "shards.npz")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29537", WORKER, two],
                       env=_env(BUCTD_CONV_MATH="bf16x6", BUCTD_DIST_BACKEND="gloo", BUCTD_SINGLE_DEVICE="1",
                                BUCTD_DDP_MODE="shards"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(two)
    assert int(got["world"]) == 2 and float(got["scale"]) == 0.5
    cfg, omodel, x, _ = recipes.build("coam_w16_96x64_colored")
    grads = []
    for rank in range(2):
        m = copy.deepcopy(omodel).double().train()
        recipes.set_dropout(m, 0.0)
        xs, js = recipes.make_inputs(cfg, x.shape[0], 900 + rank, 3)
        ts, ws = recipes.make_targets(cfg, js, 950 + rank)
        loss = oc.JointsMSELoss(True)(m(xs.double()), ts.double(), ws.double())
        loss.backward()
        grads.append({k: p.grad.numpy() for k, p in m.named_parameters() if p.grad is not None})
    rel_avg, rel_own = [], []
    gmax = max(np.linalg.norm(0.5 * (grads[0][k] + grads[1][k])) for k in grads[0])
    for k in grads[0]:
        g = got["g::" + k].astype(np.float64)
        avg = 0.5 * (grads[0][k] + grads[1][k])
        nrm = np.linalg.norm(avg)
        if nrm <= 1e-6 * gmax:      # mathematically-zero gradients (a conv bias in front of a BatchNorm): round-off only
            assert np.linalg.norm(g) <= 1e-4 * gmax, k
            continue
        rel_avg.append(np.linalg.norm(g - avg) / nrm)
        rel_own.append(np.linalg.norm(g - grads[0][k]) / nrm)
    rel_avg, rel_own = np.array(rel_avg), np.array(rel_own)
    # fp32-class arithmetic against the fp64 two-replica mean (the bar of the whole-network train-step tests) ...
    assert np.median(rel_avg) <= 2e-3 and rel_avg.max() <= 2e-2, (np.median(rel_avg), rel_avg.max())
    # ... while rank 0's own gradient is far away: the other replica's half really arrived
    assert np.median(rel_own) >= 0.2, np.median(rel_own)


def test_two_ranks_over_rccl_when_two_devices_are_visible(tmp_path):
    """The same check through the production transport: backend "nccl" (= RCCL over xGMI), one rank per GPU.  Needs two
    visible devices (the round-end 1-GPU box skips it; the 8-GPU scaling node runs it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank: fewer than two GPUs visible")
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    r = subprocess.run([sys.executable, WORKER, one], env=_env(BUCTD_CONV_MATH="bf16x6"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", WORKER, two],
                       env=_env(BUCTD_CONV_MATH="bf16x6", BUCTD_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert int(b["world"]) == 2
    # the sum of two identical fp32 gradients halved is exact, so RCCL's reduction order cannot show either
    assert np.array_equal(a["losses"], b["losses"]) and np.array_equal(a["flat"], b["flat"])


def test_one_rank_over_rccl_equals_the_plain_engine(tmp_path):
    """The production transport on a single-GPU box: a process group of ONE rank over backend "nccl" (RCCL), with
    DataParallel(exchange_in_world_of_one=True) - flat-parameter broadcast, bucketed all-reduces launched from the backward
    callbacks on the high-priority communication stream, the stream budget assertion, FusedAdam's gradient sync.  The sum over one
    rank is the identity and the scale is 1, so parameters and losses must equal the plain one-process run bit for bit."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "rccl1.npz")
    r = subprocess.run([sys.executable, WORKER, one], env=_env(BUCTD_CONV_MATH="bf16x6"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, WORKER, two],
                       env=_env(BUCTD_CONV_MATH="bf16x6", BUCTD_DDP_MODE="rccl1", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert int(b["exchanges"]) > 2 and int(b["hip_streams"]) <= 4
    assert np.array_equal(a["losses"], b["losses"]) and np.array_equal(a["flat"], b["flat"])


def test_validate_is_sharded_over_ranks(tmp_path):
    """core.function.validate under a two-rank launch (both ranks on GPU 0, gloo): every rank runs half of the batches,
    rank 0's evaluate() sees the complete, correctly ordered tables - identical to a one-process run."""
    script = tmp_path / "val_worker.py"
    script.write_text(
        "import os, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from buctd_amd import engine, models\n"
        "from buctd_amd.core.function import validate\n"
        "from buctd_amd.core.loss import JointsMSELoss\n"
        "from oracle import recipes\n"
        "rank, world, dev = engine.init_distributed()\n"
        "cfg, omodel, x, joints = recipes.build('coam_w16_96x64_colored')\n"
        "cfg.TEST.FLIP_TEST = False\n"
        "cfg.PRINT_FREQ = 100\n"
        "cfg.DEBUG = type('D', (), {'DEBUG': False})()\n"
        "net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)\n"
        "net.load_state_dict(omodel.state_dict(), strict=True)\n"
        "net = net.to(dev).eval()\n"
        "tgt, wt = recipes.make_targets(cfg, joints, 77)\n"
        "nb = 4\n"
        "class DS:\n"
        "    flip_pairs = []\n"
        "    def __len__(self): return nb * x.shape[0]\n"
        "    def evaluate(self, cfg, preds, out_dir, boxes, paths, *a):\n"
        "        np.savez(sys.argv[1], preds=preds, boxes=boxes, paths=np.array(paths))\n"
        "        return {'AP': float(preds[:, :, 2].mean())}, float(preds[:, :, 2].mean())\n"
        "def loader():\n"
        "    for i in range(nb):\n"
        "        n = x.shape[0]\n"
        "        meta = {'center': torch.full((n, 2), 40.0 + i), 'scale': torch.full((n, 2), 0.5 + 0.1 * i),\n"
        "                'score': torch.full((n,), 0.9), 'annotation_id': torch.arange(n) + 10 * i,\n"
        "                'image': [f'img_{i}_{j}.jpg' for j in range(n)]}\n"
        "        yield x * (1.0 + 0.01 * i), tgt, wt, meta\n"
        "class L:\n"
        "    def __iter__(self): return loader()\n"
        "    def __len__(self): return nb\n"
        "perf = validate(cfg, L(), DS(), net, JointsMSELoss(True), str(sys.argv[2]), str(sys.argv[2]))\n"
        "print('PERF', rank, perf)\n"
        "if world > 1:\n"
        "    import torch.distributed as dist\n"
        "    dist.barrier(); dist.destroy_process_group()\n")
    outs = []
    # OMP_NUM_THREADS=1 in both launches (torchrun's default for its workers): the recipe calibrates the oracle's BatchNorm
    # statistics on the CPU, and torch's parallel reductions round differently with the thread count
    for tag, cmd, extra in (("one", [sys.executable, str(script)], {"OMP_NUM_THREADS": "1"}),
                            ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                     "--master-addr", "127.0.0.1", "--master-port", "29535", str(script)],
                             {"BUCTD_DIST_BACKEND": "gloo", "BUCTD_SINGLE_DEVICE": "1", "OMP_NUM_THREADS": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run(cmd + [out, str(tmp_path)], env=_env(**extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
        outs.append((np.load(out), r.stdout))
    (a, _), (b, log) = outs
    assert np.array_equal(a["preds"], b["preds"]) and np.array_equal(a["boxes"], b["boxes"])
    assert list(a["paths"]) == list(b["paths"])
    perfs = [float(l.split()[2]) for l in log.splitlines() if l.startswith("PERF")]
    assert len(perfs) == 2 and perfs[0] == perfs[1], "perf_indicator is broadcast to every rank"


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (how the driver calls it) starts its two ranks itself and prints ONE JSON line from rank 0
    with n_gpus = 2, the max-over-ranks step time and the exposed gradient-exchange time.  On a 1-GPU box both ranks sit on
    GPU 0 over gloo (BUCTD_SINGLE_DEVICE / BUCTD_DIST_BACKEND are test hooks of engine.init_distributed)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "train_c2", "--batch", "4", "--no-kernel-timer"],
                       env=_env(BUCTD_DIST_BACKEND="gloo", BUCTD_SINGLE_DEVICE="1"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["global_batch"] == 8 and out["allreduce_exposed_ms_per_step"] >= 0
    _check_bucket_rows(out)


def _check_bucket_rows(out):
    """the per-bucket exchange timeline of the --gpus N line: every bucket once, exchanges begin after the backward pass
    started, in launch order on the communication stream, and their sizes add up to the gradient arena"""
    b = out["allreduce_buckets"]
    rows = b["rank0"]
    assert b["backward_ms"] > 0 and len(rows) >= 2
    assert sorted(r["bucket"] for r in rows) == list(range(len(rows)))
    assert all(r["end_ms"] >= r["start_ms"] >= 0 for r in rows)
    assert all(rows[i]["start_ms"] <= rows[i + 1]["start_ms"] + 1e-3 for i in range(len(rows) - 1))
    # overlap: a bucket's exchange is launched from the backward callbacks as soon as its last gradient kernel is enqueued, so
    # every bucket but the last (the stem's parameters, final when the backward pass is) starts INSIDE the backward pass
    assert all(r["start_ms"] < b["backward_ms"] for r in rows[:-1]), (b["backward_ms"], [r["start_ms"] for r in rows])
    assert rows[0]["start_ms"] < 0.75 * b["backward_ms"]
    # HIP stream budget under --gpus N: main + one branch stream + weight-gradient stream + communication stream
    assert out["config"].get("hip_streams", 4) <= 4
    assert abs(sum(r["mb"] for r in rows) * 2 ** 20 - 4 * out["config"]["params"]) <= 0.02 * 4 * out["config"]["params"] + 2 ** 20


def test_bench_with_four_ranks_on_one_device(tmp_path):
    """`python bench.py --gpus 4`: four ranks (all on GPU 0 over gloo on a 1-GPU box) - the launch, rendezvous, bucketed
    exchange from the backward callbacks and the max-over-ranks clock of the 4-GPU point of the scaling curve."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--workload", "train_c2", "--batch", "2", "--no-kernel-timer"],
                       env=_env(BUCTD_DIST_BACKEND="gloo", BUCTD_SINGLE_DEVICE="1"), capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp4"
    assert out["value"] > 0 and out["allreduce_exposed_ms_per_step"] >= 0
    _check_bucket_rows(out)
