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
