"""The a3-a7 kernels against the REFERENCE's own outputs (tests/golden/core.npz, target.npz, written by
oracle/make_golden.py from the imported reference functions) - not against in-test restatements:
JointsMSELoss (core/loss.py:23-41), get_max_preds / get_final_preds (core/inference.py:19-87), accuracy
(core/evaluate.py:40-70), flip_back + the flip-test merge (utils/transforms.py:16-37, core/function.py:226-236),
generate_target (dataset/JointsDataset.py:397-453)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CROWDPOSE_FLIP_PAIRS = [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11]]


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def core():
    return np.load(os.path.join(GOLD, "core.npz"))


def test_joints_mse_matches_reference_loss_and_gradient(dev, core):
    from buctd_amd import ops
    from buctd_amd.core.loss import JointsMSELoss
    pred, gt, wt = (torch.from_numpy(core[k]).to(dev) for k in ("pred", "gt", "wt"))
    loss, grad = ops.joints_mse(pred, gt, wt, True)
    assert abs(loss.item() - float(core["loss"])) <= 2e-7 * max(1.0, abs(float(core["loss"])))
    assert np.abs(grad.cpu().numpy() - core["loss_grad"]).max() <= 1e-9
    # the module (autograd) form the training loop uses
    p = pred.clone().requires_grad_(True)
    out = JointsMSELoss(True)(p, gt, wt)
    out.backward()
    assert abs(out.item() - float(core["loss"])) <= 2e-7 * max(1.0, abs(float(core["loss"])))
    assert np.abs(p.grad.cpu().numpy() - core["loss_grad"]).max() <= 1e-9


def test_decode_matches_reference_get_max_preds_and_final_preds(dev, core):
    from buctd_amd import ops
    from buctd_amd.core.inference import get_final_preds, get_max_preds
    hm = torch.from_numpy(core["hm"]).to(dev)
    preds, maxvals, idx = ops.argmax_decode(hm)
    assert np.array_equal(preds.cpu().numpy(), core["preds"])          # ties, non-positive rows, borders: exact
    assert np.array_equal(maxvals.cpu().numpy(), core["maxvals"])
    flat = core["hm"].reshape(core["hm"].shape[0], core["hm"].shape[1], -1)
    assert np.array_equal(idx.cpu().numpy(), flat.argmax(2))
    hp, hv = get_max_preds(hm)                                          # host-mirror entry point, device tensor in
    assert np.array_equal(np.asarray(hp), core["preds"]) and np.array_equal(np.asarray(hv), core["maxvals"])

    class Cfg:
        class TEST:
            POST_PROCESS = True
    fp, fm = get_final_preds(Cfg, hm, core["center"], core["scale"])
    assert np.abs(fp - core["final_preds"]).max() <= 1e-4              # image-pixel coordinates up to ~400, fp32 output
    assert np.array_equal(fm, core["maxvals"])


def test_accuracy_matches_reference(dev, core):
    from buctd_amd.core.evaluate import accuracy
    acc, avg, cnt, pred = accuracy(torch.from_numpy(core["hm"]).to(dev), torch.from_numpy(core["gt"]).to(dev))
    assert np.allclose(np.asarray(acc, dtype=np.float64), core["acc"], atol=1e-12)
    assert float(avg) == float(core["avg_acc"]) and int(cnt) == int(core["cnt"])
    assert np.array_equal(np.asarray(pred), core["preds"])


def test_flip_back_and_merge_match_reference(dev, core):
    from buctd_amd import ops
    K = core["hm"].shape[1]
    perm = list(range(K))
    for a, b in CROWDPOSE_FLIP_PAIRS:
        perm[a], perm[b] = b, a
    permd = torch.tensor(perm, dtype=torch.int32, device=dev)
    hm = torch.from_numpy(core["hm"]).to(dev)
    # 0.5 * (0 + flip_back(hm)) * 2 == the reference's flip_back (exact: halving and doubling are exact)
    fb = ops.flipback_avg(torch.zeros_like(hm), hm, permd, False) * 2.0
    assert np.array_equal(fb.cpu().numpy(), core["flip_back"])
    # the validate() merge with SHIFT_HEATMAP: 0.5 * (hm + shift(flip_back(hm reversed over the batch)))
    merged = ops.flipback_avg(hm, torch.from_numpy(core["hm"][::-1].copy()).to(dev), permd, True)
    assert np.abs(merged.cpu().numpy() - core["merged"]).max() <= 1e-7


@pytest.mark.parametrize("tag", ["crowdpose", "coco256"])
def test_gaussian_target_matches_reference_generate_target(dev, tag):
    from buctd_amd import ops
    g = np.load(os.path.join(GOLD, "target.npz"))
    hw0, hw1, iw0, iw1, sig, k = (int(v) for v in g[f"{tag}_meta"])
    joints = torch.from_numpy(g[f"{tag}_joints"]).float()[None].to(dev)
    vis = torch.from_numpy(g[f"{tag}_vis"][:, 0].copy())[None].to(dev)
    t, w = ops.gaussian_target(joints, vis, (hw0, hw1), (iw0, iw1), sig)
    assert np.array_equal(w[0].cpu().numpy(), g[f"{tag}_weight"])     # incl. the joint outside the map: weight zeroed
    assert np.abs(t[0].cpu().numpy() - g[f"{tag}_target"]).max() <= 2e-7
    assert t[0, 1].abs().max().item() == 0 and t[0, 2].max().item() > 0
