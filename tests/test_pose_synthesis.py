"""generative pose synthesis.  CPU: the counter-RNG restatement (oracle/pose_synthesis.py) reproduces
the error-type frequencies the REFERENCE function produced on the same scene (tests/golden/pose_synthesis.npz, made
by oracle/make_golden.py::pose_synthesis_case with the reference imported).  GPU: the HIP kernel reproduces the oracle
sample by sample (same generator) and, over many seeds, the same frequencies."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_synthesis.npz")


def _freq(counts):
    return counts / counts.sum(1, keepdims=True)


@pytest.mark.parametrize("dataset", ["crowdpose", "coco"])
def test_oracle_distribution_matches_reference_golden(dataset):
    from oracle import pose_synthesis as P
    g = np.load(GOLD)
    ref, runs = g[f"{dataset}_ref_counts"], int(g[f"{dataset}_runs"])
    joints, est, near, area = P.make_scene(dataset, 7)
    K, n = joints.shape[0], 250
    cnt = np.zeros((K, 6))
    for it in range(n):
        q = P.synthesize_pose(dataset, joints, est, near, area, 0, seed=50_000 + it)
        for j in range(K):
            cnt[j, 5 if not q[j, :2].any() else P.classify(q[j], joints, est, near, area, dataset, j)] += 1
        assert np.all(q[:, 2] == (1.0 if dataset == "coco" else 0.0))
    fr, fo = _freq(ref.astype(float)), cnt / n
    tol = 4.5 * np.sqrt(np.maximum(fr * (1 - fr), 2e-3) * (1 / n + 1 / runs)) + 0.01
    assert not (np.abs(fr - fo) > tol).any(), np.argwhere(np.abs(fr - fo) > tol)
    assert fr[:, 0].min() > 0.4 and fr[:, 4].max() > 0.02       # the scene exercises good and miss outcomes


@pytest.mark.gpu
@pytest.mark.parametrize("dataset", ["crowdpose", "coco"])
def test_hip_synthesis_matches_oracle_and_reference_distribution(dev, dataset):
    from oracle import pose_synthesis as P
    from buctd_amd.dataset import pose_synthesis as D
    joints, est, near, area = P.make_scene(dataset, 7)
    K = joints.shape[0]
    # sample-by-sample against the CPU twin: a batch of persons = the same scene under different person indices,
    # plus variants without neighbours / with few annotated joints / crowded
    B = 6
    J, E, N = np.stack([joints] * B), np.stack([est] * B), np.stack([near] * B)
    J[4, 6:, 2] = 0                                       # few valid joints -> other probability rows
    N[5, :, :, 2] = 0                                     # no neighbours
    ov = np.array([0, 0, 0, 3, 1, 0])
    out = D.synthesize_pose_batch(dataset, J, E, N, [area] * B, ov, seed=4242).cpu().numpy()
    for b in range(B):
        ref = P.synthesize_pose(dataset, J[b], E[b], N[b], area, int(ov[b]), seed=4242, person=b)
        assert np.abs(out[b] - ref).max() <= 1e-6, f"person {b}: HIP differs from the oracle by {np.abs(out[b] - ref).max()}"
    # reference signature, one person
    class Cfg:
        class MODEL:
            NUM_JOINTS = K
        class DATASET:
            DATASET = dataset
    one = D.synthesize_pose(Cfg, joints, est, near, area, 0, seed=77)
    assert np.abs(one - P.synthesize_pose(dataset, joints, est, near, area, 0, seed=77)).max() <= 1e-6
    # distribution over 4000 persons in one launch vs the reference's class frequencies
    g = np.load(GOLD)
    ref, runs = g[f"{dataset}_ref_counts"].astype(float), int(g[f"{dataset}_runs"])
    n = 4000
    big = D.synthesize_pose_batch(dataset, np.stack([joints] * n), np.stack([est] * n), np.stack([near] * n),
                                  [area] * n, [0] * n, seed=9).cpu().numpy()
    cnt = np.zeros((K, 6))
    for b in range(n):
        for j in range(K):
            cnt[j, 5 if not big[b, j, :2].any() else P.classify(big[b, j], joints, est, near, area, dataset, j)] += 1
    fr, fo = ref / ref.sum(1, keepdims=True), cnt / n
    tol = 4.5 * np.sqrt(np.maximum(fr * (1 - fr), 1e-3) * (1 / n + 1 / runs)) + 0.004
    assert not (np.abs(fr - fo) > tol).any(), (np.argwhere(np.abs(fr - fo) > tol), fr, fo)
