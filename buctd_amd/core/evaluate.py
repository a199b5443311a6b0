"""PCK-style training accuracy - drop-in for reference lib/core/evaluate.py:15-70 (same names, same numbers).

The reference walks joints and samples in Python loops; here the whole [N, K] distance table is one numpy
expression.  Inputs are numpy heat-maps (reference contract) or device tensors (decoded by the arg-max kernel).
"""
import numpy as np

from .inference import get_max_preds


def calc_dists(preds, target, normalize):
    """[K, N] table of normalised L2 distances; -1 where the target joint is not a usable ground truth
    (the reference's test is  target_x > 1 and target_y > 1, evaluate.py:21)."""
    p = np.asarray(preds, dtype=np.float32)
    g = np.asarray(target, dtype=np.float32)
    scale = np.asarray(normalize, dtype=np.float64)[:, None, :]
    gap = p / scale - g / scale
    usable = np.logical_and(g[..., 0] > 1, g[..., 1] > 1)
    table = np.full(usable.shape, -1.0)
    table[usable] = np.sqrt((gap[usable] ** 2).sum(-1))
    return table.T


def dist_acc(dists, thr=0.5):
    """Fraction of usable entries (!= -1) below thr; -1 when there is none."""
    usable = dists != -1
    total = int(usable.sum())
    if total == 0:
        return -1
    return float((dists[usable] < thr).sum()) / total


def pck_from_coords(pred, gt, hm_h, hm_w, thr=0.5):
    """accuracy() on already decoded arg-max coordinates [N, K, 2].  The normaliser is (H/10, W/10) applied to
    (x, y) in that order - x is divided by H/10: a quirk of the reference (evaluate.py:52-55) that is kept."""
    n, k = pred.shape[0], pred.shape[1]
    normalize = np.tile(np.array([hm_h, hm_w], dtype=np.float64) / 10.0, (n, 1))
    table = calc_dists(pred, gt, normalize)
    per_joint = np.array([dist_acc(table[j], thr) for j in range(k)], dtype=np.float64)
    scored = per_joint >= 0
    cnt = int(scored.sum())
    avg = float(per_joint[scored].sum() / cnt) if cnt else 0
    acc = np.zeros(k + 1)
    acc[1:] = per_joint
    acc[0] = avg
    return acc, avg, cnt, pred


def accuracy(output, target, hm_type='gaussian', thr=0.5):
    """Returns (acc[K+1], avg_acc, cnt, pred) like the reference: acc[0] is the mean over joints that had at least
    one usable sample, acc[1:] the per-joint values (-1 = no usable sample)."""
    if hm_type != 'gaussian':
        raise ValueError("accuracy(): only 'gaussian' heat-map targets are on the BUCTD path")
    pred, _ = get_max_preds(output)
    gt, _ = get_max_preds(target)
    return pck_from_coords(pred, gt, output.shape[2], output.shape[3], thr)
