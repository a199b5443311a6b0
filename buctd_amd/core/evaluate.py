"""PCK-style training accuracy - drop-in for reference lib/core/evaluate.py:15-70.
Accepts numpy arrays (reference contract) or device tensors (decoded on the GPU first)."""
import numpy as np

from .inference import get_max_preds


def calc_dists(preds, target, normalize):
    preds = preds.astype(np.float32)
    target = target.astype(np.float32)
    dists = np.zeros((preds.shape[1], preds.shape[0]))
    valid = (target[:, :, 0] > 1) & (target[:, :, 1] > 1)
    d = np.linalg.norm(preds / normalize[:, None, :] - target / normalize[:, None, :], axis=2)
    dists[:] = np.where(valid, d, -1).T
    return dists


def dist_acc(dists, thr=0.5):
    dist_cal = np.not_equal(dists, -1)
    num = dist_cal.sum()
    return np.less(dists[dist_cal], thr).sum() * 1.0 / num if num > 0 else -1


def accuracy(output, target, hm_type='gaussian', thr=0.5):
    idx = list(range(output.shape[1]))
    norm = 1.0
    if hm_type == 'gaussian':
        pred, _ = get_max_preds(output)
        target, _ = get_max_preds(target)
        h, w = output.shape[2], output.shape[3]
        norm = np.ones((pred.shape[0], 2)) * np.array([h, w]) / 10  # (x / (h/10), y / (w/10)): reference quirk kept
    dists = calc_dists(pred, target, norm)
    acc = np.zeros((len(idx) + 1))
    avg_acc, cnt = 0, 0
    for i in range(len(idx)):
        acc[i + 1] = dist_acc(dists[idx[i]])
        if acc[i + 1] >= 0:
            avg_acc += acc[i + 1]
            cnt += 1
    avg_acc = avg_acc / cnt if cnt != 0 else 0
    if cnt != 0:
        acc[0] = avg_acc
    return acc, avg_acc, cnt, pred
