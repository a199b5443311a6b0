"""Heat-map decoding - drop-in for reference lib/core/inference.py:19-87 (use_dark=False path).

get_max_preds keeps the reference's numpy-in / numpy-out contract for callers that already hold host
arrays; device tensors are decoded by the arg-max kernel (first-index tie break, preds zeroed where
maxval <= 0) so that validate() moves K*3 floats per person over PCIe instead of the whole heat-map.
"""
import math

import numpy as np
import torch

from .. import ops
from ..utils.transforms import transform_preds


def get_max_preds(batch_heatmaps):
    if isinstance(batch_heatmaps, torch.Tensor):
        assert batch_heatmaps.dim() == 4, 'batch_images should be 4-ndim'
        preds, maxvals, _ = ops.argmax_decode(batch_heatmaps.contiguous())
        return preds.cpu().numpy(), maxvals.cpu().numpy()
    assert isinstance(batch_heatmaps, np.ndarray), 'batch_heatmaps should be numpy.ndarray or a device tensor'
    assert batch_heatmaps.ndim == 4, 'batch_images should be 4-ndim'
    n, k, _, w = batch_heatmaps.shape
    flat = batch_heatmaps.reshape((n, k, -1))
    idx = np.argmax(flat, 2).reshape((n, k, 1))
    maxvals = np.amax(flat, 2).reshape((n, k, 1))
    preds = np.tile(idx, (1, 1, 2)).astype(np.float32)
    preds[:, :, 0] = preds[:, :, 0] % w
    preds[:, :, 1] = np.floor(preds[:, :, 1] / w)
    preds *= np.tile(np.greater(maxvals, 0.0), (1, 1, 2)).astype(np.float32)
    return preds, maxvals


def get_final_preds(config, batch_heatmaps, center, scale, use_dark=False):
    if use_dark:
        raise NotImplementedError("the DARK decoder is dead code in the reference (use_dark=False default)")
    coords, maxvals = get_max_preds(batch_heatmaps)
    hh, hw = batch_heatmaps.shape[2], batch_heatmaps.shape[3]
    if config.TEST.POST_PROCESS:
        hm = batch_heatmaps.detach().cpu().numpy() if isinstance(batch_heatmaps, torch.Tensor) else batch_heatmaps
        for n in range(coords.shape[0]):
            for p in range(coords.shape[1]):
                px = int(math.floor(coords[n][p][0] + 0.5))
                py = int(math.floor(coords[n][p][1] + 0.5))
                if 1 < px < hw - 1 and 1 < py < hh - 1:
                    h = hm[n][p]
                    diff = np.array([h[py][px + 1] - h[py][px - 1], h[py + 1][px] - h[py - 1][px]])
                    coords[n][p] += np.sign(diff) * .25
    preds = coords.copy()
    for i in range(coords.shape[0]):
        preds[i] = transform_preds(coords[i], center[i], scale[i], [hw, hh])
    return preds, maxvals
