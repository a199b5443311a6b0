"""Heat-map decoding - drop-in for reference lib/core/inference.py:19-87 (use_dark=False path).

get_max_preds keeps the reference's numpy-in / numpy-out contract for callers that already hold host
arrays; device tensors are decoded by the arg-max kernel (first-index tie break, preds zeroed where
maxval <= 0, quarter-pixel refinement included) so that validate() moves K*(2+1+2) floats per person over
PCIe instead of the whole heat-map.
"""
import numpy as np
import torch

from .. import ops
from ..utils.transforms import transform_preds


def _decode_host(heatmaps):
    """numpy [N,K,H,W] -> (coords [N,K,2] float32 zeroed where the peak is <= 0, peak values [N,K,1])."""
    n, k, _, w = heatmaps.shape
    flat = heatmaps.reshape(n, k, -1)
    where = flat.argmax(axis=2)                        # first index on ties, like np.argmax in the reference
    peak = np.take_along_axis(flat, where[..., None], axis=2)
    coords = np.stack([where % w, where // w], axis=2).astype(np.float32)
    coords *= (peak > 0.0).astype(np.float32)
    return coords, peak


def get_max_preds(batch_heatmaps):
    if isinstance(batch_heatmaps, torch.Tensor):
        assert batch_heatmaps.dim() == 4, 'batch_images should be 4-ndim'
        preds, maxvals, _ = ops.argmax_decode(batch_heatmaps.contiguous())
        return preds.cpu().numpy(), maxvals.cpu().numpy()
    assert isinstance(batch_heatmaps, np.ndarray), 'batch_heatmaps should be numpy.ndarray or a device tensor'
    assert batch_heatmaps.ndim == 4, 'batch_images should be 4-ndim'
    return _decode_host(batch_heatmaps)


def _quarter_offsets_host(heatmaps, coords):
    """POST_PROCESS (reference 68-77): +-0.25 px towards the higher neighbour for peaks strictly inside the map."""
    n, k, hh, hw = heatmaps.shape
    px = np.floor(coords[..., 0] + 0.5).astype(np.int64)
    py = np.floor(coords[..., 1] + 0.5).astype(np.int64)
    inside = (px > 1) & (px < hw - 1) & (py > 1) & (py < hh - 1)
    cx, cy = np.clip(px, 1, hw - 2), np.clip(py, 1, hh - 2)
    bi, ji = np.meshgrid(np.arange(n), np.arange(k), indexing='ij')
    dx = heatmaps[bi, ji, cy, cx + 1] - heatmaps[bi, ji, cy, cx - 1]
    dy = heatmaps[bi, ji, cy + 1, cx] - heatmaps[bi, ji, cy - 1, cx]
    return np.stack([np.sign(dx), np.sign(dy)], axis=2) * 0.25 * inside[..., None]


class DeferredFinalPreds:
    """get_final_preds in two halves for a device tensor: the decode kernel and the copies of its K * 5 floats per person
    into pinned host memory are enqueued here; final_preds(center, scale) waits for them and does the host arithmetic
    (reference inference.py:51-87).  validate() enqueues the next batch's forward between the two halves."""

    def __init__(self, config, batch_heatmaps):
        self.hh, self.hw = batch_heatmaps.shape[2], batch_heatmaps.shape[3]
        self.refine = bool(config.TEST.POST_PROCESS)
        res = ops.argmax_decode(batch_heatmaps.contiguous(), refine=self.refine)
        self.host = []
        for t in (res[0], res[1]) + ((res[3],) if self.refine else ()):
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            self.host.append(h)
        self.event = torch.cuda.Event()
        self.event.record()

    def final_preds(self, center, scale):
        self.event.synchronize()
        coords, maxvals = self.host[0].numpy(), self.host[1].numpy()
        if self.refine:
            coords = coords + self.host[2].numpy()
        preds = np.stack([transform_preds(coords[b], center[b], scale[b], [self.hw, self.hh]) for b in range(coords.shape[0])])
        return preds.astype(coords.dtype), maxvals.copy()


def get_final_preds(config, batch_heatmaps, center, scale, use_dark=False):
    if use_dark:
        raise NotImplementedError("the DARK decoder is dead code in the reference (use_dark=False default)")
    hh, hw = batch_heatmaps.shape[2], batch_heatmaps.shape[3]
    refine = bool(config.TEST.POST_PROCESS)
    if isinstance(batch_heatmaps, torch.Tensor):
        return DeferredFinalPreds(config, batch_heatmaps).final_preds(center, scale)
    else:
        coords, maxvals = get_max_preds(batch_heatmaps)
        if refine:
            coords = coords + _quarter_offsets_host(batch_heatmaps, coords).astype(coords.dtype)
    preds = np.stack([transform_preds(coords[b], center[b], scale[b], [hw, hh]) for b in range(coords.shape[0])])
    return preds.astype(coords.dtype), maxvals
