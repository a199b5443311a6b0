"""Post-evaluation suppression - drop-in for reference lib/nms/nms.py:17-200 plus the two native extensions it imports
(lib/nms/cpu_nms.pyx -> buctd_cpu_nms, host C++; lib/nms/gpu_nms.pyx + nms_kernel.cu -> buctd_nms, HIP).  Same function
names and return conventions.  None of this runs on the BUCTD path itself (keep = [] there); it serves detector-box
evaluation.  The OKS functions evaluate a whole candidate set per step as one numpy expression
instead of the reference's per-candidate Python loop."""
import ctypes as C

import numpy as np

COCO_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


def py_nms_wrapper(thresh):
    return lambda dets: nms(dets, thresh)


def cpu_nms_wrapper(thresh):
    return lambda dets: cpu_nms(dets, thresh)


def gpu_nms_wrapper(thresh, device_id):
    return lambda dets: gpu_nms(dets, thresh, device_id)


def _iou_one_to_many(box, others):
    """IoU with the +1 pixel convention of box [4] against others [m, 4]."""
    w = np.maximum(0.0, np.minimum(box[2], others[:, 2]) - np.maximum(box[0], others[:, 0]) + 1)
    h = np.maximum(0.0, np.minimum(box[3], others[:, 3]) - np.maximum(box[1], others[:, 1]) + 1)
    inter = w * h
    area = lambda b: (b[..., 2] - b[..., 0] + 1) * (b[..., 3] - b[..., 1] + 1)
    return inter / (area(box) + area(others) - inter)


def nms(dets, thresh):
    """Greedy NMS on [[x1, y1, x2, y2, score]]: walk the boxes by descending score, drop every later box whose IoU with
    a kept one exceeds thresh.  Returns the kept indices (into dets) in score order."""
    if dets.shape[0] == 0:
        return []
    remaining = dets[:, 4].argsort()[::-1]
    keep = []
    while remaining.size:
        head, rest = remaining[0], remaining[1:]
        keep.append(head)
        remaining = rest[_iou_one_to_many(dets[head, :4], dets[rest, :4]) <= thresh]
    return keep


def cpu_nms(dets, thresh):
    """lib/nms/cpu_nms.pyx:20-71 (float32 arithmetic, suppression at IoU >= thresh) through the host C++ entry point."""
    from .._C import check, lib
    d = np.ascontiguousarray(dets, dtype=np.float32)
    n = d.shape[0]
    if n == 0:
        return []
    order = np.ascontiguousarray(d[:, 4].argsort()[::-1], dtype=np.int32)
    keep = np.zeros(n, dtype=np.int32)
    num = C.c_int(0)
    check(lib().buctd_cpu_nms(d.ctypes.data_as(C.c_void_p), n, order.ctypes.data_as(C.c_void_p), float(thresh),
                              keep.ctypes.data_as(C.c_void_p), C.byref(num)), "cpu_nms")
    return [int(i) for i in keep[:num.value]]


def gpu_nms(dets, thresh, device_id=0):
    """lib/nms/gpu_nms.pyx:19-33: sort by score, suppression mask + greedy sweep on the device (buctd_nms)."""
    import torch
    from .. import ops
    from .._C import check, lib, ptr, stream_ptr
    n = dets.shape[0]
    if n == 0:
        return []
    dev = torch.device("cuda", device_id)
    with torch.cuda.device(dev):
        d = torch.as_tensor(np.ascontiguousarray(dets, dtype=np.float32)).to(dev) if not torch.is_tensor(dets) \
            else dets.to(dev, torch.float32).contiguous()
        order = torch.argsort(d[:, 4], descending=True, stable=True)
        boxes = d[order].contiguous()
        keep = torch.empty(n, dtype=torch.int32, device=dev)
        num = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = ops.workspace(lib().buctd_nms_workspace(n), dev)
        check(lib().buctd_nms(ptr(keep), ptr(num), ptr(boxes), n, boxes.shape[1], float(thresh), ptr(ws), ws.numel(),
                              stream_ptr()), "nms")
        k = int(num.item())
        return [int(i) for i in order[keep[:k].long()].cpu()]


def oks_iou(g, d, a_g, a_d, sigmas=None, in_vis_thre=None):
    """Object-keypoint similarity of pose g [3K] against poses d [m, 3K] (areas a_g, a_d [m]).  With in_vis_thre only
    the joints whose score in d exceeds it count (the reference's `list(vg > t) and list(vd > t)` evaluates to the
    second list, nms.py:90 - kept)."""
    if not isinstance(sigmas, np.ndarray):
        sigmas = COCO_SIGMAS
    variances = (sigmas * 2) ** 2
    d = np.asarray(d)
    if d.shape[0] == 0:
        return np.zeros(0)
    dx = d[:, 0::3] - g[0::3]
    dy = d[:, 1::3] - g[1::3]
    scale = (a_g + np.asarray(a_d, dtype=np.float64)) / 2 + np.spacing(1)
    e = (dx ** 2 + dy ** 2) / variances / scale[:, None] / 2
    sim = np.exp(-e)
    if in_vis_thre is None:
        return sim.sum(1) / sim.shape[1]
    counted = d[:, 2::3] > in_vis_thre
    n = counted.sum(1)
    return np.where(n > 0, (sim * counted).sum(1) / np.maximum(n, 1), 0.0)


def _db_arrays(kpts_db):
    scores = np.array([p['score'] for p in kpts_db])
    kpts = np.array([p['keypoints'].flatten() for p in kpts_db])
    areas = np.array([p['area'] for p in kpts_db])
    return scores, kpts, areas


def oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None):
    """Greedy suppression with OKS as the overlap: keep the best-scoring pose, drop poses with OKS > thresh to it."""
    if len(kpts_db) == 0:
        return []
    scores, kpts, areas = _db_arrays(kpts_db)
    remaining = scores.argsort()[::-1]
    keep = []
    while remaining.size:
        head, rest = remaining[0], remaining[1:]
        keep.append(head)
        remaining = rest[oks_iou(kpts[head], kpts[rest], areas[head], areas[rest], sigmas, in_vis_thre) <= thresh]
    return keep


def oks_merge(kpts_db_mode0, kpts_db_mode1, min_oks_thres=0.5, sigmas=None, in_vis_thre=None):
    """Append to mode-1 detections every mode-0 pose whose best OKS against the ORIGINAL mode-1 set is <= min_oks_thres
    (the returned list is kpts_db_mode1 itself, extended in place, like the reference)."""
    if len(kpts_db_mode1) == 0:
        return kpts_db_mode0
    _, k0, a0 = _db_arrays(kpts_db_mode0)
    _, k1, a1 = _db_arrays(kpts_db_mode1)
    for i in range(len(k0)):
        if oks_iou(k0[i], k1, a0[i], a1, sigmas, in_vis_thre).max() <= min_oks_thres:
            kpts_db_mode1.append(kpts_db_mode0[i])
    return kpts_db_mode1


def rescore(overlap, scores, thresh, type='gaussian'):
    assert overlap.shape[0] == scores.shape[0]
    if type == 'linear':
        hit = overlap >= thresh
        scores[hit] = scores[hit] * (1 - overlap[hit])
        return scores
    return scores * np.exp(-overlap ** 2 / thresh)


def soft_oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None):
    """Soft suppression: the scores of the remaining poses decay with their OKS to the pose just taken; at most 20
    poses are returned."""
    if len(kpts_db) == 0:
        return []
    scores, kpts, areas = _db_arrays(kpts_db)
    order = scores.argsort()[::-1]
    scores = scores[order]
    limit = 20
    keep = []
    while order.size and len(keep) < limit:
        head, order = order[0], order[1:]
        decayed = rescore(oks_iou(kpts[head], kpts[order], areas[head], areas[order], sigmas, in_vis_thre), scores[1:],
                          thresh)
        rank = decayed.argsort()[::-1]
        order, scores = order[rank], decayed[rank]
        keep.append(head)
    return np.array(keep, dtype=np.intp)
