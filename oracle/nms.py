"""TEST INFRASTRUCTURE - loop-level restatement of reference lib/nms/nms.py (nms 35-72, oks_iou 75-94, oks_nms 97-124,
soft_oks_nms 161-200, rescore 150-158), cpu_nms (lib/nms/cpu_nms.pyx:20-71) and the device NMS semantics
(lib/nms/nms_kernel.cu:23-77 mask + 123-139 sweep on score-sorted boxes).  Pinned by oracle/make_golden.py, which imports
the reference's nms.py (with the two Cython extension modules stubbed) and stores its outputs in tests/golden/nms.npz."""
import numpy as np

_SIG = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


def _iou(a, b):
    w = max(0.0, min(a[2], b[2]) - max(a[0], b[0]) + 1)
    h = max(0.0, min(a[3], b[3]) - max(a[1], b[1]) + 1)
    inter = w * h
    return inter / ((a[2] - a[0] + 1) * (a[3] - a[1] + 1) + (b[2] - b[0] + 1) * (b[3] - b[1] + 1) - inter)


def nms(dets, thresh, strict=True):
    """strict=True: suppress IoU > thresh (nms.py, nms_kernel.cu); False: >= thresh in float32 (cpu_nms.pyx)."""
    if dets.shape[0] == 0:
        return []
    d = dets if strict else dets.astype(np.float32)
    order = list(d[:, 4].argsort()[::-1])
    dead = set()
    keep = []
    for a, i in enumerate(order):
        if i in dead:
            continue
        keep.append(int(i))
        for j in order[a + 1:]:
            if j in dead:
                continue
            if strict:
                ov = _iou(d[i], d[j])
                if ov > thresh:
                    dead.add(j)
            else:
                f = np.float32
                w = max(f(0), f(min(d[i, 2], d[j, 2]) - max(d[i, 0], d[j, 0]) + f(1)))
                h = max(f(0), f(min(d[i, 3], d[j, 3]) - max(d[i, 1], d[j, 1]) + f(1)))
                inter = f(w * h)
                ai = f((d[i, 2] - d[i, 0] + f(1)) * (d[i, 3] - d[i, 1] + f(1)))
                aj = f((d[j, 2] - d[j, 0] + f(1)) * (d[j, 3] - d[j, 1] + f(1)))
                if f(inter / f(f(ai + aj) - inter)) >= f(thresh):
                    dead.add(j)
    return keep


def oks_iou(g, d, a_g, a_d, sigmas=None, in_vis_thre=None):
    sig = sigmas if isinstance(sigmas, np.ndarray) else _SIG
    var = (sig * 2) ** 2
    out = np.zeros(d.shape[0])
    for n in range(d.shape[0]):
        e = ((d[n, 0::3] - g[0::3]) ** 2 + (d[n, 1::3] - g[1::3]) ** 2) / var / ((a_g + a_d[n]) / 2 + np.spacing(1)) / 2
        if in_vis_thre is not None:
            e = e[d[n, 2::3] > in_vis_thre]     # `list(vg > t) and list(vd > t)` == the second list
        out[n] = np.sum(np.exp(-e)) / e.shape[0] if e.shape[0] != 0 else 0.0
    return out


def oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None):
    if len(kpts_db) == 0:
        return []
    scores = np.array([p['score'] for p in kpts_db])
    kpts = np.array([p['keypoints'].flatten() for p in kpts_db])
    areas = np.array([p['area'] for p in kpts_db])
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        ov = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]], sigmas, in_vis_thre)
        order = order[np.where(ov <= thresh)[0] + 1]
    return keep


def soft_oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None):
    if len(kpts_db) == 0:
        return []
    scores = np.array([p['score'] for p in kpts_db])
    kpts = np.array([p['keypoints'].flatten() for p in kpts_db])
    areas = np.array([p['area'] for p in kpts_db])
    order = scores.argsort()[::-1]
    scores = scores[order]
    keep = []
    while order.size > 0 and len(keep) < 20:
        i = order[0]
        ov = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]], sigmas, in_vis_thre)
        order = order[1:]
        scores = scores[1:] * np.exp(-ov ** 2 / thresh)
        tmp = scores.argsort()[::-1]
        order, scores = order[tmp], scores[tmp]
        keep.append(i)
    return np.array(keep, dtype=np.intp)


def make_boxes(n, seed, size=400.0):
    """Seeded detections with many overlaps: [n, 5] float32 (x1, y1, x2, y2, score), distinct scores."""
    rng = np.random.RandomState(seed)
    c = rng.rand(n, 2) * size
    wh = rng.rand(n, 2) * 80 + 20
    s = rng.permutation(n).astype(np.float32) / n + 0.001
    return np.concatenate([c - wh / 2, c + wh / 2, s[:, None]], 1).astype(np.float32)


def make_poses(n, k, seed, size=300.0):
    rng = np.random.RandomState(seed)
    base = rng.rand(max(n // 3, 1), k, 2) * size
    db = []
    for i in range(n):
        kp = base[i % base.shape[0]] + rng.randn(k, 2) * (1.0 + 4.0 * (i % 4))
        sc = rng.rand(k, 1)
        db.append({'keypoints': np.concatenate([kp, sc], 1), 'score': float(rng.rand()) + 0.01 * i,
                   'area': float(3000 + 500 * rng.rand())})
    return db
