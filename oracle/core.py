"""TEST INFRASTRUCTURE - CPU oracle of the non-network pieces of the BUCTD hot path.
numpy / plain torch restatements, each citing the reference lines it follows.  Pinned by
oracle/make_golden.py against the imported reference (core.loss, core.inference, core.evaluate,
utils.transforms) and against the executed body of JointsDataset.generate_target.
"""
import math

import numpy as np
import torch
import torch.nn as nn


class JointsMSELoss(nn.Module):
    """reference lib/core/loss.py:17-41."""

    def __init__(self, use_target_weight):
        super().__init__()
        self.criterion = nn.MSELoss(reduction="mean")
        self.use_target_weight = use_target_weight

    def forward(self, output, target, target_weight):
        b, k = output.size(0), output.size(1)
        pred = output.reshape((b, k, -1)).split(1, 1)
        gt = target.reshape((b, k, -1)).split(1, 1)
        loss = 0
        for j in range(k):
            p, g = pred[j].squeeze(), gt[j].squeeze()
            if self.use_target_weight:
                loss += 0.5 * self.criterion(p.mul(target_weight[:, j]), g.mul(target_weight[:, j]))
            else:
                loss += 0.5 * self.criterion(p, g)
        return loss / k


def joints_mse_closed_form(output, target, target_weight):
    """L = 0.5/(K*N*HW) * sum w^2 (p-g)^2; used to cross-check the module."""
    n, k = output.shape[:2]
    d = (output - target).reshape(n, k, -1)
    w2 = (target_weight.reshape(n, k, 1) ** 2) if target_weight is not None else 1.0
    return 0.5 * (w2 * d * d).sum() / (k * n * d.shape[2])


def get_max_preds(batch_heatmaps):
    """reference lib/core/inference.py:19-47 (numpy in, numpy out)."""
    assert isinstance(batch_heatmaps, np.ndarray) and batch_heatmaps.ndim == 4
    n, k, _, w = batch_heatmaps.shape
    flat = batch_heatmaps.reshape((n, k, -1))
    idx = np.argmax(flat, 2).reshape((n, k, 1))
    maxvals = np.amax(flat, 2).reshape((n, k, 1))
    preds = np.tile(idx, (1, 1, 2)).astype(np.float32)
    preds[:, :, 0] = preds[:, :, 0] % w
    preds[:, :, 1] = np.floor(preds[:, :, 1] / w)
    preds *= np.tile(np.greater(maxvals, 0.0), (1, 1, 2)).astype(np.float32)
    return preds, maxvals


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """reference lib/utils/transforms.py:86-118 with cv2.getAffineTransform replaced by the exact
    3-point linear solve it performs."""
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale])
    scale_tmp = scale * 200.0
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    sp = [0, src_w * -0.5]
    src_dir = [sp[0] * cs - sp[1] * sn, sp[0] * sn + sp[1] * cs]
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir

    def third(a, b):
        d = a - b
        return b + np.array([-d[1], d[0]], dtype=np.float32)

    src[2:, :] = third(src[0, :], src[1, :])
    dst[2:, :] = third(dst[0, :], dst[1, :])
    a, b = (dst, src) if inv else (src, dst)
    A = np.concatenate([a.astype(np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(A, b.astype(np.float64)).T  # 2x3


def affine_transform(pt, t):
    """reference lib/utils/transforms.py:121-124."""
    return np.dot(t, np.array([pt[0], pt[1], 1.]).T)[:2]


def transform_preds(coords, center, scale, output_size):
    """reference lib/utils/transforms.py:78-83."""
    out = np.zeros(coords.shape)
    t = get_affine_transform(center, scale, 0, output_size, inv=1)
    for p in range(coords.shape[0]):
        out[p, 0:2] = np.dot(t, np.array([coords[p, 0], coords[p, 1], 1.0]).T)[:2]
    return out


def get_final_preds(post_process, batch_heatmaps, center, scale):
    """reference lib/core/inference.py:51-87 (use_dark=False path)."""
    coords, maxvals = get_max_preds(batch_heatmaps)
    hh, hw = batch_heatmaps.shape[2], batch_heatmaps.shape[3]
    if post_process:
        for n in range(coords.shape[0]):
            for p in range(coords.shape[1]):
                hm = batch_heatmaps[n][p]
                px = int(math.floor(coords[n][p][0] + 0.5))
                py = int(math.floor(coords[n][p][1] + 0.5))
                if 1 < px < hw - 1 and 1 < py < hh - 1:
                    diff = np.array([hm[py][px + 1] - hm[py][px - 1], hm[py + 1][px] - hm[py - 1][px]])
                    coords[n][p] += np.sign(diff) * .25
    preds = coords.copy()
    for i in range(coords.shape[0]):
        preds[i] = transform_preds(coords[i], center[i], scale[i], [hw, hh])
    return preds, maxvals


def accuracy(output, target, thr=0.5):
    """reference lib/core/evaluate.py:15-70 (hm_type='gaussian')."""
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    h, w = output.shape[2], output.shape[3]
    norm = np.ones((pred.shape[0], 2)) * np.array([h, w]) / 10
    n, k = pred.shape[:2]
    dists = np.zeros((k, n))
    for i in range(n):
        for c in range(k):
            if tgt[i, c, 0] > 1 and tgt[i, c, 1] > 1:
                dists[c, i] = np.linalg.norm(pred[i, c, :].astype(np.float32) / norm[i] -
                                             tgt[i, c, :].astype(np.float32) / norm[i])
            else:
                dists[c, i] = -1
    acc = np.zeros(k + 1)
    avg, cnt = 0, 0
    for c in range(k):
        valid = np.not_equal(dists[c], -1)
        nv = valid.sum()
        acc[c + 1] = np.less(dists[c][valid], thr).sum() * 1.0 / nv if nv > 0 else -1
        if acc[c + 1] >= 0:
            avg += acc[c + 1]
            cnt += 1
    avg = avg / cnt if cnt != 0 else 0
    if cnt != 0:
        acc[0] = avg
    return acc, avg, cnt, pred


def flip_back(output_flipped, matched_parts):
    """reference lib/utils/transforms.py:16-30."""
    assert output_flipped.ndim == 4
    out = output_flipped[:, :, :, ::-1].copy()
    for a, b in matched_parts:
        tmp = out[:, a].copy()
        out[:, a] = out[:, b]
        out[:, b] = tmp
    return out


def fliplr_joints(joints, joints_vis, width, matched_parts):
    """reference lib/utils/transforms.py:61-75."""
    joints = joints.copy()
    joints_vis = joints_vis.copy()
    joints[:, 0] = width - joints[:, 0] - 1
    for a, b in matched_parts:
        joints[a, :], joints[b, :] = joints[b, :], joints[a, :].copy()
        joints_vis[a, :], joints_vis[b, :] = joints_vis[b, :], joints_vis[a, :].copy()
    return joints * joints_vis, joints_vis


def flip_test_merge(output, output_flipped, flip_pairs, shift):
    """reference lib/core/function.py:226-236 on numpy heat-maps."""
    of = flip_back(output_flipped, flip_pairs)
    if shift:
        sh = of.copy()
        sh[:, :, :, 1:] = of[:, :, :, 0:-1]
        of = sh
    return (output + of) * 0.5


def generate_target(joints, joints_vis, num_joints, heatmap_size, image_size, sigma, joints_weight=None):
    """reference lib/dataset/JointsDataset.py:397-453. heatmap_size / image_size are (W, H)."""
    heatmap_size = np.asarray(heatmap_size)
    image_size = np.asarray(image_size)
    target_weight = np.ones((num_joints, 1), dtype=np.float32)
    target_weight[:, 0] = joints_vis[:, 0]
    target = np.zeros((num_joints, heatmap_size[1], heatmap_size[0]), dtype=np.float32)
    tmp = sigma * 3
    for j in range(num_joints):
        stride = image_size / heatmap_size
        mu_x = int(joints[j][0] / stride[0] + 0.5)
        mu_y = int(joints[j][1] / stride[1] + 0.5)
        ul = [int(mu_x - tmp), int(mu_y - tmp)]
        br = [int(mu_x + tmp + 1), int(mu_y + tmp + 1)]
        if ul[0] >= heatmap_size[0] or ul[1] >= heatmap_size[1] or br[0] < 0 or br[1] < 0:
            target_weight[j] = 0
            continue
        size = 2 * tmp + 1
        x = np.arange(0, size, 1, np.float32)
        y = x[:, np.newaxis]
        x0 = y0 = size // 2
        g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
        g_x = max(0, -ul[0]), min(br[0], heatmap_size[0]) - ul[0]
        g_y = max(0, -ul[1]), min(br[1], heatmap_size[1]) - ul[1]
        img_x = max(0, ul[0]), min(br[0], heatmap_size[0])
        img_y = max(0, ul[1]), min(br[1], heatmap_size[1])
        if target_weight[j] > 0.5:
            target[j][img_y[0]:img_y[1], img_x[0]:img_x[1]] = g[g_y[0]:g_y[1], g_x[0]:g_x[1]]
    if joints_weight is not None:
        target_weight = np.multiply(target_weight, joints_weight)
    return target, target_weight


def gaussian_kernel_1d(ksize=15, sigma=0.0):
    """cv2.getGaussianKernel: sigma<=0 -> 0.3*((ksize-1)*0.5-1)+0.8; taps normalised to sum 1."""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2
    k = np.exp(-(x * x) / (2 * sigma * sigma))
    return k / k.sum()


def gaussian_blur_reflect101(img, ksize=15):
    """cv2.GaussianBlur(img, (15,15), 0) restated: separable, BORDER_REFLECT_101, float64.
    (cv2 is absent in this image - SURVEY 8c: pinned by hand-derived vectors in tests.)"""
    k = gaussian_kernel_1d(ksize)
    r = ksize // 2
    out = img.astype(np.float64)
    for axis in (0, 1):
        pad = [(0, 0)] * out.ndim
        pad[axis] = (r, r)
        p = np.pad(out, pad, mode="reflect")
        acc = np.zeros_like(out)
        for t in range(ksize):
            sl = [slice(None)] * out.ndim
            sl[axis] = slice(t, t + out.shape[axis])
            acc += k[t] * p[tuple(sl)]
        out = acc
    return out


def generate_heatmap(heatmap):
    """reference lib/dataset/JointsDataset.py:457-463."""
    heatmap = gaussian_blur_reflect101(heatmap)
    am = np.amax(heatmap)
    if am == 0:
        return heatmap
    heatmap /= am / 255
    return heatmap


def get_condition_image_colored(kpts, size, colors):
    """reference lib/dataset/JointsDataset.py:519-543. size = (H, W, 3) -> (H, W, 3) float64."""
    kpts = np.array(kpts).astype(int)
    z = np.zeros(size)
    for color, kpt in zip(colors, kpts):
        if 0 < kpt[0] < size[1] and 0 < kpt[1] < size[0]:
            z[kpt[1] - 1][kpt[0] - 1] = color
    return generate_heatmap(z)


def get_stacked_condition(kpts, size):
    """reference lib/dataset/JointsDataset.py:471-498. size = (H, W) -> (H, W, K) float64: one blurred impulse per joint,
    every channel peak-normalised on its own (zero channel for a joint outside the crop)."""
    kpts = np.array(kpts).astype(int)
    chans = []
    for kpt in kpts:
        z = np.zeros(size)
        if 0 < kpt[0] < size[1] and 0 < kpt[1] < size[0]:
            z[kpt[1] - 1][kpt[0] - 1] = 255
        chans.append(generate_heatmap(z))
    return np.moveaxis(np.array(chans), 0, -1)


def get_condition_image(kpts, size):
    """reference lib/dataset/JointsDataset.py:500-516. size = (H, W) -> (3, H, W) int."""
    kpts = np.array(kpts).astype(int)
    z = np.zeros(size)
    for kpt in kpts:
        if 0 < kpt[0] < size[1] and 0 < kpt[1] < size[0]:
            z[kpt[1] - 1][kpt[0] - 1] = 255
    h = np.expand_dims(generate_heatmap(z), axis=0)
    return np.repeat(h, 3, axis=0).astype(int)


CROWDPOSE_FLIP_PAIRS = [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11]]  # dataset/crowdpose.py:49-50
COCO_FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]  # dataset/coco.py:49-50
CROWDPOSE_KPT_COLORS = [[245, 53, 53], [245, 125, 45], [253, 206, 20], [206, 244, 54], [118, 253, 27],
                        [47, 254, 47], [25, 245, 113], [15, 243, 197], [14, 199, 245], [44, 126, 249],
                        [13, 13, 249], [128, 47, 249], [205, 38, 247], [245, 48, 206]]  # dataset/crowdpose.py:56
