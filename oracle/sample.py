"""TEST INFRASTRUCTURE - CPU restatement of the per-sample pipeline of reference lib/dataset/JointsDataset.py:134-361
(SURVEY 8f row f1): person box -> augmentation -> affine crop (cv2.warpAffine, INTER_LINEAR) -> ToTensor/Normalize ->
key points into crop coordinates -> Gaussian target + condition heat-map.

cv2 is not installed in this image, so cv2.warpAffine cannot be called to pin the restatement ("parity unpinned" for
the crop itself): warp_affine_u8 restates OpenCV's fixed-point bilinear warp (imgproc/src/imgwarp.cpp: warpAffine
-> WarpAffineInvoker -> remapBilinear<FixedPtCast<int, uchar, 15>>) - matrix inverted in double, source coordinates in
1/1024 px rounded to 1/32 px, weights (32-i)(32-j)*32 of 2^15, result (sum + 2^14) >> 15, BORDER_CONSTANT 0 - and
tests/test_sample_pipeline.py pins it with hand-derived vectors (identity, integer and half-pixel shifts, x2 zoom).
Everything else is pinned against the reference through oracle/core.py (affine, fliplr_joints, generate_target)."""
import numpy as np

from . import core as ocore

AB_BITS, INTER_BITS = 10, 5
AB_SCALE = 1 << AB_BITS
ROUND_DELTA = AB_SCALE // (1 << INTER_BITS) // 2


def invert_affine(m):
    """cv2.warpAffine without WARP_INVERSE_MAP: M (src -> dst) is inverted in double precision first."""
    m = np.array(m, dtype=np.float64).copy()
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = a11, m[0, 1] * -d, m[1, 0] * -d, a22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    return m


def _sat_int(v):      # saturate_cast<int>(double) = cvRound: round half to even
    return np.clip(np.rint(v), -2147483648, 2147483647).astype(np.int64)


def warp_affine_u8(src, m, dsize, flip_src=False, keep_rect=None):
    """src uint8 [H, W, C]; m 2x3 (src -> dst); dsize (w, h).  flip_src: the source is mirrored horizontally first
    (JointsDataset.py:244); keep_rect (x, y, w, h): source pixels outside are zero (NEW_AUGMENTATION, 272-285)."""
    sh, sw = src.shape[:2]
    w, h = int(dsize[0]), int(dsize[1])
    im = invert_affine(m)
    xs, ys = np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64)
    adelta = _sat_int(im[0, 0] * xs * AB_SCALE)
    bdelta = _sat_int(im[1, 0] * xs * AB_SCALE)
    x0 = _sat_int((im[0, 1] * ys + im[0, 2]) * AB_SCALE) + ROUND_DELTA
    y0 = _sat_int((im[1, 1] * ys + im[1, 2]) * AB_SCALE) + ROUND_DELTA
    X = (x0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx, fy = X & 31, Y & 31
    img = src[:, ::-1] if flip_src else src
    if keep_rect is not None:
        rx, ry, rw, rh = (int(v) for v in keep_rect)
        masked = np.zeros_like(img)
        masked[max(ry, 0):ry + rh, max(rx, 0):rx + rw] = img[max(ry, 0):ry + rh, max(rx, 0):rx + rw]
        img = masked
    pad = np.zeros((sh + 2, sw + 2, img.shape[2]), dtype=np.int64)     # BORDER_CONSTANT 0
    pad[1:-1, 1:-1] = img

    def at(yy, xx):
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        return pad[np.where(ok, yy + 1, 0), np.where(ok, xx + 1, 0)]

    w00, w01 = ((32 - fy) * (32 - fx) * 32)[..., None], ((32 - fy) * fx * 32)[..., None]
    w10, w11 = (fy * (32 - fx) * 32)[..., None], (fy * fx * 32)[..., None]
    acc = at(sy, sx) * w00 + at(sy, sx + 1) * w01 + at(sy + 1, sx) * w10 + at(sy + 1, sx + 1) * w11
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def to_tensor_normalize(img_u8, mean, std):
    """torchvision ToTensor + Normalize: HWC uint8 -> CHW float32, (v / 255 - mean) / std in float32."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return (x - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]


def xywh2cs(x, y, w, h, aspect_ratio, scale_thre, pixel_std=200):
    """dataloader.py:305-321"""
    center = np.zeros(2, dtype=np.float32)
    center[0] = x + w * 0.5
    center[1] = y + h * 0.5
    if w > aspect_ratio * h:
        h = w * 1.0 / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w * 1.0 / pixel_std, h * 1.0 / pixel_std], dtype=np.float32)
    if center[0] != -1:
        scale = scale * scale_thre
    return center, scale


def box_from_keypoints(kp, margin, img_w, img_h):
    """bounding box of the non-zero key-point coordinates +- margin, clipped to the image
    (JointsDataset.py:217-226, dataloader.py:482-491)."""
    xs, ys = kp[:, 0][np.nonzero(kp[:, 0])], kp[:, 1][np.nonzero(kp[:, 1])]
    xmin, ymin = np.clip(xs.min() - margin, 0, img_w), np.clip(ys.min() - margin, 0, img_h)
    xmax, ymax = np.clip(xs.max() + margin, 0, img_w), np.clip(ys.max() + margin, 0, img_h)
    return [xmin, ymin, xmax - xmin, ymax - ymin]


def make_sample(image_u8, joints, joints_vis, cond_joints, cond_joints_vis, center, scale, rot, flip, image_size,
                heatmap_size, sigma, flip_pairs, mean, std, colors=None, mono=False, stacked=False):
    """One sample after the random draws (center / scale / rot / flip are the post-augmentation values of
    JointsDataset.py:233-251): returns input [3+3, H, W] float32, target, target_weight, joints, cond_joints (crop)."""
    joints, joints_vis = joints.copy(), joints_vis.copy()
    cond_joints, cond_joints_vis = cond_joints.copy(), cond_joints_vis.copy()
    center = np.array(center, dtype=np.float32).copy()
    if flip:
        w_img = image_u8.shape[1]
        joints, joints_vis = ocore.fliplr_joints(joints, joints_vis, w_img, flip_pairs)
        center[0] = w_img - center[0] - 1
        cond_joints, cond_joints_vis = ocore.fliplr_joints(cond_joints, cond_joints_vis, w_img, flip_pairs)
    trans = ocore.get_affine_transform(center, scale, rot, image_size)
    crop = warp_affine_u8(image_u8, trans, image_size, flip_src=flip)
    x = to_tensor_normalize(crop, mean, std)
    for i in range(joints.shape[0]):
        if joints_vis[i, 0] > 0.0:
            joints[i, 0:2] = ocore.affine_transform(joints[i, 0:2], trans)
        if cond_joints_vis[i, 0] > 0.0:
            cond_joints[i, 0:2] = ocore.affine_transform(cond_joints[i, 0:2], trans)
    target, weight = ocore.generate_target(joints, joints_vis, joints.shape[0], heatmap_size, image_size, sigma)
    h, w = int(image_size[1]), int(image_size[0])
    if stacked:
        cond = ocore.get_stacked_condition(cond_joints[:, :2], (h, w)).transpose(2, 0, 1).astype(np.float32)
    elif mono:
        cond = ocore.get_condition_image(cond_joints, (h, w)).astype(np.float32)
    else:
        cond = ocore.get_condition_image_colored(cond_joints, (h, w, 3), colors).transpose(2, 0, 1).astype(np.float32)
    return np.concatenate([x, cond], 0), target, weight, joints, cond_joints, crop
