"""Flip-test helpers and the decode affine - drop-in for reference lib/utils/transforms.py:16-118.
numpy entry points keep the reference signatures; *_device variants run on the GPU."""
import numpy as np
import torch

from .. import ops


def flip_back(output_flipped, matched_parts):
    assert output_flipped.ndim == 4, 'output_flipped should be [batch_size, num_joints, height, width]'
    out = output_flipped[:, :, :, ::-1].copy()
    for a, b in matched_parts:
        tmp = out[:, a, :, :].copy()
        out[:, a, :, :] = out[:, b, :, :]
        out[:, b, :, :] = tmp
    return out


def flip_perm(num_joints, matched_parts, device):
    perm = list(range(num_joints))
    for a, b in matched_parts:
        perm[a], perm[b] = b, a
    return torch.tensor(perm, dtype=torch.int32, device=device)


def flip_merge_device(output, output_flipped, matched_parts, shift):
    """(output + shift(flip_back(output_flipped))) * 0.5 in one kernel (core/function.py:226-236)."""
    perm = flip_perm(output.shape[1], matched_parts, output.device)
    return ops.flipback_avg(output.contiguous(), output_flipped.contiguous(), perm, shift)


def fliplr_joints(joints, joints_vis, width, matched_parts):
    joints[:, 0] = width - joints[:, 0] - 1
    for a, b in matched_parts:
        joints[a, :], joints[b, :] = joints[b, :], joints[a, :].copy()
        joints_vis[a, :], joints_vis[b, :] = joints_vis[b, :], joints_vis[a, :].copy()
    return joints * joints_vis, joints_vis


def flip_hm(heatmap, dataset, cond_joints, cond_joints_vis):
    """Condition flip for the flip test (reference 33-58): colored conditions are re-rendered from the
    flipped key points (on the GPU), stacked ones are reversed with left/right channels swapped."""
    matched = dataset.flip_pairs
    if heatmap.shape[1] == 3:
        cj = cond_joints.cpu().numpy().copy()
        cv = cond_joints_vis.cpu().numpy().copy()
        flipped = np.stack([fliplr_joints(cj[i], cv[i], dataset.image_size[0], matched)[0] for i in range(len(cj))])
        jt = torch.from_numpy(np.ascontiguousarray(flipped)).float().to(heatmap.device)
        colors = torch.tensor(dataset.kpt_colors, dtype=torch.float32, device=heatmap.device)
        return ops.cond_render(jt, colors, int(dataset.image_size[1]), int(dataset.image_size[0]))
    if heatmap.shape[1] > 3:
        perm = list(range(heatmap.shape[1]))
        for a, b in matched:
            perm[a], perm[b] = b, a
        return heatmap.flip(3)[:, perm].contiguous()
    return heatmap.flip(3)


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """3-point affine exactly as cv2.getAffineTransform solves it (float64 linear system)."""
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale])
    scale_tmp = scale * 200.0
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    a, b = (dst, src) if inv else (src, dst)
    lhs = np.concatenate([a.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(lhs, b.astype(np.float64)).T


def affine_transform(pt, t):
    return np.dot(t, np.array([pt[0], pt[1], 1.]).T)[:2]


def get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def transform_preds(coords, center, scale, output_size):
    target_coords = np.zeros(coords.shape)
    trans = get_affine_transform(center, scale, 0, output_size, inv=1)
    for p in range(coords.shape[0]):
        target_coords[p, 0:2] = affine_transform(coords[p, 0:2], trans)
    return target_coords
