"""Flip-test helpers and the crop <-> image affine - drop-in for reference lib/utils/transforms.py:16-118.
numpy entry points keep the reference names and signatures; *_device variants run on the GPU."""
import numpy as np
import torch

from .. import ops


def _swap_table(count, matched_parts):
    """Channel permutation that exchanges every left/right pair."""
    table = np.arange(count)
    for left, right in matched_parts:
        table[left], table[right] = right, left
    return table


def flip_back(output_flipped, matched_parts):
    """Undo a horizontal flip on heat-maps [N, K, H, W]: mirror the width axis, exchange left/right joints."""
    if output_flipped.ndim != 4:
        raise AssertionError('output_flipped should be [batch_size, num_joints, height, width]')
    table = _swap_table(output_flipped.shape[1], matched_parts)
    return np.ascontiguousarray(output_flipped[:, table, :, ::-1])


def flip_perm(num_joints, matched_parts, device):
    return torch.as_tensor(_swap_table(num_joints, matched_parts), dtype=torch.int32, device=device)


def flip_merge_device(output, output_flipped, matched_parts, shift):
    """(output + shift(flip_back(output_flipped))) * 0.5 in one kernel (reference core/function.py:226-236)."""
    perm = flip_perm(output.shape[1], matched_parts, output.device)
    return ops.flipback_avg(output.contiguous(), output_flipped.contiguous(), perm, shift)


def fliplr_joints(joints, joints_vis, width, matched_parts):
    """Mirror key points [K, 3] about the vertical axis of a `width`-pixel image (in place, like the reference)
    and exchange left/right rows; invisible joints come back zeroed."""
    joints[:, 0] = width - joints[:, 0] - 1
    table = _swap_table(joints.shape[0], matched_parts)
    joints[:] = joints[table]
    joints_vis[:] = joints_vis[table]
    return joints * joints_vis, joints_vis


def flip_hm(heatmap, dataset, cond_joints, cond_joints_vis):
    """Condition flip for the flip test (reference 33-58): a 3-channel condition is re-rendered (colored) from the
    mirrored key points - on the GPU here; a stacked one is mirrored with left/right channels exchanged."""
    channels = heatmap.shape[1]
    if channels == 3:
        width, height = int(dataset.image_size[0]), int(dataset.image_size[1])
        kp = cond_joints.detach().cpu().numpy().copy()
        vis = cond_joints_vis.detach().cpu().numpy().copy()
        mirrored = [fliplr_joints(kp[b], vis[b], width, dataset.flip_pairs)[0] for b in range(kp.shape[0])]
        pts = torch.from_numpy(np.ascontiguousarray(np.stack(mirrored))).float().to(heatmap.device)
        colors = torch.tensor(dataset.kpt_colors, dtype=torch.float32, device=heatmap.device)
        return ops.cond_render(pts, colors, height, width)
    mirrored = heatmap.flip(3)
    if channels > 3:
        table = torch.as_tensor(_swap_table(channels, dataset.flip_pairs), device=heatmap.device)
        mirrored = mirrored.index_select(1, table)
    return mirrored.contiguous()


# ---------------------------------------------------------------------------------------------- affine ----
def get_dir(src_point, rot_rad):
    """src_point rotated by rot_rad (counter-clockwise in image coordinates with y down)."""
    s, c = np.sin(rot_rad), np.cos(rot_rad)
    x, y = src_point[0], src_point[1]
    return [x * c - y * s, x * s + y * c]


def get_3rd_point(a, b):
    """Third corner of the right-angled isosceles triangle on the segment a-b (float32 like the reference)."""
    dx, dy = a[0] - b[0], a[1] - b[1]
    return b + np.array([-dy, dx], dtype=np.float32)


def _triangle(origin, arm):
    """The three float32 control points the reference feeds to cv2.getAffineTransform."""
    pts = np.zeros((3, 2), dtype=np.float32)
    pts[0] = origin
    pts[1] = origin + arm
    pts[2] = get_3rd_point(pts[0], pts[1])
    return pts


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """2x3 matrix mapping the person box (center, scale*200 px, rotated by rot degrees) onto an output_size crop
    (inv=1: the reverse).  cv2.getAffineTransform solves the 3-point system in float64; so does this."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale])
    box = np.asarray(scale) * 200.0
    out_w, out_h = output_size[0], output_size[1]
    offset = box * shift
    box_pts = _triangle(center + offset, np.asarray(get_dir([0, box[0] * -0.5], np.pi * rot / 180)))
    crop_pts = _triangle(np.array([out_w * 0.5, out_h * 0.5]), np.array([0, out_w * -0.5], np.float32))
    frm, to = (crop_pts, box_pts) if inv else (box_pts, crop_pts)
    system = np.hstack([frm.astype(np.float64), np.ones((3, 1))])
    return np.linalg.solve(system, to.astype(np.float64)).T


def affine_transform(pt, t):
    return t[:, :2] @ np.array([pt[0], pt[1]], dtype=np.float64) + t[:, 2]


def transform_preds(coords, center, scale, output_size):
    """Heat-map coordinates [K, >=2] -> image coordinates through the inverse crop affine (rot = 0)."""
    t = get_affine_transform(center, scale, 0, output_size, inv=1)
    mapped = np.zeros(coords.shape)
    mapped[:, 0:2] = coords[:, 0:2].astype(np.float64) @ t[:, :2].T + t[:, 2]
    return mapped
