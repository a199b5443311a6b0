"""GPU-side sample pipeline and in-process iterative refinement.

DeviceSamplePipeline does what the DataLoader workers of the reference do per sample in
lib/dataset/JointsDataset.py:134-361 - box -> (half-body / scale / rotation / flip augmentation) -> affine crop ->
ToTensor/Normalize -> key points into crop coordinates -> Gaussian target + condition heat-map - for a whole batch:
the scalar geometry (a few dozen float64 operations per person, the same expressions as the reference) stays on the host,
every per-pixel step is one batched HIP kernel writing straight into the NCHW network input, the target and the
target weights.  Decoded images are uint8 HWC tensors already resident on the device (JPEG decoding is host I/O).

IterativeRefiner chains conditional top-down passes without touching disk: the reference runs tools/test.py three times
and hands the predictions over through the results json (scripts/test/test_BUCTD_COAM_gen_sample.sh:21,
lib/dataset/dataloader.py:454-508): prediction k -> box from its key points (+ margin, clipped) -> center / scale ->
new crop + re-rendered condition -> prediction k+1, with the rescoring of dataloader.py:596-612.
"""
import ctypes as C
import math
import random

import numpy as np
import torch

from .. import ops
from .._C import check, lib, ptr, stream_ptr
from ..core.inference import get_final_preds
from ..utils.transforms import affine_transform, fliplr_joints, get_affine_transform


class _WarpItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("H", C.c_int), ("W", C.c_int), ("flip", C.c_int), ("rx", C.c_int),
                ("ry", C.c_int), ("rw", C.c_int), ("rh", C.c_int), ("m", C.c_double * 6)]


def xywh2cs(x, y, w, h, aspect_ratio, scale_thre, pixel_std=200):
    """Box -> (center, scale) with the network's aspect ratio (reference dataloader.py:305-321)."""
    center = np.array([x + w * 0.5, y + h * 0.5], dtype=np.float32)
    if w > aspect_ratio * h:
        h = w * 1.0 / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w * 1.0 / pixel_std, h * 1.0 / pixel_std], dtype=np.float32)
    if center[0] != -1:
        scale = scale * scale_thre
    return center, scale


def box_from_keypoints(kp, margin, img_w, img_h):
    """Box of the non-zero key-point coordinates +- margin, clipped to the image (JointsDataset.py:217-226)."""
    xs, ys = kp[:, 0][np.nonzero(kp[:, 0])], kp[:, 1][np.nonzero(kp[:, 1])]
    xmin, ymin = np.clip(xs.min() - margin, 0, img_w), np.clip(ys.min() - margin, 0, img_h)
    xmax, ymax = np.clip(xs.max() + margin, 0, img_w), np.clip(ys.max() + margin, 0, img_h)
    return [xmin, ymin, xmax - xmin, ymax - ymin]


class DeviceSamplePipeline:
    def __init__(self, cfg, flip_pairs=(), upper_body_ids=(), kpt_colors=None, mean=(0.485, 0.456, 0.406),
                 std=(0.229, 0.224, 0.225), is_train=False, seed=0):
        self.cfg = cfg
        self.is_train = is_train
        self.num_joints = cfg.MODEL.NUM_JOINTS
        self.image_size = np.array(cfg.MODEL.IMAGE_SIZE)
        self.heatmap_size = np.array(cfg.MODEL.HEATMAP_SIZE)
        self.sigma = cfg.MODEL.SIGMA
        self.aspect_ratio = self.image_size[0] * 1.0 / self.image_size[1]
        ds = cfg.DATASET
        self.colored = bool(ds.COLORED)
        self.stacked = bool(ds.STACKED_CONDITION)
        self.conditional = bool(cfg.MODEL.CONDITIONAL_TOPDOWN)
        self.scale_factor = getattr(ds, "SCALE_FACTOR", 0.25)
        self.rotation_factor = getattr(ds, "ROT_FACTOR", 30)
        self.flip = bool(getattr(ds, "FLIP", True))
        self.num_joints_half_body = getattr(ds, "NUM_JOINTS_HALF_BODY", 8)
        self.prob_half_body = getattr(ds, "PROB_HALF_BODY", 0.0)
        self.bu_bbox_margin = getattr(ds, "BU_BBOX_MARGIN", 25)
        self.scale_thre = getattr(cfg.TEST, "SCALE_THRE", 1.25)
        self.flip_pairs = [list(p) for p in flip_pairs]
        self.upper_body_ids = set(upper_body_ids)
        self.kpt_colors = None if kpt_colors is None else np.asarray(kpt_colors, dtype=np.float32)
        self.mean = np.asarray(mean, dtype=np.float32)
        self.std = np.asarray(std, dtype=np.float32)
        self.np_rng = np.random.RandomState(seed)
        self.py_rng = random.Random(seed)

    # ---- host-side scalar geometry (reference expressions, float64) -----------------------------------------
    def half_body_transform(self, joints, joints_vis):
        """JointsDataset.py:90-133: box around the visible upper- or lower-body joints."""
        upper = [joints[i] for i in range(self.num_joints) if joints_vis[i][0] > 0 and i in self.upper_body_ids]
        lower = [joints[i] for i in range(self.num_joints) if joints_vis[i][0] > 0 and i not in self.upper_body_ids]
        if self.np_rng.randn() < 0.5 and len(upper) > 2:
            chosen = upper
        else:
            chosen = lower if len(lower) > 2 else upper
        if len(chosen) < 2:
            return None, None
        pts = np.array(chosen, dtype=np.float32)
        center = pts.mean(axis=0)[:2]
        lo, hi = np.amin(pts, axis=0), np.amax(pts, axis=0)
        w, h = hi[0] - lo[0], hi[1] - lo[1]
        if w > self.aspect_ratio * h:
            h = w * 1.0 / self.aspect_ratio
        elif w < self.aspect_ratio * h:
            w = h * self.aspect_ratio
        return center, np.array([w * 1.0 / 200, h * 1.0 / 200], dtype=np.float32) * 1.5

    def draw_augmentation(self, rec, center, scale):
        """The random part of JointsDataset.py:233-251 (same draws in the same order): returns center, scale, rot, flip."""
        rot, flip = 0, False
        if not self.is_train:
            return center, scale, rot, flip
        if np.sum(rec["joints_3d_vis"][:, 0]) > self.num_joints_half_body and self.np_rng.rand() < self.prob_half_body:
            c_hb, s_hb = self.half_body_transform(rec["joints_3d"], rec["joints_3d_vis"])
            if c_hb is not None and s_hb is not None:
                center, scale = c_hb, s_hb
        sf, rf = self.scale_factor, self.rotation_factor
        scale = scale * np.clip(self.np_rng.randn() * sf + 1, 1 - sf, 1 + sf)
        rot = np.clip(self.np_rng.randn() * rf, -rf * 2, rf * 2) if self.py_rng.random() <= 0.6 else 0
        flip = self.flip and self.py_rng.random() <= 0.5
        return center, scale, rot, flip

    def geometry(self, rec, aug=None):
        """Everything of a sample that is scalar: returns dict(trans, center, scale, rot, flip, joints, joints_vis,
        cond_joints, cond_joints_vis) with the key points already in crop coordinates.  aug = (center, scale, rot, flip)
        replaces the random draws (parity tests); center is the value BEFORE the flip mirrors it, like in the reference."""
        img = rec["image"]
        ih, iw = int(img.shape[0]), int(img.shape[1])
        joints = np.array(rec["joints_3d"], dtype=np.float64).copy()
        joints_vis = np.array(rec["joints_3d_vis"], dtype=np.float64).copy()
        has_cond = "cond_joints" in rec
        cj = np.array(rec["cond_joints"], dtype=np.float64).copy() if has_cond else np.zeros_like(joints)
        cv = np.array(rec["cond_joints_vis"], dtype=np.float64).copy() if has_cond else np.zeros_like(joints_vis)
        if rec.get("use_bu_bbox", False) and has_cond and cj[:, 0].sum() != 0 and cj[0, 1].sum() != 0:
            x, y, w, h = box_from_keypoints(cj, self.bu_bbox_margin, iw, ih)
            center, scale = xywh2cs(x, y, w, h, self.aspect_ratio, self.scale_thre)
        else:
            center = np.array(rec["center"], dtype=np.float32).copy()
            scale = np.array(rec["scale"], dtype=np.float32).copy()
        if aug is None:
            center, scale, rot, flip = self.draw_augmentation(rec, center, scale)
        else:
            center, scale, rot, flip = np.array(aug[0], np.float32).copy(), np.array(aug[1], np.float32), aug[2], aug[3]
        if flip:
            joints, joints_vis = fliplr_joints(joints, joints_vis, iw, self.flip_pairs)
            center[0] = iw - center[0] - 1
            if has_cond:
                cj, cv = fliplr_joints(cj, cv, iw, self.flip_pairs)
        trans = get_affine_transform(center, scale, rot, self.image_size)
        for i in range(self.num_joints):
            if joints_vis[i, 0] > 0.0:
                joints[i, 0:2] = affine_transform(joints[i, 0:2], trans)
            if has_cond and cv[i, 0] > 0.0:
                cj[i, 0:2] = affine_transform(cj[i, 0:2], trans)
        return dict(trans=trans, center=center, scale=scale, rot=rot, flip=bool(flip), joints=joints,
                    joints_vis=joints_vis, cond_joints=cj, cond_joints_vis=cv)

    # ---- batched device work -----------------------------------------------------------------------------------
    def render(self, images, geos, want_crop=False):
        """images: uint8 HWC device tensors; geos: geometry() results.  Returns input [B, 3(+3), H, W], target
        [B, K, h, w], target_weight [B, K, 1] (+ the uint8 crops) on the device."""
        dev = images[0].device
        B, K = len(images), self.num_joints
        W, H = int(self.image_size[0]), int(self.image_size[1])
        cc = (K if self.stacked else 3) if self.conditional else 0      # stacked: one condition channel per joint
        x = torch.empty((B, 3 + cc, H, W), dtype=torch.float32, device=dev)
        items = (_WarpItem * B)()
        for b, (img, g) in enumerate(zip(images, geos)):
            if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous() or not img.is_cuda:
                raise ValueError("images must be contiguous uint8 [H, W, 3] device tensors")
            items[b].src, items[b].H, items[b].W = img.data_ptr(), int(img.shape[0]), int(img.shape[1])
            items[b].flip = int(g["flip"])
            rect = g.get("keep_rect")
            items[b].rx, items[b].ry, items[b].rw, items[b].rh = (int(v) for v in rect) if rect is not None else (0, 0, 0, 0)
            for k, v in enumerate(np.asarray(g["trans"], dtype=np.float64).reshape(6)):
                items[b].m[k] = float(v)
        table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
        crop = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) if want_crop else None
        mean = (C.c_float * 3)(*self.mean.tolist())
        std = (C.c_float * 3)(*self.std.tolist())
        check(lib().buctd_warp_affine_norm(ptr(table), B, H, W, mean, std, ptr(x), x.stride(0), ptr(crop), stream_ptr()),
              "warp_affine_norm")
        # Gaussian targets: the heat-map centre mu = int(j / stride + 0.5) is evaluated here in float64 exactly like the
        # reference (JointsDataset.py:417-418).  The kernel recomputes (int)(v / stride + 0.5f), truncating toward zero as
        # well, so it is handed a v that maps back to the same mu: mu * stride for mu >= 0 and (mu - 1) * stride for mu < 0
        # (v / stride + 0.5 = mu - 0.5 truncates to mu; mu * stride would give mu + 0.5 -> mu + 1 for negative centres:
        # visible joints left of / above the crop would get a shifted Gaussian and a shifted target_weight cut-off)
        stride = self.image_size / self.heatmap_size
        jt = np.zeros((B, K, 3), dtype=np.float32)
        vis = np.zeros((B, K), dtype=np.float32)
        for b, g in enumerate(geos):
            mu_x = (g["joints"][:, 0] / stride[0] + 0.5).astype(int)
            mu_y = (g["joints"][:, 1] / stride[1] + 0.5).astype(int)
            jt[b, :, 0] = np.where(mu_x < 0, mu_x - 1, mu_x) * stride[0]
            jt[b, :, 1] = np.where(mu_y < 0, mu_y - 1, mu_y) * stride[1]
            vis[b] = g["joints_vis"][:, 0]
        target, weight = ops.gaussian_target(torch.from_numpy(jt).to(dev), torch.from_numpy(vis).to(dev),
                                             self.heatmap_size, self.image_size, self.sigma)
        if cc:
            # np.array(kpts).astype(int) truncates in float64 (JointsDataset.py:521): hand the kernel the integers
            cj = np.stack([np.trunc(g["cond_joints"][:, :2]) for g in geos]).astype(np.float32)
            cjt = torch.from_numpy(np.ascontiguousarray(cj)).to(dev)
            colors = None
            if self.colored:
                colors = torch.from_numpy(np.ascontiguousarray(self.kpt_colors[:K])).to(dev)
            ws = ops.workspace(lib().buctd_cond_render_workspace(B * K if self.stacked else B, 3, H, W), dev)
            if self.stacked:
                # get_stacked_condition (JointsDataset.py:471-498): every joint is its own single-impulse image, blurred
                # and peak-normalised on its own - B * K one-joint "images" of one channel for the render kernel
                tmp = torch.empty((B * K, 1, H, W), dtype=torch.float32, device=dev)
                check(lib().buctd_cond_render_into(ptr(cjt), 2, None, B * K, 1, 1, H, W, 0, ptr(tmp), tmp.stride(0), ptr(ws),
                                                   ws.numel(), stream_ptr()), "cond_render_into")
                x[:, 3:] = tmp.view(B, K, H, W)
            elif self.colored:
                check(lib().buctd_cond_render_into(ptr(cjt), 2, ptr(colors), B, K, 3, H, W, 0,
                                                   C.c_void_p(x[:, 3:].data_ptr()), x.stride(0), ptr(ws), ws.numel(),
                                                   stream_ptr()), "cond_render_into")
            else:
                # mono: the one blurred, int-truncated channel replicated x3 (JointsDataset.py:513-514)
                for c in range(3):
                    check(lib().buctd_cond_render_into(ptr(cjt), 2, None, B, K, 1, H, W, 1,
                                                       C.c_void_p(x[:, 3 + c:].data_ptr()), x.stride(0), ptr(ws),
                                                       ws.numel(), stream_ptr()), "cond_render_into")
        return (x, target, weight, crop) if want_crop else (x, target, weight)

    def __call__(self, records, aug=None):
        """records: dicts with 'image' (uint8 HWC device tensor), 'joints_3d', 'joints_3d_vis', 'center', 'scale' and,
        for conditional models, 'cond_joints' / 'cond_joints_vis' (+ 'score', 'annotation_id', 'use_bu_bbox').
        Returns (input, target, target_weight, meta) like a collated DataLoader batch of the reference."""
        geos = [self.geometry(r, None if aug is None else aug[i]) for i, r in enumerate(records)]
        x, target, weight = self.render([r["image"] for r in records], geos)
        meta = {
            "image": [r.get("image_file", "") for r in records],
            "joints": torch.from_numpy(np.stack([g["joints"] for g in geos])),
            "joints_vis": torch.from_numpy(np.stack([g["joints_vis"] for g in geos])),
            "cond_joints": torch.from_numpy(np.stack([g["cond_joints"] for g in geos])),
            "cond_joints_vis": torch.from_numpy(np.stack([g["cond_joints_vis"] for g in geos])),
            "center": torch.from_numpy(np.stack([g["center"] for g in geos])),
            "scale": torch.from_numpy(np.stack([g["scale"] for g in geos])),
            "rotation": torch.tensor([float(g["rot"]) for g in geos]),
            "score": torch.tensor([float(r.get("score", 1)) for r in records]),
            "annotation_id": torch.tensor([int(r.get("annotation_id", -1)) for r in records]),
        }
        return x, target, weight, meta


class IterativeRefiner:
    """BUCTD iterative refinement in one process (README.md:104 '3x iterative refinement'; reference = three CLI runs
    chained through the results json)."""

    def __init__(self, cfg, model, pipeline, in_vis_thre=None):
        self.cfg, self.model, self.pipe = cfg, model, pipeline
        self.in_vis_thre = cfg.TEST.IN_VIS_THRE if in_vis_thre is None else in_vis_thre

    @staticmethod
    def rescore(maxvals, box_score, in_vis_thre):
        """dataloader.py:596-612: mean of the key-point scores above in_vis_thre, times the box score."""
        mv = maxvals[:, :, 0]
        counted = mv > in_vis_thre
        n = counted.sum(1)
        kpt_score = np.where(n > 0, (mv * counted).sum(1) / np.maximum(n, 1), 0.0)
        return kpt_score * box_score, kpt_score

    def next_records(self, records, preds, scores):
        """prediction -> condition + box of the next pass (dataloader.py:454-508, _load_coco_pose_results)."""
        out = []
        for r, kp, sc in zip(records, preds, scores):
            ih, iw = int(r["image"].shape[0]), int(r["image"].shape[1])
            cond = np.zeros((kp.shape[0], 3), dtype=np.float64)
            cond[:, :2] = kp[:, :2]
            cond[:, 2] = kp[:, 2] if kp.shape[1] > 2 else 0.0
            x, y, w, h = box_from_keypoints(cond, self.pipe.bu_bbox_margin, iw, ih)
            c, s = xywh2cs(x, y, w, h, self.pipe.aspect_ratio, self.pipe.scale_thre)
            nr = dict(r)
            nr.update(center=c, scale=s, score=float(sc), cond_joints=cond,
                      cond_joints_vis=np.ones((kp.shape[0], 3), dtype=np.float64),
                      joints_3d=np.zeros((kp.shape[0], 3), dtype=np.float64),
                      joints_3d_vis=np.ones((kp.shape[0], 3), dtype=np.float64), use_bu_bbox=False)
            out.append(nr)
        return out

    @torch.no_grad()
    def run(self, records, passes=3):
        """Returns per pass: dict(preds [B, K, 3] image coordinates + max-val, score, box_score, keypoint_score)."""
        self.model.eval()
        history = []
        for _ in range(passes):
            geos = [self.pipe.geometry(r) for r in records]
            x, _, _ = self.pipe.render([r["image"] for r in records], geos)
            out = self.model(x)
            out = out[-1] if isinstance(out, list) else out
            center = np.stack([g["center"] for g in geos])
            scale = np.stack([g["scale"] for g in geos])
            coords, maxvals = get_final_preds(self.cfg, out, center, scale)
            box_score = np.array([float(r.get("score", 1)) for r in records])
            score, kpt_score = self.rescore(maxvals, box_score, self.in_vis_thre)
            preds = np.concatenate([coords, maxvals], axis=2)
            history.append(dict(preds=preds, score=score, box_score=box_score, keypoint_score=kpt_score,
                                center=center, scale=scale))
            records = self.next_records(records, preds, score)
        return history
