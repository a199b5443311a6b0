"""SURVEY 8f rows f1 (GPU-side sample pipeline) and f3 (in-process iterative refinement).

CPU: the oracle's restatement of cv2.warpAffine's fixed-point bilinear against hand-derived vectors (cv2 is absent
from the image: the crop is pinned this way only), and the host-side geometry of the product against the oracle.
GPU: the batched HIP pipeline (crop + normalise, Gaussian target, condition heat-map) against the oracle sample by
sample - the 8-bit crop and the normalised input bit-exact - and the refinement loop against the same loop on the oracle."""
import numpy as np
import pytest
import torch

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def test_warp_restatement_hand_derived_vectors():
    from oracle import sample as S
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (20, 30, 3)).astype(np.uint8)
    assert np.array_equal(S.warp_affine_u8(img, [[1, 0, 0], [0, 1, 0]], (30, 20)), img)            # identity
    out = S.warp_affine_u8(img, [[1, 0, 3], [0, 1, -2]], (30, 20))                                   # integer shift
    ref = np.zeros_like(img)
    ref[:18, 3:] = img[2:, :27]
    assert np.array_equal(out, ref)
    # +0.5 px in x: source coordinate x - 0.5 -> 1/1024 px: -512 + 16 + 1024 x -> >> 5 = 32 x - 16: neighbour x-1 and x
    # with weights 16/32 each -> (a + b + 1) >> 1, zero beyond the left border
    out = S.warp_affine_u8(img, [[1, 0, 0.5], [0, 1, 0]], (30, 20))
    p = np.concatenate([np.zeros((20, 1, 3), int), img.astype(int)], 1)
    assert np.array_equal(out, ((p[:, :-1] + p[:, 1:] + 1) >> 1).astype(np.uint8))
    # x2 zoom: even destination pixels copy, odd ones average their two sources
    out = S.warp_affine_u8(img, [[2, 0, 0], [0, 2, 0]], (60, 40))
    assert np.array_equal(out[::2, ::2], img)
    assert np.array_equal(out[0, 1::2][:-1], ((img[0, :-1].astype(int) + img[0, 1:] + 1) >> 1).astype(np.uint8))
    # mirrored source and keep-rectangle
    assert np.array_equal(S.warp_affine_u8(img, [[1, 0, 0], [0, 1, 0]], (30, 20), flip_src=True), img[:, ::-1])
    out = S.warp_affine_u8(img, [[1, 0, 0], [0, 1, 0]], (30, 20), keep_rect=(5, 4, 10, 6))
    ref = np.zeros_like(img)
    ref[4:10, 5:15] = img[4:10, 5:15]
    assert np.array_equal(out, ref)
    # 90 degree rotation about the origin lands on exact pixels: dst(x, y) = src(y, -x) -> first column = first row
    out = S.warp_affine_u8(img, [[0, -1, 19], [1, 0, 0]], (20, 30))
    assert np.array_equal(out[:, 19], img[0, :])


def _cfg(colored=True):
    from oracle import cfg as ocfg
    c = ocfg.hrnet_cfg(16, 14, (64, 96), "pose_hrnet_coam", use_attention=True, colored=colored,
                       stage_modules=(1, 1, 1))
    c.DATASET.update({"SCALE_FACTOR": 0.35, "ROT_FACTOR": 45, "FLIP": True, "NUM_JOINTS_HALF_BODY": 8,
                      "PROB_HALF_BODY": 0.3, "BU_BBOX_MARGIN": 25})
    c.TEST.update({"SCALE_THRE": 1.25, "IN_VIS_THRE": 0.2})
    return c


def _records(n, seed, k=14):
    from oracle import sample as S
    rng = np.random.RandomState(seed)
    recs = []
    for i in range(n):
        h, w = int(rng.randint(90, 200)), int(rng.randint(100, 260))
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        joints = np.zeros((k, 3))
        joints[:, 0], joints[:, 1] = rng.rand(k) * (w - 20) + 10, rng.rand(k) * (h - 20) + 10
        vis = np.repeat((rng.rand(k, 1) > 0.2).astype(float), 3, 1)
        vis[:, 2] = 0
        cond = joints.copy()
        cond[:, :2] += rng.randn(k, 2) * 3
        x, y, bw, bh = S.box_from_keypoints(joints, 10, w, h)
        c, s = S.xywh2cs(x, y, bw, bh, 64 / 96, 1.25)
        recs.append({"image_np": img, "joints_3d": joints, "joints_3d_vis": vis, "cond_joints": cond,
                     "cond_joints_vis": np.ones((k, 3)), "center": c, "scale": s, "score": 0.5 + 0.1 * i,
                     "annotation_id": 100 + i})
    return recs


def test_host_geometry_matches_oracle():
    from oracle import core as oc, sample as S
    from buctd_amd.dataset.pipeline import DeviceSamplePipeline, box_from_keypoints, xywh2cs
    cfg = _cfg()
    pipe = DeviceSamplePipeline(cfg, oc.CROWDPOSE_FLIP_PAIRS, range(8), oc.CROWDPOSE_KPT_COLORS, MEAN, STD, is_train=True)
    for i, r in enumerate(_records(6, 3)):
        rec = dict(r, image=torch.from_numpy(r["image_np"]))
        aug = (r["center"] + i, r["scale"] * (1 + 0.1 * i), 10.0 * i - 20, bool(i % 2))
        g = pipe.geometry(rec, aug)
        joints, vis = r["joints_3d"].copy(), r["joints_3d_vis"].copy()
        cj, cv = r["cond_joints"].copy(), r["cond_joints_vis"].copy()
        if aug[3]:
            joints, vis = oc.fliplr_joints(joints, vis, r["image_np"].shape[1], oc.CROWDPOSE_FLIP_PAIRS)
            cj, cv = oc.fliplr_joints(cj, cv, r["image_np"].shape[1], oc.CROWDPOSE_FLIP_PAIRS)
        ctr = np.array(aug[0], np.float32).copy()
        if aug[3]:
            ctr[0] = r["image_np"].shape[1] - ctr[0] - 1
        t = oc.get_affine_transform(ctr, aug[1], aug[2], [64, 96])
        assert np.abs(g["trans"] - t).max() <= 1e-9
        for k in range(14):
            if vis[k, 0] > 0:
                joints[k, :2] = oc.affine_transform(joints[k, :2], t)
            cj[k, :2] = oc.affine_transform(cj[k, :2], t)
        assert np.allclose(g["joints"], joints, atol=1e-9) and np.allclose(g["cond_joints"], cj, atol=1e-9)
        assert np.array_equal(box_from_keypoints(r["cond_joints"], 25, 300, 200),
                              S.box_from_keypoints(r["cond_joints"], 25, 300, 200))
        a, b = xywh2cs(10.5, 20.25, 80, 90, 64 / 96, 1.25), S.xywh2cs(10.5, 20.25, 80, 90, 64 / 96, 1.25)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # the random part: train-mode draws stay inside the reference's clipping ranges
    rec = dict(_records(1, 4)[0])
    rec["image"] = torch.from_numpy(rec["image_np"])
    for _ in range(50):
        c, s, rot, flip = pipe.draw_augmentation(rec, rec["center"].copy(), rec["scale"].copy())
        assert abs(rot) <= 90 and flip in (True, False) and np.all(s > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("colored", [True, False, "stacked"])
def test_device_pipeline_matches_oracle(dev, colored):
    """colored / mono (x3, int-truncated) / stacked (one peak-normalised channel per joint, JointsDataset.py:471-498)"""
    from oracle import core as oc, sample as S
    from buctd_amd.dataset.pipeline import DeviceSamplePipeline
    stacked = colored == "stacked"
    colored = bool(colored) and not stacked
    cfg = _cfg(colored)
    cfg.DATASET.STACKED_CONDITION = stacked
    pipe = DeviceSamplePipeline(cfg, oc.CROWDPOSE_FLIP_PAIRS, range(8), oc.CROWDPOSE_KPT_COLORS, MEAN, STD, is_train=True)
    recs = _records(5, 11)
    augs = [(r["center"] + np.float32(i), r["scale"] * np.float32(1 + 0.07 * i), [0, 17.5, -33, 0, 45][i], bool(i % 2))
            for i, r in enumerate(recs)]
    dev_recs = [dict(r, image=torch.from_numpy(r["image_np"]).to(dev)) for r in recs]
    geos = [pipe.geometry(r, a) for r, a in zip(dev_recs, augs)]
    x, target, weight, crop = pipe.render([r["image"] for r in dev_recs], geos, want_crop=True)
    x2, t2, w2, meta = pipe(dev_recs, augs)
    assert torch.equal(x, x2) and torch.equal(target, t2) and meta["center"].shape == (5, 2)
    for i, (r, a) in enumerate(zip(recs, augs)):
        xo, to, wo, jo, cjo, cropo = S.make_sample(r["image_np"], r["joints_3d"], r["joints_3d_vis"], r["cond_joints"],
                                                   r["cond_joints_vis"], a[0], a[1], a[2], a[3], [64, 96], [16, 24], 2,
                                                   oc.CROWDPOSE_FLIP_PAIRS, MEAN, STD, oc.CROWDPOSE_KPT_COLORS[:14],
                                                   mono=not colored, stacked=stacked)
        assert x.shape[1] == (3 + 14 if stacked else 6)
        assert np.array_equal(crop[i].cpu().numpy(), cropo), f"sample {i}: 8-bit crop differs"
        assert np.array_equal(x[i, :3].cpu().numpy(), xo[:3]), f"sample {i}: normalised crop differs"
        assert np.abs(target[i].cpu().numpy() - to).max() <= 2e-7 and np.array_equal(weight[i].cpu().numpy(), wo)
        exact = colored or stacked
        tol = 2e-3 if exact else 1.0          # mono is int-truncated: a value within 2e-3 of an integer may land below it
        dc = np.abs(x[i, 3:].cpu().numpy() - xo[3:])
        assert dc.max() <= tol and (dc > 2e-3).mean() <= (0.0 if exact else 1e-3), \
            f"sample {i}: condition differs by {dc.max()}"


@pytest.mark.gpu
def test_targets_of_visible_joints_outside_the_crop(dev):
    """Visible joints that land left of / above the crop (negative heat-map centres, JointsDataset.py:417-430): the Gaussian
    and the target_weight cut-off (br < 0) must sit where generate_target puts them - int() truncates toward zero, so
    mu = -1 comes from j / stride + 0.5 in (-2, -1]."""
    from oracle import core as oc, sample as S
    from buctd_amd.dataset.pipeline import DeviceSamplePipeline
    cfg = _cfg(True)
    pipe = DeviceSamplePipeline(cfg, oc.CROWDPOSE_FLIP_PAIRS, range(8), oc.CROWDPOSE_KPT_COLORS, MEAN, STD, is_train=True)
    recs = _records(3, 23)
    for r in recs:
        # push joints outside the box the crop is taken from: crop x in about [-40, -1] and y likewise
        c, s = r["center"], r["scale"] * 200.0
        x0, y0 = c[0] - s[0] / 2, c[1] - s[1] / 2
        for k, (fx, fy) in enumerate([(-0.02, 0.3), (-0.09, 0.5), (-0.16, 0.7), (-0.25, 0.2), (-0.40, 0.5), (0.4, -0.03),
                                      (0.6, -0.11), (0.5, -0.19), (-0.05, -0.05), (-0.21, -0.3)]):
            r["joints_3d"][k, 0], r["joints_3d"][k, 1] = x0 + fx * s[0], y0 + fy * s[1]
            r["joints_3d_vis"][k, :2] = 1
    augs = [(r["center"], r["scale"], 0.0, False) for r in recs]
    dev_recs = [dict(r, image=torch.from_numpy(r["image_np"]).to(dev)) for r in recs]
    x, target, weight, meta = pipe(dev_recs, augs)
    neg = 0
    for i, (r, a) in enumerate(zip(recs, augs)):
        xo, to, wo, jo, cjo, cropo = S.make_sample(r["image_np"], r["joints_3d"], r["joints_3d_vis"], r["cond_joints"],
                                                   r["cond_joints_vis"], a[0], a[1], a[2], a[3], [64, 96], [16, 24], 2,
                                                   oc.CROWDPOSE_FLIP_PAIRS, MEAN, STD, oc.CROWDPOSE_KPT_COLORS[:14], mono=False)
        neg += int(((jo[:10, 0] / 4 + 0.5).astype(int) < 0).sum() + ((jo[:10, 1] / 4 + 0.5).astype(int) < 0).sum())
        assert np.array_equal(weight[i].cpu().numpy(), wo), f"sample {i}: target_weight differs"
        assert np.abs(target[i].cpu().numpy() - to).max() <= 2e-7, f"sample {i}: target differs"
    assert neg >= 8, "the fixture is meant to produce negative heat-map centres"


@pytest.mark.gpu
def test_iterative_refinement_matches_oracle_loop(dev):
    """3 chained passes of a conditional model: prediction -> box / condition -> new crop -> prediction (f3)."""
    from oracle import core as oc, recipes, sample as S
    from buctd_amd import models
    from buctd_amd.dataset.pipeline import DeviceSamplePipeline, IterativeRefiner
    cfg, omodel, _, _ = recipes.build("coam_w16_96x64_colored")
    cfg.DATASET.update({"BU_BBOX_MARGIN": 25, "FLIP": False})
    cfg.TEST.update({"SCALE_THRE": 1.25, "IN_VIS_THRE": 0.2})
    m = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)
    m.load_state_dict(omodel.state_dict(), strict=True)
    m = m.to(dev).eval()
    pipe = DeviceSamplePipeline(cfg, oc.CROWDPOSE_FLIP_PAIRS, range(8), oc.CROWDPOSE_KPT_COLORS, MEAN, STD, is_train=False)
    recs = _records(3, 21)
    hist = IterativeRefiner(cfg, m, pipe).run([dict(r, image=torch.from_numpy(r["image_np"]).to(dev)) for r in recs], 3)
    assert len(hist) == 3 and hist[0]["preds"].shape == (3, 14, 3)
    # the same loop on the oracle (CPU model, numpy pipeline)
    cur = [dict(r) for r in recs]
    for p in range(3):
        xs, cs, ss = [], [], []
        for r in cur:
            xo = S.make_sample(r["image_np"], r["joints_3d"], r["joints_3d_vis"], r["cond_joints"], r["cond_joints_vis"],
                               r["center"], r["scale"], 0, False, [64, 96], [16, 24], 2, oc.CROWDPOSE_FLIP_PAIRS, MEAN,
                               STD, oc.CROWDPOSE_KPT_COLORS[:14])[0]
            xs.append(xo); cs.append(r["center"]); ss.append(r["scale"])
        with torch.no_grad():
            out = omodel(torch.from_numpy(np.stack(xs))).numpy()
        coords, maxvals = oc.get_final_preds(True, out, np.stack(cs), np.stack(ss))
        mv = maxvals[:, :, 0]
        kp_score = np.array([mv[i][mv[i] > 0.2].mean() if (mv[i] > 0.2).any() else 0.0 for i in range(len(cur))])
        score = kp_score * np.array([r["score"] for r in cur])
        h = hist[p]
        same = np.abs(h["preds"][:, :, :2] - coords).max(axis=2) <= 1e-3
        assert same.mean() >= 0.95, f"pass {p}: {100 * (1 - same.mean()):.1f}% of the key points moved"
        assert np.abs(h["preds"][:, :, 2:] - maxvals).max() <= 2e-3
        if same.all():
            assert np.allclose(h["score"], score, atol=2e-3)
        nxt = []
        for r, kp, sc in zip(cur, h["preds"], h["score"]):      # continue from the product's predictions
            cond = np.zeros((14, 3)); cond[:, :2] = kp[:, :2]; cond[:, 2] = kp[:, 2]
            x, y, w, hh = S.box_from_keypoints(cond, 25, r["image_np"].shape[1], r["image_np"].shape[0])
            c, s = S.xywh2cs(x, y, w, hh, 64 / 96, 1.25)
            nxt.append(dict(r, center=c, scale=s, score=float(sc), cond_joints=cond, cond_joints_vis=np.ones((14, 3)),
                            joints_3d=np.zeros((14, 3)), joints_3d_vis=np.ones((14, 3))))
        cur = nxt


@pytest.mark.gpu
def test_iterative_refinement_on_a_forward_graph_equals_the_eager_loop(dev):
    """The same three chained passes with the network wrapped in engine.ForwardGraph (the serving path: a few persons per
    call): every pass returns exactly what the eager network returns - predictions, scores, boxes."""
    from oracle import core as oc, recipes
    from buctd_amd import engine, models
    from buctd_amd.dataset.pipeline import DeviceSamplePipeline, IterativeRefiner
    cfg, omodel, _, _ = recipes.build("coam_w16_96x64_colored")
    cfg.DATASET.update({"BU_BBOX_MARGIN": 25, "FLIP": False})
    cfg.TEST.update({"SCALE_THRE": 1.25, "IN_VIS_THRE": 0.2})
    m = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)
    m.load_state_dict(omodel.state_dict(), strict=True)
    m = m.to(dev).eval()
    pipe = DeviceSamplePipeline(cfg, oc.CROWDPOSE_FLIP_PAIRS, range(8), oc.CROWDPOSE_KPT_COLORS, MEAN, STD, is_train=False)
    recs = _records(3, 21)

    def run(model):
        return IterativeRefiner(cfg, model, pipe).run([dict(r, image=torch.from_numpy(r["image_np"]).to(dev)) for r in recs], 3)

    eager = run(m)
    fg = engine.ForwardGraph(m, warmup=1, autoselect=False)
    graphed = run(fg)
    graphed2 = run(fg)              # all three passes replayed
    assert fg.replays >= 4
    for a, b, c in zip(eager, graphed, graphed2):
        for k in ("preds", "score", "box_score", "keypoint_score", "center", "scale"):
            assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), k
