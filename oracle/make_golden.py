"""TEST INFRASTRUCTURE - pins the oracle against the real reference and writes tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference, read-only).  It imports the reference's
own Python modules (lib/models, lib/core, lib/utils/transforms) with the stub set of SURVEY 8c
for packages this image lacks (torchvision, cv2, yacs), drives them and the oracle with the same
seeded state_dict + inputs, asserts agreement, and stores the REFERENCE's outputs as golden
vectors.  Nothing from the reference's sources is copied; only inputs/outputs (data) are kept.

    python -m oracle.make_golden            # all cases
    python -m oracle.make_golden --fast     # skip the full-size cases (BASELINE configs C1-C5)
"""
import argparse
import hashlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def install_stubs():
    """Stand-ins for packages missing from this image - used by this script only."""
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvu = types.ModuleType("torchvision.utils")

    def resize(img, size, *a, **k):  # torchvision 0.9 tensor resize: bilinear, no antialias
        return F.interpolate(img, size=tuple(size), mode="bilinear", align_corners=False)

    tvf.resize = resize
    tvt.functional = tvf
    tvt.Normalize = lambda *a, **k: None
    tv.transforms, tv.utils = tvt, tvu
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf, "torchvision.utils": tvu})
    cv2 = types.ModuleType("cv2")

    def get_affine(src, dst):
        A = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], 1)
        return np.linalg.solve(A, np.asarray(dst, np.float64)).T

    cv2.getAffineTransform = get_affine
    cv2.INTER_LINEAR = 1
    sys.modules["cv2"] = cv2
    torch.Tensor.cuda = lambda self, *a, **k: self  # reference forwards call x.cuda()


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:16]


def state_digest(model):
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()[:16]


def close(a, b, tol, what):
    err = (a - b).abs().max().item()
    assert err <= tol, f"{what}: reference vs oracle max abs diff {err:.3e} > {tol}"
    return err


def ref_model_for(name, cfg):
    import models  # reference lib/models
    mod = {"pose_hrnet": models.pose_hrnet, "pose_hrnet_coam": models.pose_hrnet_coam,
           "transpose_h": models.transpose_h, "pose_resnet": models.pose_resnet}[cfg.MODEL.NAME]
    return mod.get_pose_net(cfg, is_train=False)


def model_case(name, train):
    from oracle import recipes, core as ocore
    cfg, omodel, x, joints = recipes.build(name)
    rmodel = ref_model_for(name, cfg)
    # identical keys / shapes: strict load of the oracle's state into the reference
    rmodel.load_state_dict(omodel.state_dict(), strict=True)
    assert list(rmodel.state_dict().keys()) == list(omodel.state_dict().keys()), name + ": key order differs"
    rmodel.eval()
    with torch.no_grad():
        y_ref = rmodel(x)
        y_orc = omodel(x)
    err = close(y_ref, y_orc, 1e-5, name + " eval forward")
    rec = {"out": y_ref.numpy(), "x_sha": sha(x), "state_sha": state_digest(omodel)}
    flat = y_ref.reshape(y_ref.shape[0], y_ref.shape[1], -1)
    rec["argmax"] = flat.argmax(2).numpy().astype(np.int32)
    print(f"  {name}: eval fwd ok (ref-oracle diff {err:.2e}), out |max| {y_ref.abs().max():.3f}")
    if train:
        # one training forward/backward with dropout disabled (SURVEY 8c train-step protocol)
        tgt, wt = recipes.make_targets(cfg, joints, 77)
        crit_ref = __import__("core.loss", fromlist=["JointsMSELoss"]).JointsMSELoss(True)
        outs = {}
        for tag, m in (("ref", rmodel), ("orc", omodel)):
            m.train()
            recipes.set_dropout(m, 0.0)
            m.zero_grad()
            y = m(x)
            loss = crit_ref(y, tgt, wt) if tag == "ref" else ocore.JointsMSELoss(True)(y, tgt, wt)
            loss.backward()
            outs[tag] = (y.detach(), loss.detach(),
                         {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None},
                         {k: b.detach().clone() for k, b in m.named_buffers()})
        close(outs["ref"][0], outs["orc"][0], 1e-5, name + " train forward")
        close(outs["ref"][1], outs["orc"][1], 1e-7, name + " loss")
        for k, gref in outs["ref"][2].items():
            close(gref, outs["orc"][2][k], 1e-5 * max(1.0, gref.abs().max().item()), name + " grad " + k)
        for k, bref in outs["ref"][3].items():
            close(bref.float(), outs["orc"][3][k].float(), 1e-5, name + " buffer " + k)
        y, loss, grads, bufs = outs["ref"]
        rec["train_out"] = y.numpy()
        rec["loss"] = loss.numpy()
        names = sorted(grads)
        rec["grad_names"] = np.array(names)
        rec["grad_norms"] = np.array([grads[k].norm().item() for k in names], np.float64)
        rec["grad_sums"] = np.array([grads[k].double().sum().item() for k in names], np.float64)
        for k in ("final_layer.weight", "conv1.weight"):
            if k in grads:
                rec["grad::" + k] = grads[k].numpy()
        bnames = sorted(k for k in bufs if k.endswith("running_mean") or k.endswith("running_var"))
        rec["buf_names"] = np.array(bnames)
        rec["buf_norms"] = np.array([bufs[k].norm().item() for k in bnames], np.float64)
        rec["target_sha"] = sha(tgt)
        print(f"  {name}: train fwd/bwd ok, loss {loss.item():.6f}, {len(names)} grads pinned")
    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **rec)


def core_cases():
    """loss / decode / accuracy / flip / affine: reference functions vs oracle on seeded inputs."""
    from core.loss import JointsMSELoss as RefLoss
    from core.inference import get_max_preds as ref_gmp, get_final_preds as ref_gfp
    from core.evaluate import accuracy as ref_acc
    from utils.transforms import flip_back as ref_flip_back, fliplr_joints as ref_fliplr
    from oracle import core as oc
    g = torch.Generator().manual_seed(2024)
    N, K, H, W = 6, 14, 24, 18
    pred = torch.randn(N, K, H, W, generator=g)
    gt = torch.rand(N, K, H, W, generator=g)
    wt = (torch.rand(N, K, 1, generator=g) > 0.25).float() * (0.5 + torch.rand(N, K, 1, generator=g))
    p1 = pred.clone().requires_grad_(True)
    l_ref = RefLoss(True)(p1, gt, wt)
    l_ref.backward()
    p2 = pred.clone().requires_grad_(True)
    l_orc = oc.JointsMSELoss(True)(p2, gt, wt)
    l_orc.backward()
    close(l_ref.detach(), l_orc.detach(), 1e-8, "loss")
    close(p1.grad, p2.grad, 1e-9, "loss grad")
    close(l_ref.detach(), oc.joints_mse_closed_form(pred, gt, wt), 1e-7, "loss closed form")
    # decode incl. ties / non-positive maxima / borders
    hm = pred.numpy().copy()
    hm[0, 0] = 0.25
    hm[0, 0, 3, 4] = 2.0
    hm[0, 0, 10, 2] = 2.0
    hm[1, 1] = -1.0
    hm[2, 2] = 0.0
    hm[3, 3, 0, 0] = 9.0
    hm[3, 4, H - 1, W - 1] = 9.0
    hm[3, 5, 1, 1] = 9.0
    hm[3, 6, 2, 2] = 9.0
    pr, mv = ref_gmp(hm)
    po, mo = oc.get_max_preds(hm)
    assert np.array_equal(pr, po) and np.array_equal(mv, mo)
    center = (torch.rand(N, 2, generator=g) * 200 + 50).numpy()
    scale = (torch.rand(N, 2, generator=g) * 1.5 + 0.5).numpy()

    class C:
        class TEST:
            POST_PROCESS = True
    fr, fm = ref_gfp(C, hm.copy(), center, scale)
    fo, fmo = oc.get_final_preds(True, hm.copy(), center, scale)
    assert np.allclose(fr, fo, atol=1e-9) and np.array_equal(fm, fmo)
    ar = ref_acc(hm, gt.numpy())
    ao = oc.accuracy(hm, gt.numpy())
    assert np.allclose(ar[0], ao[0]) and ar[1] == ao[1] and ar[2] == ao[2] and np.array_equal(ar[3], ao[3])
    fb_r = ref_flip_back(hm.copy(), oc.CROWDPOSE_FLIP_PAIRS)
    fb_o = oc.flip_back(hm.copy(), oc.CROWDPOSE_FLIP_PAIRS)
    assert np.array_equal(fb_r, fb_o)
    j = (torch.rand(K, 3, generator=g) * 40).numpy()
    jv = (torch.rand(K, 3, generator=g) > 0.3).float().numpy()
    jr, jvr = ref_fliplr(j.copy(), jv.copy(), 48, oc.CROWDPOSE_FLIP_PAIRS)
    jo, jvo = oc.fliplr_joints(j, jv, 48, oc.CROWDPOSE_FLIP_PAIRS)
    assert np.array_equal(jr, jo) and np.array_equal(jvr, jvo)
    np.savez_compressed(os.path.join(OUT, "core.npz"), pred=pred.numpy(), gt=gt.numpy(), wt=wt.numpy(),
                        loss=l_ref.detach().numpy(), loss_grad=p1.grad.numpy(), hm=hm, preds=pr, maxvals=mv,
                        center=center, scale=scale, final_preds=fr, acc=ar[0], avg_acc=ar[1], cnt=ar[2],
                        flip_back=fb_r, merged=oc.flip_test_merge(hm, hm[::-1].copy(), oc.CROWDPOSE_FLIP_PAIRS, True))
    print("  core: loss / decode / final_preds / accuracy / flip_back match the reference")


def target_case():
    """generate_target: executes the reference function body (JointsDataset.py:397-453) with a dummy self
    (lib/dataset cannot be imported: cv2 / pycocotools / np.float), compares with the oracle."""
    import ast
    import textwrap
    from oracle import core as oc
    src = open(os.path.join(REF, "lib/dataset/JointsDataset.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "generate_target")
    code = textwrap.dedent("\n".join(src.splitlines()[fn.lineno - 1:fn.end_lineno]))
    ns = {"np": np}
    exec(compile(code, "generate_target", "exec"), ns)

    class Self:
        pass

    recs = {}
    for tag, (hw, iw, sig, k) in {"crowdpose": ((72, 96), (288, 384), 3, 14), "coco256": ((64, 96), (192, 256), 2, 17)}.items():
        s = Self()
        s.num_joints, s.heatmap_size, s.image_size = k, np.array(hw), np.array(iw)
        s.sigma, s.target_type, s.use_different_joints_weight = sig, "gaussian", False
        g = torch.Generator().manual_seed(5 + k)
        joints = torch.rand(k, 3, generator=g).numpy() * np.array([iw[0] * 1.4, iw[1] * 1.4, 0]) - \
            np.array([iw[0] * 0.2, iw[1] * 0.2, 0])
        joints[0, :2] = [-40.0, 10.0]      # fully outside (left)
        joints[1, :2] = [iw[0] + 60.0, 5]  # fully outside (right)
        joints[2, :2] = [-3.0, -3.0]       # partially outside, negative coords (int() truncation)
        joints[3, :2] = [iw[0] - 1.0, iw[1] - 1.0]
        vis = (torch.rand(k, 1, generator=g) > 0.2).float().repeat(1, 3).numpy()
        vis[2] = 1.0
        t_ref, w_ref = ns["generate_target"](s, joints.copy(), vis.copy())
        t_orc, w_orc = oc.generate_target(joints, vis, k, hw, iw, sig)
        assert np.array_equal(t_ref, t_orc) and np.array_equal(w_ref, w_orc), tag
        recs.update({f"{tag}_joints": joints, f"{tag}_vis": vis, f"{tag}_target": t_ref, f"{tag}_weight": w_ref,
                     f"{tag}_meta": np.array([hw[0], hw[1], iw[0], iw[1], sig, k])})
    np.savez_compressed(os.path.join(OUT, "target.npz"), **recs)
    print("  target: generate_target (executed reference body) == oracle, incl. out-of-bounds joints")


def nms_case():
    """lib/nms/nms.py imported from the reference (its two Cython extension modules stubbed - they cannot be built
    here: np.float / np.int in the .pyx sources) vs the oracle restatement; the reference's outputs become nms.npz."""
    for name in ("nms.cpu_nms", "nms.gpu_nms"):
        m = types.ModuleType(name)
        m.cpu_nms = m.gpu_nms = None
        sys.modules[name] = m
    from nms import nms as rn
    from oracle import nms as on
    rec = {}
    for tag, n, seed, thr in (("a", 300, 1, 0.3), ("b", 1000, 2, 0.5), ("c", 65, 3, 0.1)):
        d = on.make_boxes(n, seed)
        keep = [int(i) for i in rn.nms(d, thr)]
        assert keep == on.nms(d, thr), "nms " + tag
        # cpu_nms.pyx differs from nms.py only in >= vs > and float32 arithmetic: same keep list unless an overlap
        # lands within rounding of the threshold (checked not to happen on these inputs)
        assert keep == on.nms(d, thr, strict=False), "cpu_nms restatement " + tag
        rec[f"boxes_{tag}"], rec[f"thr_{tag}"], rec[f"keep_{tag}"] = d, np.float64(thr), np.array(keep, np.int64)
    for tag, n, k, seed in (("p", 40, 17, 2), ("q", 64, 14, 5)):
        db = on.make_poses(n, k, seed)
        sig = None if k == 17 else np.array([.79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89, .79, .79]) / 10.0
        kp = np.array([p["keypoints"].flatten() for p in db])
        ar = np.array([p["area"] for p in db])
        iou_r = rn.oks_iou(kp[0], kp[1:], ar[0], ar[1:], sig)
        iou_v = rn.oks_iou(kp[0], kp[1:], ar[0], ar[1:], sig, 0.3)
        assert np.allclose(iou_r, on.oks_iou(kp[0], kp[1:], ar[0], ar[1:], sig), atol=1e-15)
        assert np.allclose(iou_v, on.oks_iou(kp[0], kp[1:], ar[0], ar[1:], sig, 0.3), atol=1e-15)
        k1 = [int(i) for i in rn.oks_nms(db, 0.6, sig)]
        k2 = [int(i) for i in rn.oks_nms(db, 0.6, sig, 0.3)]
        k3 = [int(i) for i in rn.soft_oks_nms(db, 0.6, sig)]
        assert k1 == [int(i) for i in on.oks_nms(db, 0.6, sig)] and k2 == [int(i) for i in on.oks_nms(db, 0.6, sig, 0.3)]
        assert k3 == [int(i) for i in on.soft_oks_nms(db, 0.6, sig)]
        assert 0 < len(k1) < n, "the pose set must exercise suppression"
        rec.update({f"kpts_{tag}": kp, f"areas_{tag}": ar, f"scores_{tag}": np.array([p["score"] for p in db]),
                    f"oks_{tag}": iou_r, f"oksvis_{tag}": iou_v, f"oksnms_{tag}": np.array(k1, np.int64),
                    f"oksnmsvis_{tag}": np.array(k2, np.int64), f"softnms_{tag}": np.array(k3, np.int64)})
        if sig is not None:
            rec[f"sigmas_{tag}"] = sig
    np.savez_compressed(os.path.join(OUT, "nms.npz"), **rec)
    print("  nms: nms / oks_iou / oks_nms / soft_oks_nms of the reference == oracle (+ cpu_nms restatement)")


def pose_synthesis_case(runs=1500):
    """lib/dataset/pose_synthesis.py imported from the reference (pure numpy / random): its outputs on a fixed scene,
    classified geometrically into the five error types, against the counter-RNG restatement - frequencies per joint
    must agree within sampling noise.  The reference's class counts become tests/golden/pose_synthesis.npz."""
    import importlib.util
    import random
    from oracle import pose_synthesis as P
    spec = importlib.util.spec_from_file_location("ref_pose_synthesis", os.path.join(REF, "lib/dataset/pose_synthesis.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)

    class C(dict):
        __getattr__ = dict.__getitem__
    rec = {}
    for dataset in ("crowdpose", "coco"):
        K = 17 if dataset == "coco" else 14
        cfg = C(MODEL=C(NUM_JOINTS=K), DATASET=C(DATASET=dataset))
        joints, est, near, area = P.make_scene(dataset, 7)
        np.random.seed(11)
        random.seed(11)
        ref = np.zeros((K, 6), dtype=np.int64)          # 5 classes + "dropped" (all-zero output)
        orc = np.zeros((K, 6), dtype=np.int64)
        for it in range(runs):
            o = ps.synthesize_pose(cfg, joints.copy(), est.copy(), near.copy(), area, 0)
            q = P.synthesize_pose(dataset, joints, est, near, area, 0, seed=1000 + it)
            for j in range(K):
                ref[j, 5 if not o[j, :2].any() else P.classify(o[j], joints, est, near, area, dataset, j)] += 1
                orc[j, 5 if not q[j, :2].any() else P.classify(q[j], joints, est, near, area, dataset, j)] += 1
            if it == 0:
                assert np.array_equal(o[:, 2], q[:, 2]), "visibility column convention"
        fr, fo = ref / runs, orc / runs
        tol = 4.0 * np.sqrt(np.maximum(fr * (1 - fr), 1e-3) * 2 / runs) + 0.004
        bad = np.abs(fr - fo) > tol
        assert not bad.any(), f"{dataset}: class frequencies differ from the reference at {np.argwhere(bad).tolist()}:\n" \
                              f"{fr[bad]} vs {fo[bad]}"
        rec[f"{dataset}_ref_counts"], rec[f"{dataset}_runs"] = ref, np.int64(runs)
        print(f"  pose_synthesis[{dataset}]: {runs} runs, max class-frequency gap to the reference "
              f"{np.abs(fr - fo).max():.4f} (good/jitter/inv/swap/miss/dropped)")
    np.savez_compressed(os.path.join(OUT, "pose_synthesis.npz"), **rec)


def entry_batches(cfg, n_batches, batch, k=14, seed0=500, cond_channels=3, align=None):
    """The in-memory loader of the entry-point goldens: (input, target, target_weight, meta) batches, fully seeded -
    tests/test_gpu_entry.py rebuilds the same batches from the same seeds.
    align = (oracle model, train_mode): the targets of the even-numbered joints are rendered AT the arg-max of that model's
    own heat-maps (a private copy, so nothing of the caller's model changes), so that accuracy() inside train() / validate()
    sees hits as well as misses - with random targets every accuracy the entry points report is 0 and a wrong tensor handed
    to accuracy() would go unnoticed."""
    import copy
    from oracle import recipes
    out = []
    amodel = None
    if align is not None:
        amodel = copy.deepcopy(align[0])
        amodel.train(bool(align[1]))
        recipes.set_dropout(amodel, 0.0)
    for i in range(n_batches):
        x, joints = recipes.make_inputs(cfg, batch, seed0 + i, cond_channels)
        tj = joints
        if amodel is not None:
            with torch.no_grad():
                hm = amodel(x)
            hw = hm.shape[3]
            idx = hm.flatten(2).argmax(2)                                     # [B, K]
            peak = torch.stack([(idx % hw).float(), (idx // hw).float()], 2)  # heat-map (x, y)
            stride = cfg.MODEL.IMAGE_SIZE[0] / cfg.MODEL.HEATMAP_SIZE[0]
            tj = joints.clone()
            tj[:, 0::2] = peak[:, 0::2] * stride                              # generate_target: mu = int(j / stride + 0.5)
        tgt, wt = recipes.make_targets(cfg, tj, seed0 + 100 + i)
        g = torch.Generator().manual_seed(seed0 + 200 + i)
        meta = {"center": torch.rand(batch, 2, generator=g) * 100 + 50, "scale": torch.rand(batch, 2, generator=g) + 0.5,
                "score": torch.rand(batch, generator=g), "annotation_id": torch.arange(batch) + i * batch,
                "image": [f"img_{i}_{j}.jpg" for j in range(batch)],
                "cond_joints": torch.cat([joints, torch.zeros(batch, k, 1)], 2),
                "cond_joints_vis": torch.ones(batch, k, 3)}
        out.append((x, tgt, wt, meta))
    return out


def entry_case():
    """The reference's OWN entry points, lib/core/function.py:train (102-175) and validate (178-336), imported and run on
    the CPU (stubs: torchvision / cv2 / .cuda(); utils.vis - visualisation, out of scope - replaced by a no-op module;
    the fake dataset's get_condition_image_colored is the oracle's restatement of the cv2 blur).  Their
    outputs - per-iteration losses and accuracies, parameter / buffer checksums after three Adam steps, all_preds,
    all_boxes, image paths, the validation meters - become tests/golden/entry.npz."""
    vis = types.ModuleType("utils.vis")
    vis.save_debug_images = lambda *a, **k: None
    sys.modules["utils.vis"] = vis
    import core.function as rf          # the reference module
    from core.loss import JointsMSELoss as RefLoss
    from oracle import recipes, core as oc
    from oracle.cfg import Cfg

    class Writer:
        def __init__(self):
            self.scalars = []

        def add_scalar(self, k, v, s):
            self.scalars.append((k, float(v), s))

        def add_scalars(self, k, d, s):
            self.scalars.append((k, dict(d), s))

    class Dataset:
        def __init__(self, n, image_size):
            self.n, self.image_size = n, np.array(image_size)
            self.flip_pairs, self.kpt_colors = oc.CROWDPOSE_FLIP_PAIRS, oc.CROWDPOSE_KPT_COLORS
            self.captured = None

        def __len__(self):
            return self.n

        def get_condition_image_colored(self, kpts, size, colors):
            return oc.get_condition_image_colored(kpts, size, colors)

        def evaluate(self, cfg, preds, output_dir, all_boxes, img_path, *a, **k):
            self.captured = (preds.copy(), all_boxes.copy(), list(img_path))
            return {"AP": 0.5, "AP .5": 0.75}, 0.5

    rec = {}
    # ---- train(): 3 iterations, Adam lr 1e-3, dropout 0 --------------------------------------------------------------
    cfg, omodel, _, _ = recipes.build("coam_w16_96x64_colored")
    cfg.PRINT_FREQ = 1
    rmodel = ref_model_for("coam", cfg)
    rmodel.load_state_dict(omodel.state_dict(), strict=True)
    recipes.set_dropout(rmodel, 0.0)
    loader = entry_batches(cfg, 3, 2, align=(omodel, True))
    opt = torch.optim.Adam(rmodel.parameters(), lr=1e-3)
    wd = {"writer": Writer(), "train_global_steps": 0}
    rf.train(cfg, loader, rmodel, RefLoss(True), opt, 1, "/tmp", "/tmp", wd)
    assert wd["train_global_steps"] == 3
    rec["train_loss"] = np.array([v for k, v, _ in wd["writer"].scalars if k == "train_loss"])
    rec["train_acc"] = np.array([v for k, v, _ in wd["writer"].scalars if k == "train_acc"])
    sd = rmodel.state_dict()
    names = [k for k, _ in rmodel.named_parameters()]
    rec["param_names"] = np.array(names)
    rec["param_sums"] = np.array([sd[k].double().sum().item() for k in names])
    rec["param_norms"] = np.array([sd[k].double().norm().item() for k in names])
    for k in ("final_layer.weight", "conv1.weight"):
        rec["param::" + k] = sd[k].numpy()
    bnames = sorted(k for k in sd if k.endswith("running_mean") or k.endswith("running_var"))
    rec["buf_names"] = np.array(bnames)
    rec["buf_norms"] = np.array([sd[k].double().norm().item() for k in bnames])
    rec["num_batches_tracked"] = np.int64(int(sd["bn1.num_batches_tracked"]))
    rec["state_sha_before"] = np.array(state_digest(omodel))
    print(f"  entry: reference train() 3 iterations, losses {rec['train_loss']}, acc {rec['train_acc']}")
    # ---- validate(): flip test off / on, colored and mono condition -----------------------------------------------------
    for tag, recipe, flip in (("val_colored", "coam_w16_96x64_colored", False), ("val_colored_flip", "coam_w16_96x64_colored", True),
                              ("val_mono_flip", "coam_w16_96x64_mono_default_att", True)):
        cfg, omodel, _, _ = recipes.build(recipe)
        cfg.PRINT_FREQ = 1
        cfg.TEST = Cfg({"FLIP_TEST": flip, "POST_PROCESS": True, "SHIFT_HEATMAP": True})
        rmodel = ref_model_for("coam", cfg)
        rmodel.load_state_dict(omodel.state_dict(), strict=True)
        # mono: the ONE blurred, int-truncated channel replicated x3 (JointsDataset.py:500-516); under FLIP_TEST the
        # reference re-renders it COLORED (transforms.py:38-47, heatmap.shape[1] == 3) - the goldens pin that behaviour
        loader = entry_batches(cfg, 2, 2, seed0=700, cond_channels=1 if "mono" in recipe else 3, align=(omodel, False))
        ds = Dataset(4, [64, 96])
        wd = {"writer": Writer(), "valid_global_steps": 0}
        perf = rf.validate(cfg, loader, ds, rmodel, RefLoss(True), "/tmp", "/tmp", wd)
        assert perf == 0.5 and wd["valid_global_steps"] == 1
        preds, boxes, paths = ds.captured
        rec[tag + "_preds"], rec[tag + "_boxes"], rec[tag + "_paths"] = preds, boxes, np.array(paths)
        rec[tag + "_loss"] = np.float64([v for k, v, _ in wd["writer"].scalars if k == "valid_loss"][0])
        rec[tag + "_acc"] = np.float64([v for k, v, _ in wd["writer"].scalars if k == "valid_acc"][0])
        rec[tag + "_x_sha"] = np.array(sha(torch.cat([b[0] for b in loader])))
        print(f"  entry: reference validate() [{tag}] loss {rec[tag + '_loss']:.6f} acc {rec[tag + '_acc']:.3f}")
    np.savez_compressed(os.path.join(OUT, "entry.npz"), **rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fast", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    assert os.path.isdir(REF), "the reference checkout is required (build container only)"
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "lib"))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(8)
    from oracle import recipes
    print("pinning oracle against", REF)
    if not args.only:
        core_cases()
        target_case()
        nms_case()
        pose_synthesis_case()
        entry_case()
    elif args.only == "entry":
        entry_case()
        return
    small = ["prenet_w16_96x64", "coam_w16_96x64_colored", "coam_w16_96x64_mono_default_att",
             "coam_w16_96x64_stacked_2heads", "coam_w16_96x64_channel_only", "coam_w16_96x64_selfatt", "transpose_w16_96x64",
             "resnet18_96x64"]
    full = ["coam_w48_384x288", "prenet_w32_256x192", "resnet50_256x192", "prenet_w48_384x288", "transpose_a6_256x192"]
    for name in small + ([] if args.fast else full):
        if args.only and args.only != name:
            continue
        model_case(name, train=name in small)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
