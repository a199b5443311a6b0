"""train() / validate() entry points (reference lib/core/function.py:102-175, 178-336) driven end to end on the GPU
with an in-memory loader and a fake dataset, checked against the CPU oracle running the same loop semantics:
per-batch loss values, final parameters after Adam steps, the decoded all_preds handed to dataset.evaluate and
the flip-test path (device flip-merge + GPU re-rendered colored condition)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class Writer:
    def __init__(self):
        self.scalars = []

    def add_scalar(self, k, v, s):
        self.scalars.append((k, float(v), s))

    def add_scalars(self, k, d, s):
        self.scalars.append((k, dict(d), s))


class FakeDataset:
    def __init__(self, n, image_size, flip_pairs, colors):
        self.n, self.image_size, self.flip_pairs, self.kpt_colors = n, image_size, flip_pairs, colors
        self.captured = None

    def __len__(self):
        return self.n

    def evaluate(self, cfg, preds, output_dir, all_boxes, img_path, *args, **kwargs):
        self.captured = (preds.copy(), all_boxes.copy(), list(img_path))
        return {"AP": 0.5, "AP .5": 0.75}, 0.5


def _cfg_for(train_cfg, flip):
    from buctd_amd.config import cfg as base, hrnet_extra
    c = base.clone()
    c.defrost()
    c.MODEL.NAME = "pose_hrnet_coam"
    c.MODEL.NUM_JOINTS = 14
    c.MODEL.IMAGE_SIZE = [64, 96]
    c.MODEL.HEATMAP_SIZE = [16, 24]
    c.MODEL.SIGMA = 2
    c.MODEL.ATT_MODULES = [False, True, False, False]
    c.MODEL.CONDITIONAL_TOPDOWN = True
    c.MODEL.EXTRA = hrnet_extra(16, use_attention=True, modules=(1, 2, 2))
    c.DATASET.COLORED = True
    c.TRAIN.LR = 1e-3
    c.PRINT_FREQ = 1
    c.TEST.FLIP_TEST = flip
    c.TEST.POST_PROCESS = True
    c.TEST.SHIFT_HEATMAP = True
    c.freeze()
    return c


def _batches(cfg, n_batches, batch):
    from oracle import recipes, core as oc
    out = []
    for i in range(n_batches):
        x, joints = recipes.make_inputs(cfg, batch, 500 + i, 3)
        tgt, wt = recipes.make_targets(cfg, joints, 600 + i)
        meta = {"center": torch.rand(batch, 2) * 100 + 50, "scale": torch.rand(batch, 2) + 0.5,
                "score": torch.rand(batch), "annotation_id": torch.arange(batch) + i * batch,
                "image": [f"img_{i}_{j}.jpg" for j in range(batch)],
                "cond_joints": torch.cat([joints, torch.zeros(batch, 14, 1)], 2),
                "cond_joints_vis": torch.ones(batch, 14, 3)}
        out.append((x, tgt, wt, meta))
    return out


def test_train_entry_point_matches_oracle_loop(dev):
    from oracle import recipes, core as oc
    from buctd_amd import models, engine
    from buctd_amd.core.function import train
    from buctd_amd.core.loss import JointsMSELoss
    cfg = _cfg_for(True, False)
    _, omodel, _, _ = recipes.build("coam_w16_96x64_colored")
    net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)
    net.load_state_dict(omodel.state_dict(), strict=True)
    model = engine.DataParallel(net).cuda()
    optimizer = engine.get_optimizer(cfg, model)
    recipes.set_dropout(model, 0.0)
    loader = _batches(cfg, 3, 2)
    wd = {"writer": Writer(), "train_global_steps": 0}
    train(cfg, loader, model, JointsMSELoss(True).cuda(), optimizer, 1, "/tmp", "/tmp", wd)
    assert wd["train_global_steps"] == 3
    losses = [v for k, v, _ in wd["writer"].scalars if k == "train_loss"]
    # oracle: same loop with torch.optim.Adam
    om = copy.deepcopy(omodel).train()
    recipes.set_dropout(om, 0.0)
    opt = torch.optim.Adam(om.parameters(), lr=1e-3)
    ref_losses = []
    for x, tgt, wt, _ in loader:
        loss = oc.JointsMSELoss(True)(om(x), tgt, wt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    assert abs(losses[0] - ref_losses[0]) <= 1e-4 * abs(ref_losses[0]), (losses, ref_losses)
    # later steps go through Adam's sign-like first update (m/sqrt(v) ~ +-1): tiny gradient noise moves weights by lr,
    # so trajectories agree to a few percent, not to round-off
    for a, b in zip(losses[1:], ref_losses[1:]):
        assert abs(a - b) <= 5e-2 * abs(b), (losses, ref_losses)
    sd = model.module.state_dict()
    ref_sd = om.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    assert int(sd["bn1.num_batches_tracked"]) == 3
    pnames = {k for k, _ in om.named_parameters()}
    worst = max(((float((sd[k].cpu() - ref_sd[k]).abs().max()), k) for k in pnames), key=lambda t: t[0])
    # each Adam step moves a weight by at most ~lr; 3 steps with possibly opposite signs -> <= 2*3*lr
    assert worst[0] <= 6.5e-3, f"parameter {worst[1]} drifted {worst[0]} after 3 Adam steps (lr 1e-3)"
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            a, b = sd[k].cpu(), ref_sd[k]
            assert float((a - b).abs().max()) <= 0.1 * max(1.0, float(b.abs().max())), f"buffer {k}"  # trajectories diverge by ~lr per Adam step
    assert list(model.state_dict().keys())[0].startswith("module.")


@pytest.mark.parametrize("flip", [False, True])
def test_validate_entry_point_matches_oracle_loop(dev, flip):
    from oracle import recipes, core as oc
    from buctd_amd import models
    from buctd_amd.core.function import validate
    from buctd_amd.core.loss import JointsMSELoss
    cfg = _cfg_for(False, flip)
    _, omodel, _, _ = recipes.build("coam_w16_96x64_colored")
    net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)
    net.load_state_dict(omodel.state_dict(), strict=True)
    net = net.cuda()
    loader = _batches(cfg, 2, 2)
    ds = FakeDataset(4, [64, 96], oc.CROWDPOSE_FLIP_PAIRS, oc.CROWDPOSE_KPT_COLORS)
    wd = {"writer": Writer(), "valid_global_steps": 0}
    perf = validate(cfg, loader, ds, net, JointsMSELoss(True).cuda(), "/tmp", "/tmp", wd)
    assert perf == 0.5 and wd["valid_global_steps"] == 1
    preds, boxes, paths = ds.captured
    assert paths == [f"img_{i}_{j}.jpg" for i in range(2) for j in range(2)]
    # oracle loop (function.py:178-270 semantics)
    omodel.eval()
    idx = 0
    for x, tgt, wt, meta in loader:
        with torch.no_grad():
            out = omodel(x).numpy()
            if flip:
                conds = []
                for b in range(x.shape[0]):
                    fj, _ = oc.fliplr_joints(meta["cond_joints"][b].numpy(), meta["cond_joints_vis"][b].numpy(), 64,
                                              oc.CROWDPOSE_FLIP_PAIRS)
                    conds.append(oc.get_condition_image_colored(fj, (96, 64, 3), oc.CROWDPOSE_KPT_COLORS).transpose(2, 0, 1))
                xf = torch.cat([x[:, :3].flip(3), torch.from_numpy(np.stack(conds)).float()], 1)
                out = oc.flip_test_merge(out, omodel(xf).numpy(), oc.CROWDPOSE_FLIP_PAIRS, True)
        fp, mv = oc.get_final_preds(True, out, meta["center"].numpy(), meta["scale"].numpy())
        n = x.shape[0]
        # decoded arg-max positions must agree exactly except where two heat-map values tie within fp32 noise;
        # final coordinates are in image pixels (scale ~ 200 * s / 16 per heat-map pixel)
        assert np.abs(preds[idx:idx + n, :, :2] - fp).max() <= 1e-2, "final preds differ"
        assert np.abs(preds[idx:idx + n, :, 2:3] - mv).max() <= 1e-3 * max(1.0, np.abs(mv).max())
        assert np.allclose(boxes[idx:idx + n, 0:2], meta["center"].numpy())
        assert np.allclose(boxes[idx:idx + n, 4], np.prod(meta["scale"].numpy() * 200, 1))
        assert np.allclose(boxes[idx:idx + n, 6], meta["annotation_id"].numpy())
        idx += n


# ---- against the REFERENCE's own entry points (tests/golden/entry.npz, written by oracle/make_golden.py:entry_case, which
# ---- imports and runs lib/core/function.py:train / validate of the reference on the same seeded batches) -----------------
def _golden_entry():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "entry.npz"))


def test_train_entry_point_matches_reference_train(dev):
    from oracle import recipes
    from oracle.make_golden import entry_batches
    from buctd_amd import models, engine
    from buctd_amd.core.function import train
    from buctd_amd.core.loss import JointsMSELoss
    gold = _golden_entry()
    cfg = _cfg_for(True, False)
    ocfg, omodel, _, _ = recipes.build("coam_w16_96x64_colored")
    # (the recipe calibrates the BatchNorm statistics on the CPU: round-off level differences between hosts, so the initial
    # state is rebuilt here rather than compared bit for bit with the golden's digest)
    net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)
    net.load_state_dict(omodel.state_dict(), strict=True)
    model = engine.DataParallel(net).cuda()
    optimizer = engine.get_optimizer(cfg, model)
    recipes.set_dropout(model, 0.0)
    loader = entry_batches(ocfg, 3, 2, align=(omodel, True))
    wd = {"writer": Writer(), "train_global_steps": 0}
    train(cfg, loader, model, JointsMSELoss(True).cuda(), optimizer, 1, "/tmp", "/tmp", wd)
    losses = np.array([v for k, v, _ in wd["writer"].scalars if k == "train_loss"])
    accs = np.array([v for k, v, _ in wd["writer"].scalars if k == "train_acc"])
    ref = gold["train_loss"]
    assert abs(losses[0] - ref[0]) <= 1e-4 * abs(ref[0]), (losses, ref)
    # Adam's first updates are sign-like (m / sqrt(v) ~ +-1): round-off level gradient noise moves weights by lr, so the
    # trajectories agree to a few percent after the first step, not to round-off (same bar as the oracle-loop test)
    assert np.all(np.abs(losses[1:] - ref[1:]) <= 5e-2 * np.abs(ref[1:])), (losses, ref)
    # the first accuracy is that of the unchanged initial weights: it is pinned exactly, at a non-trivial value (the even
    # joints' targets sit at the net's own arg-max, oracle/make_golden.py:entry_batches); after the sign-like Adam steps a
    # borderline joint may fall on the other side of PCK's threshold (1 of ~26 counted joints = 0.04)
    assert 0.2 <= gold["train_acc"][0] <= 0.8 and accs[0] == gold["train_acc"][0], (accs, gold["train_acc"])
    assert np.all(np.abs(accs[1:] - gold["train_acc"][1:]) <= 0.08), (accs, gold["train_acc"])
    sd = model.module.state_dict()
    assert int(sd["bn1.num_batches_tracked"]) == int(gold["num_batches_tracked"]) == 3
    assert [k for k, _ in model.module.named_parameters()] == list(gold["param_names"])
    for k in ("final_layer.weight", "conv1.weight"):
        assert float((sd[k].cpu() - torch.from_numpy(gold["param::" + k])).abs().max()) <= 6.5e-3, k
    norms = np.array([float(sd[k].double().norm()) for k in gold["param_names"]])
    # every weight moves by at most ~lr per step: ||w - w_ref|| <= 6 lr sqrt(numel)
    numel = np.array([sd[k].numel() for k in gold["param_names"]])
    assert np.all(np.abs(norms - gold["param_norms"]) <= 6.5e-3 * np.sqrt(numel) + 1e-6)
    bn = np.array([float(sd[k].double().norm()) for k in gold["buf_names"]])
    assert np.all(np.abs(bn - gold["buf_norms"]) <= 0.1 * np.maximum(1.0, gold["buf_norms"]))


@pytest.mark.parametrize("tag", ["val_colored", "val_colored_flip", "val_mono_flip"])
def test_validate_entry_point_matches_reference_validate(dev, tag):
    """all_preds / all_boxes / image paths / meters of the reference's validate(): flip test off and on, colored condition
    and the mono condition (whose flipped twin the reference re-renders COLORED, transforms.py:38-47)."""
    from oracle import recipes, core as oc
    from oracle.make_golden import entry_batches
    from buctd_amd import models
    from buctd_amd.core.function import validate
    from buctd_amd.core.loss import JointsMSELoss
    gold = _golden_entry()
    mono = "mono" in tag
    recipe = "coam_w16_96x64_mono_default_att" if mono else "coam_w16_96x64_colored"
    ocfg, omodel, _, _ = recipes.build(recipe)
    cfg = _cfg_for(False, tag.endswith("flip"))
    if mono:
        from buctd_amd.config import hrnet_extra
        cfg.defrost()
        cfg.MODEL.ATT_MODULES = [False, False, True, True]
        cfg.MODEL.EXTRA = hrnet_extra(16, use_attention=True, modules=(1, 1, 1))
        cfg.DATASET.COLORED = False
        cfg.freeze()
    net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)
    net.load_state_dict(omodel.state_dict(), strict=True)
    net = net.cuda()
    loader = entry_batches(ocfg, 2, 2, seed0=700, cond_channels=1 if mono else 3, align=(omodel, False))
    ds = FakeDataset(4, [64, 96], oc.CROWDPOSE_FLIP_PAIRS, oc.CROWDPOSE_KPT_COLORS)
    wd = {"writer": Writer(), "valid_global_steps": 0}
    perf = validate(cfg, loader, ds, net, JointsMSELoss(True).cuda(), "/tmp", "/tmp", wd)
    assert perf == 0.5 and wd["valid_global_steps"] == 1
    preds, boxes, paths = ds.captured
    assert paths == list(gold[tag + "_paths"])
    assert np.allclose(boxes, gold[tag + "_boxes"], rtol=1e-6, atol=1e-6)
    rp = gold[tag + "_preds"]
    # final coordinates are image pixels (~ 200 * scale / 16 per heat-map pixel): identical arg-max, quarter-pixel
    # refinement decided by heat-map differences that are far from zero
    assert np.abs(preds[:, :, :2] - rp[:, :, :2]).max() <= 1e-2, "final preds differ from the reference's"
    assert np.abs(preds[:, :, 2] - rp[:, :, 2]).max() <= 1e-3 * max(1.0, np.abs(rp[:, :, 2]).max())
    vl = [v for k, v, _ in wd["writer"].scalars if k == "valid_loss"][0]
    va = [v for k, v, _ in wd["writer"].scalars if k == "valid_acc"][0]
    assert abs(vl - float(gold[tag + "_loss"])) <= 1e-4 * float(gold[tag + "_loss"])
    assert va == float(gold[tag + "_acc"])


def test_flip_test_as_one_forward_equals_two_forwards(dev):
    """validate() runs the flip test (reference function.py:213-236) as ONE forward over [crops | mirrored crops]: eval-mode
    networks treat every image on its own, so the prediction table must equal the one from two forwards (to the rounding of
    a split-K GEMM whose split count follows the batch: the channel-attention logits)."""
    from oracle import recipes, core as oc
    from buctd_amd import models
    from buctd_amd.core import function
    from buctd_amd.core.loss import JointsMSELoss
    cfg = _cfg_for(False, True)
    _, omodel, _, _ = recipes.build("coam_w16_96x64_colored")
    net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)
    net.load_state_dict(omodel.state_dict(), strict=True)
    net = net.cuda()
    loader = _batches(cfg, 2, 2)
    tables = []
    old = function.PAIRED_FLIP_FORWARD
    try:
        for paired in (False, True):
            function.PAIRED_FLIP_FORWARD = paired
            ds = FakeDataset(4, [64, 96], oc.CROWDPOSE_FLIP_PAIRS, oc.CROWDPOSE_KPT_COLORS)
            function.validate(cfg, loader, ds, net, JointsMSELoss(True).cuda(), "/tmp", "/tmp", None)
            tables.append(ds.captured[0])
    finally:
        function.PAIRED_FLIP_FORWARD = old
    assert np.abs(tables[0][:, :, :2] - tables[1][:, :, :2]).max() <= 1e-3      # image pixels
    assert np.abs(tables[0][:, :, 2] - tables[1][:, :, 2]).max() <= 1e-5


def test_train_with_device_prefetch_equals_the_plain_loop(dev):
    """train() copies batch i + 1 to the device beside step i (core.function.DevicePrefetch, on the weight-gradient stream);
    the reference loop copies at the top of the iteration (function.py:118-125).  Same batches, same order: the logged losses
    and the parameters after the epoch are bit-identical, with pinned and with pageable host batches."""
    import copy as _copy
    from oracle import recipes
    from buctd_amd import models, engine
    from buctd_amd.core import function
    from buctd_amd.core.loss import JointsMSELoss
    cfg = _cfg_for(True, False)
    _, omodel, _, _ = recipes.build("coam_w16_96x64_colored")
    results = []
    old = function.PREFETCH_TO_DEVICE
    try:
        for prefetch, pinned in ((False, False), (True, False), (True, True)):
            function.PREFETCH_TO_DEVICE = prefetch
            net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=False)
            net.load_state_dict(omodel.state_dict(), strict=True)
            model = engine.DataParallel(net).cuda()
            optimizer = engine.get_optimizer(cfg, model)
            recipes.set_dropout(model, 0.0)
            loader = _batches(cfg, 4, 2)
            if pinned:
                loader = [(x.pin_memory(), t.pin_memory(), w.pin_memory(), m) for x, t, w, m in loader]
            wd = {"writer": Writer(), "train_global_steps": 0}
            function.train(cfg, loader, model, JointsMSELoss(True).cuda(), optimizer, 1, "/tmp", "/tmp", wd)
            losses = [v for k, v, _ in wd["writer"].scalars if k == "train_loss"]
            results.append((losses, optimizer.flat.flat.detach().clone()))
    finally:
        function.PREFETCH_TO_DEVICE = old
    for losses, params in results[1:]:
        assert losses == results[0][0]
        assert torch.equal(params, results[0][1])
