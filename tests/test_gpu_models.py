"""Whole-network parity on the MI355X: buctd_amd models (HIP kernels through the C ABI) against
(a) the committed golden vectors produced by the REFERENCE itself (tests/golden/model_*.npz, made by
oracle/make_golden.py in the build container) and (b) the CPU oracle rebuilt here from the same seed.

Bars (north_star): eval heat-maps within an ABSOLUTE 1e-3 of the reference forward on unit-scale heat-maps (every
recipe scales its final layer by a fixed power of two so that max|y| is in [0.5, 1], oracle/recipes.py) and identical
arg-max decode indices - in the engine's default math mode (bf16x6, fp32-class) and in the exact fp32 mode; every
BASELINE config (C1-C5) at full size; train step: loss within 1e-4 relative, gradients as close to an fp64 evaluation
as the fp32 CPU path is, BN running statistics within 1e-4.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SMALL = ["prenet_w16_96x64", "coam_w16_96x64_colored", "coam_w16_96x64_mono_default_att",
         "coam_w16_96x64_stacked_2heads", "coam_w16_96x64_channel_only", "coam_w16_96x64_selfatt", "transpose_w16_96x64",
         "resnet18_96x64"]
# BASELINE.json configs at full size: C4, C2, C1, C3, C5
FULL = ["coam_w48_384x288", "prenet_w32_256x192", "resnet50_256x192", "prenet_w48_384x288", "transpose_a6_256x192"]
BAR = 1e-3     # north_star: heat-maps within 1e-3 (fp32) of the reference forward - absolute, on unit-scale heat-maps


@pytest.fixture(params=["bf16x6", "fp32"])
def math_mode(request):
    from buctd_amd import ops
    old = ops.get_conv_math()
    ops.set_conv_math(request.param)
    yield request.param
    ops.set_conv_math(old)


def product_model(cfg, oracle_model, dev):
    from buctd_amd import models
    mod = getattr(models, cfg.MODEL.NAME)
    m = mod.get_pose_net(cfg, is_train=False)
    missing = m.load_state_dict(oracle_model.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(dev)


def _oracle_grads(omodel, x, tgt, wt, dtype):
    import copy
    from oracle import recipes, core as ocore
    m = copy.deepcopy(omodel).to(dtype).train()
    recipes.set_dropout(m, 0.0)
    loss = ocore.JointsMSELoss(True)(m(x.to(dtype)), tgt.to(dtype), wt.to(dtype))
    loss.backward()
    return {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None}


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


@pytest.mark.parametrize("name", SMALL)
def test_eval_forward_matches_reference(dev, math_mode, name):
    from oracle import recipes
    gold = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    cfg, omodel, x, _ = recipes.build(name)
    with torch.no_grad():
        y_or = omodel(x)
    # the oracle rebuilt from the seed reproduces what the reference produced in the build container
    top = float(np.abs(gold["out"]).max())
    assert 0.4 <= top <= 1.0, f"recipe {name} is not unit scale: max|y| = {top}"
    drift = np.abs(y_or.numpy() - gold["out"]).max()
    assert drift <= 2e-5, f"seeded recipe drifted from the golden vector: {drift:.3e}"
    m = product_model(cfg, omodel, dev).eval()
    with torch.no_grad():
        y = m(x.to(dev))
    assert y.shape == y_or.shape and y.is_contiguous()
    err = np.abs(y.cpu().numpy() - gold["out"]).max()
    err_or = np.abs(y.cpu().numpy() - y_or.numpy()).max()
    print(f"{name} [{math_mode}]: |hip - reference| = {err:.3e}, |hip - oracle| = {err_or:.3e}, max|y| {top:.3f}")
    assert err <= BAR, f"{name}: eval heat-maps differ from the reference by {err:.3e}"
    assert err_or <= BAR, f"{name}: eval heat-maps differ from the oracle by {err_or:.3e}"
    idx = y.reshape(y.shape[0], y.shape[1], -1).argmax(2).cpu().numpy()
    assert np.array_equal(idx, y_or.reshape(y.shape[0], y.shape[1], -1).argmax(2).numpy()), f"{name}: arg-max vs oracle"
    assert np.array_equal(idx, gold["argmax"]), f"{name}: arg-max decode indices differ from the reference"
    # CPU input is accepted like in the reference (forward calls x.cuda())
    with torch.no_grad():
        y2 = m(x)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("name", SMALL)
def test_train_step_matches_reference(dev, name):
    from oracle import recipes
    from buctd_amd.core.loss import JointsMSELoss
    gold = np.load(os.path.join(GOLD, f"model_{name}.npz"), allow_pickle=False)
    cfg, omodel, x, joints = recipes.build(name)
    tgt, wt = recipes.make_targets(cfg, joints, 77)
    m = product_model(cfg, omodel, dev).train()
    recipes.set_dropout(m, 0.0)
    y = m(x.to(dev))
    loss = JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev))
    loss.backward()
    assert np.abs(y.detach().cpu().numpy() - gold["train_out"]).max() <= BAR * max(1.0, float(np.abs(gold["train_out"]).max()))
    assert rel(loss.item(), float(gold["loss"])) <= 1e-4, (loss.item(), float(gold["loss"]))
    names = [str(s) for s in gold["grad_names"]]
    params = dict(m.named_parameters())
    # fp64 evaluation of the same step on the CPU oracle = ground truth; the fp32 oracle's own distance to it
    # is the yardstick for what fp32 round-off does to this (BN-with-tiny-batch, ~60 layers deep) backward pass
    g64, g32 = _oracle_grads(omodel, x, tgt, wt, torch.float64), _oracle_grads(omodel, x, tgt, wt, torch.float32)
    # These randomly initialised nets are ill-conditioned in train mode (BatchNorm over a few dozen samples in the
    # low-resolution branches, ~60 layers deep): the fp32 CPU oracle itself sits 1e-3..5e-2 from its fp64 evaluation
    # (measured, see DESIGN.md "parity"), so a per-tensor bit-level bar is meaningless here.  The bar is statistical:
    # the HIP path must be as close to the fp64 truth as the fp32 CPU path is.  (Tight per-module bars: test_gpu_blocks.py)
    gmax = max(v.norm().item() for v in g64.values())
    e_hip, e_cpu = [], []
    for k, gn in zip(names, gold["grad_norms"]):
        g = params[k].grad
        assert g is not None, f"{k}: no gradient"
        den = g64[k].norm().item()
        if den <= 1e-6 * gmax:  # mathematically-zero gradients (conv bias in front of a BatchNorm)
            assert g.norm().item() <= 1e-4 * gmax, f"{name}: {k} should have a ~zero gradient"
            continue
        e_hip.append((g.detach().cpu().double() - g64[k]).norm().item() / den)
        e_cpu.append((g32[k].double() - g64[k]).norm().item() / den)
        assert abs(g.norm().item() - gn) <= 5e-2 * max(gn, 1e-6), \
            f"{name}: grad norm of {k}: {g.norm().item()} vs reference {gn}"
    if name.startswith("transpose"):
        # The encoder (LayerNorm, in/out projections, FFN, attention soft-max) sits downstream of every ReLU of the trunk
        # in the backward pass, so its gradients are free of the ReLU-flip chaos described below: they must be as
        # accurate as the fp32 CPU path.  (Round 1 saw "40x worse than CPU fp32" on this network: scratch/diag_transpose.py
        # localises it to ONE BatchNorm layer of the HRNet trunk whose position changes with the input seed - a flipped
        # ReLU - while every encoder tensor sits at 3e-6..2e-5 against the CPU's 1e-6..6e-6.)
        kept = [k for k, gn in zip(names, gold["grad_norms"]) if g64[k].norm().item() > 1e-6 * gmax]
        for k, eh, ec in zip(kept, e_hip, e_cpu):
            if k.startswith(("global_encoder", "final_layer", "reduce", "trans_cond")):
                assert eh <= 5 * ec + 2e-5, f"{name}: encoder gradient {k}: hip {eh:.2e} vs fp32 CPU {ec:.2e}"
    med_h, med_c = float(np.median(e_hip)), float(np.median(e_cpu))
    worst, worst_ref = max(e_hip), max(e_cpu)
    # A single ReLU whose pre-activation sits within fp32 round-off of zero flips between implementations and shifts
    # every upstream gradient by ~1e-3 (seen on both the HIP and the CPU fp32 side, channel-localised - DESIGN.md);
    # the floors below allow for such flips, a wiring / scaling bug shows up as >= 1e-1.
    if name == "coam_w16_96x64_channel_only":
        # a well-conditioned whole-network case (recipe seed chosen for it, oracle/recipes.py:SEEDS): the fp32 CPU oracle is
        # 2e-5 from fp64, so the bulk of the HIP gradients carries a tight bar.  (The worst tensors do not: a single ReLU of
        # stage3 branch 0 still flips between summation orders - both HIP math modes, not the CPU - and moves the
        # gradients of the four blocks in front of it by 5e-3..2e-2, scratch/diag_channel_only.py.)
        assert med_c <= 1e-4, "the recipe is meant to be well conditioned"
        assert med_h <= 3e-4, f"{name}: tight bar on the median: {med_h:.2e}"
    kept_all = [k for k in names if g64[k].norm().item() > 1e-6 * gmax]
    worst_k = kept_all[int(np.argmax(e_hip))]

    def bars(mh, mc, wh, wc):
        # same floors as the full-size train-step tests (2e-3 on the median, 2e-2 on the worst tensor)
        return mh <= max(3 * mc, 2e-3) and wh <= max(3 * wc, 2e-2)

    bufs = {k: v.detach().clone() for k, v in m.named_buffers()}       # (the check below may run more train-mode steps)
    gsnap = {k: params[k].grad.detach().clone() for k in ("final_layer.weight", "conv1.weight") if k in params}
    if not bars(med_h, med_c, worst, worst_ref):
        # A flipped ReLU is a property of ONE input (which pre-activation happens to sit within round-off of zero); a wiring
        # or scaling error is not.  The recipe's own input missed the bars: the SAME weights must then meet them on two other
        # seeded inputs - otherwise this is a bug, and the message names the tensor.  (No reseeding of the goldens: the
        # recipe's seed stays what it is, oracle/recipes.py.)
        others = []
        for alt in (1, 2):
            ga = torch.Generator().manual_seed(9000 + alt)
            xa = x + 0.25 * torch.randn(x.shape, generator=ga) * (torch.arange(x.shape[1]).view(1, -1, 1, 1) < 3)   # RGB only
            for q in m.parameters():
                q.grad = None
            la = JointsMSELoss(True)(m(xa.to(dev)), tgt.to(dev), wt.to(dev))
            la.backward()
            a64, a32 = _oracle_grads(omodel, xa, tgt, wt, torch.float64), _oracle_grads(omodel, xa, tgt, wt, torch.float32)
            amax = max(v.norm().item() for v in a64.values())
            eh, ec = [], []
            for k in names:
                den = a64[k].norm().item()
                if den <= 1e-6 * amax:
                    continue
                eh.append((params[k].grad.detach().cpu().double() - a64[k]).norm().item() / den)
                ec.append((a32[k].double() - a64[k]).norm().item() / den)
            others.append((float(np.median(eh)), float(np.median(ec)), max(eh), max(ec)))
        print(f"{name}: recipe input misses the bars (median hip {med_h:.2e} / cpu32 {med_c:.2e}, worst hip {worst:.2e} at "
              f"{worst_k} / cpu32 {worst_ref:.2e}); two other inputs: {others}")
        # where the error ORIGINATES: walking the parameters in backward order (last registered = closest to the loss), the
        # first tensor beyond the bar names the layer behind which a unit flipped; everything between it and the loss must be
        # clean - an error that is already in the head's gradient is not a flip inside the trunk
        order = list(reversed(list(zip(kept_all, e_hip, e_cpu))))
        onset = next((i for i, (_, eh_, ec_) in enumerate(order) if eh_ > max(3 * ec_, 1e-3)), None)
        if onset is not None:
            print(f"{name}: first tensor in backward order beyond the bar: {order[onset][0]} ({order[onset][1]:.2e}); "
                  f"{onset} tensors between it and the loss are clean")
            assert onset > 0, f"{name}: the gradient of the last layer ({order[0][0]}) is already off by {order[0][1]:.2e}"
        assert all(bars(*o) for o in others), \
            (f"{name}: grad error vs fp64 beyond the bars on the recipe input (median hip {med_h:.2e} / cpu32 {med_c:.2e}, worst "
             f"hip {worst:.2e} at {worst_k} / cpu32 {worst_ref:.2e}) AND on other inputs {others}: not a flipped ReLU")
        # the errors of a flip stay an order of magnitude under those of a wiring error (>= 1e-1 on the affected tensors)
        assert med_h <= 2e-2 and worst <= 1e-1, f"{name}: median {med_h:.2e} / worst {worst:.2e} at {worst_k}"
    print(f"{name}: grad rel err vs fp64 - median hip {med_h:.2e} / cpu32 {med_c:.2e}; max hip {worst:.2e} / cpu32 {worst_ref:.2e}")
    for k in ("final_layer.weight", "conv1.weight"):
        key = "grad::" + k
        if key in gold.files:
            g = gsnap[k].cpu().numpy()
            ref = gold[key]
            assert np.abs(g - ref).max() <= 5e-2 * max(1.0, np.abs(ref).max()), f"{name}: full grad {k}"
    for k, bn in zip([str(s) for s in gold["buf_names"]], gold["buf_norms"]):
        assert abs(bufs[k].norm().item() - bn) <= 1e-4 * max(bn, 1.0), f"{name}: buffer {k}"


def _grad_errors(m, omodel, xin, tgt, wt, dev):
    """(median hip, median cpu32, worst hip, worst cpu32, worst tensor) of the parameter gradients of one train step against
    the fp64 evaluation of the oracle."""
    from buctd_amd.core.loss import JointsMSELoss
    for q in m.parameters():
        q.grad = None
    JointsMSELoss(True)(m(xin.to(dev)), tgt.to(dev), wt.to(dev)).backward()
    g64, g32 = _oracle_grads(omodel, xin, tgt, wt, torch.float64), _oracle_grads(omodel, xin, tgt, wt, torch.float32)
    gmax = max(v.norm().item() for v in g64.values())
    params = dict(m.named_parameters())
    eh, ec, keys = [], [], []
    for k, ref in g64.items():
        den = ref.norm().item()
        if den <= 1e-6 * gmax:
            continue
        eh.append((params[k].grad.detach().cpu().double() - ref).norm().item() / den)
        ec.append((g32[k].double() - ref).norm().item() / den)
        keys.append(k)
    order = list(reversed(list(zip(keys, eh, ec))))      # backward order: last registered parameter = closest to the loss
    onset = next((i for i, (_, a, b) in enumerate(order) if a > max(3 * b, 1e-3)), None)
    _grad_errors.onset = None if onset is None else (onset, order[onset][0], order[onset][1], order[0][0])
    return float(np.median(eh)), float(np.median(ec)), max(eh), max(ec), keys[int(np.argmax(eh))]


# (the TransPose case spends 30 s in the CPU oracle's noise copies: with --runslow; the suite has to stay well inside the
# driver's 1200 s on a box whose CPUs are shared - 568 s on a box at load 35-58, 1330 s measured on a busier one)
@pytest.mark.parametrize("name", ["coam_w16_96x64_mono_default_att",
                                  pytest.param("transpose_w16_96x64", marks=pytest.mark.slow)])
def test_train_step_at_the_original_seed_differs_by_a_relu_flip_only(dev, name):
    """The goldens of these two nets use recipe seeds other than 1234 (oracle/recipes.py:SEEDS) because at 1234 one
    pre-activation sits within fp32 round-off of zero and lands on different sides of its ReLU in different implementations.
    The original seed stays a case, under the metric that tells a flip from a bug: an arithmetic, wiring or scaling defect does
    not care about 1e-6 of relative noise on the input, a flip does.  scratch/flip_probe.py on these nets (3x2-pixel branches,
    12 samples per BatchNorm channel): under such noise the worst gradient error of the HIP path, of its exact-fp32 mode AND of
    the fp32 CPU oracle against fp64 all jump between the fp32 level (5e-5) and 1e-2 .. 1e-1 from copy to copy - about one
    copy in three carries a flip in any fp32 implementation.  So: the gradients meet the statistical bars on the recipe input,
    or on at least two of up to seven copies of it with relative noise 1e-6 (same weights), while every evaluated copy keeps
    its head clean (a flip sits inside the trunk) and stays an order of magnitude under what a defect produces."""
    from oracle import recipes
    cfg, omodel, x, joints = recipes.build(name, seed=1234)
    tgt, wt = recipes.make_targets(cfg, joints, 77)
    m = product_model(cfg, omodel, dev).train()
    recipes.set_dropout(m, 0.0)

    def bars(mh, mc, wh, wc):
        return mh <= max(3 * mc, 2e-3) and wh <= max(3 * wc, 2e-2)

    met, seen = 0, []
    for k in range(8):
        xk = x
        if k:
            xk = x * (1 + 1e-6 * torch.randn(x.shape, generator=torch.Generator().manual_seed(9100 + k)))
        mh, mc, wh, wc, wk = _grad_errors(m, omodel, xk, tgt, wt, dev)
        on = _grad_errors.onset
        ok = bars(mh, mc, wh, wc)
        seen.append((k, ok, mh, wh, wk))
        print(f"{name} @ seed 1234, {'recipe input' if not k else 'noise copy %d' % k}: median hip {mh:.2e} / cpu32 {mc:.2e}, worst hip "
              f"{wh:.2e} at {wk} / cpu32 {wc:.2e}" + ("" if on is None else f"; first tensor in backward order beyond the bar: "
              f"{on[1]} ({on[2]:.2e}) after {on[0]} clean ones"))
        if not ok:
            # a flip sits INSIDE the trunk: the tensors between it and the loss - at least the head - are clean, and its effect
            # is bounded (one flipped unit moves most gradients of the 3x2-pixel branch by 5-10 %: oracle/recipes.py)
            assert on is None or on[0] > 0, f"{name} @ seed 1234: the gradient of the last layer ({on[3]}) is already off"
            assert mh <= 1e-1 and wh <= 5e-1, f"{name} @ seed 1234: median {mh:.2e} / worst {wh:.2e} at {wk}"
        met += ok
        if (k == 0 and ok) or met >= 2:
            return
    raise AssertionError(f"{name} @ seed 1234: the bars are met on {met} of {len(seen)} noise copies {seen}: not a flipped ReLU")


@pytest.mark.parametrize("name", FULL)
def test_full_size_baseline_configs_forward(dev, name):
    """Every BASELINE.json config at full size, engine default math mode: eval forward of the HIP path against the
    output the REFERENCE produced for the same seeded network and input (tests/golden/model_<name>.npz)."""
    from oracle import recipes
    from buctd_amd import ops
    assert ops.get_conv_math() == "bf16x6", "the engine's default math mode is the fp32-class bf16x6"
    gold = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    ref = gold["out"]
    cfg, omodel, x, _ = recipes.build(name)
    m = product_model(cfg, omodel, dev).eval()
    with torch.no_grad():
        y = m(x.to(dev)).cpu().numpy()
    top = float(np.abs(ref).max())
    err = np.abs(y - ref).max()
    print(f"{name} full size: |hip - reference| = {err:.3e}, max|y| {top:.3f}")
    assert 0.4 <= top <= 1.0
    assert err <= BAR, f"{name}: full-size forward differs from the reference by {err:.3e}"
    assert np.array_equal(y.reshape(y.shape[0], y.shape[1], -1).argmax(2), gold["argmax"])


# C3 is C4's trunk without the attention block: its full-size train step (52 s of CPU oracle) runs with --runslow; its
# full-size forward golden stays in the default suite
@pytest.mark.parametrize("name,tag", [("coam_w48_384x288", "C4"),
                                      pytest.param("prenet_w48_384x288", "C3", marks=pytest.mark.slow),
                                      ("prenet_w32_256x192", "C2")])
def test_full_size_train_step_vs_oracle(dev, name, tag):
    """BASELINE configs C4 (CoAM-W48 384x288, the bench workload), C3 (preNet W48 384x288) and C2 (preNet W32 256x192) at
    full size in TRAIN mode: the kernels only these sizes reach - 512-position conv tiles, the full-resolution preNet 7x7
    convolutions, fc_o on the bf16x6 GEMM and the position attention at T = 6912, the 32 / 64 / 128 / 256-channel tile plans of
    W32 - inside one forward + loss + backward, against the fp32 and fp64 CPU oracle evaluated here.  Batch 2 (the recipe's
    image and a scaled mirror image) so that the batch statistics are not degenerate."""
    from oracle import recipes
    from buctd_amd.core.loss import JointsMSELoss
    cfg, omodel, x, joints = recipes.build(name)
    x = torch.cat([x, x.flip(3) * 0.9], 0)
    joints = torch.cat([joints, joints], 0)
    tgt, wt = recipes.make_targets(cfg, joints, 77)
    m = product_model(cfg, omodel, dev).train()
    recipes.set_dropout(m, 0.0)
    y = m(x.to(dev))
    loss = JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev))
    loss.backward()
    import copy
    from oracle import core as ocore
    o32 = copy.deepcopy(omodel).train()
    recipes.set_dropout(o32, 0.0)
    y32 = o32(x)
    l32 = ocore.JointsMSELoss(True)(y32, tgt, wt)
    l32.backward()
    g64 = _oracle_grads(omodel, x, tgt, wt, torch.float64)
    g32 = {k: p.grad.detach() for k, p in o32.named_parameters() if p.grad is not None}
    top = float(y32.detach().abs().max())
    err = float((y.detach().cpu() - y32.detach()).abs().max())
    print(f"{tag} full size train: max|y| {top:.3f}, |hip - oracle| {err:.3e}, loss hip {loss.item():.6f} oracle {l32.item():.6f}")
    assert err <= BAR * max(1.0, top)
    assert rel(loss.item(), l32.item()) <= 1e-4
    params = dict(m.named_parameters())
    gmax = max(v.norm().item() for v in g64.values())
    e_hip, e_cpu, keys = [], [], []
    for k, g in g64.items():
        den = g.norm().item()
        if den <= 1e-6 * gmax:
            continue
        e_hip.append((params[k].grad.detach().cpu().double() - g).norm().item() / den)
        e_cpu.append((g32[k].double() - g).norm().item() / den)
        keys.append(k)
    med_h, med_c, worst, worst_ref = float(np.median(e_hip)), float(np.median(e_cpu)), max(e_hip), max(e_cpu)
    order = np.argsort(e_hip)[::-1][:4]
    print(f"{tag} full size: grad rel err vs fp64 - median hip {med_h:.2e} / cpu32 {med_c:.2e}; max hip {worst:.2e} / cpu32 "
          f"{worst_ref:.2e}; worst tensors: " + ", ".join(f"{keys[i]} {e_hip[i]:.1e} (cpu32 {e_cpu[i]:.1e})" for i in order))
    # the HIP path must be as close to the fp64 truth as the fp32 CPU path is (3x), with floors for a flipped ReLU
    # (DESIGN.md 4): 2e-3 on the median, 2e-2 on the worst tensor - a wiring or scaling error of 2 % in one tensor shows
    assert med_h <= max(3 * med_c, 2e-3) and worst <= max(3 * worst_ref, 2e-2)


@pytest.mark.slow
def test_c4_train_step_at_bench_batch_through_the_engine(dev):
    """The assembled BASELINE step (config C4 at the scripts' batch 32) through the SHIPPED engine path - engine.DataParallel +
    flat gradient arena + FusedAdam, group launches and streams on, exactly what bench.py times - against the CPU oracle
    evaluated here (reference lib/core/function.py:102-175).  Batch 32 selects kernels batch 2 never reaches in one piece: the
    448-position tiles, the two-family group plans, the 1024 / n weight-gradient splits, the row-streaming 1x1 kernel.
    Bars: loss 1e-4 relative, output 1e-3 absolute (unit-scale heat maps), the gradients of 24 sampled parameter tensors as
    close to an fp64 evaluation as the fp32 CPU path is (3x, floors 2e-3 median / 2e-2 worst: a flipped ReLU), and the
    parameters after the optimizer step against torch.optim.Adam on the oracle's gradients.   ~2 min of CPU: run with --runslow."""
    import copy
    from oracle import recipes, core as ocore
    from buctd_amd import engine
    from buctd_amd.core.loss import JointsMSELoss
    B = 32
    cfg, omodel, x1, joints1 = recipes.build("coam_w48_384x288")
    g = torch.Generator().manual_seed(321)
    # 32 different crops from the recipe's image: shifted, scaled copies + noise on the RGB channels
    xs = []
    for i in range(B):
        xi = torch.roll(x1, shifts=(5 * i % 41, 3 * i % 29), dims=(2, 3)) * (1.0 - 0.01 * i)
        xi[:, :3] += 0.1 * torch.randn(xi[:, :3].shape, generator=g)
        xs.append(xi)
    x = torch.cat(xs, 0)
    joints = joints1.repeat(B, 1, 1).clone()
    joints[..., :2] += torch.randn(joints[..., :2].shape, generator=g) * 6.0
    tgt, wt = recipes.make_targets(cfg, joints, 77)
    m = product_model(cfg, omodel, dev).train()
    recipes.set_dropout(m, 0.0)
    model = engine.DataParallel(m)
    optimizer = engine.get_optimizer(cfg, model)
    model.flatten()
    names = [k for k, _ in m.named_parameters()]
    p0 = {k: p.detach().clone() for k, p in m.named_parameters()}
    y = model(x.to(dev))
    loss = JointsMSELoss(True)(y, tgt.to(dev), wt.to(dev))
    optimizer.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    optimizer.step()
    torch.cuda.synchronize()
    # ---- oracle, fp32 (the reference path) and fp64 (the truth the gradient bar is measured against)
    o32 = copy.deepcopy(omodel).train()
    recipes.set_dropout(o32, 0.0)
    y32 = o32(x)
    l32 = ocore.JointsMSELoss(True)(y32, tgt, wt)
    l32.backward()
    g32 = {k: p.grad.detach() for k, p in o32.named_parameters() if p.grad is not None}
    g64 = _oracle_grads(omodel, x, tgt, wt, torch.float64)
    top = float(y32.detach().abs().max())
    err = float((y.detach().cpu() - y32.detach()).abs().max())
    print(f"C4 batch {B} through the engine: max|y| {top:.3f}, |hip - oracle| {err:.3e}, loss hip {loss.item():.6f} oracle {l32.item():.6f}")
    assert err <= BAR * max(1.0, top)
    assert rel(loss.item(), l32.item()) <= 1e-4
    # ---- 24 sampled parameter tensors, spread over the network (every 1/24 of the parameter list with a gradient)
    gmax = max(v.norm().item() for v in g64.values())
    keys = [k for k in names if k in g64 and g64[k].norm().item() > 1e-6 * gmax]
    sample = [keys[(i * len(keys)) // 24] for i in range(24)]
    e_hip = [(grads[k].cpu().double() - g64[k]).norm().item() / g64[k].norm().item() for k in sample]
    e_cpu = [(g32[k].double() - g64[k]).norm().item() / g64[k].norm().item() for k in sample]
    med_h, med_c, worst, worst_ref = float(np.median(e_hip)), float(np.median(e_cpu)), max(e_hip), max(e_cpu)
    print(f"C4 batch {B}: grad rel err vs fp64 over {len(sample)} sampled tensors - median hip {med_h:.2e} / cpu32 {med_c:.2e}; "
          f"max hip {worst:.2e} ({sample[int(np.argmax(e_hip))]}) / cpu32 {worst_ref:.2e}")
    assert med_h <= max(3 * med_c, 2e-3) and worst <= max(3 * worst_ref, 2e-2)
    # ---- the optimizer step: the first Adam step moves an element by -lr * g / (|g| + 1e-8) (bias-corrected moments of one
    # gradient).  Held against the engine's OWN gradients (captured before step()), element by element: this pins the flat
    # arena, the bucket cut and the fused kernel exactly, independent of the gradient noise a flipped ReLU leaves in a tensor
    # (where |g| is of the size of eps the move amplifies that noise: 18 % of the move for one BatchNorm bias whose gradient is
    # 1.4e-2 from fp64 - within the gradient bar above, meaningless as a bar on the optimizer)
    pn = dict(m.named_parameters())
    lr = float(cfg.TRAIN.LR)
    worst_move = 0.0
    for k in names:
        if k not in grads:
            continue
        gh = grads[k].double().cpu()
        want = -lr * gh / (gh.abs() + 1e-8)
        dh = (pn[k].detach().cpu() - p0[k].cpu()).double()
        r = float((dh - want).abs().max()) / lr
        worst_move = max(worst_move, r)
        assert r <= 2e-3, f"{k}: the optimizer step differs from Adam on the engine's own gradient by {r:.2e} lr"
    print(f"C4 batch {B}: FusedAdam step vs Adam's formula on the engine's gradients, all {len(grads)} tensors: worst element {worst_move:.2e} lr")
