"""Stream-level concurrency (ops.fork_join for HRNet branches / fuse rows, ops.conv_wgrad_async for weight gradients)
must not change a single bit: the same model, built fresh, gives identical outputs, losses and gradients with the
extra HIP streams on (from the very first call, when weights are still being re-laid-out) and off."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(dev, name, streams, train):
    from oracle import recipes
    from buctd_amd import models, ops
    from buctd_amd.core.loss import JointsMSELoss
    old = (ops._branch["on"], ops._side["on"])
    ops._branch["on"] = ops._side["on"] = streams
    try:
        cfg, omodel, x, joints = recipes.build(name)
        m = getattr(models, cfg.MODEL.NAME).get_pose_net(cfg, is_train=train)
        m.load_state_dict(omodel.state_dict(), strict=True)
        m = m.to(dev)
        if not train:
            with torch.no_grad():
                return m.eval()(x.to(dev)).cpu().numpy(), None, None
        m.train()
        recipes.set_dropout(m, 0.0)
        tgt, wt = recipes.make_targets(cfg, joints, 77)
        out = m(x.to(dev))
        loss = JointsMSELoss(True)(out, tgt.to(dev), wt.to(dev))
        loss.backward()
        grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}
        return out.detach().cpu().numpy(), loss.item(), grads
    finally:
        ops._branch["on"], ops._side["on"] = old


@pytest.mark.parametrize("name", ["prenet_w16_96x64", "coam_w16_96x64_colored"])
def test_streams_do_not_change_results(dev, name):
    y1, l1, g1 = _run(dev, name, True, True)
    y0, l0, g0 = _run(dev, name, False, True)
    assert np.array_equal(y1, y0), "forward differs with branch streams on"
    assert l1 == l0
    assert g1.keys() == g0.keys()
    bad = [(k, float(np.abs(g1[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-30))) for k in g0
           if not np.array_equal(g1[k], g0[k])]
    assert not bad, f"gradients differ with streams on ({len(bad)} of {len(g0)}): {sorted(bad, key=lambda kv: -kv[1])[:6]}"


def test_streams_first_call_eval_full_size(dev):
    """first-ever forward of a fresh full-size model on branch streams (lazy weight re-layout happens there)"""
    y1, _, _ = _run(dev, "coam_w48_384x288", True, False)
    y0, _, _ = _run(dev, "coam_w48_384x288", False, False)
    assert np.array_equal(y1, y0)


def test_fuse_row_backward_forms_the_batchnorm_sums_of_its_terms(dev):
    """FuseSum.backward (reference pose_hrnet.py:257-265) hands the conv -> BatchNorm terms of a fuse row their backward sums
    (buctd_fuse_sum_bwd_bnstat): the gradients must equal those of the path with a reduction launch per BatchNorm - to the
    rounding of a different fp32 partition of the same sums - and the reduction launches must really be gone."""
    import copy
    from buctd_amd import engine, models, ops
    from buctd_amd.config import cfg as base, hrnet_extra
    from buctd_amd.core.loss import JointsMSELoss
    c = base.clone()
    c.defrost()
    c.MODEL.NAME = "pose_hrnet"
    c.MODEL.NUM_JOINTS = 17
    c.MODEL.IMAGE_SIZE = [64, 96]
    c.MODEL.HEATMAP_SIZE = [16, 24]
    c.MODEL.CONDITIONAL_TOPDOWN = True
    c.MODEL.EXTRA = hrnet_extra(16, use_pre_net=True, modules=(1, 2, 2))
    c.freeze()
    torch.manual_seed(11)
    net = models.pose_hrnet.get_pose_net(c, is_train=True).to(dev).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 6, 96, 64, generator=g).to(dev)
    t = torch.rand(4, 17, 24, 16, generator=g).to(dev)
    w = torch.ones(4, 17, 1, device=dev)
    crit = JointsMSELoss(True)
    calls = {"n": 0}
    raw = ops.fuse_sum_bwd_bnstat

    def counted(*a, **k):
        calls["n"] += 1
        return raw(*a, **k)

    grads = {}
    old, old0 = ops._FUSE_BWD_BNSTAT, ops._FUSE_BWD_BNSTAT_S0
    ops._FUSE_BWD_BNSTAT_S0 = True         # the same-resolution terms too (the product fuses the up-sampled ones only)
    try:
        ops.fuse_sum_bwd_bnstat = counted
        for flag in (False, True):
            ops._FUSE_BWD_BNSTAT = flag      # read by ConvBnAct.forward (tags) and FuseSum (both directions) at call time
            m = copy.deepcopy(net)
            loss = crit(m(x), t, w)
            loss.backward()
            torch.cuda.synchronize()
            grads[flag] = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
            assert not ops._bwd_sums, "every accumulator a fuse row left must have been consumed"
    finally:
        ops._FUSE_BWD_BNSTAT, ops._FUSE_BWD_BNSTAT_S0 = old, old0
        ops.fuse_sum_bwd_bnstat = raw
    assert calls["n"] >= 5, calls            # fuse rows of the small net whose terms take the accumulator path
    assert grads[False].keys() == grads[True].keys()
    worst = 0.0
    for k, a in grads[False].items():
        b = grads[True][k]
        scale = float(a.abs().max()) + 1e-12
        worst = max(worst, float((a - b).abs().max()) / scale)
    assert worst <= 2e-5, worst
