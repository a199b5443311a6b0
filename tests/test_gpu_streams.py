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
