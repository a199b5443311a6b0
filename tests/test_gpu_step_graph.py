"""engine.StepGraph: forward + loss + backward of a training iteration captured once as a hipGraph and replayed
(reference loop: lib/core/function.py:102-175 - forward / zero_grad / backward / optimizer.step per batch).  A replay
launches the kernels of the eager step, on the same streams' order, so everything it produces - loss, output, parameters,
BatchNorm running statistics, optimizer state - must equal the eager engine's bit for bit, step after step, on batches
that differ from step to step."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _prenet_cfg(width=16, modules=(1, 2, 2)):
    from buctd_amd.config import cfg as base, hrnet_extra
    c = base.clone()
    c.defrost()
    c.MODEL.NAME = "pose_hrnet"
    c.MODEL.NUM_JOINTS = 17
    c.MODEL.IMAGE_SIZE = [64, 96]
    c.MODEL.HEATMAP_SIZE = [16, 24]
    c.MODEL.SIGMA = 2
    c.MODEL.PRETRAINED = ""
    c.MODEL.CONDITIONAL_TOPDOWN = True
    c.MODEL.EXTRA = hrnet_extra(width, use_pre_net=True, modules=modules)
    c.DATASET.DATASET = "coco"
    c.DATASET.COLORED = True
    c.TRAIN.LR = 1e-3
    c.freeze()
    return c


def _batch(cfg, n, seed, device):
    g = torch.Generator().manual_seed(seed)
    w, h = cfg.MODEL.IMAGE_SIZE
    hw, hh = cfg.MODEL.HEATMAP_SIZE
    k = cfg.MODEL.NUM_JOINTS
    x = torch.randn(n, 6, h, w, generator=g).to(device)
    t = torch.rand(n, k, hh, hw, generator=g).to(device)
    wt = (torch.rand(n, k, 1, generator=g) < 0.8).float().to(device)
    return x, t, wt


def _pair(cfg, device):
    from buctd_amd import engine, models
    from buctd_amd.core.loss import JointsMSELoss
    torch.manual_seed(7)
    net_a = models.pose_hrnet.get_pose_net(cfg, is_train=True).to(device)
    net_b = copy.deepcopy(net_a)
    out = []
    for net in (net_a, net_b):
        model = engine.DataParallel(net)
        opt = engine.get_optimizer(cfg, model)
        model.train()
        out.append((model, opt))
    return out, JointsMSELoss(True)


@pytest.mark.parametrize("streams", ["single", "engine"])
def test_replayed_steps_equal_eager_steps_bit_for_bit(dev, streams):
    from buctd_amd import engine
    cfg = _prenet_cfg()
    ((eager, eopt), (graphed, gopt)), crit = _pair(cfg, dev)
    step = engine.StepGraph(graphed, crit, gopt, warmup=2, streams=streams)
    for i in range(7):
        x, t, w = _batch(cfg, 4, 100 + i, dev)
        # the two engines own different arenas: set each one current before its step (as two processes would have it)
        engine.ops.set_grad_arena(eopt.flat)
        out_e = eager(x)
        loss_e = crit(out_e, t, w)
        eopt.zero_grad()
        loss_e.backward()
        eopt.step()
        engine.ops.set_grad_arena(gopt.flat)
        out_g, loss_g = step(x, t, w)
        assert torch.equal(loss_e.detach(), loss_g.detach()), (i, float(loss_e), float(loss_g))
        assert torch.equal(out_e.detach(), out_g.detach()), i
    assert step.replays == 5
    assert torch.equal(eopt.flat.flat, gopt.flat.flat)
    assert torch.equal(eopt.exp_avg, gopt.exp_avg) and torch.equal(eopt.exp_avg_sq, gopt.exp_avg_sq)
    sd_e, sd_g = eager.module.state_dict(), graphed.module.state_dict()
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k
    assert int(sd_g["bn1.num_batches_tracked"]) == 7


def test_a_ragged_batch_runs_eager_and_the_graph_survives_it(dev):
    from buctd_amd import engine
    from buctd_amd.core.loss import JointsMSELoss
    cfg = _prenet_cfg()
    ((eager, eopt), (graphed, gopt)), crit = _pair(cfg, dev)
    step = engine.StepGraph(graphed, crit, gopt, warmup=1)
    sizes = [4, 4, 4, 3, 4, 4]          # the fourth batch is the ragged tail of an epoch
    for i, n in enumerate(sizes):
        x, t, w = _batch(cfg, n, 300 + i, dev)
        engine.ops.set_grad_arena(eopt.flat)
        loss_e = crit(eager(x), t, w)
        eopt.zero_grad()
        loss_e.backward()
        eopt.step()
        engine.ops.set_grad_arena(gopt.flat)
        _, loss_g = step(x, t, w)
        assert torch.equal(loss_e.detach(), loss_g.detach()), (i, float(loss_e), float(loss_g))
    assert step.replays == 4
    assert torch.equal(eopt.flat.flat, gopt.flat.flat)
    # a caller that clears the gradients itself in front of a replayed step (the reference loop's habit) must not turn the
    # replayed gradients into "no gradient"
    x, t, w = _batch(cfg, 4, 399, dev)
    engine.ops.set_grad_arena(eopt.flat)
    loss_e = crit(eager(x), t, w)
    eopt.zero_grad()
    loss_e.backward()
    eopt.step()
    engine.ops.set_grad_arena(gopt.flat)
    gopt.zero_grad()
    _, loss_g = step(x, t, w)
    assert torch.equal(loss_e.detach(), loss_g.detach())
    assert torch.equal(eopt.flat.flat, gopt.flat.flat)


def test_train_entry_point_takes_a_step_graph(dev):
    """core.function.train(..., step_graph=...) runs the reference loop with the captured step: same losses in the log
    as the eager loop."""
    from buctd_amd import engine
    from buctd_amd.core.function import train
    from buctd_amd.core.loss import JointsMSELoss
    cfg = _prenet_cfg()
    ((eager, eopt), (graphed, gopt)), crit = _pair(cfg, dev)
    loader = []
    for i in range(5):
        x, t, w = _batch(cfg, 2, 500 + i, torch.device("cpu"))
        loader.append((x, t, w, {}))

    class Writer:
        def __init__(self):
            self.losses = []

        def add_scalar(self, k, v, s):
            if k == "train_loss":
                self.losses.append(float(v))

    c = cfg.clone()
    c.defrost()
    c.PRINT_FREQ = 1
    c.freeze()
    we, wg = {"writer": Writer(), "train_global_steps": 0}, {"writer": Writer(), "train_global_steps": 0}
    engine.ops.set_grad_arena(eopt.flat)
    train(c, loader, eager, crit, eopt, 0, "/tmp", "/tmp", we)
    engine.ops.set_grad_arena(gopt.flat)
    step = engine.StepGraph(graphed, crit, gopt, warmup=1)
    train(c, loader, graphed, crit, gopt, 0, "/tmp", "/tmp", wg, step_graph=step)
    assert step.replays == 4
    assert we["writer"].losses == wg["writer"].losses
    assert torch.equal(eopt.flat.flat, gopt.flat.flat)


def test_dropout_models_are_refused(dev):
    """The mask seed of train-mode dropout is a launch argument: a replay would repeat one mask, so the capture is refused."""
    from buctd_amd import engine, models
    from buctd_amd._C import BuctdHipError
    from buctd_amd.config import cfg as base, hrnet_extra
    from buctd_amd.core.loss import JointsMSELoss
    c = base.clone()
    c.defrost()
    c.MODEL.NAME = "pose_hrnet_coam"
    c.MODEL.NUM_JOINTS = 14
    c.MODEL.IMAGE_SIZE = [64, 96]
    c.MODEL.HEATMAP_SIZE = [16, 24]
    c.MODEL.ATT_MODULES = [False, True, False, False]
    c.MODEL.CONDITIONAL_TOPDOWN = True
    c.MODEL.EXTRA = hrnet_extra(16, use_attention=True, modules=(1, 2, 2))
    c.DATASET.COLORED = True
    c.freeze()
    torch.manual_seed(3)
    net = models.pose_hrnet_coam.get_pose_net(c, is_train=True).to(dev)
    model = engine.DataParallel(net)
    opt = engine.get_optimizer(c, model)
    model.train()
    step = engine.StepGraph(model, JointsMSELoss(True), opt, warmup=1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 6, 96, 64, generator=g).to(dev)
    t = torch.rand(2, 14, 24, 16, generator=g).to(dev)
    w = torch.ones(2, 14, 1, device=dev)
    step(x, t, w)
    with pytest.raises((BuctdHipError, NotImplementedError)):
        step(x, t, w)
    # the refusal leaves the engine usable: the eager step still runs
    loss = JointsMSELoss(True)(model(x), t, w)
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert torch.isfinite(loss)


def test_forward_graph_equals_the_eager_eval_forward(dev):
    """engine.ForwardGraph: the eval-mode forward replayed from a hipGraph (the small-batch serving path) returns the eager
    engine's heat maps bit for bit, per input signature, on changing inputs; train mode and grad mode go to the module."""
    from buctd_amd import engine, models
    cfg = _prenet_cfg()
    torch.manual_seed(9)
    net = models.pose_hrnet.get_pose_net(cfg, is_train=False).to(dev).eval()
    fg = engine.ForwardGraph(net, warmup=1, autoselect=False)
    with torch.no_grad():
        for i in range(6):
            n = 3 if i % 2 == 0 else 1                 # two signatures, interleaved
            x, _, _ = _batch(cfg, n, 700 + i, dev)
            ref = net(x)
            got = fg(x)
            assert torch.equal(ref, got), i
    assert fg.replays == 4                             # call 0 of each signature runs eager, call 1 captures and replays
    assert len(fg._graphs) == 2
    # outputs are copies: an earlier result survives the next replay of its signature
    with torch.no_grad():
        xa, _, _ = _batch(cfg, 3, 800, dev)
        xb, _, _ = _batch(cfg, 3, 801, dev)
        ya = fg(xa)
        keep = ya.clone()
        fg(xb)
        assert torch.equal(ya, keep)
    # with gradients enabled the module itself runs
    x, _, _ = _batch(cfg, 3, 802, dev)
    before = fg.replays
    y = fg(x)
    assert fg.replays == before and y.requires_grad


def test_validate_runs_on_a_forward_graph(dev):
    """validate() (reference lib/core/function.py:178-336) with the network wrapped in engine.ForwardGraph: same predictions
    table as with the plain module, flip test on."""
    import numpy as np
    from buctd_amd import engine, models
    from buctd_amd.core.function import validate
    from buctd_amd.core.loss import JointsMSELoss
    cfg = _prenet_cfg()
    c = cfg.clone()
    c.defrost()
    c.TEST.FLIP_TEST = True
    c.TEST.POST_PROCESS = True
    c.TEST.SHIFT_HEATMAP = True
    c.PRINT_FREQ = 100
    c.freeze()
    torch.manual_seed(13)
    net = models.pose_hrnet.get_pose_net(c, is_train=False).to(dev).eval()

    class Dataset:
        flip_pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
        image_size = c.MODEL.IMAGE_SIZE
        kpt_colors = [[(37 * k) % 256, (91 * k) % 256, (53 * k) % 256] for k in range(17)]

        def __init__(self, n):
            self.n, self.captured = n, None

        def __len__(self):
            return self.n

        def evaluate(self, cfg, preds, output_dir, all_boxes, img_path, *a, **k):
            self.captured = preds.copy()
            return {"AP": 0.0}, 0.0

    batches = []
    for i in range(4):
        x, t, w = _batch(c, 2, 900 + i, torch.device("cpu"))
        g = torch.Generator().manual_seed(950 + i)
        meta = {"center": torch.rand(2, 2, generator=g) * 100 + 50, "scale": torch.rand(2, 2, generator=g) + 0.5,
                "score": torch.rand(2, generator=g), "annotation_id": torch.arange(2) + 2 * i,
                "image": [f"im_{i}_{j}.jpg" for j in range(2)],
                "cond_joints": torch.cat([torch.rand(2, 17, 2, generator=g) * 60, torch.zeros(2, 17, 1)], 2),
                "cond_joints_vis": torch.ones(2, 17, 3)}
        batches.append((x, t, w, meta))
    tables = []
    for model in (net, engine.ForwardGraph(net, warmup=1)):
        ds = Dataset(8)
        validate(c, batches, ds, model, JointsMSELoss(True), "/tmp", "/tmp", None)
        tables.append(ds.captured)
    assert np.array_equal(tables[0], tables[1])


def test_autoselect_keeps_results_whichever_path_wins(dev):
    """StepGraph(autoselect=True) times its last settling step and its first two replays and keeps the faster path per
    signature; ForwardGraph does the same right after its capture.  Whichever wins, the results are the eager engine's."""
    from buctd_amd import engine
    cfg = _prenet_cfg()
    ((eager, eopt), (graphed, gopt)), crit = _pair(cfg, dev)
    step = engine.StepGraph(graphed, crit, gopt, warmup=2, autoselect=True)
    for i in range(8):
        x, t, w = _batch(cfg, 2, 1100 + i, dev)
        engine.ops.set_grad_arena(eopt.flat)
        loss_e = crit(eager(x), t, w)
        eopt.zero_grad()
        loss_e.backward()
        eopt.step()
        engine.ops.set_grad_arena(gopt.flat)
        _, loss_g = step(x, t, w)
        assert torch.equal(loss_e.detach(), loss_g.detach()), (i, float(loss_e), float(loss_g))
    assert step.replays >= 2
    assert torch.equal(eopt.flat.flat, gopt.flat.flat)
    key = next(iter(step._seen))
    assert (key in step._graphs) != (key in step._eager_only)
    net = graphed.module.eval()
    fg = engine.ForwardGraph(net, warmup=1, autoselect=True)
    with torch.no_grad():
        for i in range(4):
            x, _, _ = _batch(cfg, 2, 1200 + i, dev)
            assert torch.equal(net(x), fg(x)), i
    assert len(fg._graphs) == 1


def test_forward_graph_notices_a_checkpoint_loaded_in_place(dev):
    """Prepared filter images and folded BatchNorms are rebuilt by the eager engine when the weights change; a replay would
    keep the old ones - ForwardGraph watches the optimizer epoch and a sample of tensor versions and re-captures."""
    from buctd_amd import engine, models
    cfg = _prenet_cfg()
    torch.manual_seed(31)
    net = models.pose_hrnet.get_pose_net(cfg, is_train=False).to(dev).eval()
    torch.manual_seed(32)
    other = models.pose_hrnet.get_pose_net(cfg, is_train=False).to(dev).eval()
    fg = engine.ForwardGraph(net, warmup=1, autoselect=False)
    x, _, _ = _batch(cfg, 2, 1300, dev)
    with torch.no_grad():
        for _ in range(3):
            assert torch.equal(fg(x), net(x))
        assert fg.replays == 2
        net.load_state_dict(other.state_dict())
        want = other(x)
        for _ in range(3):
            assert torch.equal(fg(x), want)
    assert fg.replays == 4


@pytest.mark.parametrize("kind", ["prenet", "transpose"])
def test_two_lanes_return_what_the_eager_forward_returns(dev, kind):
    """ForwardGraph.submit: independent requests replay on two lanes (two graphs on the engine's two branch streams) beside
    each other; every request must get exactly the eager forward's output, whatever is in flight next to it."""
    from buctd_amd import engine, models
    if kind == "prenet":
        cfg = _prenet_cfg()
        net = models.pose_hrnet.get_pose_net(cfg, is_train=False)
    else:
        from oracle import recipes
        cfg, omodel, _, _ = recipes.build("transpose_w16_96x64")
        net = models.transpose_h.get_pose_net(cfg, is_train=False)
        net.load_state_dict(omodel.state_dict(), strict=True)
    torch.manual_seed(41)
    net = net.to(dev).eval()
    fg = engine.ForwardGraph(net, warmup=1, autoselect=False)
    w, h = cfg.MODEL.IMAGE_SIZE
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(2, 6, h, w, generator=g).to(dev) for _ in range(14)]
    with torch.no_grad():
        want = [net(x) for x in xs]
        want = [y[-1] if isinstance(y, list) else y for y in want]
        handles, got = [], []
        for i, x in enumerate(xs):
            handles.append(fg.submit(x))
            if len(handles) == 3:                       # up to three requests in flight over two lanes
                got.append(handles.pop(0).result())
        got += [hd.result() for hd in handles]
        torch.cuda.synchronize()
    got = [y[-1] if isinstance(y, list) else y for y in got]
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(want, got)):
        assert torch.equal(a, b), i
    assert fg.replays >= 11 and len(fg._graphs[next(iter(fg._graphs))]["lanes"]) == 2
