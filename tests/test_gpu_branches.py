"""Cross-branch group launches of a HighResolutionModule (reference lib/models/pose_hrnet.py:177-185, 247-249): the k-th
convolutions / BatchNorm kernels / weight gradients of all branches in one launch each (csrc/block.hip
buctd_basic_branches_*, ops.BasicBranchesFn) against one BasicChainFn per branch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_chains(dev, shapes, n, seed):
    import torch.nn as tnn
    g = torch.Generator().manual_seed(seed)
    xs, dys, chains = [], [], []
    for (N, H, W, Cn) in shapes:
        xs.append(torch.randn(N, H, W, Cn, generator=g).to(dev))
        dys.append(torch.randn(N, H, W, Cn, generator=g).to(dev))
        blocks = []
        for _ in range(n):
            ws = [tnn.Parameter((torch.randn(Cn, Cn, 3, 3, generator=g) * 0.08).contiguous(memory_format=torch.channels_last).to(dev))
                  for _ in range(2)]
            bns = []
            for _s in (1, 2):
                bn = tnn.BatchNorm2d(Cn).to(dev).train()
                with torch.no_grad():
                    bn.weight.copy_(torch.rand(Cn, generator=g) + 0.5)
                    bn.bias.copy_(torch.randn(Cn, generator=g) * 0.2)
                bns.append(bn)
            blocks.append((ws[0], bns[0], ws[1], bns[1]))
        chains.append(blocks)
    return xs, dys, chains


@pytest.mark.parametrize("shapes", [
    [(3, 24, 18, 48), (3, 12, 9, 96)],
    [(2, 24, 20, 48), (2, 12, 10, 96), (2, 6, 5, 192), (2, 3, 3, 384)],
    [(2, 16, 12, 32), (2, 8, 6, 64), (2, 4, 3, 128)],
    [(32, 96, 72, 48), (32, 48, 36, 96), (32, 24, 18, 192), (32, 12, 9, 384)],
], ids=["w48x2", "w48x4", "w32x3", "w48_full"])
def test_branches_node_equals_chains(dev, shapes):
    """Forward outputs, running statistics, input gradients and BatchNorm gradients: the same element arithmetic in the same
    order per tile -> bit-identical.  Weight gradients: the group launch cuts every convolution into fewer position splits
    (another fixed summation order) -> equal to fp32 round-off of the sums, and run-to-run bit-reproducible."""
    from buctd_amd import ops
    full = shapes[0][0] == 32
    n = 2 if full else 3
    xs, dys, chains = _make_chains(dev, shapes, n, 5)
    assert ops.group_branches_ok(xs, chains)
    params = [q for ch in chains for b in ch for q in (b[0], b[2], b[1].weight, b[1].bias, b[3].weight, b[3].bias)]
    is_w = [q.dim() == 4 for q in params]
    res = {}
    for mode in ("group", "chains", "group2"):
        for q in params:
            q.grad = None
        for ch in chains:
            for b in ch:
                for bn in (b[1], b[3]):
                    bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        xi = [x.clone().requires_grad_(True) for x in xs]
        if mode == "chains":
            ys = [ops.BasicChainFn.apply(xi[b], chains[b][0][0], chains[b]) for b in range(len(xs))]
        else:
            ys = ops.BasicBranchesFn.apply(chains[0][0][0], chains, *xi)
        torch.autograd.backward(list(ys), dys)
        torch.cuda.synchronize()
        res[mode] = ([y.detach().clone() for y in ys] + [x.grad.clone() for x in xi] +
                     [ch[-1][3].running_var.clone() for ch in chains] + [ch[0][1].running_mean.clone() for ch in chains],
                     [q.grad.clone() for q in params])
        del ys, xi
    for a, b, c in zip(res["group"][0], res["chains"][0], res["group2"][0]):
        assert torch.equal(a, c), "the group path is not run-to-run reproducible"
        assert torch.equal(a, b), "activations / input gradients / running statistics must be bit-identical to the chains"
    for a, b, c, w in zip(res["group"][1], res["chains"][1], res["group2"][1], is_w):
        assert torch.equal(a, c), "the group path is not run-to-run reproducible"
        if w:
            sc = b.abs().max().item()
            assert (a - b).abs().max().item() <= 5e-6 * sc, f"weight gradient: group vs chains {(a - b).abs().max().item():.3e} (scale {sc:.3e})"
        else:
            assert torch.equal(a, b), "BatchNorm parameter gradients must be bit-identical to the chains"


def test_group_weight_gradient_matches_fp64(dev):
    """buctd_conv3x3_wgrad_bf16x6_group against an fp64 convolution weight gradient (<= 2e-6 of the largest element)."""
    import ctypes as C
    from buctd_amd import _C
    lib = _C.lib()
    shapes = [(4, 24, 18, 48), (4, 12, 9, 96), (4, 6, 5, 192)]
    g = torch.Generator().manual_seed(11)
    items = (_C.Wg3Conv * len(shapes))()
    keep = []
    for k, (N, H, W, Cn) in enumerate(shapes):
        x = torch.randn(N, H, W, Cn, generator=g).to(dev)
        dy = torch.randn(N, H, W, Cn, generator=g).to(dev)
        dw = torch.empty(Cn, 3, 3, Cn, device=dev)
        need = lib.buctd_conv3x3_wgrad_bf16x6_group_workspace(len(shapes), N, H, W, Cn, Cn)
        assert need > 0
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        it = items[k]
        it.N, it.H, it.W, it.Ci, it.Co = N, H, W, Cn, Cn
        it.x, it.dy, it.dw, it.accumulate = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0
        it.workspace, it.workspace_bytes = ws.data_ptr(), need
        keep.append((x, dy, dw, ws))
    _C.check(lib.buctd_conv3x3_wgrad_bf16x6_group(len(shapes), items, _C.stream_ptr()), "wgrad group")
    torch.cuda.synchronize()
    for (x, dy, dw, _) in keep:
        xd = x.double().permute(0, 3, 1, 2).cpu()
        dyd = dy.double().permute(0, 3, 1, 2).cpu()
        ref = torch.nn.grad.conv2d_weight(xd, (dyd.shape[1], xd.shape[1], 3, 3), dyd, stride=1, padding=1)   # [Co][Ci][3][3]
        got = dw.cpu().double().permute(0, 3, 1, 2)
        sc = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-6 * sc


def test_module_forward_uses_group_path_and_matches_per_branch(dev):
    """HighResolutionModule.forward: the grouped branches against the per-branch chains on the module level (outputs of the
    fuse rows bit-identical)."""
    from buctd_amd import ops
    from buctd_amd import nn as bnn
    from buctd_amd.models.hrnet_common import HighResolutionModule, BasicBlock
    torch.manual_seed(9)
    m = HighResolutionModule(3, BasicBlock, [2, 2, 2], [48, 96, 192], [48, 96, 192], "SUM").to(dev).train()
    bnn.prepare_module(m)
    xs = [torch.randn(2, 16, 12, 48, device=dev), torch.randn(2, 8, 6, 96, device=dev), torch.randn(2, 4, 3, 192, device=dev)]
    outs = {}
    for on in (True, False):
        old = ops.set_group_branches(on)
        try:
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.zero_(); mod.running_var.fill_(1.0)
            xi = [x.clone().requires_grad_(True) for x in xs]
            ys = m(xi)
            torch.autograd.backward(list(ys), [torch.ones_like(y) for y in ys])
            torch.cuda.synchronize()
            outs[on] = [y.detach().clone() for y in ys] + [x.grad.clone() for x in xi]
            for p in m.parameters():
                p.grad = None
        finally:
            ops.set_group_branches(old)
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("width,batch", [(32, 8), (48, 4)])
def test_eval_branches_share_launches_and_match_the_per_branch_path(dev, width, batch):
    """Eval mode (validate(), reference lib/core/function.py:178-336): the k-th convolutions of the branches of a
    HighResolutionModule (pose_hrnet.py:177-185) go out as one launch (buctd_conv3x3_bf16x6_group_eval) - the same kernels
    on the same tiles as one launch per branch and convolution, so the heat maps are bit-identical."""
    import torch
    from buctd_amd import models, ops
    from buctd_amd.config import cfg as base, hrnet_extra
    c = base.clone()
    c.defrost()
    c.MODEL.NAME = "pose_hrnet"
    c.MODEL.NUM_JOINTS = 17
    c.MODEL.IMAGE_SIZE = [192, 256]
    c.MODEL.HEATMAP_SIZE = [48, 64]
    c.MODEL.CONDITIONAL_TOPDOWN = True
    c.MODEL.EXTRA = hrnet_extra(width, use_pre_net=True, modules=(1, 1, 1))
    c.freeze()
    torch.manual_seed(21)
    net = models.pose_hrnet.get_pose_net(c, is_train=False).to(dev).eval()
    # running statistics away from their initial values, as after training
    g = torch.Generator().manual_seed(3)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g).to(dev) * 0.1)
            m.running_var.copy_((torch.rand(m.running_var.shape, generator=g) + 0.5).to(dev))
    x = torch.randn(batch, 6, 256, 192, generator=g).to(dev)
    calls = {"n": 0}
    raw = ops.basic_branches_eval

    def counted(xs, chains):
        calls["n"] += 1
        return raw(xs, chains)

    ops.basic_branches_eval = counted
    try:
        with torch.no_grad():
            grouped = net(x)
            ops._GROUP_BRANCHES["on"] = False
            plain = net(x)
    finally:
        ops._GROUP_BRANCHES["on"] = True
        ops.basic_branches_eval = raw
    assert calls["n"] == 3, calls            # one module each in stages 2, 3 and 4
    assert torch.equal(grouped, plain)
