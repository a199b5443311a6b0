"""Tight train-mode parity of the building blocks (conv+BN+ReLU fusions, BasicBlock / Bottleneck chains,
HighResolutionModule with its fuse rows): forward, input gradient and every parameter gradient of the HIP path
against an fp64 CPU evaluation of the oracle block, bar 5e-6 relative (fp32 round-off level) and never worse than
4x the fp32 CPU oracle's own distance to fp64."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _e(a, b):
    return ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()


def _check(name, e_hip, e_cpu):
    assert e_hip <= max(5e-6, 4 * e_cpu), f"{name}: rel err vs fp64 {e_hip:.2e} (fp32 CPU {e_cpu:.2e})"


def _run(dev, name, omod, pmod, xs):
    pmod.load_state_dict(omod.state_dict(), strict=True)
    pmod = pmod.to(dev).train()
    single = not isinstance(xs, list)
    xs = [xs] if single else xs
    ref = {}
    for dt in (torch.float64, torch.float32):
        m = copy.deepcopy(omod).to(dt).train()
        xi = [x.to(dt).clone().requires_grad_(True) for x in xs]
        ys = m(xi[0]) if single else m(list(xi))
        ys = [ys] if not isinstance(ys, (list, tuple)) else list(ys)
        ws = [torch.linspace(-1, 1, y.numel(), dtype=dt).view(y.shape) for y in ys]
        sum((y * w).sum() for y, w in zip(ys, ws)).backward()
        ref[dt] = ([y.detach() for y in ys], [x.grad for x in xi], {k: p.grad for k, p in m.named_parameters()})
    xd = [x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True) for x in xs]
    yd = pmod(xd[0]) if single else pmod(list(xd))
    yd = [yd] if not isinstance(yd, (list, tuple)) else list(yd)
    for y, y64 in zip(yd, ref[torch.float64][0]):
        w = torch.linspace(-1, 1, y64.numel()).view(y64.shape).permute(0, 2, 3, 1).contiguous().to(dev)
        y.backward(w, retain_graph=True)
    y64, gx64, gp64 = ref[torch.float64]
    y32, gx32, gp32 = ref[torch.float32]
    for i in range(len(yd)):
        _check(f"{name} out{i}", _e(yd[i].detach().permute(0, 3, 1, 2), y64[i]), _e(y32[i], y64[i]))
    for i in range(len(xd)):
        _check(f"{name} dx{i}", _e(xd[i].grad.permute(0, 3, 1, 2), gx64[i]), _e(gx32[i], gx64[i]))
    for k, p in pmod.named_parameters():
        assert p.grad is not None, f"{name}: {k} got no gradient"
        _check(f"{name} d{k}", _e(p.grad, gp64[k]), _e(gp32[k], gp64[k]))


@pytest.mark.parametrize("shape", [(3, 24, 16, 16, 16, 3, 1, True), (3, 3, 2, 128, 128, 3, 1, True),
                                   (3, 6, 4, 64, 128, 3, 2, False), (3, 12, 8, 64, 16, 1, 1, False),
                                   (32, 12, 9, 384, 384, 3, 1, True), (2, 16, 12, 3, 64, 3, 2, True)])
def test_conv_bn_act(dev, shape):
    from oracle import models as om
    from buctd_amd import nn as bnn
    N, H, W, ci, co, k, st, relu = shape
    torch.manual_seed(1)
    o = om.cbr(ci, co, k, st, relu)
    p = bnn.ConvBN(bnn.Conv2d(ci, co, k, st, (k - 1) // 2, bias=False), bnn.BatchNorm2d(co),
                   bnn.ReLU(True) if relu else None)
    _run(dev, f"convbn{shape}", o, p, torch.randn(N, ci, H, W))


def test_residual_chains(dev):
    from oracle import models as om
    from buctd_amd.models import hrnet_common as hc
    torch.manual_seed(2)
    _run(dev, "basicblock", om.BasicBlock(16, 16), hc.BasicBlock(16, 16), torch.randn(3, 16, 24, 16))
    o, _ = om.make_layer(om.BasicBlock, 32, 32, 4)
    p, _ = hc.make_residual_layer(hc.BasicBlock, 32, 32, 4)
    _run(dev, "4 basic blocks", o, p, torch.randn(3, 32, 12, 8))
    o, _ = om.make_layer(om.Bottleneck, 64, 64, 2)
    p, _ = hc.make_residual_layer(hc.Bottleneck, 64, 64, 2)
    _run(dev, "2 bottlenecks (layer1 head)", o, p, torch.randn(2, 64, 12, 8))


def test_fused_bottleneck_node_is_bit_identical(dev):
    """ops.BottleneckFn (one autograd node, the skip gradient added in a data-gradient epilogue) against the three / four
    ConvBnAct nodes it replaces: same kernels, same order - every output and gradient bit for bit."""
    from buctd_amd import ops
    from buctd_amd.models import hrnet_common as hc
    torch.manual_seed(12)
    ref, _ = hc.make_residual_layer(hc.Bottleneck, 64, 64, 3)
    x0 = torch.randn(2, 12, 8, 64)
    w0 = torch.randn(2, 12, 8, 256)
    res = {}
    for fused in (True, False):
        old = ops.set_fused_bottleneck(fused)
        try:
            m = copy.deepcopy(ref).to(dev).train()
            x = x0.clone().to(dev).requires_grad_(True)
            y = m(x)
            assert (type(y.grad_fn).__name__ == "BottleneckFnBackward") == fused
            y.backward(w0.to(dev))
            torch.cuda.synchronize()
            res[fused] = (y.detach().cpu(), x.grad.cpu(), {k: p.grad.cpu() for k, p in m.named_parameters()},
                          {k: b.cpu() for k, b in m.named_buffers()})
        finally:
            ops.set_fused_bottleneck(old)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for k in res[True][2]:
        assert torch.equal(res[True][2][k], res[False][2][k]), k
    for k in res[True][3]:
        assert torch.equal(res[True][3][k], res[False][3][k]), k


def test_first_transition_as_one_node_is_bit_identical(dev):
    """ops.ForkConvBnFn: the two conv+BN+ReLU heads of transition1 read the same tensor; as one node their data gradients
    chain through a kernel epilogue - bit-identical to the two ConvBnAct nodes plus autograd's accumulation."""
    from buctd_amd import ops
    from buctd_amd.models import hrnet_common as hc
    torch.manual_seed(13)
    trunk = hc.HRNetTrunk()
    trunk.transition1 = hc.make_transition_layer([64], [16, 32])
    trunk.stage2_cfg = {"NUM_BRANCHES": 2}
    x0 = torch.randn(2, 12, 8, 64)
    res = {}
    for fused in (True, False):
        old = ops.set_fused_bottleneck(fused)
        try:
            m = copy.deepcopy(trunk).to(dev).train()
            x = x0.clone().to(dev).requires_grad_(True)
            ys = m.enter_stage(2, x, first=True)
            assert (type(ys[0].grad_fn).__name__ == "ForkConvBnFnBackward") == fused
            ((ys[0] * 0.5).sum() + (ys[1] * torch.linspace(-1, 1, ys[1].numel(), device=dev).view(ys[1].shape)).sum()).backward()
            torch.cuda.synchronize()
            res[fused] = ([y.detach().cpu() for y in ys], x.grad.cpu(), {k: p.grad.cpu() for k, p in m.named_parameters()})
        finally:
            ops.set_fused_bottleneck(old)
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    assert torch.equal(res[True][1], res[False][1])
    for k in res[True][2]:
        assert torch.equal(res[True][2][k], res[False][2][k]), k


def test_fused_nodes_leave_no_cyclic_garbage(dev):
    """The multi-convolution autograd nodes must not keep their outputs alive through a reference cycle (node -> saved
    output -> grad_fn -> node): with the cyclic collector off, device memory is the same after every iteration.  (A first
    version kept the sub-contexts' tensors in Python attributes: 3 GB per CoAM-W48 step, freed only by gc.)"""
    import gc
    from buctd_amd.models import hrnet_common as hc
    torch.manual_seed(14)
    layer, _ = hc.make_residual_layer(hc.Bottleneck, 64, 64, 3)
    trunk = hc.HRNetTrunk()
    trunk.layer1 = layer
    trunk.transition1 = hc.make_transition_layer([256], [16, 32])
    trunk.stage2_cfg = {"NUM_BRANCHES": 2}
    chain, _ = hc.make_residual_layer(hc.BasicBlock, 16, 16, 4)
    trunk.chain = chain
    trunk = trunk.to(dev).train()
    x = torch.randn(2, 24, 16, 64, device=dev)

    def it():
        xi = x.clone().requires_grad_(True)
        ys = trunk.enter_stage(2, trunk.layer1(xi), first=True)
        (trunk.chain(ys[0]).sum() + ys[1].sum()).backward()
        for p in trunk.parameters():
            p.grad = None
        torch.cuda.synchronize()
        return torch.cuda.memory_allocated()

    gc.collect()
    gc.disable()
    try:
        it()
        sizes = [it() for _ in range(4)]
    finally:
        gc.enable()
    assert len(set(sizes)) == 1, f"device memory grows without the cyclic collector: {sizes}"


@pytest.mark.parametrize("nb,mso", [(2, True), (3, True), (4, True), (4, False)])
def test_high_resolution_module(dev, nb, mso):
    from oracle import models as om
    from buctd_amd.models import hrnet_common as hc
    torch.manual_seed(3)
    ch = [16 * 2 ** i for i in range(nb)]
    o = om.HighResolutionModule(nb, om.BasicBlock, [1] * nb, list(ch), list(ch), "SUM", mso)
    p = hc.HighResolutionModule(nb, hc.BasicBlock, [1] * nb, list(ch), list(ch), "SUM", mso)
    _run(dev, f"hr{nb}", o, p, [torch.randn(3, ch[i], 24 >> i, 16 >> i) for i in range(nb)])


@pytest.mark.parametrize("shape", [(2, 24, 18, 48), (3, 12, 9, 96), (20, 96, 72, 48)])
def test_basic_block_input_bn_fusion_is_bit_identical(dev, shape, monkeypatch):
    """bf16x6 mode: conv2 of a BasicBlock applies bn1 + ReLU while staging its input, and its weight gradient rebuilds
    that input the same way (buctd_conv3x3_*_bnin).  Same bits as the unfused sequence conv1 -> bn_apply -> conv2 in the
    forward; gradients equal to the round-off of the BatchNorm-backward sums (formed in another fixed order)."""
    import torch.nn as tnn
    from buctd_amd import ops
    N, H, W, Cn = shape
    assert ops.get_conv_math() == "bf16x6"
    g = torch.Generator().manual_seed(H + Cn)
    x = torch.randn(N, H, W, Cn, generator=g).to(dev)
    w1 = tnn.Parameter((torch.randn(Cn, Cn, 3, 3, generator=g) * 0.08).contiguous(memory_format=torch.channels_last).to(dev))
    w2 = tnn.Parameter((torch.randn(Cn, Cn, 3, 3, generator=g) * 0.08).contiguous(memory_format=torch.channels_last).to(dev))
    dy = torch.randn(N, H, W, Cn, generator=g).to(dev)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setattr(ops, "_FUSE_BN_IN", flag == "1")
        assert ops.bn_in_fusable((N, H, W, Cn), w2) == (flag == "1")
        bns = []
        for s in (1, 2):
            bn = tnn.BatchNorm2d(Cn).to(dev).train()
            with torch.no_grad():
                bn.weight.copy_(torch.rand(Cn, generator=torch.Generator().manual_seed(s)) + 0.5)
                bn.bias.copy_(torch.randn(Cn, generator=torch.Generator().manual_seed(10 + s)) * 0.2)
            bns.append(bn)
        for p in (w1, w2):
            p.grad = None
        xi = x.clone().requires_grad_(True)
        y = ops.BasicBlockFn.apply(xi, w1, bns[0], w2, bns[1])
        y.backward(dy)
        torch.cuda.synchronize()
        res[flag] = [y.detach().clone(), xi.grad.clone(), w1.grad.clone(), w2.grad.clone(), bns[0].weight.grad.clone(),
                     bns[0].bias.grad.clone(), bns[1].weight.grad.clone(), bns[1].running_var.clone()]
    names = ["y", "dx", "dw1", "dw2", "dgamma1", "dbeta1", "dgamma2", "running_var2"]
    for n, a, b in zip(names, res["1"], res["0"]):
        if n in ("y", "running_var2"):
            # the forward: the very same arithmetic per element, same bits
            assert torch.equal(a, b), f"{n}: fused vs unfused differ by {(a - b).abs().max().item():.3e}"
        else:
            # the backward: the native sequence forms the BatchNorm-backward sums in the data-gradient epilogue (per row group
            # of the tile grid, buctd_conv3x3_bf16x6_bnstat), the step-by-step path with bn_bwd_reduce2_kernel (per block of
            # rows) - same element arithmetic, other (fixed) summation order: fp32 round-off of those sums
            sc = b.abs().max().item()
            err = (a - b).abs().max().item()
            assert err <= 5e-6 * sc, f"{n}: native vs step-by-step backward differ by {err:.3e} (scale {sc:.3e})"


def test_block_chain_equals_single_blocks(dev):
    """BasicChainFn (one library call per direction for a whole branch) against n BasicBlockFn calls.  Forward: the same
    launches, same bits.  Backward: inside a chain the sums of bn2's backward of block k - 1 are a by-product of block k's
    conv1 data gradient (block.hip), a single block forms them with bn_bwd_reduce2_kernel: same element arithmetic, another
    fixed summation order - equal to fp32 round-off of those sums, and run-to-run bit-reproducible."""
    import torch.nn as tnn
    from buctd_amd import ops
    N, H, W, Cn, n = 3, 12, 10, 48, 3
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, H, W, Cn, generator=g).to(dev)
    dy = torch.randn(N, H, W, Cn, generator=g).to(dev)
    blocks = []
    for k in range(n):
        ws = [tnn.Parameter((torch.randn(Cn, Cn, 3, 3, generator=g) * 0.08).contiguous(memory_format=torch.channels_last).to(dev))
              for _ in range(2)]
        bns = []
        for s in (1, 2):
            bn = tnn.BatchNorm2d(Cn).to(dev).train()
            with torch.no_grad():
                bn.weight.copy_(torch.rand(Cn, generator=g) + 0.5)
                bn.bias.copy_(torch.randn(Cn, generator=g) * 0.2)
            bns.append(bn)
        blocks.append((ws[0], bns[0], ws[1], bns[1]))
    params = [q for b in blocks for q in (b[0], b[2], b[1].weight, b[1].bias, b[3].weight, b[3].bias)]
    res = {}
    for mode in ("chain", "single", "chain2"):
        for q in params:
            q.grad = None
        for b in blocks:
            for bn in (b[1], b[3]):
                bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        xi = x.clone().requires_grad_(True)
        if mode == "single":
            y = xi
            for b in blocks:
                y = ops.BasicBlockFn.apply(y, *b)
        else:
            assert ops.native_chain_ok(tuple(xi.shape))
            y = ops.BasicChainFn.apply(xi, blocks[0][0], blocks)
        y.backward(dy)
        torch.cuda.synchronize()
        res[mode] = [y.detach().clone(), xi.grad.clone()] + [q.grad.clone() for q in params] + \
            [blocks[-1][3].running_var.clone()]
    for i, (a, b, c) in enumerate(zip(res["chain"], res["single"], res["chain2"])):
        assert torch.equal(a, c), "the chain is not run-to-run reproducible"
        if i == 0 or i == len(res["chain"]) - 1:
            assert torch.equal(a, b), "forward output / running statistics must not change"
        else:
            sc = b.abs().max().item()
            assert (a - b).abs().max().item() <= 5e-6 * sc, f"tensor {i}: chain vs single blocks {(a - b).abs().max().item():.3e} (scale {sc:.3e})"


def test_eval_bn_fold_cache_follows_training_updates(dev):
    """eval-mode scale / shift are folded once per BatchNorm and cached; a train-mode forward (running statistics updated
    through raw pointers) or a parameter update must invalidate them."""
    from buctd_amd import nn as bnn
    import torch.nn.functional as F
    torch.manual_seed(3)
    conv = bnn.Conv2d(16, 32, 3, 1, 1, bias=False).to(dev)
    bn = bnn.BatchNorm2d(32).to(dev)
    bnn.prepare_module(conv)
    x = torch.randn(2, 16, 12, 10, device=dev)

    def run(train):
        from buctd_amd import ops
        y = ops.ConvBnAct.apply(ops.nchw_to_nhwc(x, 0, 16), conv.weight, None, bn, None, True, 1, 1, train, None)
        return ops.nhwc_to_nchw(y)

    def ref():
        z = F.conv2d(x, conv.weight, None, 1, 1)
        return F.relu(F.batch_norm(z, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))

    with torch.no_grad():
        a = run(False)
        assert (a - ref()).abs().max().item() <= 1e-4
        b = run(False)                       # served from the cache
        assert torch.equal(a, b)
        run(True)                            # running statistics move
        c = run(False)
        assert (c - ref()).abs().max().item() <= 1e-4 and not torch.equal(a, c)
        bn.weight.mul_(1.5)                  # parameter update (version bump)
        d = run(False)
        assert (d - ref()).abs().max().item() <= 1e-4 and not torch.equal(c, d)
