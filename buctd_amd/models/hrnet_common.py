"""HRNet building blocks of the MI355X engine (NHWC, fused conv+BN+ReLU kernels).

Module / attribute names and construction order follow the reference so that state_dict keys are
identical (reference lib/models/pose_hrnet.py:28-98 BasicBlock/Bottleneck, 101-265
HighResolutionModule, 355-444 transition/stage builders); the forward passes are re-expressed
on the fused ops: a BasicBlock is two ConvBnAct launches, a fuse row is one FuseSum launch.
"""
import logging
import os

import torch

from .. import nn
from .. import ops

BN_MOMENTUM = 0.1
logger = logging.getLogger(__name__)


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if (self.training and self.downsample is None and self.stride == 1 and x.is_cuda and torch.is_grad_enabled()
                and self.conv1.bias is None and self.conv2.bias is None and self.bn1.training and self.bn2.training):
            nn._as_channels_last_(self.conv1.weight)
            nn._as_channels_last_(self.conv2.weight)
            return ops.BasicBlockFn.apply(x, self.conv1.weight, self.bn1, self.conv2.weight, self.bn2)
        residual = x if self.downsample is None else self.downsample(x)
        out = nn.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        return nn.conv_bn_act(out, self.conv2, self.bn2, relu=True, residual=residual)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # the one-node path runs every inner BatchNorm in train mode: a BatchNorm put in eval on its own (frozen statistics
        # while the block trains) takes the layer-by-layer path, which honours bn.training per layer
        bns = (self.bn1, self.bn2, self.bn3) + (() if self.downsample is None or len(self.downsample) < 2 else (self.downsample[1],))
        if (self.training and x.is_cuda and torch.is_grad_enabled() and ops.fused_bottleneck_on()
                and all(b.training for b in bns)
                and (self.downsample is None or type(self.downsample) is nn.ConvBN and not self.downsample._relu)):
            for c in (self.conv1, self.conv2, self.conv3) + (() if self.downsample is None else (self.downsample[0],)):
                nn._as_channels_last_(c.weight)
            return ops.BottleneckFn.apply(x, self.conv1.weight, self)
        residual = x if self.downsample is None else self.downsample(x)
        out = nn.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        out = nn.conv_bn_act(out, self.conv2, self.bn2, relu=True)
        return nn.conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=residual)


class BlockChain(nn.Chain):
    """nn.Sequential of residual blocks (same child names as the reference's: '0', '1', ...).  A chain of plain BasicBlocks
    in a training forward runs as ONE autograd node / one library call per direction (ops.BasicChainFn)."""

    def forward(self, x):
        if (len(self) > 1 and self.training and x.is_cuda and torch.is_grad_enabled() and ops.native_chain_ok(tuple(x.shape))
                and all(type(m) is BasicBlock and m.downsample is None and m.stride == 1 and m.conv1.bias is None
                        and m.conv2.bias is None and m.bn1.training and m.bn2.training
                        and m.bn1.track_running_stats == m.bn2.track_running_stats == self[0].bn1.track_running_stats for m in self)
                and ops.bn_in_fusable(tuple(x.shape), self[0].conv1.weight)):
            blocks = []
            for m in self:
                nn._as_channels_last_(m.conv1.weight)
                nn._as_channels_last_(m.conv2.weight)
                blocks.append((m.conv1.weight, m.bn1, m.conv2.weight, m.bn2))
            return ops.BasicChainFn.apply(x, self[0].conv1.weight, blocks)
        return super().forward(x)


blocks_dict = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def make_residual_layer(block, inplanes, planes, blocks, stride=1):
    """-> (nn.Chain of blocks, output channels)."""
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.ConvBN(
            nn.Conv2d(inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
            nn.BatchNorm2d(planes * block.expansion, momentum=BN_MOMENTUM))
    layers = [block(inplanes, planes, stride, downsample)]
    inplanes = planes * block.expansion
    for _ in range(1, blocks):
        layers.append(block(inplanes, planes))
    return BlockChain(*layers), inplanes


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, fuse_method,
                 multi_scale_output=True):
        super().__init__()
        self._check_branches(num_branches, blocks, num_blocks, num_inchannels, num_channels)
        self.num_inchannels = num_inchannels
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = self._make_branches(num_branches, blocks, num_blocks, num_channels)
        self.fuse_layers = self._make_fuse_layers()
        self.relu = nn.ReLU(True)

    def _check_branches(self, num_branches, blocks, num_blocks, num_inchannels, num_channels):
        for what, lst in (("NUM_BLOCKS", num_blocks), ("NUM_CHANNELS", num_channels),
                          ("NUM_INCHANNELS", num_inchannels)):
            if num_branches != len(lst):
                msg = "NUM_BRANCHES({}) <> {}({})".format(num_branches, what, len(lst))
                logger.error(msg)
                raise ValueError(msg)

    def _make_branches(self, num_branches, block, num_blocks, num_channels):
        branches = []
        for i in range(num_branches):
            chain, cout = make_residual_layer(block, self.num_inchannels[i], num_channels[i], num_blocks[i])
            self.num_inchannels[i] = cout
            branches.append(chain)
        return nn.ModuleList(branches)

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        ch = self.num_inchannels
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:
                    row.append(nn.ConvBN(nn.Conv2d(ch[j], ch[i], 1, 1, 0, bias=False), nn.BatchNorm2d(ch[i]),
                                         nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(None)
                else:
                    steps = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        cout = ch[i] if last else ch[j]
                        steps.append(nn.ConvBN(nn.Conv2d(ch[j], cout, 3, 2, 1, bias=False), nn.BatchNorm2d(cout),
                                               None if last else nn.ReLU(True)))
                    row.append(nn.Chain(*steps))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels

    def _branches_grouped(self, x):
        """All branches as ONE autograd node whose k-th convolutions share a launch (ops.BasicBranchesFn), or None when the
        module does not qualify (eval mode, other block types, shapes without a native path)."""
        nb = self.num_branches
        if not (self.training and torch.is_grad_enabled() and all(type(br) is BlockChain and len(br) > 1 for br in self.branches)):
            return None
        chains = []
        for br in self.branches:
            if not all(type(m) is BasicBlock and m.downsample is None and m.stride == 1 and m.conv1.bias is None
                       and m.conv2.bias is None and m.bn1.track_running_stats == m.bn2.track_running_stats
                       == br[0].bn1.track_running_stats and m.bn1.training and m.bn2.training for m in br):
                return None
            chains.append([(m.conv1.weight, m.bn1, m.conv2.weight, m.bn2) for m in br])
        xs = [x[i] for i in range(nb)]
        if not ops.group_branches_ok(xs, chains):
            return None
        for chain in chains:
            for (w1, _, w2, _) in chain:
                nn._as_channels_last_(w1)
                nn._as_channels_last_(w2)
        parts = ops.group_branch_partition(nb)
        if len(parts) == 1:
            return list(ops.BasicBranchesFn.apply(chains[0][0][0], chains, *xs))
        # several groups side by side on the branch streams: the streaming BatchNorm kernels of one group run under the
        # convolutions of the other
        outs = ops.fork_join([lambda p=p: ops.BasicBranchesFn.apply(chains[p[0]][0][0], [chains[i] for i in p], *[xs[i] for i in p])
                              for p in parts], [[xs[i] for i in p] for p in parts])
        ys = [None] * nb
        for p, o in zip(parts, outs):
            for i, y in zip(p, o):
                ys[i] = y
        return ys

    def _branches_grouped_eval(self, x):
        """Eval mode: the k-th convolutions of all branches in one launch (ops.basic_branches_eval), or None."""
        if self.training or torch.is_grad_enabled() or not all(type(br) is BlockChain and len(br) > 0 for br in self.branches):
            return None
        chains = []
        for br in self.branches:
            if not all(type(m) is BasicBlock and m.downsample is None and m.stride == 1 and m.conv1.bias is None
                       and m.conv2.bias is None and not m.bn1.training and not m.bn2.training
                       and m.bn1.running_mean is not None and m.bn2.running_mean is not None for m in br):
                return None
            chains.append([(m.conv1.weight, m.bn1, m.conv2.weight, m.bn2) for m in br])
        xs = [x[i] for i in range(self.num_branches)]
        for chain in chains:
            for (w1, _, w2, _) in chain:
                nn._as_channels_last_(w1)
                nn._as_channels_last_(w2)
        if not ops.eval_branches_ok(xs, chains):
            return None
        return ops.basic_branches_eval(xs, chains)

    def forward(self, x):
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        xs = self._branches_grouped(x)
        if xs is None:
            xs = self._branches_grouped_eval(x)
        if xs is None:
            xs = ops.fork_join([lambda i=i: self.branches[i](x[i]) for i in range(self.num_branches)],
                               [x[i] for i in range(self.num_branches)])
        rows = list(enumerate(self.fuse_layers))
        # every branch output feeds every fuse row: with gradients wanted, hand out aliases whose gradients are summed by one
        # n-ary add (ops.FanOut) instead of autograd's chain of two-operand adds
        fan = len(rows) > 1 and torch.is_grad_enabled() and any(t.requires_grad for t in xs)
        xr = [ops.FanOut.apply(xs[j], len(rows)) if fan and xs[j].is_cuda else [xs[j]] * len(rows)
              for j in range(self.num_branches)]

        def fuse_row(i, row):
            terms, shifts = [], []
            for j in range(self.num_branches):
                if j == i:
                    terms.append(xr[j][i])
                    shifts.append(0)
                elif j > i:
                    terms.append(row[j](xr[j][i]))   # 1x1 conv + BN at the low resolution
                    shifts.append(j - i)             # nearest up-sampling happens inside the fuse kernel
                else:
                    terms.append(row[j](xr[j][i]))
                    shifts.append(0)
            return ops.FuseSum.apply(tuple(shifts), True, *terms)

        return ops.fork_join([lambda i=i, row=row: fuse_row(i, row) for i, row in rows], [xs for _ in rows], tag=1)


def make_transition_layer(pre, cur):
    layers = []
    for i in range(len(cur)):
        if i < len(pre):
            if cur[i] != pre[i]:
                layers.append(nn.ConvBN(nn.Conv2d(pre[i], cur[i], 3, 1, 1, bias=False), nn.BatchNorm2d(cur[i]),
                                        nn.ReLU(inplace=True)))
            else:
                layers.append(None)
        else:
            steps = []
            for j in range(i + 1 - len(pre)):
                cin = pre[-1]
                cout = cur[i] if j == i - len(pre) else cin
                steps.append(nn.ConvBN(nn.Conv2d(cin, cout, 3, 2, 1, bias=False), nn.BatchNorm2d(cout),
                                       nn.ReLU(inplace=True)))
            layers.append(nn.Chain(*steps))
    return nn.ModuleList(layers)


def make_stage(layer_config, num_inchannels, multi_scale_output=True):
    block = blocks_dict[layer_config["BLOCK"]]
    modules = []
    for i in range(layer_config["NUM_MODULES"]):
        mso = multi_scale_output or i != layer_config["NUM_MODULES"] - 1
        modules.append(HighResolutionModule(layer_config["NUM_BRANCHES"], block, layer_config["NUM_BLOCKS"],
                                            num_inchannels, layer_config["NUM_CHANNELS"],
                                            layer_config["FUSE_METHOD"], mso))
        num_inchannels = modules[-1].get_num_inchannels()
    return nn.Chain(*modules), num_inchannels


class HRNetTrunk(nn.Module):
    """stem + layer1 + stage2..N; subclasses add heads.  All tensors below the API boundary are NHWC."""

    def build_trunk(self, extra, last_stage=4):
        self.conv1 = nn.Conv2d(3, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(64, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.layer1, c = make_residual_layer(Bottleneck, 64, 64, 4)
        pre = [c]
        for s in range(2, last_stage + 1):
            scfg = extra["STAGE%d" % s]
            setattr(self, "stage%d_cfg" % s, scfg)
            block = blocks_dict[scfg["BLOCK"]]
            ch = [c * block.expansion for c in scfg["NUM_CHANNELS"]]
            setattr(self, "transition%d" % (s - 1), make_transition_layer(pre, ch))
            stage, pre = make_stage(scfg, ch, multi_scale_output=(s != last_stage))
            setattr(self, "stage%d" % s, stage)
        return pre

    def stem(self, x_nhwc):
        x = nn.conv_bn_act(x_nhwc, self.conv1, self.bn1, relu=True)
        x = nn.conv_bn_act(x, self.conv2, self.bn2, relu=True)
        return self.layer1(x)

    def enter_stage(self, s, prev, first=False):
        trans = getattr(self, "transition%d" % (s - 1))
        n = getattr(self, "stage%d_cfg" % s)["NUM_BRANCHES"]
        if first:
            heads = [trans[i] if isinstance(trans[i], nn.ConvBN) else trans[i][0] if isinstance(trans[i], nn.Chain) else None
                     for i in range(n)]
            if (self.training and prev.is_cuda and torch.is_grad_enabled() and ops.fused_bottleneck_on() and n > 1
                    and all(type(h) is nn.ConvBN and h[1].training for h in heads)):
                # every new branch starts with a conv + BN + ReLU on the same tensor: one autograd node (ops.ForkConvBnFn)
                for h in heads:
                    nn._as_channels_last_(h[0].weight)
                ys = ops.ForkConvBnFn.apply(prev, heads[0][0].weight, [(h[0], h[1], h._relu) for h in heads])
                out = []
                for i in range(n):
                    y = ys[i]
                    if isinstance(trans[i], nn.Chain):
                        for m in list(trans[i])[1:]:
                            y = m(y)
                    out.append(y)
                return out
            return [trans[i](prev) if trans[i] is not None else prev for i in range(n)]
        return [trans[i](prev[-1]) if trans[i] is not None else prev[i] for i in range(n)]


def to_device_input(x):
    """The reference forwards call x.cuda() themselves (pose_hrnet_coam.py:495): CPU inputs are accepted."""
    if not x.is_cuda:
        x = x.cuda()
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous()


def init_weights_hrnet(model, pretrained="", linear=False):
    """normal(std=0.001) conv (+Linear for the CoAM variant) weights, zero biases, BN 1/0, then the
    non-strict filtered load of ImageNet weights (pose_hrnet.py:578-614, pose_hrnet_coam.py:574-609)."""
    logger.info("=> init weights from normal distribution")
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) or (linear and isinstance(m, torch.nn.Linear)):
            torch.nn.init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                torch.nn.init.constant_(m.bias, 0)
        elif isinstance(m, torch.nn.BatchNorm2d):
            torch.nn.init.constant_(m.weight, 1)
            torch.nn.init.constant_(m.bias, 0)
    if os.path.isfile(pretrained):
        state = torch.load(pretrained, map_location="cpu")
        logger.info("=> loading pretrained model {}".format(pretrained))
        keep = {k: v for k, v in state.items()
                if k.split(".")[0] in model.pretrained_layers or model.pretrained_layers[0] == "*"}
        model.load_state_dict(keep, strict=False)
    elif pretrained:
        logger.error("=> please download pre-trained models first!")
        raise ValueError("{} is not exist!".format(pretrained))
