"""TEST INFRASTRUCTURE - the CPU oracle of the BUCTD model zoo.  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this package; the product
(buctd_amd/) never does.

Plain-PyTorch (torch.nn, fp32, NCHW) restatement of the reference networks with identical
constructor semantics and state_dict keys/shapes:
  * PoseHighResolutionNet (+preNet fusion)      reference lib/models/pose_hrnet.py:274-623
  * PoseHighResolutionNet + CoAM                reference lib/models/pose_hrnet_coam.py:277-757
  * attention cores                             reference lib/models/self_attention.py:10-160
  * TransPoseH                                  reference lib/models/transpose_h.py:110-243,419-722
  * PoseResNet                                  reference lib/models/pose_resnet.py:103-305
Pinned against the imported reference by oracle/make_golden.py (state_dict key/shape equality and
forward/backward equality on seeded inputs); the results are committed under tests/golden/.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_MOM = 0.1


def resize_noaa(x, size):
    """torchvision 0.9 TF.resize on tensors == bilinear, align_corners=False, no antialias
    (pinned semantics of pose_hrnet_coam.py:755 / transpose_h.py:670)."""
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


def cbr(cin, cout, k, stride, relu, momentum=None):
    """Sequential(conv(no bias), bn[, relu]) with the reference's child indices."""
    bn = nn.BatchNorm2d(cout) if momentum is None else nn.BatchNorm2d(cout, momentum=momentum)
    mods = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=False), bn]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


class BasicBlock(nn.Module):  # pose_hrnet.py:28-57
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOM)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class Bottleneck(nn.Module):  # pose_hrnet.py:60-98
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOM)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOM)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4, momentum=BN_MOM)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


BLOCKS = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def make_layer(block, inplanes, planes, nblocks, stride=1):
    """Residual stack; returns (Sequential, out_channels). pose_hrnet.py:396-413."""
    down = None
    if stride != 1 or inplanes != planes * block.expansion:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                             nn.BatchNorm2d(planes * block.expansion, momentum=BN_MOM))
    layers = [block(inplanes, planes, stride, down)]
    for _ in range(1, nblocks):
        layers.append(block(planes * block.expansion, planes))
    return nn.Sequential(*layers), planes * block.expansion


class HighResolutionModule(nn.Module):  # pose_hrnet.py:101-265
    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels, fuse_method,
                 multi_scale_output=True):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)):
            raise ValueError("branch specification lengths disagree")
        self.num_branches = num_branches
        self.fuse_method = fuse_method
        self.multi_scale_output = multi_scale_output
        self.num_inchannels = list(num_inchannels)
        branches = []
        for i in range(num_branches):
            seq, cout = make_layer(block, self.num_inchannels[i], num_channels[i], num_blocks[i])
            self.num_inchannels[i] = cout
            branches.append(seq)
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = self._fuse()
        self.relu = nn.ReLU(True)

    def _fuse(self):
        if self.num_branches == 1:
            return None
        ch = self.num_inchannels
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(ch[j], ch[i], 1, 1, 0, bias=False), nn.BatchNorm2d(ch[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(None)
                else:
                    steps = [cbr(ch[j], ch[i] if k == i - j - 1 else ch[j], 3, 2, relu=(k != i - j - 1))
                             for k in range(i - j)]
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels

    def forward(self, xs):
        if self.num_branches == 1:
            return [self.branches[0](xs[0])]
        xs = [b(x) for b, x in zip(self.branches, xs)]
        outs = []
        for i, row in enumerate(self.fuse_layers):
            acc = None
            for j in range(self.num_branches):
                t = xs[j] if j == i else row[j](xs[j])
                acc = t if acc is None else acc + t
            outs.append(self.relu(acc))
        return outs


def make_transition(pre, cur):  # pose_hrnet.py:355-393
    layers = []
    for i, c in enumerate(cur):
        if i < len(pre):
            layers.append(cbr(pre[i], c, 3, 1, True) if c != pre[i] else None)
        else:
            steps = []
            for j in range(i + 1 - len(pre)):
                steps.append(cbr(pre[-1], c if j == i - len(pre) else pre[-1], 3, 2, True))
            layers.append(nn.Sequential(*steps))
    return nn.ModuleList(layers)


def make_stage(scfg, num_inchannels, multi_scale_output=True):  # pose_hrnet.py:415-444
    block = BLOCKS[scfg["BLOCK"]]
    mods = []
    for i in range(scfg["NUM_MODULES"]):
        mso = multi_scale_output or i != scfg["NUM_MODULES"] - 1
        mods.append(HighResolutionModule(scfg["NUM_BRANCHES"], block, scfg["NUM_BLOCKS"], num_inchannels,
                                         scfg["NUM_CHANNELS"], scfg["FUSE_METHOD"], mso))
        num_inchannels = mods[-1].get_num_inchannels()
    return nn.Sequential(*mods), num_inchannels


def hrnet_init(model, pretrained, conv_transpose=True, linear=False):
    """init_weights of pose_hrnet.py:578-614 / pose_hrnet_coam.py:574-609."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d) or (conv_transpose and isinstance(m, nn.ConvTranspose2d)) or \
                (linear and isinstance(m, nn.Linear)):
            nn.init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    if os.path.isfile(pretrained):
        sd = torch.load(pretrained)
        keep = {k: v for k, v in sd.items()
                if k.split(".")[0] in model.pretrained_layers or model.pretrained_layers[0] == "*"}
        model.load_state_dict(keep, strict=False)
    elif pretrained:
        raise ValueError("{} is not exist!".format(pretrained))


class _HRNetTrunk(nn.Module):
    """stem + layer1 + stages shared by the three HRNet-based heads."""

    def _build_trunk(self, extra, last_stage=4):
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOM)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOM)
        self.relu = nn.ReLU(inplace=True)
        self.layer1, pre = make_layer(Bottleneck, 64, 64, 4)
        pre = [pre]
        for s in range(2, last_stage + 1):
            scfg = extra["STAGE%d" % s]
            setattr(self, "stage%d_cfg" % s, scfg)
            block = BLOCKS[scfg["BLOCK"]]
            ch = [c * block.expansion for c in scfg["NUM_CHANNELS"]]
            setattr(self, "transition%d" % (s - 1), make_transition(pre, ch))
            stage, pre = make_stage(scfg, ch, multi_scale_output=(s != last_stage))
            setattr(self, "stage%d" % s, stage)
        return pre

    def _stem(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.relu(self.bn2(self.conv2(x)))
        return self.layer1(x)

    def _enter(self, s, prev, first):
        trans = getattr(self, "transition%d" % (s - 1))
        n = getattr(self, "stage%d_cfg" % s)["NUM_BRANCHES"]
        if first:
            return [trans[i](prev) if trans[i] is not None else prev for i in range(n)]
        return [trans[i](prev[-1]) if trans[i] is not None else prev[i] for i in range(n)]


class PoseHighResolutionNet(_HRNetTrunk):  # pose_hrnet.py:274-576
    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = cfg["MODEL"]["EXTRA"]
        self.cfg = cfg
        if cfg.MODEL.EXTRA.USE_PRE_NET:
            self.rgb_preNet = nn.Sequential(nn.Conv2d(3, 64, 3, 1, padding="same"), nn.BatchNorm2d(64),
                                            nn.Conv2d(64, 3, 7, 1, padding="same"), nn.BatchNorm2d(3))
            self.cond_preNet = nn.Sequential(nn.Conv2d(3, 3, 7, 1, padding="same"), nn.BatchNorm2d(3))
        pre = self._build_trunk(extra)
        k = extra["FINAL_CONV_KERNEL"]
        self.final_layer = nn.Conv2d(pre[0], cfg["MODEL"]["NUM_JOINTS"], k, 1, 1 if k == 3 else 0)
        self.pretrained_layers = extra["PRETRAINED_LAYERS"]

    def forward(self, x):
        if self.cfg.MODEL.EXTRA.USE_PRE_NET:
            if x[:, 3:].shape[1] == 0:
                raise Exception("condition is empty, please check your dataloader")
            x = self.rgb_preNet(x[:, :3]) + self.cond_preNet(x[:, 3:])
        x = self._stem(x)
        y = self.stage2(self._enter(2, x, True))
        y = self.stage3(self._enter(3, y, False))
        y = self.stage4(self._enter(4, y, False))
        return self.final_layer(y[0])

    def init_weights(self, pretrained=""):
        hrnet_init(self, pretrained)


# ----------------------------------------------------------------------------- attention ----
def _attn_init(mod):  # self_attention.py:47-59
    for m in mod.modules():
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


class ScaledDotProductAttention(nn.Module):  # self_attention.py:10-88
    def __init__(self, in_dim_q, in_dim_k, d_k, d_v, h, dropout=0.1):
        super().__init__()
        self.fc_q = nn.Linear(in_dim_q, h * d_k)
        self.fc_k = nn.Linear(in_dim_k, h * d_k)
        self.fc_v = nn.Linear(in_dim_k, h * d_v)
        self.fc_o = nn.Linear(h * d_v, in_dim_k)
        self.dropout = nn.Dropout(dropout)
        self.d_model, self.d_k, self.d_v, self.h = in_dim_k, d_k, d_v, h
        _attn_init(self)

    def forward(self, queries, keys, values):
        b, nq = queries.shape[:2]
        nk = keys.shape[1]
        q = self.fc_q(queries).view(b, nq, self.h, self.d_k).permute(0, 2, 1, 3)
        k = self.fc_k(keys).view(b, nk, self.h, self.d_k).permute(0, 2, 3, 1)
        v = self.fc_v(values).view(b, nk, self.h, self.d_v).permute(0, 2, 1, 3)
        att = self.dropout(torch.softmax(torch.matmul(q, k) / math.sqrt(self.d_k), -1))
        out = torch.matmul(att, v).permute(0, 2, 1, 3).contiguous().view(b, nq, self.h * self.d_v)
        return self.fc_o(out)


class SimplifiedScaledDotProductAttention(nn.Module):  # self_attention.py:95-160
    def __init__(self, d_model, h, dropout=0.1):
        super().__init__()
        self.d_model, self.h = d_model, h
        self.d_k = self.d_v = d_model // h
        self.fc_o = nn.Linear(h * self.d_v, d_model)
        self.dropout = nn.Dropout(dropout)
        _attn_init(self)

    def forward(self, queries, keys, values):
        b, nq = queries.shape[:2]
        nk = keys.shape[1]
        q = queries.view(b, nq, self.h, self.d_k).permute(0, 2, 1, 3)
        k = keys.view(b, nk, self.h, self.d_k).permute(0, 2, 3, 1)
        v = values.view(b, nk, self.h, self.d_v).permute(0, 2, 1, 3)
        att = self.dropout(torch.softmax(torch.matmul(q, k) / math.sqrt(self.d_k), -1))
        out = torch.matmul(att, v).permute(0, 2, 1, 3).contiguous().view(b, nq, self.h * self.d_v)
        return self.fc_o(out)


class PositionAttentionModule(nn.Module):  # pose_hrnet_coam.py:631-660
    def __init__(self, d_model=512, d_cond=3, kernel_size=3, H=7, W=7, n_heads=1, self_att=False):
        super().__init__()
        pad = (kernel_size - 1) // 2
        self.cnn = nn.Conv2d(d_model, d_model, kernel_size, padding=pad)
        self.self_att = self_att
        self.register_module("pa", None)  # slot registered before cnn_cond, like the reference module order
        if self_att:
            self.pa = ScaledDotProductAttention(d_model, d_model, d_model, d_model, n_heads)
        else:
            self.cnn_cond = nn.Conv2d(d_cond, d_cond, kernel_size, padding=pad)
            self.pa = ScaledDotProductAttention(d_cond, d_model, d_model, d_model, n_heads)

    def forward(self, x, cond=None):
        b, c = x.shape[:2]
        y = self.cnn(x).view(b, c, -1).permute(0, 2, 1)
        if self.self_att:
            return self.pa(y, y, y)
        yc = self.cnn_cond(cond).view(b, cond.shape[1], -1).permute(0, 2, 1)
        return self.pa(yc, y, y)


class ChannelAttentionModule(nn.Module):  # pose_hrnet_coam.py:662-689
    def __init__(self, d_model=512, d_cond=3, kernel_size=3, H=7, W=7, n_heads=1, self_att=False):
        super().__init__()
        pad = (kernel_size - 1) // 2
        self.cnn = nn.Conv2d(d_model, d_model, kernel_size, padding=pad)
        self.self_att = self_att
        if not self_att:
            self.cnn_cond = nn.Conv2d(d_cond, d_model, kernel_size, padding=pad)
        self.pa = SimplifiedScaledDotProductAttention(H * W, h=n_heads)

    def forward(self, x, cond=None):
        b, c = x.shape[:2]
        y = self.cnn(x).view(b, c, -1)
        if self.self_att:
            return self.pa(y, y, y)
        return self.pa(self.cnn_cond(cond).view(b, c, -1), y, y)


class DAModule(nn.Module):  # pose_hrnet_coam.py:692-725
    def __init__(self, d_model=512, d_cond=3, kernel_size=3, H=7, W=7, n_heads=1, channel_only=False):
        super().__init__()
        self.channel_only = channel_only
        if not channel_only:
            self.position_attention_module = PositionAttentionModule(d_model, d_cond, kernel_size, H, W, n_heads)
        self.channel_attention_module = ChannelAttentionModule(d_model, d_cond, kernel_size, H, W, n_heads)

    def forward(self, inp, cond):
        b, c, h, w = inp.shape
        c_out = self.channel_attention_module(inp, cond).view(b, c, h, w)
        if self.channel_only:
            return inp * c_out
        p_out = self.position_attention_module(inp, cond).permute(0, 2, 1).view(b, c, h, w)
        return inp + (p_out + c_out)


class CoAMBlock(nn.Module):  # pose_hrnet_coam.py:728-757
    def __init__(self, spat_dims, channel_list, cond_stacked, cond_colored, n_heads=1, channel_only=False):
        super().__init__()
        self.spat_dims, self.cond_color, self.cond_stacked = spat_dims, cond_colored, cond_stacked
        d_cond = cond_stacked[1] if cond_stacked[0] else (3 if cond_colored else 1)
        self.att_layers = nn.ModuleList([
            DAModule(channel_list[i], d_cond, 3, H=spat_dims[i][1], W=spat_dims[i][0], n_heads=n_heads,
                     channel_only=channel_only) for i in range(len(spat_dims))])

    def forward(self, ys, cond_hm):
        if not self.cond_color and not self.cond_stacked[0]:
            cond_hm = cond_hm[:, 0].unsqueeze(1)
        return [layer(y, resize_noaa(cond_hm, (sd[1], sd[0])))
                for layer, y, sd in zip(self.att_layers, ys, self.spat_dims)]


class SelfDAModule(nn.Module):  # pose_hrnet_coam.py:761-782
    def __init__(self, d_model=512, kernel_size=3, H=7, W=7):
        super().__init__()
        self.position_attention_module = PositionAttentionModule(d_model, None, kernel_size, H, W, self_att=True)
        self.channel_attention_module = ChannelAttentionModule(d_model, None, kernel_size, H, W, self_att=True)

    def forward(self, inp):
        b, c, h, w = inp.shape
        p = self.position_attention_module(inp).permute(0, 2, 1).view(b, c, h, w)
        return p + self.channel_attention_module(inp).view(b, c, h, w)


class SelfAttentionModule(nn.Module):  # pose_hrnet_coam.py:785-801
    def __init__(self, spat_dims, channel_list):
        super().__init__()
        self.att_layers = nn.ModuleList([SelfDAModule(channel_list[i], 3, H=spat_dims[i][0], W=spat_dims[i][1])
                                         for i in range(len(spat_dims))])

    def forward(self, ys, *args):
        return [layer(y) for layer, y in zip(self.att_layers, ys)]


class PoseHighResolutionNetCoAM(_HRNetTrunk):  # pose_hrnet_coam.py:277-572
    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = cfg["MODEL"]["EXTRA"]
        self.cfg = cfg
        pre = self._build_trunk(extra)
        k = extra["FINAL_CONV_KERNEL"]
        self.final_layer = nn.Conv2d(pre[0], cfg["MODEL"]["NUM_JOINTS"], k, 1, 1 if k == 3 else 0)
        self.pretrained_layers = extra["PRETRAINED_LAYERS"]
        heads = cfg["MODEL"]["ATTENTION_HEADS"]
        self.stage1_att = self.stage2_att = self.stage3_att = self.stage4_att = None
        self.att_config = cfg.MODEL.ATT_MODULES
        self.selfatt_config = cfg.MODEL.SELFATT_MODULES
        iw, ih = cfg.MODEL.IMAGE_SIZE
        dims = [(int(iw / s), int(ih / s)) for s in (4, 8, 16, 32)]
        for a, s in zip(self.att_config, self.selfatt_config):
            assert not a or not s
        chans = [self.stage2_cfg["NUM_CHANNELS"], self.stage3_cfg["NUM_CHANNELS"], self.stage4_cfg["NUM_CHANNELS"],
                 [self.stage4_cfg["NUM_CHANNELS"][0]]]
        spans = [dims[:2], dims[:3], dims[:], [dims[0]]]
        stacked = (cfg["DATASET"]["STACKED_CONDITION"], cfg["MODEL"]["NUM_JOINTS"])
        for i in range(4):
            if self.att_config[i]:
                setattr(self, "stage%d_att" % (i + 1),
                        CoAMBlock(spans[i], chans[i], stacked, cfg["DATASET"]["COLORED"], heads,
                                  cfg["MODEL"]["ATT_CHANNEL_ONLY"]))
            if self.selfatt_config[i]:
                setattr(self, "stage%d_att" % (i + 1), SelfAttentionModule(spans[i], chans[i]))

    def forward(self, x):
        use = self.cfg.MODEL.EXTRA.USE_ATTENTION
        cond = None
        if use:
            if x[:, 3:].shape[1] == 0:
                raise Exception("condition is empty, please check your dataloader")
            x, cond = x[:, :3], x[:, 3:]
        x = self._stem(x)
        xs = self._enter(2, x, True)
        if use and self.att_config[0]:
            xs = self.stage1_att(xs, cond)
        ys = self.stage2(xs)
        xs = self._enter(3, ys, False)
        if use and self.att_config[1]:
            xs = self.stage2_att(xs, cond)
        ys = self.stage3(xs)
        xs = self._enter(4, ys, False)
        if use and self.att_config[2]:
            xs = self.stage3_att(xs, cond)
        ys = self.stage4(xs)
        if use and self.att_config[3]:
            ys = self.stage4_att(ys, cond)
        return self.final_layer(ys[0])

    def init_weights(self, pretrained=""):
        hrnet_init(self, pretrained, linear=True)


# ----------------------------------------------------------------------------- TransPose ----
class TransformerEncoderLayer(nn.Module):  # transpose_h.py:168-243 (post-norm path)
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, src, pos=None):
        q = k = src if pos is None else src + pos
        src = self.norm1(src + self.dropout1(self.self_attn(q, k, value=src)[0]))
        ff = self.linear2(self.dropout(F.relu(self.linear1(src))))
        return self.norm2(src + self.dropout2(ff))


class TransformerEncoder(nn.Module):  # transpose_h.py:110-150
    def __init__(self, d_model, nhead, dim_feedforward, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([TransformerEncoderLayer(d_model, nhead, dim_feedforward)
                                     for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = None
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, pos=None):
        for layer in self.layers:
            src = layer(src, pos=pos)  # pe_only_at_begin=False: PE added in every layer
        return src


def sine_position_embedding(h, w, d_model, temperature=10000, scale=2 * math.pi):
    """transpose_h.py:513-537 -> [h*w, 1, d_model]."""
    ones = torch.ones(1, h, w)
    y = ones.cumsum(1, dtype=torch.float32)
    x = ones.cumsum(2, dtype=torch.float32)
    half = d_model // 2
    eps = 1e-6
    y = y / (y[:, -1:, :] + eps) * scale
    x = x / (x[:, :, -1:] + eps) * scale
    dim_t = torch.arange(half, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / half)
    px = x[:, :, :, None] / dim_t
    py = y[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    pos = torch.cat((py, px), dim=3).permute(0, 3, 1, 2)
    return pos.flatten(2).permute(2, 0, 1)


class TransPoseH(_HRNetTrunk):  # transpose_h.py:419-681
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.cfg = cfg
        extra = cfg["MODEL"]["EXTRA"]
        pre = self._build_trunk(extra, last_stage=3)
        d_model = cfg.MODEL.DIM_MODEL
        w, h = cfg.MODEL.IMAGE_SIZE
        self.reduce = nn.Conv2d(pre[0], d_model, 1, bias=False)
        if cfg.MODEL.EXTRA.USE_ATTENTION:
            self.trans_cond = nn.Conv2d(3, 16, 1, bias=False)
            d_model += 16
        pe = cfg.MODEL.POS_EMBEDDING
        assert pe in ("none", "learnable", "sine")
        self.pe_h, self.pe_w = h // 4, w // 4
        if pe == "none":
            self.pos_embedding = None
        elif pe == "learnable":
            self.pos_embedding = nn.Parameter(torch.randn(self.pe_h * self.pe_w, 1, d_model))
        else:
            self.pos_embedding = nn.Parameter(sine_position_embedding(self.pe_h, self.pe_w, d_model),
                                              requires_grad=False)
        self.global_encoder = TransformerEncoder(d_model, cfg.MODEL.N_HEAD, cfg.MODEL.DIM_FEEDFORWARD,
                                                 cfg.MODEL.ENCODER_LAYERS)
        k = extra["FINAL_CONV_KERNEL"]
        self.final_layer = nn.Conv2d(d_model, cfg["MODEL"]["NUM_JOINTS"], k, 1, 1 if k == 3 else 0)
        self.pretrained_layers = extra["PRETRAINED_LAYERS"]

    def forward(self, x):
        cond = None
        if self.cfg.MODEL.EXTRA.USE_ATTENTION:
            x, cond = x[:, :3], x[:, 3:]
            if cond.shape[1] == 0:
                raise Exception("condition is empty, please check your dataloader")
        x = self._stem(x)
        y = self.stage2(self._enter(2, x, True))
        y = self.stage3(self._enter(3, y, False))
        x = self.reduce(y[0])
        b, c, h, w = x.shape
        if cond is not None:
            xc = self.trans_cond(resize_noaa(cond, (h, w)))
            x = torch.cat((x, xc), dim=1)
            c += xc.shape[1]
        x = x.flatten(2).permute(2, 0, 1)
        x = self.global_encoder(x, pos=self.pos_embedding)
        x = x.permute(1, 2, 0).contiguous().view(b, c, h, w)
        return self.final_layer(x)

    def init_weights(self, pretrained=""):
        hrnet_init(self, pretrained)


# ------------------------------------------------------------------------------- ResNet ----
class ResBasicBlock(BasicBlock):
    pass


RESNET_SPEC = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
               101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


class PoseResNet(nn.Module):  # pose_resnet.py:103-283
    def __init__(self, block, layers, cfg, **kwargs):
        super().__init__()
        extra = cfg.MODEL.EXTRA
        self.cfg = cfg
        self.deconv_with_bias = extra.DECONV_WITH_BIAS
        if extra.USE_PRE_NET:
            self.rgb_preNet = nn.Sequential(nn.Conv2d(3, 64, 7, 1, 3), nn.BatchNorm2d(64),
                                            nn.Conv2d(64, 3, 7, 1, 3), nn.BatchNorm2d(3))
            self.cond_preNet = nn.Sequential(nn.Conv2d(3, 3, 7, 1, 3), nn.BatchNorm2d(3))
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOM)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        c = 64
        self.layer1, c = make_layer(block, c, 64, layers[0])
        self.layer2, c = make_layer(block, c, 128, layers[1], 2)
        self.layer3, c = make_layer(block, c, 256, layers[2], 2)
        self.layer4, c = make_layer(block, c, 512, layers[3], 2)
        mods = []
        for planes, kern in zip(extra.NUM_DECONV_FILTERS, extra.NUM_DECONV_KERNELS):
            pad, opad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[kern]
            mods += [nn.ConvTranspose2d(c, planes, kern, 2, pad, opad, bias=self.deconv_with_bias),
                     nn.BatchNorm2d(planes, momentum=BN_MOM), nn.ReLU(inplace=True)]
            c = planes
        assert len(mods) == 3 * extra.NUM_DECONV_LAYERS
        self.deconv_layers = nn.Sequential(*mods)
        k = extra.FINAL_CONV_KERNEL
        self.final_layer = nn.Conv2d(c, cfg.MODEL.NUM_JOINTS, k, 1, 1 if k == 3 else 0)

    def forward(self, x):
        if self.cfg.MODEL.EXTRA.USE_PRE_NET:
            x = self.rgb_preNet(x[:, :3]) + self.cond_preNet(x[:, 3:])
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.final_layer(self.deconv_layers(x))

    def init_weights(self, pretrained=""):  # pose_resnet.py:237-283 (non-pretrained branch)
        if os.path.isfile(pretrained):
            raise NotImplementedError("oracle: pretrained ResNet loading is not exercised")
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if isinstance(m, nn.ConvTranspose2d) and self.deconv_with_bias:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


def get_pose_net(cfg, is_train, **kwargs):
    """models.<NAME>.get_pose_net(cfg, is_train) of tools/train.py:92-94."""
    name = cfg.MODEL.NAME
    if name == "pose_hrnet":
        model = PoseHighResolutionNet(cfg, **kwargs)
    elif name == "pose_hrnet_coam":
        model = PoseHighResolutionNetCoAM(cfg, **kwargs)
    elif name == "transpose_h":
        model = TransPoseH(cfg, **kwargs)
    elif name == "pose_resnet":
        block, layers = RESNET_SPEC[cfg.MODEL.EXTRA.NUM_LAYERS]
        model = PoseResNet(block, layers, cfg, **kwargs)
    else:
        raise ValueError(name)
    if is_train and cfg["MODEL"]["INIT_WEIGHTS"]:
        model.init_weights(cfg["MODEL"]["PRETRAINED"])
    return model
