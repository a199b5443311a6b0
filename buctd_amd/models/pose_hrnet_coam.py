"""BUCTD-CoAM (HRNet + conditional attention) on the MI355X engine.

Drop-in for reference lib/models/pose_hrnet_coam.py: PoseHighResolutionNet (277-572),
PositionAttentionModule (631-660), ChannelAttentionModule (662-689), DAModule (692-725), CoAMBlock (728-757),
SelfDAModule / SelfAttentionModule (761-801), get_pose_net (612-618).  Same constructors and state_dict keys.
Differences are in execution only: NHWC tensors, the condition is resized straight from the NCHW input slice
by the bilinear kernel (TF.resize without antialias), and p_out + c_out + input is one fused add.
"""
from .. import nn
from .. import ops
from .hrnet_common import HRNetTrunk, init_weights_hrnet, to_device_input
from .self_attention import ScaledDotProductAttention, SimplifiedScaledDotProductAttention


class PositionAttentionModule(nn.Module):
    def __init__(self, d_model=512, d_cond=3, kernel_size=3, H=7, W=7, n_heads=1, self_att=False):
        super().__init__()
        pad = (kernel_size - 1) // 2
        self.cnn = nn.Conv2d(d_model, d_model, kernel_size=kernel_size, padding=pad)
        self.register_module("pa", None)
        self.self_att = self_att
        if self_att:
            self.pa = ScaledDotProductAttention(in_dim_q=d_model, in_dim_k=d_model, d_k=d_model, d_v=d_model, h=n_heads)
        else:
            self.cnn_cond = nn.Conv2d(d_cond, d_cond, kernel_size=kernel_size, padding=pad)
            self.pa = ScaledDotProductAttention(in_dim_q=d_cond, in_dim_k=d_model, d_k=d_model, d_v=d_model, h=n_heads)

    def forward(self, x, cond=None):
        b, h, w, c = x.shape
        y = self.cnn(x).view(b, h * w, c)
        if self.self_att:
            return self.pa(y, y, y)
        y_cond = self.cnn_cond(cond).view(b, h * w, cond.shape[3])
        return self.pa(y_cond, y, y)  # [B, T, C]


class ChannelAttentionModule(nn.Module):
    def __init__(self, d_model=512, d_cond=3, kernel_size=3, H=7, W=7, n_heads=1, self_att=False):
        super().__init__()
        pad = (kernel_size - 1) // 2
        self.cnn = nn.Conv2d(d_model, d_model, kernel_size=kernel_size, padding=pad)
        self.self_att = self_att
        if not self_att:
            self.cnn_cond = nn.Conv2d(d_cond, d_model, kernel_size=kernel_size, padding=pad)
        self.pa = SimplifiedScaledDotProductAttention(H * W, h=n_heads)

    def forward(self, x, cond=None):
        b, h, w, c = x.shape
        y = self.cnn(x).view(b, h * w, c)
        if self.self_att:
            return self.pa(y, y)
        y_cond = self.cnn_cond(cond).view(b, h * w, c)
        return self.pa(y_cond, y)  # [B, T, C] == the reference's [B, C, T] output, token-major


class DAModule(nn.Module):
    def __init__(self, d_model=512, d_cond=3, kernel_size=3, H=7, W=7, n_heads=1, channel_only=False):
        super().__init__()
        self.channel_only = channel_only
        if not channel_only:
            self.position_attention_module = PositionAttentionModule(d_model=d_model, d_cond=d_cond,
                                                                     kernel_size=kernel_size, H=H, W=W,
                                                                     n_heads=n_heads)
        self.channel_attention_module = ChannelAttentionModule(d_model=d_model, d_cond=d_cond,
                                                               kernel_size=kernel_size, H=H, W=W, n_heads=n_heads)

    def parts(self):
        """the independent attention cores of this module, as callables (input, cond) -> [B, T, C]"""
        if self.channel_only:
            return [self.channel_attention_module]
        return [self.channel_attention_module, self.position_attention_module]

    def combine(self, input, outs):
        b, h, w, c = input.shape
        if self.channel_only:  # MODEL.ATT_CHANNEL_ONLY: the channel core gates the input (pose_hrnet_coam.py:716-717)
            return ops.Mul.apply(input, outs[0].view(b, h, w, c))
        c_out, p_out = outs
        return ops.AddN.apply(input, p_out.view(b, h, w, c), c_out.view(b, h, w, c))

    def forward(self, input, cond):
        return self.combine(input, [part(input, cond) for part in self.parts()])


class CoAMBlock(nn.Module):
    def __init__(self, spat_dims, channel_list, cond_stacked, cond_colored, n_heads=1, channel_only=False):
        super().__init__()
        self.spat_dims = spat_dims
        self.cond_color = cond_colored
        self.cond_stacked = cond_stacked
        if cond_stacked[0]:
            d_cond = cond_stacked[1]
        elif cond_colored:
            d_cond = 3
        else:
            d_cond = 1
        self.d_cond = d_cond
        self.att_layers = nn.ModuleList([
            DAModule(d_model=channel_list[i], d_cond=d_cond, kernel_size=3, H=spat_dims[i][1], W=spat_dims[i][0],
                     n_heads=n_heads, channel_only=channel_only) for i in range(len(spat_dims))])

    def forward(self, y_list, x_nchw):
        """x_nchw: the full NCHW network input; the condition is its channel slice [3, 3 + d_cond)
        (mono: 'we only want one channel of the heatmap', pose_hrnet_coam.py:751-752)."""
        conds = []
        for i in range(len(y_list)):
            hh, ww = self.spat_dims[i][1], self.spat_dims[i][0]
            conds.append(ops.resize_bilinear_from_nchw(x_nchw, 3, self.d_cond, hh, ww))
        # the channel cores (token-contraction GEMMs + the 191 MB fc_o stream) and the position cores (fused attention,
        # VALU/MFMA bound) overlap on two HIP streams: +1.2 %.  (One stream per core - six - was measured and is a
        # loss: the fused position attention streams its K/V tiles out of L2 and co-running kernels evict them.)
        n = len(y_list)

        def chan():
            return [self.att_layers[i].parts()[0](y_list[i], conds[i]) for i in range(n)]

        if self.att_layers[0].channel_only:
            c_outs = chan()
            return [self.att_layers[i].combine(y_list[i], [c_outs[i]]) for i in range(n)]

        def pos():
            return [self.att_layers[i].parts()[1](y_list[i], conds[i]) for i in range(n)]

        ins = list(y_list) + conds
        c_outs, p_outs = ops.fork_join([chan, pos], [ins, ins], tag=3)
        return [self.att_layers[i].combine(y_list[i], [c_outs[i], p_outs[i]]) for i in range(n)]


class SelfDAModule(nn.Module):
    def __init__(self, d_model=512, kernel_size=3, H=7, W=7):
        super().__init__()
        self.position_attention_module = PositionAttentionModule(d_model=d_model, d_cond=None,
                                                                 kernel_size=kernel_size, H=H, W=W, self_att=True)
        self.channel_attention_module = ChannelAttentionModule(d_model=d_model, d_cond=None, kernel_size=kernel_size,
                                                               H=H, W=W, self_att=True)

    def forward(self, input):
        b, h, w, c = input.shape
        p_out = self.position_attention_module(input).view(b, h, w, c)
        c_out = self.channel_attention_module(input).view(b, h, w, c)
        return ops.AddN.apply(p_out, c_out, None)


class SelfAttentionModule(nn.Module):
    def __init__(self, spat_dims, channel_list):
        super().__init__()
        self.att_layers = nn.ModuleList([SelfDAModule(d_model=channel_list[i], kernel_size=3, H=spat_dims[i][0],
                                                      W=spat_dims[i][1]) for i in range(len(spat_dims))])

    def forward(self, y_list, *args):
        return [self.att_layers[i](y) for i, y in enumerate(y_list)]


class PoseHighResolutionNet(HRNetTrunk):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.inplanes = 64
        extra = cfg["MODEL"]["EXTRA"]
        self.cfg = cfg
        pre = self.build_trunk(extra)
        k = extra["FINAL_CONV_KERNEL"]
        self.final_layer = nn.Conv2d(in_channels=pre[0], out_channels=cfg["MODEL"]["NUM_JOINTS"], kernel_size=k,
                                     stride=1, padding=1 if k == 3 else 0)
        self.pretrained_layers = extra["PRETRAINED_LAYERS"]

        att_heads = self.cfg["MODEL"]["ATTENTION_HEADS"]
        self.stage1_att = None
        self.stage2_att = None
        self.stage3_att = None
        self.stage4_att = None
        self.att_config = cfg.MODEL.ATT_MODULES
        self.selfatt_config = cfg.MODEL.SELFATT_MODULES
        iw, ih = cfg.MODEL.IMAGE_SIZE[0], cfg.MODEL.IMAGE_SIZE[1]
        spat_dims = [(int(iw / s), int(ih / s)) for s in (4, 8, 16, 32)]
        for a, s in zip(self.att_config, self.selfatt_config):
            assert not a or not s
        spans = [spat_dims[:2], spat_dims[:3], spat_dims[:], [spat_dims[0]]]
        chans = [self.stage2_cfg["NUM_CHANNELS"], self.stage3_cfg["NUM_CHANNELS"], self.stage4_cfg["NUM_CHANNELS"],
                 [self.stage4_cfg["NUM_CHANNELS"][0]]]
        stacked = (self.cfg["DATASET"]["STACKED_CONDITION"], self.cfg["MODEL"]["NUM_JOINTS"])
        for i in range(4):
            if self.att_config[i]:
                setattr(self, "stage%d_att" % (i + 1),
                        CoAMBlock(spat_dims=spans[i], channel_list=chans[i], cond_stacked=stacked,
                                  cond_colored=self.cfg["DATASET"]["COLORED"], n_heads=att_heads,
                                  channel_only=self.cfg["MODEL"]["ATT_CHANNEL_ONLY"]))
        for i in range(4):
            if self.selfatt_config[i]:
                setattr(self, "stage%d_att" % (i + 1), SelfAttentionModule(spat_dims=spans[i], channel_list=chans[i]))

    def forward(self, x, lambda_vec=None):
        x = to_device_input(x)
        use_att = self.cfg.MODEL.EXTRA.USE_ATTENTION
        if use_att and x.shape[1] - 3 <= 0:
            raise Exception("condition is empty, please check your dataloader")
        # without attention the reference feeds every input channel to conv1 (pose_hrnet_coam.py:502-505)
        feat = self.stem(ops.nchw_to_nhwc(x, 0, 3 if use_att else x.shape[1]))
        x_list = self.enter_stage(2, feat, first=True)
        if use_att and self.att_config[0]:
            x_list = self.stage1_att(x_list, x)
        y_list = self.stage2(x_list)
        x_list = self.enter_stage(3, y_list)
        if use_att and self.att_config[1]:
            x_list = self.stage2_att(x_list, x)
        y_list = self.stage3(x_list)
        x_list = self.enter_stage(4, y_list)
        if use_att and self.att_config[2]:
            x_list = self.stage3_att(x_list, x)
        y_list = self.stage4(x_list)
        if use_att and self.att_config[3]:
            y_list = self.stage4_att(y_list, x)
        return ops.ToNCHW.apply(self.final_layer(y_list[0]))

    def init_weights(self, pretrained=""):
        init_weights_hrnet(self, pretrained, linear=True)


def get_pose_net(cfg, is_train, **kwargs):
    model = PoseHighResolutionNet(cfg, **kwargs)
    if is_train and cfg["MODEL"]["INIT_WEIGHTS"]:
        model.init_weights(cfg["MODEL"]["PRETRAINED"])
    return model
