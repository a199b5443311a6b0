"""BUCTD-preNet / plain HRNet on the MI355X engine.

Drop-in for reference lib/models/pose_hrnet.py (PoseHighResolutionNet 274-576, get_pose_net 617-623):
same constructor, same state_dict keys, NCHW fp32 in -> NCHW fp32 heat-maps out; inside, NHWC fused HIP
kernels.  preNet fusion (431-442, 452-458): x = rgb_preNet(x[:, :3]) + cond_preNet(x[:, 3:]), the sum
being folded into the second branch's BN epilogue as a residual.
"""
from .. import nn
from .. import ops
from .hrnet_common import HRNetTrunk, init_weights_hrnet, to_device_input


class PoseHighResolutionNet(HRNetTrunk):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.inplanes = 64
        extra = cfg["MODEL"]["EXTRA"]
        self.cfg = cfg
        if cfg.MODEL.EXTRA.USE_PRE_NET:
            self.rgb_preNet = self._make_preNet(3, input_image=True)
            self.cond_preNet = self._make_preNet(3, input_image=False)
        pre = self.build_trunk(extra)
        k = extra["FINAL_CONV_KERNEL"]
        self.final_layer = nn.Conv2d(in_channels=pre[0], out_channels=cfg["MODEL"]["NUM_JOINTS"], kernel_size=k,
                                     stride=1, padding=1 if k == 3 else 0)
        self.pretrained_layers = extra["PRETRAINED_LAYERS"]

    def _make_preNet(self, num_outputs, input_image=False):
        if not input_image:
            return nn.Sequential(nn.Conv2d(3, num_outputs, kernel_size=7, stride=1, padding="same"),
                                 nn.BatchNorm2d(num_outputs))
        return nn.Sequential(nn.Conv2d(3, 64, kernel_size=3, stride=1, padding="same"), nn.BatchNorm2d(64),
                             nn.Conv2d(64, num_outputs, kernel_size=7, stride=1, padding="same"),
                             nn.BatchNorm2d(num_outputs))

    def _features(self, x):
        """NCHW input -> list of NHWC stage-4 outputs."""
        x = to_device_input(x)
        if self.cfg.MODEL.EXTRA.USE_PRE_NET:
            if x.shape[1] - 3 <= 0:
                raise Exception("condition is empty, please check your dataloader")
            rgb = ops.nchw_to_nhwc(x, 0, 3)
            cond = ops.nchw_to_nhwc(x, 3, x.shape[1] - 3)
            r, c = self.rgb_preNet, self.cond_preNet
            x0 = nn.conv_bn_act(rgb, r[0], r[1])
            x0 = nn.conv_bn_act(x0, r[2], r[3])
            xin = nn.conv_bn_act(cond, c[0], c[1], residual=x0)  # x0 + x1
        else:
            xin = ops.nchw_to_nhwc(x, 0, 3) if x.shape[1] != 3 else ops.nchw_to_nhwc(x)
        x = self.stem(xin)
        y = self.stage2(self.enter_stage(2, x, first=True))
        y = self.stage3(self.enter_stage(3, y))
        return self.stage4(self.enter_stage(4, y))

    def forward(self, x, forward_feature=False, mu=None, sigma=None):
        if mu is not None:
            raise NotImplementedError("forward_lamda (MIPNet leftover, pose_hrnet.py:497-540) is not on the BUCTD path")
        y = self._features(x)
        if forward_feature:
            return ops.ToNCHW.apply(y[0])
        return ops.ToNCHW.apply(self.final_layer(y[0]))

    def init_weights(self, pretrained=""):
        init_weights_hrnet(self, pretrained)


def get_pose_net(cfg, is_train, **kwargs):
    model = PoseHighResolutionNet(cfg, **kwargs)
    if is_train and cfg["MODEL"]["INIT_WEIGHTS"]:
        model.init_weights(cfg["MODEL"]["PRETRAINED"])
    return model
