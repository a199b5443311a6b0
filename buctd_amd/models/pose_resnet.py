"""SimpleBaseline ResNet (+optional preNet) on the MI355X engine - drop-in for reference lib/models/pose_resnet.py
(PoseResNet 103-283, resnet_spec 286-292, get_pose_net 295-305): conv7x7/2 + BN + ReLU + max-pool, 4 residual
stages, 3 x (ConvTranspose2d 4x4/2 + BN + ReLU) executed by the transposed-conv (dgrad) kernel with fused BN
statistics, 1x1 head."""
import logging
import os

import torch

from .. import nn
from .. import ops
from .. import ops_seq
from .hrnet_common import BN_MOMENTUM, BasicBlock, Bottleneck, make_residual_layer, to_device_input

logger = logging.getLogger(__name__)


class PoseResNet(nn.Module):
    def __init__(self, block, layers, cfg, **kwargs):
        super().__init__()
        extra = cfg.MODEL.EXTRA
        self.inplanes = 64
        self.deconv_with_bias = extra.DECONV_WITH_BIAS
        self.cfg = cfg
        if cfg.MODEL.EXTRA.USE_PRE_NET:
            self.rgb_preNet = nn.Sequential(nn.Conv2d(3, 64, kernel_size=7, stride=1, padding=3), nn.BatchNorm2d(64),
                                            nn.Conv2d(64, 3, kernel_size=7, stride=1, padding=3), nn.BatchNorm2d(3))
            self.cond_preNet = nn.Sequential(nn.Conv2d(3, 3, kernel_size=7, stride=1, padding=3), nn.BatchNorm2d(3))
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        c = 64
        self.layer1, c = make_residual_layer(block, c, 64, layers[0])
        self.layer2, c = make_residual_layer(block, c, 128, layers[1], stride=2)
        self.layer3, c = make_residual_layer(block, c, 256, layers[2], stride=2)
        self.layer4, c = make_residual_layer(block, c, 512, layers[3], stride=2)
        self.inplanes = c
        self.deconv_layers = self._make_deconv_layer(extra.NUM_DECONV_LAYERS, extra.NUM_DECONV_FILTERS,
                                                     extra.NUM_DECONV_KERNELS)
        k = extra.FINAL_CONV_KERNEL
        self.final_layer = nn.Conv2d(in_channels=extra.NUM_DECONV_FILTERS[-1], out_channels=cfg.MODEL.NUM_JOINTS,
                                     kernel_size=k, stride=1, padding=1 if k == 3 else 0)

    def _make_deconv_layer(self, num_layers, num_filters, num_kernels):
        assert num_layers == len(num_filters) == len(num_kernels)
        layers = []
        for planes, kern in zip(num_filters, num_kernels):
            pad, opad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[kern]
            layers += [nn.ConvTranspose2d(self.inplanes, planes, kernel_size=kern, stride=2, padding=pad,
                                          output_padding=opad, bias=self.deconv_with_bias),
                       nn.BatchNorm2d(planes, momentum=BN_MOMENTUM), nn.ReLU(inplace=True)]
            self.inplanes = planes
        return nn.Sequential(*layers)

    def forward(self, x):
        x = to_device_input(x)
        if self.cfg.MODEL.EXTRA.USE_PRE_NET:
            r, c = self.rgb_preNet, self.cond_preNet
            x0 = nn.conv_bn_act(ops.nchw_to_nhwc(x, 0, 3), r[0], r[1])
            x0 = nn.conv_bn_act(x0, r[2], r[3])
            t = nn.conv_bn_act(ops.nchw_to_nhwc(x, 3, x.shape[1] - 3), c[0], c[1], residual=x0)
        else:
            t = ops.nchw_to_nhwc(x)
        t = nn.conv_bn_act(t, self.conv1, self.bn1, relu=True)
        t = ops_seq.MaxPool3x3s2.apply(t)
        t = self.layer4(self.layer3(self.layer2(self.layer1(t))))
        d = self.deconv_layers
        for i in range(0, len(d), 3):
            t = nn.conv_bn_act(t, d[i], d[i + 1], relu=True)
        return ops.ToNCHW.apply(self.final_layer(t))

    def init_weights(self, pretrained=''):
        if os.path.isfile(pretrained):
            for m in self.deconv_layers.modules():
                if isinstance(m, torch.nn.ConvTranspose2d):
                    torch.nn.init.normal_(m.weight, std=0.001)
                    if self.deconv_with_bias:
                        torch.nn.init.constant_(m.bias, 0)
                elif isinstance(m, torch.nn.BatchNorm2d):
                    torch.nn.init.constant_(m.weight, 1)
                    torch.nn.init.constant_(m.bias, 0)
            torch.nn.init.normal_(self.final_layer.weight, std=0.001)
            torch.nn.init.constant_(self.final_layer.bias, 0)
            state = torch.load(pretrained, map_location='cpu')
            logger.info('=> loading pretrained model {}'.format(pretrained))
            self.load_state_dict(state, strict=False)
            if self.cfg.MODEL.EXTRA.USE_PRE_NET:
                with torch.no_grad():
                    self.rgb_preNet[0].weight.data = state['conv1.weight'].clone()
        else:
            logger.info('=> init weights from normal distribution')
            for m in self.modules():
                if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                    torch.nn.init.normal_(m.weight, std=0.001)
                    if isinstance(m, torch.nn.ConvTranspose2d) and self.deconv_with_bias:
                        torch.nn.init.constant_(m.bias, 0)
                elif isinstance(m, torch.nn.BatchNorm2d):
                    torch.nn.init.constant_(m.weight, 1)
                    torch.nn.init.constant_(m.bias, 0)


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
               101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


def get_pose_net(cfg, is_train, **kwargs):
    block_class, layers = resnet_spec[cfg.MODEL.EXTRA.NUM_LAYERS]
    model = PoseResNet(block_class, layers, cfg, **kwargs)
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED)
    return model
