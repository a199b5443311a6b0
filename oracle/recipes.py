"""TEST INFRASTRUCTURE - seeded model / input recipes shared by oracle/make_golden.py (which runs
them through the imported reference) and the tests (which rebuild them on the GPU box, where the
reference does not exist).  Everything is generated from torch CPU generators with fixed seeds,
so a recipe is reproducible bit for bit with the same torch build.
"""
import numpy as np
import torch
import torch.nn as nn

from . import cfg as ocfg
from . import core as ocore
from . import models as omodels


def randomize(model, seed):
    """Make every tensor O(1) and asymmetric so that a 1e-3 tolerance means something
    (std=0.001 inits would make all outputs ~0)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in model.named_modules():
            if name.endswith("cnn_cond") or name.endswith("trans_cond"):
                # conditions live in [0,255]; a trained net has adapted to that scale, a random one
                # must be told, or every softmax saturates into a one-hot and parity becomes chaotic
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.4 / (255.0 * m.weight[0].numel() ** 0.5)))
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                continue
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            elif isinstance(m, nn.Linear):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (m.in_features ** -0.5))
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                fan_in = m.weight[0].numel() if isinstance(m, nn.Conv2d) else m.weight[:, 0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.4 / fan_in ** 0.5))
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return model


def make_inputs(cfg, batch, seed, cond_channels):
    """RGB ~ N(0,1); condition = rendered key-point blobs in [0,255] (SURVEY 8d synthetic inputs)."""
    g = torch.Generator().manual_seed(seed)
    w, h = cfg.MODEL.IMAGE_SIZE
    k = cfg.MODEL.NUM_JOINTS
    rgb = torch.randn(batch, 3, h, w, generator=g)
    joints = torch.rand(batch, k, 2, generator=g) * torch.tensor([w - 1.0, h - 1.0])
    if cond_channels == 0:
        return rgb, joints
    conds = []
    colors = (ocore.CROWDPOSE_KPT_COLORS * 2)[:k]
    for b in range(batch):
        if cond_channels == 3:
            c = ocore.get_condition_image_colored(joints[b].numpy(), (h, w, 3), colors).transpose(2, 0, 1)
        elif cond_channels == 1:
            c = ocore.get_condition_image(joints[b].numpy(), (h, w)).astype(np.float64)  # mono x3, int-truncated
        else:  # stacked: one blurred channel per joint (JointsDataset.py:471-498)
            chans = []
            for j in range(k):
                z = np.zeros((h, w))
                kp = joints[b, j].numpy().astype(int)
                if 0 < kp[0] < w and 0 < kp[1] < h:
                    z[kp[1] - 1][kp[0] - 1] = 255
                chans.append(ocore.generate_heatmap(z))
            c = np.stack(chans, 0)
        conds.append(torch.from_numpy(np.ascontiguousarray(c)).float())
    x = torch.cat([rgb, torch.stack(conds)], 1)
    return x, joints


def make_targets(cfg, joints, seed):
    g = torch.Generator().manual_seed(seed)
    k = cfg.MODEL.NUM_JOINTS
    b = joints.shape[0]
    vis = (torch.rand(b, k, generator=g) < 0.8).float()
    tgts, wts = [], []
    for i in range(b):
        j3 = torch.cat([joints[i], torch.zeros(k, 1)], 1).numpy()
        v3 = vis[i].view(k, 1).repeat(1, 3).numpy()
        t, w = ocore.generate_target(j3, v3, k, cfg.MODEL.HEATMAP_SIZE, cfg.MODEL.IMAGE_SIZE, cfg.MODEL.SIGMA)
        tgts.append(torch.from_numpy(t))
        wts.append(torch.from_numpy(w))
    return torch.stack(tgts), torch.stack(wts)


def _small_stage_modules(c, mods):
    for s, m in zip(("STAGE2", "STAGE3", "STAGE4"), mods):
        c.MODEL.EXTRA[s]["NUM_MODULES"] = m
    return c


CASES = {}


def case(name):
    def deco(fn):
        CASES[name] = fn
        return fn
    return deco


@case("prenet_w16_96x64")
def _prenet_small():
    c = ocfg.hrnet_cfg(16, 17, (64, 96), "pose_hrnet", use_pre_net=True, stage_modules=(1, 2, 2))
    return c, 3, 3


@case("coam_w16_96x64_colored")
def _coam_small():
    c = ocfg.hrnet_cfg(16, 14, (64, 96), "pose_hrnet_coam", use_attention=True, stage_modules=(1, 2, 2))
    return c, 3, 3


@case("coam_w16_96x64_mono_default_att")
def _coam_mono():
    # default ATT_MODULES [F,F,T,T] (config/default.py:48), mono condition (d_cond = 1)
    c = ocfg.hrnet_cfg(16, 14, (64, 96), "pose_hrnet_coam", use_attention=True, att_modules=(False, False, True, True),
                       colored=False, stage_modules=(1, 1, 1))
    return c, 2, 1


@case("coam_w16_96x64_stacked_2heads")
def _coam_stacked():
    c = ocfg.hrnet_cfg(16, 14, (64, 96), "pose_hrnet_coam", use_attention=True, att_modules=(True, True, False, False),
                       colored=False, stacked=True, heads=2, stage_modules=(1, 1, 1))
    return c, 2, 14


@case("coam_w16_96x64_channel_only")
def _coam_channel_only():
    # MODEL.ATT_CHANNEL_ONLY: DAModule returns input * c_out (pose_hrnet_coam.py:716-717)
    c = ocfg.hrnet_cfg(16, 14, (64, 96), "pose_hrnet_coam", use_attention=True, att_modules=(False, True, True, False),
                       channel_only=True, stage_modules=(1, 1, 1))
    return c, 2, 3


@case("coam_w16_96x64_selfatt")
def _coam_selfatt():
    # MODEL.SELFATT_MODULES [F,T,F,F] with ATT_MODULES all off (the constructor asserts they never coincide,
    # pose_hrnet_coam.py:461-462): SelfAttentionModule / SelfDAModule are CONSTRUCTED (their parameters are part of the
    # state_dict) but the reference's forward only ever calls stageN_att under att_config[N] (pose_hrnet_coam.py:521-562), so
    # the variant is a plain HRNet on the RGB channels carrying unused parameters - which is what this case pins
    c = ocfg.hrnet_cfg(16, 14, (64, 96), "pose_hrnet_coam", use_attention=True, att_modules=(False, False, False, False),
                       selfatt=(False, True, False, False), stage_modules=(1, 1, 1))
    return c, 2, 3


@case("transpose_w16_96x64")
def _transpose_small():
    c = ocfg.hrnet_cfg(16, 17, (64, 96), "transpose_h", use_attention=True, stage_modules=(1, 2, 2))
    c.MODEL.DIM_MODEL = 32
    c.MODEL.DIM_FEEDFORWARD = 64
    c.MODEL.ENCODER_LAYERS = 2
    return c, 2, 3


@case("resnet18_96x64")
def _resnet_small():
    c = ocfg.resnet_cfg(18, 17, (64, 96))
    return c, 2, 0


@case("coam_w48_384x288")
def _coam_full():
    c = ocfg.hrnet_cfg(48, 14, (288, 384), "pose_hrnet_coam", use_attention=True)
    return c, 1, 3


@case("prenet_w32_256x192")
def _prenet_w32():
    c = ocfg.hrnet_cfg(32, 17, (192, 256), "pose_hrnet", use_pre_net=True)
    return c, 1, 3


@case("resnet50_256x192")
def _resnet50_full():
    # BASELINE config C1: pose_resnet50 256x192 COCO-17kpt, batch 4
    c = ocfg.resnet_cfg(50, 17, (192, 256))
    return c, 4, 0


@case("prenet_w48_384x288")
def _prenet_w48():
    # BASELINE config C3: BUCTD-preNet HRNet-W48 384x288 COCO
    c = ocfg.hrnet_cfg(48, 17, (288, 384), "pose_hrnet", use_pre_net=True)
    return c, 1, 3


@case("transpose_a6_256x192")
def _transpose_full():
    # BASELINE config C5: BUCTD-TransPose-H-A6 256x192 (W48 trunk, d_model 96 + 16, 6 encoder layers, T = 3072)
    c = ocfg.hrnet_cfg(48, 17, (192, 256), "transpose_h", use_attention=True)
    return c, 1, 3


# The randomised networks emit heat-maps of |y| ~ 10..100.  north_star's parity bar is an ABSOLUTE 1e-3 on heat-maps,
# which are of unit scale in a trained network (Gaussian targets peak at 1).  Every recipe therefore scales its
# final_layer (weight and bias) by 2^-k, k fixed per recipe below so that max|y| lands in [0.5, 1] for the recipe
# input: a power of two, so the scaling itself is exact and the recipe stays a pure function of its seed.
FINAL_SCALE_LOG2 = {
    "prenet_w16_96x64": 6, "coam_w16_96x64_colored": 6, "coam_w16_96x64_mono_default_att": 7,
    "coam_w16_96x64_stacked_2heads": 5, "coam_w16_96x64_channel_only": 9, "coam_w16_96x64_selfatt": 5, "transpose_w16_96x64": 2, "resnet18_96x64": 3, "coam_w48_384x288": 6,
    "prenet_w32_256x192": 7, "resnet50_256x192": 3, "prenet_w48_384x288": 7, "transpose_a6_256x192": 3,
}


# recipe seeds other than the default: the channel-only net at seed 1234 has a ReLU pre-activation within fp32 round-off
# of zero in its train step (the fp32 CPU oracle itself sits 4e-3 from its fp64 evaluation there, against 2e-5 at seed 1)
# (round 3) the mono / default-attention net at seed 1234 has one BasicBlock pre-activation of its 3x2-pixel stage-4 branch
# within round-off of zero: which side of the ReLU it lands on depends on the summation order of the convolutions in front
# of it, and with 12 samples per BatchNorm channel there one flipped unit moves most gradients by 5-10 % (scratch/
# diag_chain_internal.py, scratch/seed_scan.py: at seed 1 the fp32 CPU oracle and every HIP math variant sit 2-3e-5 from fp64)
# The TransPose trunk is the worst conditioned of the small nets: per seed, one of {fp32 CPU oracle, HIP with / without the
# gathered bf16x6 convolutions} sits 1e-3..2e-2 from fp64 (seed: cpu32 median / HIP median - 1234: 1e-5 / 5e-3, 1: 4e-3 / 4e-5,
# 2: 2e-5 / 2e-3, 4: 2e-5 / 3e-3); seed 5 has no pre-activation near zero for any of them (2e-5 / 1e-4).
# The original seed 1234 of the last two stays a test case under a flip-aware metric (tests/test_gpu_models.py:
# test_train_step_at_the_original_seed_differs_by_a_relu_flip_only): the same weights meet the bars on perturbed inputs.
SEEDS = {"coam_w16_96x64_channel_only": 1, "coam_w16_96x64_mono_default_att": 1, "transpose_w16_96x64": 5}


def build(name, seed=None, final_scale_log2=None):
    """-> cfg, oracle model (eval mode, randomised, unit-scale heat-maps), input x [B,3+Cc,H,W], joints [B,K,2]."""
    if seed is None:
        seed = SEEDS.get(name, 1234)
    c, batch, cond_channels = CASES[name]()
    torch.manual_seed(seed)
    model = omodels.get_pose_net(c, is_train=False)
    randomize(model, seed + 1)
    x, joints = make_inputs(c, batch, seed + 2, cond_channels)
    calibrate_bn(model, x)
    k = FINAL_SCALE_LOG2[name] if final_scale_log2 is None else final_scale_log2
    with torch.no_grad():
        model.final_layer.weight.mul_(2.0 ** -k)
        if model.final_layer.bias is not None:
            model.final_layer.bias.mul_(2.0 ** -k)
    model.eval()
    return c, model, x, joints


def calibrate_bn(model, x):
    """One train-mode pass with momentum 1 so every BN's running statistics equal the batch
    statistics of the recipe input: eval-mode activations then stay O(1) through all ~300
    layers, like in a trained network (random running stats let them grow to 1e8)."""
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    saved = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    p_saved = []
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            p_saved.append((m, m.p))
            m.p = 0.0
        elif isinstance(m, nn.MultiheadAttention):
            p_saved.append((m, m.dropout))
            m.dropout = 0.0
    with torch.no_grad():
        model(x)
    for m, p in p_saved:
        if isinstance(m, nn.Dropout):
            m.p = p
        else:
            m.dropout = p
    for m, mom in zip(bns, saved):
        m.momentum = mom
        m.num_batches_tracked.zero_()


def set_dropout(model, p):
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = p
        if isinstance(m, nn.MultiheadAttention) or type(m).__name__ == "MultiheadAttention":
            m.dropout = p  # torch's module and the engine's parameter container both keep p in .dropout
