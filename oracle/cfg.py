"""TEST INFRASTRUCTURE - not part of the product path.

Minimal attribute-dict configuration for the oracle models (the reference uses yacs, which is
not installed here).  Keys and defaults follow reference lib/config/default.py:17-178 and the
experiment YAMLs (experiments/crowdpose/hrnet/w48_384x288_adam_lr1e-3.yaml:47-91).
"""
import copy


class Cfg(dict):
    """dict with attribute access, nested."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return Cfg({k: Cfg.wrap(v) for k, v in obj.items()})
        return obj


def _stage(modules, branches, width, block="BASIC"):
    return {"NUM_MODULES": modules, "NUM_BRANCHES": branches, "BLOCK": block, "NUM_BLOCKS": [4] * branches,
            "NUM_CHANNELS": [width * 2 ** i for i in range(branches)], "FUSE_METHOD": "SUM"}


def hrnet_cfg(width=48, num_joints=14, image_size=(288, 384), name="pose_hrnet", use_pre_net=False,
              use_attention=False, att_modules=(False, True, False, False), colored=True, stacked=False, heads=1,
              channel_only=False, selfatt=(False, False, False, False), stage_modules=(1, 4, 3)):
    """image_size is (W, H) like MODEL.IMAGE_SIZE."""
    w, h = image_size
    return Cfg.wrap({
        "MODEL": {
            "NAME": name, "INIT_WEIGHTS": True, "PRETRAINED": "", "NUM_JOINTS": num_joints,
            "IMAGE_SIZE": [w, h], "HEATMAP_SIZE": [w // 4, h // 4], "SIGMA": 3 if h >= 384 else 2,
            "TARGET_TYPE": "gaussian",
            "ATT_MODULES": list(att_modules), "ATT_CHANNEL_ONLY": channel_only, "ATTENTION_HEADS": heads,
            "SELFATT_MODULES": list(selfatt), "CONDITIONAL_TOPDOWN": use_pre_net or use_attention,
            "DIM_MODEL": 96, "DIM_FEEDFORWARD": 192, "N_HEAD": 1, "ENCODER_LAYERS": 6,
            "ATTENTION_ACTIVATION": "relu", "POS_EMBEDDING": "sine",
            "EXTRA": {
                "PRETRAINED_LAYERS": ["conv1", "bn1", "conv2", "bn2", "layer1", "transition1", "stage2",
                                      "transition2", "stage3", "transition3", "stage4"],
                "FINAL_CONV_KERNEL": 1,
                "STAGE2": _stage(stage_modules[0], 2, width),
                "STAGE3": _stage(stage_modules[1], 3, width),
                "STAGE4": _stage(stage_modules[2], 4, width),
                "USE_PRE_NET": use_pre_net, "USE_ATTENTION": use_attention,
            },
        },
        "DATASET": {"COLORED": colored, "STACKED_CONDITION": stacked},
        "LOSS": {"USE_TARGET_WEIGHT": True},
        "TRAIN": {"OPTIMIZER": "adam", "LR": 0.001},
        "TEST": {"FLIP_TEST": False, "POST_PROCESS": True, "SHIFT_HEATMAP": True},
    })


def resnet_cfg(num_layers=50, num_joints=17, image_size=(192, 256), use_pre_net=False):
    w, h = image_size
    c = hrnet_cfg(32, num_joints, image_size, name="pose_resnet", use_pre_net=use_pre_net)
    c.MODEL.EXTRA = Cfg.wrap({"NUM_LAYERS": num_layers, "DECONV_WITH_BIAS": False, "NUM_DECONV_LAYERS": 3,
                              "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_DECONV_KERNELS": [4, 4, 4],
                              "FINAL_CONV_KERNEL": 1, "USE_PRE_NET": use_pre_net, "USE_ATTENTION": False})
    return c


def clone(cfg):
    return Cfg.wrap(copy.deepcopy(dict(cfg)))
