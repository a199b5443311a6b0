"""Default configuration tree of the BUCTD path; same keys and default values as reference
lib/config/default.py:17-178 (the contract of every entry point that takes `cfg`)."""
import os

from .node import CfgNode as CN

_C = CN()
_C.OUTPUT_DIR = ''
_C.LOG_DIR = ''
_C.DATA_DIR = ''
_C.GPUS = (0,)
_C.WORKERS = 4
_C.PRINT_FREQ = 20
_C.AUTO_RESUME = False
_C.PIN_MEMORY = True
_C.RANK = 0
_C.EPOCH_EVAL_FREQ = 10

_C.CUDNN = CN({'BENCHMARK': True, 'DETERMINISTIC': False, 'ENABLED': True})

_C.MODEL = CN({
    'NAME': 'pose_hrnet', 'INIT_WEIGHTS': True, 'PRETRAINED': '', 'NUM_JOINTS': 17, 'TAG_PER_JOINT': True,
    'TARGET_TYPE': 'gaussian', 'IMAGE_SIZE': [256, 256], 'HEATMAP_SIZE': [64, 64], 'SIGMA': 2,
    'ATT_MODULES': [False, False, True, True], 'ATT_CHANNEL_ONLY': False, 'ATTENTION_HEADS': 1,
    'SELFATT_MODULES': [False, False, False, False], 'CONDITIONAL_TOPDOWN': False,
    'DIM_MODEL': 96, 'DIM_FEEDFORWARD': 192, 'N_HEAD': 1, 'ENCODER_LAYERS': 6, 'ATTENTION_ACTIVATION': 'relu',
    'POS_EMBEDDING': 'sine',
})
_C.MODEL.EXTRA = CN(new_allowed=True)

_C.LOSS = CN({'USE_OHKM': False, 'TOPK': 8, 'USE_TARGET_WEIGHT': True, 'USE_DIFFERENT_JOINTS_WEIGHT': False})

_C.DATASET = CN({
    'DATASET': 'mpii', 'ROOT': '', 'TRAIN_SET': 'train', 'TRAIN_IMAGE_DIR': '',
    'TRAIN_ANNOTATION_FILE': 'train2017.json', 'TEST_SET': 'valid', 'TEST_IMAGE_DIR': '',
    'TEST_ANNOTATION_FILE': 'val2017.json', 'COND_FILE': 'full_pickle.pickle', 'SYNTHESIS_POSE': False,
    'SWAP_OVERLAP': 0.0, 'DATA_FORMAT': 'jpg', 'HYBRID_JOINTS_TYPE': '', 'SELECT_DATA': False,
    'SYNTHETIC_DATASET': 'synthetic', 'SYNTHETIC_ROOT': '',
    'SYNTHETIC_TRAIN_DATASET': 'synthetic', 'SYNTHETIC_TRAIN_SET': 'train', 'SYNTHETIC_TRAIN_IMAGE_DIR': '',
    'SYNTHETIC_TRAIN_ANNOTATION_FILE': 'train2017.json', 'SYNTHETIC_TRAIN_DATASET_TYPE': 'coco_lambda_syn',
    'SYNTHETIC_TEST_DATASET': 'synthetic', 'SYNTHETIC_TEST_SET': 'valid', 'SYNTHETIC_TEST_IMAGE_DIR': '',
    'SYNTHETIC_TEST_ANNOTATION_FILE': 'val2017.json', 'SYNTHETIC_TEST_DATASET_TYPE': 'coco_lambda_syn',
    'FLIP': True, 'SCALE_FACTOR': 0.25, 'ROT_FACTOR': 30, 'PROB_HALF_BODY': 0.0, 'NUM_JOINTS_HALF_BODY': 8,
    'COLOR_RGB': False, 'BALANCED': False, 'COLORED': False, 'NEW_AUGMENTATION': True, 'BBOX_AUGMENTATION': False,
    'STACKED_CONDITION': False, 'BU_BBOX_MARGIN': 25, 'USE_COND_FILTER': False,
})

_C.TRAIN = CN({
    'LR_FACTOR': 0.1, 'LR_STEP': [90, 110], 'LR': 0.001, 'OPTIMIZER': 'adam', 'MOMENTUM': 0.9, 'WD': 0.0001,
    'NESTEROV': False, 'GAMMA1': 0.99, 'GAMMA2': 0.0, 'BEGIN_EPOCH': 0, 'END_EPOCH': 140, 'RESUME': False,
    'CHECKPOINT': '', 'BATCH_SIZE_PER_GPU': 32, 'SHUFFLE': True, 'USE_BU_BBOX': True,
})

_C.TEST = CN({
    'BATCH_SIZE_PER_GPU': 32, 'FLIP_TEST': False, 'POST_PROCESS': False, 'SHIFT_HEATMAP': False,
    'USE_GT_BBOX': False, 'USE_BU_BBOX': True, 'IMAGE_THRE': 0.1, 'NMS_THRE': 0.6, 'SOFT_NMS': False,
    'OKS_THRE': 0.5, 'IN_VIS_THRE': 0.0, 'COCO_BBOX_FILE': '', 'BBOX_THRE': 1.0, 'MODEL_FILE': '',
    'BBOX_FRACTION': 1.0, 'DECAY_THRE': 0.5, 'SCALE_THRE': 1.25,
})

_C.DEBUG = CN({'DEBUG': False, 'SAVE_BATCH_IMAGES_GT': False, 'SAVE_BATCH_IMAGES_PRED': False,
               'SAVE_HEATMAPS_GT': False, 'SAVE_HEATMAPS_PRED': False})

_C.OUTPUT_JSON = None


def update_config(cfg, args):
    """reference lib/config/default.py:180-207."""
    cfg.defrost()
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(args.opts)
    if getattr(args, 'modelDir', None):
        cfg.OUTPUT_DIR = args.modelDir
    if getattr(args, 'logDir', None):
        cfg.LOG_DIR = args.logDir
    if getattr(args, 'dataDir', None):
        cfg.DATA_DIR = args.dataDir
    cfg.DATASET.ROOT = os.path.join(cfg.DATA_DIR, cfg.DATASET.ROOT)
    cfg.MODEL.PRETRAINED = os.path.join(cfg.DATA_DIR, cfg.MODEL.PRETRAINED)
    if cfg.TEST.MODEL_FILE:
        cfg.TEST.MODEL_FILE = os.path.join(cfg.DATA_DIR, cfg.TEST.MODEL_FILE)
    cfg.freeze()


def hrnet_extra(width, use_pre_net=False, use_attention=False, modules=(1, 4, 3)):
    """MODEL.EXTRA of experiments/*/hrnet/w{32,48}_384x288_adam_lr1e-3.yaml:39-91 for a given width."""
    def stage(m, b):
        return {'NUM_MODULES': m, 'NUM_BRANCHES': b, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [4] * b,
                'NUM_CHANNELS': [width * 2 ** i for i in range(b)], 'FUSE_METHOD': 'SUM'}
    return CN({
        'PRETRAINED_LAYERS': ['conv1', 'bn1', 'conv2', 'bn2', 'layer1', 'transition1', 'stage2', 'transition2',
                              'stage3', 'transition3', 'stage4'],
        'FINAL_CONV_KERNEL': 1, 'STAGE2': stage(modules[0], 2), 'STAGE3': stage(modules[1], 3),
        'STAGE4': stage(modules[2], 4), 'USE_PRE_NET': use_pre_net, 'USE_ATTENTION': use_attention,
    }, new_allowed=True)
