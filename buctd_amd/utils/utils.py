"""Boundary helpers of the training driver - drop-in for the pieces of reference lib/utils/utils.py the path
uses: get_optimizer (258-274), save_checkpoint (303-308), create_logger (220-255, without the dataset-specific
directory logic that needs the external data tree)."""
import logging
import os
import time

import torch

from ..engine import get_optimizer  # noqa: F401  (same name / signature as the reference helper)


def save_checkpoint(states, is_best, output_dir, filename='checkpoint.pth'):
    """Writes the checkpoint dict; a best-so-far run additionally leaves its bare best_state_dict as
    model_best.pth (what tools/test.py loads).  Same files as the reference helper."""
    target = os.path.join(output_dir, filename)
    torch.save(states, target)
    best = states.get('best_state_dict') if (is_best and 'state_dict' in states) else None
    if best is not None:
        torch.save(best, os.path.join(output_dir, 'model_best.pth'))
    return target


def create_logger(cfg, cfg_name, phase='train'):
    """<OUTPUT_DIR>/<dataset>/<model>/<cfg name>/ for logs and checkpoints, <LOG_DIR>/.../<cfg name>_<time>/ for
    tensorboard; returns (logger, output dir, tensorboard dir)."""
    stamp = time.strftime('%Y-%m-%d-%H-%M')
    stem = os.path.splitext(os.path.basename(cfg_name))[0]
    leaf = os.path.join(cfg.DATASET.DATASET, cfg.MODEL.NAME)
    out_dir = os.path.join(cfg.OUTPUT_DIR or 'output', leaf, stem)
    tb_dir = os.path.join(cfg.LOG_DIR or 'log', leaf, stem + '_' + stamp)
    for d in (out_dir, tb_dir):
        os.makedirs(d, exist_ok=True)
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    file_handler = logging.FileHandler(os.path.join(out_dir, f'{stem}_{stamp}_{phase}.log'))
    file_handler.setFormatter(logging.Formatter('%(asctime)-15s %(message)s'))
    root.addHandler(file_handler)
    root.addHandler(logging.StreamHandler())
    return root, str(out_dir), str(tb_dir)
