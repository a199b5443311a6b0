"""Boundary helpers of the training driver - drop-in for the pieces of reference lib/utils/utils.py the path
uses: get_optimizer (258-274), save_checkpoint (303-308), create_logger (220-255, without the dataset-specific
directory logic that needs the external data tree)."""
import logging
import os
import time

import torch

from ..engine import get_optimizer  # noqa: F401  (same name / signature as the reference helper)


def save_checkpoint(states, is_best, output_dir, filename='checkpoint.pth'):
    torch.save(states, os.path.join(output_dir, filename))
    if is_best and 'state_dict' in states:
        torch.save(states['best_state_dict'], os.path.join(output_dir, 'model_best.pth'))


def create_logger(cfg, cfg_name, phase='train'):
    root = cfg.OUTPUT_DIR or 'output'
    name = os.path.basename(cfg_name).split('.')[0]
    final_output_dir = os.path.join(root, cfg.DATASET.DATASET, cfg.MODEL.NAME, name)
    os.makedirs(final_output_dir, exist_ok=True)
    time_str = time.strftime('%Y-%m-%d-%H-%M')
    log_file = os.path.join(final_output_dir, '{}_{}_{}.log'.format(name, time_str, phase))
    logging.basicConfig(filename=log_file, format='%(asctime)-15s %(message)s')
    log = logging.getLogger()
    log.setLevel(logging.INFO)
    logging.getLogger('').addHandler(logging.StreamHandler())
    tb_dir = os.path.join(cfg.LOG_DIR or 'log', cfg.DATASET.DATASET, cfg.MODEL.NAME, name + '_' + time_str)
    os.makedirs(tb_dir, exist_ok=True)
    return log, str(final_output_dir), str(tb_dir)
