"""train() / validate() - drop-in for reference lib/core/function.py:102-175 and 178-336 (same signatures,
same side effects on model / optimizer / writer_dict / returned perf indicator).

What changed underneath (not in behaviour):
  * heat-maps never cross PCIe: accuracy and decoding read K*(2+1) floats per person produced by the arg-max
    kernel, the flip test is merged on the GPU (flip_back + 1-px shift + average in one kernel), the flipped
    colored condition is re-rendered on the GPU;
  * the per-iteration host syncs of the reference (loss.item() at 141, the full-heat-map D2H + numpy arg-max at
    143-145) are deferred: loss scalars and decoded key points are copied asynchronously and folded into the
    meters one iteration later, so the GPU queue never drains inside the loop.
"""
import logging
import os
import time

import numpy as np
import torch

from .. import ops
from ..utils.transforms import flip_hm, flip_merge_device
from .evaluate import calc_dists, dist_acc
from .inference import get_final_preds

logger = logging.getLogger(__name__)


class AverageMeter(object):
    """Computes and stores the average and current value"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count if self.count != 0 else 0


def _accuracy_from_preds(pred, target, h, w, thr=0.5):
    """evaluate.accuracy() on already decoded arg-max coordinates (numpy [N,K,2])."""
    norm = np.ones((pred.shape[0], 2)) * np.array([h, w]) / 10
    dists = calc_dists(pred, target, norm)
    k = pred.shape[1]
    acc = np.zeros(k + 1)
    avg_acc, cnt = 0, 0
    for i in range(k):
        acc[i + 1] = dist_acc(dists[i], thr)
        if acc[i + 1] >= 0:
            avg_acc += acc[i + 1]
            cnt += 1
    avg_acc = avg_acc / cnt if cnt != 0 else 0
    if cnt != 0:
        acc[0] = avg_acc
    return acc, avg_acc, cnt, pred


class _DeferredStats:
    """Loss scalar + decoded key points of one iteration, travelling to the host asynchronously."""

    def __init__(self, loss, output, target, n):
        with torch.no_grad():
            p_out, _, _ = ops.argmax_decode(output.detach().contiguous())
            p_tgt, _, _ = ops.argmax_decode(target.contiguous())
        self.h, self.w = output.shape[2], output.shape[3]
        self.n = n
        self.loss = torch.empty((), dtype=torch.float32, pin_memory=True)
        self.p_out = torch.empty(p_out.shape, dtype=torch.float32, pin_memory=True)
        self.p_tgt = torch.empty(p_tgt.shape, dtype=torch.float32, pin_memory=True)
        self.loss.copy_(loss.detach(), non_blocking=True)
        self.p_out.copy_(p_out, non_blocking=True)
        self.p_tgt.copy_(p_tgt, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def resolve(self, losses, acc):
        self.event.synchronize()
        losses.update(float(self.loss), self.n)
        _, avg_acc, cnt, pred = _accuracy_from_preds(self.p_out.numpy(), self.p_tgt.numpy(), self.h, self.w)
        acc.update(avg_acc, cnt)
        return pred


def train(config, train_loader, model, criterion, optimizer, epoch, output_dir, tb_log_dir, writer_dict,
          print_prefix=''):
    batch_time = AverageMeter()
    data_time = AverageMeter()
    losses = AverageMeter()
    acc = AverageMeter()
    model.train()

    end = time.time()
    pending = None
    for i, (input, target, target_weight, meta) in enumerate(train_loader):
        data_time.update(time.time() - end)
        if not config.MODEL.CONDITIONAL_TOPDOWN:
            input = input[:, :3]
        input = input.cuda(non_blocking=True)
        outputs = model(input)
        target = target.cuda(non_blocking=True)
        target_weight = target_weight.cuda(non_blocking=True)
        if isinstance(outputs, list):
            loss = criterion(outputs[0], target, target_weight)
            for output in outputs[1:]:
                loss = loss + criterion(output, target, target_weight)
            output = outputs[-1]
        else:
            output = outputs
            loss = criterion(output, target, target_weight)

        optimizer.zero_grad()
        loss.backward()
        optimizer.step()

        # stats of the previous iteration are on the host by now; this iteration's are queued behind the step
        if pending is not None:
            pending.resolve(losses, acc)
        pending = _DeferredStats(loss, output, target, input.size(0))

        batch_time.update(time.time() - end)
        end = time.time()

        if i % config.PRINT_FREQ == 0:
            pred = pending.resolve(losses, acc)
            pending = None
            msg = 'Epoch: [{0}][{1}/{2}]\t' \
                  'Time {batch_time.val:.3f}s ({batch_time.avg:.3f}s)\t' \
                  'Speed {speed:.1f} samples/s\t' \
                  'Data {data_time.val:.3f}s ({data_time.avg:.3f}s)\t' \
                  'Loss {loss.val:.5f} ({loss.avg:.5f})\t' \
                  'Accuracy {acc.val:.3f} ({acc.avg:.3f})'.format(
                      epoch, i, len(train_loader), batch_time=batch_time,
                      speed=input.size(0) / max(batch_time.val, 1e-9), data_time=data_time, loss=losses, acc=acc)
            logger.info(msg)
            if writer_dict:
                writer = writer_dict['writer']
                global_steps = writer_dict['train_global_steps']
                writer.add_scalar('train_loss', losses.val, global_steps)
                writer.add_scalar('train_acc', acc.val, global_steps)
                writer_dict['train_global_steps'] = global_steps + 1
            if epoch % 50 == 0 and config.DEBUG.DEBUG:
                _save_debug_images(config, input, meta, target, pred * 4, output,
                                   '{}_epoch_{}_iter_{}_{}'.format(os.path.join(output_dir, 'train'), epoch, i,
                                                                   print_prefix))
    if pending is not None:
        pending.resolve(losses, acc)
    return


def _save_debug_images(config, input, meta, target, pred, output, prefix, output_dir=None):
    """Debug JPEG grids (reference utils/vis.py:416-472) need cv2 / torchvision, which the target image lacks;
    they are outside the hot path - the hook stays so that callers do not break."""
    logger.debug('debug image dump skipped (%s)', prefix)


def validate(config, val_loader, val_dataset, model, criterion, output_dir, tb_log_dir, writer_dict=None, epoch=-1,
             print_prefix=''):
    batch_time = AverageMeter()
    losses = AverageMeter()
    acc = AverageMeter()
    model.eval()

    num_samples = len(val_dataset)
    all_preds = np.zeros((num_samples, config.MODEL.NUM_JOINTS, 3), dtype=np.float32)
    all_boxes = np.zeros((num_samples, 6 + 1))
    image_path = []
    filenames = []
    imgnums = []
    idx = 0

    with torch.no_grad():
        end = time.time()
        for i, (input, target, target_weight, meta) in enumerate(val_loader):
            if not config.MODEL.CONDITIONAL_TOPDOWN:
                input = input[:, :3]
            input = input.cuda(non_blocking=True)
            outputs = model(input)
            output = outputs[-1] if isinstance(outputs, list) else outputs

            if config.TEST.FLIP_TEST:
                if config.MODEL.CONDITIONAL_TOPDOWN:
                    cond_f = flip_hm(input[:, 3:], val_dataset, meta['cond_joints'], meta['cond_joints_vis'])
                    input_flipped = torch.cat((input[:, :3].flip(3), cond_f.to(input.device)), dim=1)
                else:
                    input_flipped = input.flip(3)
                outputs_flipped = model(input_flipped)
                output_flipped = outputs_flipped[-1] if isinstance(outputs_flipped, list) else outputs_flipped
                output = flip_merge_device(output, output_flipped, val_dataset.flip_pairs,
                                           bool(config.TEST.SHIFT_HEATMAP))

            target = target.cuda(non_blocking=True)
            target_weight = target_weight.cuda(non_blocking=True)
            loss = criterion(output, target, target_weight)
            num_images = input.size(0)
            stats = _DeferredStats(loss, output, target, num_images)

            c = meta['center'].numpy()
            s = meta['scale'].numpy()
            score = meta['score'].numpy()
            annotation_id = meta['annotation_id'].numpy()
            preds, maxvals = get_final_preds(config, output, c, s)
            pred = stats.resolve(losses, acc)

            batch_time.update(time.time() - end)
            end = time.time()

            all_preds[idx:idx + num_images, :, 0:2] = preds[:, :, 0:2]
            all_preds[idx:idx + num_images, :, 2:3] = maxvals
            all_boxes[idx:idx + num_images, 0:2] = c[:, 0:2]
            all_boxes[idx:idx + num_images, 2:4] = s[:, 0:2]
            all_boxes[idx:idx + num_images, 4] = np.prod(s * 200, 1)
            all_boxes[idx:idx + num_images, 5] = score
            all_boxes[idx:idx + num_images, 6] = annotation_id
            image_path.extend(meta['image'])
            idx += num_images

            if (i % config.PRINT_FREQ == 0) or (i == (len(val_loader) - 1)):
                msg = 'Test: [{0}/{1}]\t' \
                      'Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t' \
                      'Loss {loss.val:.6f} ({loss.avg:.6f})\t' \
                      'Accuracy {acc.val:.3f} ({acc.avg:.3f})'.format(
                          i, len(val_loader) - 1, batch_time=batch_time, loss=losses, acc=acc)
                logger.info(msg)
                if config.DEBUG.DEBUG:
                    _save_debug_images(config, input, meta, target, pred * 4, output,
                                       '{}_epoch_{:09d}_iter_{}_{}'.format(os.path.join(output_dir, 'val'), epoch, i,
                                                                           print_prefix), output_dir=output_dir)

        name_values, perf_indicator = val_dataset.evaluate(config, all_preds, output_dir, all_boxes, image_path, epoch,
                                                           filenames, imgnums)
        model_name = config.MODEL.NAME
        if isinstance(name_values, list):
            for name_value in name_values:
                _print_name_value(name_value, model_name)
        else:
            _print_name_value(name_values, model_name)

        if writer_dict:
            writer = writer_dict['writer']
            global_steps = writer_dict['valid_global_steps']
            writer.add_scalar('valid_loss', losses.avg, global_steps)
            writer.add_scalar('valid_acc', acc.avg, global_steps)
            if isinstance(name_values, list):
                for name_value in name_values:
                    writer.add_scalars('valid', dict(name_value), global_steps)
            else:
                writer.add_scalars('valid', dict(name_values), global_steps)
            writer_dict['valid_global_steps'] = global_steps + 1
    return perf_indicator


def _print_name_value(name_value, full_arch_name):
    names = name_value.keys()
    values = name_value.values()
    num_values = len(name_value)
    logger.info('| Arch ' + ' '.join(['| {}'.format(name) for name in names]) + ' |')
    logger.info('|---' * (num_values + 1) + '|')
    if len(full_arch_name) > 15:
        full_arch_name = full_arch_name[:8] + '...'
    logger.info('| ' + full_arch_name + ' ' + ' '.join(['| {:.3f}'.format(value) for value in values]) + ' |')
