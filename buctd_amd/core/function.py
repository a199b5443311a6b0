"""train() / validate() - drop-in for reference lib/core/function.py:102-175 and 178-336 (same signatures,
same side effects on model / optimizer / writer_dict, same log lines, same returned perf indicator).

What changed underneath (not in behaviour):
  * heat-maps never cross PCIe: accuracy and decoding read K*(2+1) floats per person produced by the arg-max
    kernel (the quarter-pixel refinement of get_final_preds included), the flip test is merged on the GPU
    (flip_back + 1-px shift + average in one kernel), the flipped colored condition is re-rendered on the GPU;
  * the per-iteration host syncs of the reference (loss.item() at 141, the full-heat-map D2H + numpy arg-max at
    143-145) are deferred: loss scalars and decoded key points are copied asynchronously and folded into the
    meters one iteration later, so the GPU queue never drains inside the loop;
  * under a one-process-per-GPU launch validate() shards the batches of val_loader over the ranks and
    all-gathers all_preds / all_boxes / image paths before rank 0 calls val_dataset.evaluate.
"""
import logging
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..utils.transforms import flip_hm, flip_merge_device
from .evaluate import pck_from_coords
from .inference import DeferredFinalPreds, get_final_preds

logger = logging.getLogger(__name__)


class AverageMeter(object):
    """Running mean with the reference's public fields (val, avg, sum, count)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count = self.count + n
        self.avg = self.sum / self.count if self.count else 0


_ZERO_COPY = ops._SW["STATS_ZERO_COPY"] == "1"


class _DeferredStats:
    """Loss scalar + decoded key points of one iteration, travelling to the host asynchronously."""

    def __init__(self, loss, output, target, n):
        self.h, self.w = output.shape[2], output.shape[3]
        self.n = n
        shape = (output.shape[0], output.shape[1], 2)
        self.loss = torch.empty((), dtype=torch.float32, pin_memory=True)
        self.p_out = torch.empty(shape, dtype=torch.float32, pin_memory=True)
        self.p_tgt = torch.empty(shape, dtype=torch.float32, pin_memory=True)
        with torch.no_grad():
            if _ZERO_COPY:
                # the decode kernels write the 3.5 KB of coordinates (and a one-float kernel the loss) straight into pinned
                # host memory: three copy-engine transfers less per step, and the next step's first kernel does not queue
                # behind a copy (the gap in front of nchw_to_nhwc in profiles/r06_critical_path.txt)
                ops.argmax_decode(output.detach().contiguous(), preds_out=self.p_out)
                ops.argmax_decode(target.contiguous(), preds_out=self.p_tgt)
                ops.scalar_to_host(loss.detach(), self.loss)
            else:
                p_out, _, _ = ops.argmax_decode(output.detach().contiguous())
                p_tgt, _, _ = ops.argmax_decode(target.contiguous())
                self.loss.copy_(loss.detach(), non_blocking=True)
                self.p_out.copy_(p_out, non_blocking=True)
                self.p_tgt.copy_(p_tgt, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def resolve(self, losses, acc):
        self.event.synchronize()
        losses.update(float(self.loss), self.n)
        _, avg_acc, cnt, pred = pck_from_coords(self.p_out.numpy(), self.p_tgt.numpy(), self.h, self.w)
        acc.update(avg_acc, cnt)
        return pred


def _forward_with_loss(model, criterion, input, target, target_weight):
    """A model may return one heat-map tensor or a list of them (reference function.py:127-133): the loss is summed
    over the list, the last entry is the prediction."""
    outputs = model(input)
    heads = outputs if isinstance(outputs, list) else [outputs]
    loss = None
    for head in heads:
        term = criterion(head, target, target_weight)
        loss = term if loss is None else loss + term
    return heads[-1], loss


def _train_line(epoch, i, total, batch_time, data_time, losses, acc, batch):
    speed = batch / max(batch_time.val, 1e-9)
    return (f'Epoch: [{epoch}][{i}/{total}]\t'
            f'Time {batch_time.val:.3f}s ({batch_time.avg:.3f}s)\t'
            f'Speed {speed:.1f} samples/s\t'
            f'Data {data_time.val:.3f}s ({data_time.avg:.3f}s)\t'
            f'Loss {losses.val:.5f} ({losses.avg:.5f})\t'
            f'Accuracy {acc.val:.3f} ({acc.avg:.3f})')


PREFETCH_TO_DEVICE = True      # train(): host-to-device copies of the next batch beside the current step (DevicePrefetch)


class DevicePrefetch:
    """Iterates a loader of (input, target, target_weight, meta) host batches and yields them with the three tensors on the
    device.  The reference loop copies a batch at the top of its iteration (function.py:118-125: .cuda(non_blocking=True)) -
    on the compute stream, i.e. in front of the forward pass (85 MB per CoAM-W48 batch of 32: ~2 ms of every step).  Here
    the copies of batch i + 1 are enqueued on the engine's weight-gradient stream BEFORE step i is enqueued: that stream is
    idle during the forward pass, the copies run on the DMA engines beside it, and the main stream only waits for an event
    that has long passed.  No new HIP stream (a fifth one costs 25-32 %: DESIGN.md 6).  Pinned host batches
    (DataLoader(pin_memory=True), as the reference builds its loaders) make the copies asynchronous; pageable ones still work."""

    def __init__(self, loader, conditional=True, device=None):
        self.loader, self.conditional = loader, conditional
        self.device = device if device is not None else torch.device('cuda', torch.cuda.current_device())

    def __len__(self):
        return len(self.loader)

    def _enqueue(self, batch):
        input, target, target_weight, meta = batch
        if not self.conditional:
            input = input[:, :3]
        side = ops.copy_stream(self.device)
        if side is None:
            return (input.cuda(non_blocking=True), target.cuda(non_blocking=True), target_weight.cuda(non_blocking=True)), meta, None
        with torch.cuda.stream(side):
            dev = tuple(t.to(self.device, non_blocking=True) for t in (input, target, target_weight))
            ev = torch.cuda.Event()
            ev.record(side)
        return dev, meta, ev

    def _ready(self, item):
        dev, meta, ev = item
        if ev is not None:
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)
            for t in dev:
                t.record_stream(main)
        return dev[0], dev[1], dev[2], meta

    def __iter__(self):
        it = iter(self.loader)
        try:
            cur = self._enqueue(next(it))
        except StopIteration:
            return
        for host in it:
            nxt = self._enqueue(host)          # batch i + 1 goes on its way before step i is enqueued
            yield self._ready(cur)
            cur = nxt
        yield self._ready(cur)


def train(config, train_loader, model, criterion, optimizer, epoch, output_dir, tb_log_dir, writer_dict,
          print_prefix='', step_graph=None):
    """reference lib/core/function.py:102-175, same positional signature.  step_graph (an engine.StepGraph built on the same
    model / criterion / optimizer; not in the reference): the device work of an iteration is replayed from a hipGraph instead
    of being enqueued kernel by kernel - for host-bound configurations (HRNet-W32 at 256x192)."""
    batch_time, data_time, losses, acc = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    model.train()
    conditional = bool(config.MODEL.CONDITIONAL_TOPDOWN)
    tick = time.time()
    pending = None
    batches = DevicePrefetch(train_loader, conditional) if PREFETCH_TO_DEVICE and torch.cuda.is_available() else train_loader
    for i, (input, target, target_weight, meta) in enumerate(batches):
        data_time.update(time.time() - tick)
        if not input.is_cuda:
            input = (input if conditional else input[:, :3]).cuda(non_blocking=True)
            target = target.cuda(non_blocking=True)
            target_weight = target_weight.cuda(non_blocking=True)
        if step_graph is not None:
            output, loss = step_graph(input, target, target_weight)
        else:
            output, loss = _forward_with_loss(model, criterion, input, target, target_weight)

            optimizer.zero_grad()
            loss.backward()
            optimizer.step()

        # stats of the previous iteration are on the host by now; this iteration's are queued behind the step
        if pending is not None:
            pending.resolve(losses, acc)
        pending = _DeferredStats(loss, output, target, input.size(0))

        now = time.time()
        batch_time.update(now - tick)
        tick = now

        if i % config.PRINT_FREQ != 0:
            continue
        pred = pending.resolve(losses, acc)
        pending = None
        logger.info(_train_line(epoch, i, len(train_loader), batch_time, data_time, losses, acc, input.size(0)))
        if writer_dict:
            step = writer_dict['train_global_steps']
            writer_dict['writer'].add_scalar('train_loss', losses.val, step)
            writer_dict['writer'].add_scalar('train_acc', acc.val, step)
            writer_dict['train_global_steps'] = step + 1
        if config.DEBUG.DEBUG and epoch % 50 == 0:
            tag = '{}_epoch_{}_iter_{}_{}'.format(os.path.join(output_dir, 'train'), epoch, i, print_prefix)
            _save_debug_images(config, input, meta, target, pred * 4, output, tag)
    if pending is not None:
        pending.resolve(losses, acc)


def _save_debug_images(config, input, meta, target, pred, output, prefix, output_dir=None):
    """Debug JPEG grids (reference utils/vis.py:416-472) need cv2 / torchvision, which the target image lacks;
    they are outside the hot path - the hook stays so that callers do not break."""
    logger.debug('debug image dump skipped (%s)', prefix)


# ------------------------------------------------------------------------------------------ validate ----
def _dist_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_validation_shards(all_preds, all_boxes, image_path, filled):
    """Merge the per-rank result tables of a sharded validate(): every rank filled the rows of the batches it
    processed (`filled` [num_samples] bool, disjoint across ranks) and the matching entries of image_path
    (a list with None elsewhere).  After the call every rank holds the complete tables.  The tables are tiny
    (K*3 + 7 floats per person), so one all-reduce of the masked tables is the whole exchange."""
    rank, world = _dist_world()
    if world == 1:
        return all_preds, all_boxes, image_path
    backend = dist.get_backend()
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    mask = np.asarray(filled, dtype=bool)
    packed = np.concatenate([np.where(mask[:, None], all_preds.reshape(len(mask), -1), 0).astype(np.float64),
                             np.where(mask[:, None], all_boxes, 0).astype(np.float64),
                             mask[:, None].astype(np.float64)], axis=1)
    t = torch.from_numpy(packed).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    merged = t.cpu().numpy()
    owners = merged[:, -1]
    if not np.all(owners == 1):
        raise RuntimeError('validate(): %d samples were processed by no rank or by several' % int((owners != 1).sum()))
    kcols = all_preds.shape[1] * all_preds.shape[2]
    preds = merged[:, :kcols].reshape(all_preds.shape).astype(all_preds.dtype)
    boxes = merged[:, kcols:kcols + all_boxes.shape[1]].astype(all_boxes.dtype)
    paths = [None] * world
    dist.all_gather_object(paths, [(i, p) for i, p in enumerate(image_path) if p is not None])
    full = list(image_path)
    for part in paths:
        for i, p in part:
            full[i] = p
    return preds, boxes, full


class _RankBatchSampler:
    """The batches i = rank, rank + world, ... of a sequential, unshuffled loader: index lists only, so the DataLoader workers
    of a rank decode, augment and collate its own share of the validation set and nothing else."""

    def __init__(self, num_samples, batch_size, drop_last, rank, world):
        n = num_samples // batch_size if drop_last else (num_samples + batch_size - 1) // batch_size
        self.plan = [(i, i * batch_size, min(num_samples, (i + 1) * batch_size)) for i in range(n) if i % world == rank]

    def __iter__(self):
        for _, a, b in self.plan:
            yield list(range(a, b))

    def __len__(self):
        return len(self.plan)


def _rank_batches(val_loader, rank, world):
    """Yields (batch index, first row, batch) for the batches this rank owns.  A torch DataLoader over a sequential sampler
    is re-created with a per-rank batch sampler (the data path is sharded, not just the model compute); any other iterable
    of batches is walked in full and foreign batches are skipped (their rows are still counted)."""
    from torch.utils.data import DataLoader, SequentialSampler
    if world > 1 and isinstance(val_loader, DataLoader) and isinstance(getattr(val_loader, 'sampler', None), SequentialSampler) \
            and val_loader.batch_size is not None:
        bs = _RankBatchSampler(len(val_loader.dataset), val_loader.batch_size, val_loader.drop_last, rank, world)
        kw = dict(num_workers=val_loader.num_workers, collate_fn=val_loader.collate_fn, pin_memory=val_loader.pin_memory,
                  worker_init_fn=val_loader.worker_init_fn, timeout=val_loader.timeout)
        if val_loader.num_workers > 0:
            kw.update(prefetch_factor=val_loader.prefetch_factor, persistent_workers=False)
        mine = DataLoader(val_loader.dataset, batch_sampler=bs, **kw)
        for (i, a, _), batch in zip(bs.plan, mine):
            yield i, a, batch
        return
    cursor = 0
    for i, batch in enumerate(val_loader):
        count = batch[0].size(0)
        if world == 1 or i % world == rank:
            yield i, cursor, batch
        cursor += count


def _prefetched(rank_batches, conditional):
    """(i, row0, host batch) -> (i, row0, device batch): batch i + 1 is copied beside the forward of batch i (DevicePrefetch)."""
    if not (PREFETCH_TO_DEVICE and torch.cuda.is_available()):
        yield from rank_batches
        return
    pf = DevicePrefetch(None, conditional)
    prev = None
    for i, row0, batch in rank_batches:
        item = (i, row0, pf._enqueue(batch))
        if prev is not None:
            yield prev[0], prev[1], pf._ready(prev[2])
        prev = item
    if prev is not None:
        yield prev[0], prev[1], pf._ready(prev[2])


class _Enqueued:
    """The result of a forward that is already enqueued on the current stream (a network without submit())."""

    def __init__(self, out):
        self.out = out

    def result(self):
        return self.out


def _submit(model, x):
    """-> handle with .result(): through the network's own submit() (engine.ForwardGraph: graph lanes) where it has one."""
    submit = getattr(model, 'submit', None)
    return submit(x) if callable(submit) else _Enqueued(model(x))


def _mirrored_input(config, val_dataset, input, meta):
    """The input of the flip test (reference function.py:213-225): the mirrored crop, with the condition re-rendered from
    the mirrored condition key points."""
    if config.MODEL.CONDITIONAL_TOPDOWN:
        cond = flip_hm(input[:, 3:], val_dataset, meta['cond_joints'], meta['cond_joints_vis'])
        return torch.cat((input[:, :3].flip(3), cond.to(input.device)), dim=1)
    return input.flip(3)


# The flip test as ONE forward over [crops | mirrored crops]: eval-mode networks treat every image independently (running
# BatchNorm statistics, per-image attention), so the two halves equal the reference's two forwards - one pass through the
# launch sequence instead of two (validation at TEST.BATCH_SIZE <= 16 is bound by the host's ~7 ms per forward), twice the
# rows per launch above that.  Off: two forwards, the reference's literal order.
PAIRED_FLIP_FORWARD = True


def validate(config, val_loader, val_dataset, model, criterion, output_dir, tb_log_dir, writer_dict=None, epoch=-1,
             print_prefix=''):
    batch_time, losses, acc = AverageMeter(), AverageMeter(), AverageMeter()
    model.eval()
    rank, world = _dist_world()
    num_samples = len(val_dataset)
    num_joints = config.MODEL.NUM_JOINTS
    all_preds = np.zeros((num_samples, num_joints, 3), dtype=np.float32)
    all_boxes = np.zeros((num_samples, 6 + 1))
    filled = np.zeros(num_samples, dtype=bool)
    image_path = [None] * num_samples if world > 1 else []
    filenames, imgnums = [], []
    last = len(val_loader) - 1
    conditional = bool(config.MODEL.CONDITIONAL_TOPDOWN)

    # The loop is a three-stage software pipeline over the batches (same results, same order as the reference's loop):
    #   enqueue(i)      the forward of batch i (both halves of the flip test) - through model.submit() where the network offers
    #                   one (engine.ForwardGraph: two graph lanes), so that it runs beside what follows
    #   device_part(i)  flip merge, loss, accuracy decode, final-prediction decode + their copies into pinned memory
    #   host_part(i)    waits for those few KB, fills the tables, logs
    # in the order device_part(i - 1), enqueue(i), host_part(i - 1) (enqueue(i) first where the forward has a lane of its own):
    # the host arithmetic of a batch runs while the GPU is busy with the next batch's forward.
    def enqueue(i, row0, batch):
        input, target, target_weight, meta = batch
        if not input.is_cuda:
            input = (input if conditional else input[:, :3]).cuda(non_blocking=True)
        n = input.size(0)
        if not config.TEST.FLIP_TEST:
            handles = (_submit(model, input),)
        elif PAIRED_FLIP_FORWARD:
            handles = (_submit(model, torch.cat((input, _mirrored_input(config, val_dataset, input, meta)), dim=0)),)
        else:
            handles = (_submit(model, input), _submit(model, _mirrored_input(config, val_dataset, input, meta)))
        return dict(i=i, rows=slice(row0, row0 + n), count=n, input=input, target=target, target_weight=target_weight,
                    meta=meta, handles=handles)

    def device_part(it):
        outs = [h.result() for h in it.pop('handles')]
        outs = [o[-1] if isinstance(o, list) else o for o in outs]
        output = outs[0]
        if config.TEST.FLIP_TEST:
            n = it['count']
            output, flipped = (output[:n], output[n:]) if len(outs) == 1 else (outs[0], outs[1])
            output = flip_merge_device(output.contiguous(), flipped.contiguous(), val_dataset.flip_pairs,
                                       bool(config.TEST.SHIFT_HEATMAP))
        it['target'] = it['target'].cuda(non_blocking=True)
        target_weight = it.pop('target_weight').cuda(non_blocking=True)
        loss = criterion(output, it['target'], target_weight)
        it['stats'] = _DeferredStats(loss, output, it['target'], it['count'])
        it['decode'] = DeferredFinalPreds(config, output)
        it['output'] = output

    def host_part(it, tick):
        i, rows, meta = it['i'], it['rows'], it['meta']
        center = meta['center'].numpy()
        scale = meta['scale'].numpy()
        preds, maxvals = it['decode'].final_preds(center, scale)
        pred = it['stats'].resolve(losses, acc)
        now = time.time()
        batch_time.update(now - tick)
        all_preds[rows, :, 0:2] = preds[:, :, 0:2]
        all_preds[rows, :, 2:3] = maxvals
        all_boxes[rows, 0:2] = center[:, 0:2]
        all_boxes[rows, 2:4] = scale[:, 0:2]
        all_boxes[rows, 4] = np.prod(scale * 200, 1)
        all_boxes[rows, 5] = meta['score'].numpy()
        all_boxes[rows, 6] = meta['annotation_id'].numpy()
        filled[rows] = True
        if world > 1:
            image_path[rows] = list(meta['image'])
        else:
            image_path.extend(meta['image'])
        if i % config.PRINT_FREQ == 0 or i == last:
            logger.info(f'Test: [{i}/{last}]\t'
                        f'Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                        f'Loss {losses.val:.6f} ({losses.avg:.6f})\t'
                        f'Accuracy {acc.val:.3f} ({acc.avg:.3f})')
            if config.DEBUG.DEBUG:
                tag = '{}_epoch_{:09d}_iter_{}_{}'.format(os.path.join(output_dir, 'val'), epoch, i, print_prefix)
                _save_debug_images(config, it['input'], meta, it['target'], pred * 4, it['output'], tag, output_dir=output_dir)
        return now

    with torch.no_grad():
        tick = time.time()
        pending = None
        lanes = callable(getattr(model, 'submit', None))
        for i, row0, batch in _prefetched(_rank_batches(val_loader, rank, world), conditional):
            if lanes:
                # the forward runs on a lane stream of its own: enqueue it first, so that it overlaps the previous batch's
                # forward on the other lane (a lane only waits for what the current stream holds at submit time)
                cur = enqueue(i, row0, batch)
                if pending is not None:
                    device_part(pending)
            else:
                # one stream: the previous batch's decode must sit in front of this forward, or the host would wait for it
                if pending is not None:
                    device_part(pending)
                cur = enqueue(i, row0, batch)
            if pending is not None:
                tick = host_part(pending, tick)
            pending = cur
        if pending is not None:
            device_part(pending)
            tick = host_part(pending, tick)

        if world > 1:
            all_preds, all_boxes, image_path = gather_validation_shards(all_preds, all_boxes, image_path, filled)
            stat = torch.tensor([losses.sum, losses.count, acc.sum, acc.count], dtype=torch.float64)
            if dist.get_backend() == 'nccl':
                stat = stat.cuda()
            dist.all_reduce(stat)
            stat = stat.cpu().tolist()
            losses.avg = stat[0] / stat[1] if stat[1] else 0
            acc.avg = stat[2] / stat[3] if stat[3] else 0

        if rank == 0:
            name_values, perf_indicator = val_dataset.evaluate(config, all_preds, output_dir, all_boxes, image_path,
                                                               epoch, filenames, imgnums)
            tables = name_values if isinstance(name_values, list) else [name_values]
            for table in tables:
                _print_name_value(table, config.MODEL.NAME)
            if writer_dict:
                step = writer_dict['valid_global_steps']
                writer = writer_dict['writer']
                writer.add_scalar('valid_loss', losses.avg, step)
                writer.add_scalar('valid_acc', acc.avg, step)
                for table in tables:
                    writer.add_scalars('valid', dict(table), step)
                writer_dict['valid_global_steps'] = step + 1
        else:
            perf_indicator = None
        if world > 1:
            box = [perf_indicator]
            dist.broadcast_object_list(box, src=0)
            perf_indicator = box[0]
    return perf_indicator


def _print_name_value(name_value, full_arch_name):
    """The markdown table of the reference (function.py:340-357): header, separator, one row of 3-decimal values;
    architecture names longer than 15 characters are abbreviated to 8 + '...'."""
    arch = full_arch_name if len(full_arch_name) <= 15 else full_arch_name[:8] + '...'
    header = ' '.join('| {}'.format(k) for k in name_value.keys())
    row = ' '.join('| {:.3f}'.format(v) for v in name_value.values())
    logger.info('| Arch ' + header + ' |')
    logger.info('|---' * (len(name_value) + 1) + '|')
    logger.info('| ' + arch + ' ' + row + ' |')
