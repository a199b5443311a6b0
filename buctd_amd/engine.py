"""Training runtime of the MI355X engine: flat parameter / gradient arenas, the fused Adam step and the
one-process-per-GPU data-parallel wrapper (RCCL all-reduce over xGMI).

What it replaces in the reference:
  * torch.nn.DataParallel(model).cuda() (tools/train.py:147): one process, per-iteration parameter broadcast,
    scatter / gather and reduce-add to GPU 0.  Here every rank holds resident parameters; the only per-step
    exchange is a bucketed all-reduce of the flat fp32 gradient, launched bucket by bucket on a side stream while
    the backward pass is still producing earlier layers' gradients.
  * optim.Adam(model.parameters(), lr) (lib/utils/utils.py:268-272): ~1800 tensors -> one fused kernel over the
    flat arena (buctd_adam_step), gradient averaging (1/world) folded in.
BatchNorm statistics stay per replica, like under nn.DataParallel (no SyncBN); buffers of rank 0 are the ones a
checkpoint sees (broadcast_buffers()).
"""
import os

import torch
import torch.distributed as dist

from . import nn as bnn
from . import ops

_ALIGN = 4  # floats: every tensor starts on a 16-byte boundary of the arena


class FlatParams:
    """Moves all parameters of `module` into one contiguous fp32 buffer (views keep logical shape and the
    channels_last strides of conv weights) and owns a same-layout gradient arena the backward kernels write into."""

    def __init__(self, module):
        bnn.prepare_module(module)
        self.params = []
        seen = set()
        for p in module.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise ValueError("module has no parameters")
        dev = self.params[0].device
        self.offsets = {}
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("all parameters must be fp32 on one device")
            self.offsets[id(p)] = off
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self._gviews = {}
        for p in self.params:
            v = self._view(self.flat, p)
            v.copy_(p.data)
            p.data = v
            self._gviews[id(p)] = self._view(self.grad, p)
        self.ready_cb = None
        ops.set_grad_arena(self)

    def _view(self, buf, p):
        o = self.offsets[id(p)]
        return torch.as_strided(buf, p.shape, p.stride(), o)

    def span(self, p):
        o = self.offsets[id(p)]
        return o, o + p.numel()

    # ops.grad_target protocol -------------------------------------------------------------
    def __call__(self, p):
        return self._gviews.get(id(p))

    def owns(self, p):
        return id(p) in self._gviews

    def grad_is_arena(self, p):
        return p.grad is not None and p.grad.data_ptr() == self._gviews[id(p)].data_ptr()

    def collect(self):
        """Make the arena hold the current gradient of every parameter (zero where there is none)."""
        for p in self.params:
            g = self._gviews[id(p)]
            if p.grad is None:
                if p.requires_grad:
                    g.zero_()
            elif not self.grad_is_arena(p):
                g.copy_(p.grad)
                p.grad = g


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) semantics on the flat arena."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_sync=None):
        if not isinstance(flat, FlatParams):
            raise TypeError("FusedAdam works on an engine.FlatParams arena")
        super().__init__(flat.params, dict(lr=lr, betas=betas, eps=eps))
        self.flat = flat
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0
        self.grad_sync = grad_sync  # callable returning the gradient scale (1/world) after syncing

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used on the BUCTD path")
        self.flat.collect()
        gscale = self.grad_sync() if self.grad_sync is not None else 1.0
        self.step_count += 1
        g = self.param_groups[0]
        ops.adam_step(self.flat.flat, self.flat.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0],
                      g["betas"][1], g["eps"], self.step_count, gscale)
        ops.weights_updated()  # the kernel wrote the parameters through raw pointers: drop prepared filter images

    def zero_grad(self, set_to_none=True):
        for p in self.flat.params:
            p.grad = None

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


class GradBuckets:
    """Contiguous slices of the gradient arena, cut so that a bucket closes when the backward pass (which
    produces gradients roughly in reverse registration order) has written all of its tensors."""

    def __init__(self, flat, bucket_bytes=48 << 20):
        self.flat = flat
        target = bucket_bytes // 4
        self.buckets = []  # (start, end, set(param ids))
        cur_ids, cur_end, cur_start = set(), None, None
        for p in reversed(flat.params):
            s, e = flat.span(p)
            e = flat.offsets[id(p)] + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            if cur_end is None:
                cur_end = e
            cur_start = s
            cur_ids.add(id(p))
            if cur_end - cur_start >= target:
                self.buckets.append((cur_start, cur_end, cur_ids))
                cur_ids, cur_end = set(), None
        if cur_ids:
            self.buckets.append((cur_start, cur_end, cur_ids))
        self.of_param = {}
        for i, (_, _, ids) in enumerate(self.buckets):
            for pid in ids:
                self.of_param[pid] = i


class DataParallel(torch.nn.Module):
    """Drop-in for torch.nn.DataParallel in tools/train.py:147 / tools/test.py:134 under a
    one-process-per-GPU launch (torchrun): exposes .module, forwards to it, keeps replicas identical.

    Without an initialised process group (single GPU) it is a transparent wrapper."""

    def __init__(self, module, device_ids=None, bucket_bytes=48 << 20, overlap=True):
        super().__init__()
        self.module = module
        self.flat = None
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.bucket_bytes = bucket_bytes
        self.overlap = overlap
        self._handles = []
        self._pending = None
        self._comm_stream = None

    # -- construction-time helpers ---------------------------------------------------------
    def cuda(self, device=None):
        self.module.cuda(device)
        return self

    def flatten(self):
        if self.flat is None:
            self.flat = FlatParams(self.module)
            if self.world > 1:
                dist.broadcast(self.flat.flat, src=0)
                self.broadcast_buffers()
                self.buckets = GradBuckets(self.flat, self.bucket_bytes)
                if self.overlap and self.flat.flat.is_cuda:
                    self._comm_stream = torch.cuda.Stream()
                    ops.set_grad_ready_callback(self._grad_ready)
        return self.flat

    def broadcast_buffers(self):
        """rank 0's BatchNorm running statistics win, like replica 0 under nn.DataParallel."""
        if self.world > 1:
            for b in self.module.buffers():
                dist.broadcast(b, src=0)

    def forward(self, *args, **kwargs):
        if self.world > 1 and self.flat is not None:
            self._start_step()
        return self.module(*args, **kwargs)

    # -- gradient exchange -----------------------------------------------------------------
    def _start_step(self):
        self._handles = []
        self._pending = [len(ids) for (_, _, ids) in self.buckets.buckets]

    def _launch(self, i):
        s, e, _ = self.buckets.buckets[i]
        view = self.flat.grad[s:e]
        if self._comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                ops.wait_side_stream(self._comm_stream)   # weight gradients are produced on ops' side stream
                self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
        else:
            if view.is_cuda:
                ops.wait_side_stream()
            self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def _grad_ready(self, p):
        """Called by the backward ops right after the kernels writing p.grad were enqueued."""
        if self._pending is None:
            return
        i = self.buckets.of_param.get(id(p))
        if i is None or not self.flat.grad_is_arena(p):
            return
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def sync_gradients(self):
        """Finish the all-reduce of every bucket; returns the scale that turns the summed gradient into the
        gradient of the global-batch mean loss (what nn.DataParallel computes)."""
        if self.world == 1:
            return 1.0
        if self._pending is None:
            self._start_step()
        for i, left in enumerate(self._pending):
            if left != 0:  # not launched during backward (overlap off, or a parameter without gradient)
                self._pending[i] = 0
                self._launch(i)
        for h in self._handles:
            h.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._handles, self._pending = [], None
        return 1.0 / self.world

    # -- nn.Module plumbing so that checkpoints carry the reference's 'module.' prefix ------
    def state_dict(self, *args, **kwargs):
        return super().state_dict(*args, **kwargs)


def get_optimizer(cfg, model):
    """reference lib/utils/utils.py:258-274: Adam(lr) for 'adam' (no weight decay), SGD otherwise - here the
    fused flat-arena Adam, wired to the data-parallel gradient exchange when `model` is an engine.DataParallel."""
    if cfg.TRAIN.OPTIMIZER != "adam":
        raise NotImplementedError("every BUCTD recipe trains with Adam (get_optimizer: 'we only use adam')")
    if isinstance(model, DataParallel):
        flat = model.flatten()
        return FusedAdam(flat, lr=cfg.TRAIN.LR, grad_sync=model.sync_gradients)
    return FusedAdam(FlatParams(model), lr=cfg.TRAIN.LR)


def init_distributed():
    """One process per GPU (torchrun env: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        if os.environ.get("BUCTD_SINGLE_DEVICE") == "1":   # test hook: every rank on GPU 0 (needs a non-RCCL backend)
            local = 0
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        backend = os.environ.get("BUCTD_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    else:
        device = torch.device("cpu")
        backend = "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, device
