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
import time

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

    def grad_runs(self):
        """Contiguous [start, end) element runs of the arena that cover exactly the parameters holding a gradient right now
        (p.grad is not None) - torch.optim skips every other parameter entirely (no weight decay, no momentum), e.g.
        TransPose's frozen pos_embedding (transpose_h.py:129) and constructed-but-unused modules.  One run when every
        parameter has a gradient (the usual case)."""
        runs = []
        for p in self.params:
            if p.grad is None:
                continue
            s = self.offsets[id(p)]
            e = s + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            if runs and runs[-1][1] == s:
                runs[-1][1] = e
            else:
                runs.append([s, e])
        return [(s, e) for s, e in runs]


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
        ops.weights_updated()  # the kernel wrote the parameters through raw pointers: prepared filter images are stale
        ops.refresh_prepared(self.flat.flat.device)   # ... and are rebuilt here, by one launch behind the Adam kernel
        if self.flat.flat.is_cuda:
            ops.acc_pool.reset(self.flat.flat.device)  # BatchNorm accumulators of the step: one fill, every stream is joined here

    def zero_grad(self, set_to_none=True):
        self.zeroed = getattr(self, "zeroed", 0) + 1      # (StepGraph: the gradient views a capture installed are gone)
        for p in self.flat.params:
            p.grad = None

    def state_dict(self):
        """torch.optim.Adam's checkpoint format (what the reference stores under checkpoint['optimizer'],
        tools/train.py:243-266): per-parameter 'step' / 'exp_avg' / 'exp_avg_sq' keyed by parameter index."""
        state = {}
        if self.step_count > 0:
            for i, p in enumerate(self.flat.params):
                o, e = self.flat.span(p)
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": torch.as_strided(self.exp_avg, p.shape, p.stride(), o).clone(),
                            "exp_avg_sq": torch.as_strided(self.exp_avg_sq, p.shape, p.stride(), o).clone()}
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            for k, v in (("weight_decay", 0), ("amsgrad", False), ("maximize", False)):
                d.setdefault(k, v)
            d["params"] = list(range(len(self.flat.params)))
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts a torch.optim.Adam state_dict (reference checkpoints) or this class's round-1 flat format."""
        if "state" not in sd:                      # round-1 private format: flat arenas
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        else:
            state = sd["state"]
            steps = set()
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            for i, p in enumerate(self.flat.params):
                st = state.get(i, state.get(str(i)))
                if st is None:
                    continue
                o, _ = self.flat.span(p)
                for name, arena in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                    src = st[name]
                    if tuple(src.shape) != tuple(p.shape):
                        raise ValueError(f"optimizer state of parameter {i}: shape {tuple(src.shape)} != {tuple(p.shape)}")
                    torch.as_strided(arena, p.shape, p.stride(), o).copy_(src)
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError("FusedAdam keeps one step counter: per-parameter steps differ in this state_dict")
            self.step_count = steps.pop() if steps else 0
        for g, src in zip(self.param_groups, sd["param_groups"]):
            for k, v in src.items():
                if k == "params":
                    continue
                if k in ("weight_decay",) and v not in (0, 0.0):
                    raise ValueError("FusedAdam implements Adam without weight decay (the BUCTD recipe)")
                if k in ("amsgrad", "maximize") and v:
                    raise ValueError(f"FusedAdam does not implement {k}")
                g[k] = v


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD(lr, momentum, weight_decay, nesterov) semantics on the flat arena (one kernel per step) - the 'sgd'
    branch of the reference's get_optimizer (lib/utils/utils.py:260-267).  Reads and writes torch.optim.SGD state_dicts."""

    def __init__(self, flat, lr=1e-3, momentum=0.0, weight_decay=0.0, nesterov=False, grad_sync=None):
        if not isinstance(flat, FlatParams):
            raise TypeError("FusedSGD works on an engine.FlatParams arena")
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(flat.params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay,
                                           nesterov=nesterov))
        self.flat = flat
        self.buf = torch.zeros_like(flat.flat) if momentum else None
        self.step_count = 0
        self.grad_sync = grad_sync

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used on the BUCTD path")
        self.flat.collect()
        gscale = self.grad_sync() if self.grad_sync is not None else 1.0
        g = self.param_groups[0]
        # torch.optim.SGD touches only parameters that hold a gradient: a parameter with p.grad None keeps its value (no
        # weight decay) and its momentum buffer.  A zero momentum buffer makes "buf = momentum * buf + g" the first-step
        # initialisation "buf = g" (dampening 0), so parameters that join later need no flag of their own.
        for s, e in self.flat.grad_runs():
            ops.sgd_step(self.flat.flat[s:e], self.flat.grad[s:e], None if self.buf is None else self.buf[s:e], g["lr"],
                         g["momentum"], g["weight_decay"], g["nesterov"], False, gscale)
        self.step_count += 1
        ops.weights_updated()
        ops.refresh_prepared(self.flat.flat.device)
        if self.flat.flat.is_cuda:
            ops.acc_pool.reset(self.flat.flat.device)

    def zero_grad(self, set_to_none=True):
        self.zeroed = getattr(self, "zeroed", 0) + 1
        for p in self.flat.params:
            p.grad = None

    def state_dict(self):
        state = {}
        if self.buf is not None and self.step_count > 0:
            for i, p in enumerate(self.flat.params):
                o, _ = self.flat.span(p)
                state[i] = {"momentum_buffer": torch.as_strided(self.buf, p.shape, p.stride(), o).clone()}
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            for k, v in (("maximize", False), ("foreach", None), ("differentiable", False), ("fused", None)):
                d.setdefault(k, v)
            d["params"] = list(range(len(self.flat.params)))
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        state = sd.get("state", {})
        loaded = 0
        for i, p in enumerate(self.flat.params):
            st = state.get(i, state.get(str(i)))
            if st is None or st.get("momentum_buffer") is None:
                continue
            if self.buf is None:
                raise ValueError("FusedSGD was built without momentum but the state_dict carries momentum buffers")
            o, _ = self.flat.span(p)
            torch.as_strided(self.buf, p.shape, p.stride(), o).copy_(st["momentum_buffer"])
            loaded += 1
        self.step_count = 1 if loaded else 0
        for g, src in zip(self.param_groups, sd["param_groups"]):
            for k, v in src.items():
                if k == "params":
                    continue
                if k == "dampening" and v not in (0, 0.0):
                    raise ValueError("FusedSGD implements dampening = 0 (the reference's recipe)")
                if k == "maximize" and v:
                    raise ValueError("FusedSGD does not implement maximize")
                g[k] = v


class GradBuckets:
    """Contiguous slices of the gradient arena, cut so that a bucket closes when the backward pass (which
    produces gradients roughly in reverse registration order) has written all of its tensors."""

    def __init__(self, flat, bucket_bytes=48 << 20):
        self.flat = flat
        target = bucket_bytes // 4
        self.buckets = []  # (start, end, set(param ids))
        cur_ids, cur_end, cur_start = set(), None, None
        def close():
            nonlocal cur_ids, cur_end
            if cur_ids:
                self.buckets.append((cur_start, cur_end, cur_ids))
            cur_ids, cur_end = set(), None

        for p in reversed(flat.params):
            s, _ = flat.span(p)
            e = flat.offsets[id(p)] + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            if e - s >= target:
                # a tensor as large as a bucket (CoAM fc_o: 191 MB of the 462 MB) travels alone, so that its
                # all-reduce starts the moment its gradient GEMM is enqueued instead of waiting for neighbours
                close()
                self.buckets.append((s, e, {id(p)}))
                continue
            if cur_end is None:
                cur_end = e
            cur_start = s
            cur_ids.add(id(p))
            if cur_end - cur_start >= target:
                close()
        close()
        self.of_param = {}
        for i, (_, _, ids) in enumerate(self.buckets):
            for pid in ids:
                self.of_param[pid] = i


class DataParallel(torch.nn.Module):
    """Drop-in for torch.nn.DataParallel in tools/train.py:147 / tools/test.py:134 under a
    one-process-per-GPU launch (torchrun): exposes .module, forwards to it, keeps replicas identical.

    Without an initialised process group (single GPU) it is a transparent wrapper."""

    def __init__(self, module, device_ids=None, bucket_bytes=48 << 20, overlap=True, exchange_in_world_of_one=False):
        super().__init__()
        self.module = module
        self.flat = None
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        # exchange_in_world_of_one: run the whole gradient exchange (buckets, communication stream, collectives) in an
        # initialised process group of ONE rank too - the sum over one rank is the identity, so the step must equal the
        # plain engine's bit for bit; the hardware test of the RCCL transport on a single-GPU box
        self._exchange = self.world > 1 or (exchange_in_world_of_one and dist.is_available() and dist.is_initialized())
        self.bucket_bytes = bucket_bytes
        self.overlap = overlap
        self._handles = []
        self._pending = False
        self._dirty = False
        self._entry_stream = None
        self._comm_stream = None
        self._collective_in_stream = False
        # bench.py: set to a list to get one (bucket index, bytes, start event, end event) per bucket exchange on the
        # communication stream (timing events; None = off, the production setting)
        self.bucket_trace = None

    # -- construction-time helpers ---------------------------------------------------------
    def cuda(self, device=None):
        self.module.cuda(device)
        return self

    def flatten(self):
        if self.flat is None:
            self.flat = FlatParams(self.module)
            if self._exchange:
                dist.broadcast(self.flat.flat, src=0)
                self.broadcast_buffers()
                self.buckets = GradBuckets(self.flat, self.bucket_bytes)
                self._collective_in_stream = self.flat.flat.is_cuda and dist.get_backend() == "nccl"
                if self.overlap and self.flat.flat.is_cuda:
                    # the exchange runs at the HIGHEST stream priority: an RCCL kernel occupies a few CUs per channel and
                    # must not queue behind the four compute streams' workgroups (it would start when they drain, i.e.
                    # exposed); what it takes away from them is its channel count (RCCL's default: <= 32 CUs of 256)
                    dev = self.flat.flat.device
                    self._comm_stream = _comm_streams.get(dev.index) or torch.cuda.Stream(device=dev, priority=-1)
                    # HIP stream budget: the default runtime serves FOUR hardware queues and any fifth stream costs 25-32 %
                    # (DESIGN.md 3.12, 6).  With the communication stream the compute side gets main + ONE branch stream
                    # (the two group-launch families of a HighResolutionModule, the fuse rows) + the weight-gradient stream.
                    ops.set_branch_max(1)
                    ops.set_grad_ready_callback(self._grad_ready)
        return self.flat

    def broadcast_buffers(self):
        """rank 0's BatchNorm running statistics win, like replica 0 under nn.DataParallel."""
        if self._exchange:
            for b in self.module.buffers():
                dist.broadcast(b, src=0)

    def forward(self, *args, **kwargs):
        if self._exchange and self.flat is not None:
            self._start_step()
        return self.module(*args, **kwargs)

    # -- gradient exchange -----------------------------------------------------------------
    def _start_step(self):
        self._handles = []
        self._dirty = False
        nb = len(self.buckets.buckets)
        self._ready = [set() for _ in range(nb)]       # parameter ids whose gradient is final, per bucket
        self._streams = [dict() for _ in range(nb)]    # streams that produced gradient work of the bucket
        self._launched = [False] * nb
        self._pending = True
        if self.flat.flat.is_cuda:
            # the stream the backward pass is entered on: autograd replays nodes of the main path here
            self._entry_stream = torch.cuda.current_stream(self.flat.flat.device)

    MAX_HIP_STREAMS = 4      # main + branch + weight-gradient + communication (the hardware queues of the default runtime)

    _budget_warned = False

    def _check_stream_budget(self, device):
        """a PERFORMANCE condition, checked from autograd callbacks: it warns once, it never raises"""
        n = 1 + len(ops.compute_streams(device)) + (1 if self._comm_stream is not None else 0)
        if n > self.MAX_HIP_STREAMS and not DataParallel._budget_warned:
            DataParallel._budget_warned = True
            import warnings
            warnings.warn(f"{n} HIP streams on {device} (main + {len(ops.compute_streams(device))} compute + communication): "
                          "more than four hardware queues cost 25-32 % (DESIGN.md 6)", RuntimeWarning, stacklevel=2)

    def _launch(self, i):
        s, e, _ = self.buckets.buckets[i]
        view = self.flat.grad[s:e]
        self._launched[i] = True
        if self._comm_stream is not None:
            self._check_stream_budget(view.device)
            # one event per stream that wrote into this bucket, recorded now: stream order makes it cover the
            # bucket's kernels on that stream (and nothing of streams that did not contribute)
            streams = dict(self._streams[i])
            for st in (torch.cuda.current_stream(view.device), self._entry_stream):
                streams[st.cuda_stream] = st
            for st in streams.values():
                ev = torch.cuda.Event()
                ev.record(st)
                self._comm_stream.wait_event(ev)
            with torch.cuda.stream(self._comm_stream):
                if self.bucket_trace is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self._comm_stream)
                if self._collective_in_stream:
                    # RCCL: a collective issued with async_op=False is enqueued on the CURRENT stream - the communication
                    # stream - and returns at once.  async_op=True would run it on the process group's own internal stream
                    # behind an event: a FIFTH HIP stream, i.e. the 25 % cliff of DESIGN.md 6 (measured with a one-rank
                    # group: 470 -> 351 images/s).  sync_gradients joins the communication stream, no handles needed.
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=False)
                else:
                    self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
                if self.bucket_trace is not None:
                    if not self._collective_in_stream:
                        self._handles[-1].wait()      # stream-side wait only: orders e1 behind the collective
                    e1.record(self._comm_stream)
                    self.bucket_trace.append((i, 4 * (e - s), e0, e1))
        else:
            if view.is_cuda:
                ops.wait_side_stream()
            if self._collective_in_stream:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=False)     # in the current stream, as above
            else:
                self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def _grad_ready(self, p):
        """Called by the backward ops right after the kernels writing p.grad were enqueued (on the current stream
        and, for weight gradients, on ops' side stream)."""
        if not self._pending:
            return
        i = self.buckets.of_param.get(id(p))
        if i is None or not self.flat.grad_is_arena(p):
            return
        if self._launched[i] or id(p) in self._ready[i]:
            # a second gradient write to an already counted parameter (shared weight, or a second backward before
            # step()): the early all-reduce would miss it -> redo everything synchronously in sync_gradients
            self._dirty = True
            return
        if p.is_cuda:
            cur = torch.cuda.current_stream(p.device)
            self._streams[i][cur.cuda_stream] = cur
            for st in ops.side_streams(p.device):
                self._streams[i][st.cuda_stream] = st
        self._ready[i].add(id(p))
        if len(self._ready[i]) == len(self.buckets.buckets[i][2]):
            self._launch(i)

    def sync_gradients(self):
        """Finish the all-reduce of every bucket; returns the scale that turns the summed gradient into the
        gradient of the global-batch mean loss (what nn.DataParallel computes)."""
        if not self._exchange:
            return 1.0
        if not self._pending:
            self._start_step()
        if self._dirty:
            # buckets already summed over the ranks cannot be summed again: drain what is in flight, reset the step state
            # (so that the next step starts clean instead of failing for ever) and tell the caller
            for h in self._handles:
                h.wait()
            if self._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self._comm_stream)
            self._handles, self._pending, self._dirty = [], False, False
            raise RuntimeError("engine.DataParallel: a parameter received a second gradient after its bucket was "
                               "counted (shared weights / gradient accumulation need overlap=False); this step's "
                               "gradients are not usable")
        if self.flat.flat.is_cuda:
            ops.wait_side_stream()          # buckets launched here see every gradient kernel enqueued so far
        for i in range(len(self.buckets.buckets)):
            if not self._launched[i]:       # not closed during backward (overlap off, or a parameter without gradient)
                self._launch(i)
        for h in self._handles:
            h.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._handles, self._pending = [], False
        return 1.0 / self.world

    # (checkpoints carry the reference's 'module.' prefix through nn.Module's own state_dict: the wrapped net is `self.module`)


def get_optimizer(cfg, model):
    """reference lib/utils/utils.py:258-274: SGD(lr, MOMENTUM, WD, NESTEROV) for 'sgd', Adam(lr) for 'adam' - here the fused
    flat-arena optimizers, wired to the data-parallel gradient exchange when `model` is an engine.DataParallel."""
    if isinstance(model, DataParallel):
        flat, sync = model.flatten(), model.sync_gradients
    else:
        flat, sync = FlatParams(model), None
    if cfg.TRAIN.OPTIMIZER == "sgd":
        return FusedSGD(flat, lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WD,
                        nesterov=cfg.TRAIN.NESTEROV, grad_sync=sync)
    if cfg.TRAIN.OPTIMIZER == "adam":
        return FusedAdam(flat, lr=cfg.TRAIN.LR, grad_sync=sync)
    return None       # the reference returns None for any other name (utils.py:259, 274)


# Stream-capture error mode of the two graph classes: "thread_local" - a DataLoader's pin-memory thread (hipHostMalloc) or any
# other thread of the process may call the runtime while a step is being captured; "global" would turn such a call into a
# failed capture.  The kernels autograd's device thread launches into the capturing streams are captured either way.
_CAPTURE_MODE = "thread_local"


class StepGraph:
    """The device work of one training iteration - forward, loss, backward, gradient collection - captured ONCE as a hipGraph
    and replayed per step; the optimizer step stays an eager launch behind it (its bias corrections are launch arguments).

    Why: the eager engine enqueues ~1000-1600 kernels per step through Python (autograd nodes + ctypes calls): 23-35 ms of
    host time.  HRNet-W32 at 256x192 (BASELINE config C2) has 17 ms of GPU work per batch of 32, so the eager step is bound
    by the host; a replay is one hipGraphLaunch.  The streams the engine forks (branch streams, weight-gradient stream) become
    parallel branches of the graph; nothing about the kernels or their order on a stream changes, so a replayed step is
    bit-identical to the eager one (tests/test_gpu_step_graph.py).

    Use (what core.function.train does when handed one):  `output, loss = step(input, target, target_weight)` in place of
    forward / zero_grad / backward / optimizer.step().  The first `warmup` calls with a given input signature run the eager
    engine (they are ordinary training steps: caches, workspaces, streams and LDS limits settle); the next call captures;
    every later call copies the batch into the graph's static input buffers and replays.  A batch of another shape (the last,
    ragged batch of an epoch) runs the eager engine.  `output` and `loss` are the graph's static tensors: read or copy them
    before the next call.

    Not capturable, and refused loudly: train-mode dropout (the mask seed is a launch argument: CoAM / TransPose run eager),
    a gradient exchange over a process group (the buckets are launched from host callbacks)."""

    def __init__(self, model, criterion, optimizer, warmup=3, streams="single", allow_repeated_dropout_masks=False,
                 autoselect=False):
        if streams not in ("single", "engine"):
            raise ValueError("streams: 'single' (a linear graph) or 'engine' (the engine's branch / weight-gradient streams "
                             "become parallel branches of the graph)")
        if not isinstance(optimizer, (FusedAdam, FusedSGD)):
            raise TypeError("StepGraph replays into a flat gradient arena: it needs engine.FusedAdam / FusedSGD")
        if isinstance(model, DataParallel) and model._exchange:
            raise NotImplementedError("StepGraph: the bucketed gradient exchange is launched from host callbacks and is not "
                                      "captured; run the eager engine under a process group")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.warmup = int(warmup)
        self.streams = streams
        self._allow_seeds = bool(allow_repeated_dropout_masks)     # measurement only: every replay repeats one mask
        # autoselect: the last eager settling step and the first two replays of a signature are timed (device drained around
        # them - they are ordinary training steps on the caller's batches) and the signature keeps the faster path: a linear
        # graph wins below ~batch 32 on HRNet-W32 and loses the stream concurrency of the eager engine above
        self.autoselect = bool(autoselect)
        self._eager_s = {}
        self._eager_only = set()
        self._seen = {}
        self._graphs = {}
        self._dropout = False
        self.replays = 0

    @staticmethod
    def _signature(tensors):
        return tuple((tuple(t.shape), t.dtype, t.device) for t in tensors)

    def _forward_backward(self, x, target, weight):
        outputs = self.model(x)
        heads = outputs if isinstance(outputs, list) else [outputs]
        loss = None
        for head in heads:
            term = self.criterion(head, target, weight)
            loss = term if loss is None else loss + term
        self.optimizer.zero_grad()
        loss.backward()
        return heads[-1], loss

    def _capture(self, x, target, weight):
        dev = x.device
        flat = self.optimizer.flat
        static = [t.clone() for t in (x, target, weight)]
        bns = [m for m in self.model.modules() if isinstance(m, bnn.BatchNorm2d)]
        before = [m._pending_batches for m in bns]
        graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad()
        ops.begin_capture(self._allow_seeds)
        forks = ops.set_stream_forks(self.streams == "engine")
        try:
            with torch.cuda.graph(graph, capture_error_mode=_CAPTURE_MODE):
                ops.acc_pool.rebase(dev)              # the pool's ordering event becomes an edge of the graph
                out, loss = self._forward_backward(*static)
                flat.collect()                        # gradients autograd accumulated outside the arena are copied per replay
                ops.acc_pool.zero_used(dev)           # a replay leaves its statistics accumulators zeroed for the next one
                used = ops.acc_pool.used(dev)
        except BaseException:
            # a failed capture must not leave the pool ordered behind a captured event, nor gradient views of a graph that
            # does not exist: the eager engine stays usable
            ops.set_stream_forks(*forks)
            ops.end_capture()
            ops.acc_pool.rebase(dev, ops.acc_pool.used(dev))
            self.optimizer.zero_grad()
            raise
        ops.set_stream_forks(*forks)
        keep = ops.end_capture()
        ops.acc_pool.rebase(dev, used)                # eager edge: zero before the first replay, fresh (uncaptured) event
        grads = [(p, p.grad) for p in flat.params if p.grad is not None]
        counts = [(m, m._pending_batches - b) for m, b in zip(bns, before) if m._pending_batches != b]
        for m, b in zip(bns, before):
            m._pending_batches = b                    # the capture enqueued nothing: its batches are counted per replay
        return {"graph": graph, "static": static, "out": out, "loss": loss, "bn_counts": counts, "keep": keep, "clocked": 0,
                "grads": grads, "zeroed": getattr(self.optimizer, "zeroed", 0)}

    def __call__(self, x, target, weight):
        key = self._signature((x, target, weight))
        g = self._graphs.get(key)
        if g is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n < self.warmup or not x.is_cuda or not self.model.training or key in self._eager_only:
                drawn = ops.seeds_drawn()
                clocked = self.autoselect and x.is_cuda and n == self.warmup - 1 and n > 0
                if clocked:
                    torch.cuda.synchronize(x.device)
                    t0 = time.perf_counter()
                out, loss = self._forward_backward(x, target, weight)
                self.optimizer.step()
                if clocked:
                    torch.cuda.synchronize(x.device)
                    self._eager_s[key] = time.perf_counter() - t0
                self._dropout = self._dropout or ops.seeds_drawn() != drawn
                return out, loss
            if self._dropout and not self._allow_seeds:
                raise NotImplementedError("StepGraph: this model draws train-mode dropout masks; the mask seed is a launch "
                                          "argument, a replay would repeat one mask - run the eager engine")
            g = self._graphs[key] = self._capture(x, target, weight)
        for dst, src in zip(g["static"], (x, target, weight)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        clocked = self.autoselect and key in self._eager_s and g["clocked"] < 2
        if clocked:
            torch.cuda.synchronize(x.device)
            t0 = time.perf_counter()
        g["graph"].replay()
        for m, c in g["bn_counts"]:
            m._pending_batches += c
        self.replays += 1
        if getattr(self.optimizer, "zeroed", 0) != g["zeroed"]:
            # somebody cleared the gradients since the capture (an eager step in between, the caller's own zero_grad): the
            # replay wrote into the arena views the capture had installed - hand them to the parameters again, or the
            # optimizer would take "no gradient" for "zero"
            for p, gr in g["grads"]:
                p.grad = gr
            g["zeroed"] = getattr(self.optimizer, "zeroed", 0)
        self.optimizer.step()
        if clocked:
            torch.cuda.synchronize(x.device)
            g["clocked"] += 1
            g["replay_s"] = min(g.get("replay_s", 1e9), time.perf_counter() - t0)
            if g["clocked"] == 2 and g["replay_s"] > self._eager_s[key]:
                # the eager engine is faster for this signature: drop the graph (its outputs stay valid tensors)
                self._eager_only.add(key)
                out, loss = g["out"], g["loss"]
                del self._graphs[key]
                return out, loss
        return g["out"], g["loss"]

    def static_inputs(self, x, target, weight):
        """The static input buffers of the graph captured for this signature (None before the capture): a loader that writes
        its batches there saves the per-step copy."""
        g = self._graphs.get(self._signature((x, target, weight)))
        return None if g is None else tuple(g["static"])


class ForwardGraph(torch.nn.Module):
    """Eval-mode forward of `module` replayed from a hipGraph, one graph per input signature - the serving path: top-down pose
    inference runs the network on the few person crops of an image (tools/inference.py, lib/core/function.py:178-336 at small
    TEST.BATCH_SIZE), where the eager engine's ~400-900 launches cost more host time than the kernels take on the GPU.

    Drop-in for the module inside validate() / dataset.pipeline.IterativeRefiner: `net = engine.ForwardGraph(net)`; calls in
    train mode, with gradients enabled, or on a CPU tensor go straight to the module.  The first `warmup` calls of a signature
    run the eager engine (filter images, folded BatchNorms and workspaces settle), the next one captures a LINEAR graph (the
    runtime re-enqueues a graph with cross-queue edges node by node: DESIGN.md 3.14l), later ones copy the input into the
    graph's static buffer and replay.  Outputs are copies by default (`static_output=True` hands out the graph's own output
    tensors: valid until the next call of the same signature).  Parameters are read at replay time, so a checkpoint loaded in
    place is picked up - but anything keyed on the weights (prepared filter images, folded BatchNorms) is rebuilt by the eager
    engine, not by a replay: engine.FusedAdam steps and in-place rewrites (load_state_dict: a sample of tensor versions is
    watched) drop the graphs automatically; `reset()` does it by hand."""

    def __init__(self, module, warmup=2, static_output=False, max_graphs=8, autoselect=True):
        super().__init__()
        self.module = module
        self.warmup, self.static_output, self.max_graphs = int(warmup), bool(static_output), int(max_graphs)
        # autoselect: right after a capture the eager forward and the replay are timed (three runs each, device drained) and
        # the signature keeps the faster one - a linear graph loses the stream concurrency of the eager engine, which is worth
        # ~10 % from 16 persons per call on (profiles/r06_forward_graph_by_batch.txt)
        self.autoselect = bool(autoselect)
        self._seen, self._graphs = {}, {}
        self._epoch = None
        self._probe = None
        self._capture_streams = []
        self.replays = 0

    def reset(self):
        self._seen, self._graphs = {}, {}

    def forward(self, x, *args, **kwargs):
        if (self.module.training or torch.is_grad_enabled() or args or kwargs or not torch.is_tensor(x) or not x.is_cuda):
            return self.module(x, *args, **kwargs)
        # an optimizer step (epoch) or an in-place rewrite of the parameters (tensor versions of a sample of them: a
        # load_state_dict bumps every one) makes the eager engine rebuild prepared filter images and folded BatchNorms - a
        # replay would not: drop the graphs
        if self._probe is None:
            ps = list(self.module.parameters()) + list(self.module.buffers())
            self._probe = ps[::max(1, len(ps) // 16)]
        epoch = (ops.weights_epoch(), tuple(p._version for p in self._probe))
        if epoch != self._epoch:
            self.reset()
            self._epoch = epoch
        key = (tuple(x.shape), x.dtype, x.device, x.stride())
        g = self._graphs.get(key)
        if g is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n < self.warmup or len(self._graphs) >= self.max_graphs:
                return self.module(x)
            g = self._graphs[key] = self._capture(x)
        if not g["use"]:
            return self.module(x)
        if g["x"].data_ptr() != x.data_ptr():
            g["x"].copy_(x, non_blocking=True)
        g["graph"].replay()
        self.replays += 1
        out = g["out"]
        if self.static_output:
            return out
        return [o.clone() for o in out] if isinstance(out, list) else out.clone()

    # ---- two lanes: independent requests beside each other -------------------------------------------------------------
    def submit(self, x):
        """Asynchronous forward for a serving loop: returns a handle at once, `handle.result()` makes the current stream wait
        and returns the output (a copy).  Requests of a captured signature alternate between `LANES` graphs that replay on
        the engine's two branch streams (no new HIP stream): the launch-bound kernels of independent requests run beside
        each other - one person per call keeps ~180 kernels of a few microseconds each in flight, far from filling the
        chip.  Submit a few requests before collecting the first.  Anything that cannot be replayed (train mode, gradients,
        a signature still settling, autoselect preferring the eager engine) is computed on the spot."""
        if (self.module.training or torch.is_grad_enabled() or not torch.is_tensor(x) or not x.is_cuda):
            return _Ready(self.module(x))
        key = (tuple(x.shape), x.dtype, x.device, x.stride())
        g = self._graphs.get(key)
        if g is None or not g["use"]:
            return _Ready(self.forward(x))          # settles / captures lane 0 (and watches the weights)
        probe = (ops.weights_epoch(), tuple(p._version for p in self._probe))
        if probe != self._epoch:
            return _Ready(self.forward(x))
        lanes = g.setdefault("lanes", [g])
        while len(lanes) < self.LANES:              # further lanes: their own graph, buffers and workspaces
            lanes.append(self._capture(x, lane=len(lanes), clock=False))
        k = g["next"] = (g.get("next", -1) + 1) % self.LANES
        lane = lanes[k]
        cur = torch.cuda.current_stream(x.device)
        st = ops.lane_stream(x.device, k)
        st.wait_stream(cur)                         # x is ready; the lane's previous request has been copied out (stream order)
        with torch.cuda.stream(st):
            lane["x"].copy_(x, non_blocking=True)
            lane["graph"].replay()
            out = lane["out"]
            out = [o.clone() for o in out] if isinstance(out, list) else out.clone()
            ev = torch.cuda.Event()
            ev.record(st)
        x.record_stream(st)
        self.replays += 1
        return _Pending(out, ev, x.device)

    LANES = 2

    def _capture(self, x, lane=0, clock=True):
        static = x.clone()
        graph = torch.cuda.CUDAGraph()
        # every lane is captured on a stream of its own: the engine's scratch buffers are kept per stream, so two lanes that
        # replay beside each other never share one (the capture streams themselves never execute anything)
        if lane >= len(self._capture_streams):
            self._capture_streams.extend(torch.cuda.Stream(device=x.device) for _ in range(lane + 1 - len(self._capture_streams)))
        ops.begin_capture(allow_seeds=True)     # eval mode: the attention cores draw a seed but drop nothing (p = 0)
        forks = ops.set_stream_forks(False)
        try:
            with torch.cuda.graph(graph, stream=self._capture_streams[lane], capture_error_mode=_CAPTURE_MODE):
                out = self.module(static)
        finally:
            ops.set_stream_forks(*forks)
            keep = ops.end_capture()
        if not clock:
            return {"graph": graph, "x": static, "out": out, "keep": keep, "use": True}
        use = True
        if self.autoselect:
            import time

            def clock(fn):
                fn()
                torch.cuda.synchronize(x.device)
                t0 = time.perf_counter()
                for _ in range(3):
                    fn()
                torch.cuda.synchronize(x.device)
                return time.perf_counter() - t0
            use = clock(graph.replay) <= clock(lambda: self.module(static))
        return {"graph": graph, "x": static, "out": out, "keep": keep, "use": use}


class _Ready:
    """A result that is already enqueued on the current stream (ForwardGraph.submit when nothing was replayed)."""

    def __init__(self, out):
        self._out = out

    def result(self):
        return self._out


class _Pending:
    """A request replaying on a lane stream: result() orders the current stream behind it."""

    def __init__(self, out, event, device):
        self._out, self._event, self._device = out, event, device

    def result(self):
        cur = torch.cuda.current_stream(self._device)
        cur.wait_event(self._event)
        for o in (self._out if isinstance(self._out, list) else [self._out]):
            o.record_stream(cur)
        return self._out


_comm_streams = {}      # device index -> the communication stream reserved by reserve_streams


def reserve_streams(device, data_parallel):
    """Create AND use the HIP streams of the engine on `device` before anything else creates streams there - in particular
    before the RCCL communicator, which opens internal streams of its own.  The runtime binds a stream to one of its four
    hardware queues when the stream is first used; the first four get a queue each, later ones share - and two busy streams
    on one queue serialise (a wait of one blocks the kernels of the other).  Measured on one MI355X with a one-rank RCCL
    group: process group first, engine streams later = 352-357 images/s; engine streams first = 479 (the plain engine: 470-479).
    data_parallel: reserve main + ONE branch stream + weight-gradient stream + communication stream (and cap the branch
    streams at one); otherwise main + the branch streams + the weight-gradient stream."""
    if device.type != "cuda":
        return
    streams = []
    if data_parallel:
        ops.set_branch_max(1)
        if device.index not in _comm_streams:
            _comm_streams[device.index] = torch.cuda.Stream(device=device, priority=-1)
    streams += ops.reserve_compute_streams(device)
    if data_parallel:
        streams.append(_comm_streams[device.index])
    probe = torch.zeros(64, device=device)
    main = torch.cuda.current_stream(device)
    for st in streams:
        st.wait_stream(main)
        with torch.cuda.stream(st):
            probe.add_(1.0)
        main.wait_stream(st)
    torch.cuda.synchronize(device)


def init_distributed():
    """One process per GPU (torchrun env: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). Returns (rank, world, device).
    Two TEST hooks are read here and nowhere else in the engine (tests/test_gpu_ddp.py runs several ranks on the one GPU of a
    test box): BUCTD_SINGLE_DEVICE=1 puts every rank on GPU 0, BUCTD_DIST_BACKEND picks the process-group backend ("gloo" for
    that case; the default "nccl" is RCCL on ROCm).  bench.py reads BUCTD_BENCH_* (its CPU-baseline child, first-touch priming)."""
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
        reserve_streams(device, data_parallel=True)     # before RCCL opens its own streams
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    # every replica draws its own dropout masks (like the replicas of nn.DataParallel), derived from the user's seed
    ops.manual_seed(ops.mix_seed(torch.initial_seed(), rank))
    return rank, world, device
