"""Tensor-level wrappers over the C ABI (buctd_amd/_C.py) plus the autograd Functions the
model mirror is written with.

Internal layout: activations are contiguous fp32 NHWC tensors ``[N, H, W, C]`` (token tensors
``[B, T, C]`` are the same memory), conv weights are logical OIHW Parameters stored
channels_last (= physical ``[Co][R][S][Ci]``), Linear weights ``[out, in]``.

Nothing here falls back to eager PyTorch arithmetic: every numeric op is a libbuctd_hip.so
kernel; torch is used for allocation (caching allocator), streams and autograd bookkeeping.
Parameter gradients are written by the kernels straight into ``param.grad`` (allocated from a
flat arena when one is registered, see buctd_amd/engine.py) instead of being returned to
autograd, so no AccumulateGrad add kernels run.
"""
import ctypes as C
import math
import os

import torch

from . import _C
from ._C import ConvDesc, MatmulDesc, check, lib, ptr, stream_ptr

# Engine switches and their product defaults.  The product reads NO environment variable for them: experiments (scratch/ A/B
# scripts) set BUCTD_TUNING=1, and only then buctd_amd/_tuning.py is imported and overrides entries from BUCTD_<NAME>.
# (The only other variables the package looks at are the two test hooks of engine.init_distributed - BUCTD_SINGLE_DEVICE,
# BUCTD_DIST_BACKEND - documented there.)
_SW = {"CONV_MATH": "bf16x6", "PREP_BATCH": "1", "GCONV_X6": "1", "GCONV_MASK": "15", "NATIVE_BLOCK": "1", "FUSED_BOTTLENECK": "1",
       "FUSE_BN_IN": "1", "FC_O_X6": "1", "MHA_X6": "1", "MHA_PRESPLIT": "1", "ATTN_X6": "1", "WGRAD_STREAM": "1", "WGRAD_STREAMS": "1",
       "WGRAD_PRIO": "-1", "BRANCH_STREAMS": "1", "BRANCH_MAX": "2", "BRANCH_PRIO": "0", "C3_PERSISTENT": "0", "STATS_ZERO_COPY": "1", "FUSE_BWD_BNSTAT": "0", "FUSE_BWD_BNSTAT_S0": "0"}
if os.environ.get("BUCTD_TUNING") == "1":
    from . import _tuning
    _tuning.override(_SW)

# hipGraph capture of a training step (engine.StepGraph)
# --------------------------------------------------------------------------------------
# While a step is being captured, a wait on an event that was recorded BEFORE the capture began must not be issued: HIP
# refuses it (hipErrorStreamCaptureIsolation), and it is not needed - the capture starts behind a device synchronisation
# and a replay is launched into the stream that carried the eager work in front of it.  Events recorded inside the capture
# are registered here and waited on as usual (they are edges of the graph).
_capture = {"on": False, "events": set(), "allow_seeds": False}


def capturing():
    return _capture["on"]


def _new_event(stream=None):
    """An event recorded now on `stream` (default: the current one) that other streams may have to wait on later."""
    ev = torch.cuda.Event()
    ev.record(stream) if stream is not None else ev.record()
    if _capture["on"]:
        _capture["events"].add(id(ev))
        _capture.setdefault("keep", []).append(ev)     # ids stay unique while the capture lasts
    return ev


def _wait_event(stream, ev):
    if _capture["on"] and id(ev) not in _capture["events"]:
        return
    stream.wait_event(ev)


def begin_capture(allow_seeds=False):
    _capture.update(on=True, events=set(), keep=[], allow_seeds=bool(allow_seeds))


def end_capture():
    """-> what must stay alive as long as the captured graph does (events, workspaces outgrown during the capture)"""
    keep = _capture.get("keep", [])
    _capture.update(on=False, events=set(), keep=[], allow_seeds=False)
    return keep


# --------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------
# workspace + RNG seed bookkeeping
# --------------------------------------------------------------------------------------
_workspaces = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream); ops on one stream run in order, so
    sharing it between consecutive ops is safe."""
    key = (device.index, _C.current_stream_handle())
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def workspace_on(stream, nbytes, device):
    """The scratch buffer of `stream` (a torch.cuda.Stream) without making it current: entering a stream context costs
    ~15 us of host time, per BasicBlock backward.  The buffer is never returned to the allocator, so which stream it
    was allocated under does not matter."""
    key = (device.index, stream.cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        # growth (rare: the first step visits increasingly large shapes): the old buffer may still be written by kernels
        # queued on `stream`, and it was allocated under whatever stream was current - it is kept alive until an event
        # recorded on `stream` here has passed (checked at the next growth), then handed back to the allocator
        if _capture["on"]:
            # an event query is not permitted while a stream captures; the outgrown buffer simply stays alive with the graph
            if buf is not None:
                _capture["keep"].append(buf)
        else:
            _retired_workspaces[:] = [(b, ev) for b, ev in _retired_workspaces if not ev.query()]
            if buf is not None:
                ev = torch.cuda.Event()
                ev.record(stream)
                _retired_workspaces.append((buf, ev))
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        buf.record_stream(stream)
        _workspaces[key] = buf
    return buf


_retired_workspaces = []     # (buffer, event on the stream that may still use it)


# --------------------------------------------------------------------------------------
# zeroed accumulator pool (BatchNorm statistics without finalize launches, csrc/bn_acc.h)
# --------------------------------------------------------------------------------------
class _AccPool:
    """Hands out slices of a ZEROED device buffer.  A BatchNorm statistics accumulator must be zero when its producing
    kernel starts and is dead once its consumer ran - within one forward or backward call - so a slice is never returned:
    `reset()` (the optimizer step: every stream has been joined) zeroes what was handed out with ONE fill and rewinds; a
    process that never resets (tests, plain forward passes) gets a fresh zeroed chunk whenever the current one is used up.
    Streams other than the one that zeroed a chunk are ordered behind the fill by an event, once."""

    CHUNK = 32 << 20

    def __init__(self):
        self.state = {}

    def _new_chunk(self, device, nbytes):
        cur = torch.cuda.current_stream(device)
        buf = torch.zeros(max(self.CHUNK, nbytes), dtype=torch.uint8, device=device)
        ev = _new_event(cur)
        gen = self.state[device.index]["gen"] + 1 if device.index in self.state else 0
        st = {"buf": buf, "off": 0, "ev": ev, "stream": cur.cuda_stream, "seen": {cur.cuda_stream}, "gen": gen}
        self.state[device.index] = st
        return st

    def generation(self, device_index):
        """Counts the rewinds / chunk changes of a device's pool: a slice taken under an older generation has been handed
        out again (AccRef checks it)."""
        st = self.state.get(device_index)
        return st["gen"] if st is not None else -1

    def take(self, nbytes, device):
        """-> device address of `nbytes` zero bytes (256-byte aligned), usable on the current stream"""
        nbytes = (int(nbytes) + 255) & ~255
        st = self.state.get(device.index)
        if st is None or st["off"] + nbytes > st["buf"].numel():
            st = self._new_chunk(device, nbytes)
        if _C.current_stream_handle() not in st["seen"]:       # (the raw handle: building a Stream object costs ~5 us)
            cur = torch.cuda.current_stream(device)
            _wait_event(cur, st["ev"])
            st["buf"].record_stream(cur)
            st["seen"].add(cur.cuda_stream)
        p = st["buf"].data_ptr() + st["off"]
        st["off"] += nbytes
        return p

    def reset(self, device):
        """Call where every stream that used the pool has been joined into the current one (FusedAdam.step)."""
        st = self.state.get(device.index)
        if st is None or st["off"] == 0:
            return
        self.rebase(device, st["off"])

    def rebase(self, device, nbytes=0):
        """Zero the first `nbytes` of the current chunk on the current stream, rewind, and make the current stream the one
        every other stream orders itself behind.  engine.StepGraph calls it at both edges of a capture: inside, so that the
        ordering event of the pool is an edge of the graph; behind, so that eager code never meets a captured event."""
        st = self.state.get(device.index)
        if st is None:
            return 0
        cur = torch.cuda.current_stream(device)
        if nbytes:
            st["buf"][:nbytes].zero_()
        high = st["off"]
        st["off"] = 0
        st["gen"] += 1
        st["ev"], st["stream"], st["seen"] = _new_event(cur), cur.cuda_stream, {cur.cuda_stream}
        st["buf"].record_stream(cur)
        return high

    def used(self, device):
        st = self.state.get(device.index)
        return st["off"] if st is not None else 0

    def zero_used(self, device):
        """Zero what has been handed out so far (current stream), without rewinding."""
        st = self.state.get(device.index)
        if st is not None and st["off"]:
            st["buf"][:st["off"]].zero_()


acc_pool = _AccPool()


def acc_bytes(Cn):
    return _memo(("accb", Cn), lambda: int(lib().buctd_bn_acc_bytes(Cn)))


def step_boundary(device):
    """Rewind the accumulator pool: every accumulator handed out so far is dead (its consumer ran) and every stream that used
    one has been joined into the current stream.  The fused optimizers call this from step(); a training loop around any OTHER
    optimizer (get_optimizer returns None for names the reference does not know either) calls it once per step - without it
    the pool stays correct but takes, and zero-fills, a fresh 32 MB chunk whenever the current one is used up."""
    acc_pool.reset(device)


class AccRef:
    """A statistics accumulator of Cn channels taken from the pool (device address only: the pool owns the memory).  `ptr`
    refuses to serve a reference kept across a rewind of the pool - its slice has been handed out again."""
    __slots__ = ("_ptr", "Cn", "_dev", "_gen")

    def __init__(self, Cn, device, n=1):
        self.Cn = Cn
        self._ptr = acc_pool.take(n * acc_bytes(Cn), device)
        self._dev = device.index
        self._gen = acc_pool.generation(device.index)

    @property
    def ptr(self):
        if acc_pool.generation(self._dev) != self._gen:
            raise _C.BuctdHipError("stale statistics accumulator: the pool was rewound (optimizer step / ops.step_boundary) after "
                                   "this reference was taken")
        return self._ptr


_seed_state = {"seed": None, "counter": 0}


def mix_seed(seed, rank=0):
    """splitmix64 of (seed, rank): distinct, well-spread dropout streams per replica."""
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(rank) + 1)) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def manual_seed(seed):
    _seed_state["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _seed_state["counter"] = 0


def seeds_drawn():
    """How many dropout seeds have been handed out (engine.StepGraph: did a warm-up step draw any?)."""
    return _seed_state["counter"]


def next_seed():
    if _capture["on"] and not _capture["allow_seeds"]:
        raise _C.BuctdHipError("a dropout seed was drawn while a step graph was being captured: the seed is a launch argument, "
                               "a replay would repeat this step's mask - models with train-mode dropout (CoAM, TransPose) "
                               "run the eager engine")
    if _seed_state["seed"] is None:
        # first use without an explicit ops.manual_seed: follow torch.manual_seed (and the rank, if a process group
        # is up), so that runs honour the user's seed and replicas draw different masks
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        manual_seed(mix_seed(torch.initial_seed(), rank))
    _seed_state["counter"] += 1
    return (_seed_state["seed"] * 0x9E3779B97F4A7C15 + _seed_state["counter"] * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def _f32(t, name="tensor"):
    if t.dtype != torch.float32:
        raise _C.BuctdHipError(f"{name}: fp32 expected, got {t.dtype}")
    if not t.is_contiguous():
        raise _C.BuctdHipError(f"{name}: contiguous tensor expected")
    return t


def weight_rsc(w):
    """Check that the weight memory is [Co][R][S][Ci]: a channels_last OIHW conv weight, or a
    2-d [out][in] Linear weight (== [out][1][1][in])."""
    if w.dim() == 2:
        if not w.is_contiguous():
            raise _C.BuctdHipError("Linear weight must be contiguous")
    elif w.dim() != 4:
        raise _C.BuctdHipError("conv weight must be 4-d OIHW or 2-d [out,in]")
    elif not w.is_contiguous(memory_format=torch.channels_last):
        raise _C.BuctdHipError("conv weight must be stored channels_last (call buctd_amd.nn.prepare_module)")
    return w


def _wshape(w):
    return tuple(w.shape) if w.dim() == 4 else (w.shape[0], w.shape[1], 1, 1)


def _new_like_weight(w):
    return torch.empty_like(w, memory_format=torch.preserve_format)


# --------------------------------------------------------------------------------------
# parameter-gradient sink
# --------------------------------------------------------------------------------------
_grad_arena = None


def set_grad_arena(arena):
    """arena(param) -> preallocated grad view (or None). Installed by engine.FlatParams."""
    global _grad_arena
    _grad_arena = arena


_grad_ready_cb = None


def set_grad_ready_callback(cb):
    """cb(param) is invoked right after the kernels that finalise param.grad were enqueued
    (engine.DataParallel uses it to start the bucket's all-reduce while backward continues)."""
    global _grad_ready_cb
    _grad_ready_cb = cb


def grad_done(*params):
    if _grad_ready_cb is not None:
        for p in params:
            if p is not None and p.requires_grad:
                _grad_ready_cb(p)


def grad_target(p):
    """Returns (tensor to write the gradient into, accumulate flag)."""
    if p.grad is not None:
        return p.grad, 1
    g = _grad_arena(p) if _grad_arena is not None else None
    if g is None:
        g = torch.empty_like(p, memory_format=torch.preserve_format)
    p.grad = g
    return g, 0


# --------------------------------------------------------------------------------------
# raw ops
# --------------------------------------------------------------------------------------
# Host-side memo of pure per-shape facts (descriptor structs, "does kernel X take this shape", statistics grouping): a
# train step asks the library the same ~60 questions 1500 times; a dict lookup is 20x cheaper than the ctypes call.
_desc_cache = {}
_shape_memo = {}


def conv_desc(x_shape, w_shape, stride, pad):
    key = (tuple(x_shape), tuple(w_shape), stride, pad)
    d = _desc_cache.get(key)
    if d is not None:
        return d
    N, H, W, Ci = x_shape
    Co, Ci2, R, S = w_shape
    if Ci2 != Ci:
        raise _C.BuctdHipError(f"conv: input has {Ci} channels, weight expects {Ci2}")
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - S) // stride + 1
    d = ConvDesc(N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo)
    if len(_desc_cache) < 4096:
        _desc_cache[key] = d          # read-only by convention: every consumer passes it by reference to the library
    return d


def _memo(key, fn):
    v = _shape_memo.get(key)
    if v is None:
        v = fn()
        if len(_shape_memo) < 16384:
            _shape_memo[key] = v
    return v


_CONV_MATH_MODES = ("fp32", "bf16x6", "bf16x3")
_conv_math = {"mode": _SW["CONV_MATH"]}


def set_conv_math(mode):
    """How the 3x3 / stride-1 / pad-1 convolutions (forward, data gradient, weight gradient) are computed:
    'fp32'   - v_mfma_f32_16x16x4_f32, bitwise an fmaf chain (conv.hip), 157 TFLOP/s peak;
    'bf16x6' - the default: fp32 operands split EXACTLY into three bf16 pieces, six bf16 MFMAs per product with fp32
               accumulation (conv3x3.hip); the dropped piece products are below one fp32 rounding, so this is
               fp32-class arithmetic at 2.5 PF / 6 = 417 TFLOP/s-equivalent;
    'bf16x3' - optional reduced precision (two pieces, three MFMAs, ~2^-16 per product); also switches the fused
               CoAM attention to its bf16 variants.  Never used for a parity claim or the headline benchmark.
    Everything else (1x1, strided, 7x7 convolutions, GEMMs) is fp32 MFMA in every mode."""
    if mode not in _CONV_MATH_MODES:
        raise ValueError(f"conv math mode must be one of {_CONV_MATH_MODES}")
    _conv_math["mode"] = mode


if _conv_math["mode"] not in _CONV_MATH_MODES:
    raise ValueError(f"BUCTD_CONV_MATH must be one of {_CONV_MATH_MODES}")


def get_conv_math():
    return _conv_math["mode"]


def _c3fn(suffix):
    """libbuctd_hip entry point of the current split mode, e.g. buctd_conv3x3_bf16x6_prep."""
    return getattr(lib(), "buctd_conv3x3_" + _conv_math["mode"] + suffix)


def _bf16x3_ok(d):
    """True when this convolution takes the split-bf16 3x3 kernel in the current math mode."""
    mode = _conv_math["mode"]
    if mode == "fp32" or d.R != 3 or d.S != 3 or d.stride != 1 or d.pad != 1:
        return False
    return _memo(("c3ok", mode, d.N, d.H, d.W, d.Ci, d.Co),
                 lambda: _c3fn("_supported")(d.N, d.H, d.W, d.Ci, d.Co) == 1)


_weights_epoch = {"n": 0}
# After an optimizer step every prepared filter image of the model is rebuilt by ONE launch on the stream of the step
# (refresh_prepared, called by FusedAdam) instead of ~430 small launches in front of the convolutions that need them:
# off the critical path of the next forward / backward, 430 launches less per step.  BUCTD_PREP_BATCH=0: lazy refresh.
_PREP_BATCH = _SW["PREP_BATCH"] == "1"
_prep_registry = {"weights": [], "table": None, "table_key": None, "event": None, "stream": None, "waited": set()}


def weights_epoch():
    """Counts the in-place parameter rewrites announced by weights_updated() (the fused optimizers)."""
    return _weights_epoch["n"]


def weights_updated():
    """Called by whoever rewrites parameters through raw pointers (FusedAdam): invalidates prepared filter images."""
    _weights_epoch["n"] += 1


def refresh_prepared(device):
    """Rebuild all registered prepared filter images now (one launch on the current stream)."""
    if _PREP_BATCH and _conv_math["mode"] != "fp32":
        _prep_all(device)


class _PrepItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wprep", C.c_void_p), ("Ci", C.c_int), ("Co", C.c_int), ("flip", C.c_int),
                ("reserved", C.c_int), ("piece_begin", C.c_long)]


def _prep_all(device):
    """Refresh every registered prepared image whose filter was rewritten in place (same storage, new epoch) with ONE
    launch instead of one per filter and direction (~430 per CoAM-W48 train step)."""
    epoch = _weights_epoch["n"]
    mode = _conv_math["mode"]
    np_pieces = 3 if mode == "bf16x6" else 2
    unit = 24 if np_pieces == 3 else 16          # image bytes per piece of the batched kernel
    items, live, total = [], [], 0
    for ref in _prep_registry["weights"]:
        w = ref()
        if w is None or not w.is_cuda or w.device != device:
            continue
        cache = getattr(w, "_buctd_prep", None)
        if cache is None or cache[0][0] != w.data_ptr():
            continue
        live.append(ref)
        if cache[0] == (w.data_ptr(), w._version, epoch, mode) or cache[0][3] != mode:
            continue
        Co, Ci = _wshape(w)[0], _wshape(w)[1]
        for flip in (0, 1):
            img = cache[1 + flip]
            if img is None:
                continue
            items.append((w.data_ptr(), img.data_ptr(), Ci, Co, flip, np_pieces, total))
            total += img.numel() // unit
        cache[0] = (w.data_ptr(), w._version, epoch, mode)
    _prep_registry["weights"] = live
    launched = False
    if items:
        key = tuple(items)
        if _prep_registry["table_key"] != key:
            arr = (_PrepItem * len(items))(*[_PrepItem(*it) for it in items])
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            _prep_registry["table"] = host.to(device)
            _prep_registry["table_key"] = key
        check(lib().buctd_conv3x3_bf16x3_prep_batched(ptr(_prep_registry["table"]), len(items), total, stream_ptr()),
              "conv3x3 prep_batched")
        launched = True
    launched = _gprep_all(device, epoch) or launched
    if not launched:
        return
    ev = torch.cuda.Event()
    ev.record()
    _prep_registry["event"], _prep_registry["stream"] = ev, torch.cuda.current_stream(device).cuda_stream
    _prep_registry["waited"] = set()


def _prep_wait(device):
    """A batched refresh ran on the optimizer's stream: other streams order themselves behind it once."""
    ev = _prep_registry["event"]
    if ev is not None and not _capture["on"]:     # a capture is ordered behind the refresh as a whole
        cur = _C.current_stream_handle()
        if cur != _prep_registry["stream"] and cur not in _prep_registry["waited"]:
            torch.cuda.current_stream(device).wait_event(ev)
            _prep_registry["waited"].add(cur)


_gprep_registry = {"weights": [], "table": None, "table_key": None}


def _gprep_all(device, epoch):
    """The gathered-convolution images (csrc/conv_gather_x6.hip) of every registered filter in ONE launch; True if launched."""
    if _conv_math["mode"] != "bf16x6":
        return False
    items, live = [], []
    for ref in _gprep_registry["weights"]:
        w = ref()
        if w is None or not w.is_cuda or w.device != device:
            continue
        cache = getattr(w, "_buctd_gprep", None)
        if cache is None or cache[0][0] != w.data_ptr():
            continue
        live.append(ref)
        if cache[0] == (w.data_ptr(), w._version, epoch):
            continue
        Co, Ci = _wshape(w)[0], _wshape(w)[1]
        for direction in (0, 1):
            img = cache[2 + direction]
            if img is not None:
                items.append((cache[1], Ci, Co, w.data_ptr(), direction, img.data_ptr()))
        cache[0] = (w.data_ptr(), w._version, epoch)
    _gprep_registry["weights"] = live
    if not items:
        return False
    key = tuple(items)
    if _gprep_registry["table_key"] != key:
        isz = int(lib().buctd_gconv_x6_prep_item_bytes())
        buf = C.create_string_buffer(isz * len(items))
        base = C.addressof(buf)
        for i, (kind, Ci, Co, wptr, direction, iptr) in enumerate(items):
            check(lib().buctd_gconv_x6_prep_item(kind, Ci, Co, C.c_void_p(wptr), direction, C.c_void_p(iptr),
                                                 C.c_void_p(base + i * isz)), "gconv_x6_prep_item")
        _gprep_registry["table"] = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(device)
        _gprep_registry["table_key"] = key
    check(lib().buctd_gconv_x6_prep_batched(ptr(_gprep_registry["table"]), len(items), stream_ptr()), "gconv_x6_prep_batched")
    return True


def _conv3x3_prepared(w, flip):
    """bf16 hi|lo stage image of a 3x3 filter (conv3x3.hip), cached on the weight tensor until it changes."""
    import weakref
    Co, Ci = _wshape(w)[0], _wshape(w)[1]
    epoch = _weights_epoch["n"]
    key = (w.data_ptr(), w._version, epoch, _conv_math["mode"])
    cache = getattr(w, "_buctd_prep", None)
    if cache is not None and cache[0] != key and cache[0][:2] == key[:2] and cache[0][3:] == key[3:] and _PREP_BATCH:
        _prep_all(w.device)       # rewritten in place by the optimizer kernel: batch-refresh all registered images
    if cache is None or cache[0] != key:
        first = cache is None
        cache = [key, None, None]
        try:
            w._buctd_prep = cache
            if first:
                _prep_registry["weights"].append(weakref.ref(w))
        except (AttributeError, RuntimeError, TypeError):
            pass
    if cache[1 + flip] is None:
        nbytes = _c3fn("_prep_bytes")(Ci, Co, flip)
        img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        check(_c3fn("_prep")(Ci, Co, ptr(w), flip, ptr(img), stream_ptr()), "conv3x3 prep")
        cache[1 + flip] = img
    _prep_wait(w.device)
    return cache[1 + flip]


def _conv3x3_bf16x3(x, w, flip, cin, cout, bias, scale, shift, residual, relu, stats, in_bn=None):
    N, H, W = x.shape[0], x.shape[1], x.shape[2]
    wp = _conv3x3_prepared(w, flip)
    y = torch.empty((N, H, W, cout), dtype=torch.float32, device=x.device)
    part = counts = info = None
    acc_mode = (_conv_math["mode"] == "bf16x6" and bias is None and scale is None
                and (stats == "acc" or isinstance(in_bn, BnAccInput)))
    if acc_mode:
        # statistics as integer accumulators (csrc/bn_acc.h): no partials, no finalize launch
        acc = AccRef(cout, x.device) if stats else None
        st = in_bn.struct() if isinstance(in_bn, BnAccInput) else None
        if in_bn is not None and st is None:
            raise _C.BuctdHipError("conv3x3: the accumulator path takes its input BatchNorm as a BnAccInput")
        check(lib().buctd_conv3x3_bf16x6_acc(N, H, W, cin, cout, ptr(x), ptr(wp), ptr(residual), int(bool(relu)), ptr(y),
                                             C.c_void_p(acc.ptr) if acc is not None else None,
                                             C.byref(st) if st is not None else None,
                                             ptr(in_bn.gamma) if st is not None else None,
                                             ptr(in_bn.beta) if st is not None else None,
                                             int(bool(in_bn.relu)) if st is not None else 0, stream_ptr()), "conv3x3_bf16x6_acc")
        return (y, acc, ("acc",)) if stats else y
    if isinstance(in_bn, BnAccInput):
        raise _C.BuctdHipError("conv3x3: BnAccInput needs the bf16x6 accumulator path (no bias / eval scale)")
    if stats:
        def groups():
            ng, rpg = C.c_int(), C.c_int()
            check(_c3fn("_stats_groups")(N, H, W, cin, cout, C.byref(ng), C.byref(rpg)), "conv3x3 groups")
            return ng.value, rpg.value
        ngv, rpgv = _memo(("c3grp", _conv_math["mode"], N, H, W, cin, cout), groups)
        part = torch.empty((ngv, cout, 2), dtype=torch.float32, device=x.device)
        counts = torch.empty(ngv, dtype=torch.int32, device=x.device)
        info = (ngv, rpgv, counts)
    if in_bn is not None:
        mean, invstd, gamma, beta, in_relu = in_bn
        check(lib().buctd_conv3x3_bf16x6_bnin(N, H, W, cin, cout, ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift),
                                              ptr(residual), int(bool(relu)), ptr(y), ptr(part), ptr(counts), ptr(mean),
                                              ptr(invstd), ptr(gamma), ptr(beta), int(bool(in_relu)), stream_ptr()),
              "conv3x3_bf16x6_bnin")
        return (y, part, info) if stats else y
    check(_c3fn("")(N, H, W, cin, cout, ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift),
                    ptr(residual), int(bool(relu)), ptr(y), ptr(part), ptr(counts), stream_ptr()),
          "conv3x3 (split bf16)")
    return (y, part, info) if stats else y


# ---- gathered bf16x6 convolutions (csrc/conv_gather_x6.hip): 1x1 and stride-2 3x3, forward + data gradient ----------------
_GCONV_X6 = _SW["GCONV_X6"] != "0"
_GCONV_MASK = int(_SW["GCONV_MASK"])     # experiments: bit 0/1 = 1x1 forward / data gradient, 2/3 = stride-2 3x3


def _gconv_kind(d):
    """1: 1x1 / stride 1 / pad 0; 2: 3x3 / stride 2 / pad 1; 0: neither."""
    if d.R == 1 and d.S == 1 and d.stride == 1 and d.pad == 0:
        return 1
    if d.R == 3 and d.S == 3 and d.stride == 2 and d.pad == 1:
        return 2
    return 0


def _gconv_ok(d, direction):
    """True when this convolution (direction 0 forward, 1 data gradient) takes the gathered bf16x6 kernel."""
    if _conv_math["mode"] != "bf16x6" or not _GCONV_X6:
        return False
    kind = _gconv_kind(d)
    if kind and not (_GCONV_MASK >> (2 * (kind - 1) + direction)) & 1:
        return False
    return kind != 0 and _memo(("gcok", kind, d.N, d.H, d.W, d.Ci, d.Co, direction),
                               lambda: lib().buctd_gconv_x6_supported(kind, d.N, d.H, d.W, d.Ci, d.Co, direction) == 1)


def _gconv_prepared(w, kind, direction):
    """weight image of the gathered kernels, cached on the weight tensor until it changes; filters rewritten in place by
    the optimizer kernel are refreshed together by _prep_all (one launch per step)."""
    import weakref
    Co, Ci = _wshape(w)[0], _wshape(w)[1]
    key = (w.data_ptr(), w._version, _weights_epoch["n"])
    cache = getattr(w, "_buctd_gprep", None)
    if cache is not None and cache[0] != key and cache[0][:2] == key[:2] and _PREP_BATCH:
        _prep_all(w.device)           # same storage, new optimizer epoch: batch-refresh every registered image
    if cache is None or cache[0] != key or cache[1] != kind:
        first = cache is None
        cache = [key, kind, None, None]
        try:
            w._buctd_gprep = cache
            if first:                 # one registry entry per weight tensor, however often its cache key changes
                _gprep_registry["weights"].append(weakref.ref(w))
        except (AttributeError, RuntimeError, TypeError):
            pass
    if cache[2 + direction] is None:
        nbytes = _memo(("gcpb", kind, Ci, Co, direction), lambda: int(lib().buctd_gconv_x6_prep_bytes(kind, Ci, Co, direction)))
        img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        check(lib().buctd_gconv_x6_prep(kind, Ci, Co, ptr(w), direction, ptr(img), stream_ptr()), "gconv_x6_prep")
        cache[2 + direction] = img
    _prep_wait(w.device)
    return cache[2 + direction]


def _gconv_fwd(x, w, d, bias, scale, shift, residual, relu, stats):
    kind = _gconv_kind(d)
    wp = _gconv_prepared(w, kind, 0)
    y = torch.empty((d.N, d.Ho, d.Wo, d.Co), dtype=torch.float32, device=x.device)
    part = counts = info = None
    if stats == "acc" and scale is None and residual is None and not relu:
        acc = AccRef(d.Co, x.device)
        check(lib().buctd_gconv_x6_fwd_acc(kind, d.N, d.H, d.W, d.Ci, d.Co, ptr(x), ptr(wp), ptr(bias), ptr(y),
                                           C.c_void_p(acc.ptr), stream_ptr()), "gconv_x6_fwd_acc")
        return y, acc, ("acc",)
    if stats:
        def groups():
            ng, rpg = C.c_int(), C.c_int()
            check(lib().buctd_gconv_x6_stats_groups(kind, d.N, d.H, d.W, d.Ci, d.Co, C.byref(ng), C.byref(rpg)), "gconv groups")
            return ng.value, rpg.value
        ngv, rpgv = _memo(("gcgrp", kind, d.N, d.H, d.W, d.Ci, d.Co), groups)
        part = torch.empty((ngv, d.Co, 2), dtype=torch.float32, device=x.device)
        counts = torch.empty(ngv, dtype=torch.int32, device=x.device)
        info = (ngv, rpgv, counts)
    check(lib().buctd_gconv_x6_fwd(kind, d.N, d.H, d.W, d.Ci, d.Co, ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift),
                                   ptr(residual), int(bool(relu)), ptr(y), ptr(part), ptr(counts), stream_ptr()), "gconv_x6_fwd")
    return (y, part, info) if stats else y


_NATIVE_BLOCK = _SW["NATIVE_BLOCK"] == "1"
if _SW["C3_PERSISTENT"] == "1":      # experiments: the persistent form of the train-mode 3x3 launches (include/buctd_hip.h)
    lib().buctd_conv3x3_bf16x6_persistent(1)
_FUSED_BOTTLENECK = {"on": _SW["FUSED_BOTTLENECK"] == "1"}


def fused_bottleneck_on():
    """Bottlenecks of a training forward as one autograd node (BottleneckFn); off = three / four ConvBnAct nodes (tests)."""
    return _FUSED_BOTTLENECK["on"]


def set_fused_bottleneck(on):
    old = _FUSED_BOTTLENECK["on"]
    _FUSED_BOTTLENECK["on"] = bool(on)
    return old
# experiment switches, read ONCE at import (the hot path consults module constants, never the environment)
_FUSE_BN_IN = _SW["FUSE_BN_IN"] == "1"
_FUSE_BWD_BNSTAT = _SW["FUSE_BWD_BNSTAT"] == "1"
_FUSE_BWD_BNSTAT_S0 = _SW["FUSE_BWD_BNSTAT_S0"] == "1"
_FC_O_X6 = _SW["FC_O_X6"] != "0"
_MHA_X6 = _SW["MHA_X6"] != "0"
_MHA_PRESPLIT = _SW["MHA_PRESPLIT"] != "0"
_ATTN_X6 = _SW["ATTN_X6"] != "0"
# optional veto: callable(x_shape) -> True sends a BasicBlock through the step-by-step path (bench.py brackets every launch
# of its roofline shape with HIP events, which it can only do from the host mirror)
native_block_veto = {"fn": None}


def native_chain_ok(x_shape):
    """True when a chain of BasicBlocks on this activation shape may take the one-call-per-direction path."""
    return (_NATIVE_BLOCK and _conv_math["mode"] == "bf16x6" and native_block_veto["fn"] is None)


def bn_in_fusable(x_shape, w):
    """True when a 3x3/s1/p1 convolution of this shape can apply its producer's BatchNorm(+ReLU) while staging its
    input (bf16x6 kernel): conv_fwd(..., in_bn=...) and conv_wgrad(..., x_bn=...)."""
    if _conv_math["mode"] != "bf16x6" or not _FUSE_BN_IN:
        return False
    ws = _wshape(w)
    if ws[2] != 3 or ws[3] != 3:
        return False
    d = conv_desc(x_shape, ws, 1, 1)
    return _memo(("bnin", d.N, d.H, d.W, d.Ci, d.Co),
                 lambda: (lib().buctd_conv3x3_bf16x6_supported(d.N, d.H, d.W, d.Ci, d.Co) == 1 and
                          lib().buctd_conv3x3_wgrad_bf16x6_supported(d.N, d.H, d.W, d.Ci, d.Co) == 1))


def conv_fwd(x, w, bias=None, stride=1, pad=0, scale=None, shift=None, residual=None, relu=False, stats=False,
             in_bn=None):
    """in_bn = (mean, invstd, gamma, beta, relu) or a BnAccInput: x is the raw output of the producing convolution and its
    BatchNorm (+ReLU) is applied while the input is staged (only where bn_in_fusable() says so).
    stats: True -> (y, Welford partials, info) for bn_finalize; "acc" -> (y, AccRef, ("acc",)) where the kernel supports the
    accumulator form (csrc/bn_acc.h: no finalize launch), the partials form elsewhere."""
    _f32(x, "conv input")
    weight_rsc(w)
    d = conv_desc(x.shape, _wshape(w), stride, pad)
    if in_bn is not None and not (_conv_math["mode"] == "bf16x6" and _bf16x3_ok(d)):
        raise _C.BuctdHipError("conv_fwd: in_bn needs the bf16x6 3x3 kernel (check bn_in_fusable first)")
    if _bf16x3_ok(d):
        return _conv3x3_bf16x3(x, w, 0, d.Ci, d.Co, bias, scale, shift, residual, relu, stats, in_bn)
    if _gconv_ok(d, 0):
        return _gconv_fwd(x, w, d, bias, scale, shift, residual, relu, stats)
    y = torch.empty((d.N, d.Ho, d.Wo, d.Co), dtype=torch.float32, device=x.device)
    part = None
    info = None
    if stats and scale is None and residual is None and not relu and \
            _memo(("thin", d.N, d.H, d.W, d.Ci, d.Co, d.R, d.S, d.stride, d.pad), lambda: lib().buctd_conv2d_fwd_thin(C.byref(d)) == 1):
        # <= 4 output channels at full resolution (preNet 7x7): the thin kernel has no fused epilogue; the BatchNorm
        # partials of its 3-channel result are one cheap extra pass
        check(lib().buctd_conv2d_fwd(C.byref(d), ptr(x), ptr(w), ptr(bias), None, None, None, 0, ptr(y), None, stream_ptr()),
              "conv2d_fwd")
        part, info = bn_stats(y)
        return y, part, info
    if stats:
        def groups():
            ng_, rpg_ = C.c_int(), C.c_int()
            check(lib().buctd_conv2d_stats_groups(C.byref(d), 0, C.byref(ng_), C.byref(rpg_)), "conv2d_stats_groups")
            return ng_, rpg_
        ng, rpg = _memo(("c2grp", 0, d.N, d.H, d.W, d.Ci, d.Co, d.R, d.S, d.stride, d.pad), groups)
        part = torch.empty((ng.value, d.Co, 2), dtype=torch.float32, device=x.device)
        info = (ng.value, rpg.value)
    check(lib().buctd_conv2d_fwd(C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(scale), ptr(shift), ptr(residual),
                                 int(bool(relu)), ptr(y), ptr(part), stream_ptr()), "conv2d_fwd")
    return (y, part, info) if stats else y


def conv_dgrad(dy, w, x_shape, stride=1, pad=0, bias=None, stats=False, residual=None):
    """dx of a convolution == forward of a transposed convolution.  residual (not with stats): added to dx - the
    gradient arriving through a skip connection - in the kernel epilogue where the kernel supports it."""
    _f32(dy, "conv dgrad input")
    weight_rsc(w)
    d = conv_desc(x_shape, _wshape(w), stride, pad)
    if tuple(dy.shape) != (d.N, d.Ho, d.Wo, d.Co):
        raise _C.BuctdHipError(f"conv_dgrad: dy shape {tuple(dy.shape)} != {(d.N, d.Ho, d.Wo, d.Co)}")
    if _bf16x3_ok(d) and _memo(("c3ok", _conv_math["mode"], d.N, d.H, d.W, d.Co, d.Ci),
                               lambda: _c3fn("_supported")(d.N, d.H, d.W, d.Co, d.Ci) == 1):
        return _conv3x3_bf16x3(dy, w, 1, d.Co, d.Ci, bias, None, None, residual, False, stats)
    if residual is not None and stats:
        raise _C.BuctdHipError("conv_dgrad: residual and stats do not combine")
    if bias is None and not stats and _gconv_ok(d, 1):
        dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
        kind = _gconv_kind(d)
        check(lib().buctd_gconv_x6_dgrad(kind, d.N, d.H, d.W, d.Ci, d.Co, ptr(dy), ptr(_gconv_prepared(w, kind, 1)),
                                         ptr(residual), ptr(dx), stream_ptr()), "gconv_x6_dgrad")
        return dx
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    part = None
    info = None
    if stats:
        def groups():
            ng_, rpg_ = C.c_int(), C.c_int()
            check(lib().buctd_conv2d_stats_groups(C.byref(d), 1, C.byref(ng_), C.byref(rpg_)), "conv2d_stats_groups")
            return ng_, rpg_
        ng, rpg = _memo(("c2grp", 1, d.N, d.H, d.W, d.Ci, d.Co, d.R, d.S, d.stride, d.pad), groups)
        part = torch.empty((ng.value, d.Ci, 2), dtype=torch.float32, device=dy.device)
        info = (ng.value, rpg.value)
    check(lib().buctd_conv2d_dgrad(C.byref(d), ptr(dy), ptr(w), ptr(bias), ptr(dx), ptr(part), stream_ptr()),
          "conv2d_dgrad")
    if residual is not None:
        add(dx, residual, out=dx)
    return (dx, part, info) if stats else dx


_GCONV_WGRAD = {"on": True}      # 1x1 / stride-2 weight gradients on the bf16x6 kernel (off: the exact-fp32 MFMA kernel; tests)


def conv_wgrad(x, dy, w_like, stride=1, pad=0, out=None, accumulate=0, x_bn=None, stream=None):
    """x_bn = (mean, invstd, gamma, beta, relu): like conv_fwd's in_bn, for the X operand of the weight gradient.
    stream: a torch.cuda.Stream to launch on WITHOUT making it current (entering a stream context costs ~15 us of host time
    per call - more than the launch); None = the current stream."""
    if stream is None:
        workspace_ = workspace
        stream_ptr_ = stream_ptr
    else:
        workspace_ = lambda nbytes, device: workspace_on(stream, nbytes, device)
        stream_ptr_ = lambda: stream.cuda_stream
    _f32(x, "conv wgrad x")
    _f32(dy, "conv wgrad dy")
    d = conv_desc(x.shape, _wshape(w_like), stride, pad)
    if out is None:
        out = _new_like_weight(w_like)
        accumulate = 0
    weight_rsc(out)
    mode = _conv_math["mode"]
    if mode != "fp32" and d.R == 3 and d.S == 3 and d.stride == 1 and d.pad == 1:
        fn = getattr(lib(), "buctd_conv3x3_wgrad_" + mode)
        need = _memo(("wg3", mode, d.N, d.H, d.W, d.Ci, d.Co),
                     lambda: (getattr(lib(), "buctd_conv3x3_wgrad_" + mode + "_workspace")(d.N, d.H, d.W, d.Ci, d.Co)
                              if getattr(lib(), "buctd_conv3x3_wgrad_" + mode + "_supported")(d.N, d.H, d.W, d.Ci, d.Co) == 1
                              else -1))
        if need >= 0:
            ws = workspace_(need, x.device)
            if x_bn is not None:
                if mode != "bf16x6":
                    raise _C.BuctdHipError("conv_wgrad: x_bn needs the bf16x6 kernel")
                mean, invstd, gamma, beta, x_relu = x_bn
                check(lib().buctd_conv3x3_wgrad_bf16x6_bnin(d.N, d.H, d.W, d.Ci, d.Co, ptr(x), ptr(dy), ptr(out),
                                                            int(accumulate), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
                                                            int(bool(x_relu)), ptr(ws), ws.numel(), stream_ptr_()),
                      "conv3x3_wgrad_bf16x6_bnin")
                return out
            check(fn(d.N, d.H, d.W, d.Ci, d.Co, ptr(x), ptr(dy), ptr(out), int(accumulate), ptr(ws), ws.numel(),
                     stream_ptr_()), "conv3x3_wgrad (split bf16)")
            return out
    if x_bn is not None:
        raise _C.BuctdHipError("conv_wgrad: x_bn needs the bf16x6 3x3 kernel (check bn_in_fusable first)")
    kind = _gconv_kind(d) if mode == "bf16x6" and _GCONV_WGRAD["on"] else 0
    if kind:
        need = _memo(("gwg", kind, d.N, d.H, d.W, d.Ci, d.Co),
                     lambda: (int(lib().buctd_gconv_wgrad_x6_workspace(kind, d.N, d.H, d.W, d.Ci, d.Co))
                              if lib().buctd_gconv_wgrad_x6_supported(kind, d.N, d.H, d.W, d.Ci, d.Co) == 1 else -1))
        if need >= 0:
            ws = workspace_(need, x.device)
            check(lib().buctd_gconv_wgrad_x6(kind, d.N, d.H, d.W, d.Ci, d.Co, ptr(x), ptr(dy), ptr(out), int(accumulate), ptr(ws),
                                             ws.numel(), stream_ptr_()), "gconv_wgrad_x6")
            return out
    need = lib().buctd_conv2d_wgrad_workspace(C.byref(d))
    ws = workspace_(need, x.device)
    check(lib().buctd_conv2d_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(out), int(accumulate), ptr(ws), ws.numel(),
                                   stream_ptr_()), "conv2d_wgrad")
    return out


# Weight gradients are off the critical path of the backward pass (nothing downstream of dgrad needs them), so they
# run on a second HIP stream and fill the CUs that the latency-bound BN / dgrad chain leaves idle.  The join is queued
# as an autograd end-of-backward callback, so param.grad is complete on the caller's stream when backward() returns.
_side = {"on": _SW["WGRAD_STREAM"] == "1", "streams": {}, "joined": True, "rr": 0,
         "n": max(1, int(_SW["WGRAD_STREAMS"]))}   # more than one measured slower (L2 contention)


# The weight-gradient stream runs at HIGH HIP priority: the critical path of the backward pass runs along it for half of the
# time (profiles/r04_critical_path.txt: 12.9 ms of weight gradients + 3.5 ms of their slab reductions on the chain), so its
# kernels should get free workgroup slots before the main stream's: 454.0 -> 457.2 img/s (interleaved A/B on one box, round
# 4; the branch streams at high priority cost 1 %).
_SIDE_PRIO = int(_SW["WGRAD_PRIO"])


def _side_stream(device):
    """Weight-gradient streams, used round-robin (consecutive layers' gradients are independent of each other)."""
    _side["rr"] = (_side["rr"] + 1) % _side["n"]
    key = (device.index, _side["rr"])
    st = _side["streams"].get(key)
    if st is None:
        st = torch.cuda.Stream(device=device, priority=_SIDE_PRIO)
        _side["streams"][key] = st
    return st


def copy_stream(device):
    """The stream host-to-device copies of the NEXT batch go to (core.function.DevicePrefetch): the weight-gradient stream -
    idle during the forward pass, and no additional HIP stream.  None where the engine forks no such stream."""
    if not (_side["on"] and device.type == "cuda"):
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(), 0)
    st = _side["streams"].get(key)
    if st is None:
        st = torch.cuda.Stream(device=device, priority=_SIDE_PRIO)
        _side["streams"][key] = st
    return st


def side_streams(device):
    """The weight-gradient streams created so far on `device`."""
    return [st for (idx, _), st in _side["streams"].items() if idx == device.index]


def wait_side_stream(stream=None):
    """Make `stream` (default: the current one) wait for everything enqueued so far on the streams this module owns
    (weight-gradient stream and branch streams)."""
    for (idx, _), st in _side["streams"].items():
        target = stream if stream is not None else torch.cuda.current_stream(torch.device("cuda", idx))
        target.wait_stream(st)
    for (idx, _), st in _branch["streams"].items():
        target = stream if stream is not None else torch.cuda.current_stream(torch.device("cuda", idx))
        if target != st:
            target.wait_stream(st)


# Independent sub-graphs (the parallel resolution branches of an HRNet module, the rows of its fuse stage) are
# enqueued on separate HIP streams: the low-resolution branches are launch/latency bound and hide under the
# bandwidth-bound high-resolution one.  Autograd replays every backward node on the stream of its forward op, so the
# backward pass inherits the same concurrency.
_branch = {"on": _SW["BRANCH_STREAMS"] == "1", "streams": {}}


# main + 2 branch streams + the weight-gradient stream = the 4 HIP hardware queues: no two streams share a queue by
# accident (HRNet branches 2 and 3, the cheapest, share the last stream): 453 -> 468 img/s
_BRANCH_MAX = int(_SW["BRANCH_MAX"])


def set_stream_forks(branch, wgrad=None):
    """Turn the engine's stream forks on / off (branch streams of fork_join, the weight-gradient stream); returns the previous
    pair.  engine.StepGraph captures a LINEAR graph with both off: on ROCm 7.2 a replay pays ~70 us of host time per
    cross-queue edge of the graph."""
    old = (_branch["on"], _side["on"])
    _branch["on"] = bool(branch)
    _side["on"] = bool(branch if wgrad is None else wgrad)
    return old


def set_branch_max(n):
    """Cap on the number of branch streams (besides the main stream).  The default runtime serves four hardware queues and any
    fifth HIP stream costs 25-32 %: engine.DataParallel lowers the cap to 1 when it adds its communication
    stream (main + 1 branch + weight-gradient + communication = 4).  Returns the previous cap."""
    global _BRANCH_MAX
    old, _BRANCH_MAX = _BRANCH_MAX, max(0, int(n))
    return old


def compute_streams(device):
    """The HIP streams this module launches on besides the current one on `device` (branch + weight-gradient streams)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    # branch streams above the current cap were created before it was lowered and are no longer handed out
    return ([st for (d, i), st in _branch["streams"].items() if d == idx and i <= _BRANCH_MAX] +
            [st for (d, _), st in _side["streams"].items() if d == idx])


def reserve_compute_streams(device):
    """Create the branch streams (up to the cap) and the weight-gradient stream(s) of `device` now; returns them.  See
    engine.reserve_streams: streams are bound to hardware queues in the order of their first use."""
    out = []
    if _branch["on"]:
        out += [_branch_stream(device, i) for i in range(1, _BRANCH_MAX + 1)]
    if _side["on"]:
        for rr in range(_side["n"]):
            key = (device.index, rr)
            if key not in _side["streams"]:
                _side["streams"][key] = torch.cuda.Stream(device=device, priority=_SIDE_PRIO)
            out.append(_side["streams"][key])
    return out


def _branch_stream(device, i):
    i = min(i, _BRANCH_MAX)     # branches beyond the cap share the last branch stream
    key = (device.index, i)
    st = _branch["streams"].get(key)
    if st is None:
        st = torch.cuda.Stream(device=device, priority=int(_SW["BRANCH_PRIO"]))
        _branch["streams"][key] = st
    return st


def lane_stream(device, k):
    """Branch stream k + 1 of `device` (engine.ForwardGraph.submit replays its k-th lane there)."""
    return _branch_stream(device, k + 1)


def _record(t, stream):
    if torch.is_tensor(t):
        if t.is_cuda:
            t.record_stream(stream)
    elif isinstance(t, (list, tuple)):
        for u in t:
            _record(u, stream)


def fork_join(fns, inputs, tag=0):
    """Run fns[i]() concurrently: fns[0] on the current stream, the others on branch streams that first wait for the
    work enqueued so far; returns their results once the current stream has been made to wait for all of them.
    inputs[i]: the tensors fns[i] reads (so that the allocator knows they are in use on that stream)."""
    dev = None
    for t in inputs:
        for u in (t if isinstance(t, (list, tuple)) else [t]):
            if torch.is_tensor(u) and u.is_cuda:
                dev = u.device
    if not _branch["on"] or dev is None or len(fns) < 2:
        return [f() for f in fns]
    main = torch.cuda.current_stream(dev)
    start = torch.cuda.Event()
    start.record(main)
    outs = [None] * len(fns)
    done = []
    outs[0] = fns[0]()      # same op creation order as the serial path: autograd then accumulates in the same order
    for i in range(1, len(fns)):
        st = _branch_stream(dev, i)
        if st == main:          # nested fork on a branch stream: run inline
            outs[i] = fns[i]()
            continue
        st.wait_event(start)
        _record(inputs[i], st)
        with torch.cuda.stream(st):
            outs[i] = fns[i]()
            ev = torch.cuda.Event()
            ev.record(st)
        _record(outs[i], main)
        done.append(ev)
    for ev in done:
        main.wait_event(ev)
    return outs


def _join_side():
    _side["joined"] = True
    _bwd_sums.clear()
    wait_side_stream()


def _queue_join():
    """Queue the end-of-backward join once per backward pass.  Keyed on the autograd graph-task id, so a backward that
    died with an exception (its callback never ran) cannot leave the next one without a join."""
    task = torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else None
    if task is not None and task >= 0:
        if _side.get("task") == task:
            return
        _side["task"] = task
    elif not _side["joined"]:
        return
    _side["joined"] = False
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_join_side)
    except RuntimeError:      # not inside a backward pass: join right away
        _join_side()


def conv_wgrad_async(x, dy, w_like, stride, pad, out, accumulate, x_bn=None):
    """conv_wgrad on the side stream (backward passes only: the join rides on the autograd engine's final callback)."""
    if not (_side["on"] and x.is_cuda):
        if x.is_cuda and _branch["on"]:
            _queue_join()     # parameter gradients may be written on branch streams: still join them at the end
        return conv_wgrad(x, dy, w_like, stride, pad, out=out, accumulate=accumulate, x_bn=x_bn)
    side = _side_stream(x.device)
    # fork: the side stream waits for what the current stream has enqueued so far (one cached event in the library: creating
    # an Event object and entering a stream context per call cost ~25 us of host time - the step of HRNet-W32 is host-bound)
    check(lib().buctd_stream_fork(stream_ptr(), side.cuda_stream), "stream_fork")
    conv_wgrad(x, dy, w_like, stride, pad, out=out, accumulate=accumulate, x_bn=x_bn, stream=side)
    x.record_stream(side)
    dy.record_stream(side)
    if x_bn is not None:
        x_bn[0].record_stream(side)
        x_bn[1].record_stream(side)
    _queue_join()
    return out


def matmul(A, B, Cout, *, batch, M, N, K, a_layout, b_layout, lda, ldb, ldc, stride_a=0, stride_b=0, stride_c=0,
           Kc=None, gsa=0, gsbk=0, Nc=None, gsbn=0, gsc=0, alpha=1.0, bias=None, bias_axis=0,
           a_off=0, b_off=0, c_off=0):
    d = MatmulDesc(batch, M, N, K, a_layout, b_layout, lda, ldb, ldc, stride_a, stride_b, stride_c,
                   K if Kc is None else Kc, gsa, gsbk, N if Nc is None else Nc, gsbn, gsc, alpha, bias_axis)
    need = lib().buctd_matmul_workspace(C.byref(d))
    ws = workspace(need, A.device) if need else None
    pa = C.c_void_p(A.data_ptr() + 4 * a_off)
    pb = C.c_void_p(B.data_ptr() + 4 * b_off)
    pc = C.c_void_p(Cout.data_ptr() + 4 * c_off)
    check(lib().buctd_matmul(C.byref(d), pa, pb, ptr(bias), pc, ptr(ws), ws.numel() if ws is not None else 0,
                             stream_ptr()), "matmul")
    return Cout


def x6_image(src, V, K, role, *, vs, ks, vg=0, vgs=0, kg=0, kgs=0, off=0, out=None):
    """Prepared bf16x6 image of the logical matrix X[v][k] = src.flat[off + (v // vg) * vgs + (v % vg) * vs +
    (k // kg) * kgs + (k % kg) * ks] (vg / kg = 0: plain strides); role 0: A operand (rows of C), 1: B operand."""
    need = lib().buctd_x6_image_bytes(V, K, role)
    if out is None or out.numel() < need:
        out = torch.empty(need, dtype=torch.uint8, device=src.device)
    check(lib().buctd_x6_image(C.c_void_p(src.data_ptr() + 4 * off), V, K, vg, vgs, vs, kg, kgs, ks, role, ptr(out),
                               stream_ptr()), "x6_image")
    return out


def x6_gemm(a_image, b_image, Cout, M, N, K, *, ldc, Nc=0, gsc=0, bias=None, bias_axis=0, alpha=1.0, c_off=0):
    """Cout(m, n) = alpha * sum_k A(m, k) B(k, n) (+ bias) from two prepared images (bf16x6 arithmetic)."""
    check(lib().buctd_x6_gemm(M, N, K, ptr(a_image), ptr(b_image), ptr(bias), bias_axis, alpha,
                              C.c_void_p(Cout.data_ptr() + 4 * c_off), ldc, Nc, gsc, stream_ptr()), "x6_gemm")
    return Cout


def _x6_weight_image(w, V, K, transposed):
    """bf16x6 image of a Linear weight w [V or K rows] as the A operand, cached on the tensor until it changes
    (version / optimizer epoch).  transposed: the logical matrix is w^T (X[v][k] = w[k][v])."""
    key = (w.data_ptr(), w._version, _weights_epoch["n"])
    cache = getattr(w, "_buctd_x6img", None)
    if cache is None or cache["key"] != key:
        cache = {"key": key}
        try:
            w._buctd_x6img = cache
        except (AttributeError, RuntimeError, TypeError):
            pass
    tag = "t" if transposed else "n"
    if tag not in cache:
        ld = w.shape[1]
        cache[tag] = x6_image(w, V, K, 0, vs=1, ks=ld) if transposed else x6_image(w, V, K, 0, vs=ld, ks=1)
        cache[tag + "_ev"] = _new_event()
        cache[tag + "_stream"] = torch.cuda.current_stream(w.device).cuda_stream
    elif torch.cuda.current_stream(w.device).cuda_stream != cache[tag + "_stream"]:
        _wait_event(torch.cuda.current_stream(w.device), cache[tag + "_ev"])
    return cache[tag]


def fc_o_x6_ok(T, rows):
    """the bf16x6 GEMM pays once the product is large (its images are padded to 128 x 192 x 128 tiles)"""
    return _conv_math["mode"] == "bf16x6" and T >= 512 and rows >= 192 and _FC_O_X6


def bn_finalize(part, info, rows, Cn, eps, momentum, running_mean, running_var):
    """info = (ngroups, rows_per_group[, per-group valid-row counts tensor])."""
    mean = torch.empty(Cn, dtype=torch.float32, device=part.device)
    invstd = torch.empty(Cn, dtype=torch.float32, device=part.device)
    counts = info[2] if len(info) > 2 else None
    check(lib().buctd_bn_finalize(ptr(part), ptr(counts), info[0], info[1], rows, Cn, eps, momentum, ptr(mean), ptr(invstd),
                                  ptr(running_mean), ptr(running_var), stream_ptr()), "bn_finalize")
    return mean, invstd


class BnAccInput:
    """The BatchNorm(+ReLU) of a producing layer whose statistics are still in its accumulator: the consumer derives
    mean / invstd itself and its first workgroup writes them to `mean` / `invstd` (fresh [Cn] tensors kept for the backward
    pass) and updates the running statistics.  Passed as conv_fwd(in_bn=...)."""

    def __init__(self, acc, rows, bn, training=True, relu=True):
        Cn = acc.Cn
        dev = bn.weight.device
        self.acc, self.rows, self.gamma, self.beta, self.relu = acc, int(rows), bn.weight, bn.bias, relu
        self.eps = bn.eps
        self.momentum = 0.1 if bn.momentum is None else bn.momentum
        stat = torch.empty((2, Cn), dtype=torch.float32, device=dev)
        self.mean, self.invstd = stat[0], stat[1]
        track = bn.track_running_stats and training
        self.rm = bn.running_mean if track else None
        self.rv = bn.running_var if track else None

    def struct(self):
        st = _C.BnAccIn()
        st.acc, st.rows, st.eps, st.momentum = self.acc.ptr, self.rows, self.eps, self.momentum
        st.mean_out, st.invstd_out = self.mean.data_ptr(), self.invstd.data_ptr()
        st.running_mean = self.rm.data_ptr() if self.rm is not None else None
        st.running_var = self.rv.data_ptr() if self.rv is not None else None
        return st


def bn_apply_acc(z, bnin, residual=None, relu=False):
    """y = act(bn(z) (+ residual)) with the statistics taken from bnin.acc; bnin.mean / bnin.invstd are filled on the way."""
    Cn = z.shape[-1]
    y = torch.empty_like(z)
    st = bnin.struct()
    check(lib().buctd_bn_apply_acc(ptr(z), C.byref(st), ptr(bnin.gamma), ptr(bnin.beta), ptr(residual), int(bool(relu)), ptr(y),
                                   z.numel() // Cn, Cn, stream_ptr()), "bn_apply_acc")
    return y


def bn_acc_ok(Cn):
    return Cn % 4 == 0 and Cn <= 1024


def bn_stats(z):
    Cn = z.shape[-1]
    rows = z.numel() // Cn
    ng, rpg = C.c_int(), C.c_int()
    check(lib().buctd_bn_stats_groups(rows, Cn, C.byref(ng), C.byref(rpg)), "bn_stats_groups")
    part = torch.empty((ng.value, Cn, 2), dtype=torch.float32, device=z.device)
    check(lib().buctd_bn_stats(ptr(z), rows, Cn, ptr(part), None, None, stream_ptr()), "bn_stats")
    return part, (ng.value, rpg.value)


def bn_apply(z, mean, invstd, gamma, beta, residual=None, relu=False):
    Cn = z.shape[-1]
    y = torch.empty_like(z)
    check(lib().buctd_bn_apply(ptr(z), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(residual), int(bool(relu)),
                               ptr(y), z.numel() // Cn, Cn, stream_ptr()), "bn_apply")
    return y


def bn_bwd(dy, y, z, mean, invstd, gamma, relu, want_dres, dgamma, dbeta, accumulate, beta=None, acc=None, acc_ready=False):
    """y=None with relu: the ReLU mask is rebuilt from z, gamma and beta (forward without residual only).
    acc / acc_ready: an accumulator that already holds the backward sums (formed by the data gradient that produced dy)."""
    Cn = z.shape[-1]
    rows = z.numel() // Cn
    dz = torch.empty_like(z)
    dres = torch.empty_like(z) if want_dres else None
    if relu and y is None and beta is None:
        raise _C.BuctdHipError("bn_bwd: ReLU backward needs the forward output or beta")
    if bn_acc_ok(Cn):
        # reduction into an integer accumulator, decoded by the apply kernel: two launches, no finalize in between
        acc = acc if acc is not None else AccRef(Cn, z.device)
        check(lib().buctd_bn_bwd_acc(ptr(dy), ptr(y) if relu else None, ptr(z), ptr(mean), ptr(invstd), ptr(gamma),
                                     ptr(beta) if (relu and y is None) else None, int(bool(relu)), rows, Cn, ptr(dz), ptr(dres),
                                     ptr(dgamma), ptr(dbeta), int(accumulate), C.c_void_p(acc.ptr), int(bool(acc_ready)),
                                     stream_ptr()), "bn_bwd_acc")
        return dz, dres
    need = lib().buctd_bn_bwd_workspace(rows, Cn)
    ws = workspace(need, z.device)
    check(lib().buctd_bn_bwd(ptr(dy), ptr(y) if relu else None, ptr(z), ptr(mean), ptr(invstd), ptr(gamma),
                             ptr(beta) if (relu and y is None) else None, int(bool(relu)), rows, Cn, ptr(dz), ptr(dres), ptr(dgamma), ptr(dbeta), int(accumulate),
                             ptr(ws), ws.numel(), stream_ptr()), "bn_bwd")
    return dz, dres


def bn_fold_cached(bn, gamma, beta, eps):
    """eval-mode scale / shift of a BatchNorm module, folded once and kept on the module until a parameter, a running
    statistic (tensor versions) or the optimizer epoch changes - inference runs the same 146 folds on every forward
    otherwise (4.4 us each: 0.64 ms of a 37 ms TransPose step)."""
    rm, rv = bn.running_mean, bn.running_var
    # the train path updates the running statistics through raw pointers (no version bump): it counts batches on the
    # module (nn.BatchNorm2d.count_batch) or bumps num_batches_tracked - both are part of the key
    nbt = bn.num_batches_tracked
    key = (gamma.data_ptr(), gamma._version, beta._version, rm.data_ptr(), rm._version, rv._version, _weights_epoch["n"],
           float(eps), getattr(bn, "_pending_batches", 0), nbt._version if nbt is not None else 0)
    cache = getattr(bn, "_buctd_fold", None)
    cur = torch.cuda.current_stream(gamma.device)
    if cache is None or cache[0] != key:
        scale, shift = bn_fold(gamma, beta, rm, rv, eps)
        ev = _new_event()
        cache = (key, scale, shift, ev, cur.cuda_stream, set())
        try:
            object.__setattr__(bn, "_buctd_fold", cache)
        except (AttributeError, TypeError):
            return scale, shift
    elif cur.cuda_stream != cache[4] and cur.cuda_stream not in cache[5]:
        _wait_event(cur, cache[3])            # folded on another stream: order this stream behind it, once
        cache[5].add(cur.cuda_stream)
    return cache[1], cache[2]


def bn_fold(gamma, beta, rm, rv, eps):
    Cn = gamma.numel()
    scale = torch.empty(Cn, dtype=torch.float32, device=gamma.device)
    shift = torch.empty(Cn, dtype=torch.float32, device=gamma.device)
    check(lib().buctd_bn_fold(ptr(gamma), ptr(beta), ptr(rm), ptr(rv), eps, Cn, ptr(scale), ptr(shift), stream_ptr()),
          "bn_fold")
    return scale, shift


def add(a, b=None, relu=False, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(lib().buctd_add(ptr(a), ptr(b), ptr(out), a.numel(), int(bool(relu)), stream_ptr()), "add")
    return out


def mul(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(lib().buctd_mul(ptr(a), ptr(b), ptr(out), a.numel(), stream_ptr()), "mul")
    return out


class Mul(torch.autograd.Function):
    """out = a * b, DAModule's channel-only gate `input * c_out` (pose_hrnet_coam.py:716-717)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _contig(a), _contig(b)
        ctx.save_for_backward(a, b)
        return mul(a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = _contig(dy)
        return mul(dy, b), mul(dy, a)


def scale(x, dev_scalar=None, alpha=1.0, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(lib().buctd_scale(ptr(x), ptr(dev_scalar), alpha, ptr(out), x.numel(), stream_ptr()), "scale")
    return out


def relu_bwd(dy, y):
    dx = torch.empty_like(dy)
    check(lib().buctd_relu_bwd(ptr(dy), ptr(y), ptr(dx), dy.numel(), stream_ptr()), "relu_bwd")
    return dx


def colsum(x2d_like, Cn, out, accumulate):
    rows = x2d_like.numel() // Cn
    need = lib().buctd_colsum_workspace(rows, Cn)
    ws = workspace(need, x2d_like.device)
    check(lib().buctd_colsum(ptr(x2d_like), rows, Cn, ptr(out), int(accumulate), ptr(ws), ws.numel(), stream_ptr()),
          "colsum")
    return out


def nchw_to_nhwc(x, c0=0, cc=None):
    _f32(x, "nchw input")
    N, Ct, H, W = x.shape
    cc = Ct - c0 if cc is None else cc
    y = torch.empty((N, H, W, cc), dtype=torch.float32, device=x.device)
    check(lib().buctd_nchw_to_nhwc(ptr(x), N, Ct, c0, cc, H, W, ptr(y), stream_ptr()), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x):
    _f32(x, "nhwc input")
    N, H, W, Cn = x.shape
    y = torch.empty((N, Cn, H, W), dtype=torch.float32, device=x.device)
    check(lib().buctd_nhwc_to_nchw(ptr(x), N, Cn, H, W, ptr(y), stream_ptr()), "nhwc_to_nchw")
    return y


def fuse_sum(terms, shifts, relu=True):
    """terms[j]: [N, H>>s_j, W>>s_j, C]; output resolution = that of the shift-0 term."""
    base = terms[shifts.index(0)]
    N, H, W, Cn = base.shape
    out = torch.empty_like(base)
    n = len(terms)
    arr = (C.c_void_p * n)(*[t.data_ptr() for t in terms])
    sh = (C.c_int * n)(*shifts)
    check(lib().buctd_fuse_sum(arr, sh, n, N, H, W, Cn, int(bool(relu)), ptr(out), stream_ptr()), "fuse_sum")
    return out


def fuse_sum_bwd(dy, y, shift):
    N, H, W, Cn = dy.shape
    g = torch.empty((N, H >> shift, W >> shift, Cn), dtype=torch.float32, device=dy.device)
    check(lib().buctd_fuse_sum_bwd(ptr(dy), ptr(y), shift, N, H, W, Cn, ptr(g), stream_ptr()), "fuse_sum_bwd")
    return g


# (gradient address, z address) -> AccRef holding the BatchNorm-backward sums of that pair, formed by the kernel that wrote the
# gradient (fuse_sum_bwd_bnstat); ConvBnAct.backward pops its entry.  Emptied at the end of every backward pass (_join_side).
_bwd_sums = {}


def fuse_sum_bwd_bnstat(dy, y, shift, bns):
    """fuse_sum_bwd + the BatchNorm-backward sums of up to three conv -> BatchNorm terms of this shift (bns: (z, mean, invstd)
    each) in one pass; the accumulators are left in _bwd_sums for the terms' ConvBnAct.backward."""
    N, H, W, Cn = dy.shape
    g = torch.empty((N, H >> shift, W >> shift, Cn), dtype=torch.float32, device=dy.device)
    n = len(bns)
    accs = [AccRef(Cn, dy.device) for _ in bns]
    zs = (C.c_void_p * n)(*[b[0].data_ptr() for b in bns])
    ms = (C.c_void_p * n)(*[b[1].data_ptr() for b in bns])
    iv = (C.c_void_p * n)(*[b[2].data_ptr() for b in bns])
    ac = (C.c_void_p * n)(*[a.ptr for a in accs])
    check(lib().buctd_fuse_sum_bwd_bnstat(ptr(dy), ptr(y), shift, N, H, W, Cn, ptr(g), n, zs, ms, iv, ac, stream_ptr()),
          "fuse_sum_bwd_bnstat")
    for b, a in zip(bns, accs):
        _bwd_sums[(g.data_ptr(), b[0].data_ptr())] = a
    _queue_join()       # the registry is emptied with the backward pass
    return g


def resize_bilinear_from_nchw(x, c0, cc, Ho, Wo):
    _f32(x, "resize input")
    N, Ct, H, W = x.shape
    y = torch.empty((N, Ho, Wo, cc), dtype=torch.float32, device=x.device)
    check(lib().buctd_resize_bilinear(ptr(x), N, Ct, c0, cc, H, W, Ho, Wo, ptr(y), stream_ptr()), "resize_bilinear")
    return y


def softmax_dropout_fwd(s, L, scale, p_drop, seed, inplace=True):
    rows = s.numel() // L
    p = s if inplace else torch.empty_like(s)
    pd = torch.empty_like(s) if p_drop > 0 else p
    check(lib().buctd_softmax_dropout_fwd(ptr(s), rows, L, scale, p_drop, seed, ptr(p), ptr(pd), stream_ptr()),
          "softmax_dropout_fwd")
    return p, pd


def softmax_dropout_bwd(dpd, p, L, scale, p_drop, seed, inplace=True):
    rows = p.numel() // L
    ds = dpd if inplace else torch.empty_like(dpd)
    check(lib().buctd_softmax_dropout_bwd(ptr(dpd), ptr(p), rows, L, scale, p_drop, seed, ptr(ds), stream_ptr()),
          "softmax_dropout_bwd")
    return ds


def joints_mse(pred, gt, w, want_grad, gscale=1.0):
    N, K = pred.shape[0], pred.shape[1]
    HW = pred.numel() // (N * K)
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    ws = workspace(N * K * 4, pred.device)
    check(lib().buctd_joints_mse(ptr(pred), ptr(gt), ptr(w), N, K, HW, ptr(loss), ptr(grad), gscale, ptr(ws),
                                 ws.numel(), stream_ptr()), "joints_mse")
    return loss, grad


def argmax_decode(hm, refine=False, preds_out=None):
    """refine: also return the quarter-pixel offsets of get_final_preds' POST_PROCESS step ([N,K,2]).
    preds_out: a pinned host tensor [N,K,2] the kernel writes the coordinates into directly (the GPU maps pinned memory)."""
    N, K, H, W = hm.shape
    preds = preds_out if preds_out is not None else torch.empty((N, K, 2), dtype=torch.float32, device=hm.device)
    if preds_out is not None and not (preds_out.is_pinned() and preds_out.is_contiguous() and tuple(preds_out.shape) == (N, K, 2)
                                      and preds_out.dtype == torch.float32):
        raise _C.BuctdHipError("argmax_decode: preds_out must be a pinned contiguous fp32 [N, K, 2] host tensor")
    maxvals = torch.empty((N, K, 1), dtype=torch.float32, device=hm.device)
    idx = torch.empty((N, K), dtype=torch.int32, device=hm.device)
    if refine:
        quarter = torch.empty((N, K, 2), dtype=torch.float32, device=hm.device)
        check(lib().buctd_argmax_decode_refined(ptr(hm), N * K, H, W, preds.data_ptr(), ptr(maxvals), ptr(idx), ptr(quarter),
                                                stream_ptr()), "argmax_decode_refined")
        return preds, maxvals, idx, quarter
    check(lib().buctd_argmax_decode(ptr(hm), N * K, H, W, preds.data_ptr(), ptr(maxvals), ptr(idx), stream_ptr()),
          "argmax_decode")
    return preds, maxvals, idx


def scalar_to_host(src, dst):
    """One float from device memory into a pinned host tensor, written by a kernel on the current stream (no copy engine)."""
    if not (dst.is_pinned() and dst.dtype == torch.float32 and dst.numel() == 1 and src.numel() == 1):
        raise _C.BuctdHipError("scalar_to_host: one fp32 value into a pinned host tensor")
    check(lib().buctd_copy_channels(ptr(src), 1, 1, 0, dst.data_ptr(), 1, 0, 1, stream_ptr()), "scalar_to_host")
    return dst


def gaussian_target(joints, vis, heatmap_size, image_size, sigma):
    """joints [B,K,3] (crop pixels), vis [B,K]; sizes are (W, H) like the reference cfg."""
    B, K = joints.shape[0], joints.shape[1]
    Wh, Hh = int(heatmap_size[0]), int(heatmap_size[1])
    target = torch.empty((B, K, Hh, Wh), dtype=torch.float32, device=joints.device)
    weight = torch.empty((B, K, 1), dtype=torch.float32, device=joints.device)
    check(lib().buctd_gaussian_target(ptr(joints), ptr(vis), B, K, Hh, Wh, image_size[0] / heatmap_size[0],
                                      image_size[1] / heatmap_size[1], float(sigma), ptr(target), ptr(weight),
                                      stream_ptr()), "gaussian_target")
    return target, weight


def cond_render(joints, colors, H, W, truncate=False):
    """joints [B,K,>=2] crop pixels; colors [K,Cc] or None (mono 255) -> [B,Cc,H,W] in [0,255]."""
    B, K, js = joints.shape
    Cc = 1 if colors is None else colors.shape[1]
    cond = torch.empty((B, Cc, H, W), dtype=torch.float32, device=joints.device)
    need = lib().buctd_cond_render_workspace(B, Cc, H, W)
    ws = workspace(need, joints.device)
    check(lib().buctd_cond_render(ptr(joints), js, ptr(colors), B, K, Cc, H, W, int(bool(truncate)), ptr(cond),
                                  ptr(ws), ws.numel(), stream_ptr()), "cond_render")
    return cond


def flipback_avg(a, b, perm, shift):
    N, K, H, W = a.shape
    out = torch.empty_like(a)
    check(lib().buctd_flipback_avg(ptr(a), ptr(b), ptr(perm), N, K, H, W, int(bool(shift)), ptr(out), stream_ptr()),
          "flipback_avg")
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, gscale=1.0):
    check(lib().buctd_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, gscale,
                                stream_ptr()), "adam_step")


def sgd_step(p, g, buf, lr, momentum, weight_decay, nesterov, first, gscale=1.0):
    check(lib().buctd_sgd_step(ptr(p), ptr(g), ptr(buf), p.numel(), lr, momentum, weight_decay, int(bool(nesterov)),
                               int(bool(first)), gscale, stream_ptr()), "sgd_step")


def layernorm_fwd(x, gamma, beta, eps):
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    invstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().buctd_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), rows, Cn, eps, ptr(y), ptr(mean), ptr(invstd),
                                    stream_ptr()), "layernorm_fwd")
    return y, mean, invstd


def add_layernorm_fwd(a, b, gamma, beta, eps, keep_sum=True):
    """LayerNorm(a + b) in one pass; -> y, mean, invstd, a + b (None unless keep_sum: the backward pass needs it)."""
    Cn = a.shape[-1]
    rows = a.numel() // Cn
    y = torch.empty_like(a)
    s = torch.empty_like(a) if keep_sum else None
    mean = torch.empty(rows, dtype=torch.float32, device=a.device)
    invstd = torch.empty(rows, dtype=torch.float32, device=a.device)
    check(lib().buctd_add_layernorm_fwd(ptr(a), ptr(b), ptr(gamma), ptr(beta), rows, Cn, eps, ptr(s), ptr(y), ptr(mean),
                                        ptr(invstd), stream_ptr()), "add_layernorm_fwd")
    return y, mean, invstd, s


def layernorm_bwd(dy, x, mean, invstd, gamma, dgamma, dbeta, accumulate):
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    dx = torch.empty_like(x)
    need = lib().buctd_layernorm_bwd_workspace(rows, Cn)
    ws = workspace(need, x.device)
    check(lib().buctd_layernorm_bwd(ptr(dy), ptr(x), ptr(mean), ptr(invstd), ptr(gamma), rows, Cn, ptr(dx),
                                    ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), ws.numel(), stream_ptr()),
          "layernorm_bwd")
    return dx


def dropout(x, p_drop, seed):
    y = torch.empty_like(x)
    check(lib().buctd_dropout(ptr(x), ptr(y), x.numel(), p_drop, seed, stream_ptr()), "dropout")
    return y


def maxpool3x3s2_fwd(x):
    N, H, W, Cn = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, Ho, Wo, Cn), dtype=torch.float32, device=x.device)
    idx = torch.empty((N, Ho, Wo, Cn), dtype=torch.int32, device=x.device)
    check(lib().buctd_maxpool3x3s2_fwd(ptr(x), N, H, W, Cn, ptr(y), ptr(idx), stream_ptr()), "maxpool_fwd")
    return y, idx


def maxpool3x3s2_bwd(dy, idx, x_shape):
    N, H, W, Cn = x_shape
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    check(lib().buctd_maxpool3x3s2_bwd(ptr(dy), ptr(idx), N, H, W, Cn, ptr(dx), stream_ptr()), "maxpool_bwd")
    return dx


# --------------------------------------------------------------------------------------
# autograd Functions
# --------------------------------------------------------------------------------------
def _contig(g):
    return g if g.is_contiguous() else g.contiguous()


class ConvBnAct(torch.autograd.Function):
    """conv (+bias) -> BatchNorm2d -> (+residual) -> (ReLU), train or eval mode.

    Mirrors the conv/bn/relu triplets of reference lib/models/pose_hrnet.py:44-57 (BasicBlock),
    81-99 (Bottleneck) and the nn.Sequential(conv, bn[, relu]) groups of 201-242, 404-432.
    """

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, bn, residual, relu, stride, pad, training, transposed_shape):
        gamma, beta = bn.weight, bn.bias
        eps = bn.eps
        ctx.meta = (bn, conv_w, conv_b, relu, stride, pad, training, transposed_shape, tuple(x.shape))
        if training or bn.running_mean is None:
            momentum = 0.1 if bn.momentum is None else bn.momentum
            if transposed_shape is None:
                z, part, info = conv_fwd(x, conv_w, conv_b, stride, pad, stats="acc")
            else:
                z, part, info = conv_dgrad(x, conv_w, transposed_shape, stride, pad, bias=conv_b, stats="acc")
            Cn = z.shape[-1]
            rows = z.numel() // Cn
            track = bn.track_running_stats and training
            if info[0] == "acc":
                bnin = BnAccInput(part, rows, bn, training)
                y = bn_apply_acc(z, bnin, residual, relu)
                mean, invstd = bnin.mean, bnin.invstd
            else:
                mean, invstd = bn_finalize(part, info, rows, Cn, eps, momentum,
                                           bn.running_mean if track else None, bn.running_var if track else None)
                y = bn_apply(z, mean, invstd, gamma, beta, residual, relu)
            if track:
                bn.count_batch() if hasattr(bn, "count_batch") else bn.num_batches_tracked.add_(1)
            ctx.has_res = residual is not None
            # without a residual the ReLU mask is rebuilt from z in the backward kernels: y is not kept (nor re-read)
            ctx.save_for_backward(x, z, mean, invstd, y if (relu and ctx.has_res) else None)
            if _FUSE_BWD_BNSTAT and info[0] == "acc" and not relu and residual is None and bn_acc_ok(Cn):
                # a plain conv -> BatchNorm output: a FuseSum consuming it can form this BatchNorm's backward sums while it
                # writes the gradient (FuseSum.backward); z / mean / invstd are kept alive by this node anyway
                y._buctd_bn = (z, mean, invstd)
            return y
        scale, shift = bn_fold_cached(bn, gamma, beta, eps)
        if transposed_shape is None:
            y = conv_fwd(x, conv_w, conv_b, stride, pad, scale=scale, shift=shift, residual=residual, relu=relu)
        else:
            z = conv_dgrad(x, conv_w, transposed_shape, stride, pad, bias=conv_b)
            Cn = z.shape[-1]
            zero = torch.zeros(Cn, dtype=torch.float32, device=z.device)
            y = bn_apply(z, zero, scale, torch.ones_like(zero), shift, residual, relu)
        ctx.save_for_backward()
        ctx.eval_mode = True
        return y

    @staticmethod
    def backward(ctx, dy):
        if getattr(ctx, "eval_mode", False):
            raise _C.BuctdHipError("backward through eval-mode BatchNorm is not on the BUCTD path")
        bn, conv_w, conv_b, relu, stride, pad, training, transposed_shape, x_shape = ctx.meta
        x, z, mean, invstd, y = ctx.saved_tensors
        dy = _contig(dy)
        dgamma, acc_g = grad_target(bn.weight)
        dbeta, acc_b = grad_target(bn.bias)
        assert acc_g == acc_b
        ready = _bwd_sums.pop((dy.data_ptr(), z.data_ptr()), None) if _bwd_sums else None
        dz, dres = bn_bwd(dy, y, z, mean, invstd, bn.weight, relu, ctx.has_res and relu, dgamma, dbeta, acc_g,
                          beta=bn.bias, acc=ready, acc_ready=ready is not None)
        if ctx.has_res and not relu:
            dres = dy
        dx = None
        if transposed_shape is None:
            if ctx.needs_input_grad[0]:
                # dx_residual: a gradient that reaches x on another path of the same autograd node (BottleneckFn) - added
                # in the epilogue of the data-gradient kernel instead of by an accumulation pass of autograd
                dx = conv_dgrad(dz, conv_w, x_shape, stride, pad, residual=getattr(ctx, "dx_residual", None))
            dw, acc_w = grad_target(conv_w)
            conv_wgrad_async(x, dz, conv_w, stride, pad, dw, acc_w)
        else:
            # forward was a transposed conv: its data gradient is a plain conv, its weight gradient
            # swaps the roles of input and output gradient
            if ctx.needs_input_grad[0]:
                dx = conv_fwd(dz, conv_w, None, stride, pad)
            dw, acc_w = grad_target(conv_w)
            conv_wgrad_async(dz, x, conv_w, stride, pad, dw, acc_w)
        if conv_b is not None:
            db, acc = grad_target(conv_b)
            colsum(dz, dz.shape[-1], db, acc)
        grad_done(bn.weight, bn.bias, conv_w, conv_b)
        return dx, None, None, None, dres, None, None, None, None, None


class _SubCtx:
    """What ConvBnAct.forward / backward need of an autograd context, for nodes that chain several of them.  The tensors a
    sub-context saves are handed to the REAL context (_pack_subs) - an output of the node kept in a Python attribute would
    close a reference cycle (node -> tensor -> grad_fn -> node) that only the cyclic garbage collector frees: 3 GB of device
    memory per CoAM-W48 step."""

    def __init__(self, needs_x=True):
        self.needs_input_grad = (needs_x,)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _pack_subs(ctx, subs):
    """Move the sub-contexts' saved tensors into ctx.save_for_backward; ctx.sub_layout remembers where each one goes."""
    flat, layout = [], []
    for c in subs:
        if c is None:
            layout.append(None)
            continue
        layout.append(tuple(t is not None for t in c.saved_tensors))
        flat.extend(t for t in c.saved_tensors if t is not None)
        c.saved_tensors = ()
    ctx.save_for_backward(*flat)
    ctx.sub_layout = layout


def _unpack_subs(ctx, subs):
    it = iter(ctx.saved_tensors)
    for c, lay in zip(subs, ctx.sub_layout):
        if c is not None:
            c.saved_tensors = tuple(next(it) if has else None for has in lay)


class BottleneckFn(torch.autograd.Function):
    """One autograd node for the residual Bottleneck (reference lib/models/pose_hrnet.py:60-98, train mode):
    y = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + residual),  residual = x or downsample(x).
    The same kernels in the same order as the three (four) ConvBnAct nodes it replaces; what the single node buys: the
    gradient of the skip path is added in the epilogue of the data-gradient kernel that produces the other gradient of x,
    where autograd ran an accumulation pass over the 256-channel stem-resolution tensor (three reads / writes of 226 MB per
    block at CoAM-W48) - and the host walks one node instead of four."""

    @staticmethod
    def forward(ctx, x, w1, m):
        need = ctx.needs_input_grad[0]
        c1, c2, c3 = _SubCtx(need), _SubCtx(), _SubCtx()
        cd = None
        residual = x
        if m.downsample is not None:
            cd = _SubCtx(need)
            ds_conv, ds_bn = m.downsample[0], m.downsample[1]
            s, p = ds_conv._geom()
            residual = ConvBnAct.forward(cd, x, ds_conv.weight, ds_conv.bias, ds_bn, None, False, s, p, True, None)
        out = ConvBnAct.forward(c1, x, m.conv1.weight, None, m.bn1, None, True, 1, 0, True, None)
        out = ConvBnAct.forward(c2, out, m.conv2.weight, None, m.bn2, None, True, m.stride, 1, True, None)
        y = ConvBnAct.forward(c3, out, m.conv3.weight, None, m.bn3, residual, True, 1, 0, True, None)
        ctx.sub = (c1, c2, c3, cd)
        _pack_subs(ctx, ctx.sub)
        return y

    @staticmethod
    def backward(ctx, dy):
        c1, c2, c3, cd = ctx.sub
        _unpack_subs(ctx, ctx.sub)
        r3 = ConvBnAct.backward(c3, dy)
        d_out2, dres = r3[0], r3[4]
        d_out1 = ConvBnAct.backward(c2, d_out2)[0]
        dx = None
        if ctx.needs_input_grad[0]:
            if cd is not None:
                dres = ConvBnAct.backward(cd, dres)[0]
            c1.dx_residual = dres
        elif cd is not None:
            ConvBnAct.backward(cd, dres)
        dx = ConvBnAct.backward(c1, d_out1)[0]
        for c in ctx.sub:
            if c is not None:
                c.saved_tensors = ()
        return dx, None, None


class ForkConvBnFn(torch.autograd.Function):
    """Several conv -> BatchNorm (-> ReLU) groups reading the SAME input as one autograd node (the first transition of
    HRNet, reference lib/models/pose_hrnet.py:374-402: layer1's 256-channel output feeds both new branches): the data
    gradients chain through each other's epilogue (dx = dgrad_1(dz_1) + dgrad_0(dz_0)) instead of meeting in an
    accumulation pass of autograd.  Same kernels as the separate ConvBnAct nodes."""

    @staticmethod
    def forward(ctx, x, w0, groups):
        need = ctx.needs_input_grad[0]
        subs, outs = [], []
        for conv, bn, relu in groups:
            c = _SubCtx(need)
            s, p = conv._geom()
            outs.append(ConvBnAct.forward(c, x, conv.weight, conv.bias, bn, None, relu, s, p, True, None))
            subs.append(c)
        ctx.sub = subs
        _pack_subs(ctx, subs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        subs = ctx.sub
        _unpack_subs(ctx, subs)
        dx = None
        for c, dy in zip(subs, dys):
            if dy is None:      # this output took no part in the loss
                continue
            c.dx_residual = dx
            r = ConvBnAct.backward(c, dy)[0]
            dx = r if r is not None else dx
        for c in subs:
            c.saved_tensors = ()
        return dx, None, None


class BasicBlockFn(torch.autograd.Function):
    """One autograd node for the whole residual BasicBlock (reference lib/models/pose_hrnet.py:28-57, train mode, stride
    1, no downsample):  y = relu(bn2(conv2(relu(bn1(conv1(x))))) + x).
    Same kernels as two ConvBnAct nodes; what the single node buys: the gradient of the skip connection is added in the
    epilogue of conv1's data-gradient kernel instead of by an autograd accumulation kernel (one launch and three
    tensor passes per block), and the host walks one node instead of two plus an accumulation."""

    @staticmethod
    def forward(ctx, x, w1, bn1, w2, bn2):
        # In the bf16x6 mode conv2 applies bn1 + ReLU while it stages its input: the tensor y1 = relu(bn1(conv1(x)))
        # never exists in HBM (one kernel and two tensor passes less per block; conv2's weight gradient rebuilds it the
        # same way).  Bit-identical to the unfused sequence: the staged value is bn_apply's own expression.
        fuse = bn_in_fusable(tuple(x.shape), w2)
        veto = native_block_veto["fn"]
        if (fuse and _NATIVE_BLOCK and bn_in_fusable(tuple(x.shape), w1) and bn1.track_running_stats == bn2.track_running_stats
                and not (veto is not None and veto(tuple(x.shape)))):
            return BasicBlockFn._forward_native(ctx, x, w1, bn1, w2, bn2)
        z1, part, info = conv_fwd(x, w1, None, 1, 1, stats="acc")
        Cn = z1.shape[-1]
        rows = z1.numel() // Cn
        y1 = None
        if info[0] == "acc":
            # statistics as accumulators: the consumer of a BatchNorm decodes them, no finalize launch
            bnin1 = BnAccInput(part, rows, bn1, True, relu=True)
            if fuse:
                z2, part2, info2 = conv_fwd(z1, w2, None, 1, 1, stats="acc", in_bn=bnin1)
            else:
                y1 = bn_apply_acc(z1, bnin1, None, True)
                z2, part2, info2 = conv_fwd(y1, w2, None, 1, 1, stats="acc")
            bnin2 = BnAccInput(part2, rows, bn2, True)
            y = bn_apply_acc(z2, bnin2, x, True)
            mean1, invstd1, mean2, invstd2 = bnin1.mean, bnin1.invstd, bnin2.mean, bnin2.invstd
        else:
            def fin(bn, part, info):
                track = bn.track_running_stats
                return bn_finalize(part, info, rows, Cn, bn.eps, 0.1 if bn.momentum is None else bn.momentum,
                                   bn.running_mean if track else None, bn.running_var if track else None)
            mean1, invstd1 = fin(bn1, part, info)
            if fuse:
                z2, part2, info2 = conv_fwd(z1, w2, None, 1, 1, stats=True, in_bn=(mean1, invstd1, bn1.weight, bn1.bias, True))
            else:
                y1 = bn_apply(z1, mean1, invstd1, bn1.weight, bn1.bias, None, True)
                z2, part2, info2 = conv_fwd(y1, w2, None, 1, 1, stats=True)
            mean2, invstd2 = fin(bn2, part2, info2)
            y = bn_apply(z2, mean2, invstd2, bn2.weight, bn2.bias, x, True)
        for bn in (bn1, bn2):
            if bn.track_running_stats:
                bn.count_batch() if hasattr(bn, "count_batch") else bn.num_batches_tracked.add_(1)
        ctx.meta = (w1, bn1, w2, bn2, fuse)
        ctx.save_for_backward(x, z1, mean1, invstd1, None if fuse else y1, z2, mean2, invstd2, y)
        return y

    @staticmethod
    def _forward_native(ctx, x, w1, bn1, w2, bn2):
        """The same five kernels through ONE library call (block.hip): the nine ctypes calls and dozen small allocations of
        the step-by-step path cost ~95 us of host time per block - more than HRNet-W32 needs on the GPU."""
        N, H, W, Cn = x.shape
        dev = x.device
        act = torch.empty((3, N, H, W, Cn), dtype=torch.float32, device=dev)       # z1 | z2 | y
        stat = torch.empty((4, Cn), dtype=torch.float32, device=dev)               # mean1 | invstd1 | mean2 | invstd2
        d = _C.BasicBlockDesc()
        d.N, d.H, d.W, d.C = N, H, W, Cn
        d.x = x.data_ptr()
        d.w1_fwd = _conv3x3_prepared(w1, 0).data_ptr()
        d.w2_fwd = _conv3x3_prepared(w2, 0).data_ptr()
        d.gamma1, d.beta1 = bn1.weight.data_ptr(), bn1.bias.data_ptr()
        d.gamma2, d.beta2 = bn2.weight.data_ptr(), bn2.bias.data_ptr()
        track = bn1.track_running_stats
        if track:
            d.running_mean1, d.running_var1 = bn1.running_mean.data_ptr(), bn1.running_var.data_ptr()
            d.running_mean2, d.running_var2 = bn2.running_mean.data_ptr(), bn2.running_var.data_ptr()
        d.eps1, d.momentum1 = bn1.eps, 0.1 if bn1.momentum is None else bn1.momentum
        d.eps2, d.momentum2 = bn2.eps, 0.1 if bn2.momentum is None else bn2.momentum
        base, step = act.data_ptr(), 4 * N * H * W * Cn
        d.z1, d.z2, d.y = base, base + step, base + 2 * step
        d.acc, d.stat = AccRef(Cn, dev, 2).ptr, stat.data_ptr()
        check(lib().buctd_basic_block_fwd_train(C.byref(d), stream_ptr()), "basic_block_fwd_train")
        if track:
            for bn in (bn1, bn2):
                bn.count_batch() if hasattr(bn, "count_batch") else bn.num_batches_tracked.add_(1)
        y = act[2]
        ctx.meta = (w1, bn1, w2, bn2, "native")
        ctx.save_for_backward(x, act, stat)
        return y

    @staticmethod
    def _backward_native(ctx, dy):
        w1, bn1, w2, bn2, _ = ctx.meta
        x, act, stat = ctx.saved_tensors
        dy = _contig(dy)
        N, H, W, Cn = x.shape
        dev = x.device
        want_dx = ctx.needs_input_grad[0]
        tmp = torch.empty((5 if want_dx else 4, N, H, W, Cn), dtype=torch.float32, device=dev)   # dz2 | dres | dy1 | dz1 | dx
        d = _C.BasicBlockDesc()
        d.N, d.H, d.W, d.C = N, H, W, Cn
        d.x = x.data_ptr()
        d.w1_fwd = d.w2_fwd = 0
        d.w1_bwd = _conv3x3_prepared(w1, 1).data_ptr()
        d.w2_bwd = _conv3x3_prepared(w2, 1).data_ptr()
        d.gamma1, d.beta1 = bn1.weight.data_ptr(), bn1.bias.data_ptr()
        d.gamma2, d.beta2 = bn2.weight.data_ptr(), bn2.bias.data_ptr()
        base, step = act.data_ptr(), 4 * N * H * W * Cn
        d.z1, d.z2, d.y = base, base + step, base + 2 * step
        d.stat = stat.data_ptr()
        g = _C.BasicBlockGrads()
        tb = tmp.data_ptr()
        g.dy, g.dz2, g.dres, g.dy1, g.dz1 = dy.data_ptr(), tb, tb + step, tb + 2 * step, tb + 3 * step
        g.dx = tb + 4 * step if want_dx else 0
        dg2, acc_g2 = grad_target(bn2.weight)
        db2, acc_b2 = grad_target(bn2.bias)
        dw2, acc_w2 = grad_target(w2)
        dg1, acc_g1 = grad_target(bn1.weight)
        db1, acc_b1 = grad_target(bn1.bias)
        dw1, acc_w1 = grad_target(w1)
        assert acc_g2 == acc_b2 and acc_g1 == acc_b1
        weight_rsc(dw1)
        weight_rsc(dw2)
        g.dw1, g.dw2 = dw1.data_ptr(), dw2.data_ptr()
        g.dgamma1, g.dbeta1, g.dgamma2, g.dbeta2 = dg1.data_ptr(), db1.data_ptr(), dg2.data_ptr(), db2.data_ptr()
        g.acc_w1, g.acc_w2, g.acc_bn1, g.acc_bn2 = int(acc_w1), int(acc_w2), int(acc_g1), int(acc_g2)
        g.bn_acc = AccRef(Cn, dev, 2).ptr
        main = torch.cuda.current_stream(dev)
        use_side = _side["on"]
        side = _side_stream(dev) if use_side else main
        need = _memo(("wg3", "bf16x6", N, H, W, Cn, Cn),
                     lambda: (lib().buctd_conv3x3_wgrad_bf16x6_workspace(N, H, W, Cn, Cn)
                              if lib().buctd_conv3x3_wgrad_bf16x6_supported(N, H, W, Cn, Cn) == 1 else -1))
        wg_ws = workspace_on(side, need, dev)      # the side stream's own scratch buffer
        g.wg_ws, g.wg_ws_bytes = wg_ws.data_ptr(), wg_ws.numel()
        check(lib().buctd_basic_block_bwd(C.byref(d), C.byref(g), main.cuda_stream, side.cuda_stream if use_side else None),
              "basic_block_bwd")
        if use_side:
            for t in (x, act, stat, tmp):
                t.record_stream(side)
            _queue_join()
        elif _branch["on"]:
            _queue_join()
        grad_done(bn2.weight, bn2.bias, w2)
        grad_done(bn1.weight, bn1.bias, w1)
        return (tmp[4] if want_dx else None), None, None, None, None

    @staticmethod
    def backward(ctx, dy):
        if ctx.meta[4] == "native":
            return BasicBlockFn._backward_native(ctx, dy)
        w1, bn1, w2, bn2, fuse = ctx.meta
        x, z1, mean1, invstd1, y1, z2, mean2, invstd2, y2 = ctx.saved_tensors
        dy = _contig(dy)
        # conv2 / bn2 (+ skip): dres = masked upstream gradient
        dg, acc_g = grad_target(bn2.weight)
        db, acc_b = grad_target(bn2.bias)
        assert acc_g == acc_b
        dz2, dres = bn_bwd(dy, y2, z2, mean2, invstd2, bn2.weight, True, True, dg, db, acc_g)
        dy1 = conv_dgrad(dz2, w2, tuple(z1.shape), 1, 1)
        dw, acc_w = grad_target(w2)
        if fuse:
            conv_wgrad_async(z1, dz2, w2, 1, 1, dw, acc_w, x_bn=(mean1, invstd1, bn1.weight, bn1.bias, True))
        else:
            conv_wgrad_async(y1, dz2, w2, 1, 1, dw, acc_w)
        grad_done(bn2.weight, bn2.bias, w2)
        # conv1 / bn1: ReLU mask rebuilt from z1; the skip gradient joins in the dgrad epilogue
        dg, acc_g = grad_target(bn1.weight)
        db, acc_b = grad_target(bn1.bias)
        assert acc_g == acc_b
        dz1, _ = bn_bwd(dy1, None, z1, mean1, invstd1, bn1.weight, True, False, dg, db, acc_g, beta=bn1.bias)
        dx = conv_dgrad(dz1, w1, tuple(x.shape), 1, 1, residual=dres) if ctx.needs_input_grad[0] else None
        dw, acc_w = grad_target(w1)
        conv_wgrad_async(x, dz1, w1, 1, 1, dw, acc_w)
        grad_done(bn1.weight, bn1.bias, w1)
        return dx, None, None, None, None


class BasicChainFn(torch.autograd.Function):
    """A chain of residual BasicBlocks (an HRNet branch: pose_hrnet.py:165-185) as ONE autograd node and one library call
    per direction (block.hip: buctd_basic_chain_*): the launches of BasicBlockFn's native path, a quarter of its host work
    per block.  blocks: [(w1, bn1, w2, bn2), ...]; w_first only makes autograd build the node when x needs no gradient."""

    @staticmethod
    def forward(ctx, x, w_first, blocks):
        n = len(blocks)
        N, H, W, Cn = x.shape
        dev = x.device
        act = torch.empty((n, 3, N, H, W, Cn), dtype=torch.float32, device=dev)        # per block: z1 | z2 | y
        stat = torch.empty((n, 4, Cn), dtype=torch.float32, device=dev)
        descs = (_C.BasicBlockDesc * n)()
        step = 4 * N * H * W * Cn
        accb = acc_bytes(Cn)
        abase, pbase, sbase = act.data_ptr(), AccRef(Cn, dev, 2 * n).ptr, stat.data_ptr()
        xin = x.data_ptr()
        for k, (w1, bn1, w2, bn2) in enumerate(blocks):
            d = descs[k]
            d.N, d.H, d.W, d.C = N, H, W, Cn
            d.x = xin
            d.w1_fwd = _conv3x3_prepared(w1, 0).data_ptr()
            d.w2_fwd = _conv3x3_prepared(w2, 0).data_ptr()
            d.gamma1, d.beta1 = bn1.weight.data_ptr(), bn1.bias.data_ptr()
            d.gamma2, d.beta2 = bn2.weight.data_ptr(), bn2.bias.data_ptr()
            if bn1.track_running_stats:
                d.running_mean1, d.running_var1 = bn1.running_mean.data_ptr(), bn1.running_var.data_ptr()
                d.running_mean2, d.running_var2 = bn2.running_mean.data_ptr(), bn2.running_var.data_ptr()
                bn1.count_batch() if hasattr(bn1, "count_batch") else bn1.num_batches_tracked.add_(1)
                bn2.count_batch() if hasattr(bn2, "count_batch") else bn2.num_batches_tracked.add_(1)
            d.eps1, d.momentum1 = bn1.eps, 0.1 if bn1.momentum is None else bn1.momentum
            d.eps2, d.momentum2 = bn2.eps, 0.1 if bn2.momentum is None else bn2.momentum
            b0 = abase + 3 * step * k
            d.z1, d.z2, d.y = b0, b0 + step, b0 + 2 * step
            d.acc, d.stat = pbase + k * 2 * accb, sbase + k * 4 * Cn * 4
            xin = d.y
        check(lib().buctd_basic_chain_fwd_train(n, descs, stream_ptr()), "basic_chain_fwd_train")
        ctx.blocks = blocks
        ctx.save_for_backward(x, act, stat)
        return act[n - 1, 2]

    @staticmethod
    def backward(ctx, dy):
        blocks = ctx.blocks
        n = len(blocks)
        x, act, stat = ctx.saved_tensors
        dy = _contig(dy)
        N, H, W, Cn = x.shape
        dev = x.device
        want_dx = ctx.needs_input_grad[0]
        tmp = torch.empty((n, 5, N, H, W, Cn), dtype=torch.float32, device=dev)     # per block: dz2 | dres | dy1 | dz1 | dx
        step = 4 * N * H * W * Cn
        abase, sbase, tb = act.data_ptr(), stat.data_ptr(), tmp.data_ptr()
        descs = (_C.BasicBlockDesc * n)()
        grads = (_C.BasicBlockGrads * n)()
        main = torch.cuda.current_stream(dev)
        use_side = _side["on"]
        side = _side_stream(dev) if use_side else main
        accb = acc_bytes(Cn)
        bn_acc = AccRef(Cn, dev, 2 * n).ptr
        need = _memo(("wg3", "bf16x6", N, H, W, Cn, Cn),
                     lambda: (lib().buctd_conv3x3_wgrad_bf16x6_workspace(N, H, W, Cn, Cn)
                              if lib().buctd_conv3x3_wgrad_bf16x6_supported(N, H, W, Cn, Cn) == 1 else -1))
        wg_ws = workspace_on(side, need, dev)
        xin = x.data_ptr()
        for k, (w1, bn1, w2, bn2) in enumerate(blocks):
            d, g = descs[k], grads[k]
            d.N, d.H, d.W, d.C = N, H, W, Cn
            d.x = xin
            d.w1_bwd = _conv3x3_prepared(w1, 1).data_ptr()
            d.w2_bwd = _conv3x3_prepared(w2, 1).data_ptr()
            d.gamma1, d.beta1 = bn1.weight.data_ptr(), bn1.bias.data_ptr()
            d.gamma2, d.beta2 = bn2.weight.data_ptr(), bn2.bias.data_ptr()
            b0 = abase + 3 * step * k
            d.z1, d.z2, d.y = b0, b0 + step, b0 + 2 * step
            d.stat = sbase + k * 4 * Cn * 4
            xin = d.y
            t0 = tb + 5 * step * k
            g.dy = dy.data_ptr() if k == n - 1 else tb + 5 * step * (k + 1) + 4 * step      # the next block's dx
            g.dz2, g.dres, g.dy1, g.dz1 = t0, t0 + step, t0 + 2 * step, t0 + 3 * step
            g.dx = t0 + 4 * step if (k > 0 or want_dx) else 0
            dg2, acc_g2 = grad_target(bn2.weight)
            db2, acc_b2 = grad_target(bn2.bias)
            dw2, acc_w2 = grad_target(w2)
            dg1, acc_g1 = grad_target(bn1.weight)
            db1, acc_b1 = grad_target(bn1.bias)
            dw1, acc_w1 = grad_target(w1)
            assert acc_g2 == acc_b2 and acc_g1 == acc_b1
            weight_rsc(dw1)
            weight_rsc(dw2)
            g.dw1, g.dw2 = dw1.data_ptr(), dw2.data_ptr()
            g.dgamma1, g.dbeta1, g.dgamma2, g.dbeta2 = dg1.data_ptr(), db1.data_ptr(), dg2.data_ptr(), db2.data_ptr()
            g.acc_w1, g.acc_w2, g.acc_bn1, g.acc_bn2 = int(acc_w1), int(acc_w2), int(acc_g1), int(acc_g2)
            g.bn_acc = bn_acc + k * 2 * accb
            g.wg_ws, g.wg_ws_bytes = wg_ws.data_ptr(), wg_ws.numel()
        check(lib().buctd_basic_chain_bwd(n, descs, grads, main.cuda_stream, side.cuda_stream if use_side else None),
              "basic_chain_bwd")
        if use_side:
            for t in (x, act, stat, tmp, dy):
                t.record_stream(side)
            _queue_join()
        elif _branch["on"]:
            _queue_join()
        for (w1, bn1, w2, bn2) in reversed(blocks):
            grad_done(bn2.weight, bn2.bias, w2)
            grad_done(bn1.weight, bn1.bias, w1)
        return (tmp[0, 4] if want_dx else None), None, None


_GROUP_BRANCHES = {"on": True}


def set_group_branches(on):
    """HighResolutionModule branches as one autograd node with cross-branch group launches (BasicBranchesFn); off = one
    BasicChainFn per branch on the branch streams (tests compare the two)."""
    old = _GROUP_BRANCHES["on"]
    _GROUP_BRANCHES["on"] = bool(on)
    return old


# 2: two group launch families side by side on two streams (branches {0, 1} | {2, 3}): C4 464.9 against 454.0 img/s with one
# family and 462.5 with one chain per branch; C2 1170 against 1100 / 971 (one box, 20 steps)
_GROUP_PARTS = {"n": 2}


def set_group_parts(n):
    old = _GROUP_PARTS["n"]
    _GROUP_PARTS["n"] = int(n)
    return old


_GROUP_PARTITION = {}      # branches -> explicit partition (experiments: ops.set_group_partition)


def set_group_partition(nb, parts):
    """Explicit assignment of the branches of an nb-branch module to group-launch families, e.g. (3, [[0, 1], [2]])."""
    assert sorted(i for p in parts for i in p) == list(range(nb))
    _GROUP_PARTITION[nb] = [list(p) for p in parts]


def group_branch_partition(nb):
    """Branch indices per group launch family: one group of all branches, or two groups side by side on two streams."""
    if nb in _GROUP_PARTITION:
        return _GROUP_PARTITION[nb]
    if _GROUP_PARTS["n"] <= 1 or nb < 3:
        return [list(range(nb))]
    if nb == 3:
        return [[0], [1, 2]] if _GROUP_PARTS["n"] == 2 else [[0, 2], [1]]
    return [[0, 1], [2, 3]] if _GROUP_PARTS["n"] == 2 else [[0, 3], [1, 2]]


def group_branches_ok(xs, chains):
    """True when the branches of a HighResolutionModule (activations xs[b], chains[b] = [(w1, bn1, w2, bn2), ...]) may run
    as ONE native call per direction with cross-branch group launches."""
    if not (_GROUP_BRANCHES["on"] and 1 <= len(xs) <= 4 and len({len(c) for c in chains}) == 1):
        return False
    cfs = set()
    for x, chain in zip(xs, chains):
        if not (x.is_cuda and x.dim() == 4 and native_chain_ok(tuple(x.shape)) and bn_in_fusable(tuple(x.shape), chain[0][0])):
            return False
        Cn = x.shape[-1]
        cfs.add(3 if Cn % 48 == 0 else 2 if Cn % 32 == 0 else 0)
    return len(cfs) == 1 and 0 not in cfs


def eval_branches_ok(xs, chains):
    """The eval-mode counterpart of group_branches_ok: every branch convolution has a bf16x6 3x3 kernel."""
    if not (_GROUP_BRANCHES["on"] and _conv_math["mode"] == "bf16x6" and 2 <= len(xs) <= 4 and len({len(c) for c in chains}) == 1):
        return False
    for x, chain in zip(xs, chains):
        if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous()):
            return False
        for (w1, _, w2, _) in chain:
            if not (_bf16x3_ok(conv_desc(x.shape, _wshape(w1), 1, 1)) and _bf16x3_ok(conv_desc(x.shape, _wshape(w2), 1, 1))):
                return False
            if _wshape(w1)[0] != x.shape[-1] or _wshape(w2)[0] != x.shape[-1]:
                return False

    def one_kernel():
        # do the tile shapes of all branches live in one group kernel?  (small maps at one or two persons per call do not:
        # then the branches keep their own launches on the branch streams)
        items = (_C.C3Conv * len(xs))()
        for it, x in zip(items, xs):
            it.N, it.H, it.W, it.Ci, it.Co = x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.shape[3]
            it.x = it.wprep = it.y = xs[0].data_ptr()      # no launch: the query only reads the shapes
        return lib().buctd_conv3x3_bf16x6_group_workgroups(len(xs), items) > 0
    return _memo(("evalgrp",) + tuple(tuple(x.shape) for x in xs), one_kernel)


def basic_branches_eval(xs, chains):
    """Eval-mode forward of the branches of a HighResolutionModule (pose_hrnet.py:177-185): the k-th convolutions of all
    branches - folded BatchNorm, skip connection and ReLU in the epilogue - share a launch (buctd_conv3x3_bf16x6_group_eval).
    The same kernels on the same tiles as the per-branch path: bit-identical, 2 n launches instead of 2 n nb."""
    nb, n = len(xs), len(chains[0])
    cur = list(xs)
    for k in range(n):
        folded = [(bn_fold_cached(ch[k][1], ch[k][1].weight, ch[k][1].bias, ch[k][1].eps),
                   bn_fold_cached(ch[k][3], ch[k][3].weight, ch[k][3].bias, ch[k][3].eps)) for ch in chains]
        mids = [torch.empty_like(x) for x in cur]
        outs = [torch.empty_like(x) for x in cur]
        for half, dst in ((0, mids), (1, outs)):
            items = (_C.C3ConvEval * nb)()
            for b in range(nb):
                x = cur[b] if half == 0 else mids[b]
                w = chains[b][k][2 * half]
                weight_rsc(w)
                N, H, W, Cn = x.shape
                sc, sh = folded[b][half]
                it = items[b]
                it.N, it.H, it.W, it.Ci, it.Co = N, H, W, Cn, _wshape(w)[0]
                it.x, it.wprep = x.data_ptr(), _conv3x3_prepared(w, 0).data_ptr()
                it.scale, it.shift = sc.data_ptr(), sh.data_ptr()
                it.residual = cur[b].data_ptr() if half == 1 else None
                it.relu = 1
                it.y = dst[b].data_ptr()
            check(lib().buctd_conv3x3_bf16x6_group_eval(nb, items, stream_ptr()), "conv3x3_group_eval")
        cur = outs
    return cur


class BasicBranchesFn(torch.autograd.Function):
    """The branches of a HighResolutionModule (pose_hrnet.py:177-185, 247-249) - nb chains of n residual BasicBlocks on maps of
    different size - as ONE autograd node and one library call per direction (block.hip: buctd_basic_branches_*).  The k-th
    convolutions of all branches are one launch (several de-phased rounds of workgroups instead of nb phase-locked rounds on
    nb streams), so are the BatchNorm applies / backwards and the weight gradients.
    chains[b] = [(w1, bn1, w2, bn2), ...]; w_first only makes autograd build the node when no input needs a gradient."""

    @staticmethod
    def forward(ctx, w_first, chains, *xs):
        nb, n = len(xs), len(chains[0])
        dev = xs[0].device
        descs = (_C.BasicBlockDesc * (nb * n))()
        acts, stats = [], []
        for b, x in enumerate(xs):
            N, H, W, Cn = x.shape
            act = torch.empty((n, 3, N, H, W, Cn), dtype=torch.float32, device=dev)        # per block: z1 | z2 | y
            stat = torch.empty((n, 4, Cn), dtype=torch.float32, device=dev)
            acts.append(act)
            stats.append(stat)
            step = 4 * N * H * W * Cn
            accb = acc_bytes(Cn)
            abase, pbase, sbase = act.data_ptr(), AccRef(Cn, dev, 2 * n).ptr, stat.data_ptr()
            xin = x.data_ptr()
            for k, (w1, bn1, w2, bn2) in enumerate(chains[b]):
                d = descs[b * n + k]
                d.N, d.H, d.W, d.C = N, H, W, Cn
                d.x = xin
                d.w1_fwd = _conv3x3_prepared(w1, 0).data_ptr()
                d.w2_fwd = _conv3x3_prepared(w2, 0).data_ptr()
                d.gamma1, d.beta1 = bn1.weight.data_ptr(), bn1.bias.data_ptr()
                d.gamma2, d.beta2 = bn2.weight.data_ptr(), bn2.bias.data_ptr()
                if bn1.track_running_stats:
                    d.running_mean1, d.running_var1 = bn1.running_mean.data_ptr(), bn1.running_var.data_ptr()
                    d.running_mean2, d.running_var2 = bn2.running_mean.data_ptr(), bn2.running_var.data_ptr()
                    bn1.count_batch() if hasattr(bn1, "count_batch") else bn1.num_batches_tracked.add_(1)
                    bn2.count_batch() if hasattr(bn2, "count_batch") else bn2.num_batches_tracked.add_(1)
                d.eps1, d.momentum1 = bn1.eps, 0.1 if bn1.momentum is None else bn1.momentum
                d.eps2, d.momentum2 = bn2.eps, 0.1 if bn2.momentum is None else bn2.momentum
                b0 = abase + 3 * step * k
                d.z1, d.z2, d.y = b0, b0 + step, b0 + 2 * step
                d.acc, d.stat = pbase + k * 2 * accb, sbase + k * 4 * Cn * 4
                xin = d.y
        check(lib().buctd_basic_branches_fwd_train(nb, n, descs, stream_ptr()), "basic_branches_fwd_train")
        ctx.chains = chains
        ctx.save_for_backward(*xs, *acts, *stats)
        return tuple(act[n - 1, 2] for act in acts)

    @staticmethod
    def backward(ctx, *dys):
        chains = ctx.chains
        nb, n = len(chains), len(chains[0])
        saved = ctx.saved_tensors
        xs, acts, stats = saved[:nb], saved[nb:2 * nb], saved[2 * nb:]
        dev = xs[0].device
        descs = (_C.BasicBlockDesc * (nb * n))()
        grads = (_C.BasicBlockGrads * (nb * n))()
        main = torch.cuda.current_stream(dev)
        use_side = _side["on"]
        side = _side_stream(dev) if use_side else main
        needs = []
        for x in xs:
            N, H, W, Cn = x.shape
            needs.append(_memo(("wg3g", nb, N, H, W, Cn), lambda: (int(lib().buctd_conv3x3_wgrad_bf16x6_group_workspace(
                nb, N, H, W, Cn, Cn)) + 255) & ~255))
        wg_ws = workspace_on(side, sum(needs), dev)     # one slab area per branch: the branches share a launch
        ws_ptr = wg_ws.data_ptr()
        tmps, keep = [], []
        for b, x in enumerate(xs):
            N, H, W, Cn = x.shape
            dy = dys[b]
            dy = torch.zeros_like(x) if dy is None else _contig(dy)
            keep.append(dy)
            want_dx = ctx.needs_input_grad[2 + b]
            tmp = torch.empty((n, 5, N, H, W, Cn), dtype=torch.float32, device=dev)   # per block: dz2 | dres | dy1 | dz1 | dx
            tmps.append(tmp)
            step = 4 * N * H * W * Cn
            abase, sbase, tb = acts[b].data_ptr(), stats[b].data_ptr(), tmp.data_ptr()
            accb = acc_bytes(Cn)
            bn_acc = AccRef(Cn, dev, 2 * n).ptr
            xin = x.data_ptr()
            for k, (w1, bn1, w2, bn2) in enumerate(chains[b]):
                d, g = descs[b * n + k], grads[b * n + k]
                d.N, d.H, d.W, d.C = N, H, W, Cn
                d.x = xin
                d.w1_bwd = _conv3x3_prepared(w1, 1).data_ptr()
                d.w2_bwd = _conv3x3_prepared(w2, 1).data_ptr()
                d.gamma1, d.beta1 = bn1.weight.data_ptr(), bn1.bias.data_ptr()
                d.gamma2, d.beta2 = bn2.weight.data_ptr(), bn2.bias.data_ptr()
                b0 = abase + 3 * step * k
                d.z1, d.z2, d.y = b0, b0 + step, b0 + 2 * step
                d.stat = sbase + k * 4 * Cn * 4
                xin = d.y
                t0 = tb + 5 * step * k
                g.dy = dy.data_ptr() if k == n - 1 else tb + 5 * step * (k + 1) + 4 * step      # the next block's dx
                g.dz2, g.dres, g.dy1, g.dz1 = t0, t0 + step, t0 + 2 * step, t0 + 3 * step
                g.dx = t0 + 4 * step if (k > 0 or want_dx) else 0
                dg2, acc_g2 = grad_target(bn2.weight)
                db2, acc_b2 = grad_target(bn2.bias)
                dw2, acc_w2 = grad_target(w2)
                dg1, acc_g1 = grad_target(bn1.weight)
                db1, acc_b1 = grad_target(bn1.bias)
                dw1, acc_w1 = grad_target(w1)
                assert acc_g2 == acc_b2 and acc_g1 == acc_b1
                weight_rsc(dw1)
                weight_rsc(dw2)
                g.dw1, g.dw2 = dw1.data_ptr(), dw2.data_ptr()
                g.dgamma1, g.dbeta1, g.dgamma2, g.dbeta2 = dg1.data_ptr(), db1.data_ptr(), dg2.data_ptr(), db2.data_ptr()
                g.acc_w1, g.acc_w2, g.acc_bn1, g.acc_bn2 = int(acc_w1), int(acc_w2), int(acc_g1), int(acc_g2)
                g.bn_acc = bn_acc + k * 2 * accb
                g.wg_ws, g.wg_ws_bytes = ws_ptr, needs[b]
            ws_ptr += needs[b]
        check(lib().buctd_basic_branches_bwd(nb, n, descs, grads, main.cuda_stream, side.cuda_stream if use_side else None),
              "basic_branches_bwd")
        if use_side:
            for t in list(xs) + list(acts) + list(stats) + tmps + keep:
                t.record_stream(side)
            _queue_join()
        elif _branch["on"]:
            _queue_join()
        for k in range(n - 1, -1, -1):
            for b in range(nb):
                w1, bn1, w2, bn2 = chains[b][k]
                grad_done(bn2.weight, bn2.bias, w2)
                grad_done(bn1.weight, bn1.bias, w1)
        return (None, None) + tuple(tmps[b][0, 4] if ctx.needs_input_grad[2 + b] else None for b in range(nb))


class Conv(torch.autograd.Function):
    """conv / Linear-as-1x1-conv with bias, optional fused residual + ReLU (no BatchNorm)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, relu):
        y = conv_fwd(x, w, b, stride, pad, relu=relu)
        ctx.meta = (w, b, stride, pad, relu, tuple(x.shape))
        ctx.save_for_backward(x, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        w, b, stride, pad, relu, x_shape = ctx.meta
        x, y = ctx.saved_tensors
        dy = _contig(dy)
        if relu:
            dy = relu_bwd(dy, y)
        dx = conv_dgrad(dy, w, x_shape, stride, pad) if ctx.needs_input_grad[0] else None
        if w.requires_grad:
            dw, acc = grad_target(w)
            conv_wgrad_async(x, dy, w, stride, pad, dw, acc)
        if b is not None and b.requires_grad:
            db, acc = grad_target(b)
            colsum(dy, dy.shape[-1], db, acc)
        grad_done(w, b)
        return dx, None, None, None, None, None


class FuseSum(torch.autograd.Function):
    """relu(sum_j nearest_upsample(term_j)) - reference lib/models/pose_hrnet.py:257-265."""

    @staticmethod
    def forward(ctx, shifts, relu, *terms):
        out = fuse_sum(list(terms), list(shifts), relu)
        ctx.shifts, ctx.relu = shifts, relu
        ctx.save_for_backward(out if relu else None)
        # terms that are plain conv -> BatchNorm outputs (ConvBnAct.forward tags them): (z, mean, invstd) of their BatchNorm
        ctx.term_bn = tuple(getattr(t, "_buctd_bn", None) for t in terms) if _FUSE_BWD_BNSTAT else None
        return out

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _contig(dy)
        cache = {}
        grads = []
        for j, s in enumerate(ctx.shifts):
            if not ctx.needs_input_grad[2 + j]:
                grads.append(None)
                continue
            if s not in cache:
                if s > 0 or ctx.relu:
                    bns = []
                    # s > 0 only: the shift-0 gradient also feeds the identity term - the branch chain, which is the critical
                    # path of the backward pass - and must not wait for extra passes over the terms' z (measured: -0.8 %)
                    if ctx.term_bn is not None and (s > 0 or _FUSE_BWD_BNSTAT_S0):
                        bns = [ctx.term_bn[k] for k, sk in enumerate(ctx.shifts)
                               if sk == s and ctx.needs_input_grad[2 + k] and ctx.term_bn[k] is not None][:3]
                    cache[s] = fuse_sum_bwd_bnstat(dy, y, s, bns) if bns else fuse_sum_bwd(dy, y, s)
                else:
                    cache[s] = dy
            grads.append(cache[s])
        ctx.term_bn = None
        return (None, None, *grads)


class AddN(torch.autograd.Function):
    """out = a + b (+ c), used for DAModule's input + (p_out + c_out) (pose_hrnet_coam.py:724)."""

    @staticmethod
    def forward(ctx, a, b, c):
        out = add(a, b)
        if c is not None:
            out = add(out, c, out=out)
        ctx.n = 3 if c is not None else 2
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy, (dy if ctx.n == 3 else None)


def add_n(terms):
    """terms[0] + terms[1] (+ ...) in as few passes as possible (up to four terms per launch), left to right."""
    acc = terms[0]
    k = 1
    while k < len(terms):
        group = [acc] + list(terms[k:k + 3])
        k += 3
        out = torch.empty_like(acc)
        arr = (C.c_void_p * len(group))(*[t.data_ptr() for t in group])
        check(lib().buctd_add_n(arr, len(group), ptr(out), acc.numel(), stream_ptr()), "add_n")
        acc = out
    return acc


class FanOut(torch.autograd.Function):
    """Hands a tensor to n consumers as n aliases and sums their gradients itself - one n-ary add instead of autograd's
    chain of n - 1 two-operand ATen adds (66 launches per CoAM-W48 step).  The sum runs in consumer order."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [_contig(g) for g in grads if g is not None]
        if not gs:
            return None, None
        return (gs[0] if len(gs) == 1 else add_n(gs)), None


class ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return nhwc_to_nchw(x)

    @staticmethod
    def backward(ctx, dy):
        return nchw_to_nhwc(_contig(dy))


class PositionAttention(torch.autograd.Function):
    """softmax(q k^T / sqrt(dk)) -> dropout -> . v for h heads (self_attention.py:74-86).
    q [B,Tq,h*dk], k [B,Tk,h*dk], v [B,Tk,h*dv] -> [B,Tq,h*dv].
    Packed form (nn.MultiheadAttention self-attention): k=None and q = [B,T,2*h*dk] holding q | k side by side,
    as produced by one fused input-projection GEMM; the backward then returns one packed gradient."""

    @staticmethod
    def forward(ctx, q, k, v, h, p_drop, training):
        packed = k is None
        B, Tq = q.shape[0], q.shape[1]
        ldq = q.shape[2]
        hdk = ldq // 2 if packed else ldq
        kt, k_off, ldk = (q, hdk, ldq) if packed else (k, 0, k.shape[2])
        Tk = kt.shape[1]
        dk, dv = hdk // h, v.shape[2] // h
        scale = 1.0 / math.sqrt(dk)
        p_eff = p_drop if training else 0.0
        seed = next_seed()
        S = torch.empty((B, h, Tq, Tk), dtype=torch.float32, device=q.device)
        for i in range(h):
            matmul(q, kt, S, batch=B, M=Tq, N=Tk, K=dk, a_layout=0, b_layout=0, lda=ldq, ldb=ldk, ldc=Tk,
                   stride_a=Tq * ldq, stride_b=Tk * ldk, stride_c=h * Tq * Tk, a_off=i * dk, b_off=k_off + i * dk,
                   c_off=i * Tq * Tk)
        P, Pd = softmax_dropout_fwd(S, Tk, scale, p_eff, seed)
        O = torch.empty((B, Tq, h * dv), dtype=torch.float32, device=q.device)
        for i in range(h):
            matmul(Pd, v, O, batch=B, M=Tq, N=dv, K=Tk, a_layout=0, b_layout=1, lda=Tk, ldb=h * dv, ldc=h * dv,
                   stride_a=h * Tq * Tk, stride_b=Tk * h * dv, stride_c=Tq * h * dv, a_off=i * Tq * Tk,
                   b_off=i * dv, c_off=i * dv)
        ctx.meta = (h, dk, dv, scale, p_eff, seed, packed)
        ctx.save_for_backward(q, k, v, P, Pd if p_eff > 0 else None)
        return O

    @staticmethod
    def backward(ctx, dO):
        h, dk, dv, scale, p_eff, seed, packed = ctx.meta
        q, k, v, P, Pd = ctx.saved_tensors
        if Pd is None:
            Pd = P
        dO = _contig(dO)
        B, Tq, ldq = q.shape
        hdk = h * dk
        kt, k_off, ldk = (q, hdk, ldq) if packed else (k, 0, k.shape[2])
        Tk = kt.shape[1]
        hdv = h * dv
        dq = torch.empty_like(q)
        dkk = dq if packed else torch.empty_like(k)
        dvv = torch.empty_like(v)
        dP = torch.empty_like(P)
        for i in range(h):
            # dV = Pd^T dO
            matmul(Pd, dO, dvv, batch=B, M=Tk, N=dv, K=Tq, a_layout=1, b_layout=1, lda=Tk, ldb=hdv, ldc=hdv,
                   stride_a=h * Tq * Tk, stride_b=Tq * hdv, stride_c=Tk * hdv, a_off=i * Tq * Tk, b_off=i * dv,
                   c_off=i * dv)
            # dPd = dO V^T
            matmul(dO, v, dP, batch=B, M=Tq, N=Tk, K=dv, a_layout=0, b_layout=0, lda=hdv, ldb=hdv, ldc=Tk,
                   stride_a=Tq * hdv, stride_b=Tk * hdv, stride_c=h * Tq * Tk, a_off=i * dv, b_off=i * dv,
                   c_off=i * Tq * Tk)
        dS = softmax_dropout_bwd(dP, P, Tk, scale, p_eff, seed)
        for i in range(h):
            # dq = dS k ; dk = dS^T q
            matmul(dS, kt, dq, batch=B, M=Tq, N=dk, K=Tk, a_layout=0, b_layout=1, lda=Tk, ldb=ldk, ldc=ldq,
                   stride_a=h * Tq * Tk, stride_b=Tk * ldk, stride_c=Tq * ldq, a_off=i * Tq * Tk, b_off=k_off + i * dk,
                   c_off=i * dk)
            matmul(dS, q, dkk, batch=B, M=Tk, N=dk, K=Tq, a_layout=1, b_layout=1, lda=Tk, ldb=ldq, ldc=ldk,
                   stride_a=h * Tq * Tk, stride_b=Tq * ldq, stride_c=Tk * ldk, a_off=i * Tq * Tk, b_off=i * dk,
                   c_off=k_off + i * dk)
        return dq, (None if packed else dkk), dvv, None, None, None


def mha_fused_ok(T, d, h=1):
    """True when the fused flash-style self-attention forward (attn_mha.hip) covers this shape."""
    return h == 1 and lib().buctd_mha_fwd_supported(int(T), int(d)) == 1


def mha_fwd(qk, v, scale=None):
    """softmax(scale * q k^T) v for one head without the T x T matrix.  qk [B, T, 2d] = q | k side by side (the packed
    nn.MultiheadAttention input projection), v [B, T, d] -> [B, T, d].  Inference / no-dropout path."""
    _f32(qk, "mha qk")
    _f32(v, "mha v")
    B, T, two_d = qk.shape
    d = two_d // 2
    out = torch.empty((B, T, d), dtype=torch.float32, device=qk.device)
    kptr = C.c_void_p(qk.data_ptr() + 4 * d)
    # default math mode: both products in bf16x6 (fp32 class on the bf16 matrix cores); fp32 mode: exact fp32 MFMA
    sc = (1.0 / math.sqrt(d)) if scale is None else float(scale)
    if _conv_math["mode"] == "bf16x6" and _MHA_X6:
        if _MHA_PRESPLIT:       # keys / values split once into the workspace, streamed by DMA (same bits, faster)
            nb = _memo(("mhaws", B, T, d), lambda: lib().buctd_mha_fwd_bf16x6_workspace(B, T, d))
            ws = workspace(nb, qk.device)
            check(lib().buctd_mha_fwd_bf16x6_ws(B, T, d, ptr(qk), kptr, ptr(v), two_d, v.shape[2], sc, ptr(out), None,
                                                ptr(ws), ws.numel(), stream_ptr()), "mha_fwd")
            return out
        fn = lib().buctd_mha_fwd_bf16x6
    else:
        fn = lib().buctd_mha_fwd
    check(fn(B, T, d, ptr(qk), kptr, ptr(v), two_d, v.shape[2], sc, ptr(out), None, stream_ptr()), "mha_fwd")
    return out


def mha_train_ok(T, d, h=1):
    """True when the fused training attention (attn_mha_train.hip: forward with dropout + lse, flash-style backward) covers
    this shape."""
    return h == 1 and _memo(("mhatr", int(T), int(d)), lambda: lib().buctd_mha_train_supported(int(T), int(d)) == 1)


class FusedMHA(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) -> dropout -> . v for ONE head without the T x T matrix, forward and backward (reference
    nn.MultiheadAttention inside transpose_h.py:192-197, training).  qk [B, T, 2d] = q | k side by side (the packed input
    projection), v [B, T, d] -> [B, T, d]; the backward returns the packed gradient d(q | k) and dv."""

    @staticmethod
    def forward(ctx, qk, v, p_drop, training):
        qk, v = _contig(qk), _contig(v)
        _f32(qk, "mha qk")
        _f32(v, "mha v")
        B, T, two_d = qk.shape
        d = two_d // 2
        scale = 1.0 / math.sqrt(d)
        p_eff = float(p_drop) if training else 0.0
        seed = next_seed()
        out = torch.empty((B, T, d), dtype=torch.float32, device=qk.device)
        lse = torch.empty((B, T), dtype=torch.float32, device=qk.device)
        check(lib().buctd_mha_fwd_train(B, T, d, ptr(qk), C.c_void_p(qk.data_ptr() + 4 * d), ptr(v), two_d, v.shape[2], scale,
                                        p_eff, seed, ptr(out), ptr(lse), stream_ptr()), "mha_fwd_train")
        ctx.meta = (scale, p_eff, seed)
        ctx.save_for_backward(qk, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        scale, p_eff, seed = ctx.meta
        qk, v, out, lse = ctx.saved_tensors
        dout = _contig(dout)
        B, T, two_d = qk.shape
        d = two_d // 2
        dqk = torch.empty_like(qk)
        dv = torch.empty_like(v)
        ws = workspace(lib().buctd_mha_bwd_workspace(B, T), qk.device)
        check(lib().buctd_mha_bwd(B, T, d, ptr(qk), C.c_void_p(qk.data_ptr() + 4 * d), ptr(v), two_d, v.shape[2], ptr(out),
                                  ptr(dout), ptr(lse), scale, p_eff, seed, ptr(dqk), C.c_void_p(dqk.data_ptr() + 4 * d), two_d,
                                  ptr(dv), dv.shape[2], ptr(ws), ws.numel(), stream_ptr()), "mha_bwd")
        return dqk, dv, None, None


def attn_smallqk_ok(T, d_in, C, h=1):
    """True when the fused narrow-contraction attention (attn_smallqk.hip) covers this shape."""
    if h != 1 or d_in + 1 > 20:
        return False
    R4 = _smallqk_r4(d_in)
    return bool(lib().buctd_attn_smallqk_supported(T, R4, C))


def _smallqk_r4(d_in):
    r = d_in + 1
    return 4 if r <= 4 else 8 if r <= 8 else 16 if r <= 16 else 20


class SmallQKAttention(torch.autograd.Function):
    """fc_q -> softmax(q k^T / sqrt(C)) -> dropout -> . v (self_attention.py:74-86, h = 1) without the T x T matrix.
    yq [B,T,d] are the raw condition tokens (fc_q's input), wq [C,d] / bq [C] fc_q's parameters, k / v [B,T,C].
    fc_q is folded into the keys: logits = [yq, 1] ([wq | bq]^T k^T), so the kernels contract over R4 <= 20
    channels on the VALU and keep the MFMA for the T- and C-contractions (attn_smallqk.hip)."""

    @staticmethod
    def forward(ctx, yq, wq, bq, k, v, p_drop, training):
        B, T, d = yq.shape
        Cn = k.shape[2]
        R4 = _smallqk_r4(d)
        dev = yq.device
        qp = torch.zeros((B, T, R4), dtype=torch.float32, device=dev)
        qp[:, :, :d] = yq
        qp[:, :, d] = 1.0
        w4t = torch.zeros((R4, Cn), dtype=torch.float32, device=dev)       # rows: wq^T, bq, 0-pad
        w4t[:d] = wq.detach().t()
        w4t[d] = bq.detach()
        k = _contig(k)
        v = _contig(v)
        kp = torch.empty((B, T, R4), dtype=torch.float32, device=dev)
        matmul(k, w4t, kp, batch=1, M=B * T, N=R4, K=Cn, a_layout=0, b_layout=0, lda=Cn, ldb=Cn, ldc=R4)
        scale = 1.0 / math.sqrt(Cn)
        p_eff = float(p_drop) if training else 0.0
        seed = next_seed()
        out = torch.empty((B, T, Cn), dtype=torch.float32, device=dev)
        m = torch.empty((B, T), dtype=torch.float32, device=dev)
        linv = torch.empty((B, T), dtype=torch.float32, device=dev)
        # contractions over T and C on the bf16 matrix cores: two pieces per operand in the bf16x3 mode, three (fp32
        # class) in the default bf16x6 mode, the exact fp32 MFMA kernels in the fp32 mode
        b3 = {"bf16x3": 1, "bf16x6": 2}.get(_conv_math["mode"], 0)
        if not _ATTN_X6 and b3 == 2:
            b3 = 0
        check(lib().buctd_attn_smallqk_fwd(B, T, R4, Cn, ptr(qp), ptr(kp), ptr(v), scale, p_eff, seed, b3, ptr(out), ptr(m),
                                           ptr(linv), stream_ptr()), "attn_smallqk_fwd")
        ctx.meta = (d, R4, scale, p_eff, seed, b3)
        ctx.save_for_backward(qp, kp, k, v, out, m, linv, w4t, wq, bq)
        return out

    @staticmethod
    def backward(ctx, dout):
        d, R4, scale, p_eff, seed, b3 = ctx.meta
        qp, kp, k, v, out, m, linv, w4t, wq, bq = ctx.saved_tensors
        dout = _contig(dout)
        B, T, Cn = v.shape
        dev = v.device
        dqp = torch.empty_like(qp)
        dkp = torch.empty_like(kp)
        dv = torch.empty_like(v)
        dvec = torch.empty((B, T), dtype=torch.float32, device=dev)
        check(lib().buctd_attn_smallqk_bwd(B, T, R4, Cn, ptr(qp), ptr(kp), ptr(v), ptr(out), ptr(dout), ptr(m), ptr(linv),
                                           scale, p_eff, seed, b3, ptr(dqp), ptr(dkp), ptr(dv), ptr(dvec), stream_ptr()),
              "attn_smallqk_bwd")
        # k' = k w4t^T  ->  dk = dk' w4t ; dw4t = dk'^T k
        dk = torch.empty_like(k)
        matmul(dkp, w4t, dk, batch=1, M=B * T, N=Cn, K=R4, a_layout=0, b_layout=1, lda=R4, ldb=Cn, ldc=Cn)
        dw4t = torch.empty_like(w4t)
        matmul(dkp, k, dw4t, batch=1, M=R4, N=Cn, K=B * T, a_layout=1, b_layout=1, lda=R4, ldb=Cn, ldc=Cn)
        dyq = dqp[:, :, :d].contiguous() if ctx.needs_input_grad[0] else None
        gw, acc = grad_target(wq)
        if acc:
            gw.add_(dw4t[:d].t())
        else:
            gw.copy_(dw4t[:d].t())
        grad_done(wq)
        gb, acc = grad_target(bq)
        if acc:
            gb.add_(dw4t[d])
        else:
            gb.copy_(dw4t[d])
        grad_done(bq)
        return dyq, None, None, dk, dv, None, None


class ChannelAttention(torch.autograd.Function):
    """SimplifiedScaledDotProductAttention + fc_o on channel-major queries (self_attention.py:146-159)
    evaluated on NHWC token tensors: qn [B,T,C] (condition features), yn [B,T,C] (keys = values).
    Logits are [B,h,C,C] with the reduction over the T/h tokens of each head; fc_o = Linear(T,T)."""

    @staticmethod
    def forward(ctx, qn, yn, fc_w, fc_b, h, p_drop, training):
        B, T, Cn = yn.shape
        dk = T // h
        scale = 1.0 / math.sqrt(dk)
        p_eff = p_drop if training else 0.0
        seed = next_seed()
        Lg = torch.empty((B, h, Cn, Cn), dtype=torch.float32, device=yn.device)
        for i in range(h):
            # L[c1][c2] = sum_t qn[t][c1] yn[t][c2]
            matmul(qn, yn, Lg, batch=B, M=Cn, N=Cn, K=dk, a_layout=1, b_layout=1, lda=Cn, ldb=Cn, ldc=Cn,
                   stride_a=T * Cn, stride_b=T * Cn, stride_c=h * Cn * Cn, a_off=i * dk * Cn, b_off=i * dk * Cn,
                   c_off=i * Cn * Cn)
        A, Ad = softmax_dropout_fwd(Lg, Cn, scale, p_eff, seed)
        on = torch.empty((B, T, Cn), dtype=torch.float32, device=yn.device)
        for i in range(h):
            # on[t][c1] = sum_c2 yn[t][c2] Ad[c1][c2]   (tokens of head i)
            matmul(yn, Ad, on, batch=B, M=dk, N=Cn, K=Cn, a_layout=0, b_layout=0, lda=Cn, ldb=Cn, ldc=Cn,
                   stride_a=T * Cn, stride_b=h * Cn * Cn, stride_c=T * Cn, a_off=i * dk * Cn, b_off=i * Cn * Cn,
                   c_off=i * dk * Cn)
        # fc_o over the token axis for all images at once: out[b][t'][c] = sum_t W[t'][t] on[b][t][c] + bias[t']
        out = torch.empty_like(on)
        if fc_o_x6_ok(T, B * Cn):
            x6_gemm(_x6_weight_image(fc_w, T, T, False),
                    x6_image(on, B * Cn, T, 1, vg=Cn, vgs=T * Cn, vs=1, ks=Cn), out, T, B * Cn, T, ldc=Cn, Nc=Cn,
                    gsc=T * Cn, bias=fc_b, bias_axis=1)
        else:
            matmul(fc_w, on, out, batch=1, M=T, N=B * Cn, K=T, a_layout=0, b_layout=1, lda=T, ldb=Cn, ldc=Cn,
                   Nc=Cn, gsbn=T * Cn, gsc=T * Cn, bias=fc_b, bias_axis=1)
        ctx.meta = (fc_w, fc_b, h, dk, scale, p_eff, seed)
        ctx.save_for_backward(qn, yn, A, Ad if p_eff > 0 else None, on)
        return out

    @staticmethod
    def backward(ctx, dout):
        fc_w, fc_b, h, dk, scale, p_eff, seed = ctx.meta
        qn, yn, A, Ad, on = ctx.saved_tensors
        if Ad is None:
            Ad = A
        dout = _contig(dout)
        B, T, Cn = yn.shape
        dev = yn.device
        # fc_o parameter gradients
        if fc_w.requires_grad:
            dw, acc = grad_target(fc_w)
            tgt = dw if not acc else torch.empty_like(dw)
            if fc_o_x6_ok(T, B * Cn):
                x6_gemm(x6_image(dout, T, B * Cn, 0, vs=Cn, ks=1, kg=Cn, kgs=T * Cn),
                        x6_image(on, T, B * Cn, 1, vs=Cn, ks=1, kg=Cn, kgs=T * Cn), tgt, T, T, B * Cn, ldc=T)
            else:
                matmul(dout, on, tgt, batch=1, M=T, N=T, K=B * Cn, a_layout=0, b_layout=0, lda=Cn, ldb=Cn, ldc=T,
                       Kc=Cn, gsa=T * Cn, gsbk=T * Cn)
            if acc:
                add(dw, tgt, out=dw)
        if fc_b is not None and fc_b.requires_grad:
            db, acc = grad_target(fc_b)
            ones = torch.ones(B * Cn, dtype=torch.float32, device=dev)
            tgt = db if not acc else torch.empty_like(db)
            matmul(dout, ones, tgt, batch=1, M=T, N=1, K=B * Cn, a_layout=0, b_layout=0, lda=Cn, ldb=B * Cn, ldc=1,
                   Kc=Cn, gsa=T * Cn, gsbk=Cn)
            if acc:
                add(db, tgt, out=db)
        grad_done(fc_w, fc_b)
        # d_on[b][t][c] = sum_t' W[t'][t] dout[b][t'][c]
        don = torch.empty_like(on)
        if fc_o_x6_ok(T, B * Cn):
            x6_gemm(_x6_weight_image(fc_w, T, T, True),
                    x6_image(dout, B * Cn, T, 1, vg=Cn, vgs=T * Cn, vs=1, ks=Cn), don, T, B * Cn, T, ldc=Cn, Nc=Cn,
                    gsc=T * Cn)
        else:
            matmul(fc_w, dout, don, batch=1, M=T, N=B * Cn, K=T, a_layout=1, b_layout=1, lda=T, ldb=Cn, ldc=Cn,
                   Nc=Cn, gsbn=T * Cn, gsc=T * Cn)
        dAd = torch.empty_like(A)
        dyn_v = torch.empty_like(yn)
        for i in range(h):
            # dAd[c1][c2] = sum_t don[t][c1] yn[t][c2]
            matmul(don, yn, dAd, batch=B, M=Cn, N=Cn, K=dk, a_layout=1, b_layout=1, lda=Cn, ldb=Cn, ldc=Cn,
                   stride_a=T * Cn, stride_b=T * Cn, stride_c=h * Cn * Cn, a_off=i * dk * Cn, b_off=i * dk * Cn,
                   c_off=i * Cn * Cn)
            # dV[t][c2] = sum_c1 don[t][c1] Ad[c1][c2]
            matmul(don, Ad, dyn_v, batch=B, M=dk, N=Cn, K=Cn, a_layout=0, b_layout=1, lda=Cn, ldb=Cn, ldc=Cn,
                   stride_a=T * Cn, stride_b=h * Cn * Cn, stride_c=T * Cn, a_off=i * dk * Cn, b_off=i * Cn * Cn,
                   c_off=i * dk * Cn)
        dL = softmax_dropout_bwd(dAd, A, Cn, scale, p_eff, seed)
        dqn = torch.empty_like(qn)
        dyn_k = torch.empty_like(yn)
        for i in range(h):
            # dq[t][c1] = sum_c2 yn[t][c2] dL[c1][c2] ; dK[t][c2] = sum_c1 qn[t][c1] dL[c1][c2]
            matmul(yn, dL, dqn, batch=B, M=dk, N=Cn, K=Cn, a_layout=0, b_layout=0, lda=Cn, ldb=Cn, ldc=Cn,
                   stride_a=T * Cn, stride_b=h * Cn * Cn, stride_c=T * Cn, a_off=i * dk * Cn, b_off=i * Cn * Cn,
                   c_off=i * dk * Cn)
            matmul(qn, dL, dyn_k, batch=B, M=dk, N=Cn, K=Cn, a_layout=0, b_layout=1, lda=Cn, ldb=Cn, ldc=Cn,
                   stride_a=T * Cn, stride_b=h * Cn * Cn, stride_c=T * Cn, a_off=i * dk * Cn, b_off=i * Cn * Cn,
                   c_off=i * dk * Cn)
        dyn = add(dyn_v, dyn_k, out=dyn_v)
        return dqn, dyn, None, None, None, None, None


class JointsMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, w):
        loss, grad = joints_mse(pred, gt, w, want_grad=pred.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, dl):
        (grad,) = ctx.saved_tensors
        # chain rule through the scalar loss, on device (no host sync)
        if grad is None:
            return None, None, None
        return scale(grad, dl.reshape(1).contiguous()), None, None
