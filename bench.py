#!/usr/bin/env python
"""Headline benchmark: training throughput (images/s) of BUCTD-CoAM-W48 at 384x288 on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = the full per-batch work of reference lib/core/function.py:train (102-175) on one synthetic batch
already resident in HBM: forward (train-mode BN, dropout on), JointsMSELoss, zero_grad, backward, gradient
all-reduce (N > 1), Adam step, plus the arg-max decode of output and target for the accuracy meter.
Weak scaling: 32 images per GPU (scripts/train/train_BUCTD_COAM_gen_sample.sh:17).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: fp32 MFMA (v_mfma_f32_16x16x4_f32) peak = fp32 vector peak; dense bf16 MFMA peak; HBM3E spec
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBPS = 8000.0
MFMAS_PER_PRODUCT = {"bf16x6": 6, "bf16x3": 3}
# Counters cannot be read inside the timed run.  HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction) and
# the kernels' durations INSIDE the step (rocprofv3 --kernel-trace of this very command) are therefore measured by
# scratch/r04_profiles.sh on the build that is committed and quoted from the files it wrote - with their name and date - and
# only for the math mode / batch / shape they were collected at.  A missing file means null, never a stale constant.
def _profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def _box_probe_start():
    """rocm-smi as a child process: shader / memory clock, socket power, temperatures - of the box this line was measured on"""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        return subprocess.Popen([exe, "-d", "0", "--showclocks", "--showpower", "--showtemp", "--json"], stdout=subprocess.PIPE,
                                stderr=subprocess.DEVNULL, text=True)
    except OSError:
        return None


def _box_probe_collect(proc):
    if proc is None:
        return None
    try:
        out, _ = proc.communicate(timeout=20)
        card = next(iter(json.loads(out).values()))
    except Exception:      # noqa: BLE001 - a diagnostic, never a reason to lose the bench line
        return None
    keep = {}
    for k, v in card.items():
        kl = k.lower()
        if any(w in kl for w in ("sclk", "mclk", "power", "temperature")) and "voltage" not in kl:
            keep[k] = v
    return keep or None


def _serialised_census(tag):
    """kernel time per step with every stream serialised, from the committed census of this round (scratch/serial_census.sh)"""
    name = "r06_kernel_trace_stats_serialised_streams.txt" if tag == "c4" else f"r06_kernel_trace_stats_serialised_streams_{tag}.txt"
    try:
        import re
        txt = open(os.path.join(ROOT, "profiles", name)).read()
        m = re.search(r"kernel time ([0-9.]+) ms per step", txt)
        b = re.search(r"serialised streams: ([0-9.]+) ([0-9.]+)", txt)
        return {"kernel_ms_per_step": float(m.group(1)) if m else None, "ms_per_step": float(b.group(2)) if b else None,
                "source": "profiles/" + name}
    except OSError:
        return None


def coam_w48_cfg(batch, colored=True):
    from buctd_amd.config import cfg as base, hrnet_extra
    c = base.clone()
    c.defrost()
    c.MODEL.NAME = "pose_hrnet_coam"
    c.MODEL.NUM_JOINTS = 14
    c.MODEL.IMAGE_SIZE = [288, 384]
    c.MODEL.HEATMAP_SIZE = [72, 96]
    c.MODEL.SIGMA = 3
    c.MODEL.PRETRAINED = ""
    c.MODEL.ATT_MODULES = [False, True, False, False]
    c.MODEL.CONDITIONAL_TOPDOWN = True
    c.MODEL.EXTRA = hrnet_extra(48, use_attention=True)
    c.DATASET.DATASET = "crowdpose"
    c.DATASET.COLORED = bool(colored)
    c.TRAIN.BATCH_SIZE_PER_GPU = batch
    c.TRAIN.LR = 0.002
    c.freeze()
    return c


def prenet_cfg(batch, width, image_size):
    """BASELINE configs C2 (W32 256x192) / C3 (W48 384x288): BUCTD-preNet HRNet, COCO 17 key points, colored condition
    stacked on the RGB crop (experiments/coco/hrnet/*prenet*: USE_PRE_NET)."""
    from buctd_amd.config import cfg as base, hrnet_extra
    c = base.clone()
    c.defrost()
    c.MODEL.NAME = "pose_hrnet"
    c.MODEL.NUM_JOINTS = 17
    c.MODEL.IMAGE_SIZE = list(image_size)
    c.MODEL.HEATMAP_SIZE = [image_size[0] // 4, image_size[1] // 4]
    c.MODEL.SIGMA = 3 if image_size[1] >= 384 else 2
    c.MODEL.PRETRAINED = ""
    c.MODEL.CONDITIONAL_TOPDOWN = True
    c.MODEL.EXTRA = hrnet_extra(width, use_pre_net=True)
    c.DATASET.DATASET = "coco"
    c.DATASET.COLORED = True
    c.TRAIN.BATCH_SIZE_PER_GPU = batch
    c.TRAIN.LR = 0.001
    c.freeze()
    return c


# workload -> (cfg builder, model module, metric, description, roofline shape (C, H, W) = stage-2..4 branch 0)
TRAIN_WORKLOADS = {
    "train_c4": (lambda b: coam_w48_cfg(b), "pose_hrnet_coam", "images/sec (train) BUCTD-CoAM-W48 384x288",
                 "BUCTD-CoAM-W48 (pose_hrnet_coam, ATT_MODULES [F,T,F,F], colored condition) 384x288 CrowdPose-14kpt",
                 (48, 96, 72)),
    "train_c3": (lambda b: prenet_cfg(b, 48, (288, 384)), "pose_hrnet", "images/sec (train) BUCTD-preNet-W48 384x288",
                 "BUCTD-preNet HRNet-W48 (pose_hrnet, USE_PRE_NET, colored condition) 384x288 COCO-17kpt", (48, 96, 72)),
    "train_c2": (lambda b: prenet_cfg(b, 32, (192, 256)), "pose_hrnet", "images/sec (train) BUCTD-preNet-W32 256x192",
                 "BUCTD-preNet HRNet-W32 (pose_hrnet, USE_PRE_NET, colored condition) 256x192 COCO-17kpt", (32, 64, 48)),
}


CROWDPOSE_COLORS = [[245, 53, 53], [245, 125, 45], [253, 206, 20], [206, 244, 54], [118, 253, 27], [47, 254, 47],
                    [25, 245, 113], [15, 243, 197], [14, 199, 245], [44, 126, 249], [13, 13, 249], [128, 47, 249],
                    [205, 38, 247], [245, 48, 206]]


def synthetic_batch(cfg, batch, device, seed):
    """RGB ~ N(0,1); colored condition rendered from synthesized key points (GT + generative noise); Gaussian
    targets sigma 3; target_weight ~ Bernoulli(0.8). Generated on the device, outside the timed region."""
    from buctd_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    w, h = cfg.MODEL.IMAGE_SIZE
    k = cfg.MODEL.NUM_JOINTS
    rgb = torch.randn(batch, 3, h, w, generator=g).to(device)
    gt = torch.rand(batch, k, 2, generator=g) * torch.tensor([w - 1.0, h - 1.0])
    # generative-noise condition: the device port of lib/dataset/pose_synthesis.py perturbs the ground
    # truth (jitter / miss / inversion / swap with a neighbouring person's joints), as the "generative sampling" recipes do
    from buctd_amd.dataset.pose_synthesis import synthesize_pose_batch
    j3 = torch.cat([gt, torch.ones(batch, k, 1)], 2).double()
    near = torch.cat([(gt + torch.randn(batch, k, 2, generator=g) * 40.0), torch.ones(batch, k, 1)], 2).double()[:, None]
    area = torch.full((batch,), float(w * h) * 0.5, dtype=torch.float64)
    syn = synthesize_pose_batch(cfg.DATASET.DATASET, j3.numpy(), j3.numpy(), near.numpy(), area.numpy(),
                                [1] * batch, seed, device=device)
    cond_j = syn[:, :, :2].float().contiguous()
    colors = torch.tensor((CROWDPOSE_COLORS * 2)[:k], dtype=torch.float32, device=device)
    if not cfg.DATASET.COLORED:
        # mono condition (north_star's literal "4-channel crops"; pose_hrnet_coam.py:750-757): one white heat-map channel
        colors = torch.ones(k, 1, dtype=torch.float32, device=device)
    cond = ops.cond_render(cond_j, colors, h, w)
    x = torch.cat([rgb, cond], 1).contiguous()
    joints3 = torch.cat([gt, torch.zeros(batch, k, 1)], 2).to(device).contiguous()
    vis = (torch.rand(batch, k, generator=g) < 0.8).float().to(device)
    target, weight = ops.gaussian_target(joints3, vis, cfg.MODEL.HEATMAP_SIZE, cfg.MODEL.IMAGE_SIZE, cfg.MODEL.SIGMA)
    return x, target, weight


class KernelTimer:
    """HIP events (torch.cuda.Event = hipEvent) recorded on the launching stream right before and after every launch
    of the roofline shape - the 3x3 48 -> 48 convolution at 96x72 (HRNet stage-2/3/4 branch 0) - during the timed
    steps: forward, data-gradient and weight-gradient launches are kept apart."""

    def __init__(self, batch, shape=(48, 96, 72)):
        self.batch = batch
        self.shape = shape
        self.pairs = {"fwd": [], "dgrad": [], "wgrad": []}
        self.enabled = False

    def match(self, d):
        c, h, w = self.shape
        return (d.R == 3 and d.stride == 1 and d.pad == 1 and d.Ci == c and d.Co == c and d.H == h and d.W == w
                and d.N == self.batch)

    def timed(self, kind, fn, stream=None):
        """events on the stream the kernel is launched on: the current one, or the explicit stream of a weight gradient"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream) if stream is not None else e0.record()
        out = fn()
        e1.record(stream) if stream is not None else e1.record()
        self.pairs[kind].append((e0, e1))
        return out

    def mean_us(self, kind):
        ps = self.pairs[kind]
        return (1e3 * sum(a.elapsed_time(b) for a, b in ps) / len(ps), len(ps)) if ps else (None, 0)

    def reset(self):
        for k in self.pairs:
            self.pairs[k] = []


def install_timer(timer):
    from buctd_amd import ops
    raw_fwd, raw_dgrad, raw_wgrad = ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad

    def conv_fwd(x, w, bias=None, stride=1, pad=0, **kw):
        if timer.enabled:
            d = ops.conv_desc(x.shape, ops._wshape(w), stride, pad)
            if timer.match(d):
                if ops._bf16x3_ok(d):
                    ops._conv3x3_prepared(w, 0)   # the per-weight-version filter re-layout is its own kernel
                return timer.timed("fwd", lambda: raw_fwd(x, w, bias, stride, pad, **kw))
        return raw_fwd(x, w, bias, stride, pad, **kw)

    def conv_dgrad(dy, w, x_shape, stride=1, pad=0, **kw):
        if timer.enabled:
            d = ops.conv_desc(x_shape, ops._wshape(w), stride, pad)
            if timer.match(d):
                if ops._bf16x3_ok(d):
                    ops._conv3x3_prepared(w, 1)
                return timer.timed("dgrad", lambda: raw_dgrad(dy, w, x_shape, stride, pad, **kw))
        return raw_dgrad(dy, w, x_shape, stride, pad, **kw)

    def conv_wgrad(x, dy, w_like, stride=1, pad=0, **kw):
        if timer.enabled:
            d = ops.conv_desc(x.shape, ops._wshape(w_like), stride, pad)
            if timer.match(d):
                return timer.timed("wgrad", lambda: raw_wgrad(x, dy, w_like, stride, pad, **kw), kw.get("stream"))
        return raw_wgrad(x, dy, w_like, stride, pad, **kw)

    ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad = conv_fwd, conv_dgrad, conv_wgrad
    # BasicBlocks normally go through one native call per direction (block.hip); the blocks of the roofline shape take the
    # step-by-step path while the timer is on, so that every one of their launches is bracketed individually
    # (every 4th such block: enough samples - >300 launches per kind - at a quarter of the perturbation)
    c, h, w = timer.shape
    seen = {"n": 0}

    def veto(xs):
        if not (timer.enabled and xs == (timer.batch, h, w, c)):
            return False
        seen["n"] += 1
        return seen["n"] % 4 == 0
    ops.native_block_veto["fn"] = veto


def roofline_entry(math, batch, kind, in_step, solo, traffic, shape=(48, 96, 72), kernel_us=None, sources=None):
    """One roofline object.  Algorithmic work of one launch (SURVEY 8d x batch): 2*N*H*W*C*C*9 FLOP (C4: 96x72, 48
    channels) and fp32 bytes in + out + weights.  in_step / solo = (average launch us, launches); kernel_us = the kernel's
    average DURATION inside the step from the committed rocprofv3 trace of this command (None: not collected)."""
    n = batch
    cw, hh, ww = shape
    flops = 2.0 * n * hh * ww * cw * cw * 9
    if kind == "wgrad":
        bytes_ = 4.0 * (2 * n * cw * hh * ww) + 4.0 * 9 * cw * cw       # read x and dy once, write dW
    else:
        bytes_ = 4.0 * (2 * n * cw * hh * ww) + 4.0 * 9 * cw * cw       # read x, write y, read W
    t_head = in_step[0] if in_step[0] else solo[0]
    if not t_head:
        return None
    names = {"fwd": "forward + data-gradient launches", "wgrad": "weight-gradient launches (kernel + slab reduction)"}
    if math == "fp32":
        peak, unit_note = PEAK_FP32_MFMA_TFLOPS, "exact fp32 MFMA (v_mfma_f32_16x16x4_f32) = the fp32 vector peak"
        kernel = "conv_gemm_kernel / conv_wgrad_kernel (fp32 MFMA)"
    else:
        k = MFMAS_PER_PRODUCT[math]
        peak = PEAK_BF16_MFMA_TFLOPS / k
        unit_note = (f"dense bf16 MFMA peak 2500 TFLOP/s / {k} MFMAs per fp32 product (split operands) = "
                     f"{peak:.1f} TFLOP/s-equivalent")
        kernel = {("bf16x6", "fwd"): "conv3x3_x6_kernel<MF=7,NF=3,WM=4,WN=1> (448-position tiles, one round of 506 workgroups)",
                  ("bf16x3", "fwd"): "conv3x3_split_kernel<NP=2,4,3,4,1>"}.get(
                      (math, "fwd" if kind == "fwd" else "wgrad"),
                      "conv3x3_wgrad_split_kernel<NP=%d,3> + wg3_reduce_kernel" % (3 if math == "bf16x6" else 2))
    hbm_us = bytes_ / (PEAK_HBM_GBPS * 1e3)
    mfma_us = flops / (peak * 1e6)
    bound = "mfma" if mfma_us >= hbm_us else "hbm"

    def frac(us):
        return round((mfma_us if bound == "mfma" else hbm_us) / us, 4) if us else None

    t_ach = kernel_us if kernel_us else t_head      # `achieved` / `frac` follow the in-step kernel duration where it is known
    tf = flops / t_ach / 1e6
    gbps = bytes_ / t_ach / 1e3
    if shape != (48, 96, 72):
        kernel = kernel.split("<")[0] + "<...>"
    return {"kernel": f"{kernel}: 3x3 {cw}->{cw} @{hh}x{ww} N={n} (HRNet branch 0), {names[kind]}",
            "bound": bound,
            "achieved": round(tf, 2) if bound == "mfma" else round(gbps, 1),
            "peak": round(peak, 1) if bound == "mfma" else PEAK_HBM_GBPS,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
            # frac = the reproducible one: the kernel's own duration inside the step (committed rocprofv3 trace); without a trace
            # for this configuration, the live event bracket
            "frac": frac(kernel_us) if kernel_us else frac(t_head),
            "frac_source": "frac_in_step_kernel" if kernel_us else "frac_event_bracket",
            "avg_kernel_us_in_step": kernel_us, "frac_in_step_kernel": frac(kernel_us),
            "frac_event_bracket": frac(t_head),
            "traffic": traffic, "traffic_unit": "HBM bytes/launch (PMC, standalone launches)" if traffic else None,
            "sources": sources,
            "algorithmic_flops": flops, "algorithmic_bytes": bytes_,
            "avg_launch_us": round(t_head, 2), "launches_timed": in_step[1] if in_step[0] else solo[1],
            "avg_launch_us_solo": round(solo[0], 2) if solo[0] else None, "launches_timed_solo": solo[1],
            "frac_solo": frac(solo[0]),
            "hbm_gbps": round(gbps, 1), "hbm_frac": round(gbps / PEAK_HBM_GBPS, 4),
            "roof_times_us": {"mfma": round(mfma_us, 2), "hbm": round(hbm_us, 2)},
            "timing": "HIP events on the launching stream around each launch. avg_launch_us / frac_event_bracket: every launch of this "
                      "shape inside the step (event to event: the step keeps 4 streams busy, so it contains the co-runners' "
                      "share of the GPU and the time the launch queues behind them); avg_kernel_us_in_step / "
                      "frac_in_step_kernel: the kernel's own duration inside the step, from the committed rocprofv3 "
                      "trace of this command (sources); avg_launch_us_solo / frac_solo: 60 launches of the same kernel "
                      "alone on the GPU right after the timed steps and 150 warm-up rounds (settled clock)",
            "note": f"binding roof = {bound}: {unit_note}; HBM roof 8 TB/s on {bytes_ / 1e6:.1f} MB/launch"}


def _c3conv_array(batch, shapes, tensors, stats):
    from buctd_amd import _C
    arr = (_C.C3Conv * len(shapes))()
    for d, (cw, hh, ww), t in zip(arr, shapes, tensors):
        d.N, d.H, d.W, d.Ci, d.Co = batch, hh, ww, cw, cw
        d.x, d.wprep, d.y = t["x"].data_ptr(), t["w"].data_ptr(), t["y"].data_ptr()
        d.stats_acc = t["acc"].data_ptr() if stats else None
    return arr


def group_launch_probe(batch, rshape, device):
    """The DOMINANT launch of the step alone on the GPU: the two-member group launch of the forward / data-gradient 3x3
    convolutions (branches 0 and 1 of a HighResolutionModule: C @ HxW and 2C @ H/2xW/2, equal FLOPs) with the BatchNorm
    statistics accumulators, and the matching weight-gradient group launch (kernel + slab reduction) - 60 launches each after a
    warm-up, HIP events on the launching stream.  Also asks the library for the grid of the forward launch, by which the
    committed kernel trace is searched for the same launch INSIDE the step."""
    from buctd_amd import _C, ops
    lib = _C.lib()
    cw, hh, ww = rshape
    shapes = [(cw, hh, ww), (2 * cw, hh // 2, ww // 2)]
    main = torch.cuda.current_stream()
    ts = []
    for (c_, h_, w_) in shapes:
        wt = (torch.randn(c_, c_, 3, 3, device=device) * 0.05).contiguous(memory_format=torch.channels_last)
        ts.append({"x": torch.randn(batch, h_, w_, c_, device=device), "y": torch.empty(batch, h_, w_, c_, device=device),
                   "dw": torch.empty(c_, 3, 3, c_, device=device), "w": ops._conv3x3_prepared(wt, 0),
                   "acc": torch.zeros(int(lib.buctd_bn_acc_bytes(c_)) // 8, dtype=torch.int64, device=device)})
    arr = _c3conv_array(batch, shapes, ts, True)
    wgs = int(lib.buctd_conv3x3_bf16x6_group_workgroups(2, arr))
    wg = (_C.Wg3Conv * 2)()
    wss = []
    for it, (c_, h_, w_), t in zip(wg, shapes, ts):
        ws = torch.empty(int(lib.buctd_conv3x3_wgrad_bf16x6_group_workspace(2, batch, h_, w_, c_, c_)), dtype=torch.uint8, device=device)
        wss.append(ws)
        it.N, it.H, it.W, it.Ci, it.Co = batch, h_, w_, c_, c_
        it.x, it.dy, it.dw, it.accumulate = t["x"].data_ptr(), t["y"].data_ptr(), t["dw"].data_ptr(), 0
        it.workspace, it.workspace_bytes = ws.data_ptr(), ws.numel()

    def timed(fn, reps=60, warm=120):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        for _ in range(reps):
            fn()
        b.record(main)
        b.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    wg_wgs = int(lib.buctd_conv3x3_wgrad_bf16x6_group_workgroups(2, wg))
    fwd_us = timed(lambda: _C.check(lib.buctd_conv3x3_bf16x6_group(2, arr, main.cuda_stream), "group"))
    wg_us = timed(lambda: _C.check(lib.buctd_conv3x3_wgrad_bf16x6_group(2, wg, main.cuda_stream), "wgrad group"))
    return {"fwd_us": fwd_us, "wgrad_us": wg_us, "fwd_workgroups": wgs, "wgrad_workgroups": wg_wgs, "shapes": shapes, "launches": 60}


def roofline_objects(args, rshape, tag, in_step, solo, grp):
    """`roofline` = the dominant kernel of the step by time: conv3x3_x6_lean_kernel, the forward / data-gradient launches of the
    3x3 BasicBlock convolutions (29 % of the kernel time of a step; most of them two-member GROUP launches, the entry's
    launch); `roofline_wgrad` = the weight-gradient group launch of the same two convolutions (kernel + slab reduction);
    `roofline_single` = the single-convolution launches of the roofline shape (event-bracketed inside the step, as before).
    Durations INSIDE the step and the PMC traffic come from this round's committed profiles of this command (file + date
    quoted; null when absent or when the configuration differs from the profiled one); solo durations are measured live."""
    cw, hh, ww = rshape
    quoted = args.conv_math == "bf16x6" and args.batch == 32 and args.condition == "colored"
    trace = _profile_json(f"r06_in_step_kernel_us_{tag}.json") if quoted else None
    pmc = _profile_json("r06_pmc_traffic.json") if quoted and rshape == (48, 96, 72) else None
    k6 = MFMAS_PER_PRODUCT.get(args.conv_math, 6)
    peak = PEAK_FP32_MFMA_TFLOPS if args.conv_math == "fp32" else PEAK_BF16_MFMA_TFLOPS / k6
    n = args.batch
    flops1 = 2.0 * n * hh * ww * cw * cw * 9                      # one convolution (every HRNet branch costs the same)
    bytes1 = lambda c_, h_, w_: 4.0 * (2 * n * c_ * h_ * w_) + 4.0 * 9 * c_ * c_      # x in, y out (or x, dy in), filter

    def grid_us(name, grid):
        if not trace:
            return None
        rows = [v for k, v in trace["by_grid"].items() if name in k and (grid is None or k.endswith("|" + grid))]
        if not rows:
            return None
        return sum(v["ms_per_step"] for v in rows) / sum(v["calls_per_step"] for v in rows) * 1e3

    def obj(kernel, what, flops, bytes_, us_step, us_solo, launches, traffic, extra=None):
        mfma_us, hbm_us = flops / (peak * 1e6), bytes_ / (PEAK_HBM_GBPS * 1e3)
        bound = "mfma" if mfma_us >= hbm_us else "hbm"
        roof = mfma_us if bound == "mfma" else hbm_us
        t = us_step or us_solo
        if not t:
            return None
        o = {"kernel": kernel, "launch": what, "bound": bound,
             "achieved": round(flops / t / 1e6, 2) if bound == "mfma" else round(bytes_ / t / 1e3, 1),
             "peak": round(peak, 1) if bound == "mfma" else PEAK_HBM_GBPS, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
             "frac": round(roof / t, 4), "frac_source": "avg_kernel_us_in_step" if us_step else "avg_launch_us_solo",
             "avg_kernel_us_in_step": round(us_step, 1) if us_step else None,
             "frac_in_step_kernel": round(roof / us_step, 4) if us_step else None,
             "avg_launch_us_solo": round(us_solo, 1) if us_solo else None, "frac_solo": round(roof / us_solo, 4) if us_solo else None,
             "launches_timed_solo": launches,
             "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)" if traffic else None,
             "algorithmic_flops": flops, "algorithmic_bytes": bytes_,
             "roof_times_us": {"mfma": round(mfma_us, 2), "hbm": round(hbm_us, 2)},
             "hbm_gbps": round(bytes_ / t / 1e3, 1), "hbm_frac": round(bytes_ / t / 1e3 / PEAK_HBM_GBPS, 4),
             "sources": ({"avg_kernel_us_in_step": f"{trace['source']}, {trace['date']}"} if trace and us_step else {}) |
                        ({"traffic": f"{pmc['source']}, {pmc['date']}"} if pmc and traffic else {}),
             "note": (f"binding roof = {bound}: dense bf16 MFMA peak 2500 TFLOP/s / {k6} MFMAs per fp32 product = {peak:.1f} "
                      f"TFLOP/s-equivalent; HBM roof 8 TB/s on {bytes_ / 1e6:.1f} MB/launch.  frac = in-step kernel duration "
                      "from the committed rocprofv3 trace of this command where available, else the live solo launch; the chip "
                      "clocks ~1.8-1.9 GHz under a pure MFMA stream (the kernel with everything but its MFMAs removed reaches "
                      "0.75 of this nominal roof)")}
        if extra:
            o.update(extra)
        return o

    res = {}
    if args.conv_math != "bf16x6":
        # the other math modes keep the single-launch entries only
        def merge(a, b):
            m = a[1] + b[1]
            return ((a[0] * a[1] + b[0] * b[1]) / m, m) if m else (None, 0)
        main = roofline_entry(args.conv_math, args.batch, "fwd", merge(in_step["fwd"], in_step["dgrad"]),
                              merge(solo["fwd"], solo["dgrad"]), None, rshape, None, None)
        wg = roofline_entry(args.conv_math, args.batch, "wgrad", in_step["wgrad"], solo["wgrad"], None, rshape, None, None)
        if main is not None:
            res["roofline"] = main
        if wg is not None:
            res["roofline_wgrad"] = wg
        return res
    fam = 0 if cw == 48 else 1
    shapes = [(cw, hh, ww), (2 * cw, hh // 2, ww // 2)]
    b2 = sum(bytes1(*sh) for sh in shapes)
    g_grid = f"{grp['fwd_workgroups'] * 256},1,1" if grp and grp.get("fwd_workgroups") else None
    us_g = grid_us(f"conv3x3_x6_lean_kernel<{fam},", g_grid) if g_grid else None
    res["roofline"] = obj(
        f"conv3x3_x6_lean_kernel<{fam}, option set> (csrc/conv3x3_lean.hip): train-mode 3x3 bf16x6 convolution, forward and data gradient",
        f"group launch of two BasicBlock convolutions of one HighResolutionModule layer-step: {cw}->{cw} @{hh}x{ww} + "
        f"{2 * cw}->{2 * cw} @{hh // 2}x{ww // 2}, N={n}, with the BatchNorm accumulator epilogue; grid {g_grid}",
        2 * flops1, b2, us_g, grp["fwd_us"] if grp else None, grp["launches"] if grp else 0,
        round(pmc["fwd_group"]["bytes"]) if pmc and "fwd_group" in pmc else None,
        {"all_launches_in_step": ({"calls_per_step": round(sum(v["calls_per_step"] for k, v in trace["by_grid"].items() if "conv3x3_x6_lean_kernel" in k), 1),
                                   "ms_per_step": round(sum(v["ms_per_step"] for k, v in trace["by_grid"].items() if "conv3x3_x6_lean_kernel" in k), 3)}
                                  if trace else None)})
    w_grid = f"{grp['wgrad_workgroups'] * 256},1,1" if grp and grp.get("wgrad_workgroups", 0) > 0 else None
    us_wg = grid_us("conv3x3_wgrad_group_kernel", w_grid) if w_grid else None
    us_red = grid_us("wg3_reduce_group_kernel", None)
    if trace and us_red:
        # the slab reduction that follows the two-member weight-gradient launches: five chunk pairs (1 + 4)
        us_red = grid_us("wg3_reduce_group_kernel", "86016,5,1") or us_red
    res["roofline_wgrad"] = obj(
        f"conv3x3_wgrad_group_kernel<3, {3 if cw == 48 else 2}> + wg3_reduce_group_kernel (csrc/conv3x3_wgrad.hip)",
        f"weight gradients of the same two convolutions in one launch (grid {w_grid}) + their slab reduction, N={n}",
        2 * flops1, b2, (us_wg + us_red) if (us_wg and us_red) else None, grp["wgrad_us"] if grp else None,
        grp["launches"] if grp else 0, round(pmc["wgrad_group"]["bytes"]) if pmc and "wgrad_group" in pmc else None)

    def merge(a, b):
        m = a[1] + b[1]
        return ((a[0] * a[1] + b[0] * b[1]) / m, m) if m else (None, 0)
    P_ = n * (hh + 1) * (ww + 1) + ww + 1
    s_grid = f"{((((P_ + 447) // 448) + 7) // 8) * 8 * 256},1,1" if cw == 48 else None
    us_s = grid_us(f"conv3x3_x6_lean_kernel<{fam},", s_grid) if s_grid else None
    single = roofline_entry(args.conv_math, args.batch, "fwd", merge(in_step["fwd"], in_step["dgrad"]),
                            merge(solo["fwd"], solo["dgrad"]), round((pmc["fwd"]["bytes"] + pmc["dgrad"]["bytes"]) / 2) if pmc and "fwd" in pmc else None,
                            rshape, round(us_s, 1) if us_s else None,
                            {"avg_kernel_us_in_step": f"{trace['source']}, {trace['date']}"} if trace and us_s else None)
    if single is not None:
        single["kernel"] = (f"conv3x3_x6_lean_kernel<{fam}, option set>, single-convolution launches: 3x3 {cw}->{cw} @{hh}x{ww} N={n} "
                            "(HRNet branch 0; 448-position tiles, one round of 506 workgroups)")
        res["roofline_single"] = single
    for k in [k for k, v in res.items() if v is None]:
        del res[k]
    return res


def _host_cores(cap=32):
    """CPU threads this process may really use: scheduler affinity, clipped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def _cpu_baseline_worker(steps=2, batch=2):
    """The CPU oracle (plain-PyTorch restatement of the reference, pinned against it) timed on this box's host
    cores on a bounded sample of the same workload: CoAM-W48 384x288 train steps at batch 2."""
    sys.path.insert(0, ROOT)
    cores = _host_cores()
    torch.set_num_threads(cores)
    from oracle import cfg as ocfg, models as omodels, core as ocore, recipes
    c = ocfg.hrnet_cfg(48, 14, (288, 384), "pose_hrnet_coam", use_attention=True)
    torch.manual_seed(0)
    m = omodels.get_pose_net(c, is_train=True).train()
    x, joints = recipes.make_inputs(c, batch, 3, 3)
    tgt, wt = recipes.make_targets(c, joints, 4)
    opt = torch.optim.Adam(m.parameters(), lr=2e-3)
    crit = ocore.JointsMSELoss(True)

    def step():
        loss = crit(m(x), tgt, wt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    step()  # warm-up
    t0 = time.time()
    for _ in range(steps):
        step()
    dt = time.time() - t0
    return {"value": batch * steps / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} train steps of CoAM-W48 384x288 at batch {batch} (after 1 warm-up), torch CPU fp32, "
                      f"{cores} threads"}


def cpu_baseline(limit_s=300):
    """Runs the CPU leg in a child process (no GPU visible, hard time limit) so a starved host cannot stall the bench."""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", BUCTD_BENCH_CPU_WORKER="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True,
                           timeout=limit_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        note = "CPU oracle leg failed: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:200]
    except subprocess.TimeoutExpired:
        note = f"CPU oracle leg did not finish 3 batch-2 train steps within {limit_s} s on this host"
    return {"value": None, "unit": "images/s", "cores": _host_cores(), "kind": "port", "sample": note}


COCO_COLORS = [[127, 0, 255], [97, 46, 254], [67, 91, 252], [37, 134, 249], [7, 173, 245], [22, 206, 239],
               [52, 232, 231], [82, 248, 222], [112, 254, 212], [142, 249, 201], [172, 233, 188], [202, 207, 174],
               [232, 174, 159], [255, 135, 142], [255, 92, 126], [255, 47, 108], [255, 0, 90]]


def transpose_a6_cfg(batch):
    from buctd_amd.config import cfg as base, hrnet_extra
    c = base.clone()
    c.defrost()
    c.MODEL.NAME = "transpose_h"
    c.MODEL.NUM_JOINTS = 17
    c.MODEL.IMAGE_SIZE = [192, 256]
    c.MODEL.HEATMAP_SIZE = [48, 64]
    c.MODEL.SIGMA = 2
    c.MODEL.PRETRAINED = ""
    c.MODEL.CONDITIONAL_TOPDOWN = True
    c.MODEL.EXTRA = hrnet_extra(48, use_attention=True)
    c.DATASET.DATASET = "coco"
    c.DATASET.COLORED = True
    c.TEST.BATCH_SIZE_PER_GPU = batch
    c.freeze()
    return c


def emit(out, rank, world, in_group=False):
    """rank 0's ONE JSON line as the LAST line of the job's stdout.  RCCL writes a banner ("Librccl path : ...") through C stdio;
    on a pipe it stays in every rank's buffer until the process exits - behind the JSON line.  So: every rank flushes its C and
    Python buffers, the ranks meet, then rank 0 prints."""
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if world > 1 or in_group:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)


def bench_infer_c5(args, rank, world, device):
    """BASELINE config C5: BUCTD-TransPose-H-A6 256x192, 3x iterative-refinement inference, persons/s:
    one step = three chained passes, each forward -> device arg-max decode -> colored condition re-rendered from the
    decoded key points (crop coordinates) -> next forward, on 32 persons resident in HBM; eval mode."""
    from buctd_amd import models, ops
    cfg = transpose_a6_cfg(args.batch)
    torch.manual_seed(1234)
    net = models.transpose_h.get_pose_net(cfg, is_train=False).to(device).eval()
    w, h = cfg.MODEL.IMAGE_SIZE
    k = cfg.MODEL.NUM_JOINTS
    g = torch.Generator(device="cpu").manual_seed(7 + rank)
    rgb = torch.randn(args.batch, 3, h, w, generator=g).to(device)
    joints0 = (torch.rand(args.batch, k, 2, generator=g) * torch.tensor([w - 1.0, h - 1.0])).to(device).contiguous()
    colors = torch.tensor(COCO_COLORS[:k], dtype=torch.float32, device=device)
    x = torch.empty(args.batch, 6, h, w, device=device)
    x[:, :3] = rgb
    mha = {"pairs": [], "on": False}
    raw_mha = ops.mha_fwd

    def timed_mha(qk, v, scale=None):
        if not mha["on"]:
            return raw_mha(qk, v, scale)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = raw_mha(qk, v, scale)
        e1.record()
        mha["pairs"].append((e0, e1))
        return out

    ops.mha_fwd = timed_mha

    @torch.no_grad()
    def step():
        joints = joints0
        for _ in range(3):
            need = ops.lib().buctd_cond_render_workspace(args.batch, 3, h, w)
            ws = ops.workspace(need, device)
            ops.check(ops.lib().buctd_cond_render_into(ops.ptr(joints), 2, ops.ptr(colors), args.batch, k, 3, h, w, 0,
                                                       ops.C.c_void_p(x[:, 3:].data_ptr()), x.stride(0), ops.ptr(ws),
                                                       ws.numel(), ops.stream_ptr()), "cond_render_into")
            out = net(x)
            preds, _, _ = ops.argmax_decode(out)
            joints = ops.scale(preds, alpha=4.0)          # heat-map -> crop coordinates (stride 4)
        return joints

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    mha["on"] = not args.no_kernel_timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    mha["on"] = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        emit(None, rank, world)
        return
    persons = args.batch * world * args.steps
    out = {"metric": "persons/sec (3x iterative-refinement inference) BUCTD-TransPose-H-A6 256x192",
           "value": round(persons / dt, 3), "unit": "persons/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (3x3 convs and attention products bf16x6: operands split exactly into 3 bf16 pieces, 6 bf16 MFMAs "
                    "per product, fp32 accumulate - fp32-class)" if args.conv_math == "bf16x6" else "f32",
           "data": "synthetic (N(0,1) RGB; first condition from uniform key points, later ones from the decoded predictions)",
           "config": {"workload": "BUCTD-TransPose-H-A6 (transpose_h, W48 trunk, d_model 96+16, 6 encoder layers, "
                                  "T = 3072) 256x192 COCO-17kpt, eval: 3 chained passes per person (forward -> arg-max "
                                  "decode -> colored condition re-render -> forward)",
                      "global_batch": args.batch * world, "batch_per_gpu": args.batch, "input": "N x 6 x 256 x 192 fp32",
                      "params": sum(p.numel() for p in net.parameters()), "parallelism": f"dp{world} (replicas)",
                      "conv_math": args.conv_math, "forwards_per_step": 3}}
    if mha["pairs"]:
        us = 1e3 * sum(a.elapsed_time(b) for a, b in mha["pairs"]) / len(mha["pairs"])
        T, d = (h // 4) * (w // 4), 112
        flops = 4.0 * args.batch * T * T * d
        bytes_ = 4.0 * args.batch * T * d * 4          # q, k, v read + o written once
        tf = flops / us / 1e6
        x6 = args.conv_math == "bf16x6" and ops._MHA_X6
        peak = PEAK_BF16_MFMA_TFLOPS / 6 if x6 else PEAK_FP32_MFMA_TFLOPS
        pre = x6 and ops._MHA_PRESPLIT
        kname = ("mha_kv_split_kernel + mha_fwd_x6q_kernel" if pre else "mha_fwd_x6_kernel") if x6 else "mha_fwd_kernel"
        out["roofline"] = {"kernel": f"{kname}<7>: fused self-attention forward, "
                                     f"T={T} d={d} N={args.batch} (TransPose encoder layer)", "bound": "mfma",
                           "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                           "traffic": None, "algorithmic_flops": flops, "algorithmic_bytes": bytes_,
                           "avg_launch_us": round(us, 1), "launches_timed": len(mha["pairs"]),
                           "hbm_frac": round(bytes_ / us / 1e3 / PEAK_HBM_GBPS, 4),
                           "timing": "HIP events on the launching stream around every call in the timed steps "
                                     "(single stream: no co-runners)" +
                                     ("; a call = the K/V pre-split launch (6 B/element image, ~45 us) + the attention launch"
                                      if pre else ""),
                           "note": ("bf16x6: dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMAs per fp32 product = 416.7 TFLOP/s-"
                                    "equivalent" if x6 else "exact fp32 MFMA (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s peak)") +
                                   " binds: 4 T^2 d FLOP against 4 T d floats of HBM traffic per image"}
    emit(out, rank, world)


def self_launch(n):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def first_touch(device):
    """Write a large block of device memory once and hand it back to the driver before anything is measured.  Device memory
    that no process has used since a box came up costs 10-50 ms per GB on its first touch (0.4 ms later; measured on the
    gpurun boxes, scratch/first_run_probe2.py), and the caching allocator still maps a few new segments in the first dozen
    steps after the warm-up: as the first process on a fresh box a short run measured that instead of the step.  Nothing of
    the measured work moves; several ranks on one device (tests) skip it."""
    if os.environ.get("BUCTD_BENCH_NO_FIRST_TOUCH") == "1" or os.environ.get("BUCTD_SINGLE_DEVICE") == "1":
        return
    free = torch.cuda.mem_get_info(device)[0]
    n = min(int(free * 0.5), 128 << 30) >> 30
    blocks = [torch.empty(1 << 30, dtype=torch.uint8, device=device).fill_(0) for _ in range(n)]
    torch.cuda.synchronize(device)
    del blocks
    torch.cuda.empty_cache()


def main():
    if os.environ.get("BUCTD_BENCH_CPU_WORKER") == "1":
        print(json.dumps(_cpu_baseline_worker()))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="train_c4", choices=["train_c4", "train_c3", "train_c2", "infer_c5"],
                    help="train_c4 (default): the headline metric, CoAM-W48 384x288 training images/s; train_c3 / train_c2: "
                         "the preNet configs (HRNet-W48 384x288, HRNet-W32 256x192); infer_c5: TransPose-H-A6 256x192 3x "
                         "iterative-refinement inference persons/s (BASELINE config C5)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (scripts: 32, W48 YAML: 24)")
    ap.add_argument("--conv-math", default="bf16x6",
                    choices=["fp32", "bf16x6", "bf16x3"],
                    help="how the 3x3/s1 convs (fwd, dgrad, wgrad) are computed: bf16x6 (default) = fp32-class, exact "
                         "3-way bf16 split, 6 MFMAs per product; fp32 = v_mfma_f32_16x16x4_f32; bf16x3 = optional "
                         "reduced precision (~2^-16 per product) - not a headline mode")
    ap.add_argument("--condition", default="colored", choices=["colored", "mono"],
                    help="train_c4 only: colored = the CrowdPose recipe (6-channel input, the headline); mono = one white "
                         "condition channel (4-channel input, north_star's literal variant)")
    ap.add_argument("--one-rank-exchange", action="store_true",
                    help="--gpus 1 only: initialise an RCCL process group of ONE rank and run the whole gradient exchange "
                         "(buckets, communication stream, collectives, 1 branch stream) - the per-GPU cost of the data-parallel "
                         "machinery without a second GPU; not the headline configuration")
    ap.add_argument("--step-graph", action="store_true",
                    help="--gpus 1, models without train-mode dropout (train_c2 / train_c3): forward + loss + backward are captured "
                         "once as a hipGraph (engine.StepGraph) and replayed per step; the same kernels, bit-identical results, one "
                         "hipGraphLaunch instead of ~1000 host-side launches - the host-bound C2 step becomes GPU-bound")
    ap.add_argument("--host-input", default="resident", choices=["resident", "copy", "prefetch"],
                    help="where the batch is when a step starts: resident = in HBM (the contract of `value`); copy = pinned host "
                         "memory, copied on the compute stream at the top of the step as the reference loop does; prefetch = pinned "
                         "host memory through core.function.DevicePrefetch (what train() does: the next batch's copies beside the "
                         "current step).  copy / prefetch give the PCIe-inclusive rate quoted in DESIGN.md - never the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, torch.distributed.run on this node,
        # rendezvous on 127.0.0.1) and pass rank 0's JSON line through
        raise SystemExit(self_launch(args.gpus))

    from buctd_amd import engine, models, ops
    from buctd_amd.core.function import _DeferredStats, AverageMeter
    from buctd_amd.core.loss import JointsMSELoss

    rank, world, device = engine.init_distributed()
    if device.type != "cuda":
        raise SystemExit("bench.py needs a ROCm device: buctd_amd has no CPU path")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    ops.set_conv_math(args.conv_math)
    first_touch(device)
    if args.workload == "infer_c5":
        bench_infer_c5(args, rank, world, device)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    make_cfg, module, metric, describe, rshape = TRAIN_WORKLOADS[args.workload]
    cfg = make_cfg(args.batch)
    n_in = 6
    if args.condition == "mono":
        # north_star's literal "synthetic 384x288 4-channel crops": RGB + one mono condition heat map (pose_hrnet_coam.py:750-757)
        if args.workload != "train_c4":
            raise SystemExit("--condition mono is a variant of train_c4 (the preNet recipes stack a colored condition)")
        cfg = coam_w48_cfg(args.batch, colored=False)
        describe = describe.replace("colored condition", "mono condition (4-channel input)")
        n_in = 4
    torch.manual_seed(1234)
    ops.manual_seed(1234 + rank)
    net = getattr(models, module).get_pose_net(cfg, is_train=True).to(device)
    if args.one_rank_exchange:
        if world != 1:
            raise SystemExit("--one-rank-exchange is a --gpus 1 measurement")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        engine.reserve_streams(device, data_parallel=True)     # what engine.init_distributed does under --gpus N
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        describe += " [one-rank RCCL exchange on]"
    model = engine.DataParallel(net, exchange_in_world_of_one=args.one_rank_exchange)
    optimizer = engine.get_optimizer(cfg, model)
    if world == 1:
        model.flatten()
    criterion = JointsMSELoss(cfg.LOSS.USE_TARGET_WEIGHT)
    x, target, weight = synthetic_batch(cfg, args.batch, device, seed=100 + rank)
    timer = KernelTimer(args.batch, rshape)
    losses, acc = AverageMeter(), AverageMeter()
    model.train()
    state = {"pending": None}

    graph_step = None
    if args.step_graph:
        if world != 1 or args.one_rank_exchange:
            raise SystemExit("--step-graph captures the single-GPU step (the gradient exchange is launched from host callbacks)")
        graph_step = engine.StepGraph(model, criterion, optimizer, warmup=2)
        describe += " [forward + loss + backward replayed from a hipGraph]"

    feed = None
    if args.host_input != "resident":
        from buctd_amd.core.function import DevicePrefetch
        host = tuple(t.cpu().pin_memory() for t in (x, target, weight))
        describe += f" [batch from pinned host memory every step: {args.host_input}]"

        def endless():
            while True:
                yield host + ({},)
        feed = iter(DevicePrefetch(endless(), True, device)) if args.host_input == "prefetch" else None

    def step():
        if args.host_input == "prefetch":
            x_, target_, weight_, _ = next(feed)
        elif args.host_input == "copy":
            x_, target_, weight_ = (t.cuda(non_blocking=True) for t in host)
        else:
            x_, target_, weight_ = x, target, weight
        if graph_step is not None:
            out, loss = graph_step(x_, target_, weight_)
        else:
            out = model(x_)
            loss = criterion(out, target_, weight_)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        if state["pending"] is not None:
            state["pending"].resolve(losses, acc)
        state["pending"] = _DeferredStats(loss, out, target_, args.batch)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    while graph_step is not None and graph_step.replays == 0:
        step()          # eager settling steps + the capture, in front of the warm-up steps the command line asks for
    for _ in range(args.warmup):
        step()
    fence()
    # the timed region runs the engine exactly as shipped: no per-launch events, no step-by-step blocks.  ONE event per step
    # on the main stream (read after the fence) gives the spread of the steps; the clock / power probe is a child process
    # started in the middle of the region (it reads the SMU while the GPU is under THIS load) and collected afterwards.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    probe = None
    t0 = time.perf_counter()
    for k in range(args.steps):
        marks[k].record()
        if k == args.steps // 2 and rank == 0:
            probe = _box_probe_start()
        step()
    marks[args.steps].record()
    fence()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    box = _box_probe_collect(probe) if rank == 0 else None
    # separate short pass for the in-step kernel durations: HIP events on the launching stream around the launches of the
    # roofline shape (every 4th BasicBlock of that shape goes through the step-by-step path so that its launches can be
    # bracketed); not part of `value`
    in_step = {k: (None, 0) for k in ("fwd", "dgrad", "wgrad")}
    if not args.no_kernel_timer:
        graph_step = None       # the bracketed launches below are eager launches
        install_timer(timer)
        step()
        fence()
        timer.enabled = True
        for _ in range(max(2, min(4, args.steps))):
            step()
        fence()
        timer.enabled = False
        in_step = {k: timer.mean_us(k) for k in ("fwd", "dgrad", "wgrad")}
    # exposed gradient-exchange time per step (N > 1): a pass with the device drained in front of and behind the exchange
    comm_ms = None
    if world > 1 or args.one_rank_exchange:
        ts = []
        raw_sync = model.sync_gradients

        def timed_sync():
            torch.cuda.synchronize()
            a = time.perf_counter()
            r = raw_sync()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - a)
            return r
        optimizer.grad_sync = timed_sync
        for _ in range(3):
            step()
        fence()
        optimizer.grad_sync = raw_sync
        t = torch.tensor([sum(ts) / len(ts)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm_ms = round(1e3 * float(t.item()), 3)
        # one more step with timing events around every bucket's exchange on the communication stream: when (ms after the
        # backward pass started on the GPU) each bucket's all-reduce began and ended, against the end of the backward pass -
        # what is hidden behind the backward and what sticks out shows at a glance
        model.bucket_trace = []
        bw0, bw1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        outp = model(x)
        lossp = criterion(outp, target, weight)
        optimizer.zero_grad()
        bw0.record()
        lossp.backward()
        bw1.record()
        optimizer.step()
        fence()
        bucket_rows = [{"bucket": i, "mb": round(nb / 2 ** 20, 1), "start_ms": round(bw0.elapsed_time(e0), 2),
                        "end_ms": round(bw0.elapsed_time(e1), 2)} for i, nb, e0, e1 in model.bucket_trace]
        backward_ms = round(bw0.elapsed_time(bw1), 2)
        model.bucket_trace = None
    solo = {k: (None, 0) for k in in_step}
    grp = None
    if not args.no_kernel_timer and rank == 0:
        # the same kernels alone on the GPU: 60 launches each (after 150 warm-up rounds) on a stage-4-branch-0 sized activation with one of the
        # model's own 48 -> 48 filters (forward with the BN-statistics epilogue, as in the step)
        timer.reset()
        wsel = next(p for p in net.parameters() if tuple(p.shape) == (rshape[0], rshape[0], 3, 3))
        xs = torch.randn(args.batch, rshape[1], rshape[2], rshape[0], device=device)
        dys = torch.randn(args.batch, rshape[1], rshape[2], rshape[0], device=device)
        gw = torch.empty_like(wsel)
        for it in range(210):                 # 150 untimed rounds first: the chip ramps its clock over ~20-30 ms of load
            timer.enabled = it >= 150
            ops.conv_fwd(xs, wsel, None, 1, 1, stats=True)
            ops.conv_dgrad(dys, wsel, tuple(xs.shape), 1, 1)
            ops.conv_wgrad(xs, dys, wsel, 1, 1, out=gw, accumulate=0)
        torch.cuda.synchronize()
        timer.enabled = False
        solo = {k: timer.mean_us(k) for k in ("fwd", "dgrad", "wgrad")}
        if args.conv_math == "bf16x6":
            grp = group_launch_probe(args.batch, rshape, device)
    state["pending"].resolve(losses, acc)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        global_batch = args.batch * world
        value = global_batch * args.steps / dt
        out = {
            "metric": metric, "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x6": "f32 (3x3 convs: operands split exactly into 3 bf16 pieces, 6 bf16 MFMAs "
                                "per product, fp32 accumulate - fp32-class)",
                      "bf16x3": "f32 with REDUCED-PRECISION 3x3 convs (2 bf16 pieces, ~2^-16 per product) - not a "
                                "headline mode"}[args.conv_math],
            "data": "synthetic (N(0,1) RGB; condition = device pose synthesis of uniform ground-truth key points; Gaussian targets)",
            "config": {"workload": describe + " full train step: fwd + JointsMSE + bwd + grad all-reduce + Adam + arg-max "
                                              "accuracy decode",
                       "global_batch": global_batch, "batch_per_gpu": args.batch,
                       "input": f"N x {n_in} x {cfg.MODEL.IMAGE_SIZE[1]} x {cfg.MODEL.IMAGE_SIZE[0]} fp32",
                       "params": sum(p.numel() for p in net.parameters()), "parallelism": f"dp{world}",
                       # HIP streams of this rank: main + branch / weight-gradient streams (+ communication under --gpus N)
                       "hip_streams": 1 + len(ops.compute_streams(device)) + (1 if world > 1 or args.one_rank_exchange else 0),
                       "conv_math": args.conv_math, "loss": round(losses.avg, 6), "step_graph": bool(args.step_graph), "host_input": args.host_input},
            # is a slow line a slow box or a slow build?  the spread of the timed steps (GPU time between one event per step
            # on the main stream) and what the SMU reported in the middle of the timed region
            "step_ms": {"min": round(step_ms[0], 3), "median": round(step_ms[len(step_ms) // 2], 3), "max": round(step_ms[-1], 3)},
            "box": box,
        }
        if comm_ms is not None:
            out["allreduce_exposed_ms_per_step"] = comm_ms
            out["allreduce"] = ("flat fp32 gradient arena in ~48 MB buckets (tensors >= a bucket travel alone), one "
                                "RCCL all-reduce per bucket on a highest-priority communication stream, launched from the "
                                "backward pass as soon as the bucket's last gradient kernel is enqueued")
            out["allreduce_buckets"] = {"backward_ms": backward_ms, "rank0": bucket_rows,
                                        "note": "start / end of every bucket's all-reduce on the communication stream, ms after "
                                                "the backward pass started on the GPU (one extra step behind the timed region)"}
        tag = {"train_c4": "c4", "train_c3": "c3", "train_c2": "c2"}[args.workload]
        out["serialised"] = _serialised_census(tag)
        out.update(roofline_objects(args, rshape, tag, in_step, solo, grp))
        if world == 1 and not args.no_cpu_baseline and args.workload == "train_c4":
            out["cpu_baseline"] = cpu_baseline()
    emit(out, rank, world, in_group=args.one_rank_exchange)
    if world > 1 or args.one_rank_exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
