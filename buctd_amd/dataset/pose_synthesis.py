"""Generative pose synthesis on the device - drop-in for reference lib/dataset/pose_synthesis.py:779-817
(synthesize_pose and its coco / crowdpose variants 234-775): the training-time condition of every "generative
sampling" recipe (DATASET.SYNTHESIS_POSE True).  synthesize_pose keeps the reference signature for one person;
synthesize_pose_batch does a whole batch in one kernel launch (one wavefront per person and joint)."""
import ctypes as C

import numpy as np
import torch

from .._C import check, lib, ptr, stream_ptr

_MAXK = 32


class _Tables(C.Structure):
    _fields_ = [("sigmas", C.c_double * _MAXK), ("pair", C.c_int * _MAXK), ("jitter_cls", C.c_int * _MAXK),
                ("miss_cls", C.c_int * _MAXK), ("inv_cls", C.c_int * _MAXK), ("swap_cls", C.c_int * _MAXK),
                ("jitter_p", (C.c_double * 3) * 2), ("miss_p", (C.c_double * 3) * 3), ("inv_p", C.c_double * 3),
                ("swap_p", (C.c_double * 3) * 2), ("out_vis", C.c_double)]


def _joint_classes(dataset, k):
    """Per-joint probability classes of the reference's if / elif ladders (pose_synthesis.py:56-72, 91-112, 146-152,
    176-190 for crowdpose; 561-575, 597-618, 651-657, 680-694 for coco).  Crowdpose joints 12 and 13 fall through the
    jitter ladder and inherit what joint 11 left in the variable: class 0."""
    if dataset == "coco":
        jit = [0 if (j == 0 or 13 <= j <= 16) else (1 if 1 <= j <= 10 else 2) for j in range(k)]
        miss = [0 if j <= 4 else (1 if j in (5, 6, 15, 16) else 2) for j in range(k)]
        inv = [0 if j <= 4 else (1 if 5 <= j <= 10 else 2) for j in range(k)]
        swap = list(inv)
        sym = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16)]
        sig = [.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]
        vis = 1.0
    elif dataset == "crowdpose":
        jit = [0 if 8 <= j <= 11 else (1 if j <= 5 else (2 if j <= 7 else 0)) for j in range(k)]
        miss = [0 if j in (12, 13) else (1 if j in (0, 1, 8, 9) else 2) for j in range(k)]
        inv = [0 if j >= 12 else (1 if j <= 5 else 2) for j in range(k)]
        swap = [0 if j in (12, 13) else (1 if j <= 5 else 2) for j in range(k)]
        sym = [(0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (10, 11)]
        sig = [.79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89, .79, .79]
        vis = 0.0
    else:
        raise NotImplementedError("pose synthesis: the BUCTD recipes use the coco and crowdpose variants")
    return jit, miss, inv, swap, sym, np.array(sig) / 10.0, vis


def make_tables(dataset, num_joints):
    jit, miss, inv, swap, sym, sig, vis = _joint_classes(dataset, num_joints)
    if len(sig) != num_joints or num_joints > _MAXK:
        raise ValueError(f"{dataset} has {len(sig)} key points, MODEL.NUM_JOINTS is {num_joints}")
    t = _Tables()
    pair = [-1] * _MAXK
    for a, b in sym:
        pair[a], pair[b] = b, a
    for j in range(_MAXK):
        t.pair[j] = pair[j]
        t.sigmas[j] = float(sig[j]) if j < num_joints else 0.0
        t.jitter_cls[j] = jit[j] if j < num_joints else 0
        t.miss_cls[j] = miss[j] if j < num_joints else 0
        t.inv_cls[j] = inv[j] if j < num_joints else 0
        t.swap_cls[j] = swap[j] if j < num_joints else 0
    for r, row in enumerate([[0.15, 0.20, 0.25], [0.10, 0.15, 0.20]]):
        for c, v in enumerate(row):
            t.jitter_p[r][c] = v
    for r, row in enumerate([[0.15, 0.20, 0.25], [0.10, 0.13, 0.15], [0.02, 0.05, 0.10]]):
        for c, v in enumerate(row):
            t.miss_p[r][c] = v
    for c, v in enumerate([0.01, 0.03, 0.06]):
        t.inv_p[c] = v
    for r, row in enumerate([[0.02, 0.15, 0.10], [0.01, 0.06, 0.03]]):
        for c, v in enumerate(row):
            t.swap_p[r][c] = v
    t.out_vis = vis
    return t


def synthesize_pose_batch(dataset, joints, estimated_joints, near_joints, area, num_overlap, seed, device=None):
    """joints, estimated_joints [B, K, 3]; near_joints [B, M, K, 3] (pad absent neighbours with visibility 0);
    area [B]; num_overlap [B].  numpy arrays or tensors; returns a float64 device tensor [B, K, 3]."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device

    def dev64(a):
        return torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a, dtype=torch.float64).to(dev).contiguous()
    J, E, A = dev64(joints), dev64(estimated_joints), dev64(area)
    B, K = J.shape[0], J.shape[1]
    NR = dev64(near_joints) if near_joints is not None and len(near_joints) else None
    M = 0 if NR is None else NR.shape[1]
    ov = torch.as_tensor(np.asarray(num_overlap), dtype=torch.int32).to(dev).contiguous()
    out = torch.empty((B, K, 3), dtype=torch.float64, device=dev)
    tables = make_tables(dataset, K)
    check(lib().buctd_synthesize_pose(C.byref(tables), ptr(J), ptr(E), ptr(NR), ptr(A), ptr(ov), B, K, M,
                                      C.c_ulonglong(int(seed) & 0xFFFFFFFFFFFFFFFF), ptr(out), stream_ptr()),
          "synthesize_pose")
    return out


_calls = {"n": 0}


def synthesize_pose(cfg, joints, estimated_joints, near_joints, area, num_overlap, seed=None):
    """Reference signature (one person, numpy in / numpy out).  seed None: torch's seed mixed with a call counter."""
    if seed is None:
        _calls["n"] += 1
        seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + _calls["n"]) & 0xFFFFFFFFFFFFFFFF
    near = np.asarray(near_joints, dtype=np.float64).reshape(-1, cfg.MODEL.NUM_JOINTS, 3)
    out = synthesize_pose_batch(cfg.DATASET.DATASET, np.asarray(joints)[None], np.asarray(estimated_joints)[None],
                                near[None] if near.shape[0] else None, [area], [num_overlap], seed)
    return out[0].cpu().numpy()
