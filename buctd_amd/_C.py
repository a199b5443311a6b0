"""ctypes binding of libbuctd_hip.so (the C ABI declared in include/buctd_hip.h).

The library is the only compute back end of the package: there is no CPU or eager
PyTorch fallback.  Importing this module without the built library, or calling an op
with tensors that are not on a ROCm device, raises.

torch is imported first on purpose: the shared object depends on libamdhip64.so.7 by
SONAME, and the dynamic loader then binds it to the HIP runtime PyTorch already loaded,
so device pointers and streams are shared with the caching allocator.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbuctd_hip.so")
# The product loader reads no environment.  Experiments that need another build of the same sources go through
# scratch/run_alt.py, which sets LIB_PATH before the first lib() call.


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "Ci", "Co", "R", "S", "stride", "pad", "Ho", "Wo")]


class BasicBlockDesc(C.Structure):     # mirrors buctd_basic_block
    _fields_ = ([(n, C.c_int) for n in ("N", "H", "W", "C")] +
                [(n, C.c_void_p) for n in ("x", "w1_fwd", "w2_fwd", "w1_bwd", "w2_bwd", "gamma1", "beta1", "gamma2", "beta2",
                                           "running_mean1", "running_var1", "running_mean2", "running_var2")] +
                [(n, C.c_float) for n in ("eps1", "momentum1", "eps2", "momentum2")] +
                [(n, C.c_void_p) for n in ("z1", "z2", "y", "acc", "stat")])


class BasicBlockGrads(C.Structure):    # mirrors buctd_basic_block_grads
    _fields_ = ([(n, C.c_void_p) for n in ("dy", "dz2", "dres", "dy1", "dz1", "dx", "dw1", "dw2", "dgamma1", "dbeta1",
                                           "dgamma2", "dbeta2")] +
                [(n, C.c_int) for n in ("acc_w1", "acc_w2", "acc_bn1", "acc_bn2")] +
                [("bn_acc", C.c_void_p), ("wg_ws", C.c_void_p), ("wg_ws_bytes", C.c_size_t)])


class BnAccIn(C.Structure):            # mirrors buctd_bn_acc_in
    _fields_ = [("acc", C.c_void_p), ("rows", C.c_long), ("eps", C.c_float), ("momentum", C.c_float),
                ("mean_out", C.c_void_p), ("invstd_out", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p)]


class C3Conv(C.Structure):             # mirrors buctd_c3_conv
    _fields_ = ([(n, C.c_int) for n in ("N", "H", "W", "Ci", "Co")] +
                [(n, C.c_void_p) for n in ("x", "wprep", "residual")] + [("relu", C.c_int)] +
                [(n, C.c_void_p) for n in ("y", "stats_acc")] + [("in_bn", C.POINTER(BnAccIn))] +
                [(n, C.c_void_p) for n in ("in_gamma", "in_beta")] + [("in_relu", C.c_int)] +
                [(n, C.c_void_p) for n in ("bn_z", "bn_y", "bn_mean", "bn_invstd", "bn_gamma", "bn_beta", "bn_acc")])


class C3ConvEval(C.Structure):         # mirrors buctd_c3_conv_eval
    _fields_ = ([(n, C.c_int) for n in ("N", "H", "W", "Ci", "Co")] +
                [(n, C.c_void_p) for n in ("x", "wprep", "scale", "shift", "residual")] + [("relu", C.c_int), ("y", C.c_void_p)])


class BnApplyItem(C.Structure):        # mirrors buctd_bn_apply_item
    _fields_ = [("z", C.c_void_p), ("st", BnAccIn), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("residual", C.c_void_p),
                ("relu", C.c_int), ("y", C.c_void_p), ("rows", C.c_long), ("C", C.c_int)]


class BnBwdItem(C.Structure):          # mirrors buctd_bn_bwd_item
    _fields_ = ([(n, C.c_void_p) for n in ("dy", "y", "z", "mean", "invstd", "gamma", "beta")] +
                [("relu", C.c_int), ("rows", C.c_long), ("C", C.c_int)] +
                [(n, C.c_void_p) for n in ("dz", "dres", "dgamma", "dbeta")] +
                [("accumulate", C.c_int), ("acc", C.c_void_p), ("acc_ready", C.c_int)])


class Wg3Conv(C.Structure):            # mirrors buctd_wg3_conv
    _fields_ = ([(n, C.c_int) for n in ("N", "H", "W", "Ci", "Co")] +
                [(n, C.c_void_p) for n in ("x", "dy", "dw")] + [("accumulate", C.c_int)] +
                [(n, C.c_void_p) for n in ("x_mean", "x_invstd", "x_gamma", "x_beta")] + [("x_relu", C.c_int)] +
                [("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)])


class MatmulDesc(C.Structure):
    _fields_ = [
        ("batch", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a_layout", C.c_int), ("b_layout", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
        ("stride_a", C.c_long), ("stride_b", C.c_long), ("stride_c", C.c_long),
        ("Kc", C.c_int), ("group_stride_a", C.c_long), ("group_stride_bk", C.c_long),
        ("Nc", C.c_int), ("group_stride_bn", C.c_long), ("group_stride_c", C.c_long),
        ("alpha", C.c_float), ("bias_axis", C.c_int),
    ]


_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float
_U64 = C.c_uint64
_SZ = C.c_size_t
_PD = C.POINTER(ConvDesc)
_PM = C.POINTER(MatmulDesc)
_PI = C.POINTER(C.c_int)

# name -> (restype, argtypes); mirrors include/buctd_hip.h one to one
SIGNATURES = {
    "buctd_version": (_I, []),
    "buctd_last_error": (C.c_char_p, []),
    "buctd_conv2d_fwd": (_I, [_PD, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "buctd_conv2d_fwd_thin": (_I, [_PD]),
    "buctd_conv2d_dgrad": (_I, [_PD, _P, _P, _P, _P, _P, _P]),
    "buctd_conv2d_stats_groups": (_I, [_PD, _I, _PI, _PI]),
    "buctd_conv2d_wgrad_workspace": (_SZ, [_PD]),
    "buctd_conv2d_wgrad": (_I, [_PD, _P, _P, _P, _I, _P, _SZ, _P]),
    "buctd_matmul_workspace": (_SZ, [_PM]),
    "buctd_matmul": (_I, [_PM, _P, _P, _P, _P, _P, _SZ, _P]),
    "buctd_bn_finalize": (_I, [_P, _P, _I, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P]),
    "buctd_conv3x3_bf16x3_supported": (_I, [_I, _I, _I, _I, _I]),
    "buctd_conv3x3_bf16x3_stats_groups": (_I, [_I, _I, _I, _I, _I, _PI, _PI]),
    "buctd_conv3x3_bf16x3_prep_bytes": (_SZ, [_I, _I, _I]),
    "buctd_conv3x3_bf16x3_prep": (_I, [_I, _I, _P, _I, _P, _P]),
    "buctd_conv3x3_bf16x3_prep_batched": (_I, [_P, _I, _L, _P]),
    "buctd_conv3x3_bf16x3": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "buctd_conv3x3_wgrad_bf16x3_supported": (_I, [_I, _I, _I, _I, _I]),
    "buctd_conv3x3_wgrad_bf16x3_workspace": (_SZ, [_I, _I, _I, _I, _I]),
    "buctd_conv3x3_wgrad_bf16x3": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _SZ, _P]),
    "buctd_conv3x3_bf16x6_supported": (_I, [_I, _I, _I, _I, _I]),
    "buctd_conv3x3_bf16x6_stats_groups": (_I, [_I, _I, _I, _I, _I, _PI, _PI]),
    "buctd_conv3x3_bf16x6_prep_bytes": (_SZ, [_I, _I, _I]),
    "buctd_conv3x3_bf16x6_prep": (_I, [_I, _I, _P, _I, _P, _P]),
    "buctd_conv3x3_bf16x6": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "buctd_conv3x3_bf16x6_bnin": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "buctd_conv3x3_bf16x6_bnstat": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "buctd_bn_bwd_from_partials": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _P, _I, _P, _P, _P, _P, _I, _P, _SZ, _P]),
    "buctd_conv3x3_wgrad_bf16x6_bnin": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _SZ, _P]),
    "buctd_conv3x3_wgrad_bf16x6_supported": (_I, [_I, _I, _I, _I, _I]),
    "buctd_conv3x3_wgrad_bf16x6_workspace": (_SZ, [_I, _I, _I, _I, _I]),
    "buctd_conv3x3_wgrad_bf16x6": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _SZ, _P]),
    "buctd_bn_stats": (_I, [_P, _L, _I, _P, _PI, _PI, _P]),
    "buctd_bn_stats_groups": (_I, [_L, _I, _PI, _PI]),
    "buctd_bn_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _L, _I, _P]),
    "buctd_bn_bwd_workspace": (_SZ, [_L, _I]),
    "buctd_bn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P, _I, _P, _SZ, _P]),
    "buctd_bn_fold": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "buctd_bn_acc_bytes": (_SZ, [_I]),
    "buctd_bn_apply_acc": (_I, [_P, C.POINTER(BnAccIn), _P, _P, _P, _I, _P, _L, _I, _P]),
    "buctd_bn_bwd_acc": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P, _I, _P, _I, _P]),
    "buctd_conv3x3_bf16x6_acc": (_I, [_I] * 5 + [_P, _P, _P, _I, _P, _P, C.POINTER(BnAccIn), _P, _P, _I, _P]),
    "buctd_conv3x3_bf16x6_bnstat_acc": (_I, [_I] * 5 + [_P] * 12),
    "buctd_gconv_x6_fwd_acc": (_I, [_I] * 6 + [_P] * 6),
    "buctd_conv3x3_bf16x6_group": (_I, [_I, C.POINTER(C3Conv), _P]),
    "buctd_conv3x3_bf16x6_group_eval": (_I, [_I, C.POINTER(C3ConvEval), _P]),
    "buctd_conv3x3_bf16x6_persistent": (_I, [_I]),
    "buctd_conv3x3_bf16x6_group_workgroups": (_I, [_I, C.POINTER(C3Conv)]),
    "buctd_gconv_wgrad_x6_supported": (_I, [_I] * 6),
    "buctd_gconv_wgrad_x6_workspace": (_SZ, [_I] * 6),
    "buctd_gconv_wgrad_x6": (_I, [_I] * 6 + [_P, _P, _P, _I, _P, _SZ, _P]),
    "buctd_bn_apply_acc_group": (_I, [_I, C.POINTER(BnApplyItem), _P]),
    "buctd_bn_bwd_acc_group": (_I, [_I, C.POINTER(BnBwdItem), _P]),
    "buctd_conv3x3_wgrad_bf16x6_group_workspace": (_SZ, [_I] * 6),
    "buctd_conv3x3_wgrad_bf16x6_group": (_I, [_I, C.POINTER(Wg3Conv), _P]),
    "buctd_conv3x3_wgrad_bf16x6_group_workgroups": (_I, [_I, C.POINTER(Wg3Conv)]),
    "buctd_stream_fork": (_I, [_P, _P]),
    "buctd_basic_branches_fwd_train": (_I, [_I, _I, C.POINTER(BasicBlockDesc), _P]),
    "buctd_basic_branches_bwd": (_I, [_I, _I, C.POINTER(BasicBlockDesc), C.POINTER(BasicBlockGrads), _P, _P]),
    "buctd_basic_block_fwd_train": (_I, [C.POINTER(BasicBlockDesc), _P]),
    "buctd_basic_block_bwd": (_I, [C.POINTER(BasicBlockDesc), C.POINTER(BasicBlockGrads), _P, _P]),
    "buctd_basic_chain_fwd_train": (_I, [_I, C.POINTER(BasicBlockDesc), _P]),
    "buctd_basic_chain_bwd": (_I, [_I, C.POINTER(BasicBlockDesc), C.POINTER(BasicBlockGrads), _P, _P]),
    "buctd_x6_image_dims": (_I, [_I, _I, _I, _P, _P]),
    "buctd_x6_image_bytes": (C.c_size_t, [_I, _I, _I]),
    "buctd_x6_image": (_I, [_P, _I, _I, _I, _L, _L, _I, _L, _L, _I, _P, _P]),
    "buctd_x6_gemm": (_I, [_I, _I, _I, _P, _P, _P, _I, _F, _P, _L, _I, _L, _P]),
    "buctd_add": (_I, [_P, _P, _P, _L, _I, _P]),
    "buctd_add_n": (_I, [C.POINTER(_P), _I, _P, _L, _P]),
    "buctd_mul": (_I, [_P, _P, _P, _L, _P]),
    "buctd_scale": (_I, [_P, _P, _F, _P, _L, _P]),
    "buctd_copy_channels": (_I, [_P, _L, _I, _I, _P, _I, _I, _I, _P]),
    "buctd_add_bcast": (_I, [_P, _P, _P, _L, _L, _P]),
    "buctd_relu_bwd": (_I, [_P, _P, _P, _L, _P]),
    "buctd_colsum_workspace": (_SZ, [_L, _I]),
    "buctd_colsum": (_I, [_P, _L, _I, _P, _I, _P, _SZ, _P]),
    "buctd_nchw_to_nhwc": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "buctd_nhwc_to_nchw": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "buctd_fuse_sum": (_I, [C.POINTER(_P), _PI, _I, _I, _I, _I, _I, _I, _P, _P]),
    "buctd_fuse_sum_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "buctd_fuse_sum_bwd_bnstat": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                       C.POINTER(_P), _P]),
    "buctd_resize_bilinear": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "buctd_maxpool3x3s2_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "buctd_maxpool3x3s2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "buctd_softmax_dropout_fwd": (_I, [_P, _L, _I, _F, _F, _U64, _P, _P, _P]),
    "buctd_softmax_dropout_bwd": (_I, [_P, _P, _L, _I, _F, _F, _U64, _P, _P]),
    "buctd_dropout": (_I, [_P, _P, _L, _F, _U64, _P]),
    "buctd_attn_smallqk_supported": (_I, [_I, _I, _I]),
    "buctd_attn_smallqk_fwd": (_I, [_I, _I, _I, _I, _P, _P, _P, _F, _F, _U64, _I, _P, _P, _P, _P]),
    "buctd_attn_smallqk_bwd": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _U64, _I, _P, _P, _P, _P, _P]),
    "buctd_layernorm_fwd": (_I, [_P, _P, _P, _L, _I, _F, _P, _P, _P, _P]),
    "buctd_add_layernorm_fwd": (_I, [_P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P, _P]),
    "buctd_layernorm_bwd_workspace": (_SZ, [_L, _I]),
    "buctd_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _I, _P, _SZ, _P]),
    "buctd_joints_mse": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _F, _P, _SZ, _P]),
    "buctd_argmax_decode": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "buctd_warp_affine_norm": (_I, [_P, _I, _I, _I, _P, _P, _P, _L, _P, _P]),
    "buctd_cond_render_into": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P, _SZ, _P]),
    "buctd_synthesize_pose": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, C.c_ulonglong, _P, _P]),
    "buctd_mha_fwd_supported": (_I, [_I, _I]),
    "buctd_mha_fwd": (_I, [_I, _I, _I, _P, _P, _P, _I, _I, _F, _P, _P, _P]),
    "buctd_mha_fwd_bf16x6": (_I, [_I, _I, _I, _P, _P, _P, _I, _I, _F, _P, _P, _P]),
    "buctd_gconv_x6_supported": (_I, [_I] * 7),
    "buctd_gconv_x6_prep_bytes": (_SZ, [_I, _I, _I, _I]),
    "buctd_gconv_x6_prep": (_I, [_I, _I, _I, _P, _I, _P, _P]),
    "buctd_gconv_x6_prep_item_bytes": (_SZ, []),
    "buctd_gconv_x6_prep_item": (_I, [_I, _I, _I, _P, _I, _P, _P]),
    "buctd_gconv_x6_prep_batched": (_I, [_P, _I, _P]),
    "buctd_gconv_x6_stats_groups": (_I, [_I] * 6 + [_PI, _PI]),
    "buctd_gconv_x6_fwd": (_I, [_I] * 6 + [_P] * 6 + [_I, _P, _P, _P, _P]),
    "buctd_gconv_x6_dgrad": (_I, [_I] * 6 + [_P] * 4 + [_P]),
    "buctd_mha_fwd_bf16x6_workspace": (_SZ, [_I, _I, _I]),
    "buctd_mha_fwd_bf16x6_ws": (_I, [_I, _I, _I, _P, _P, _P, _I, _I, _F, _P, _P, _P, _SZ, _P]),
    "buctd_mha_train_supported": (_I, [_I, _I]),
    "buctd_mha_fwd_train": (_I, [_I, _I, _I, _P, _P, _P, _I, _I, _F, _F, _U64, _P, _P, _P]),
    "buctd_mha_bwd_workspace": (_SZ, [_I, _I]),
    "buctd_mha_bwd": (_I, [_I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _F, _F, _U64, _P, _P, _I, _P, _I, _P, _SZ, _P]),
    "buctd_nms_workspace": (_SZ, [_I]),
    "buctd_nms": (_I, [_P, _P, _P, _I, _I, _F, _P, _SZ, _P]),
    "buctd_cpu_nms": (_I, [_P, _I, _P, _F, _P, _P]),
    "buctd_argmax_decode_refined": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "buctd_gaussian_target": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _F, _P, _P, _P]),
    "buctd_cond_render_workspace": (_SZ, [_I, _I, _I, _I]),
    "buctd_cond_render": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    "buctd_flipback_avg": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "buctd_sgd_step": (_I, [_P, _P, _P, _L, _F, _F, _F, _I, _I, _F, _P]),
    "buctd_adam_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P]),
}

_lib = None


class BuctdHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise BuctdHipError(
                f"{LIB_PATH} is missing - build it with `make` (or __graft_entry__.build()); "
                "buctd_amd has no CPU / eager fallback")
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().buctd_last_error()
        raise BuctdHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def current_stream_handle():
    """Raw hipStream_t of torch's current stream as an int.  torch.cuda.current_stream() builds a Stream object through
    several Python layers (~8 us); with ~4000 kernel launches per train step that alone was a third of the host time."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def stream_ptr():
    """hipStream_t of the current stream as a plain int (every bound function declares c_void_p argtypes: ctypes converts
    an int itself; building a c_void_p object per argument cost ~2 ms of host time per train step)."""
    return current_stream_handle()


def ptr(t):
    """Device pointer (plain int) of a contiguous fp32 (or int32) ROCm tensor, None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise BuctdHipError("buctd_amd ops need tensors on the ROCm device (no CPU path)")
    return t.data_ptr()
