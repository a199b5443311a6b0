"""Layer library of the MI355X engine.

The classes subclass their ``torch.nn`` namesakes purely as *parameter containers* so that
constructor signatures, default initialisation, ``state_dict()`` keys / shapes and
``load_state_dict`` behave exactly like the reference's ``nn.Module`` tree
(reference lib/models/pose_hrnet.py builds everything from nn.Conv2d / nn.BatchNorm2d /
nn.Linear / nn.Sequential).  Their ``forward`` never calls ATen math: it routes NHWC tensors
through buctd_amd.ops (HIP kernels).  Conv weights are kept channels_last in memory
(= [Co][R][S][Ci]) - logical shape and checkpoint contents are unchanged.
"""
import torch
import torch.nn as tnn

from . import ops

Module = tnn.Module
ModuleList = tnn.ModuleList
Parameter = tnn.Parameter


def _as_channels_last_(p):
    if p.dim() == 4 and not p.is_contiguous(memory_format=torch.channels_last):
        old = p.data
        p.data = old.contiguous(memory_format=torch.channels_last)
        if old.is_cuda:
            # the first forward may run on a branch stream (ops.fork_join): keep the old storage away from the
            # allocator until that stream's copy has read it
            old.record_stream(torch.cuda.current_stream(old.device))


def prepare_module(module):
    """Put every 4-d conv weight of `module` into channels_last memory (idempotent).
    Called lazily by the conv layers and by engine.FlatParams."""
    for m in module.modules():
        if isinstance(m, (Conv2d, ConvTranspose2d)):
            _as_channels_last_(m.weight)
    return module


class Conv2d(tnn.Conv2d):
    """nn.Conv2d container; forward = NHWC implicit-GEMM conv (ops.Conv)."""

    def _geom(self):
        k, s, p = self.kernel_size, self.stride, self.padding
        if isinstance(p, str):
            if p != "same":
                raise ValueError("only integer or 'same' padding is supported")
            p = ((k[0] - 1) // 2, (k[1] - 1) // 2)
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or self.groups != 1 or self.dilation != (1, 1):
            raise ValueError("buctd_amd.nn.Conv2d supports square, symmetric, dense convolutions only")
        return s[0], p[0]

    def forward(self, x, relu=False):
        _as_channels_last_(self.weight)
        stride, pad = self._geom()
        return ops.Conv.apply(x, self.weight, self.bias, stride, pad, relu)


class ConvTranspose2d(tnn.ConvTranspose2d):
    """nn.ConvTranspose2d container (weight [Cin][Cout][k][k]); only used fused with BN."""

    def _geom(self):
        k, s, p, op = self.kernel_size, self.stride, self.padding, self.output_padding
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or self.groups != 1:
            raise ValueError("square, symmetric transposed convolutions only")
        return k[0], s[0], p[0], op[0]

    def out_shape(self, x_shape):
        k, s, p, op = self._geom()
        N, H, W, _ = x_shape
        return (N, (H - 1) * s - 2 * p + k + op, (W - 1) * s - 2 * p + k + op, self.out_channels)


class BatchNorm2d(tnn.BatchNorm2d):
    """nn.BatchNorm2d container. Stand-alone forward (no producing conv) is not needed on the path;
    BN always runs fused behind a conv via conv_bn_act().  num_batches_tracked (only consumed when
    momentum=None, which the path never uses) is counted on the host and written back when a
    state_dict is taken, instead of launching one int64 add kernel per BN per step."""

    _pending_batches = 0

    def count_batch(self):
        self._pending_batches += 1

    def flush_batches(self):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked += self._pending_batches
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_batches()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x):
        raise RuntimeError("buctd_amd.nn.BatchNorm2d runs fused with its convolution (conv_bn_act)")


class Linear(tnn.Linear):
    """nn.Linear on token tensors [B, T, Cin] -> [B, T, Cout], executed as a 1x1 conv: the
    [out, in] weight is bit-identical to an [out][1][1][in] conv filter."""

    def forward(self, x, relu=False):
        B, T, Cin = x.shape
        # any 2-D arrangement of the B * T rows is the same 1x1 convolution; as ONE image of B rows the zero-padded position
        # space of the gathered kernels carries one pad row per B (as B images of one row: one per row - half the tiles)
        y = ops.Conv.apply(x.view(1, B, T, Cin), self.weight, self.bias, 1, 0, relu)
        return y.view(B, T, self.out_features)


class ReLU(tnn.ReLU):
    def forward(self, x):
        raise RuntimeError("ReLU runs fused into the producing kernel")


class Upsample(tnn.Upsample):
    def forward(self, x):
        raise RuntimeError("nearest up-sampling runs fused into the fuse-sum kernel")


class Sequential(tnn.Sequential):
    pass


def conv_bn_act(x, conv, bn, relu=False, residual=None):
    """Fused conv -> BN -> (+residual) -> (ReLU) on NHWC tensors (train or eval)."""
    _as_channels_last_(conv.weight)
    if isinstance(conv, ConvTranspose2d):
        k, s, p, op = conv._geom()
        # ConvTranspose2d(Cin->Cout) == data gradient of a Conv2d(Cout->Cin) with the same weight
        return ops.ConvBnAct.apply(x, conv.weight, conv.bias, bn, residual, relu, s, p, bn.training,
                                   conv.out_shape(tuple(x.shape)))
    stride, pad = conv._geom()
    return ops.ConvBnAct.apply(x, conv.weight, conv.bias, bn, residual, relu, stride, pad, bn.training, None)


class ConvBN(Sequential):
    """nn.Sequential(Conv2d, BatchNorm2d[, ReLU | Upsample]) with the reference's child indices
    ('0','1','2'), executed as one fused op. `act` says whether child 2 is a ReLU."""

    def __init__(self, conv, bn, tail=None):
        mods = [conv, bn] + ([tail] if tail is not None else [])
        super().__init__(*mods)
        self._relu = isinstance(tail, tnn.ReLU)

    def forward(self, x, residual=None):
        return conv_bn_act(x, self[0], self[1], relu=self._relu, residual=residual)


class Chain(Sequential):
    """nn.Sequential of modules that each take/return one NHWC tensor."""

    def forward(self, x):
        for m in self:
            x = m(x)
        return x
