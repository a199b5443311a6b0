"""Attention cores of CoAM on the MI355X engine (token tensors are [B, T, C] == flattened NHWC).

Mirrors reference lib/models/self_attention.py: ScaledDotProductAttention (10-88: fc_q/k/v/o, softmax,
dropout 0.1) and SimplifiedScaledDotProductAttention (95-160: no q/k/v projections, fc_o only) with the
same parameter names and the same constructor-time init (Linear weights normal(std=0.001), bias 0).
"""
import torch

from .. import nn
from .. import ops


def _init_linear(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                torch.nn.init.constant_(m.bias, 0)


class ScaledDotProductAttention(nn.Module):
    def __init__(self, in_dim_q, in_dim_k, d_k, d_v, h, dropout=.1, rev=False):
        super().__init__()
        d_model = in_dim_q if rev else in_dim_k
        self.fc_q = nn.Linear(in_dim_q, h * d_k)
        self.fc_k = nn.Linear(in_dim_k, h * d_k)
        self.fc_v = nn.Linear(in_dim_k, h * d_v)
        self.fc_o = nn.Linear(h * d_v, d_model)
        self.dropout = torch.nn.Dropout(dropout)  # holds p; the mask is generated inside the softmax kernel
        self.d_model, self.d_k, self.d_v, self.h = d_model, d_k, d_v, h
        self.fused = True  # take the fused narrow-contraction kernels whenever the shape allows (tests toggle this)
        _init_linear(self)

    def forward(self, queries, keys, values, attention_mask=None, attention_weights=None):
        if attention_mask is not None or attention_weights is not None:
            raise NotImplementedError("attention_mask / attention_weights are never passed on the BUCTD path")
        k = self.fc_k(keys)
        v = self.fc_v(values)
        if self.fused and ops.attn_smallqk_ok(queries.shape[1], queries.shape[2], self.h * self.d_k, self.h) \
                and self.d_k == self.d_v and keys.shape[1] == queries.shape[1]:
            # narrow fc_q (condition channels): fold it into the keys and never materialise the T x T matrix
            out = ops.SmallQKAttention.apply(queries, self.fc_q.weight, self.fc_q.bias, k, v, float(self.dropout.p),
                                             self.training)
            return self.fc_o(out)
        q = self.fc_q(queries)
        out = ops.PositionAttention.apply(q, k, v, self.h, float(self.dropout.p), self.training)
        return self.fc_o(out)


class SimplifiedScaledDotProductAttention(nn.Module):
    """Channel attention: the reference feeds [B, C, T] tensors (d_model = T = H*W). Here the same
    quantities arrive token-major ([B, T, C]); ops.ChannelAttention contracts over T accordingly."""

    def __init__(self, d_model, h, dropout=.1):
        super().__init__()
        self.d_model = d_model
        self.d_k = self.d_v = d_model // h
        self.h = h
        self.fc_o = nn.Linear(h * self.d_v, d_model)
        self.dropout = torch.nn.Dropout(dropout)
        _init_linear(self)

    def forward(self, queries_tokens, keys_tokens, values_tokens=None, attention_mask=None, attention_weights=None):
        if attention_mask is not None or attention_weights is not None:
            raise NotImplementedError("attention_mask / attention_weights are never passed on the BUCTD path")
        if values_tokens is not None and values_tokens is not keys_tokens:
            raise NotImplementedError("keys and values are always the same tensor on the BUCTD path")
        return ops.ChannelAttention.apply(queries_tokens, keys_tokens, self.fc_o.weight, self.fc_o.bias, self.h,
                                          float(self.dropout.p), self.training)
