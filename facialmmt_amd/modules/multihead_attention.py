"""Multi-head attention of the cross-modal encoder on the HIP path; host-side mirror of the reference's
modules/multihead_attention.py (fairseq-style, time-major, packed in_proj_weight (3E,E)).

forward(query, key, value, attn_mask=None, need_weights=False) -> (attn, attn_weights).  The head-averaged attention weights
the reference returns (ref :133-134) are discarded by every caller (CrossmodalTransformer.py:147,151) and the fused kernel never
materialises the (B*nH, Lq, Lk) probability tensor, so by default None is returned in their place; `need_weights=True`
recomputes them (fmmt_mha_avg_weights: post-dropout probabilities averaged over the heads, (B, Lq, Lk), no gradient).
Unsupported options raise: attn_mask, add_bias_kv, add_zero_attn."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import Parameter

from .. import ops


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, attn_dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.attn_dropout = attn_dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.scaling = self.head_dim ** -0.5
        self.in_proj_weight = Parameter(torch.Tensor(3 * embed_dim, embed_dim))
        self.register_parameter('in_proj_bias', None)
        if bias:
            self.in_proj_bias = Parameter(torch.Tensor(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        if add_bias_kv or add_zero_attn:
            raise NotImplementedError("facialmmt_amd HIP path: add_bias_kv / add_zero_attn (unused by the reference model)")
        self.bias_k = self.bias_v = None
        self.add_zero_attn = add_zero_attn
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.in_proj_bias is not None:
            nn.init.constant_(self.in_proj_bias, 0.)
            nn.init.constant_(self.out_proj.bias, 0.)

    def _proj(self, x, start, end):
        b = self.in_proj_bias[start:end] if self.in_proj_bias is not None else None
        return ops.linear(x, self.in_proj_weight[start:end], b)

    def in_proj_q(self, query):
        return self._proj(query, 0, self.embed_dim)

    def in_proj_k(self, key):
        return self._proj(key, self.embed_dim, 2 * self.embed_dim)

    def in_proj_v(self, value):
        return self._proj(value, 2 * self.embed_dim, 3 * self.embed_dim)

    def in_proj_kv(self, key):
        return self._proj(key, self.embed_dim, 3 * self.embed_dim)

    def attend(self, query, key, value, res=None, need_weights=False, segments=None):
        """out_proj(softmax(q k^T) v) [+ res]; `res` fuses the caller's residual add into the GEMM epilogue.
        need_weights: return (out, head-averaged attention weights (B, Lq, Lk)) instead of out.
        segments: ((q0, Lq, k0, Lk), ...) -- query / key hold several independent sequences stacked along time, rows q0..q0+Lq of
        the queries attend to rows k0..k0+Lk of the keys (ops.MhaSegFn): every projection runs once over all of them."""
        tgt_len, bsz, embed_dim = query.size()
        assert embed_dim == self.embed_dim
        assert key.size() == value.size()
        E = self.embed_dim
        p = self.attn_dropout if self.training else 0.0
        if segments is not None:
            assert key is value and not need_weights, "stacked sequences: packed [k | v] projection, no attention weights"
            seeds = ops.draw_seed(query.device, len(segments)) if p > 0 else 0
            q, kproj = ops.in_proj_q_kv(query, key, self.in_proj_weight, self.in_proj_bias)
            ctx = ops.mha_core_segments(q, kproj, segments, self.num_heads, self.scaling, p, seeds)
            return ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, res)
        # fresh seed per call, drawn on the device: no host sync, and a captured hipGraph re-draws it on replay
        seed = ops.draw_seed(query.device) if p > 0 else 0
        if key is value or (key.data_ptr() == value.data_ptr() and key.shape == value.shape):
            # two GEMMs (q: N = E; [k | v]: N = 2E) whose weight gradients land in one packed (3E, E) gradient: ops.InProjFn
            q, kproj = ops.in_proj_q_kv(query, key, self.in_proj_weight, self.in_proj_bias)
            vproj = None
        else:
            q, kproj, vproj = self.in_proj_q(query), self.in_proj_k(key), self.in_proj_v(value)
        if not need_weights:
            ctx = ops.mha_core(q, kproj, vproj, self.num_heads, self.scaling, p, seed)
            return ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, res)
        ctx, lse = ops.mha_core(q, kproj, vproj, self.num_heads, self.scaling, p, seed, return_lse=True)
        weights = ops.mha_avg_weights(q, kproj, lse, self.num_heads, self.scaling, p, seed)
        return ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, res), weights

    def forward(self, query, key, value, attn_mask=None, need_weights=False):
        """Time x Batch x Channel in, (attn (Lq,B,E), attn_weights) out (ref :51-135); attn_weights is None unless need_weights."""
        if attn_mask is not None:
            raise NotImplementedError("facialmmt_amd HIP path: attn_mask (the reference model always passes None)")
        if need_weights:
            return self.attend(query, key, value, need_weights=True)
        return self.attend(query, key, value), None
