"""Per-modality self-attention encoder (BERT-style, post-LN, TF LayerNorm) and additive-attention
pooling on the HIP path; host-side mirror of the reference's modules/Transformer.py with identical class
names, constructor signatures and state_dict keys.

Every Linear / LayerNorm / attention core below is a libfmmt_hip kernel (ops.linear, ops.mlp,
ops.layer_norm, ops.mha_core with the extended attention mask as its `key_bias`); like the rest of the hot
path there is no PyTorch / CPU formulation behind it -- CPU tensors raise FmmtError.  What stays as torch
elementwise ops on (B, L)-sized tensors: the position-embedding add, hidden-state dropout (p = 0.1 between a
dense layer and its residual add, ref :121,134) and the tanh / masked-softmax / weighted-sum tail of
AdditiveAttention.

Layout: the reference is batch-major (B, L, H); the attention kernel is time-major (row t of (L, B, H) at
(t*B + b)*ld), so MELDTransEncoder transposes once on entry and once on exit and runs its layers time-major
(all other ops are per-token and layout-agnostic).  The sub-modules keep the reference's batch-major
`forward` signatures for API compatibility and expose the time-major body as `forward_tm`.

Compute dtype: `compute_dtype` (set by the owning model from `config.compute_dtype`; None = the input's
dtype) selects bf16 (throughput) or fp32 (parity) for the kernels; the result is returned in the input's dtype.
The reference hard-codes .cuda() for the position ids (ref :213); here they follow the input's device."""
from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _key_bias(attention_mask, B, L):
    """(B,1,1,L) extended attention mask ((1-m) * -10000, src/models.py:157,164) -> (B, L) fp32 logit bias"""
    if attention_mask is None:
        return None
    kb = attention_mask.reshape(-1, attention_mask.shape[-1]).to(torch.float32)
    if kb.shape[0] == 1 and B > 1:
        kb = kb.expand(B, L)
    assert kb.shape == (B, L), f"attention_mask must broadcast over heads and queries: expected (B,1,1,L) = ({B},1,1,{L})"
    return kb.contiguous()


class AdditiveAttention(nn.Module):
    """score_t = value(tanh(P h_t + Q query_vector)); masked softmax over t; weighted sum (ref :8-45)."""

    def __init__(self, inputs_dim, hidden_dim):
        super().__init__()
        self.query_vector = nn.Parameter(torch.randn(inputs_dim))
        self.value = nn.Linear(hidden_dim, 1)
        self.P = nn.Linear(inputs_dim, hidden_dim)
        self.Q = nn.Linear(inputs_dim, hidden_dim)
        self.tanh = nn.Tanh()
        self.compute_dtype = None

    def forward(self, inputs, mask=None):
        B, L, _ = inputs.size()
        if L == 1:
            return inputs.squeeze(), 1
        cd = self.compute_dtype or inputs.dtype
        # P h_t: the one real GEMM of the pooling (B*L x H x H) on the HIP path; Q query_vector is a single row
        ph = ops.linear(inputs.to(cd), self.P.weight, self.P.bias).to(inputs.dtype)
        qq = ops.linear(self.query_vector.to(cd).unsqueeze(0), self.Q.weight, self.Q.bias).to(inputs.dtype)
        scores = F.linear(torch.tanh(ph + qq), self.value.weight.to(inputs.dtype), self.value.bias.to(inputs.dtype)).squeeze(-1)
        if mask is not None:
            scores = scores.masked_fill(mask == 0., float('-inf'))
        alpha = F.softmax(scores.float(), dim=-1).to(inputs.dtype).view(B, 1, L)
        return torch.bmm(alpha, inputs).squeeze(dim=1), alpha


class LayerNorm(nn.Module):
    """TF-style layer norm, epsilon inside the square root (ref :48-61) -- the same arithmetic as
    fmmt_layernorm_fwd's rsqrt(var + eps)."""

    def __init__(self, hidden_size, eps=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.variance_epsilon)


class SelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def forward_tm(self, x, key_bias):
        """x (L, B, H) time-major; key_bias (B, L) or None.  softmax(q k^T / sqrt(d) + mask) v with dropout on the
        probabilities (ref :86-103); the (B, nH, L, L) tensors are never materialised."""
        p = self.dropout.p if self.training else 0.0
        seed = ops.draw_seed(x.device) if p > 0 else 0
        scale = 1.0 / math.sqrt(self.attention_head_size)
        if PACKED_QKV and self.query.bias is not None and self.key.bias is not None and self.value.bias is not None:
            # the three projections as one launch into a packed (L*B, 3H) buffer the attention core works on in place (ops.SelfAttnQkvFn)
            return ops.self_attention_qkv(x, self.query.weight, self.query.bias, self.key.weight, self.key.bias, self.value.weight, self.value.bias,
                                          self.num_attention_heads, scale, p, seed, key_bias)
        q = ops.linear(x, self.query.weight, self.query.bias)
        k = ops.linear(x, self.key.weight, self.key.bias)
        v = ops.linear(x, self.value.weight, self.value.bias)
        return ops.mha_core(q, k, v, self.num_attention_heads, scale, p, seed, key_bias)

    def forward(self, hidden_states, attention_mask):
        B, L, _ = hidden_states.shape
        out = self.forward_tm(hidden_states.transpose(0, 1).contiguous(), _key_bias(attention_mask, B, L))
        return out.transpose(0, 1).contiguous()


def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


class TransformerIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        self.intermediate_act_fn = gelu

    def forward(self, hidden_states):
        # standalone use only: inside the encoder layer the erf GELU is the epilogue of this GEMM (ops.mlp)
        return self.intermediate_act_fn(ops.linear(hidden_states, self.dense.weight, self.dense.bias))


FUSED_TAILS = True           # module constant (tests patch it): False = dropout, add and LayerNorm as separate launches
PACKED_QKV = True            # False = query / key / value as three Linear launches (and 3 x 3 in the backward)


def _tail(m, h, input_tensor):
    """LayerNorm(dropout(h) + input) of a training-mode sublayer (ref :121-123,134-136; `m` owns .dropout and .LayerNorm).  bf16 on the GPU: ONE
    launch (ops.dropadd_layer_norm: counter-based dropout replayed in the backward, bf16 rounding op by op as the three torch / HIP launches it
    replaces); otherwise those three."""
    ln = m.LayerNorm
    if FUSED_TAILS and h.is_cuda and h.dtype == torch.bfloat16 and input_tensor.dtype == torch.bfloat16 and h.shape[-1] % 8 == 0 and h.shape[-1] <= 2048:
        seed = ops.draw_seed(h.device)                               # drawn on the device: graph-replay safe
        return ops.dropadd_layer_norm(h, input_tensor, ln.weight, ln.bias, ln.variance_epsilon, m.dropout.p, seed)
    return ln(m.dropout(h) + input_tensor)


class Residual_Norm(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        if self.training and self.dropout.p > 0:
            return _tail(self, ops.linear(hidden_states, self.dense.weight, self.dense.bias), input_tensor)
        # no dropout between dense and residual: the residual add is the GEMM epilogue
        return self.LayerNorm(ops.linear(hidden_states, self.dense.weight, self.dense.bias, res=input_tensor))


class Output_Residual_Norm(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        if self.training and self.dropout.p > 0:
            return _tail(self, ops.linear(hidden_states, self.dense.weight, self.dense.bias), input_tensor)
        return self.LayerNorm(ops.linear(hidden_states, self.dense.weight, self.dense.bias, res=input_tensor))


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.selfatt = SelfAttention(config)
        self.dense_norm = Residual_Norm(config)

    def forward_tm(self, x, key_bias):
        return self.dense_norm(self.selfatt.forward_tm(x, key_bias), x)

    def forward(self, input_tensor, attention_mask):
        return self.dense_norm(self.selfatt(input_tensor, attention_mask), input_tensor)


class TransformerEnoderLayer(nn.Module):      # (sic) the reference's spelling is part of the API
    def __init__(self, config):
        super().__init__()
        self.transformer_self_attention = MultiHeadSelfAttention(config)
        self.intermediate = TransformerIntermediate(config)
        self.output = Output_Residual_Norm(config)

    def _ffn(self, a):
        """LN(dropout(dense2(gelu(dense1(a)))) + a)  (ref :109-137): fc1 + erf GELU + fc2 as ops.mlp (GELU in fc1's
        epilogue, GELU' in the epilogue of fc2's input-gradient GEMM)."""
        out, inter = self.output, self.intermediate
        if self.training and out.dropout.p > 0:
            return _tail(out, ops.mlp(a, inter.dense.weight, inter.dense.bias, out.dense.weight, out.dense.bias), a)
        return out.LayerNorm(ops.mlp(a, inter.dense.weight, inter.dense.bias, out.dense.weight, out.dense.bias, res=a))

    def forward_tm(self, x, key_bias):
        return self._ffn(self.transformer_self_attention.forward_tm(x, key_bias))

    def forward(self, inputs, attention_mask):
        return self._ffn(self.transformer_self_attention(inputs, attention_mask))


class MELDTransEncoder(nn.Module):
    def __init__(self, config, layer_num, get_max_lens, hidden_size):
        super().__init__()
        self.position_embeddings = nn.Embedding(get_max_lens, hidden_size)
        layer = TransformerEnoderLayer(config)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(layer_num)])
        self.compute_dtype = None

    def forward(self, feature_input, attention_mask, output_all_encoded_layers=False):
        B, L, _ = feature_input.shape
        pos = self.position_embeddings(torch.arange(L, dtype=torch.long, device=feature_input.device))
        cd = self.compute_dtype or feature_input.dtype
        x = (feature_input + pos.unsqueeze(0)).transpose(0, 1).contiguous().to(cd)          # (L, B, H) time-major
        kb = _key_bias(attention_mask, B, L)
        for layer_module in self.layer:
            x = layer_module.forward_tm(x, kb)
        return x.transpose(0, 1).to(feature_input.dtype)
