"""Per-modality self-attention encoder (BERT-style, post-LN, TF LayerNorm) and additive-attention
pooling; host-side mirror of the reference's modules/Transformer.py with identical class names and
state_dict keys.  SURVEY.md 8f rank 1: this stack is *next* in line for native kernels; in round 1 it
runs as stock PyTorch-ROCm ops on the GPU (it is on the logits path but outside the section-8 hot path),
written device-agnostically (the reference hard-codes .cuda(), ref :213)."""
from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class AdditiveAttention(nn.Module):
    """score_t = value(tanh(P h_t + Q query_vector)); masked softmax over t; weighted sum (ref :8-45)."""

    def __init__(self, inputs_dim, hidden_dim):
        super().__init__()
        self.query_vector = nn.Parameter(torch.randn(inputs_dim))
        self.value = nn.Linear(hidden_dim, 1)
        self.P = nn.Linear(inputs_dim, hidden_dim)
        self.Q = nn.Linear(inputs_dim, hidden_dim)
        self.tanh = nn.Tanh()

    def forward(self, inputs, mask=None):
        B, L, _ = inputs.size()
        if L == 1:
            return inputs.squeeze(), 1
        scores = self.value(self.tanh(self.P(inputs) + self.Q(self.query_vector))).squeeze(-1)     # (B, L)
        if mask is not None:
            scores = scores.masked_fill(mask == 0., float('-inf'))
        alpha = F.softmax(scores, dim=-1).view(B, 1, L)
        return torch.bmm(alpha, inputs).squeeze(dim=1), alpha


class LayerNorm(nn.Module):
    """TF-style layer norm, epsilon inside the square root (ref :48-61)."""

    def __init__(self, hidden_size, eps=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        return F.layer_norm(x, x.shape[-1:], self.weight, self.bias, self.variance_epsilon)


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

    def transpose_for_scores(self, x):
        return x.view(*x.shape[:-1], self.num_attention_heads, self.attention_head_size).permute(0, 2, 1, 3)

    def forward(self, hidden_states, attention_mask):
        q = self.transpose_for_scores(self.query(hidden_states))
        k = self.transpose_for_scores(self.key(hidden_states))
        v = self.transpose_for_scores(self.value(hidden_states))
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.attention_head_size) + attention_mask
        probs = self.dropout(torch.softmax(scores, dim=-1))
        ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
        return ctx.view(*ctx.shape[:-2], self.all_head_size)


def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


class TransformerIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        self.intermediate_act_fn = gelu

    def forward(self, hidden_states):
        return self.intermediate_act_fn(self.dense(hidden_states))


class Residual_Norm(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


class Output_Residual_Norm(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.selfatt = SelfAttention(config)
        self.dense_norm = Residual_Norm(config)

    def forward(self, input_tensor, attention_mask):
        return self.dense_norm(self.selfatt(input_tensor, attention_mask), input_tensor)


class TransformerEnoderLayer(nn.Module):      # (sic) the reference's spelling is part of the API
    def __init__(self, config):
        super().__init__()
        self.transformer_self_attention = MultiHeadSelfAttention(config)
        self.intermediate = TransformerIntermediate(config)
        self.output = Output_Residual_Norm(config)

    def forward(self, inputs, attention_mask):
        a = self.transformer_self_attention(inputs, attention_mask)
        return self.output(self.intermediate(a), a)


class MELDTransEncoder(nn.Module):
    def __init__(self, config, layer_num, get_max_lens, hidden_size):
        super().__init__()
        self.position_embeddings = nn.Embedding(get_max_lens, hidden_size)
        layer = TransformerEnoderLayer(config)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(layer_num)])

    def forward(self, feature_input, attention_mask, output_all_encoded_layers=False):
        L = feature_input.shape[1]
        pos = self.position_embeddings(torch.arange(L, dtype=torch.long, device=feature_input.device))
        x = feature_input + pos.unsqueeze(0)
        for layer_module in self.layer:
            x = layer_module(x, attention_mask)
        return x
