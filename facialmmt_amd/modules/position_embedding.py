"""Sinusoidal positional embedding of the cross-modal encoder; host-side mirror of the reference's
modules/position_embedding.py (same class / function names, `_float_tensor` buffer in the state_dict).

Quirk reproduced on purpose: the encoder feeds *feature channel 0* to this module as
if it were token ids, so a time step whose channel-0 value is exactly 0.0 is "padding" and receives the
all-zero table row; every other step t gets row t+1.  On the hot path the gather is fused with the
sqrt(E) scaling in one kernel (ops.posemb_scale); this module only owns the table."""
from __future__ import annotations

import math

import torch
import torch.nn as nn


def make_positions(tensor, padding_idx, left_pad):
    """positions start at padding_idx+1; padding symbols keep padding_idx (ref :8-27)."""
    seq_len = tensor.size(1)
    pos = torch.arange(padding_idx + 1, padding_idx + 1 + seq_len, device=tensor.device).expand_as(tensor)
    mask = tensor.ne(padding_idx)
    if left_pad:
        pos = pos - seq_len + mask.long().sum(dim=1, keepdim=True)
    return torch.where(mask, pos, torch.full_like(pos, padding_idx)).long()


class SinusoidalPositionalEmbedding(nn.Module):
    def __init__(self, embedding_dim, padding_idx=0, left_pad=0, init_size=128):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.padding_idx = padding_idx
        self.left_pad = left_pad
        self.weights = dict()                       # device -> fp32 table, grown on demand
        self.register_buffer('_float_tensor', torch.FloatTensor(1))

    @staticmethod
    def get_embedding(num_embeddings, embedding_dim, padding_idx=None):
        """row p = [sin(p f_k) | cos(p f_k)], f_k = exp(-k ln(1e4)/(E/2-1)); halves concatenated (ref :44-61)."""
        half = embedding_dim // 2
        freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
        ang = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
        emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(num_embeddings, -1)
        if embedding_dim % 2 == 1:
            emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
        if padding_idx is not None:
            emb[padding_idx, :] = 0
        return emb

    def table(self, seq_len: int, device) -> torch.Tensor:
        """fp32 table with at least padding_idx + 1 + seq_len rows on `device`."""
        need = self.padding_idx + 1 + seq_len
        key = str(device)
        t = self.weights.get(key)
        if t is None or t.size(0) < need:
            t = self.get_embedding(max(need, 128), self.embedding_dim, self.padding_idx).to(device).contiguous()
            self.weights[key] = t
        return t

    def forward(self, input):
        """input (bsz, seqlen) of "token ids" -> (bsz, seqlen, E), detached (ref :63-76)."""
        bsz, seq_len = input.size()
        t = self.table(seq_len, input.device).type_as(self._float_tensor)
        positions = make_positions(input, self.padding_idx, self.left_pad)
        return t.index_select(0, positions.reshape(-1)).reshape(bsz, seq_len, -1).detach()

    def max_positions(self):
        return int(1e5)
