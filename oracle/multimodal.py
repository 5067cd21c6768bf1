"""ORACLE (test infrastructure, never shipped or timed as the product).

CPU restatement of the callers around the hot path: modules/Transformer.py (per-modality self-attention
encoder + additive-attention pooling) and src/models.py's MultiModalTransformerForClassification.forward
/ meld_utt_transformer.forward, as pure functions over a state_dict.  The text encoder is whatever
callable the test passes in (a deterministic stand-in).  Pinned by tests/golden/multimodal.npz."""
from __future__ import annotations

import math

import torch

from .crossmodal import crossmodal_encoder
from .swin import gelu_erf


def tf_layer_norm(x, w, b, eps):
    """modules/Transformer.py:48-61 (epsilon inside the square root)."""
    u = x.mean(-1, keepdim=True)
    s = ((x - u) ** 2).mean(-1, keepdim=True)
    return w * ((x - u) / torch.sqrt(s + eps)) + b


def lin(sd, pre, x):
    return x @ sd[pre + "weight"].t() + sd[pre + "bias"]


def meld_encoder(sd, pre, x, ext_mask, n_layers, n_heads=12, eps=1e-12):
    """MELDTransEncoder.forward, eval mode (modules/Transformer.py:196-226): learned position embedding for
    positions 0..L-1 added to the input, then post-LN layers: self-attention (scores/sqrt(hd) + additive
    mask, :92-101) -> dense + residual + LN (:136-140) -> dense + gelu (:128-131) -> dense + residual + LN (:150-154)."""
    B, L, Hd = x.shape
    x = x + sd[pre + "position_embeddings.weight"][:L][None]
    hd = Hd // n_heads
    for i in range(n_layers):
        p = f"{pre}layer.{i}."
        a = p + "transformer_self_attention."
        def heads(t):
            return t.reshape(B, L, n_heads, hd).permute(0, 2, 1, 3)
        q, k, v = heads(lin(sd, a + "selfatt.query.", x)), heads(lin(sd, a + "selfatt.key.", x)), heads(lin(sd, a + "selfatt.value.", x))
        s = q @ k.transpose(-1, -2) / math.sqrt(hd) + ext_mask
        ctx = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, L, Hd)
        att = tf_layer_norm(lin(sd, a + "dense_norm.dense.", ctx) + x, sd[a + "dense_norm.LayerNorm.weight"], sd[a + "dense_norm.LayerNorm.bias"], eps)
        inter = gelu_erf(lin(sd, p + "intermediate.dense.", att))
        x = tf_layer_norm(lin(sd, p + "output.dense.", inter) + att, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return x


def additive_attention(sd, pre, x, mask):
    """AdditiveAttention.forward (modules/Transformer.py:24-45): value(tanh(P x + Q query)) -> -inf where
    mask == 0 -> softmax over time -> weighted sum."""
    sc = lin(sd, pre + "value.", torch.tanh(lin(sd, pre + "P.", x) + lin(sd, pre + "Q.", sd[pre + "query_vector"]))).squeeze(-1)
    sc = sc.masked_fill(mask == 0., float("-inf"))
    return (torch.softmax(sc, -1)[:, None] @ x).squeeze(1)


def slice_target_utterance_loop(text_feats, sep_mask, utt_idx, max_len, roberta):
    """Literal restatement of the token-slicing double loop (src/models.py:112-150)."""
    B, T, Hd = text_feats.shape
    out = torch.zeros(B, max_len, Hd, dtype=text_feats.dtype)
    msk = torch.zeros(B, max_len)
    for i in range(B):
        u = int(utt_idx[i])
        seps = []
        for index, value in enumerate(sep_mask[i].tolist()):
            if value == 1:
                seps.append(index)
                if u == 0:
                    n = min(index - 1, max_len)
                    out[i, :n] = text_feats[i, 1:n + 1]
                    msk[i, :n] = 1
                    break
                elif u > 0 and u + 1 == len(seps):
                    cur, prev = seps[-1], seps[-2]
                    gap = 2 if roberta else 1
                    n = min(cur - prev - gap, max_len)
                    out[i, :n] = text_feats[i, prev + gap:prev + gap + n]
                    msk[i, :n] = 1
                    break
    return out, msk


def multimodal_logits(sd, plm, cfg, ids, attn_mask, sep_mask, audio, audio_mask, vision, vision_mask, utt_idx, roberta=True):
    """MultiModalTransformerForClassification.forward in eval mode (src/models.py:95-188)."""
    text = lin(sd, "text_linear.", plm(ids, attn_mask)[0])
    t_feat, t_mask = slice_target_utterance_loop(text, sep_mask, utt_idx, cfg.get_text_utt_max_lens, roberta)
    a_ext = (1.0 - audio_mask[:, None, None]) * -10000.0
    a = meld_encoder(sd, "audio_utt_transformer.", lin(sd, "audio_linear.", audio), a_ext, cfg.audio_utt_Transformernum, cfg.num_attention_heads, cfg.layer_norm_eps)
    v_ext = (1.0 - vision_mask[:, None, None]) * -10000.0
    v = meld_encoder(sd, "vision_utt_transformer.", lin(sd, "vision_linear.", vision), v_ext, cfg.vision_utt_Transformernum, cfg.num_attention_heads, cfg.layer_norm_eps)
    t_tm, a_tm, v_tm = t_feat.transpose(0, 1), a.transpose(0, 1), v.transpose(0, 1)
    ta = torch.cat((crossmodal_encoder(sd, t_tm, a_tm, a_tm, cfg.crossmodal_num_heads_TA, "CrossModalTrans_TA."),
                    crossmodal_encoder(sd, a_tm, t_tm, t_tm, cfg.crossmodal_num_heads_TA, "CrossModalTrans_TA.")), 0)
    v_x = crossmodal_encoder(sd, v_tm, ta, ta, cfg.crossmodal_num_heads_TA_V, "CrossModalTrans_TA_V.")
    ta_x = crossmodal_encoder(sd, ta, v_tm, v_tm, cfg.crossmodal_num_heads_TA_V, "CrossModalTrans_TA_V.")
    final = torch.cat((ta_x, v_x), 0).transpose(0, 1)
    mask = torch.cat((t_mask, audio_mask, vision_mask), 1)
    return lin(sd, "classifier.", additive_attention(sd, "attention.", final, mask))


def meld_utt_logits(sd, cfg, x, mask):
    """meld_utt_transformer.forward in eval mode (src/models.py:209-223)."""
    ext = (1.0 - mask[:, None, None]) * -10000.0
    h = meld_encoder(sd, "utt_transformer.", lin(sd, "modality_linear.", x), ext, cfg.vision_utt_Transformernum, cfg.num_attention_heads, cfg.layer_norm_eps)
    return lin(sd, "classifier.", additive_attention(sd, "attention.", h, mask))
