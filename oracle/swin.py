"""ORACLE (test infrastructure, never shipped or timed as the product).

CPU restatement of the reference's Swin-tiny facial encoder, written as pure functions over a
``state_dict`` (name -> tensor).  Works in fp32 or fp64 on any torch device; autograd through these
functions is the gradient oracle.  Every function cites the reference lines it restates
(paths relative to /root/reference).  Pinned by tests/golden/*.npz, which were produced by importing
the reference itself (oracle/gen_golden.py) -- see tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Deliberate differences in *formulation* (results identical): roll / window_partition /
window_reverse are never materialised; a single gather index per (H, W, shift) maps window slots to
image tokens, which is exactly what the HIP kernels fold into their address arithmetic.
"""
from __future__ import annotations

import math
from functools import lru_cache

import torch
import torch.nn.functional as F

WS = 7  # window size, modules/SwinTransformer/swin_conf.yaml:20

# Swin-tiny geometry fixed by swin_conf.yaml:4-22
DEPTHS = (2, 2, 6, 2)
HEADS = (3, 6, 12, 24)
EMBED = 96
GRID0 = 56


# ---------------------------------------------------------------------------------------------
# index tables
# ---------------------------------------------------------------------------------------------
@lru_cache(maxsize=None)
def window_token_index(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """(nW, ws*ws) int64: token id (row-major in the H x W grid) held by slot p of window w.

    Restates roll(-shift) + window_partition (Swin_Transformer.py:244,33-45): slot (i, j) of window
    (wy, wx) reads pixel ((wy*ws+i+shift) % H, (wx*ws+j+shift) % W).  window_reverse + roll(+shift)
    (:48-62,261) is the scatter through the same table."""
    wy, wx, i, j = torch.meshgrid(torch.arange(H // ws), torch.arange(W // ws), torch.arange(ws),
                                  torch.arange(ws), indexing="ij")
    h = (wy * ws + i + shift) % H
    w = (wx * ws + j + shift) % W
    return (h * W + w).reshape(-1, ws * ws)


@lru_cache(maxsize=None)
def shift_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """(nW, ws*ws, ws*ws) float of {0, -100}; restates Swin_Transformer.py:208-227.

    Region ids live on *shifted* coordinates: rows [0,H-ws) -> 0, [H-ws,H-shift) -> 1, [H-shift,H) -> 2
    (same for columns), id = 3*row_region + col_region; a pair of slots with different ids gets -100."""
    def region(n):
        r = torch.zeros(n, dtype=torch.long)
        r[n - ws:n - shift] = 1
        r[n - shift:] = 2
        return r
    rid = region(H)[:, None] * 3 + region(W)[None, :]                    # (H, W) on shifted coords
    wy, wx, i, j = torch.meshgrid(torch.arange(H // ws), torch.arange(W // ws), torch.arange(ws),
                                  torch.arange(ws), indexing="ij")
    ids = rid[wy * ws + i, wx * ws + j].reshape(-1, ws * ws)              # (nW, 49)
    diff = ids[:, None, :] != ids[:, :, None]
    return torch.where(diff, torch.tensor(-100.0), torch.tensor(0.0))


@lru_cache(maxsize=None)
def relative_position_index(ws: int) -> torch.Tensor:
    """(ws*ws, ws*ws) int64 into the (2ws-1)^2 bias table; restates Swin_Transformer.py:93-102."""
    p = torch.arange(ws * ws)
    pi, pj = p // ws, p % ws
    di = pi[:, None] - pi[None, :] + ws - 1
    dj = pj[:, None] - pj[None, :] + ws - 1
    return di * (2 * ws - 1) + dj


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
def gelu_erf(x):
    """nn.GELU() default = exact erf form (Swin_Transformer.py:15,20)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def patch_embed(sd, x, pre="patch_embed."):
    """Conv2d(3,96,k=4,s=4) -> (N, 3136, 96) -> LayerNorm(96).  Swin_Transformer.py:407,419-421.

    Written as a patch-matrix product: patch (py,px) row = [c][ky][kx] flattened (K=48)."""
    N, Cin, H, W = x.shape
    wgt = sd[pre + "proj.weight"]                                          # (96, 3, 4, 4)
    P = wgt.shape[-1]
    patches = x.reshape(N, Cin, H // P, P, W // P, P).permute(0, 2, 4, 1, 3, 5).reshape(N, (H // P) * (W // P), Cin * P * P)
    y = patches @ wgt.reshape(wgt.shape[0], -1).t() + sd[pre + "proj.bias"]
    return layer_norm(y, sd[pre + "norm.weight"], sd[pre + "norm.bias"])


def window_attention(sd, pre, xw, num_heads, mask=None):
    """W-MSA on windows xw (B_, 49, C).  Swin_Transformer.py:113-144.

    q is scaled by head_dim**-0.5 *before* QK^T (:123-124); bias = table[index] (:126-129);
    mask (nW,49,49) broadcast over batch and heads (:131-134); softmax over keys."""
    B_, T, C = xw.shape
    hd = C // num_heads
    qkv = xw @ sd[pre + "qkv.weight"].t() + sd[pre + "qkv.bias"]           # (B_, T, 3C): [q | k | v], head-major inside
    qkv = qkv.reshape(B_, T, 3, num_heads, hd)
    q = qkv[:, :, 0].transpose(1, 2) * (hd ** -0.5)                        # (B_, nH, T, hd)
    k = qkv[:, :, 1].transpose(1, 2)
    v = qkv[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2)                                            # (B_, nH, T, T)
    table = sd[pre + "relative_position_bias_table"]                       # (169, nH)
    idx = relative_position_index(WS).to(table.device)
    s = s + table[idx.reshape(-1)].reshape(T, T, num_heads).permute(2, 0, 1)
    if mask is not None:
        nW = mask.shape[0]
        s = (s.reshape(B_ // nW, nW, num_heads, T, T) + mask.to(s)[None, :, None]).reshape(B_, num_heads, T, T)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B_, T, C)
    return o @ sd[pre + "proj.weight"].t() + sd[pre + "proj.bias"]


def swin_block(sd, pre, x, H, W, num_heads, shift, drop_path_scale=None):
    """One SwinTransformerBlock (Swin_Transformer.py:233-270).

    drop_path_scale: None (eval / rate 0) or a pair of per-sample (N,) multipliers
    (already divided by keep-prob, timm DropPath semantics) for the two residual branches."""
    N, L, C = x.shape
    if min(H, W) <= WS:                                                    # :192-195
        shift = 0
    idx = window_token_index(H, W, WS, shift).to(x.device)                 # (nW, 49)
    nW = idx.shape[0]
    xn = layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    xw = xn[:, idx.reshape(-1)].reshape(N * nW, WS * WS, C)                # gather == roll+partition
    mask = shift_mask(H, W, WS, shift).to(x.device) if shift > 0 else None
    aw = window_attention(sd, pre + "attn.", xw, num_heads, mask)
    a = torch.empty_like(x, dtype=aw.dtype)                                # (aw's dtype: the same function also runs under bf16 autocast on the GPU)
    a[:, idx.reshape(-1)] = aw.reshape(N, nW * WS * WS, C)                 # scatter == reverse+roll back
    if drop_path_scale is not None:
        a = a * drop_path_scale[0].to(a)[:, None, None]
    x = x + a
    h = layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    h = gelu_erf(h @ sd[pre + "mlp.fc1.weight"].t() + sd[pre + "mlp.fc1.bias"])
    h = h @ sd[pre + "mlp.fc2.weight"].t() + sd[pre + "mlp.fc2.bias"]
    if drop_path_scale is not None:
        h = h * drop_path_scale[1].to(h)[:, None, None]
    return x + h


def patch_merging(sd, pre, x, H, W):
    """2x2 neighbour concat in order (even,even),(odd,even),(even,odd),(odd,odd) -> LN(4C) ->
    Linear(4C,2C,no bias).  Swin_Transformer.py:316-326."""
    N, L, C = x.shape
    g = x.reshape(N, H // 2, 2, W // 2, 2, C)                              # [n, h2, dh, w2, dw, c]
    cat = torch.cat([g[:, :, 0, :, 0], g[:, :, 1, :, 0], g[:, :, 0, :, 1], g[:, :, 1, :, 1]], dim=-1)
    cat = cat.reshape(N, (H // 2) * (W // 2), 4 * C)
    cat = layer_norm(cat, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return cat @ sd[pre + "reduction.weight"].t()


def batch_norm_1d(sd, pre, x, training, eps=1e-5):
    """nn.BatchNorm1d(512) (Swin_Transformer.py:494): eval uses running stats; train uses biased
    batch variance for normalisation.  (Running-stat update is a side effect tested separately.)"""
    if training:
        mu = x.mean(0)
        var = x.var(0, unbiased=False)
    else:
        mu, var = sd[pre + "running_mean"], sd[pre + "running_var"]
    return (x - mu) * torch.rsqrt(var + eps) * sd[pre + "weight"] + sd[pre + "bias"]


def output_head(sd, x, training=False, pre="output_layer."):
    """LayerNorm(768) -> flatten (token-major, then channel) -> Linear(37632,512) -> BatchNorm1d.
    Swin_Transformer.py:491-494."""
    x = layer_norm(x, sd[pre + "0.weight"], sd[pre + "0.bias"])
    x = x.reshape(x.shape[0], -1) @ sd[pre + "2.weight"].t() + sd[pre + "2.bias"]
    return batch_norm_1d(sd, pre + "3.", x, training)


def swin_forward_features(sd, x, training=False, drop_path_scales=None, return_stages=False):
    """patch_embed -> 4 stages (blocks [+ PatchMerging]) -> output head.  Swin_Transformer.py:515-531.

    drop_path_scales: optional list of 12 entries (one per block, see swin_block)."""
    x = patch_embed(sd, x)
    H = W = GRID0
    blk = 0
    stages = []
    for s, (depth, nh) in enumerate(zip(DEPTHS, HEADS)):
        for d in range(depth):
            dps = drop_path_scales[blk] if drop_path_scales is not None else None
            x = swin_block(sd, f"layers.{s}.blocks.{d}.", x, H, W, nh, 0 if d % 2 == 0 else WS // 2, dps)
            blk += 1
        if s < len(DEPTHS) - 1:
            x = patch_merging(sd, f"layers.{s}.downsample.", x, H, W)
            H, W = H // 2, W // 2
        stages.append(x)
    out = output_head(sd, x, training)
    return (out, stages) if return_stages else out


def swin_forward(sd, x, training=False, drop_path_scales=None):
    """SwinTransformer.forward (Swin_Transformer.py:533-541): a batch of one is duplicated so that
    BatchNorm sees two samples, and row 0 is returned."""
    if x.shape[0] == 1:
        return swin_forward_features(sd, torch.cat([x, x], 0), training, drop_path_scales)[:1]
    return swin_forward_features(sd, x, training, drop_path_scales)


def swin_affwild_logits(sd, x, training=False, drop_path_scales=None):
    """SwinForAffwildClassification.forward without the Gumbel step (src/models.py:27-30):
    swin -> Linear(512,64) -> ReLU -> Linear(64,7).  Keys: swin.*, linear.*, classifier.*"""
    swin_sd = {k[len("swin."):]: v for k, v in sd.items() if k.startswith("swin.")}
    f = swin_forward(swin_sd, x, training, drop_path_scales)
    h = torch.relu(f @ sd["linear.weight"].t() + sd["linear.bias"])
    return h @ sd["classifier.weight"].t() + sd["classifier.bias"]


# MACs per frame from the reference's own flops() formulas (Swin_Transformer.py:149-160,276-288,
# 333-337,424-429), used by bench.py for the roofline's algorithmic work.
def swin_macs_per_frame() -> dict:
    out = {}
    out["patch_embed"] = GRID0 * GRID0 * EMBED * 3 * 16 + GRID0 * GRID0 * EMBED
    H = GRID0
    C = EMBED
    total = out["patch_embed"]
    for s, (depth, nh) in enumerate(zip(DEPTHS, HEADS)):
        T = WS * WS
        attn = T * C * 3 * C + 2 * nh * T * (C // nh) * T + T * C * C
        blk = C * H * H + (H * H // T) * attn + 2 * H * H * C * C * 4 + C * H * H
        st = depth * blk
        if s < 3:
            st += H * H * C + (H // 2) * (H // 2) * 4 * C * 2 * C
        out[f"stage{s}"] = st
        total += st
        H //= 2
        C *= 2
    out["head"] = 49 * 768 * 512
    out["total"] = total + out["head"]
    return out
