"""`torch.ops.fmmt.*`: the hot-path kernels as registered PyTorch custom operators (torch.library), with autograd
formulas and shape ("fake") functions, for callers that want dispatcher-visible ops (`torch.ops.fmmt.*`)
instead of the `torch.autograd.Function`s of `facialmmt_amd/ops.py`.  Both front ends are thin: every operator body is
the same call into the C ABI (`include/fmmt.h`) through `ops.*_raw` -- there is no second implementation, and no CPU
kernel is registered (device_types="cuda": a CPU tensor raises from the dispatcher).

Registered (forward ops return what their backward needs as extra outputs, as custom ops must):

    fmmt::linear(x, weight, bias?, res?, rowscale?, rows_per_scale) -> y                  nn.Linear (+ residual, DropPath scale)
    fmmt::mlp(x, w1, b1, w2, b2, res?, rowscale?, rows_per_scale) -> (y, h_dg, h_act)     Mlp of a Swin block (fused for C = 96 / 192); h_dg = what the backward needs
                                                                                           of the pre-activation: its gelu' (round 6; the pre-activation itself where ops.mlp_saves_dg says no)
    fmmt::layer_norm(x, gamma, beta, eps) -> (y, mean, rstd)                              nn.LayerNorm
    fmmt::window_attention(qkv, table, index, n_img, H, W, heads, shift, scale) -> (out, lse)   W-MSA / SW-MSA core on token-order qkv
    fmmt::patch_embed_u8(img_u8, flavour, bf16) -> cols                                    input pre-step + patch gather (no gradient)
    fmmt::layer_norm_merge(x, gamma, beta, eps, merge_hw) -> (y, mean, rstd)              PatchMerging's 2x2 gather + LayerNorm(4C)
    fmmt::linear_splitk(x, weight, bias?) -> y                                            the 37632 -> 512 embedding head (split-K launch)
    fmmt::batch_norm_1d(x, gamma, beta, running_mean, running_var, momentum, eps, training) -> (y, save_mean, save_invstd, running_mean', running_var')
    fmmt::mha(q, k, v?, key_bias?, heads, scale, dropout_p, seed) -> (out, lse)           cross-modal attention core (packed k|v when v is None)
    fmmt::posemb_scale(x, table, scale) -> y                                              sqrt(E) x + sinusoidal position embedding
    fmmt::window_block(x, ln_w, ln_b, eps, wqkv, bqkv?, wproj, bproj?, table, index, n_img, H, W, heads, shift, scale, rowscale?)
                       -> (y, xn, attn_out, mean, rstd, lse)                               norm1 -> (S)W-MSA -> proj -> residual (C = 96: one launch; 192: four), recompute backward

The nn.Modules of facialmmt_amd/modules keep using ops.py (fewer dispatcher hops per launch); tests/test_gpu_torch_ops.py
holds the two front ends bit-identical, forward and backward, and runs torch.library.opcheck on each operator."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import EPI_GELU, EPI_GELU_BWD, EPI_GELU_DG, EPI_MUL_AUX

_LIB = "fmmt"


def _cast(w: Tensor, dtype, transpose: bool = False) -> Tensor:
    """weight in the activation dtype (no shadow cache here: the dispatcher hands operators fresh tensor wrappers)"""
    w = w.detach().to(dtype)
    return w.t().contiguous() if transpose else w.contiguous()


# ------------------------------------------------------------------------------------------------ linear
@torch.library.custom_op(f"{_LIB}::linear", mutates_args=(), device_types="cuda")
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], res: Optional[Tensor], rowscale: Optional[Tensor], rows_per_scale: int) -> Tensor:
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    res2 = res.reshape(-1, weight.shape[0]).contiguous() if res is not None else None
    y = ops.linear_raw(x2, _cast(weight, x.dtype), bias, res=res2, rowscale=rowscale, rows_per_scale=rows_per_scale)
    return y.reshape(*x.shape[:-1], weight.shape[0])


@linear.register_fake
def _(x, weight, bias, res, rowscale, rows_per_scale):
    return x.new_empty(*x.shape[:-1], weight.shape[0])


def _linear_setup(ctx, inputs, output):
    x, weight, bias, res, rowscale, rps = inputs
    ctx.save_for_backward(x, weight, rowscale)
    ctx.has_bias, ctx.has_res, ctx.rps = bias is not None, res is not None, rps


def _linear_backward(ctx, dy):
    x, weight, rowscale = ctx.saved_tensors
    N = weight.shape[0]
    dy2 = dy.reshape(-1, N).contiguous()
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = ops.linear_raw(dy2, _cast(weight, dy2.dtype, True), None, rowscale=rowscale, rows_per_scale=ctx.rps).reshape(x.shape)
    if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
        dw, db = ops.wgrad_raw(dy2, x2, ctx.has_bias, rowscale, ctx.rps)
    return dx, dw, db, (dy if ctx.has_res else None), None, None


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


# ------------------------------------------------------------------------------------------------ mlp
@torch.library.custom_op(f"{_LIB}::mlp", mutates_args=(), device_types="cuda")
def mlp(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, res: Optional[Tensor], rowscale: Optional[Tensor],
        rows_per_scale: int) -> Tuple[Tensor, Tensor, Tensor]:
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    w1l, w2l = _cast(w1, x.dtype), _cast(w2, x.dtype)
    res2 = res.reshape(-1, w2.shape[0]).contiguous() if res is not None else None
    h_pre = torch.empty((x2.shape[0], w1.shape[0]), dtype=x.dtype, device=x.device)
    dg = ops.mlp_saves_dg(x2, w1, w2, b1, b2)
    if ops._mlp_fusable(x2, w1, w2, b1, b2):
        h = torch.empty_like(h_pre)
        y = ops.mlp_fused_raw(x2, w1l, b1.float().contiguous(), w2l, b2.float().contiguous(), res2, rowscale, rows_per_scale, h_pre, h, dg)
    else:
        h = ops.linear_raw(x2, w1l, b1, epi=EPI_GELU_DG if dg else EPI_GELU, y_pre=h_pre)
        y = ops.linear_raw(h, w2l, b2, res=res2, rowscale=rowscale, rows_per_scale=rows_per_scale)
    return y.reshape(*x.shape[:-1], w2.shape[0]), h_pre, h


@mlp.register_fake
def _(x, w1, b1, w2, b2, res, rowscale, rows_per_scale):
    m = x.numel() // x.shape[-1]
    return x.new_empty(*x.shape[:-1], w2.shape[0]), x.new_empty(m, w1.shape[0]), x.new_empty(m, w1.shape[0])


def _mlp_setup(ctx, inputs, output):
    x, w1, b1, w2, b2, res, rowscale, rps = inputs
    _, h_pre, h = output
    ctx.save_for_backward(x, w1, w2, h_pre, h, rowscale)
    ctx.has_res, ctx.rps = res is not None, rps
    ctx.dg = ops.mlp_saves_dg(x.reshape(-1, x.shape[-1]), w1, w2, b1, b2)
    ctx.set_materialize_grads(False)


def _mlp_backward(ctx, dy, _dpre, _dact):
    x, w1, w2, h_pre, h, rowscale = ctx.saved_tensors
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    dy2 = dy.reshape(-1, w2.shape[0]).contiguous()
    dh = ops.linear_raw(dy2, _cast(w2, dy2.dtype, True), None, epi=EPI_MUL_AUX if ctx.dg else EPI_GELU_BWD, aux=h_pre, rowscale=rowscale, rows_per_scale=ctx.rps)
    dw2, db2 = ops.wgrad_raw(dy2, h, True, rowscale, ctx.rps)
    dx = ops.linear_raw(dh, _cast(w1, dy2.dtype, True), None).reshape(x.shape) if ctx.needs_input_grad[0] else None
    dw1, db1 = ops.wgrad_raw(dh, x2, True)
    return dx, dw1, db1, dw2, db2, (dy if ctx.has_res else None), None, None


mlp.register_autograd(_mlp_backward, setup_context=_mlp_setup)


# ------------------------------------------------------------------------------------------------ layer_norm
@torch.library.custom_op(f"{_LIB}::layer_norm", mutates_args=(), device_types="cuda")
def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    from . import _lib
    lib = _lib.load()
    x = x.contiguous()
    C = x.shape[-1]
    M = x.numel() // C
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    g, b = gamma.float().contiguous(), beta.float().contiguous()
    rc = lib.fmmt_layernorm_fwd(_lib.dtype_code(x.dtype), M, C, x.data_ptr(), g.data_ptr(), b.data_ptr(), eps, y.data_ptr(), mean.data_ptr(),
                                rstd.data_ptr(), 0, ops._st())
    _lib.check(rc, f"fmmt_layernorm_fwd(M={M},C={C})")
    return y, mean, rstd


@layer_norm.register_fake
def _(x, gamma, beta, eps):
    m = x.numel() // x.shape[-1]
    return torch.empty_like(x), x.new_empty(m, dtype=torch.float32), x.new_empty(m, dtype=torch.float32)


def _ln_setup(ctx, inputs, output):
    x, gamma, beta, eps = inputs
    _, mean, rstd = output
    ctx.save_for_backward(x, mean, rstd, gamma)
    ctx.set_materialize_grads(False)


def _ln_backward(ctx, dy, _dmean, _drstd):
    from . import _lib
    x, mean, rstd, gamma = ctx.saved_tensors
    lib = _lib.load()
    x = x.contiguous()
    C = x.shape[-1]
    M = x.numel() // C
    dy = dy.contiguous()
    g = gamma.float().contiguous()
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    nbytes = lib.fmmt_layernorm_bwd_workspace(C)
    ws = ops._ws(nbytes, x.device)
    rc = lib.fmmt_layernorm_bwd(_lib.dtype_code(x.dtype), M, C, dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), None,
                                dx.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), nbytes, ops._st())
    _lib.check(rc, f"fmmt_layernorm_bwd(M={M},C={C})")
    return dx, dg, db, None


layer_norm.register_autograd(_ln_backward, setup_context=_ln_setup)


# ------------------------------------------------------------------------------------------------ window attention
@torch.library.custom_op(f"{_LIB}::window_attention", mutates_args=(), device_types="cuda")
def window_attention(qkv: Tensor, table: Tensor, index_i32: Tensor, n_img: int, H: int, W: int, num_heads: int, shift: int,
                     scale: float) -> Tuple[Tensor, Tensor]:
    """qkv (n_img*H*W, 3C) in token order; table (169, heads) fp32; the shift mask is the standard SW-MSA one (derived from
    the window coordinates inside the kernel)."""
    from . import _lib
    lib = _lib.load()
    qkv = qkv.contiguous()
    C = qkv.shape[-1] // 3
    nW = (H // 7) * (W // 7)
    out = torch.empty((n_img * H * W, C), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((n_img * nW * num_heads * 49,), dtype=torch.float32, device=qkv.device)
    tab = table.float().contiguous()
    mask = None
    if shift:
        from .modules.SwinTransformer.Swin_Transformer import build_shift_mask
        mask = build_shift_mask(H, W, 7, shift).to(qkv.device)
    rc = lib.fmmt_window_attn_fwd(_lib.dtype_code(qkv.dtype), n_img, H, W, C, num_heads, shift, qkv.data_ptr(), tab.data_ptr(), index_i32.data_ptr(),
                                  mask.data_ptr() if mask is not None else None, nW if mask is not None else 0, 1 if mask is not None else 0,
                                  scale, out.data_ptr(), lse.data_ptr(), ops._st())
    _lib.check(rc, "fmmt_window_attn_fwd")
    return out, lse


@window_attention.register_fake
def _(qkv, table, index_i32, n_img, H, W, num_heads, shift, scale):
    C = qkv.shape[-1] // 3
    return qkv.new_empty(n_img * H * W, C), qkv.new_empty(n_img * (H // 7) * (W // 7) * num_heads * 49, dtype=torch.float32)


def _wa_setup(ctx, inputs, output):
    qkv, table, index_i32, n_img, H, W, num_heads, shift, scale = inputs
    out, lse = output
    ctx.save_for_backward(qkv, out, lse, table, index_i32)
    ctx.cfg = (n_img, H, W, num_heads, shift, scale)
    ctx.set_materialize_grads(False)


def _wa_backward(ctx, dout, _dlse):
    from . import _lib
    qkv, out, lse, table, index_i32 = ctx.saved_tensors
    n_img, H, W, num_heads, shift, scale = ctx.cfg
    lib = _lib.load()
    qkv = qkv.contiguous()
    C = qkv.shape[-1] // 3
    nW = (H // 7) * (W // 7)
    tab = table.float().contiguous()
    mask = None
    if shift:
        from .modules.SwinTransformer.Swin_Transformer import build_shift_mask
        mask = build_shift_mask(H, W, 7, shift).to(qkv.device)
    dout = dout.contiguous()
    dqkv = torch.empty_like(qkv)
    dtable = torch.empty_like(tab)
    nbytes = lib.fmmt_window_attn_bwd_workspace(num_heads)
    ws = ops._ws(nbytes, qkv.device)
    rc = lib.fmmt_window_attn_bwd(_lib.dtype_code(qkv.dtype), n_img, H, W, C, num_heads, shift, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                  lse.data_ptr(), tab.data_ptr(), index_i32.data_ptr(), mask.data_ptr() if mask is not None else None,
                                  nW if mask is not None else 0, 1 if mask is not None else 0, scale, dqkv.data_ptr(), dtable.data_ptr(),
                                  ws.data_ptr(), nbytes, ops._st())
    _lib.check(rc, "fmmt_window_attn_bwd")
    return dqkv, dtable.to(table.dtype), None, None, None, None, None, None, None


window_attention.register_autograd(_wa_backward, setup_context=_wa_setup)


# ------------------------------------------------------------------------------------------------ input pre-step
@torch.library.custom_op(f"{_LIB}::patch_embed_u8", mutates_args=(), device_types="cuda")
def patch_embed_u8(img_u8: Tensor, flavour: str, bf16: bool) -> Tensor:
    return ops.patch_embed_u8(img_u8, flavour, torch.bfloat16 if bf16 else torch.float32)


@patch_embed_u8.register_fake
def _(img_u8, flavour, bf16):
    return img_u8.new_empty(img_u8.shape[0] * 3136, 48, dtype=torch.bfloat16 if bf16 else torch.float32)


# ------------------------------------------------------------------------------------------------ PatchMerging gather + LayerNorm
def _ln_call(x, gamma, beta, eps, merge_hw):
    from . import _lib
    lib = _lib.load()
    x = x.contiguous()
    if merge_hw:
        n, L, Cq = x.shape
        M, C = n * (merge_hw // 2) ** 2, 4 * Cq
        y = torch.empty((n, (merge_hw // 2) ** 2, C), dtype=x.dtype, device=x.device)
    else:
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    g, b = gamma.float().contiguous(), beta.float().contiguous()
    rc = lib.fmmt_layernorm_fwd(_lib.dtype_code(x.dtype), M, C, x.data_ptr(), g.data_ptr(), b.data_ptr(), eps, y.data_ptr(), mean.data_ptr(),
                                rstd.data_ptr(), merge_hw, ops._st())
    _lib.check(rc, f"fmmt_layernorm_fwd(M={M},C={C},merge={merge_hw})")
    return y, mean, rstd


@torch.library.custom_op(f"{_LIB}::layer_norm_merge", mutates_args=(), device_types="cuda")
def layer_norm_merge(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, merge_hw: int) -> Tuple[Tensor, Tensor, Tensor]:
    """x (n, H*W, C/4) token grid, H = W = merge_hw -> LayerNorm over the 2x2-neighbour concat (n, H*W/4, C)  (Swin_Transformer.py:316-325)"""
    return _ln_call(x, gamma, beta, eps, merge_hw)


@layer_norm_merge.register_fake
def _(x, gamma, beta, eps, merge_hw):
    n, L, Cq = x.shape
    m = n * (merge_hw // 2) ** 2
    return x.new_empty(n, (merge_hw // 2) ** 2, 4 * Cq), x.new_empty(m, dtype=torch.float32), x.new_empty(m, dtype=torch.float32)


def _lnm_setup(ctx, inputs, output):
    x, gamma, beta, eps, merge_hw = inputs
    _, mean, rstd = output
    ctx.save_for_backward(x, mean, rstd, gamma)
    ctx.merge_hw = merge_hw
    ctx.set_materialize_grads(False)


def _lnm_backward(ctx, dy, _dmean, _drstd):
    from . import _lib
    x, mean, rstd, gamma = ctx.saved_tensors
    lib = _lib.load()
    x = x.contiguous()
    n, L, Cq = x.shape
    M, C = n * (ctx.merge_hw // 2) ** 2, 4 * Cq
    dy = dy.contiguous()
    g = gamma.float().contiguous()
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    nbytes = lib.fmmt_layernorm_bwd_workspace(C)
    ws = ops._ws(nbytes, x.device)
    rc = lib.fmmt_layernorm_bwd(_lib.dtype_code(x.dtype), M, C, dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), None,
                                dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ctx.merge_hw, ws.data_ptr(), nbytes, ops._st())
    _lib.check(rc, f"fmmt_layernorm_bwd(M={M},C={C},merge={ctx.merge_hw})")
    return dx, dg, db, None, None


layer_norm_merge.register_autograd(_lnm_backward, setup_context=_lnm_setup)


# ------------------------------------------------------------------------------------------------ split-K head Linear
@torch.library.custom_op(f"{_LIB}::linear_splitk", mutates_args=(), device_types="cuda")
def linear_splitk(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """y = x w^T + b for few rows and a very long K (the 49*768 -> 512 embedding head, Swin_Transformer.py:493): fmmt_linear_fwd_splitk
    where the library says the shape is a split-K shape, fmmt_linear_fwd otherwise (ops.linear_raw makes that choice)"""
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    return ops.linear_raw(x2, _cast(weight, x.dtype), bias).reshape(*x.shape[:-1], weight.shape[0])


@linear_splitk.register_fake
def _(x, weight, bias):
    return x.new_empty(*x.shape[:-1], weight.shape[0])


def _lsk_setup(ctx, inputs, output):
    x, weight, bias = inputs
    ctx.save_for_backward(x, weight)
    ctx.has_bias = bias is not None


def _lsk_backward(ctx, dy):
    x, weight = ctx.saved_tensors
    dy2 = dy.reshape(-1, weight.shape[0]).contiguous()
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    dx = ops.linear_raw(dy2, _cast(weight, dy2.dtype, True), None).reshape(x.shape) if ctx.needs_input_grad[0] else None
    dw, db = ops.wgrad_raw(dy2, x2, ctx.has_bias)
    return dx, dw, db


linear_splitk.register_autograd(_lsk_backward, setup_context=_lsk_setup)


# ------------------------------------------------------------------------------------------------ BatchNorm1d of the embedding head
@torch.library.custom_op(f"{_LIB}::batch_norm_1d", mutates_args=(), device_types="cuda")
def batch_norm_1d(x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor, momentum: float, eps: float,
                  training: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """functional form (an operator with an autograd formula may not mutate its inputs): the updated running statistics are
    returned as outputs 3 and 4 -- nn.BatchNorm1d semantics are `running_mean.copy_(out[3]); running_var.copy_(out[4])`"""
    rm, rv = running_mean.clone(), running_var.clone()
    y, sm, si = ops.batch_norm_1d_fwd_raw(x.contiguous(), gamma.float().contiguous(), beta.float().contiguous(), rm, rv, momentum, eps, training)
    return y, sm, si, rm, rv


@batch_norm_1d.register_fake
def _(x, gamma, beta, running_mean, running_var, momentum, eps, training):
    C = x.shape[1]
    return (torch.empty_like(x), x.new_empty(C, dtype=torch.float32), x.new_empty(C, dtype=torch.float32),
            torch.empty_like(running_mean), torch.empty_like(running_var))


def _bn_setup(ctx, inputs, output):
    x, gamma, beta, rm, rv, momentum, eps, training = inputs
    _, sm, si, _rm, _rv = output
    ctx.save_for_backward(x, gamma, sm, si)
    ctx.training = bool(training)
    ctx.set_materialize_grads(False)


def _bn_backward(ctx, dy, _dsm, _dsi, _drm, _drv):
    x, gamma, sm, si = ctx.saved_tensors
    dx, dg, db = ops.batch_norm_1d_bwd_raw(dy.contiguous(), x.contiguous(), gamma.float().contiguous(), sm, si, ctx.training)
    return dx, dg, db, None, None, None, None, None


batch_norm_1d.register_autograd(_bn_backward, setup_context=_bn_setup)


# ------------------------------------------------------------------------------------------------ cross-modal attention core
@torch.library.custom_op(f"{_LIB}::mha", mutates_args=(), device_types="cuda")
def mha(q: Tensor, k: Tensor, v: Optional[Tensor], key_bias: Optional[Tensor], num_heads: int, scale: float, dropout_p: float,
        seed: int) -> Tuple[Tensor, Tensor]:
    """q (Lq, B, E), k / v (Lk, B, E) time-major -- or k = the packed (Lk, B, 2E) [k | v] projection and v None; key_bias (B, Lk) fp32
    additive logit bias; dropout mask = hash(seed, element), replayed by the backward (multihead_attention.py:85-128)"""
    kb = key_bias.to(torch.float32).contiguous() if key_bias is not None else None
    return ops.mha_fwd_raw(q.contiguous(), k.contiguous(), v.contiguous() if v is not None else None, num_heads, scale, dropout_p, seed, None, kb)


@mha.register_fake
def _(q, k, v, key_bias, num_heads, scale, dropout_p, seed):
    Lq, B, E = q.shape
    return torch.empty_like(q), q.new_empty(B * num_heads * Lq, dtype=torch.float32)


def _mha_setup(ctx, inputs, output):
    q, k, v, key_bias, num_heads, scale, dropout_p, seed = inputs
    out, lse = output
    ctx.save_for_backward(q, k, v, key_bias, out, lse)
    ctx.cfg = (num_heads, scale, dropout_p, seed)
    ctx.set_materialize_grads(False)


def _mha_backward(ctx, dout, _dlse):
    q, k, v, key_bias, out, lse = ctx.saved_tensors
    num_heads, scale, dropout_p, seed = ctx.cfg
    kb = key_bias.to(torch.float32).contiguous() if key_bias is not None else None
    dq, dk, dv = ops.mha_bwd_raw(q.contiguous(), k.contiguous(), v.contiguous() if v is not None else None, out, dout.contiguous(), lse,
                                 num_heads, scale, dropout_p, seed, None, kb)
    return dq, dk, dv, None, None, None, None, None


mha.register_autograd(_mha_backward, setup_context=_mha_setup)


# ------------------------------------------------------------------------------------------------ cross-modal input embedding
@torch.library.custom_op(f"{_LIB}::posemb_scale", mutates_args=(), device_types="cuda")
def posemb_scale(x: Tensor, table: Tensor, scale: float) -> Tensor:
    """y = scale * x + table[position], position of (t, b) = t + 1 where x[t, b, 0] != 0, else 0 (CrossmodalTransformer.py:63-66)"""
    from . import _lib
    x = x.contiguous()
    L, B, E = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.load().fmmt_posemb_scale_fwd(_lib.dtype_code(x.dtype), L, B, E, x.data_ptr(), table.data_ptr(), scale, y.data_ptr(), ops._st()),
               f"fmmt_posemb_scale_fwd(L={L},B={B},E={E})")
    return y


@posemb_scale.register_fake
def _(x, table, scale):
    return torch.empty_like(x)


def _pe_setup(ctx, inputs, output):
    ctx.scale = inputs[2]


def _pe_backward(ctx, dy):
    from . import _lib
    dy = dy.contiguous()
    dx = torch.empty_like(dy)
    _lib.check(_lib.load().fmmt_scale(_lib.dtype_code(dy.dtype), dy.numel(), dy.data_ptr(), ctx.scale, dx.data_ptr(), ops._st()), "fmmt_scale")
    return dx, None, None


posemb_scale.register_autograd(_pe_backward, setup_context=_pe_setup)


# ------------------------------------------------------------------------------------------------ fused attention half of a Swin block
@torch.library.custom_op(f"{_LIB}::window_block", mutates_args=(), device_types="cuda")
def window_block(x: Tensor, ln_w: Tensor, ln_b: Tensor, eps: float, wqkv: Tensor, bqkv: Optional[Tensor], wproj: Tensor, bproj: Optional[Tensor],
                 table: Tensor, index_i32: Tensor, n_img: int, H: int, W: int, num_heads: int, shift: int, scale: float,
                 rowscale: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """y = x + rowscale * (proj(W-MSA(LN(x) wqkv^T + bqkv)) + bproj) (bf16; C = 96: one launch, fmmt_window_block_fwd; C = 192: four launches); shift > 0 uses the
    standard SW-MSA mask of (H, W, shift).  Also returns LN(x), the attention output, the row statistics and the log-sum-exp."""
    C = x.shape[-1]
    mask = None
    if shift:
        from .modules.SwinTransformer.Swin_Transformer import build_shift_mask
        mask = build_shift_mask(H, W, 7, shift).to(x.device)
    y, xn, o, mean, rstd, lse = ops.window_block_raw(
        x.reshape(-1, C).contiguous(), n_img, H, W, num_heads, shift, ln_w.float().contiguous(), ln_b.float().contiguous(), eps, _cast(wqkv, x.dtype),
        bqkv.float().contiguous() if bqkv is not None else None, _cast(wproj, x.dtype), bproj.float().contiguous() if bproj is not None else None,
        table.float().contiguous(), index_i32, scale, rowscale, True, mask)
    return y.reshape(x.shape), xn, o, mean, rstd, lse


@window_block.register_fake
def _(x, ln_w, ln_b, eps, wqkv, bqkv, wproj, bproj, table, index_i32, n_img, H, W, num_heads, shift, scale, rowscale):
    C = x.shape[-1]
    m = x.numel() // C
    f32 = dict(dtype=torch.float32)
    return (torch.empty_like(x), x.new_empty(m, C), x.new_empty(m, C), x.new_empty(m, **f32), x.new_empty(m, **f32),
            x.new_empty(n_img * (H // 7) * (W // 7) * num_heads * 49, **f32))


def _wb_setup(ctx, inputs, output):
    x, ln_w, ln_b, eps, wqkv, bqkv, wproj, bproj, table, index_i32, n_img, H, W, num_heads, shift, scale, rowscale = inputs
    _, xn, o, mean, rstd, lse = output
    ctx.save_for_backward(x, xn, o, mean, rstd, lse, ln_w, wqkv, bqkv, wproj, table, index_i32, rowscale)
    ctx.cfg = (n_img, H, W, num_heads, shift, scale, bproj is not None)
    ctx.set_materialize_grads(False)


def _wb_backward(ctx, dy, *_unused):
    x, xn, o, mean, rstd, lse, ln_w, wqkv, bqkv, wproj, table, index_i32, rowscale = ctx.saved_tensors
    n_img, H, W, num_heads, shift, scale, has_bproj = ctx.cfg
    mask = None
    if shift:
        from .modules.SwinTransformer.Swin_Transformer import build_shift_mask
        mask = build_shift_mask(H, W, 7, shift).to(x.device)
    C = x.shape[-1]
    g = ops.window_block_backward(dy, x.reshape(-1, C).contiguous(), xn, o, mean, rstd, lse, ln_w.float().contiguous(), wqkv, bqkv, wproj,
                                  table.float().contiguous(), index_i32, mask, rowscale, (n_img, H, W, C, num_heads, shift, scale, x.shape))
    dx, dg, db, dwq, dbq, dwp, dbp, dtable = g
    return (dx, dg, db, None, dwq, dbq, dwp, dbp if has_bproj else None, dtable.to(table.dtype), None, None, None, None, None, None, None, None)


window_block.register_autograd(_wb_backward, setup_context=_wb_setup)
