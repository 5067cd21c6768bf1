"""`torch.ops.fmmt.*`: the hot-path kernels as registered PyTorch custom operators (torch.library), with autograd
formulas and shape ("fake") functions, for callers that want dispatcher-visible ops (`torch.ops.fmmt.*`)
instead of the `torch.autograd.Function`s of `facialmmt_amd/ops.py`.  Both front ends are thin: every operator body is
the same call into the C ABI (`include/fmmt.h`) through `ops.*_raw` -- there is no second implementation, and no CPU
kernel is registered (device_types="cuda": a CPU tensor raises from the dispatcher).

Registered (forward ops return what their backward needs as extra outputs, as custom ops must):

    fmmt::linear(x, weight, bias?, res?, rowscale?, rows_per_scale) -> y                  nn.Linear (+ residual, DropPath scale)
    fmmt::mlp(x, w1, b1, w2, b2, res?, rowscale?, rows_per_scale) -> (y, h_pre, h_act)    Mlp of a Swin block (fused for C = 96 / 192)
    fmmt::layer_norm(x, gamma, beta, eps) -> (y, mean, rstd)                              nn.LayerNorm
    fmmt::window_attention(qkv, table, index, n_img, H, W, heads, shift, scale) -> (out, lse)   W-MSA / SW-MSA core on token-order qkv
    fmmt::patch_embed_u8(img_u8, flavour, bf16) -> cols                                    input pre-step + patch gather (no gradient)

The nn.Modules of facialmmt_amd/modules keep using ops.py (fewer dispatcher hops per launch); tests/test_gpu_torch_ops.py
holds the two front ends bit-identical, forward and backward, and runs torch.library.opcheck on each operator."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import EPI_GELU, EPI_GELU_BWD

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
    if ops._mlp_fusable(x2, w1, w2, b1, b2):
        h = torch.empty_like(h_pre)
        y = ops.mlp_fused_raw(x2, w1l, b1.float().contiguous(), w2l, b2.float().contiguous(), res2, rowscale, rows_per_scale, h_pre, h)
    else:
        h = ops.linear_raw(x2, w1l, b1, epi=EPI_GELU, y_pre=h_pre)
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
    ctx.set_materialize_grads(False)


def _mlp_backward(ctx, dy, _dpre, _dact):
    x, w1, w2, h_pre, h, rowscale = ctx.saved_tensors
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    dy2 = dy.reshape(-1, w2.shape[0]).contiguous()
    dh = ops.linear_raw(dy2, _cast(w2, dy2.dtype, True), None, epi=EPI_GELU_BWD, aux=h_pre, rowscale=rowscale, rows_per_scale=ctx.rps)
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
