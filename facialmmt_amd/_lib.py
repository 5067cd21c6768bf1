"""ctypes binding of libfmmt_hip.so (the C ABI declared in include/fmmt.h).

This is exactly the binding a maintainer of the reference would add: device
pointers come from ``tensor.data_ptr()``, the stream from ``torch.cuda.current_stream().cuda_stream``.
There is no fallback: a missing library or a non-zero return code raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfmmt_hip.so")

F32, BF16 = 0, 1
GENERIC = 0x100          # dtype flag of the fused Mlp entry points: the element-type-generic restatement (include/fmmt.h)
EPI_NONE, EPI_GELU, EPI_GELU_BWD, EPI_GELU_DG, EPI_MUL_AUX = 0, 1, 2, 3, 4
SAVE_DG = 0x200
BATCH_MAJOR = 0x400          # include/fmmt.h FMMT_BATCH_MAJOR: fmmt_mha_fwd / _bwd on (batch, tokens, hidden) operands
# include/fmmt.h FMMT_SAVE_DG: dtype flag of the fused Mlp entry points (h_pre = gelu'(pre-activation))
RESIZE_PIL, RESIZE_CV2 = 0, 1

_p, _i, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); mirrors include/fmmt.h one to one (tests/test_host_cpu.py::test_header_and_ctypes_signatures_agree checks both ways)
SIGNATURES = {
    "fmmt_version": (_i, []),
    "fmmt_linear_fwd": (_i, [_i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p]),
    "fmmt_linear_splitk_workspace": (_sz, [_i, _i, _i]),
    "fmmt_linear_fwd_splitk": (_i, [_i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _p, _sz, _p]),
    "fmmt_linear_wgrad_workspace": (_sz, [_i, _i, _i, _i]),
    "fmmt_linear_wgrad": (_i, [_i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _p, _sz, _p]),
    "fmmt_linear_wgrad_partials": (_i, [_i, _i, _i, _i, _p, _i, _p, _i, _i, _p, _i, _i, _p, _sz, _p]),
    "fmmt_mlp_fwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "fmmt_mlp_ln_fwd": (_i, [_i, _i, _i, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p]),
    "fmmt_mlp_bwd_input": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "fmmt_linear_ln_bwd_workspace": (_sz, [_i]),
    "fmmt_linear_ln_bwd": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "fmmt_mlp_ln_bwd_input_workspace": (_sz, [_i]),
    "fmmt_mlp_ln_bwd_input": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "fmmt_linear_wgrad_finish": (_i, [_i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "fmmt_layernorm_fwd": (_i, [_i, _i, _i, _p, _p, _p, _f, _p, _p, _p, _i, _p]),
    "fmmt_layernorm_bwd_workspace": (_sz, [_i]),
    "fmmt_layernorm_bwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _sz, _p]),
    "fmmt_window_attn_fwd": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p]),
    "fmmt_window_attn_bwd_workspace": (_sz, [_i]),
    "fmmt_window_attn_bwd": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _sz, _p]),
    "fmmt_window_block_fwd": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fmmt_window_block_fwd_ref": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fmmt_window_block_attn_bwd": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _sz, _p]),
    "fmmt_window_block_attn_bwd_ref": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _sz, _p]),
    "fmmt_linear_fwd_seg3": (_i, [_i, _i, _i, _i, _p, _i, _p, _p, _p, _i, _i, _p, _p, _p, _p, _i, _p]),
    "fmmt_select_frames_fwd": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _f, _p, _p, _p, _p]),
    "fmmt_select_frames_bwd": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "fmmt_mha_fwd": (_i, [_i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i, _f, _p, _f, _u64, _p, _p, _i, _p, _p]),
    "fmmt_mha_bwd": (_i, [_i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i, _f, _p, _f, _u64, _p, _p, _p, _i, _p, _p, _i, _p, _p, _i, _p]),
    "fmmt_mha_avg_weights": (_i, [_i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _f, _p, _f, _u64, _p, _p, _p, _p]),
    "fmmt_patch_im2col": (_i, [_i, _i, _p, _p, _p]),
    "fmmt_patch_col2im": (_i, [_i, _i, _p, _p, _p]),
    "fmmt_batchnorm1d_fwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p]),
    "fmmt_batchnorm1d_bwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "fmmt_posemb_scale_fwd": (_i, [_i, _i, _i, _i, _p, _p, _f, _p, _p]),
    "fmmt_scale": (_i, [_i, _sz, _p, _f, _p, _p]),
    "fmmt_colsum": (_i, [_i, _i, _i, _i, _p, _i, _p, _p]),
    "fmmt_cast_batch": (_i, [_i, _i, _p, _p]),
    "fmmt_layernorm_bwd_bf16_workspace": (_sz, [_i, _i]),
    "fmmt_layernorm_bwd_bf16": (_i, [_i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "fmmt_dropadd_ln_fwd": (_i, [_i, _i, _i, _f, _p, _p, _p, _p, _f, _u64, _p, _u64, _p, _p, _p]),
    "fmmt_dropadd_ln_bwd_workspace": (_sz, [_i, _i]),
    "fmmt_dropadd_ln_bwd": (_i, [_i, _i, _i, _f, _p, _p, _p, _f, _u64, _p, _u64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "fmmt_plm_dropadd_ln_fwd": (_i, [_i, _i, _f, _p, _p, _p, _p, _f, _u64, _p, _u64, _p, _p, _p]),
    "fmmt_embedding_bwd": (_i, [_i, _i, _i, _p, C.c_int64, _p, _p, _p]),
    "fmmt_plm_gelu_bwd_colsum_workspace": (_sz, [_i, _i]),
    "fmmt_plm_gelu_bwd_colsum": (_i, [_i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "fmmt_plm_dropadd_ln_bwd_workspace": (_sz, [_i, _i]),
    "fmmt_plm_dropadd_ln_bwd": (_i, [_i, _i, _f, _p, _p, _p, _f, _u64, _p, _u64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "fmmt_grad_handover": (_i, [_i, _i, _p, _p, _p, _p]),
    "fmmt_adamw_batch": (_i, [_i, _i, _p, _p, _p, _p, _f, _f, _f, _f, _f, _i, _p]),
    "fmmt_resize_table": (_i, [_i, _i, _i, _p, _p]),
    "fmmt_resize_band_rows": (_i, [_p, _i]),
    "fmmt_patch_embed_u8": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "fmmt_patch_embed_ln_fwd": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p]),
    "fmmt_patch_embed_u8_ln_fwd": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p]),
}

FMMT_EINVAL, FMMT_EALIGN, FMMT_EWORKSPACE = -1, -2, -3
_ERR = {-1: "FMMT_EINVAL (bad shape / unsupported size)", -2: "FMMT_EALIGN (pointer or leading dimension not 16-byte aligned)",
        -3: "FMMT_EWORKSPACE (workspace too small)"}

_lib = None


class FmmtError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises FmmtError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FmmtError(f"{LIB_PATH} is missing: build it with `python -m facialmmt_amd.build` "
                        f"(or __graft_entry__.build()); there is no CPU / PyTorch fallback for the hot path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = _ERR.get(rc, f"hipError_t {rc}")
        raise FmmtError(f"{what} failed: {msg}")


def dtype_code(t) -> int:
    import torch
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise FmmtError(f"unsupported activation dtype {t}: the HIP path computes in float32 (parity) or bfloat16 (throughput)")
