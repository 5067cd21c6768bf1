// Argument block of the fused Mlp kernels (mlp_fused.hip) and of their element-type-generic restatement (mlp_ref.hip).
#pragma once
#include "fmmt_common.h"

struct MlpArgs {                                              // (global: it crosses translation units, like WaArgs)
    int M;
    const bf16* x;
    const bf16* w1;
    const float* b1;
    const bf16* w2;
    const float* b2;
    const bf16* res;
    const float* rowscale;
    int rows_per_scale;
    bf16* y;
    bf16* h_pre;
    bf16* h_act;
    int tiles;
    // LN mode (fmmt_mlp_ln_fwd): x is the block's residual stream, the Mlp runs on LayerNorm(x) formed in registers, res == x
    const float* ln_g;
    const float* ln_b;
    float eps;
    bf16* xn;
    float* mean;
    float* rstd;
    // LN-backward epilogue of the input-gradient kernel (fmmt_mlp_ln_bwd_input): the LayerNorm's input, and per-workgroup partial sums
    const bf16* ln_x;
    float* ln_part;
    int dg;                                                   // FMMT_SAVE_DG: h_pre is gelu'(pre-activation), not the pre-activation (forward: written; backward: read)
};

// mlp_ref.hip: the generic instantiations (el = FMMT_F32 / FMMT_BF16; the activation / weight pointers of MlpArgs reinterpreted)
int fmmt_mlp_ref_fwd_launch(int el, int C, bool ln, const MlpArgs& a, hipStream_t st);
int fmmt_mlp_ref_bwd_launch(int el, int C, bool lnb, const MlpArgs& a, hipStream_t st);
