// Declarations shared by the fused block-half forward kernel (wblock.hip, bf16 throughput path) and its element-type-generic
// restatement (wblock_ref.hip: the same kernel written over a fragment-type trait, instantiated for fp32 operands so that the
// kernel's ALGORITHM -- window / shift addressing, fragment-order weight rows, mask derivation, base-2 softmax with the row sums
// taken after the second product -- is held to the reference-generated goldens at 1e-3, independent of bf16 rounding).
#pragma once
#include "fmmt_common.h"
#include "../../include/fmmt.h"
#include "wattn_geom.h"

namespace {

struct WbArgs {
    int n_img, H, W, shift;
    const bf16* x;                  // (the generic instantiation reinterprets the activation / weight pointers as its element type)
    const float* ln_g;
    const float* ln_b;
    float eps;
    const bf16* wqkv;
    const float* bqkv;
    const bf16* wproj;
    const float* bproj;
    const float* table;
    const int32_t* index;
    float scale;
    const float* rowscale;
    bf16* y;
    bf16* xn;
    bf16* o;
    float* mean;
    float* rstd;
    float* lse;
    int B_;
};

constexpr int WB_BP = 68;            // bias row pitch in floats (272 B: 16 query rows fall on 16 different 16-byte slots)

template <int C>
struct WbLds {
    static constexpr int NH = C / 32, PITCH = C + 8;
    static constexpr int W_BYTES = 4 * C * PITCH * 2;                 // 3C rows of Wqkv + C rows of Wproj, fragment order
    static constexpr int BIAS_BYTES = NH * TOK * WB_BP * 4;
    static constexpr int VEC_BYTES = (C + C + 3 * C + C) * 4;         // gamma, beta, bqkv, bproj
    static constexpr int TOTAL = W_BYTES + BIAS_BYTES + VEC_BYTES;
};

// LDS weight row d -> source row.  Rows are stored in FRAGMENT order: a 16-row MFMA tile reads 16 consecutive LDS rows.
// d < 3C : ((head * 3 + part) * 2 + nt) * 16 + i  <->  Wqkv row part * C + head * 32 + (i >> 2) * 8 + nt * 4 + (i & 3)
// d >= 3C: 3C + (c * 2 + nt) * 16 + i             <->  Wproj row c * 32 + (i >> 2) * 8 + nt * 4 + (i & 3)
template <int C>
__device__ __forceinline__ int wb_src_row(int d, bool& is_proj) {
    is_proj = d >= 3 * C;
    const int dd = is_proj ? d - 3 * C : d;
    const int blk = dd >> 5, nt = (dd >> 4) & 1, i = dd & 15;
    const int within = (i >> 2) * 8 + nt * 4 + (i & 3);
    if (is_proj) return blk * 32 + within;
    const int head = blk / 3, part = blk - head * 3;
    return part * C + head * 32 + within;
}

constexpr float WB_LOG2E = 1.4426950408889634f, WB_LN2 = 0.6931471805599453f;

}  // namespace

int fmmt_wblock_ref_fwd_launch(int dtype, const void* args, hipStream_t st);     // wblock_ref.hip (args: a WbArgs)
