// PatchEmbed's projection + bias + LayerNorm on ONE 16-patch tile of a wave (Swin_Transformer.py:392-422), shared by patch_ln.hip (the patch
// matrix read from memory) and preproc.hip (the patch matrix formed in LDS by the uint8 pre-step of the same launch).
//   D = W (A side, 16-channel tiles in the chan_of<24> order) . cols^T: a lane ends up with the channels c * 32 + lg * 8 + e of patch li -- the whole
//   row in the four lanes li + 16 g, so the LayerNorm is in-lane sums + two swaps.  K = 48 = one 32-deep step plus one whose upper half is zero.
//   x_pre = T(acc + bias); statistics over the ROUNDED x_pre values of the row (the ones fmmt_layernorm_bwd finds when it re-reads x_pre).
#pragma once
#include "gemm_common.h"
#include "elem_trait.h"

template <typename T>
struct PatchLnParams {                                      // per-lane: twelve weight fragments, 3 x 24 per-channel constants
    typename ElemTrait<T>::frag wf[6][2];
    float bia[24], gam[24], bet[24];
};

template <typename T>
__device__ __forceinline__ void patch_ln_load(PatchLnParams<T>& P, const T* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, int li, int lg) {
    using E = ElemTrait<T>;
    constexpr int K = 48, NT = 6, KS = 3;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ch = chan_of<4 * NT>(nt, li >> 2, li & 3);              // the channel that row li of accumulator tile nt holds
        P.wf[nt][0] = E::ld(w + (size_t)ch * K + lg * 8);
        P.wf[nt][1] = lg < 2 ? E::ld(w + (size_t)ch * K + 32 + lg * 8) : E::zero();
    }
#pragma unroll
    for (int c = 0; c < KS; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = c * 32 + lg * 8 + e;
            P.bia[c * 8 + e] = bias ? bias[ch] : 0.f;
            P.gam[c * 8 + e] = gamma[ch];
            P.bet[c * 8 + e] = beta[ch];
        }
}

// c0 / c1: the lane's two B fragments of patch `tok` (k = lg * 8 .. + 7 and, for lg < 2, 32 + lg * 8 .. + 7; zero otherwise)
template <typename T>
__device__ __forceinline__ void patch_ln_tile(const PatchLnParams<T>& P, const typename ElemTrait<T>::frag& c0, const typename ElemTrait<T>::frag& c1, float eps,
                                              size_t tok, bool valid, int lg, T* __restrict__ x_pre, T* __restrict__ y, float* __restrict__ mean_out,
                                              float* __restrict__ rstd_out) {
    using E = ElemTrait<T>;
    using F = typename E::frag;
    constexpr int C = 96, NT = 6, KS = 3;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        acc[nt] = E::mma(P.wf[nt][0], c0, f32x4{0.f, 0.f, 0.f, 0.f});
        acc[nt] = E::mma(P.wf[nt][1], c1, acc[nt]);
    }
    float v[KS * 8], sum = 0.f;
#pragma unroll
    for (int c = 0; c < KS; ++c) {
        F o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = E::cv(acc[2 * c + (e >> 2)][e & 3] + P.bia[c * 8 + e]);
            v[c * 8 + e] = (float)o[e];
            sum += v[c * 8 + e];
        }
        if (x_pre && valid) E::st(x_pre + tok * C + c * 32 + lg * 8, o);
    }
    const float mean = swap_sum(sum) * (1.0f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < KS * 8; ++i) {
        v[i] -= mean;
        q += v[i] * v[i];
    }
    const float rstd = rsqrtf(swap_sum(q) * (1.0f / (float)C) + eps);
    if (valid) {
#pragma unroll
        for (int c = 0; c < KS; ++c) {
            F o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = E::cv(v[c * 8 + e] * rstd * P.gam[c * 8 + e] + P.bet[c * 8 + e]);
            E::st(y + tok * C + c * 32 + lg * 8, o);
        }
        if (mean_out && lg == 0) {
            mean_out[tok] = mean;
            rstd_out[tok] = rstd;
        }
    }
}


// The same tile with every per-channel constant loaded where it is used (weights -> products -> bias -> statistics -> gamma / beta) instead of held in a
// PatchLnParams for the kernel's life: for a caller that runs ONE tile per wave (preproc.hip: the uint8 pre-step's workgroup) the 120 registers of the
// parameter block only cost occupancy -- 172 registers, two workgroups per CU under a latency-bound gather.  Same operations in the same order as
// patch_ln_load + patch_ln_tile: bit-identical results.
template <typename T>
__device__ __forceinline__ void patch_ln_tile_lean(const T* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const typename ElemTrait<T>::frag& c0, const typename ElemTrait<T>::frag& c1,
                                                   float eps, size_t tok, bool valid, int li, int lg, T* __restrict__ x_pre, T* __restrict__ y,
                                                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    using E = ElemTrait<T>;
    using F = typename E::frag;
    constexpr int C = 96, NT = 6, KS = 3, K = 48;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ch = chan_of<4 * NT>(nt, li >> 2, li & 3);
        const F w0 = E::ld(w + (size_t)ch * K + lg * 8);
        const F w1 = lg < 2 ? E::ld(w + (size_t)ch * K + 32 + lg * 8) : E::zero();
        acc[nt] = E::mma(w0, c0, f32x4{0.f, 0.f, 0.f, 0.f});
        acc[nt] = E::mma(w1, c1, acc[nt]);
    }
    float v[KS * 8], sum = 0.f;
#pragma unroll
    for (int c = 0; c < KS; ++c) {
        F o;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            b0 = *reinterpret_cast<const f32x4*>(bias + c * 32 + lg * 8);
            b1 = *reinterpret_cast<const f32x4*>(bias + c * 32 + lg * 8 + 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = E::cv(acc[2 * c + (e >> 2)][e & 3] + (e < 4 ? b0[e & 3] : b1[e & 3]));
            v[c * 8 + e] = (float)o[e];
            sum += v[c * 8 + e];
        }
        if (x_pre && valid) E::st(x_pre + tok * C + c * 32 + lg * 8, o);
    }
    const float mean = swap_sum(sum) * (1.0f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < KS * 8; ++i) {
        v[i] -= mean;
        q += v[i] * v[i];
    }
    const float rstd = rsqrtf(swap_sum(q) * (1.0f / (float)C) + eps);
    if (valid) {
#pragma unroll
        for (int c = 0; c < KS; ++c) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c * 32 + lg * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + c * 32 + lg * 8 + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(beta + c * 32 + lg * 8), t1 = *reinterpret_cast<const f32x4*>(beta + c * 32 + lg * 8 + 4);
            F o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = E::cv(v[c * 8 + e] * rstd * (e < 4 ? g0[e & 3] : g1[e & 3]) + (e < 4 ? t0[e & 3] : t1[e & 3]));
            E::st(y + tok * C + c * 32 + lg * 8, o);
        }
        if (mean_out && lg == 0) {
            mean_out[tok] = mean;
            rstd_out[tok] = rstd;
        }
    }
}
