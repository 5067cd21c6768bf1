// Device-side helpers shared by every kernel of libfmmt_hip (gfx950 / CDNA4 only).
//
// Conventions used throughout csrc/:
//   * a wavefront is 64 lanes; kernels hard-code 64.
//   * activations are token-major row-major matrices [rows][channels]; element type T is
//     float (parity mode) or __bf16 (throughput mode); every reduction / accumulator is fp32.
//   * MFMA operand maps (cdna_hip_programming.md section 3):
//       16x16x32 bf16 : A[i = lane&15][k = (lane>>4)*8 + e], B[k = (lane>>4)*8 + e][j = lane&15], e = 0..7
//       16x16x4  f32  : A[i = lane&15][k = lane>>4],         B[k = lane>>4][j = lane&15]
//       C/D (both)    : D[i = (lane>>4)*4 + r][j = lane&15], r = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gelu_poly_data.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define FMMT_DT_F32 0
#define FMMT_DT_BF16 1

// Dispatch constants.  Each of these was an environment switch while its A/B was open (rounds 1-2: tile shapes, ring depths, epilogue
// routes, thresholds ...); the measured winner is compiled in and the name stays as the label DESIGN.md refers to.  The library
// reads NO environment variable.
constexpr int fmmt_const(const char*, int winner) { return winner; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: one flag word per call site, one bit per device (a process
// that drives several GPUs -- a C-ABI consumer without torch's one-process-per-GPU habit -- sets it on each), atomically (launchers may be
// called from autograd's worker threads).  Returns 0 or the hipError.
struct FmmtLdsOnce {
    unsigned long long done = 0;
    int set(const void* fn, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        const unsigned long long bit = 1ull << (dev & 63);
        if (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit) return 0;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
        return 0;
    }
};

#define FMMT_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// 16-byte vector access.  Vec<T>::N elements per 16 B.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec<bf16> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

template <typename T> __device__ __forceinline__ Vec<T> ldvec(const T* p) {
    Vec<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void stvec(T* p, const Vec<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}
template <typename T> __device__ __forceinline__ Vec<T> zerovec() {
    Vec<T> r;
#pragma unroll
    for (int i = 0; i < Vec<T>::N; ++i) r.set(i, 0.f);
    return r;
}

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// ---------------------------------------------------------------------------------------------
// exact (erf) GELU and its derivative -- nn.GELU() / F.gelu default
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Throughput (bf16) mode.  History.  Round 1: libm erff, ~32 VALU instructions per element.  Round 2: Abramowitz-Stegun 7.1.26 with v_rcp + v_exp (quarter rate),
// ~20 issue slots per element: 57 % of the fused Mlp's cycles.  Round 3: a (value, slope) table in LDS, 6 VALU + 1 ds_read_b64 per element -- each lookup a
// dependent LDS round trip (the fused Mlp forward's waves spent 45 % of their cycles parked on s_waitcnt: profiles/r03_pmc_mlp_fused_fwd_sq1.md).  Round 4: Phi and
// gelu' as odd polynomials around 1/2 in packed fp32 math (8 / 10 terms, argument clamped to +-4 / +-4.5; tools/gen_gelu_poly.py still prints them):
// |error| <= 5e-5 ABSOLUTE -- i.e. no accuracy at all where the functions are smaller than that: gelu(x) for x < -3.5 (|gelu| < 8e-4) came out with tens of
// percent of relative error and either sign (round-4 ADVICE, round-5 VERDICT).
typedef __attribute__((ext_vector_type(2))) float f32x2;
// Round 6: the forms the bf16 kernels evaluate (gelu_poly_data.h, tools/gen_gelu_poly.py).  With a = max(-|x|, -R):
//     gelu(x)  = max(x, 0) + a 2^L(a),          L ~ log2 Phi on [-9, 0], degree 6        (gelu(x) = x + gelu(-x) for x > 0; the product is <= 0 always)
//     gelu'(x) = x < 0 ? g(a) : 1 - g(a),       g = 2^(c a^2) S(a), S degree 8            (gelu'(x) = 1 - gelu'(-x); the Gaussian carries the decay)
// relative error <= 4.6e-4 (gelu, x in [-8, 0)) / 7.1e-4 (gelu', [-8, -1.5]) and the exact sign on the negative tail, 6.8e-5 / 6.0e-5 absolute
// everywhere; per element 1 v_max + 6 / 8 v_fmaak + ONE v_exp_f32 + 3 / 6 plain VALU -- no packed fp32 (profiles/r05_issue_rates.txt: a v_pk_fma_f32
// beside MFMAs costs ~12 cycles, a plain VALU instruction hides).  tests/support_op_cases.py::t_gelu_tail sweeps [-8, 8] through every kernel family.
// Measured against the round-4 forms (round 6, same call, profiles/r06_gelu_forms.txt): GELU epilogues and the fused Mlp forward unchanged (they are bound by
// their two output streams), GELU' launches +4 % -- which is why the forward now STORES gelu' (gelu_both_exp_f below) and no bench kernel evaluates it alone.
template <int N> __device__ __forceinline__ float fmmt_horner(const float (&c)[N], float a) {
    float q = c[N - 1];
#pragma unroll
    for (int k = N - 2; k >= 0; --k) q = __builtin_fmaf(q, a, c[k]);
    return q;
}
__device__ __forceinline__ float gelu_exp_f(float x) {
    const float a = fmaxf(-fabsf(x), -FMMT_GELU_L_R);
    const float t = a * __builtin_amdgcn_exp2f(fmmt_horner(fmmt_gelu_log2phi_poly, a));
    return fmaxf(x, 0.f) + t;
}
__device__ __forceinline__ float gelu_grad_exp_f(float x) {
    const float a = fmaxf(-fabsf(x), -FMMT_GELU_S_R);
    const float e = __builtin_amdgcn_exp2f((a * FMMT_GELU_S_C) * a);
    const float g = e * fmmt_horner(fmmt_gelu_grad_s_poly, a);
    return x < 0.f ? g : 1.0f - g;                          // (1/2 + sign(x) (1/2 - g) would cancel the tail away: g itself for x < 0)
}
// gelu AND gelu' of one value (the forward epilogues that store the derivative for the backward, FMMT_EPI_GELU_DG): Phi(a) = 2^L(a) serves both,
// phi(a) = 2^(c a^2 + log2(1 / sqrt(2 pi))) is one more exponential -- gelu'(a) = Phi(a) + a phi(a); 16 plain VALU + 2 v_exp_f32 for the pair
// (separately 11 + 1 and 15 + 1).  Relative error of gelu' on the tail: Phi's 4.6e-4 scaled by Phi / |gelu'| ~ 1 / a^2.
__device__ __forceinline__ void gelu_both_exp_f(float x, float& g, float& dg) {
    const float a = fmaxf(-fabsf(x), -FMMT_GELU_L_R);
    const float cdf = __builtin_amdgcn_exp2f(fmmt_horner(fmmt_gelu_log2phi_poly, a));
    const float pdf = __builtin_amdgcn_exp2f(__builtin_fmaf(a * FMMT_GELU_S_C, a, -1.3257480647361592f));     // log2(1 / sqrt(2 pi))
    g = fmaxf(x, 0.f) + a * cdf;
    const float ga = __builtin_fmaf(a, pdf, cdf);
    dg = x < 0.f ? ga : 1.0f - ga;
}
// v[e] <- gelu(v[e]) / v[e] <- v[e] * gelu'(pre[e]), n = 2 NP elements
template <int NP> __device__ __forceinline__ void gelu_poly_inplace(float* v) {
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j) v[j] = gelu_exp_f(v[j]);
}
template <int NP> __device__ __forceinline__ void gelu_grad_poly_mul_inplace(float* v, const float* pre) {
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j) v[j] *= gelu_grad_exp_f(pre[j]);
}

// element-type dispatch used by the GEMM epilogues: exact for float, packed polynomial for bf16 (n = 4 or a multiple of 8)
// IL = pairs evaluated in lockstep (4; 2 where the registers are short)
template <typename T, int IL = 4> __device__ __forceinline__ void gelu_inplace(float* v, int n) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < n; ++e) v[e] = gelu_f(v[e]);
    } else {
        if (n == 4) return gelu_poly_inplace<2>(v);
#pragma unroll
        for (int e = 0; e < n; e += 2 * IL) gelu_poly_inplace<IL>(v + e);
    }
}
// v[e] <- gelu(v[e]), d[e] <- gelu'(v[e]) (before the update): exact for float, the shared-exponential form for bf16
template <typename T> __device__ __forceinline__ void gelu_both_inplace(float* v, float* d, int n) {
#pragma unroll
    for (int e = 0; e < n; ++e) {
        const float x = v[e];
        if constexpr (sizeof(T) == 4) {
            v[e] = gelu_f(x);
            d[e] = gelu_grad_f(x);
        } else {
            gelu_both_exp_f(x, v[e], d[e]);
        }
    }
}
template <typename T, int IL = 4> __device__ __forceinline__ void gelu_grad_mul_inplace(float* v, const float* pre, int n) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < n; ++e) v[e] *= gelu_grad_f(pre[e]);
    } else {
        if (n == 4) return gelu_grad_poly_mul_inplace<2>(v, pre);
#pragma unroll
        for (int e = 0; e < n; e += 2 * IL) gelu_grad_poly_mul_inplace<IL>(v + e, pre + e);
    }
}

// ---------------------------------------------------------------------------------------------
// reductions inside a 16-lane group (DPP row) and a full 64-lane wave
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = group16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Sum / max over the four lanes li + 16 g (the lanes that share an MFMA column / a token row) through the gfx950 row / half swaps (v_permlane16_swap exchanges the odd
// 16-lane rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower
// half of the second): VALU only, no LDS round trip (ds_bpermute) on the softmax's dependency chain.
__device__ __forceinline__ float swap_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    float a = __uint_as_float(r[0]), b = __uint_as_float(r[1]), m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    a = __uint_as_float(q[0]);
    b = __uint_as_float(q[1]);
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ float swap_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// ---------------------------------------------------------------------------------------------
// keep-mask of the attention-probability dropout (fmmt_mha_fwd / _bwd; both formulations, replayed bit-identically in the backward).
// splitmix64 per element was ~60 issue slots of 64-bit multiplies against ~10 for the softmax element itself (round 6: the 4 x 512-token text
// shape ran 39 us without dropout, 75 with).  One pair of 32-bit avalanche mixes (two multiplies each; lowbias32 and the first rounds of
// triple32, public-domain constants of the hash-prospector search) now serves the FOUR consecutive keys 4g .. 4g+3 of a query row, 16 bits per
// decision: keep = field >= round(p * 2^16), kept values scaled by 2^16 / (2^16 - threshold) (the exact inverse of the realised keep rate).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32a(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t mix32b(uint32_t x) {
    x ^= x >> 17; x *= 0xed5ad4bbU; x ^= x >> 11; x *= 0xac4c1b51U; x ^= x >> 15;
    return x;
}
struct AttnDrop {
    uint32_t key, thresh, G;      // hashed seed; threshold of a 16-bit field; 4-key groups per query row
    float inv;
};
__device__ __forceinline__ AttnDrop attn_drop_setup(float p, uint64_t seed, int Lk) {
    AttnDrop d;
    d.key = mix32a((uint32_t)seed ^ 0x85ebca6bU) + mix32b((uint32_t)(seed >> 32) ^ 0xc2b2ae35U);
    const uint32_t t = (uint32_t)(p * 65536.0f + 0.5f);
    d.thresh = t > 65535u ? 65535u : t;
    d.inv = 65536.0f / (65536.0f - (float)d.thresh);
    d.G = (uint32_t)(Lk + 3) >> 2;
    return d;
}
// the two words of group `g` of query row `row` (row = (batch * heads + head) * Lq + query; 32-bit wrap-around is harmless)
__device__ __forceinline__ void attn_drop_words(const AttnDrop& d, uint32_t row, uint32_t g, uint32_t& a, uint32_t& b) {
    const uint32_t x = row * d.G + g + d.key;
    a = mix32a(x);
    b = mix32b(x);
}
// multiplier of key 4g + f from the group's words: fields 0, 1 = low / high half of a, fields 2, 3 of b
__device__ __forceinline__ float attn_drop_field(const AttnDrop& d, uint32_t a, uint32_t b, int f) {
    const uint32_t w = (f & 2) ? b : a;
    const uint32_t v = (f & 1) ? (w >> 16) : (w & 0xFFFFu);
    return v >= d.thresh ? d.inv : 0.f;
}

// The same generator for the elementwise dropout of the fused sublayer tails (plm_fused.hip): elements 4g .. 4g+3 of call site `salt` share a pair of words.
struct ElemDrop {
    uint32_t key, thresh;
    float inv;
};
__device__ __forceinline__ ElemDrop elem_drop_setup(float p, uint64_t seed, uint64_t salt) {
    ElemDrop d;
    d.key = mix32a((uint32_t)seed ^ 0x85ebca6bU) + mix32b((uint32_t)(seed >> 32) ^ 0xc2b2ae35U) + mix32a((uint32_t)salt ^ 0x27d4eb2fU) + mix32b((uint32_t)(salt >> 32) ^ 0x165667b1U);
    const uint32_t t = (uint32_t)(p * 65536.0f + 0.5f);
    d.thresh = t > 65535u ? 65535u : t;
    d.inv = 65536.0f / (65536.0f - (float)d.thresh);
    return d;
}
// keep flags of elements 4g .. 4g+3 (bit f of the result: element 4g + f is kept)
__device__ __forceinline__ uint32_t elem_keep4(const ElemDrop& d, uint32_t g) {
    const uint32_t x = g + d.key, a = mix32a(x), b = mix32b(x);
    return ((a & 0xFFFFu) >= d.thresh ? 1u : 0u) | ((a >> 16) >= d.thresh ? 2u : 0u) | ((b & 0xFFFFu) >= d.thresh ? 4u : 0u) | ((b >> 16) >= d.thresh ? 8u : 0u);
}

// per-sample (DropPath) multiplier lookup: rowscale == nullptr -> 1
__device__ __forceinline__ float row_scale(const float* rowscale, int row, int rows_per_scale) {
    return rowscale ? rowscale[row / rows_per_scale] : 1.0f;
}
