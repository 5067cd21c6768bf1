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
#include "gelu_lut_data.h"

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

// Throughput (bf16) mode: the same erf GELU evaluated two elements at a time with packed fp32 math.
//   erfc(|u|) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-u^2),  t = 1 / (1 + 0.3275911 |u|)   (Abramowitz-Stegun
//   7.1.26, |error| <= 1.5e-7), u = x / sqrt(2);  Phi(x) = 1 - erfc/2 for x >= 0 and erfc/2 for x < 0 (no
//   cancellation in the negative tail).  exp(-u^2) = exp(-x^2/2) is also the Gaussian of GELU', so the
//   derivative costs one extra FMA.  |gelu error| <= 5e-7 absolute and <= 2e-3 relative in the far negative tail
//   -- both below bf16 rounding of the stored result; the fp32 (parity) kernels keep erff.
// The libm erff used above costs ~32 VALU instructions per element and made the fc1 / fc2-dgrad epilogues
// VALU-bound (64 outputs per lane per tile against 48..96 MFMAs); this form is ~8 issue slots per element.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& cdf, f32x2& gauss) {
    const f32x2 u = x * 0.70710678118654752f;
    const f32x2 au = {__builtin_fabsf(u.x), __builtin_fabsf(u.y)};
    const f32x2 d = au * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 e2 = u * u * -1.4426950408889634f;
    gauss = f32x2{__builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y)};
    f32x2 q = t * 1.061405429f + -1.453152027f;
    q = q * t + 1.421413741f;
    q = q * t + -0.284496736f;
    q = q * t + 0.254829592f;
    const f32x2 h = q * t * gauss * 0.5f;              // erfc(|u|) / 2
    cdf = f32x2{u.x >= 0.f ? 1.0f - h.x : h.x, u.y >= 0.f ? 1.0f - h.y : h.y};
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    f32x2 cdf, g;
    gelu_parts2(x, cdf, g);
    return x * cdf;
}
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
    f32x2 cdf, g;
    gelu_parts2(x, cdf, g);
    return x * g * 0.3989422804014327f + cdf;
}
// Table form for the bf16 kernels whose epilogue is VALU-bound (fused Mlp: 32 activations per lane and 64-channel stage against 36 MFMAs;
// GELU / GELU' epilogues of the GEMMs): the polynomial above is ~20 issue slots per element with its two quarter-rate transcendentals
// (57 % of the fused Mlp's cycles).  Phi(x) -- or gelu'(x) = Phi(x) + x phi(x) -- tabulated on [-8, 8) in steps of 1/64 as
// (value, difference to the next entry) pairs (gelu_lut_data.h, generated in double precision by tools/gen_gelu_lut.py), copied into
// LDS (8 KB) by the kernel, linear interpolation: fma, med3, cvt, fract, address, ds_read_b64, fma = 6 VALU + 1 LDS read.
// Interpolation error h^2/8 |f''|: <= 8e-6 absolute for Phi (relative <= 2e-3 out to x = -8: the curvature decays with the function),
// <= 3e-5 for gelu' -- below the bf16 rounding of what is stored.  The fp32 (parity) kernels keep erff.
constexpr int GELU_LUT_N = 1024;
constexpr int GELU_LUT_BYTES = GELU_LUT_N * 8;
typedef __attribute__((ext_vector_type(2))) float lut2_t;
// copy a table into LDS: 512 chunks of 16 bytes; the caller makes it visible (barrier) before the first gelu_lut()
__device__ __forceinline__ void gelu_lut_copy(lut2_t* tab, const float* src, int tid, int nthreads) {
    for (int i = tid; i < GELU_LUT_BYTES / 16; i += nthreads)
        reinterpret_cast<f32x4*>(tab)[i] = reinterpret_cast<const f32x4*>(src)[i];
}
__device__ __forceinline__ float gelu_lut(const lut2_t* tab, float x) {
    float t = __builtin_fmaf(x, 64.0f, (float)(GELU_LUT_N / 2));
    t = __builtin_amdgcn_fmed3f(t, 0.0f, (float)GELU_LUT_N - 0.001f);
    const lut2_t e = tab[(int)t];
    return __builtin_fmaf(__builtin_amdgcn_fractf(t), e.y, e.x);
}
// v[e] <- gelu(v[e]) / v[e] <- v[e] * gelu'(pre[e]) through a table in LDS (nullptr: the polynomial)
template <typename T> __device__ __forceinline__ void gelu_inplace_lut(const lut2_t* tab, float* v, int n);
template <typename T> __device__ __forceinline__ void gelu_grad_mul_inplace_lut(const lut2_t* tab, float* v, const float* pre, int n);

// element-type dispatch used by the GEMM epilogue: exact for float, packed-fast for bf16
template <typename T> __device__ __forceinline__ void gelu_inplace(float* v, int n) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < n; ++e) v[e] = gelu_f(v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < n; e += 2) {
            const f32x2 r = gelu_fast2(f32x2{v[e], v[e + 1]});
            v[e] = r.x;
            v[e + 1] = r.y;
        }
    }
}
template <typename T> __device__ __forceinline__ void gelu_grad_mul_inplace(float* v, const float* pre, int n) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < n; ++e) v[e] *= gelu_grad_f(pre[e]);
    } else {
#pragma unroll
        for (int e = 0; e < n; e += 2) {
            const f32x2 r = gelu_grad_fast2(f32x2{pre[e], pre[e + 1]});
            v[e] *= r.x;
            v[e + 1] *= r.y;
        }
    }
}
template <typename T> __device__ __forceinline__ void gelu_inplace_lut(const lut2_t* tab, float* v, int n) {
    if (sizeof(T) == 4 || !tab) return gelu_inplace<T>(v, n);
#pragma unroll
    for (int e = 0; e < n; ++e) v[e] *= gelu_lut(tab, v[e]);
}
template <typename T> __device__ __forceinline__ void gelu_grad_mul_inplace_lut(const lut2_t* tab, float* v, const float* pre, int n) {
    if (sizeof(T) == 4 || !tab) return gelu_grad_mul_inplace<T>(v, pre, n);
#pragma unroll
    for (int e = 0; e < n; ++e) v[e] *= gelu_lut(tab, pre[e]);
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
// counter-based RNG for attention-probability dropout (replayed bit-identically in backward):
// splitmix64 of (seed, element index) -> 24-bit uniform
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// per-sample (DropPath) multiplier lookup: rowscale == nullptr -> 1
__device__ __forceinline__ float row_scale(const float* rowscale, int row, int rows_per_scale) {
    return rowscale ? rowscale[row / rows_per_scale] : 1.0f;
}
