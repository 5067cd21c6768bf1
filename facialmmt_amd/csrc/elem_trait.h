// Element-type trait of the parity instantiations (lin_lnbwd.hip, patch_ln.hip): what a kernel restated over `T` needs from its element
// type -- the 8-element operand fragment of one lane, its 16-byte loads / stores, the 32-deep 16x16 product, the output conversion.
//   bf16   fragment = 8 bf16 (one VGPR quad), product = one v_mfma_f32_16x16x32_bf16
//   float  fragment = 8 fp32, product = 8 x v_mfma_f32_16x16x4_f32 (element e of every lane's fragment is one 4-deep step: lane (li, lg)
//          holds k = lg * 8 + e, so the 8 steps cover the same 32 k as the bf16 instruction, in the same accumulator layout)
// mlp_ref.hip / wblock_ref.hip / wattn_bwd_ref.hip carry the same trait locally (round 4).
#pragma once
#include "gemm_common.h"

typedef __attribute__((ext_vector_type(8))) float f32x8;

template <typename T> struct ElemTrait;
template <> struct ElemTrait<bf16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ frag ld(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ void st(bf16* p, const frag& f) { *reinterpret_cast<bf16x8*>(p) = f; }
    static __device__ __forceinline__ frag zero() {
        frag z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
        return z;
    }
    static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ bf16 cv(float v) { return (bf16)v; }
};
template <> struct ElemTrait<float> {
    typedef f32x8 frag;
    static __device__ __forceinline__ frag ld(const float* p) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        return frag{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    }
    static __device__ __forceinline__ void st(float* p, const frag& f) {
        *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
    static __device__ __forceinline__ frag zero() { return frag{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, f32x4 c) {
#pragma unroll
        for (int e = 0; e < 8; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ float cv(float v) { return v; }
};
