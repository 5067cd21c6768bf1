// The fused attention half of a Swin block (wblock.hip) restated over an ELEMENT-TYPE TRAIT, for parity.
//
//     y = x + rowscale * ( proj( W-MSA / SW-MSA( LN1(x) . Wqkv^T + bqkv ) ) + bproj )        Swin_Transformer.py:233-266, :113-144
//
// wblock.hip's kernel is built on the fragment layouts of the bf16 16x16x32 matrix instruction: a lane's 16 bytes are 8 consecutive
// channels of one token, two accumulator tiles ARE the next product's operand fragment, and so on.  Round 3 had no fp32 form of it, so
// the reference-generated goldens reached it only through bf16 rounding (3e-2 of scale) -- its index arithmetic, mask derivation and
// softmax were never shown to be right to 1e-3.  This file is the same kernel, statement for statement, with the fragment type and
// the matrix product behind a trait:
//
//   bf16   fragment = 8 x bf16 (16 bytes), product = v_mfma_f32_16x16x32_bf16              -- the production arithmetic
//   float  fragment = 8 x fp32 (32 bytes), product = 8 x v_mfma_f32_16x16x4_f32, element e of both fragments per instruction:
//          the 16x16x4 instruction takes A[i = lane & 15][k = lane >> 4] and B[k = lane >> 4][j = lane & 15] and returns the SAME
//          accumulator layout D[4 (lane >> 4) + r][lane & 15]; fed element e of a lane's 8-element fragment it contracts the k-slots
//          {8 g + e : g = 0..3}, and e = 0..7 covers the 32-deep block -- every layout argument of wblock.hip carries over unchanged.
//
// The fp32 instantiation keeps no weights in LDS (they would take 146 KB) and reads its weight fragments from memory through the same
// fragment-order row map (wb_src_row); nothing is rounded to bf16 anywhere.  It is a parity instrument, not a fast kernel (it spills),
// reachable through fmmt_window_block_fwd(dtype = FMMT_F32).  Tests (tests/test_gpu_wblock.py): the bf16 instantiation of THIS
// template against the production kernel (same values up to the packed-vs-scalar LayerNorm arithmetic), and the fp32 instantiation
// against the reference's block_s0_shift{0,3} goldens and an fp64 restatement at 1e-3.
#include "wblock_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) float f32x8;

template <typename T> struct WbEl;
template <> struct WbEl<bf16> {
    typedef bf16x8 frag;
    static constexpr bool LDS_WEIGHTS = true;
    static __device__ __forceinline__ frag ld(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ void st(bf16* p, const frag& f) { *reinterpret_cast<bf16x8*>(p) = f; }
    static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ bf16 cv(float v) { return (bf16)v; }
};
template <> struct WbEl<float> {
    typedef f32x8 frag;
    static constexpr bool LDS_WEIGHTS = false;
    static __device__ __forceinline__ frag ld(const float* p) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        return frag{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    }
    static __device__ __forceinline__ void st(float* p, const frag& f) {
        *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
    static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, f32x4 c) {
#pragma unroll
        for (int e = 0; e < 8; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ float cv(float v) { return v; }
};

template <typename T, int C, int NW>
__global__ __launch_bounds__(NW * 64) void wblock_ref_fwd_kernel(WbArgs p) {
    using E = WbEl<T>;
    using F = typename E::frag;
    using L = WbLds<C>;
    constexpr int NH = L::NH, KS = C / 32, PITCH = L::PITCH, BP = WB_BP;
    constexpr int WB = E::LDS_WEIGHTS ? L::W_BYTES : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Wl = reinterpret_cast<T*>(smem);
    float* Bs = reinterpret_cast<float*>(smem + WB);
    float* Gm = Bs + NH * TOK * BP;
    float* Bt = Gm + C;
    float* Bq = Bt + C;
    float* Bp = Bq + 3 * C;
    const T* xg = reinterpret_cast<const T*>(p.x);
    const T* wqkv = reinterpret_cast<const T*>(p.wqkv);
    const T* wproj = reinterpret_cast<const T*>(p.wproj);
    T* yg = reinterpret_cast<T*>(p.y);
    T* xng = reinterpret_cast<T*>(p.xn);
    T* og = reinterpret_cast<T*>(p.o);

    // ---------------------------------------------------------------- stage the operands every window shares
    if constexpr (E::LDS_WEIGHTS) {
        for (int q = threadIdx.x; q < 4 * C * (C / 8); q += NW * 64) {
            const int d = q / (C / 8), ch = q - d * (C / 8);
            bool is_proj;
            const int sr = wb_src_row<C>(d, is_proj);
            E::st(Wl + d * PITCH + ch * 8, E::ld((is_proj ? wproj : wqkv) + (size_t)sr * C + ch * 8));
        }
    }
    for (int t = threadIdx.x; t < NH * TOK * BP; t += NW * 64) {
        const int h = t / (TOK * BP), r = t - h * (TOK * BP), q = r / BP, k = r - q * BP;
        Bs[t] = k < TOK ? p.table[p.index[q * TOK + k] * NH + h] * WB_LOG2E : NEG_BIG;
    }
    for (int t = threadIdx.x; t < C; t += NW * 64) {
        Gm[t] = p.ln_g[t];
        Bt[t] = p.ln_b[t];
        Bp[t] = p.bproj ? p.bproj[t] : 0.f;
    }
    for (int t = threadIdx.x; t < 3 * C; t += NW * 64) Bq[t] = p.bqkv ? p.bqkv[t] : 0.f;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const LaneGeom G = lane_geom(li, lg, p.shift);
    const int nwt = gridDim.x * NW, B_ = p.B_;
    const float invC = 1.0f / (float)C;
    const float sc2 = p.scale * WB_LOG2E;
    F ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = E::cv(1.0f);
    const int cperm[2] = {(li >> 2) * 8 + (li & 3), (li >> 2) * 8 + 4 + (li & 3)};

    // weight fragment: row d of the fragment-order image (wblock.hip's LDS image), channels col .. col + 7
    auto wfrag = [&](int d, int col) -> F {
        if constexpr (E::LDS_WEIGHTS) {
            return E::ld(Wl + d * PITCH + col);
        } else {
            bool is_proj;
            const int sr = wb_src_row<C>(d, is_proj);
            return E::ld((is_proj ? wproj : wqkv) + (size_t)sr * C + col);
        }
    };

    for (int b_ = blockIdx.x * NW + wave; b_ < B_; b_ += nwt) {
        const WinPos P = win_pos(p, b_);
        int tok[4];
        F nrm[4][KS];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            tok[t] = (int)tok_of(p, P, G.di[t], G.dj[t]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) nrm[t][ks] = E::ld(xg + (size_t)tok[t] * C + lg * 8 + ks * 32);
        }
        const float rs = p.rowscale ? p.rowscale[P.img] : 1.0f;
        const bool masked = p.shift > 0 && (P.lastrow || P.lastcol);

        // ------------------------------------------------------------ LayerNorm (Swin_Transformer.py:239,243): even / odd partial sums as in
        // the production kernel's packed arithmetic
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v[KS * 8];
            float sx = 0.f, sy = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[ks * 8 + 2 * j] = (float)nrm[t][ks][2 * j];
                    v[ks * 8 + 2 * j + 1] = (float)nrm[t][ks][2 * j + 1];
                    sx += v[ks * 8 + 2 * j];
                    sy += v[ks * 8 + 2 * j + 1];
                }
            const float mean = swap_sum(sx + sy) * invC;
            float qx = 0.f, qy = 0.f;
#pragma unroll
            for (int e = 0; e < KS * 8; e += 2) {
                v[e] -= mean;
                v[e + 1] -= mean;
                qx += v[e] * v[e];
                qy += v[e + 1] * v[e + 1];
            }
            const float rstd = rsqrtf(swap_sum(qx + qy) * invC + p.eps);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                F o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = E::cv(v[ks * 8 + e] * rstd * Gm[ks * 32 + lg * 8 + e] + Bt[ks * 32 + lg * 8 + e]);
                nrm[t][ks] = o;
                if (xng && G.valid[t]) E::st(xng + (size_t)tok[t] * C + ks * 32 + lg * 8, o);
            }
            if (p.mean && G.valid[t] && lg == 0) {
                p.mean[tok[t]] = mean;
                p.rstd[tok[t]] = rstd;
            }
        }

        F of[NH][4];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            // -------------------------------------------------------- q^T, k^T of head h: [32 channels][64 tokens]
            F qk[2][4];
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                f32x4 acc[2][4];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[nt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const F wf = wfrag(((h * 3 + part) * 2 + nt) * 16 + li, ks * 32 + lg * 8);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[nt][t] = E::mma(wf, nrm[t][ks], acc[nt][t]);
                    }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    F f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f[r] = E::cv(acc[0][t][r] + Bq[part * C + h * 32 + lg * 8 + r]);
                        f[4 + r] = E::cv(acc[1][t][r] + Bq[part * C + h * 32 + lg * 8 + 4 + r]);
                    }
                    qk[part][t] = f;
                }
            }
            // -------------------------------------------------------- v of head h, transposed product: [64 tokens][32 channels]
            F vT[2][2];
            {
                f32x4 acc[4][2];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) acc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const F wf = wfrag(((h * 3 + 2) * 2 + dt) * 16 + li, ks * 32 + lg * 8);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t][dt] = E::mma(nrm[t][ks], wf, acc[t][dt]);
                    }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const float bv = Bq[2 * C + h * 32 + cperm[dt]];
#pragma unroll
                    for (int ks2 = 0; ks2 < 2; ++ks2) {
                        F f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = E::cv(acc[2 * ks2][dt][r] + bv);
                            f[4 + r] = E::cv(acc[2 * ks2 + 1][dt][r] + bv);
                        }
                        vT[ks2][dt] = f;
                    }
                }
            }
            // -------------------------------------------------------- attention core of head h (S^T = K . Q^T, base-2 softmax, O^T = V^T . P^T)
            const float* bh = Bs + h * TOK * BP;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const int q = qt * 16 + li;
                const float* brow = bh + (q < TOK ? q : TOK - 1) * BP + lg * 4;
                f32x4 s[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4 a = E::mma(qk[1][kt], qk[0][qt], f32x4{0.f, 0.f, 0.f, 0.f});
                    s[kt] = a * sc2 + *reinterpret_cast<const f32x4*>(brow + kt * 16);
                }
                if (masked) {
                    const unsigned mb = std_mask_bits(G, P, qt);
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[kt][r] += ((mb >> (kt * 4 + r)) & 1u) ? -100.0f * WB_LOG2E : 0.0f;
                }
                float m = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
                for (int kt = 1; kt < 4; ++kt) m = fmaxf(m, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
                m = swap_max(m);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - m);
                F pb0, pb1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pb0[r] = E::cv(s[0][r]); pb0[4 + r] = E::cv(s[1][r]);
                    pb1[r] = E::cv(s[2][r]); pb1[4 + r] = E::cv(s[3][r]);
                }
                f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f}, ls = {0.f, 0.f, 0.f, 0.f};
                o0 = E::mma(vT[0][0], pb0, o0);
                o1 = E::mma(vT[0][1], pb0, o1);
                ls = E::mma(ones, pb0, ls);
                o0 = E::mma(vT[1][0], pb1, o0);
                o1 = E::mma(vT[1][1], pb1, o1);
                ls = E::mma(ones, pb1, ls);
                const float l = ls[0];
                const float inv = sizeof(T) == 2 ? __builtin_amdgcn_rcpf(l) : 1.0f / l;
                o0 *= inv;
                o1 *= inv;
                F ob;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ob[r] = E::cv(o0[r]);
                    ob[4 + r] = E::cv(o1[r]);
                }
                of[h][qt] = ob;
                if (G.valid[qt]) {
                    if (og) E::st(og + (size_t)tok[qt] * C + h * HD + lg * 8, ob);
                    if (lg == 0) p.lse[((size_t)b_ * NH + h) * TOK + q] = (m + (sizeof(T) == 2 ? __builtin_amdgcn_logf(l) : log2f(l))) * WB_LN2;
                }
            }
        }

        // ------------------------------------------------------------ proj + bias, DropPath scale, residual (Swin_Transformer.py:142,266)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc[2 * KS];
#pragma unroll
            for (int cn = 0; cn < 2 * KS; ++cn) acc[cn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int cn = 0; cn < 2 * KS; ++cn) acc[cn] = E::mma(wfrag(3 * C + cn * 16 + li, h * 32 + lg * 8), of[h][t], acc[cn]);
#pragma unroll
            for (int c = 0; c < KS; ++c) {
                const F xr = E::ld(xg + (size_t)tok[t] * C + c * 32 + lg * 8);
                F o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = E::cv((acc[2 * c][r] + Bp[c * 32 + lg * 8 + r]) * rs + (float)xr[r]);
                    o[4 + r] = E::cv((acc[2 * c + 1][r] + Bp[c * 32 + lg * 8 + 4 + r]) * rs + (float)xr[4 + r]);
                }
                if (G.valid[t]) E::st(yg + (size_t)tok[t] * C + c * 32 + lg * 8, o);
            }
        }
    }
}

template <typename T, int C, int NW>
int wb_ref_launch(const WbArgs& a, hipStream_t st) {
    constexpr int lds = (WbEl<T>::LDS_WEIGHTS ? WbLds<C>::W_BYTES : 0) + WbLds<C>::BIAS_BYTES + WbLds<C>::VEC_BYTES;
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&wblock_ref_fwd_kernel<T, C, NW>), lds)) return rc_;
    const int need = (a.B_ + NW - 1) / NW;
    const int grid = need < 256 ? need : 256;
    hipLaunchKernelGGL((wblock_ref_fwd_kernel<T, C, NW>), dim3(grid), dim3(NW * 64), lds, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// dtype FMMT_F32: fp32 operands (parity); FMMT_BF16: the bf16 instantiation of the SAME template (tests compare it with wblock.hip's kernel)
int fmmt_wblock_ref_fwd_launch(int dtype, const void* args, hipStream_t st) {
    const WbArgs& a = *reinterpret_cast<const WbArgs*>(args);
    if (dtype == FMMT_F32) return wb_ref_launch<float, 96, 4>(a, st);
    return wb_ref_launch<bf16, 96, 4>(a, st);
}
