// The fused Mlp kernels of Swin stages 0 / 1 (mlp_fused.hip) restated over an ELEMENT-TYPE TRAIT, for parity -- the companion of
// wblock_ref.hip / wattn_bwd_ref.hip.
//
//   forward      y = res + rowscale * ( gelu(x' . W1^T + b1) . W2^T + b2 ),  x' = x or LayerNorm(x) formed on the operand fragments
//                (Swin_Transformer.py:14-30, :267-268); saves LN(x), row statistics, pre-activation, activation
//   backward     dh = rowscale * (dy . W2) * gelu'(h_pre),  dx = dh . W1, optionally with the backward of norm2 as the tile epilogue
//                (dx = LayerNorm'(.) + dy, d(gamma) / d(beta) as per-workgroup partial sums)
//
// Same decomposition and the same index arithmetic as the production kernels: a workgroup of 8 waves walks 256-token tiles, wave w
// owns tokens [32 w, 32 w + 32) for both products; product 1 is D1[hidden][token] with the hidden-row permutation chan_of<8> that makes
// a lane's two accumulator tiles the 8 consecutive hidden channels of its token -- the B fragment of product 2; product 2 accumulates
// D2[channel][token] with chan_of<4 NT2>; LayerNorm rows sit in the four lanes li + 16 g.  What the trait replaces: fragments of 8
// fp32, the 32-deep product as 8 x v_mfma_f32_16x16x4_f32, erf / its derivative instead of the packed polynomials; and the weights are read
// as fragments straight from memory instead of streaming through the DMA ring (how operands reach the CU is not part of the
// algorithm being checked).  Selected by the FMMT_GENERIC dtype flag (bf16) or by dtype FMMT_F32 on the fused entry points.
#include "gemm_common.h"
#include "mlp_args.h"

namespace {

typedef __attribute__((ext_vector_type(8))) float f32x8;

template <typename T> struct MrEl;
template <> struct MrEl<bf16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ frag ld(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ void st(bf16* p, const frag& f) { *reinterpret_cast<bf16x8*>(p) = f; }
    static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ bf16 cv(float v) { return (bf16)v; }
};
template <> struct MrEl<float> {
    typedef f32x8 frag;
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

// ---------------------------------------------------------------------------------------------------------------- forward
template <typename T, int C, bool LN>
__global__ __launch_bounds__(512) void mlp_ref_fwd_kernel(MlpArgs p) {
    using E = MrEl<T>;
    using F = typename E::frag;
    constexpr int H = 4 * C, HS = 64, NS = H / HS, KS = C / 32, NT2 = C / 16, CW2 = 4 * NT2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const T* xg = reinterpret_cast<const T*>(p.x);
    const T* w1 = reinterpret_cast<const T*>(p.w1);
    const T* w2 = reinterpret_cast<const T*>(p.w2);
    T* hpre = reinterpret_cast<T*>(p.h_pre);
    T* hact = reinterpret_cast<T*>(p.h_act);
    T* xng = reinterpret_cast<T*>(p.xn);

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int t0 = tile * 256 + wave * 32;
        F xf[2][KS];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int tok = min(t0 + mt * 16 + li, p.M - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[mt][ks] = E::ld(xg + (size_t)tok * C + ks * 32 + lg * 8);
        }
        if constexpr (LN) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int tok = t0 + mt * 16 + li;
                float v[KS * 8];
                float sum = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[ks * 8 + e] = (float)xf[mt][ks][e];
                        sum += v[ks * 8 + e];
                    }
                const float mean = swap_sum(sum) * (1.0f / (float)C);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < KS * 8; ++e) {
                    v[e] -= mean;
                    q += v[e] * v[e];
                }
                const float rstd = rsqrtf(swap_sum(q) * (1.0f / (float)C) + p.eps);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    F o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = E::cv(v[ks * 8 + e] * rstd * p.ln_g[ks * 32 + lg * 8 + e] + p.ln_b[ks * 32 + lg * 8 + e]);
                    xf[mt][ks] = o;
                    if (xng && tok < p.M) E::st(xng + (size_t)tok * C + ks * 32 + lg * 8, o);
                }
                if (p.mean && tok < p.M && lg == 0) {
                    p.mean[tok] = mean;
                    p.rstd[tok] = rstd;
                }
            }
        }
        f32x4 acc2[2][NT2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int hs = 0; hs < NS; ++hs) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int h0 = hs * HS + blk * 32;                         // the 32 hidden channels of this K block of product 2
                f32x4 acc1[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const F wf = E::ld(w1 + (size_t)(h0 + chan_of<8>(nt, li >> 2, li & 3)) * C + ks * 32 + lg * 8);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc1[mt][nt] = E::mma(wf, xf[mt][ks], acc1[mt][nt]);
                    }
                F hf[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc1[mt][0][r] + p.b1[h0 + lg * 8 + r];
                        v[4 + r] = acc1[mt][1][r] + p.b1[h0 + lg * 8 + 4 + r];
                    }
                    const int tok = t0 + mt * 16 + li;
                    F pre8;
                    if (p.dg) {                              // FMMT_SAVE_DG: the derivative in place of the pre-activation (production: mlp_fused.hip, DG)
                        float d[8];
                        gelu_both_inplace<T>(v, d, 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) pre8[e] = E::cv(d[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) pre8[e] = E::cv(v[e]);
                        gelu_inplace<T>(v, 8);               // fp32: erff; bf16: the production kernels' form (fmmt_common.h)
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) hf[mt][e] = E::cv(v[e]);
                    if (tok < p.M) {
                        const size_t off = (size_t)tok * H + h0 + lg * 8;
                        if (hpre) E::st(hpre + off, pre8);
                        if (hact) E::st(hact + off, hf[mt]);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) {
                    const F wf2 = E::ld(w2 + (size_t)chan_of<CW2>(nt, li >> 2, li & 3) * H + h0 + lg * 8);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc2[mt][nt] = E::mma(wf2, hf[mt], acc2[mt][nt]);
                }
            }
        }
        LinArgs e{};
        e.M = p.M;
        e.N = C;
        e.bias = p.b2;
        e.y = p.y;
        e.ldy = C;
        e.res = LN ? (const void*)p.x : (const void*)p.res;
        e.ldres = C;
        e.rowscale = p.rowscale;
        e.rows_per_scale = p.rows_per_scale;
        nt_epilogue<T, 2, NT2>(e, acc2, t0, 0, li, lg);
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward (input)
// p.x = dy, p.w1 = W2^T [4C][C], p.w2 = W1^T [C][4C], p.h_pre = saved pre-activation, p.h_act = dh (out), p.y = dx (out)
template <typename T, int C, bool LNB>
__global__ __launch_bounds__(512) void mlp_ref_bwd_kernel(MlpArgs p) {
    using E = MrEl<T>;
    using F = typename E::frag;
    constexpr int H = 4 * C, HS = 64, NS = H / HS, KS = C / 32, NT2 = C / 16, CW2 = 4 * NT2;
    constexpr int NV = KS * 8;
    __shared__ float colsum_s[8][2 * C];                     // LNB: per-wave d(gamma) | d(beta) column sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const T* dyg = reinterpret_cast<const T*>(p.x);
    const T* w2t = reinterpret_cast<const T*>(p.w1);
    const T* w1t = reinterpret_cast<const T*>(p.w2);
    const T* hpre = reinterpret_cast<const T*>(p.h_pre);
    T* dhg = reinterpret_cast<T*>(p.h_act);
    T* dxg = reinterpret_cast<T*>(p.y);
    const T* lnx = reinterpret_cast<const T*>(p.ln_x);
    float dg[NV], db[NV];                                    // LNB: this lane's column sums (its 8 KS channels), over its tokens
#pragma unroll
    for (int v = 0; v < NV; ++v) dg[v] = db[v] = 0.f;

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int t0 = tile * 256 + wave * 32;
        F xf[2][KS];
        float rsv[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int tok = min(t0 + mt * 16 + li, p.M - 1);
            rsv[mt] = row_scale(p.rowscale, tok, p.rows_per_scale);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[mt][ks] = E::ld(dyg + (size_t)tok * C + ks * 32 + lg * 8);
        }
        f32x4 acc2[2][NT2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int hs = 0; hs < NS; ++hs) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int h0 = hs * HS + blk * 32;
                f32x4 acc1[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const F wf = E::ld(w2t + (size_t)(h0 + chan_of<8>(nt, li >> 2, li & 3)) * C + ks * 32 + lg * 8);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc1[mt][nt] = E::mma(wf, xf[mt][ks], acc1[mt][nt]);
                    }
                F hf[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int tokc = min(t0 + mt * 16 + li, p.M - 1), tok = t0 + mt * 16 + li;
                    const F ax = E::ld(hpre + (size_t)tokc * H + h0 + lg * 8);
                    float a[8], pre[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        a[e] = e < 4 ? acc1[mt][0][e] : acc1[mt][1][e - 4];
                        pre[e] = (float)ax[e];
                    }
                    if (p.dg) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] *= pre[e];
                    } else {
                        gelu_grad_mul_inplace<T>(a, pre, 8); // fp32: erff / expf; bf16: the production kernels' form
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) hf[mt][e] = E::cv(a[e] * rsv[mt]);
                    if (tok < p.M) E::st(dhg + (size_t)tok * H + h0 + lg * 8, hf[mt]);
                }
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) {
                    const F wf2 = E::ld(w1t + (size_t)chan_of<CW2>(nt, li >> 2, li & 3) * H + h0 + lg * 8);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc2[mt][nt] = E::mma(wf2, hf[mt], acc2[mt][nt]);
                }
            }
        }
        if constexpr (LNB) {
            // product 2's accumulator tile is d(LN out) of token li in the LayerNorm input's fragment layout (channels c * 32 + lg * 8 + e)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int tok = t0 + mt * 16 + li;
                const bool valid = tok < p.M;
                const int tokc = valid ? tok : p.M - 1;
                const float mean = p.mean[tokc], rstd = p.rstd[tokc];
                F lx[KS];
#pragma unroll
                for (int c = 0; c < KS; ++c) lx[c] = E::ld(lnx + (size_t)tokc * C + c * 32 + lg * 8);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int c = 0; c < KS; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = acc2[mt][2 * c + (e >> 2)][e & 3];
                        const float xh = ((float)lx[c][e] - mean) * rstd;
                        const float gm = d * p.ln_g[c * 32 + lg * 8 + e];
                        s1 += gm;
                        s2 += gm * xh;
                    }
                s1 = swap_sum(s1) * (1.0f / (float)C);
                s2 = swap_sum(s2) * (1.0f / (float)C);
#pragma unroll
                for (int c = 0; c < KS; ++c) {
                    F o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = acc2[mt][2 * c + (e >> 2)][e & 3];
                        const float xh = ((float)lx[c][e] - mean) * rstd;
                        const float gm = d * p.ln_g[c * 32 + lg * 8 + e];
                        o[e] = E::cv(rstd * (gm - s1 - xh * s2) + (float)xf[mt][c][e]);
                        if (valid) {
                            dg[c * 8 + e] += d * xh;
                            db[c * 8 + e] += d;
                        }
                    }
                    if (valid) E::st(dxg + (size_t)tok * C + c * 32 + lg * 8, o);
                }
            }
        } else {
            LinArgs e{};
            e.M = p.M;
            e.N = C;
            e.y = p.y;
            e.ldy = C;
            nt_epilogue<T, 2, NT2>(e, acc2, t0, 0, li, lg);
        }
    }
    if constexpr (LNB) {
        // column sums: over the 16 token lanes of a lane group (fixed order), then over the waves; one row of partials per workgroup
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float a = dg[v], b = db[v];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                a += __shfl_xor(a, m, 64);
                b += __shfl_xor(b, m, 64);
            }
            if (li == 0) {
                const int ch = (v >> 3) * 32 + lg * 8 + (v & 7);
                colsum_s[wave][ch] = a;
                colsum_s[wave][C + ch] = b;
            }
        }
        __syncthreads();
        if (tid < 2 * C) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += colsum_s[w][tid];
            p.ln_part[(size_t)blockIdx.x * 2 * C + tid] = a;
        }
    }
}

template <typename T, int C, bool LN>
int mr_launch_fwd(const MlpArgs& a, hipStream_t st) {
    const int grid = a.tiles < 256 ? a.tiles : 256;
    hipLaunchKernelGGL((mlp_ref_fwd_kernel<T, C, LN>), dim3(grid), dim3(512), 0, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}
template <typename T, int C, bool LNB>
int mr_launch_bwd(const MlpArgs& a, hipStream_t st) {
    const int grid = a.tiles < 256 ? a.tiles : 256;
    hipLaunchKernelGGL((mlp_ref_bwd_kernel<T, C, LNB>), dim3(grid), dim3(512), 0, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// el: FMMT_F32 / FMMT_BF16; a.tiles = ceil(M / 256); returns the grid size used through *grid (the LN-backward's partial rows)
int fmmt_mlp_ref_fwd_launch(int el, int C, bool ln, const MlpArgs& a, hipStream_t st) {
    if (el == FMMT_F32) {
        if (C == 96) return ln ? mr_launch_fwd<float, 96, true>(a, st) : mr_launch_fwd<float, 96, false>(a, st);
        return ln ? mr_launch_fwd<float, 192, true>(a, st) : mr_launch_fwd<float, 192, false>(a, st);
    }
    if (C == 96) return ln ? mr_launch_fwd<bf16, 96, true>(a, st) : mr_launch_fwd<bf16, 96, false>(a, st);
    return ln ? mr_launch_fwd<bf16, 192, true>(a, st) : mr_launch_fwd<bf16, 192, false>(a, st);
}
int fmmt_mlp_ref_bwd_launch(int el, int C, bool lnb, const MlpArgs& a, hipStream_t st) {
    if (el == FMMT_F32) {
        if (C == 96) return lnb ? mr_launch_bwd<float, 96, true>(a, st) : mr_launch_bwd<float, 96, false>(a, st);
        return lnb ? mr_launch_bwd<float, 192, true>(a, st) : mr_launch_bwd<float, 192, false>(a, st);
    }
    if (C == 96) return lnb ? mr_launch_bwd<bf16, 96, true>(a, st) : mr_launch_bwd<bf16, 96, false>(a, st);
    return lnb ? mr_launch_bwd<bf16, 192, true>(a, st) : mr_launch_bwd<bf16, 192, false>(a, st);
}
