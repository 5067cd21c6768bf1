// Input gradient of "LayerNorm -> Linear" in ONE launch (gfx950, bf16; C = 96 LayerNorm channels, K = 3C = 288 Linear outputs):
//
//     dx = LayerNorm'( dz . W ; x, mean, rstd, gamma ) + dres          dz [M, K], W [K, C], x / dres / dx [M, C]
//
// the tail of the stage-0 attention half's backward (Swin_Transformer.py:239-243: attn(norm1(x)) -- dz = d(loss)/d(qkv), W = Wqkv, dres =
// the gradient of the block's residual stream).  As two launches (fmmt_linear_fwd on W^T, fmmt_layernorm_bwd) d(LN out) is written and
// read back (770 MB at 640 frames) and x / dres / dx move in a launch of their own: 0.31 + 0.27 ms per block against the 2.3 GB this kernel
// moves.  It is the "d(xn) = dqkv . Wqkv -> LayerNorm' in-wave -> write dx + dy residual once" part of the one-kernel backward the
// round-3 review asks for, as its own launch behind fmmt_window_block_attn_bwd (the rest of that fusion is priced in NOTES.md R4.3).
//
// Decomposition = product 2 and the LNB epilogue of mlp_fused_bwd_kernel (mlp_fused.hip) without the ring: W^T [C][K] (55 KB) is staged
// once per workgroup in 32-deep K blocks of 64-byte swizzled rows; a wave owns 32 tokens of a 256-token tile, its dz rows ARE the B
// fragments of the 16x16x32 MFMA (token li, 8 consecutive k per lane: 16-byte loads straight from memory), the accumulator tile is d(LN
// out) of those tokens with a token's 96 channels in the four lanes li + 16 g -- the layout of the x fragments, so LayerNorm' is in-lane
// sums + two cross-lane steps.  The dz fragments travel in three chunks through two register buffers, each requested one chunk (or, the
// next tile's first, one epilogue) ahead; x / statistics of a tile are requested in front of its MFMAs, dres per token tile at the head of its LayerNorm'.  d(gamma) / d(beta): per-lane
// accumulators over the persistent tile loop, summed over the row's 16 tokens by DPP, over the waves through LDS, one row of partial
// sums per workgroup, finished in fixed order by mlp_ln_part_reduce_kernel's twin below.
#include "gemm_common.h"
#include "elem_trait.h"

namespace {

struct DlArgs {
    int M, tiles;
    const bf16* dz;        // [M][K]            (the generic twin below reads these five as its own element type)
    const bf16* wt;        // [C][K] = W^T
    const bf16* x;         // [M][C] LayerNorm input
    const float* mean;
    const float* rstd;
    const float* gamma;
    const bf16* dres;      // [M][C] or nullptr
    bf16* dx;              // [M][C]
    float* part;           // [grid][2 C]
};

__device__ __forceinline__ float dl_row16_sum(float v) {     // sum over the 16 lanes of a DPP row, fixed order (as mlp_fused.hip)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

template <int C, int K, bool HASRES>
__global__ __launch_bounds__(512) void lin_lnbwd_kernel(DlArgs p) {
    using T = bf16;
    constexpr int KB = K / 32, KS = C / 32, NT2 = C / 16, CW2 = 4 * NT2;
    constexpr int NV = KS * 8, NOWN = 2 * NV / 16;
    static_assert(K % 32 == 0 && C % 32 == 0 && (2 * NV) % 16 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Wl = reinterpret_cast<T*>(smem);                                       // [KB][C][32], 64-byte rows, 16-byte chunks swizzled
    float* gam_s = reinterpret_cast<float*>(smem + (size_t)KB * C * 32 * sizeof(T));
    float* slot_s = gam_s + C;                                                // [8 waves][4 lg][2 NV]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    auto swz = [](int row) { return ((row >> 3) ^ (row >> 2)) & 3; };
    for (int q = tid; q < C * (K / 8); q += 512) {
        const int r = q / (K / 8), ch = q - r * (K / 8), b = ch >> 2, c4 = ch & 3;
        *reinterpret_cast<bf16x8*>(Wl + (b * C + r) * 32 + ((c4 ^ swz(r)) << 3)) = *reinterpret_cast<const bf16x8*>(p.wt + (size_t)r * K + ch * 8);
    }
    if (tid < C) gam_s[tid] = p.gamma[tid];
    __syncthreads();
    int woff[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
        const int r = chan_of<CW2>(nt, li >> 2, li & 3);
        woff[nt] = r * 32 + ((lg ^ swz(r)) << 3);
    }
    // d(gamma) / d(beta): this lane's tokens' contributions to its NV channels, summed over the whole tile loop in registers; the 16 lanes
    // of a DPP row (= 16 tokens, same channels) meet once, after the loop (per tile that reduction was ~1 k VALU instructions per wave:
    // most of the epilogue)
    float dga[NV], dba[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) dga[i] = dba[i] = 0.f;

    const int G = gridDim.x;
    // dz fragments in chunks of three K blocks (96 of the 288 columns), two register buffers: chunk i + 1 is requested in front of chunk
    // i's MFMAs, the next tile's first chunk in front of the epilogue (all 18 fragments of a tile at once, plus the next tile's, spilled)
    constexpr int CH = 3, NCH = KB / CH;
    static_assert(KB % CH == 0 && NCH == 3, "three chunks");
    bf16x8 qa[2][CH], qb[2][CH];
    auto load_chunk = [&](int tile, int ch, bf16x8 (&dst)[2][CH]) {
        const int t0 = tile * 256 + wave * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int tok = min(t0 + mt * 16 + li, p.M - 1);
#pragma unroll
            for (int k = 0; k < CH; ++k) dst[mt][k] = *reinterpret_cast<const bf16x8*>(p.dz + (size_t)tok * K + (ch * CH + k) * 32 + lg * 8);
        }
    };
    f32x4 acc[2][NT2];
    auto mma_chunk = [&](int ch, const bf16x8 (&src)[2][CH]) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wl + (ch * CH + k) * (C * 32) + woff[nt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, src[mt][k], acc[mt][nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);               // (without the fences the scheduler hoists all 54 weight-fragment reads and spills)
        }
    };
    int tile = blockIdx.x;
    if (tile < p.tiles) load_chunk(tile, 0, qa);
    for (; tile < p.tiles; tile += G) {
        const int t0 = tile * 256 + wave * 32;
        // this tile's LayerNorm operands: requested now, used behind the MFMAs
        bf16x8 lx[2][KS];
        float lmean[2], lrstd[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int tok = min(t0 + mt * 16 + li, p.M - 1);
            lmean[mt] = p.mean[tok];
            lrstd[mt] = p.rstd[tok];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                lx[mt][ks] = *reinterpret_cast<const bf16x8*>(p.x + (size_t)tok * C + ks * 32 + lg * 8);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        load_chunk(tile, 1, qb);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk(0, qa);
        load_chunk(tile, 2, qa);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk(1, qb);
        mma_chunk(2, qa);
        if (tile + G < p.tiles) load_chunk(tile + G, 0, qa);
        __builtin_amdgcn_sched_barrier(0);
        // LayerNorm' on the accumulator tile (the LNB epilogue of mlp_fused_bwd_kernel): one token tile at a time
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            // one token tile at a time, the row's values formed twice (sums, then outputs) rather than kept: registers
            __builtin_amdgcn_sched_barrier(0);
            const int tok = t0 + mt * 16 + li;
            const bool valid = tok < p.M;
            const float mean = lmean[mt], rstd = lrstd[mt];
            bf16x8 dr[KS];                                   // the residual gradient of this token tile: requested here, used behind the row sums
            if constexpr (HASRES) {
                const int tokc = min(tok, p.M - 1);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) dr[ks] = *reinterpret_cast<const bf16x8*>(p.dres + (size_t)tokc * C + ks * 32 + lg * 8);
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < KS; ++c) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam_s + c * 32 + lg * 8), g1 = *reinterpret_cast<const f32x4*>(gam_s + c * 32 + lg * 8 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = acc[mt][2 * c + (e >> 2)][e & 3];
                    const float xh = ((float)lx[mt][c][e] - mean) * rstd;
                    const float gm = d * (e < 4 ? g0[e & 3] : g1[e & 3]);
                    s1 += gm;
                    s2 += gm * xh;
                }
            }
            s1 = swap_sum(s1) * (1.0f / (float)C);
            s2 = swap_sum(s2) * (1.0f / (float)C);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < KS; ++c) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam_s + c * 32 + lg * 8), g1 = *reinterpret_cast<const f32x4*>(gam_s + c * 32 + lg * 8 + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int v = c * 8 + e;
                    const float d = acc[mt][2 * c + (e >> 2)][e & 3];
                    const float xh = ((float)lx[mt][c][e] - mean) * rstd;
                    const float gm = d * (e < 4 ? g0[e & 3] : g1[e & 3]);
                    o[e] = (bf16)(rstd * (gm - s1 - xh * s2) + (HASRES ? (float)dr[c][e] : 0.f));
                    const float dv = valid ? d : 0.f;
                    dga[v] += dv * xh;
                    dba[v] += dv;
                }
                int tk = tok;
                asm volatile("" : "+v"(tk));                 // the store address is formed here (hoisted, the six 64-bit addresses spilled)
                if (valid) *reinterpret_cast<bf16x8*>(p.dx + (size_t)tk * C + c * 32 + lg * 8) = o;
            }
        }
    }
    // d(gamma) | d(beta): per-wave slots -> fixed-order sum over the waves -> one row of partial sums per workgroup
    float own[NOWN];
#pragma unroll
    for (int i = 0; i < NOWN; ++i) own[i] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float sg = dl_row16_sum(dga[v]), sb = dl_row16_sum(dba[v]);
        if (li == (v & 15)) own[v >> 4] = sg;
        if (li == ((v + NV) & 15)) own[(v + NV) >> 4] = sb;
    }
#pragma unroll
    for (int k = 0; k < NOWN; ++k) slot_s[(wave * 4 + lg) * (2 * NV) + k * 16 + li] = own[k];
    __syncthreads();
    if (tid < 2 * C) {
        const int kind = tid / C, ch = tid % C;
        const int v = kind * NV + (ch >> 5) * 8 + (ch & 7), lgc = (ch >> 3) & 3;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) a += slot_s[(w * 4 + lgc) * (2 * NV) + v];
        p.part[(size_t)blockIdx.x * 2 * C + tid] = a;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same kernel restated over an ELEMENT-TYPE TRAIT, for parity (the companion of mlp_ref.hip / wblock_ref.hip / wattn_bwd_ref.hip):
// same decomposition and index arithmetic -- 256-token tiles, wave w owns tokens [32 w, 32 w + 32), dz rows as the B fragments, W^T rows
// chan_of<CW2> as the A fragments, the accumulator tile = d(LN out) with a token's C channels in the four lanes li + 16 g, LayerNorm' as
// in-lane sums + two cross-lane steps, d(gamma) / d(beta) as per-lane sums over the tile loop -> DPP row sum -> per-wave LDS slots ->
// one partial row per workgroup.  What the trait replaces: fragments of 8 fp32 and the 32-deep product as 8 x v_mfma_f32_16x16x4_f32;
// and the weight fragments are read straight from memory (how operands reach the CU is not part of the algorithm being checked).
// fp32: nothing is rounded -- held to the reference's block goldens at 1e-3.  bf16 | FMMT_GENERIC: held against the kernel above.
template <typename T, int C, int K, bool HASRES>
__global__ __launch_bounds__(512) void lin_lnbwd_ref_kernel(DlArgs p) {
    using E = ElemTrait<T>;
    using F = typename E::frag;
    constexpr int KB = K / 32, KS = C / 32, NT2 = C / 16, CW2 = 4 * NT2;
    constexpr int NV = KS * 8, NOWN = 2 * NV / 16;
    __shared__ float slot_s[8 * 4 * 2 * NV];
    const T* dzg = reinterpret_cast<const T*>(p.dz);
    const T* wtg = reinterpret_cast<const T*>(p.wt);
    const T* xg = reinterpret_cast<const T*>(p.x);
    const T* drg = reinterpret_cast<const T*>(p.dres);
    T* dxg = reinterpret_cast<T*>(p.dx);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    float dga[NV], dba[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) dga[i] = dba[i] = 0.f;
    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int t0 = tile * 256 + wave * 32;
        f32x4 acc[2][NT2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kb = 0; kb < KB; ++kb) {
            F zf[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) zf[mt] = E::ld(dzg + (size_t)min(t0 + mt * 16 + li, p.M - 1) * K + kb * 32 + lg * 8);
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                const F wf = E::ld(wtg + (size_t)chan_of<CW2>(nt, li >> 2, li & 3) * K + kb * 32 + lg * 8);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = E::mma(wf, zf[mt], acc[mt][nt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int tok = t0 + mt * 16 + li, tokc = min(tok, p.M - 1);
            const bool valid = tok < p.M;
            const float mean = p.mean[tokc], rstd = p.rstd[tokc];
            F lx[KS], dr[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                lx[ks] = E::ld(xg + (size_t)tokc * C + ks * 32 + lg * 8);
                if constexpr (HASRES) dr[ks] = E::ld(drg + (size_t)tokc * C + ks * 32 + lg * 8);
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < KS; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = acc[mt][2 * c + (e >> 2)][e & 3];
                    const float xh = ((float)lx[c][e] - mean) * rstd;
                    const float gm = d * p.gamma[c * 32 + lg * 8 + e];
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
                    const int v = c * 8 + e;
                    const float d = acc[mt][2 * c + (e >> 2)][e & 3];
                    const float xh = ((float)lx[c][e] - mean) * rstd;
                    const float gm = d * p.gamma[c * 32 + lg * 8 + e];
                    o[e] = E::cv(rstd * (gm - s1 - xh * s2) + (HASRES ? (float)dr[c][e] : 0.f));
                    const float dv = valid ? d : 0.f;
                    dga[v] += dv * xh;
                    dba[v] += dv;
                }
                if (valid) E::st(dxg + (size_t)tok * C + c * 32 + lg * 8, o);
            }
        }
    }
    float own[NOWN];
#pragma unroll
    for (int i = 0; i < NOWN; ++i) own[i] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float sg = dl_row16_sum(dga[v]), sb = dl_row16_sum(dba[v]);
        if (li == (v & 15)) own[v >> 4] = sg;
        if (li == ((v + NV) & 15)) own[(v + NV) >> 4] = sb;
    }
#pragma unroll
    for (int k = 0; k < NOWN; ++k) slot_s[(wave * 4 + lg) * (2 * NV) + k * 16 + li] = own[k];
    __syncthreads();
    if (tid < 2 * C) {
        const int kind = tid / C, ch = tid % C;
        const int v = kind * NV + (ch >> 5) * 8 + (ch & 7), lgc = (ch >> 3) & 3;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) a += slot_s[(w * 4 + lgc) * (2 * NV) + v];
        p.part[(size_t)blockIdx.x * 2 * C + tid] = a;
    }
}

// rows of per-workgroup partial sums [nblocks][2 C] -> d(gamma) [C], d(beta) [C]; fixed order
__global__ __launch_bounds__(1024) void lin_ln_part_reduce_kernel(const float* __restrict__ part, int nblocks, int C, float* dgamma, float* dbeta) {
    __shared__ float red[16][64];
    const int tc = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tc;
    float a = 0.f;
    if (i < 2 * C)
        for (int b = tg; b < nblocks; b += 16) a += part[(size_t)b * 2 * C + i];
    red[tg][tc] = a;
    __syncthreads();
    if (tg == 0 && i < 2 * C) {
        a = red[0][tc];
#pragma unroll
        for (int g = 1; g < 16; ++g) a += red[g][tc];
        if (i < C) dgamma[i] = a;
        else dbeta[i - C] = a;
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" size_t fmmt_linear_ln_bwd_workspace(int C) { return (size_t)256 * 2 * C * sizeof(float); }

extern "C" int fmmt_linear_ln_bwd(int dtype, int M, int C, int K, const void* dz, const void* wt, const void* x, const float* mean, const float* rstd,
                                  const float* ln_gamma, const void* dres, void* dx, float* dgamma, float* dbeta, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    const int el = dtype & 0xff;
    const bool generic = (dtype & FMMT_GENERIC) || el == FMMT_F32;                          // parity instantiations (lin_lnbwd_ref_kernel)
    if ((el != FMMT_BF16 && el != FMMT_F32) || M <= 0 || C != 96 || K != 288) return FMMT_EINVAL;   // other shapes: fmmt_linear_fwd + fmmt_layernorm_bwd
    if (!dz || !wt || !x || !mean || !rstd || !ln_gamma || !dx || !dgamma || !dbeta || !workspace) return FMMT_EINVAL;
    if (workspace_bytes < fmmt_linear_ln_bwd_workspace(C)) return FMMT_EWORKSPACE;
    if (!al16(dz) || !al16(wt) || !al16(x) || !al16(dx) || (dres && !al16(dres)) || !al16(workspace)) return FMMT_EALIGN;
    DlArgs a{};
    a.M = M; a.tiles = (M + 255) / 256;
    a.dz = (const bf16*)dz; a.wt = (const bf16*)wt; a.x = (const bf16*)x; a.mean = mean; a.rstd = rstd; a.gamma = ln_gamma;
    a.dres = (const bf16*)dres; a.dx = (bf16*)dx; a.part = (float*)workspace;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    constexpr int lds = (288 / 32) * 96 * 32 * 2 + 96 * 4 + 8 * 4 * 2 * 24 * 4;
    const int grid = a.tiles < 256 ? a.tiles : 256;
    if (generic) {
        if (el == FMMT_F32) {
            if (dres) hipLaunchKernelGGL((lin_lnbwd_ref_kernel<float, 96, 288, true>), dim3(grid), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((lin_lnbwd_ref_kernel<float, 96, 288, false>), dim3(grid), dim3(512), 0, st, a);
        } else {
            if (dres) hipLaunchKernelGGL((lin_lnbwd_ref_kernel<bf16, 96, 288, true>), dim3(grid), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((lin_lnbwd_ref_kernel<bf16, 96, 288, false>), dim3(grid), dim3(512), 0, st, a);
        }
    } else if (dres) {
        static FmmtLdsOnce lds_once;
        if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&lin_lnbwd_kernel<96, 288, true>), lds)) return rc_;
        hipLaunchKernelGGL((lin_lnbwd_kernel<96, 288, true>), dim3(grid), dim3(512), lds, st, a);
    } else {
        static FmmtLdsOnce lds_once;
        if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&lin_lnbwd_kernel<96, 288, false>), lds)) return rc_;
        hipLaunchKernelGGL((lin_lnbwd_kernel<96, 288, false>), dim3(grid), dim3(512), lds, st, a);
    }
    FMMT_CHECK_LAUNCH();
    hipLaunchKernelGGL(lin_ln_part_reduce_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, st, (const float*)workspace, grid, C, dgamma, dbeta);
    FMMT_CHECK_LAUNCH();
    return 0;
}
