// Pieces of the Linear kernels shared by gemm.hip and mlp_fused.hip: the argument block, the output-channel permutation
// of a wave tile and the vectorised epilogue (bias, GELU / GELU', DropPath row scale, residual, 16-byte row stores).
#pragma once
#include "fmmt_common.h"
#include "../../include/fmmt.h"

namespace {

struct LinArgs {
    int M, N, K;
    const void* x; int ldx;
    const void* w; int ldw;
    const float* bias;
    void* y; int ldy;
    void* y_pre;
    int epi;
    const void* aux; int ldaux;
    const void* res; int ldres;
    const float* rowscale; int rows_per_scale;
    int tiles_n, tiles_m;
    int reserved;      // launcher-to-kernel A/B bits: 4 direct p256 epilogue, 8 no GELU slab, 16 no block slab, 32 no wave slab
    int ksplit;        // K range per blockIdx.y (split-K); 0 = no split
    float* part;       // split-K: fp32 partials [split][M][N] instead of the epilogue
    // Segmented weight operand (fmmt_linear_fwd_seg3; few-token direct-to-LDS kernels only): wseg_mode 1 = along N (output channels
    // [s * wseg, (s + 1) * wseg) come from weight s = w / w1 / w2, bias / bias1 / bias2), 2 = along K (w_s is [N][wseg], the product sums over
    // the three K ranges of x); 0 = one weight.  A tile (64 channels) / a K step (64) never straddles a segment: wseg % 64 == 0.
    const void* w1; const void* w2;
    const float* bias1; const float* bias2;
    int wseg, wseg_mode;
};

// Output-channel permutation of a wave tile.  MFMA row i = 4*g + r of n-tile nt becomes output channel
//   full 8-wide chunk c = (4*nt)/8 :  c*32 + g*8 + (4*nt)%8 + r     (4 lanes x 16 B = 64 contiguous bytes per row)
//   4-wide tail (CW % 8 != 0)      :  (CW/8)*32 + g*4 + r
// so that every epilogue access of a lane is a 16-byte vector and the four lanes that share a token row
// cover one contiguous 64-byte span per store instruction.
template <int CW>
__device__ __forceinline__ int chan_of(int nt, int g, int r) {
    const int t0 = nt * 4;
    return (t0 < (CW / 8) * 8) ? (t0 / 8) * 32 + g * 8 + (t0 % 8) + r : (CW / 8) * 32 + g * 4 + r;
}

// The M x N epilogue operand of a wave tile (residual OR GELU' pre-activation; bf16) and the DropPath scale of its rows, loaded
// into registers ahead of the epilogue: the persistent kernel issues these loads at the head of a tile's LAST K step, in front
// of that step's DMA, so that the in-order vmcnt has them back when the accumulators are ready (an epilogue that loads its
// operand itself stalls every wave of the workgroup for a full memory latency per tile -- a third of a K = 384 tile's life).
template <int MT, int NT>
struct EpiPre {
    bf16x8 full[MT][NT / 2 > 0 ? NT / 2 : 1];
    bf16x4 tail[MT];
    float rs[MT];
};
template <int MT, int NT>
__device__ __forceinline__ void nt_epilogue_prefetch(const LinArgs& p, EpiPre<MT, NT>& pre, int mbase, int nbase, int li, int lg) {
    const bool use_aux = p.epi == FMMT_EPI_GELU_BWD || p.epi == FMMT_EPI_MUL_AUX;
    const bf16* __restrict__ src = reinterpret_cast<const bf16*>(use_aux ? p.aux : p.res);
    const int ld = use_aux ? p.ldaux : p.ldres;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int m = min(mbase + a * 16 + li, p.M - 1);     // ragged last panel: a valid row, never stored
        pre.rs[a] = row_scale(p.rowscale, m, p.rows_per_scale);
        if (src) {
#pragma unroll
            for (int c = 0; c < NT / 2; ++c)
                pre.full[a][c] = *reinterpret_cast<const bf16x8*>(src + (size_t)m * ld + nbase + chan_of<4 * NT>(2 * c, lg, 0));
            if constexpr (NT % 2) pre.tail[a] = *reinterpret_cast<const bf16x4*>(src + (size_t)m * ld + nbase + chan_of<4 * NT>(NT - 1, lg, 0));
        }
    }
}

// Epilogue of one wave tile (MT x NT MFMA tiles at rows mbase.., channels nbase..): bias, GELU / GELU',
// DropPath row scale, residual, 16-byte stores; or raw fp32 partials for the split-K path.
// NOOPS: the launch is known to carry no residual, no GELU' operand and no DropPath scale (the persistent kernel without
// operand prefetch): their loads are compiled out -- a global load that only EXISTS in the epilogue makes the compiler guard
// every later reuse of its destination register with s_waitcnt vmcnt(0), also in the K loop's DMA issue.
template <typename T, int MT, int NT, bool BIAS_DONE = false, bool PRE = false, bool NOOPS = false>
__device__ __forceinline__ void nt_epilogue(const LinArgs& p, f32x4 (&acc)[MT][NT], int mbase, int nbase, int li, int lg,
                                            const EpiPre<MT, NT>* pre = nullptr) {
    constexpr int VEC = Vec<T>::N;
    T* __restrict__ yg = reinterpret_cast<T*>(p.y);
    T* __restrict__ ypre = reinterpret_cast<T*>(p.y_pre);
    const T* __restrict__ auxg = reinterpret_cast<const T*>(p.aux);
    const T* __restrict__ resg = reinterpret_cast<const T*>(p.res);
        if (p.part) {                                      // split-K: raw fp32 partial sums, finished by another kernel
            float* pp = p.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int m = mbase + a * 16 + li;
                if (m >= p.M) continue;
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int n = nbase + chan_of<4 * NT>(b, lg, 0);
                    if (n + 4 <= p.N) *reinterpret_cast<f32x4*>(pp + (size_t)m * p.N + n) = acc[a][b];
                }
            }
            return;
        }
        // per lane and token row: CW/VEC vector chunks (fp32: one n-tile = 4 channels = 16 B; bf16: two
        // n-tiles = 8 channels = 16 B) plus, for bf16 with odd NT, a 4-channel (8-byte) tail
        constexpr int TPC = VEC / 4;                       // n-tiles per 16-byte chunk
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int m = mbase + a * 16 + li;
            if (m >= p.M) continue;
            float rs;
            if constexpr (NOOPS) rs = 1.0f;
            else if constexpr (PRE) rs = pre->rs[a];
            else rs = row_scale(p.rowscale, m, p.rows_per_scale);
#pragma unroll
            for (int b0 = 0; b0 < NT; b0 += TPC) {
                const int w = (NT - b0 >= TPC) ? VEC : 4;      // chunk width (compile-time after unrolling)
                const int n = nbase + chan_of<4 * NT>(b0, lg, 0);
                if (n + w > p.N) continue;
                float v[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = (e < w) ? acc[a][b0 + (e >> 2)][e & 3] : 0.f;
                if (!BIAS_DONE && p.bias) {
#pragma unroll
                    for (int e4 = 0; e4 < VEC; e4 += 4)
                        if (e4 < w) {
                            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + n + e4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[e4 + r] += bb[r];
                        }
                }
                auto load_chunk = [&](const T* base, int ld, float* out) {
                    if constexpr (PRE) {                     // bf16 only: the operand is already in registers
                        if (w == VEC) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) out[e] = (float)pre->full[a][b0 / 2][e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) out[e] = (float)pre->tail[a][e];
                        }
                        return;
                    }
                    if (w == VEC) {
                        const Vec<T> t = ldvec<T>(base + (size_t)m * ld + n);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) out[e] = t.get(e);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) out[e] = to_f32(base[(size_t)m * ld + n + e]);
                    }
                };
                auto store_chunk = [&](T* base, int ld, const float* in) {
                    if (w == VEC) {
                        Vec<T> t;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) t.set(e, in[e]);
                        stvec<T>(base + (size_t)m * ld + n, t);
                    } else {
                        T o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(in[e]);
                        *reinterpret_cast<uint2*>(base + (size_t)m * ld + n) = *reinterpret_cast<const uint2*>(o);   // bf16 only (VEC == 8)
                    }
                };
                if (p.epi == FMMT_EPI_GELU) {
                    if (ypre) store_chunk(ypre, p.ldy, v);
                    gelu_inplace<T>(v, VEC);
                } else if (p.epi == FMMT_EPI_GELU_DG) {         // y = gelu(v), y_pre = gelu'(v): the backward multiplies (FMMT_EPI_MUL_AUX), no polynomial there
                    float d[VEC];
                    gelu_both_inplace<T>(v, d, VEC);
                    if (ypre) store_chunk(ypre, p.ldy, d);
                } else if (!NOOPS && p.epi == FMMT_EPI_GELU_BWD) {
                    float ax[VEC];
                    load_chunk(auxg, p.ldaux, ax);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) ax[e] = (e < w) ? ax[e] : 0.f;
                    gelu_grad_mul_inplace<T>(v, ax, VEC);
                } else if (!NOOPS && p.epi == FMMT_EPI_MUL_AUX) {
                    float ax[VEC];
                    load_chunk(auxg, p.ldaux, ax);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] *= (e < w) ? ax[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] *= rs;
                if (!NOOPS && resg) {
                    float rx[VEC];
                    load_chunk(resg, p.ldres, rx);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += (e < w) ? rx[e] : 0.f;
                }
                store_chunk(yg, p.ldy, v);
            }
        }
}

// Plain / bias epilogue of a 256-row block tile through an fp32 LDS slab (the K loop's stages are free by then).  Straight from the
// accumulator layout a store touches 16 token rows x 64 bytes, which the CU's vector-memory path takes at ~7 B per cycle; the
// stage-0/1 launches, whose K loop is three to six steps, spend much of their life there.  Here the waves park their
// accumulators, 128 token rows per pass, and all threads then walk the slab row by row, 8 channels (16 bytes of output) per
// thread, so that every store instruction writes complete 128-byte lines.  Same fp32 values, same rounding as nt_epilogue.
// (With a residual / GELU' operand or a GELU the same scheme measured 0-14 % SLOWER: two co-resident workgroups already hide one
// another's epilogue, and the slab's four barriers outweigh what the operand loads gain; those launches keep nt_epilogue.)
// MT x NT MFMA tiles per wave, waves (256 / (16 MT)) x NWN, block tile 256 x BN, NTHREADS threads.
template <typename T, int MT, int NT, int NWN, int BN, int NTHREADS>
__device__ __forceinline__ void nt_epilogue_slab(const LinArgs& p, f32x4 (&acc)[MT][NT], char* lds, int wm, int wn, int li, int lg, int tid,
                                                 int m0, int n0) {
    static_assert(sizeof(T) == 2, "bf16 path");
    constexpr int CW = 4 * NT, WN = BN / NWN, RPW = MT * 16, GPP = 128 / RPW, PITCH = BN * 4 + 16, CPR = BN / 8;
    constexpr int PER = 128 * CPR / NTHREADS;
    static_assert(128 % RPW == 0 && (128 * CPR) % NTHREADS == 0, "slab geometry");
    T* __restrict__ yg = reinterpret_cast<T*>(p.y);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // the K loop's last fragment reads / the previous pass's slab reads are done
        if (wm / GPP == pass) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    *reinterpret_cast<f32x4*>(lds + ((wm % GPP) * RPW + a * 16 + li) * PITCH + (wn * WN + chan_of<CW>(b, lg, 0)) * 4) = acc[a][b];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int q = i * NTHREADS + tid, r = q / CPR, cc = q - r * CPR;
            const int m = m0 + pass * 128 + r, n = n0 + cc * 8;
            if (m < p.M) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(lds + r * PITCH + cc * 32);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(lds + r * PITCH + cc * 32 + 16);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                if (p.bias) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                }
                Vec<T> t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t.set(e, v[e]);
                stvec<T>(yg + (size_t)m * p.ldy + n, t);
            }
        }
    }
}

// Every epilogue (GELU + pre-activation, GELU', DropPath scale, residual) of a 64 x 64 WAVE tile through a wave-private fp32
// slab, one 16-row fragment at a time: 64 channels of bf16 are exactly one 128-byte line, so after the transposition every store
// AND every load of the residual / GELU' operand is eight complete lines per instruction instead of sixteen 64-byte pieces.
// No barrier (a wave's LDS operations execute in order), so two co-resident workgroups keep hiding one another's epilogue.
// Same fp32 arithmetic in the same order as nt_epilogue.  wslab: this wave's 16 x 272 bytes.
template <typename T>
__device__ __forceinline__ void nt_epilogue_wslab(const LinArgs& p, f32x4 (&acc)[4][4], char* wslab, int lane, int li, int lg, int mbase, int nbase) {
    static_assert(sizeof(T) == 2, "bf16 path");
    constexpr int PITCH = 64 * 4 + 16;
    T* __restrict__ yg = reinterpret_cast<T*>(p.y);
    T* __restrict__ ypre = reinterpret_cast<T*>(p.y_pre);
    const T* __restrict__ auxg = reinterpret_cast<const T*>(p.aux);
    const T* __restrict__ resg = reinterpret_cast<const T*>(p.res);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(wslab + li * PITCH + chan_of<16>(b, lg, 0) * 4) = acc[a][b];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * 64 + lane, r = q >> 3, cc = q & 7;
            const int m = mbase + a * 16 + r, n = nbase + cc * 8;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(wslab + r * PITCH + cc * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(wslab + r * PITCH + cc * 32 + 16);
            if (m < p.M) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                if (p.bias) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                }
                const float rs = row_scale(p.rowscale, m, p.rows_per_scale);
                auto put = [&](T* base, int ld) {
                    Vec<T> t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t.set(e, v[e]);
                    stvec<T>(base + (size_t)m * ld + n, t);
                };
                if (p.epi == FMMT_EPI_GELU) {
                    if (ypre) put(ypre, p.ldy);
                    gelu_inplace<T>(v, 8);
                } else if (p.epi == FMMT_EPI_GELU_DG) {
                    float d[8];
                    gelu_both_inplace<T>(v, d, 8);
                    if (ypre) {
                        Vec<T> t;
#pragma unroll
                        for (int e = 0; e < 8; ++e) t.set(e, d[e]);
                        stvec<T>(ypre + (size_t)m * p.ldy + n, t);
                    }
                } else if (p.epi == FMMT_EPI_GELU_BWD) {
                    const Vec<T> t = ldvec<T>(auxg + (size_t)m * p.ldaux + n);
                    float ax[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) ax[e] = t.get(e);
                    gelu_grad_mul_inplace<T>(v, ax, 8);
                } else if (p.epi == FMMT_EPI_MUL_AUX) {
                    const Vec<T> t = ldvec<T>(auxg + (size_t)m * p.ldaux + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= t.get(e);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= rs;
                if (resg) {
                    const Vec<T> t = ldvec<T>(resg + (size_t)m * p.ldres + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += t.get(e);
                }
                put(yg, p.ldy);
            }
        }
    }
}

}  // namespace
