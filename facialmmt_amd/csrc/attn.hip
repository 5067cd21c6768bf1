// Attention cores for gfx950 (round-1 formulation: one query row -- or one key row -- per lane).
//
// Window attention: a 7x7 window has 49 tokens and head_dim 32, so a whole (window, head) problem fits
// one 64-lane wave with lane p owning token p: scores, softmax and the P.V row are lane-local (no
// cross-lane reduction at all), K/V rows are broadcast from LDS.  roll / window_partition /
// window_reverse are pure address arithmetic: slot (i,j) of window (wy,wx) is pixel
// ((wy*7+i+shift) % H, (wx*7+j+shift) % W) for loads of q,k,v AND for the store of the output.
// A workgroup (4 waves) is pinned to one head and walks windows, so the dense (49x49) relative
// position bias of that head is gathered once into LDS.
//
// Backward recomputes probabilities from the saved log-sum-exp (flash style) in two passes:
// row-owner pass (dq, d bias) and column-owner pass (dk, dv); d(bias table) is reduced through
// per-wave partials in a fixed order (deterministic, no atomics).
#include "fmmt_common.h"
#include "../../include/fmmt.h"
#include "wattn_args.h"
#include "mha_args.h"
#include <stdlib.h>

namespace {

constexpr int TOK = 49;      // tokens per window
constexpr int WS = 7;        // window side
constexpr int HD = 32;       // head dim of every Swin-tiny stage
constexpr int KP = 36;       // LDS row pitch (floats) of a 49 x 32 tile: 16-byte aligned rows
constexpr int BP = 50;       // LDS row pitch of the 49 x 49 bias tile
constexpr int WA_BWD_WAVES_PER_HEAD_MAX = 256;   // partial d(bias) tiles per head (waves for the fp32 kernel, workgroups for the MFMA kernel)


template <typename T> __device__ __forceinline__ void load_row32(const T* p, float* r) {
    constexpr int VEC = Vec<T>::N;
#pragma unroll
    for (int c = 0; c < HD / VEC; ++c) {
        const Vec<T> v = ldvec<T>(p + c * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) r[c * VEC + e] = v.get(e);
    }
}
template <typename T> __device__ __forceinline__ void store_row32(T* p, const float* r) {
    constexpr int VEC = Vec<T>::N;
#pragma unroll
    for (int c = 0; c < HD / VEC; ++c) {
        Vec<T> v;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v.set(e, r[c * VEC + e]);
        stvec<T>(p + c * VEC, v);
    }
}
__device__ __forceinline__ void lds_put_row32(float* s, const float* r) {
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) *reinterpret_cast<f32x4*>(s + c * 4) = f32x4{r[c * 4], r[c * 4 + 1], r[c * 4 + 2], r[c * 4 + 3]};
}
__device__ __forceinline__ float lds_dot32(const float* s, const float* r) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + c * 4);
        a += v[0] * r[c * 4] + v[1] * r[c * 4 + 1] + v[2] * r[c * 4 + 2] + v[3] * r[c * 4 + 3];
    }
    return a;
}
__device__ __forceinline__ void lds_axpy32(float* acc, float a, const float* s) {
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + c * 4);
        acc[c * 4] += a * v[0];
        acc[c * 4 + 1] += a * v[1];
        acc[c * 4 + 2] += a * v[2];
        acc[c * 4 + 3] += a * v[3];
    }
}

// token id of slot p of window b_ (image-major), see file header
__device__ __forceinline__ size_t window_token(const WaArgs& p, int b_, int slot) {
    const int nWx = p.W / WS, nW = (p.H / WS) * nWx;
    const int img = b_ / nW, w = b_ - img * nW;
    const int wy = w / nWx, wx = w - wy * nWx;
    const int i = slot / WS, j = slot - i * WS;
    int hh = wy * WS + i + p.shift;
    if (hh >= p.H) hh -= p.H;
    int ww = wx * WS + j + p.shift;
    if (ww >= p.W) ww -= p.W;
    return (size_t)img * p.H * p.W + (size_t)hh * p.W + ww;
}

__device__ __forceinline__ void fill_bias(const WaArgs& p, int head, float* Bs) {
    for (int t = threadIdx.x; t < TOK * TOK; t += 256) {
        const int i = t / TOK, j = t - i * TOK;
        Bs[i * BP + j] = p.table[p.index[t] * p.nH + head];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wattn_fwd_kernel(WaArgs p) {
    __shared__ __attribute__((aligned(16))) float Ks[4][TOK * KP];
    __shared__ __attribute__((aligned(16))) float Vs[4][TOK * KP];
    __shared__ float Bs[TOK * BP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.x / p.groups_per_head, grp = blockIdx.x - head * p.groups_per_head;
    const int nW = (p.H / WS) * (p.W / WS), B_ = p.n_img * nW;
    const int stride = p.groups_per_head * 4;
    const int slot = lane < TOK ? lane : TOK - 1;
    const T* __restrict__ qkv = reinterpret_cast<const T*>(p.qkv);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);

    fill_bias(p, head, Bs);
    const int iters = (B_ + stride - 1) / stride;
    for (int it = 0; it < iters; ++it) {
        const int b_raw = it * stride + grp * 4 + wave;
        const bool active = b_raw < B_ && lane < TOK;
        const int b_ = b_raw < B_ ? b_raw : B_ - 1;
        const size_t tok = window_token(p, b_, slot);
        const T* row = qkv + tok * 3 * p.C + head * HD;
        float q[HD], tmp[HD];
        load_row32<T>(row, q);
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] *= p.scale;
        __syncthreads();                                  // previous iteration done with Ks/Vs (and Bs filled)
        load_row32<T>(row + p.C, tmp);
        lds_put_row32(&Ks[wave][slot * KP], tmp);
        load_row32<T>(row + 2 * p.C, tmp);
        lds_put_row32(&Vs[wave][slot * KP], tmp);
        __syncthreads();

        const float* mrow = p.mask ? p.mask + ((size_t)(b_ % p.nW_mask) * TOK + slot) * TOK : nullptr;
        // online softmax over 7 tiles of 7 keys (one window row of keys per tile)
        float o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = 0.f;
        float mx = -INFINITY, l = 0.f;
#pragma unroll 1
        for (int j0 = 0; j0 < TOK; j0 += WS) {
            float s[WS];
            float tmax = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < WS; ++jj) {
                float v = lds_dot32(&Ks[wave][(j0 + jj) * KP], q) + Bs[slot * BP + j0 + jj];
                if (mrow) v += mrow[j0 + jj];
                s[jj] = v;
                tmax = fmaxf(tmax, v);
            }
            const float mnew = fmaxf(mx, tmax);
            const float alpha = __expf(mx - mnew);
            l *= alpha;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] *= alpha;
#pragma unroll
            for (int jj = 0; jj < WS; ++jj) {
                const float e = __expf(s[jj] - mnew);
                l += e;
                lds_axpy32(o, e, &Vs[wave][(j0 + jj) * KP]);
            }
            mx = mnew;
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= inv;
        if (active) {
            store_row32<T>(out + tok * p.C + head * HD, o);
            p.lse[((size_t)b_ * p.nH + head) * TOK + slot] = mx + __logf(l);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wattn_bwd_kernel(WaArgs p) {
    __shared__ __attribute__((aligned(16))) float As[4][TOK * KP];   // pass R: K      pass C: scaled Q
    __shared__ __attribute__((aligned(16))) float Cs[4][TOK * KP];   // pass R: V      pass C: dO
    __shared__ float DB[4][TOK * BP];                                // per-wave d(bias) accumulators
    __shared__ float Ls[4][64], Ds[4][64];
    __shared__ float Bs[TOK * BP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.x / p.groups_per_head, grp = blockIdx.x - head * p.groups_per_head;
    const int nW = (p.H / WS) * (p.W / WS), B_ = p.n_img * nW;
    const int stride = p.groups_per_head * 4;
    const int slot = lane < TOK ? lane : TOK - 1;
    const T* __restrict__ qkv = reinterpret_cast<const T*>(p.qkv);
    const T* __restrict__ og = reinterpret_cast<const T*>(p.out);
    const T* __restrict__ dog = reinterpret_cast<const T*>(p.dout);
    T* __restrict__ dqkv = reinterpret_cast<T*>(p.dqkv);

    fill_bias(p, head, Bs);
    for (int t = lane; t < TOK * BP; t += 64) DB[wave][t] = 0.f;

    const int iters = (B_ + stride - 1) / stride;
    for (int it = 0; it < iters; ++it) {
        const int b_raw = it * stride + grp * 4 + wave;
        const bool active = b_raw < B_ && lane < TOK;
        const int b_ = b_raw < B_ ? b_raw : B_ - 1;
        const size_t tok = window_token(p, b_, slot);
        const T* row = qkv + tok * 3 * p.C + head * HD;
        const float* mbase = p.mask ? p.mask + (size_t)(b_ % p.nW_mask) * TOK * TOK : nullptr;

        float q[HD], dO[HD];
        load_row32<T>(row, q);
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] *= p.scale;
        load_row32<T>(dog + tok * p.C + head * HD, dO);
        float delta = 0.f;
        {
            float o[HD];
            load_row32<T>(og + tok * p.C + head * HD, o);
#pragma unroll
            for (int d = 0; d < HD; ++d) delta += o[d] * dO[d];
        }
        const float lse = p.lse[((size_t)b_ * p.nH + head) * TOK + slot];

        __syncthreads();                                  // previous iteration's pass C finished (Bs/DB ready)
        {
            float tmp[HD];
            load_row32<T>(row + p.C, tmp);
            lds_put_row32(&As[wave][slot * KP], tmp);
            load_row32<T>(row + 2 * p.C, tmp);
            lds_put_row32(&Cs[wave][slot * KP], tmp);
        }
        __syncthreads();

        // ---- pass R: lane owns query row `slot`
        {
            float dq[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = 0.f;
#pragma unroll 2
            for (int j = 0; j < TOK; ++j) {
                float s = lds_dot32(&As[wave][j * KP], q) + Bs[slot * BP + j];
                if (mbase) s += mbase[slot * TOK + j];
                const float pij = __expf(s - lse);
                const float dP = lds_dot32(&Cs[wave][j * KP], dO);
                const float dS = pij * (dP - delta);
                if (active) DB[wave][slot * BP + j] += dS;       // lane-private address, wave-private tile
                lds_axpy32(dq, dS, &As[wave][j * KP]);
            }
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] *= p.scale;
            if (active) store_row32<T>(dqkv + tok * 3 * p.C + head * HD, dq);
        }
        // own key / value rows back into registers, then the tiles are recycled for scaled Q and dO
        float kreg[HD], vreg[HD];
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(&As[wave][slot * KP + c * 4]);
            const f32x4 vv = *reinterpret_cast<const f32x4*>(&Cs[wave][slot * KP + c * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { kreg[c * 4 + e] = kv[e]; vreg[c * 4 + e] = vv[e]; }
        }
        __syncthreads();
        lds_put_row32(&As[wave][slot * KP], q);
        lds_put_row32(&Cs[wave][slot * KP], dO);
        Ls[wave][lane] = lse;
        Ds[wave][lane] = delta;
        __syncthreads();

        // ---- pass C: lane owns key/value row `slot`
        {
            float dk[HD], dv[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) dk[d] = dv[d] = 0.f;
#pragma unroll 2
            for (int i = 0; i < TOK; ++i) {
                float s = lds_dot32(&As[wave][i * KP], kreg) + Bs[i * BP + slot];
                if (mbase) s += mbase[i * TOK + slot];
                const float pij = __expf(s - Ls[wave][i]);
                lds_axpy32(dv, pij, &Cs[wave][i * KP]);
                const float dP = lds_dot32(&Cs[wave][i * KP], vreg);
                const float dS = pij * (dP - Ds[wave][i]);
                lds_axpy32(dk, dS, &As[wave][i * KP]);
            }
            if (active) {
                store_row32<T>(dqkv + tok * 3 * p.C + p.C + head * HD, dk);
                store_row32<T>(dqkv + tok * 3 * p.C + 2 * p.C + head * HD, dv);
            }
        }
    }
    __syncthreads();
    float* part = p.part + ((size_t)head * stride + grp * 4 + wave) * TOK * TOK;
    for (int t = lane; t < TOK * TOK; t += 64) part[t] = DB[wave][(t / TOK) * BP + (t % TOK)];
}

// d(bias table), deterministic two-stage reduction of the per-wave partials:
//   dense[h][e] = sum_w part[h][w][e]           (one thread per (h, e): coalesced over e)
//   dtable[r][h] = sum_{e : index[e] == r} dense[h][e]
__global__ __launch_bounds__(256) void wattn_dense_kernel(const float* __restrict__ part, int nH, int waves, float* __restrict__ dense) {
    // 256 threads = 64 entries x 4 partial groups; fixed-order tree
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tx;
    float a = 0.f;
    if (t < nH * TOK * TOK) {
        const int h = t / (TOK * TOK), e = t - h * TOK * TOK;
        const float* pe = part + (size_t)h * waves * TOK * TOK + e;
        for (int w = ty; w < waves; w += 4) a += pe[(size_t)w * TOK * TOK];
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && t < nH * TOK * TOK) dense[t] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

__global__ __launch_bounds__(64) void wattn_dtable_kernel(const float* __restrict__ dense, const int32_t* __restrict__ index,
                                                          int nH, float* __restrict__ dtable) {
    // one wave per table entry (r, h): lanes stride over the 2401 (i,j) pairs, fixed-order wave reduction
    const int t = blockIdx.x;
    const int r = t / nH, h = t - r * nH;
    float a = 0.f;
    for (int e = threadIdx.x; e < TOK * TOK; e += 64)
        if (index[e] == r) a += dense[h * TOK * TOK + e];
    a = wave_sum(a);
    if (threadIdx.x == 0) dtable[t] = a;
}

int wa_groups_per_head(int B_, int nH, bool bwd, bool mfma) {
    // Persistent workgroups pinned to a head.  The grid must not exceed what is co-resident (a second
    // round of workgroups would find its share of windows already sized for the full grid):
    //   MFMA fwd: 36 KB LDS, 116 VGPR -> 4 per CU (1024);  MFMA bwd (two waves per window): 57 KB LDS, <= 256
    //   registers -> 2 per CU (512);
    //   fp32 VALU kernels: ~8 / 1-2 per CU, parity path only.
    static const int fwd_wgs = fmmt_const("FMMT_WA_FWD_WGS", 1024);
    const int target = mfma ? (bwd ? 512 : fwd_wgs) : 2048;
    int g = target / nH;
    const int wpi = (mfma && bwd) ? 2 : 4;      // windows per workgroup iteration (MFMA backward: two waves per window)
    const int maxg = (B_ + wpi - 1) / wpi;
    if (g > maxg) g = maxg;
    const int cap = mfma ? WA_BWD_WAVES_PER_HEAD_MAX : WA_BWD_WAVES_PER_HEAD_MAX / 4;
    if (bwd && g > cap) g = cap;
    if (mfma && g >= 8) g &= ~7;          // multiple of 8: lets the kernel keep a window group's heads on one XCD
    return g < 1 ? 1 : g;
}

int wa_xcd_grouped(int groups_per_head, bool mfma) {
    static const int on = fmmt_const("FMMT_WA_XCD", 1);
    return (on && mfma && groups_per_head % 8 == 0) ? 1 : 0;
}

int wa_check(int dtype, int n_img, int H, int W, int C, int nH, int shift) {
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    if (n_img <= 0 || H <= 0 || W <= 0 || H % WS || W % WS) return FMMT_EINVAL;
    if (nH <= 0 || C != nH * HD) return FMMT_EINVAL;
    if (shift < 0 || shift >= WS) return FMMT_EINVAL;
    return 0;
}

// =============================================================================================
// Cross-modal multi-head attention (head_dim 64 or 32), time-major operands.
// =============================================================================================

template <typename T, int D> __device__ __forceinline__ void load_row(const T* p, float* r) {
    constexpr int VEC = Vec<T>::N;
#pragma unroll
    for (int c = 0; c < D / VEC; ++c) {
        const Vec<T> v = ldvec<T>(p + c * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) r[c * VEC + e] = v.get(e);
    }
}
template <typename T, int D> __device__ __forceinline__ void store_row(T* p, const float* r) {
    constexpr int VEC = Vec<T>::N;
#pragma unroll
    for (int c = 0; c < D / VEC; ++c) {
        Vec<T> v;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v.set(e, r[c * VEC + e]);
        stvec<T>(p + c * VEC, v);
    }
}
template <int D> __device__ __forceinline__ float ldot(const float* s, const float* r) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + c * 4);
        a += v[0] * r[c * 4] + v[1] * r[c * 4 + 1] + v[2] * r[c * 4 + 2] + v[3] * r[c * 4 + 3];
    }
    return a;
}
template <int D> __device__ __forceinline__ void laxpy(float* acc, float a, const float* s) {
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + c * 4);
        acc[c * 4] += a * v[0];
        acc[c * 4 + 1] += a * v[1];
        acc[c * 4 + 2] += a * v[2];
        acc[c * 4 + 3] += a * v[3];
    }
}

constexpr int MT = 32;   // rows of the "other side" staged in LDS per step

// stage MT rows [r0, r0+MT) of a time-major tensor (row t -> base + (t*B + b)*ld + h*D) as fp32 into
// LDS [MT][D+4]; 64 lanes: lane -> (row = lane/2, half = lane&1)
template <typename T, int D, bool QROWS>
__device__ __forceinline__ void stage_rows(const MhaArgs& p, const T* base, int ld, int b, int h, int r0, int L, float mul,
                                           float* S, int lane) {
    constexpr int P = D + 4;
    const int r = lane >> 1, half = lane & 1;
    const int t = min(r0 + r, L - 1);
    float tmp[D / 2];
    load_row<T, D / 2>(base + (QROWS ? p.rq(t, b) : p.rk(t, b)) * ld + h * D + half * (D / 2), tmp);
#pragma unroll
    for (int c = 0; c < D / 8; ++c)
        *reinterpret_cast<f32x4*>(S + r * P + half * (D / 2) + c * 4) =
            f32x4{tmp[c * 4] * mul, tmp[c * 4 + 1] * mul, tmp[c * 4 + 2] * mul, tmp[c * 4 + 3] * mul};
}

__device__ __forceinline__ float keep_scale(const MhaArgs& p, int bh, int i, int j) {
    if (p.drop_p <= 0.f) return 1.f;
    const AttnDrop d = attn_drop_setup(p.drop_p, p.seed_dev ? *p.seed_dev : p.seed, p.Lk);
    uint32_t a, b;
    attn_drop_words(d, (uint32_t)bh * (uint32_t)p.Lq + (uint32_t)i, (uint32_t)j >> 2, a, b);
    return attn_drop_field(d, a, b, j & 3);
}
// additive logit bias of key j (the BERT-style "(1 - mask) * -10000" extended attention mask); 0 if absent
__device__ __forceinline__ float key_bias(const MhaArgs& p, int b, int j) {
    return p.key_bias ? p.key_bias[(size_t)b * p.Lk + min(j, p.Lk - 1)] : 0.f;
}

// forward: lane owns a query row; keys/values streamed through LDS in tiles of MT; online softmax
template <typename T, int D>
__global__ __launch_bounds__(64) void mha_fwd_kernel(MhaArgs p) {
    constexpr int P = D + 4;
    __shared__ __attribute__((aligned(16))) float Ks[MT * P];
    __shared__ __attribute__((aligned(16))) float Vs[MT * P];
    const int lane = threadIdx.x;
    const int bh = blockIdx.y, b = bh / p.nH, h = bh - b * p.nH;
    const int i_raw = blockIdx.x * 64 + lane;
    const bool active = i_raw < p.Lq;
    const int i = active ? i_raw : p.Lq - 1;
    const T* qg = reinterpret_cast<const T*>(p.q);
    float q[D], o[D];
    load_row<T, D>(qg + p.rq(i, b) * p.ldq + h * D, q);
#pragma unroll
    for (int d = 0; d < D; ++d) { q[d] *= p.scale; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < p.Lk; j0 += MT) {
        __syncthreads();
        stage_rows<T, D, false>(p, reinterpret_cast<const T*>(p.k), p.ldkv, b, h, j0, p.Lk, 1.f, Ks, lane);
        stage_rows<T, D, false>(p, reinterpret_cast<const T*>(p.v), p.ldkv, b, h, j0, p.Lk, 1.f, Vs, lane);
        __syncthreads();
#pragma unroll 1
        for (int js = 0; js < MT; js += 8) {
            if (j0 + js >= p.Lk) break;
            float s[8];
            float tmax = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                s[jj] = (j0 + js + jj < p.Lk) ? ldot<D>(Ks + (js + jj) * P, q) + key_bias(p, b, j0 + js + jj) : -INFINITY;
                tmax = fmaxf(tmax, s[jj]);
            }
            const float mnew = fmaxf(m, tmax);
            const float alpha = __expf(m - mnew);          // m = -inf on the first tile -> 0
            l *= alpha;
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] *= alpha;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float e = __expf(s[jj] - mnew);       // -inf -> 0
                l += e;
                const float ks = keep_scale(p, bh, i, j0 + js + jj);
                laxpy<D>(o, e * ks, Vs + (js + jj) * P);
            }
            m = mnew;
        }
    }
    if (active) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] *= inv;
        store_row<T, D>(reinterpret_cast<T*>(p.out) + p.rq(i, b) * p.ldo + h * D, o);
        p.lse[(size_t)bh * p.Lq + i] = m + __logf(l);
    }
}

// backward, row-owner pass: dq_i = scale * sum_j dS_ij k_j
template <typename T, int D>
__global__ __launch_bounds__(64) void mha_bwd_dq_kernel(MhaArgs p) {
    constexpr int P = D + 4;
    __shared__ __attribute__((aligned(16))) float Ks[MT * P];
    __shared__ __attribute__((aligned(16))) float Vs[MT * P];
    const int lane = threadIdx.x;
    const int bh = blockIdx.y, b = bh / p.nH, h = bh - b * p.nH;
    const int i_raw = blockIdx.x * 64 + lane;
    const bool active = i_raw < p.Lq;
    const int i = active ? i_raw : p.Lq - 1;
    float q[D], dO[D], dq[D];
    load_row<T, D>(reinterpret_cast<const T*>(p.q) + p.rq(i, b) * p.ldq + h * D, q);
    load_row<T, D>(reinterpret_cast<const T*>(p.dout) + p.rq(i, b) * p.ldo + h * D, dO);
    float delta = 0.f;
    {
        float o[D];
        load_row<T, D>(reinterpret_cast<const T*>(p.out) + p.rq(i, b) * p.ldo + h * D, o);
#pragma unroll
        for (int d = 0; d < D; ++d) delta += o[d] * dO[d];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) { q[d] *= p.scale; dq[d] = 0.f; }
    const float lse = p.lse[(size_t)bh * p.Lq + i];
    for (int j0 = 0; j0 < p.Lk; j0 += MT) {
        __syncthreads();
        stage_rows<T, D, false>(p, reinterpret_cast<const T*>(p.k), p.ldkv, b, h, j0, p.Lk, 1.f, Ks, lane);
        stage_rows<T, D, false>(p, reinterpret_cast<const T*>(p.v), p.ldkv, b, h, j0, p.Lk, 1.f, Vs, lane);
        __syncthreads();
#pragma unroll 2
        for (int j = 0; j < MT; ++j) {
            if (j0 + j >= p.Lk) break;
            const float pij = __expf(ldot<D>(Ks + j * P, q) + key_bias(p, b, j0 + j) - lse);
            const float dP = ldot<D>(Vs + j * P, dO) * keep_scale(p, bh, i, j0 + j);
            laxpy<D>(dq, pij * (dP - delta), Ks + j * P);
        }
    }
    if (active) {
#pragma unroll
        for (int d = 0; d < D; ++d) dq[d] *= p.scale;
        store_row<T, D>(reinterpret_cast<T*>(p.dq) + p.rq(i, b) * p.lddq + h * D, dq);
    }
}

// backward, column-owner passes.  WHICH = 0: dv_j = sum_i pdrop_ij dO_i ; WHICH = 1: dk_j = sum_i dS_ij (scale q_i)
template <typename T, int D, int WHICH>
__global__ __launch_bounds__(64) void mha_bwd_dkv_kernel(MhaArgs p) {
    constexpr int P = D + 4;
    __shared__ __attribute__((aligned(16))) float Qs[MT * P];
    __shared__ __attribute__((aligned(16))) float Gs[MT * P];
    __shared__ float Ls[MT], Dl[MT];
    const int lane = threadIdx.x;
    const int bh = blockIdx.y, b = bh / p.nH, h = bh - b * p.nH;
    const int j_raw = blockIdx.x * 64 + lane;
    const bool active = j_raw < p.Lk;
    const int j = active ? j_raw : p.Lk - 1;
    float kreg[D], acc[D];
    load_row<T, D>(reinterpret_cast<const T*>(p.k) + p.rk(j, b) * p.ldkv + h * D, kreg);
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    const float kb = key_bias(p, b, j);
    float vreg[WHICH == 1 ? D : 1];
    if constexpr (WHICH == 1) load_row<T, D>(reinterpret_cast<const T*>(p.v) + p.rk(j, b) * p.ldkv + h * D, vreg);
    const T* og = reinterpret_cast<const T*>(p.out);
    const T* dog = reinterpret_cast<const T*>(p.dout);
    for (int i0 = 0; i0 < p.Lq; i0 += MT) {
        __syncthreads();
        stage_rows<T, D, true>(p, reinterpret_cast<const T*>(p.q), p.ldq, b, h, i0, p.Lq, p.scale, Qs, lane);
        stage_rows<T, D, true>(p, dog, p.ldo, b, h, i0, p.Lq, 1.f, Gs, lane);
        if (lane < MT) {
            const int i = min(i0 + lane, p.Lq - 1);
            Ls[lane] = p.lse[(size_t)bh * p.Lq + i];
            if constexpr (WHICH == 1) {
                float o[D], g[D];
                load_row<T, D>(og + p.rq(i, b) * p.ldo + h * D, o);
                load_row<T, D>(dog + p.rq(i, b) * p.ldo + h * D, g);
                float dl = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) dl += o[d] * g[d];
                Dl[lane] = dl;
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int ii = 0; ii < MT; ++ii) {
            if (i0 + ii >= p.Lq) break;
            const float pij = __expf(ldot<D>(Qs + ii * P, kreg) + kb - Ls[ii]);
            const float ks = keep_scale(p, bh, i0 + ii, j);
            if constexpr (WHICH == 0) {
                laxpy<D>(acc, pij * ks, Gs + ii * P);
            } else {
                const float dP = ldot<D>(Gs + ii * P, vreg) * ks;
                laxpy<D>(acc, pij * (dP - Dl[ii]), Qs + ii * P);
            }
        }
    }
    if (active) {
        T* dst = reinterpret_cast<T*>(WHICH == 0 ? p.dv : p.dk);
        store_row<T, D>(dst + p.rk(j, b) * p.lddkv + h * D, acc);
    }
}

int mha_check(int dtype, int Lq, int Lk, int B, int E, int nH) {
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    if (Lq <= 0 || Lk <= 0 || B <= 0 || nH <= 0 || E % nH) return FMMT_EINVAL;
    const int hd = E / nH;
    if (hd != 64 && hd != 32) return FMMT_EINVAL;
    return 0;
}


// Head-averaged (post-dropout) attention weights, multihead_attention.py:133-134: `attn_weights.view(bsz, heads, tgt, src).sum(1) / heads`.
// No caller of the reference uses them (CrossmodalTransformer.py:147,151 discard them), so the attention kernels never form
// the (B*heads, Lq, Lk) tensor; this kernel recomputes the probabilities from q, k and the saved log-sum-exp on request.
// One block per (query row, batch), thread j owns key j (strided); fp32 arithmetic for either storage type.
template <typename T>
__global__ __launch_bounds__(128) void mha_avg_weights_kernel(MhaArgs p, float* __restrict__ w) {
    extern __shared__ float qs[];                                   // the query row, all heads, pre-scaled
    const int i = blockIdx.x, b = blockIdx.y, hd = p.E / p.nH;
    const T* qg = reinterpret_cast<const T*>(p.q) + p.rq(i, b) * p.ldq;
    const T* kg = reinterpret_cast<const T*>(p.k);
    for (int c = threadIdx.x; c < p.E; c += 128) qs[c] = to_f32(qg[c]) * p.scale;
    __syncthreads();
    for (int j = threadIdx.x; j < p.Lk; j += 128) {
        const T* kr = kg + p.rk(j, b) * p.ldkv;
        const float kb = key_bias(p, b, j);
        float acc = 0.f;
        for (int h = 0; h < p.nH; ++h) {
            const int bh = b * p.nH + h;
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qs[h * hd + d] * to_f32(kr[h * hd + d]);
            acc += __expf(s + kb - p.lse[(size_t)bh * p.Lq + i]) * keep_scale(p, bh, i, j);
        }
        w[((size_t)b * p.Lq + i) * p.Lk + j] = acc / (float)p.nH;
    }
}

}  // namespace

extern "C" int fmmt_window_attn_fwd(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                                    const void* qkv, const float* table, const int32_t* index,
                                    const float* mask, int nW_mask, int mask_is_shift, float scale,
                                    void* out, float* lse, void* stream) {
    if (int e = wa_check(dtype, n_img, H, W, C, num_heads, shift)) return e;
    if (mask && nW_mask <= 0) return FMMT_EINVAL;
    WaArgs a{};
    a.n_img = n_img; a.H = H; a.W = W; a.C = C; a.nH = num_heads; a.shift = shift; a.qkv = qkv; a.table = table;
    a.index = index; a.mask = mask; a.nW_mask = nW_mask; a.mask_is_shift = mask_is_shift; a.scale = scale; a.out = out; a.lse = lse;
    const int B_ = n_img * (H / WS) * (W / WS);
    a.groups_per_head = wa_groups_per_head(B_, num_heads, false, dtype == FMMT_BF16);
    a.xcd_grouped = wa_xcd_grouped(a.groups_per_head, dtype == FMMT_BF16);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(num_heads * a.groups_per_head);
    if (dtype == FMMT_BF16) return fmmt_wattn_mfma_fwd_launch(a, (int)grid.x, st);     // matrix-core path
    hipLaunchKernelGGL(wattn_fwd_kernel<float>, grid, dim3(256), 0, st, a);             // exact-fp32 parity path
    FMMT_CHECK_LAUNCH();
    return 0;
}

static size_t wa_part_bytes(int num_heads) { return ((size_t)num_heads * (WA_BWD_WAVES_PER_HEAD_MAX + 1) * TOK * TOK * sizeof(float) + 255) / 256 * 256; }

extern "C" size_t fmmt_window_attn_bwd_workspace(int num_heads) {
    return wa_part_bytes(num_heads) + WA_SINK_BYTES;        // per-wave partials + dense, then the store sink (wattn_args.h)
}

extern "C" int fmmt_window_attn_bwd(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                                    const void* qkv, const void* out, const void* dout, const float* lse,
                                    const float* table, const int32_t* index, const float* mask, int nW_mask,
                                    int mask_is_shift, float scale, void* dqkv, float* dtable,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    // parity instantiations (wattn_bwd_ref.hip, operands read instead of recomputed): FMMT_F32 with no mask or the standard SW-MSA mask, and
    // FMMT_BF16 | FMMT_GENERIC (the bf16 instantiation of the same template, for the test that holds it against the production kernel).
    // fp32 with an explicit mask tensor stays on the VALU kernel below (a different algorithm; tests use it as a cross-check).
    const bool want_generic = (dtype & FMMT_GENERIC) != 0;
    dtype &= 0xff;
    if (int e = wa_check(dtype, n_img, H, W, C, num_heads, shift)) return e;
    if (mask && nW_mask <= 0) return FMMT_EINVAL;
    if (workspace_bytes < fmmt_window_attn_bwd_workspace(num_heads)) return FMMT_EWORKSPACE;
    const bool generic_ok = !mask || (mask_is_shift && shift > 0);
    if (want_generic && !generic_ok) return FMMT_EINVAL;
    const bool generic = want_generic || (dtype == FMMT_F32 && generic_ok);
    WaArgs a{};
    a.n_img = n_img; a.H = H; a.W = W; a.C = C; a.nH = num_heads; a.shift = shift; a.qkv = qkv; a.table = table;
    a.index = index; a.mask = mask; a.nW_mask = nW_mask; a.mask_is_shift = mask_is_shift; a.scale = scale; a.out = const_cast<void*>(out);
    a.lse = const_cast<float*>(lse); a.dout = dout; a.dqkv = dqkv; a.part = reinterpret_cast<float*>(workspace);
    a.sink = reinterpret_cast<char*>(workspace) + wa_part_bytes(num_heads);
    const int B_ = n_img * (H / WS) * (W / WS);
    a.groups_per_head = wa_groups_per_head(B_, num_heads, true, dtype == FMMT_BF16);
    a.xcd_grouped = wa_xcd_grouped(a.groups_per_head, dtype == FMMT_BF16);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (generic) {
        a.groups_per_head = fmmt_wattn_bwd_groups(B_, num_heads, 1);                    // one pair of waves (one window at a time) per workgroup
        a.xcd_grouped = 0;
        if (int rc = fmmt_wattn_bwd_ref_launch(dtype, a, num_heads * a.groups_per_head, st)) return rc;
        return fmmt_wattn_dtable_finish(a.part, num_heads, a.groups_per_head, index, dtable, st);
    }
    dim3 grid(num_heads * a.groups_per_head);
    if (dtype == FMMT_BF16) {
        if (int rc = fmmt_wattn_mfma_bwd_launch(a, (int)grid.x, st)) return rc;
    } else {
        hipLaunchKernelGGL(wattn_bwd_kernel<float>, grid, dim3(256), 0, st, a);
        FMMT_CHECK_LAUNCH();
    }
    return fmmt_wattn_dtable_finish(a.part, num_heads, dtype == FMMT_BF16 ? a.groups_per_head : a.groups_per_head * 4, index, dtable, st);
}

static int wba_run(bool generic, int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                   const void* xn, const void* dy, const void* attn_out, const float* lse,
                   const void* wqkv, const float* bqkv, const void* wproj,
                   const float* table, const int32_t* index, float scale, const float* rowscale,
                   void* dqkv, float* dtable, void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = wa_check(dtype, n_img, H, W, C, num_heads, shift)) return e;
    generic = generic || dtype == FMMT_F32;                                              // fp32 = the generic restatement (wattn_bwd_ref.hip), C = 96
    if ((dtype != FMMT_BF16 && dtype != FMMT_F32) || (C != 96 && C != 192) || (generic && C != 96)) return FMMT_EINVAL;   // other widths: fmmt_window_attn_bwd on a materialised qkv
    if (!xn || !dy || !attn_out || !lse || !wqkv || !wproj || !table || !index || !dqkv || !dtable || !workspace) return FMMT_EINVAL;
    if (workspace_bytes < fmmt_window_attn_bwd_workspace(num_heads)) return FMMT_EWORKSPACE;
    WaArgs a{};
    a.n_img = n_img; a.H = H; a.W = W; a.C = C; a.nH = num_heads; a.shift = shift; a.table = table; a.index = index; a.scale = scale;
    a.mask_is_shift = shift > 0; a.out = const_cast<void*>(attn_out); a.lse = const_cast<float*>(lse); a.dout = dy; a.dqkv = dqkv;
    a.part = reinterpret_cast<float*>(workspace);
    a.sink = reinterpret_cast<char*>(workspace) + wa_part_bytes(num_heads);
    a.xn = xn; a.wqkv = wqkv; a.bqkv = bqkv; a.wproj = wproj; a.rowscale = rowscale;
    const int B_ = n_img * (H / WS) * (W / WS);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (generic) {
        a.groups_per_head = fmmt_wattn_bwd_groups(B_, num_heads, 1);                    // one pair of waves (one window at a time) per workgroup
        if (int rc = fmmt_wattn_bwd_ref_launch(dtype, a, num_heads * a.groups_per_head, st)) return rc;
        return fmmt_wattn_dtable_finish(a.part, num_heads, a.groups_per_head, index, dtable, st);
    }
    a.groups_per_head = fmmt_wattn_bwd_groups(B_, num_heads, 4);
    a.xcd_grouped = wa_xcd_grouped(a.groups_per_head, true);
    if (int rc = fmmt_wattn_mfma_bwd_rc_launch(a, num_heads * a.groups_per_head, st)) return rc;
    return fmmt_wattn_dtable_finish(a.part, num_heads, a.groups_per_head, index, dtable, st);
}

extern "C" int fmmt_window_block_attn_bwd(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                                          const void* xn, const void* dy, const void* attn_out, const float* lse,
                                          const void* wqkv, const float* bqkv, const void* wproj,
                                          const float* table, const int32_t* index, float scale, const float* rowscale,
                                          void* dqkv, float* dtable, void* workspace, size_t workspace_bytes, void* stream) {
    return wba_run(false, dtype, n_img, H, W, C, num_heads, shift, xn, dy, attn_out, lse, wqkv, bqkv, wproj, table, index, scale, rowscale,
                   dqkv, dtable, workspace, workspace_bytes, stream);
}

// the element-type-generic restatement (wattn_bwd_ref.hip): FMMT_F32 = what fmmt_window_block_attn_bwd(FMMT_F32) runs; FMMT_BF16 = the
// generic template's bf16 instantiation, compared with the production kernel by tests/test_gpu_wblock.py
extern "C" int fmmt_window_block_attn_bwd_ref(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                                              const void* xn, const void* dy, const void* attn_out, const float* lse,
                                              const void* wqkv, const float* bqkv, const void* wproj,
                                              const float* table, const int32_t* index, float scale, const float* rowscale,
                                              void* dqkv, float* dtable, void* workspace, size_t workspace_bytes, void* stream) {
    return wba_run(true, dtype, n_img, H, W, C, num_heads, shift, xn, dy, attn_out, lse, wqkv, bqkv, wproj, table, index, scale, rowscale,
                   dqkv, dtable, workspace, workspace_bytes, stream);
}

// per-workgroup dense d(bias) partials [num_heads][parts_per_head][49 * 49] (in a workspace of fmmt_window_attn_bwd_workspace bytes)
// -> dtable [169][num_heads], fixed summation order
int fmmt_wattn_dtable_finish(float* part, int num_heads, int parts_per_head, const int32_t* index, float* dtable, hipStream_t st) {
    const int nt = (2 * WS - 1) * (2 * WS - 1) * num_heads;
    float* dense = part + (size_t)num_heads * WA_BWD_WAVES_PER_HEAD_MAX * TOK * TOK;
    const int nd = num_heads * TOK * TOK;
    hipLaunchKernelGGL(wattn_dense_kernel, dim3((nd + 63) / 64), dim3(256), 0, st, part, num_heads, parts_per_head, dense);
    FMMT_CHECK_LAUNCH();
    hipLaunchKernelGGL(wattn_dtable_kernel, dim3(nt), dim3(64), 0, st, dense, index, num_heads, dtable);
    FMMT_CHECK_LAUNCH();
    return 0;
}

int fmmt_wattn_bwd_groups(int B_, int num_heads, int windows_per_iteration) {
    int g = 256 / num_heads;                                 // one 8-wave workgroup per CU
    const int maxg = (B_ + windows_per_iteration - 1) / windows_per_iteration;
    if (g > maxg) g = maxg;
    if (g > WA_BWD_WAVES_PER_HEAD_MAX) g = WA_BWD_WAVES_PER_HEAD_MAX;
    if (g >= 8) g &= ~7;
    return g < 1 ? 1 : g;
}

#define FMMT_MHA_DISPATCH(KERNEL, GRIDX)                                                             \
    do {                                                                                             \
        dim3 grid((GRIDX), B * num_heads);                                                           \
        const int hd = E / num_heads;                                                                \
        if (dtype == FMMT_BF16) {                                                                    \
            if (hd == 64) hipLaunchKernelGGL((KERNEL(bf16, 64)), grid, dim3(64), 0, st, a);          \
            else hipLaunchKernelGGL((KERNEL(bf16, 32)), grid, dim3(64), 0, st, a);                   \
        } else {                                                                                     \
            if (hd == 64) hipLaunchKernelGGL((KERNEL(float, 64)), grid, dim3(64), 0, st, a);         \
            else hipLaunchKernelGGL((KERNEL(float, 32)), grid, dim3(64), 0, st, a);                  \
        }                                                                                            \
        FMMT_CHECK_LAUNCH();                                                                         \
    } while (0)

#define K_FWD(T, D) mha_fwd_kernel<T, D>
#define K_DQ(T, D) mha_bwd_dq_kernel<T, D>
#define K_DV(T, D) mha_bwd_dkv_kernel<T, D, 0>
#define K_DK(T, D) mha_bwd_dkv_kernel<T, D, 1>

extern "C" int fmmt_mha_avg_weights(int dtype, int Lq, int Lk, int B, int E, int num_heads, const void* q, int ldq, const void* k, int ldkv,
                                    float scale, const float* key_bias, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                    const float* lse, float* weights, void* stream) {
    if (int e = mha_check(dtype, Lq, Lk, B, E, num_heads)) return e;
    if (!q || !k || !lse || !weights || dropout_p < 0.f || dropout_p >= 1.f) return FMMT_EINVAL;
    MhaArgs a{};
    a.Lq = Lq; a.Lk = Lk; a.B = B; a.E = E; a.nH = num_heads; a.q = q; a.ldq = ldq; a.k = k; a.ldkv = ldkv; a.scale = scale;
    a.key_bias = key_bias; a.drop_p = dropout_p; a.seed = seed; a.seed_dev = seed_dev; a.lse = const_cast<float*>(lse);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FMMT_BF16) hipLaunchKernelGGL(mha_avg_weights_kernel<bf16>, dim3(Lq, B), dim3(128), (size_t)E * sizeof(float), st, a, weights);
    else hipLaunchKernelGGL(mha_avg_weights_kernel<float>, dim3(Lq, B), dim3(128), (size_t)E * sizeof(float), st, a, weights);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_mha_fwd(int dtype, int Lq, int Lk, int B, int E, int num_heads,
                            const void* q, int ldq, const void* k, const void* v, int ldkv, float scale, const float* key_bias,
                            float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* out, int ldo, float* lse, void* stream) {
    const int bm = (dtype & FMMT_BATCH_MAJOR) ? 1 : 0;
    dtype &= ~FMMT_BATCH_MAJOR;
    if (int e = mha_check(dtype, Lq, Lk, B, E, num_heads)) return e;
    if (dropout_p < 0.f || dropout_p >= 1.f) return FMMT_EINVAL;
    MhaArgs a{};
    a.Lq = Lq; a.Lk = Lk; a.B = B; a.E = E; a.nH = num_heads; a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ldkv = ldkv;
    a.scale = scale; a.key_bias = key_bias; a.drop_p = dropout_p; a.seed = seed; a.seed_dev = seed_dev; a.out = out; a.ldo = ldo; a.lse = lse; a.bm = bm;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FMMT_BF16 && E / num_heads == 64 && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0)
        return fmmt_mha_mfma_fwd_launch(a, st);                      // matrix-core path
    FMMT_MHA_DISPATCH(K_FWD, (Lq + 63) / 64);
    return 0;
}

extern "C" int fmmt_mha_bwd(int dtype, int Lq, int Lk, int B, int E, int num_heads,
                            const void* q, int ldq, const void* k, const void* v, int ldkv, float scale, const float* key_bias,
                            float dropout_p, uint64_t seed, const uint64_t* seed_dev, const void* out, const void* dout, int ldo,
                            const float* lse, void* dq, int lddq, void* dk, void* dv, int lddkv, void* stream) {
    const int bm = (dtype & FMMT_BATCH_MAJOR) ? 1 : 0;
    dtype &= ~FMMT_BATCH_MAJOR;
    if (int e = mha_check(dtype, Lq, Lk, B, E, num_heads)) return e;
    if (dropout_p < 0.f || dropout_p >= 1.f) return FMMT_EINVAL;
    MhaArgs a{};
    a.Lq = Lq; a.Lk = Lk; a.B = B; a.E = E; a.nH = num_heads; a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ldkv = ldkv;
    a.scale = scale; a.key_bias = key_bias; a.drop_p = dropout_p; a.seed = seed; a.seed_dev = seed_dev; a.out = const_cast<void*>(out); a.ldo = ldo;
    a.lse = const_cast<float*>(lse); a.dout = dout; a.dq = dq; a.lddq = lddq; a.dk = dk; a.dv = dv; a.lddkv = lddkv; a.bm = bm;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FMMT_BF16 && E / num_heads == 64 && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0 && lddq % 8 == 0 && lddkv % 8 == 0)
        return fmmt_mha_mfma_bwd_launch(a, st);                      // matrix-core path
    FMMT_MHA_DISPATCH(K_DQ, (Lq + 63) / 64);
    FMMT_MHA_DISPATCH(K_DV, (Lk + 63) / 64);
    FMMT_MHA_DISPATCH(K_DK, (Lk + 63) / 64);
    return 0;
}
