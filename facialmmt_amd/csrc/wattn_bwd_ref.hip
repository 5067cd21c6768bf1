// The recompute backward of the fused block half (wattn_mfma.hip: wattn_mfma_bwd_kernel<MM, 4, 96>, behind fmmt_window_block_attn_bwd)
// restated over an ELEMENT-TYPE TRAIT, for parity -- the companion of wblock_ref.hip.
//
//     dqkv, d(table) of WindowAttention (Swin_Transformer.py:113-144) from LN1(x), dy (gradient of the block half's OUTPUT), the saved
//     attention output and log-sum-exp; q / k / v and d(attention output) = rowscale * dy . Wproj[:, head] re-formed inside the kernel
//
// Same algorithm as the production kernel, statement for statement: two waves per (window, head), wave h of the pair owns token tiles
// {2h, 2h + 1} as QUERY tiles in pass 1 (dQ, d(bias)) and as KEY tiles in pass 2 (dK, dV); the partner's fragments come from the pair's
// natural-layout LDS tiles; probabilities are recomputed from the log-sum-exp in both orientations; d(bias) accumulates in registers
// and is reduced per workgroup in wave order.  What differs is what the trait hides: fragments of 8 fp32 (two 16-byte accesses), the
// 32-deep product as 8 x v_mfma_f32_16x16x4_f32 (same accumulator layout), the transposed LDS fragment as 8 scalar reads instead of
// ds_read_b64_tr_b16, no prefetch of the next problem (an implementation detail of the fast kernel, not of the algorithm), one pair
// per workgroup (the fp32 tiles are twice the bytes), and nothing rounded to bf16.  The bf16 instantiation of this template exists so
// that a test can hold it against the production kernel; the fp32 one is what fmmt_window_block_attn_bwd(FMMT_F32) runs and what the
// reference-generated gradient goldens reach at 1e-3.
#include "fmmt_common.h"
#include "../../include/fmmt.h"
#include "wattn_args.h"
#include "wattn_geom.h"

namespace {

typedef __attribute__((ext_vector_type(8))) float f32x8;

template <typename T> struct WrEl;
template <> struct WrEl<bf16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ frag ld(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ void st(bf16* p, const frag& f) { *reinterpret_cast<bf16x8*>(p) = f; }
    static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ bf16 cv(float v) { return (bf16)v; }
};
template <> struct WrEl<float> {
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

template <typename T> __device__ __forceinline__ typename WrEl<T>::frag wr_zero() {
    typename WrEl<T>::frag z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = WrEl<T>::cv(0.f);
    return z;
}
template <typename T> __device__ __forceinline__ typename WrEl<T>::frag wr_pack(const float* lo4, const float* hi4) {
    typename WrEl<T>::frag v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = WrEl<T>::cv(lo4[e]); v[4 + e] = WrEl<T>::cv(hi4[e]); }
    return v;
}
// wattn_geom.h::tr_fragT without the transposing LDS instruction: A-operand fragment of X^T for X stored [row][TP] in LDS.
// MFMA row i <-> channel (i >> 2) * 8 + dt * 4 + (i & 3); k-slot (g, e) <-> row r0 + e (e < 4), r0 + 16 + (e - 4) (e >= 4), r0 = 32 ks + 4 g
template <typename T> __device__ __forceinline__ typename WrEl<T>::frag wr_fragT(const T* tile, int r0, int dt, int li) {
    const int c = (li >> 2) * 8 + dt * 4 + (li & 3);
    typename WrEl<T>::frag v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(r0 + (e < 4 ? e : 16 + e - 4)) * TP + c];
    return v;
}

struct RSlot { int di, dj; bool valid; };
__device__ __forceinline__ RSlot rslot_of(int slot) {
    RSlot S;
    S.valid = slot < TOK;
    const int cs = S.valid ? slot : TOK - 1;
    S.di = cs / WS;
    S.dj = cs - S.di * WS;
    return S;
}

template <typename T, int RC> struct WrLds {
    static constexpr int TILE = 64 * TP * (int)sizeof(T);
    static constexpr int TILES = 4 * TILE, STATS = 2 * 64 * 4, BIAS = 64 * BPM * 4;
    static constexpr int WP = RC + 8;
    static constexpr int WGT = RC ? 128 * WP * (int)sizeof(T) + 96 * 4 : 0;              // RC = 0: q / k / v / d(out) are read, no weights
    static constexpr int TOTAL = TILES + STATS + BIAS + WGT;
};

// MM: 0 = no mask, 1 = the standard SW-MSA mask derived from window coordinates
template <typename T, int MM, int RC>
__global__ __launch_bounds__(128) void wattn_bwd_ref_kernel(WaArgs p) {
    using E = WrEl<T>;
    using F = typename E::frag;
    using L = WrLds<T, RC>;
    constexpr int KS = RC / 32;
    extern __shared__ __attribute__((aligned(16))) char smem_ref[];
    T* kt_ = reinterpret_cast<T*>(smem_ref);
    T* qt_ = kt_ + 64 * TP;
    T* gt_ = qt_ + 64 * TP;
    T* vt_ = gt_ + 64 * TP;
    float* Ls = reinterpret_cast<float*>(smem_ref + L::TILES);
    float* Dl = Ls + 64;
    float* Bs = reinterpret_cast<float*>(smem_ref + L::TILES + L::STATS);
    T* Wh = reinterpret_cast<T*>(smem_ref + L::TILES + L::STATS + L::BIAS);          // 96 rows of Wqkv + 32 "rows" of Wproj^T, fragment order
    float* bq = reinterpret_cast<float*>(Wh + 128 * L::WP);
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int head = blockIdx.x / p.groups_per_head, grp = blockIdx.x - head * p.groups_per_head;
    const int nW = (p.H / WS) * (p.W / WS), B_ = p.n_img * nW;
    const int stride = p.groups_per_head;
    const T* __restrict__ og = reinterpret_cast<const T*>(p.out);
    const T* __restrict__ dog = reinterpret_cast<const T*>(p.dout);
    const T* __restrict__ xng = reinterpret_cast<const T*>(p.xn);
    T* __restrict__ dqkv = reinterpret_cast<T*>(p.dqkv);
    const LaneGeom G = lane_geom(li, lg, p.shift);
    const RSlot own[2] = {rslot_of((2 * h) * 16 + li), rslot_of((2 * h + 1) * 16 + li)};

    for (int t = threadIdx.x; t < 64 * BPM; t += 128) {
        const int q = t / BPM, k = t - q * BPM;
        Bs[t] = (q < TOK && k < TOK) ? p.table[p.index[q * TOK + k] * p.nH + head] : NEG_BIG;
    }
    if constexpr (RC > 0) {
        // rows in FRAGMENT order: (part * 2 + nt) * 16 + i <-> Wqkv row part * C + head * 32 + (i >> 2) * 8 + nt * 4 + (i & 3);
        // 96 + nt * 16 + i <-> COLUMN head * 32 + (i >> 2) * 8 + nt * 4 + (i & 3) of Wproj (row c of Wproj -> LDS column c)
        const T* wq = reinterpret_cast<const T*>(p.wqkv);
        const T* wp = reinterpret_cast<const T*>(p.wproj);
        for (int q = threadIdx.x; q < 96 * RC; q += 128) {
            const int d = q / RC, c = q - d * RC;
            const int part = d >> 5, nt = (d >> 4) & 1, i = d & 15;
            const int sr = part * RC + head * HD + (i >> 2) * 8 + nt * 4 + (i & 3);
            Wh[d * L::WP + c] = wq[(size_t)sr * RC + c];
        }
        for (int q = threadIdx.x; q < 32 * RC; q += 128) {
            const int c = q >> 5, d = q & 31, nt = d >> 4, i = d & 15;
            Wh[(96 + d) * L::WP + c] = wp[(size_t)c * RC + head * HD + (i >> 2) * 8 + nt * 4 + (i & 3)];
        }
        for (int t = threadIdx.x; t < 96; t += 128) bq[t] = p.bqkv ? p.bqkv[(t >> 5) * RC + head * HD + (t & 31)] : 0.f;
    }
    f32x4 dbias[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) dbias[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int iters = (B_ + stride - 1) / stride;
    for (int it = 0; it < iters; ++it) {
        const int b_raw = it * stride + grp;
        const bool wactive = b_raw < B_;
        const int b_ = wactive ? b_raw : B_ - 1;
        const WinPos P = win_pos(p, b_);
        size_t tok[2];
        F qf[2], kf[2], gf[2], vv[2];
        float ls[2], dl[2];
        __syncthreads();
        if constexpr (RC == 0) {
            // the materialised form (fmmt_window_attn_bwd; production: wattn_mfma_bwd_kernel<MM, 2, 0>): this head's q / k / v rows of the
            // qkv tensor and d(attention output) ARE the fragments -- token li of the tile, head dims lg * 8 .. + 7
            const T* qkvg = reinterpret_cast<const T*>(p.qkv);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                tok[a] = tok_of(p, P, own[a].di, own[a].dj);
                const T* row = qkvg + tok[a] * 3 * p.C + head * HD + lg * 8;
                qf[a] = E::ld(row);
                kf[a] = E::ld(row + p.C);
                vv[a] = E::ld(row + 2 * p.C);
                gf[a] = E::ld(dog + tok[a] * p.C + head * HD + lg * 8);
            }
        } else {
            F xf[2][KS], yf[2][KS];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                tok[a] = tok_of(p, P, own[a].di, own[a].dj);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    xf[a][ks] = E::ld(xng + tok[a] * RC + lg * 8 + ks * 32);
                    yf[a][ks] = E::ld(dog + tok[a] * RC + lg * 8 + ks * 32);
                }
            }
            const float rs = p.rowscale ? p.rowscale[P.img] : 1.0f;
            // part 0..2: q, k, v = LN1(x) . W^T + b ; part 3: d(attention output) = rowscale * dy . Wproj[:, head]
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                f32x4 acc[2][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int a = 0; a < 2; ++a) acc[nt][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const F wf = E::ld(Wh + ((part * 2 + nt) * 16 + li) * L::WP + ks * 32 + lg * 8);
#pragma unroll
                        for (int a = 0; a < 2; ++a) acc[nt][a] = E::mma(wf, part == 3 ? yf[a][ks] : xf[a][ks], acc[nt][a]);
                    }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    F f;
                    if (part < 3) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = E::cv(acc[0][a][r] + bq[part * 32 + lg * 8 + r]);
                            f[4 + r] = E::cv(acc[1][a][r] + bq[part * 32 + lg * 8 + 4 + r]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = E::cv(acc[0][a][r] * rs);
                            f[4 + r] = E::cv(acc[1][a][r] * rs);
                        }
                    }
                    if (part == 0) qf[a] = f;
                    else if (part == 1) kf[a] = f;
                    else if (part == 2) vv[a] = f;
                    else gf[a] = f;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int slot = (2 * h + a) * 16 + li;
            const F of = E::ld(og + tok[a] * p.C + head * HD + lg * 8);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += (float)gf[a][e] * (float)of[e];
            dl[a] = xor_sum(d);
            ls[a] = p.lse[((size_t)b_ * p.nH + head) * TOK + (slot < TOK ? slot : TOK - 1)];
            const int off = slot * TP + lg * 8;
            E::st(kt_ + off, own[a].valid ? kf[a] : wr_zero<T>());
            E::st(qt_ + off, own[a].valid ? qf[a] : wr_zero<T>());
            E::st(gt_ + off, own[a].valid ? gf[a] : wr_zero<T>());
            E::st(vt_ + off, own[a].valid ? vv[a] : wr_zero<T>());
            if (lg == 0) {
                Ls[slot] = ls[a];
                Dl[slot] = dl[a];
            }
        }
        __syncthreads();

        // ------------------------------------------------ pass 1: own QUERY tiles -> dQ, d bias
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int qt = 2 * h + a;
            const int q = qt * 16 + li;
            float ds[16];
            const unsigned mb = (MM == 1) ? std_mask_bits(G, P, qt) : 0u;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const F kfk = E::ld(kt_ + (kt * 16 + li) * TP + lg * 8);
                const f32x4 sa = E::mma(kfk, qf[a], f32x4{0.f, 0.f, 0.f, 0.f});
                const F vfk = E::ld(vt_ + (kt * 16 + li) * TP + lg * 8);
                const f32x4 dp = E::mma(vfk, gf[a], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = sa[r] * p.scale + Bs[q * BPM + kt * 16 + lg * 4 + r];
                    if constexpr (MM == 1) s += ((mb >> (kt * 4 + r)) & 1u) ? -100.0f : 0.0f;
                    const float pij = sizeof(T) == 2 ? __expf(s - ls[a]) : expf(s - ls[a]);
                    const float d = pij * (dp[r] - dl[a]);
                    ds[kt * 4 + r] = d;
                    if (wactive) dbias[a][kt][r] += d;
                }
            }
            const F d0 = wr_pack<T>(&ds[0], &ds[4]), d1 = wr_pack<T>(&ds[8], &ds[12]);
            F kT[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) kT[ks][dt] = wr_fragT<T>(kt_, 32 * ks + 4 * lg, dt, li);
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            a0 = E::mma(kT[0][0], d0, a0);
            a0 = E::mma(kT[1][0], d1, a0);
            a1 = E::mma(kT[0][1], d0, a1);
            a1 = E::mma(kT[1][1], d1, a1);
            if (wactive && own[a].valid) {
                F ob;
#pragma unroll
                for (int r = 0; r < 4; ++r) { ob[r] = E::cv(a0[r] * p.scale); ob[4 + r] = E::cv(a1[r] * p.scale); }
                E::st(dqkv + tok[a] * 3 * p.C + head * HD + lg * 8, ob);
            }
        }

        // ------------------------------------------------ pass 2: own KEY tiles -> dK, dV
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int kt = 2 * h + a;
            const int key = kt * 16 + li;
            const F vfk = E::ld(vt_ + (kt * 16 + li) * TP + lg * 8);
            float pp[16], ds[16];
            const unsigned mb = (MM == 1) ? std_mask_bits(G, P, kt) : 0u;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const F qfq = E::ld(qt_ + (qt * 16 + li) * TP + lg * 8);
                const F gfq = E::ld(gt_ + (qt * 16 + li) * TP + lg * 8);
                const f32x4 sa = E::mma(qfq, kf[a], f32x4{0.f, 0.f, 0.f, 0.f});
                const f32x4 dp = E::mma(gfq, vfk, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qt * 16 + lg * 4 + r;
                    float s = sa[r] * p.scale + Bs[q * BPM + key];
                    if constexpr (MM == 1) s += ((mb >> (qt * 4 + r)) & 1u) ? -100.0f : 0.0f;
                    const float pij = sizeof(T) == 2 ? __expf(s - Ls[q]) : expf(s - Ls[q]);
                    pp[qt * 4 + r] = pij;
                    ds[qt * 4 + r] = pij * (dp[r] - Dl[q]);
                }
            }
            const F p0 = wr_pack<T>(&pp[0], &pp[4]), p1 = wr_pack<T>(&pp[8], &pp[12]);
            const F d0 = wr_pack<T>(&ds[0], &ds[4]), d1 = wr_pack<T>(&ds[8], &ds[12]);
            F gT[2][2], qT[2][2];
#pragma unroll
            for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    gT[qs][dt] = wr_fragT<T>(gt_, 32 * qs + 4 * lg, dt, li);
                    qT[qs][dt] = wr_fragT<T>(qt_, 32 * qs + 4 * lg, dt, li);
                }
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f}, k0 = {0.f, 0.f, 0.f, 0.f}, k1 = {0.f, 0.f, 0.f, 0.f};
            v0 = E::mma(gT[0][0], p0, v0);
            v0 = E::mma(gT[1][0], p1, v0);
            v1 = E::mma(gT[0][1], p0, v1);
            v1 = E::mma(gT[1][1], p1, v1);
            k0 = E::mma(qT[0][0], d0, k0);
            k0 = E::mma(qT[1][0], d1, k0);
            k1 = E::mma(qT[0][1], d0, k1);
            k1 = E::mma(qT[1][1], d1, k1);
            if (wactive && own[a].valid) {
                F kb, vb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    kb[r] = E::cv(k0[r] * p.scale); kb[4 + r] = E::cv(k1[r] * p.scale);
                    vb[r] = E::cv(v0[r]); vb[4 + r] = E::cv(v1[r]);
                }
                T* dst = dqkv + tok[a] * 3 * p.C + head * HD + lg * 8;
                E::st(dst + p.C, kb);
                E::st(dst + 2 * p.C, vb);
            }
        }
    }
    // d(bias) of the workgroup: the two waves write their own query rows (wave h holds rows of tiles 2h, 2h + 1)
    __syncthreads();
    float* acc = reinterpret_cast<float*>(kt_);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int q = (2 * h + a) * 16 + li;
        if (q >= TOK) continue;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + lg * 4 + r;
                if (key < TOK) acc[q * TOK + key] = dbias[a][kt][r];
            }
    }
    __syncthreads();
    float* part = p.part + ((size_t)head * p.groups_per_head + grp) * TOK * TOK;
    for (int t = threadIdx.x; t < TOK * TOK; t += 128) part[t] = acc[t];
}

template <typename T, int MM, int RC>
int wr_launch(const WaArgs& a, int grid, hipStream_t st) {
    constexpr int lds = WrLds<T, RC>::TOTAL;
    static_assert(lds <= 160 * 1024, "LDS");
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&wattn_bwd_ref_kernel<T, MM, RC>), lds)) return rc_;
    hipLaunchKernelGGL((wattn_bwd_ref_kernel<T, MM, RC>), dim3(grid), dim3(128), lds, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// generic restatement of the backward; head-major grid of num_heads * a.groups_per_head workgroups (one pair each).
//   a.xn != nullptr   the recompute form behind fmmt_window_block_attn_bwd, C = 96
//   a.xn == nullptr   the materialised form behind fmmt_window_attn_bwd (q / k / v / d(out) read from a.qkv / a.dout), any width of 32-wide
//                     heads; mask none or the standard SW-MSA mask (an explicit mask tensor: FMMT_EINVAL, the caller keeps its other path)
int fmmt_wattn_bwd_ref_launch(int dtype, const WaArgs& a, int grid, hipStream_t st) {
    if (!a.xn) {
        if (a.mask && !a.mask_is_shift) return FMMT_EINVAL;
        const bool masked = a.mask != nullptr && a.shift > 0;
        if (dtype == FMMT_F32) return masked ? wr_launch<float, 1, 0>(a, grid, st) : wr_launch<float, 0, 0>(a, grid, st);
        return masked ? wr_launch<bf16, 1, 0>(a, grid, st) : wr_launch<bf16, 0, 0>(a, grid, st);
    }
    const bool masked = a.shift > 0;
    if (a.C != 96) return FMMT_EINVAL;
    if (dtype == FMMT_F32) return masked ? wr_launch<float, 1, 96>(a, grid, st) : wr_launch<float, 0, 96>(a, grid, st);
    return masked ? wr_launch<bf16, 1, 96>(a, grid, st) : wr_launch<bf16, 0, 96>(a, grid, st);
}
