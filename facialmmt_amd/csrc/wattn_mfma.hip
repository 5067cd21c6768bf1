// (Shifted-)window attention core on the matrix cores -- bf16 throughput path (gfx950).
//
// One wave owns one (window, head) problem: 49 tokens (padded to 64) x head_dim 32.
//   S^T = K . Q^T      16 x mfma_f32_16x16x32_bf16 (K, Q fragments are 16-byte loads straight from the
//                      token-order qkv matrix: an MFMA fragment row IS 8 contiguous channels of one token)
//   softmax            computed on S^T: a lane holds ONE query column (its 16 keys per query tile), so the
//                      row max / sum need only two cross-lane steps (xor 16, xor 32)
//   O^T = V^T . P^T    16 x MFMA; P^T is consumed from the registers it was produced in (the k-slot order of
//                      the contraction is free, so it is chosen to match the accumulator layout); V^T comes
//                      from a natural-layout LDS tile through the LDS transpose read (ds_read_b64_tr_b16),
//                      with the output-channel permutation that makes every lane store 8 contiguous channels.
// roll / window_partition / window_reverse are address arithmetic, as in attn.hip.  A workgroup (4 waves)
// is pinned to one head and walks windows, so that head's dense 49x49 bias sits in LDS.
//
// Backward recomputes P from the saved log-sum-exp twice: once with lanes owning query columns (dQ, d bias
// accumulated in registers across windows) and once with lanes owning key columns (dK, dV); two waves share a
// (window, head) problem, each owning half of the token tiles in both passes.  d(bias table) is reduced
// through per-workgroup partials in a fixed order.
#include <type_traits>
#include "fmmt_common.h"
#include "../../include/fmmt.h"
#include "wattn_args.h"
#include "wattn_geom.h"

namespace {

constexpr float WA_LOG2E = 1.4426950408889634f;

// mul = log2(e) for the kernels whose softmax runs on base-2 exponentials (v_exp_f32 is base 2: one multiply per logit less)
__device__ __forceinline__ void fill_bias_mfma(const WaArgs& p, int head, float* Bs, int nthreads = 256, float mul = 1.0f) {
    for (int t = threadIdx.x; t < 64 * BPM; t += nthreads) {
        const int q = t / BPM, k = t - q * BPM;
        Bs[t] = (q < TOK && k < TOK) ? p.table[p.index[q * TOK + k] * p.nH + head] * mul : NEG_BIG;
    }
}

// blockIdx -> (head, window group).  The hardware places block b on XCD b % 8, and each XCD has its own L2.  A head
// slice of a token is 64 B, half an L2 line, so the heads of one window group should run on the SAME XCD at about the
// same time: with groups_per_head a multiple of 8 (the launcher rounds it), blocks are numbered
// b = (grp / 8) * 8 * nH + head * 8 + grp % 8, i.e. b % 8 == grp % 8 for every head.  (FMMT_WA_XCD=0: head-major.)
__device__ __forceinline__ void head_group_of_block(const WaArgs& p, int& head, int& grp) {
    const int b = blockIdx.x;
    if (p.xcd_grouped) {
        const int band = 8 * p.nH, q = b / band, o = b - q * band;
        head = o >> 3;
        grp = q * 8 + (o & 7);
    } else {
        head = b / p.groups_per_head;
        grp = b - head * p.groups_per_head;
    }
}

// =============================================================================================
// MM: 0 = no mask, 1 = standard SW-MSA mask derived from window coordinates, 2 = arbitrary (nW,49,49) tensor
template <int MM>
__global__ __launch_bounds__(256) void wattn_mfma_fwd_kernel(WaArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 Vt[4][64 * TP];
    __shared__ __attribute__((aligned(16))) float Bs[64 * BPM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    int head, grp;
    head_group_of_block(p, head, grp);
    const int nW = (p.H / WS) * (p.W / WS), B_ = p.n_img * nW;
    const int stride = p.groups_per_head * 4;
    const bf16* __restrict__ qkv = reinterpret_cast<const bf16*>(p.qkv);
    bf16* __restrict__ out = reinterpret_cast<bf16*>(p.out);
    const LaneGeom G = lane_geom(li, lg, p.shift);
    bf16* vt = Vt[wave];

    fill_bias_mfma(p, head, Bs);
    const int iters = (B_ + stride - 1) / stride;
    for (int it = 0; it < iters; ++it) {
        const int b_raw = it * stride + grp * 4 + wave;
        const bool wactive = b_raw < B_;
        const int b_ = wactive ? b_raw : B_ - 1;
        const WinPos P = win_pos(p, b_);
        size_t tok[4];
        bf16x8 qf[4], kf[4];
        __syncthreads();                                   // previous iteration's reads of Vt are done
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            tok[t] = tok_of(p, P, G.di[t], G.dj[t]);
            const bf16* row = qkv + tok[t] * 3 * p.C + head * HD + lg * 8;
            qf[t] = ld_frag(row);
            kf[t] = ld_frag(row + p.C);
            const bf16x8 vv = G.valid[t] ? ld_frag(row + 2 * p.C) : zero_frag();
            *reinterpret_cast<bf16x8*>(vt + (t * 16 + li) * TP + lg * 8) = vv;
        }
        __syncthreads();
        bf16x8 vT[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vT[ks][dt] = tr_fragT(vt, 32 * ks + 4 * lg, dt, li);

#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const int q = qt * 16 + li;
            float s[16];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[q * BPM + kt * 16 + lg * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt * 4 + r] = a[r] * p.scale + b[r];
            }
            if constexpr (MM == 1) {
                const unsigned mb = std_mask_bits(G, P, qt);
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] += ((mb >> e) & 1u) ? -100.0f : 0.0f;
            } else if constexpr (MM == 2) {
                const float* mrow = p.mask + ((size_t)(b_ % p.nW_mask) * TOK + (q < TOK ? q : TOK - 1)) * TOK;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 16 + lg * 4 + r;
                        s[kt * 4 + r] += mrow[key < TOK ? key : TOK - 1];      // clamped, branch-free (pad keys are -1e30 already)
                    }
            }
            float m = s[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) m = fmaxf(m, s[e]);
            m = xor_max(m);
            float l = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = __expf(s[e] - m);
                l += s[e];
            }
            l = xor_sum(l);
            const float inv = 1.0f / l;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] *= inv;
            const bf16x8 pb0 = pack8(&s[0], &s[4]), pb1 = pack8(&s[8], &s[12]);
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
            o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT[0][0], pb0, o0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT[1][0], pb1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT[0][1], pb0, o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT[1][1], pb1, o1, 0, 0, 0);
            if (wactive && G.valid[qt]) {
                bf16x8 ob;
#pragma unroll
                for (int r = 0; r < 4; ++r) { ob[r] = (bf16)o0[r]; ob[4 + r] = (bf16)o1[r]; }
                *reinterpret_cast<bf16x8*>(out + tok[qt] * p.C + head * HD + lg * 8) = ob;
                if (lg == 0) p.lse[((size_t)b_ * p.nH + head) * TOK + q] = m + __logf(l);
            }
        }
    }
}

// =============================================================================================
// Backward, two waves per (window, head).  A one-wave-per-window formulation (measured: 6.5 ms/step against 5.3)
// needs 344-412 registers (64 for d(bias), 64 for the q/k/v/dO fragments of all four token tiles, ...), i.e. one
// wave per SIMD and nothing to hide the global-load -> LDS -> MFMA -> store chain behind; forcing it to 256
// registers spills and is 2x slower.  Here wave h of a pair owns token tiles {2h, 2h+1}: as
// QUERY tiles in pass 1 (dQ, d(bias): 2 x 4 accumulator tiles instead of 4 x 4) and as KEY tiles in pass 2
// (dK, dV).  Fragments of the partner's tiles come from the pair's natural-layout LDS tiles (K, Q, dO,
// V).  A workgroup (2 pairs) walks two windows per iteration with 57 KB of LDS: two workgroups per CU, two
// waves per SIMD.
// =============================================================================================
struct Slot { int di, dj; bool valid; };
__device__ __forceinline__ Slot slot_of(int slot) {
    Slot S;
    S.valid = slot < TOK;
    const int cs = S.valid ? slot : TOK - 1;
    S.di = cs / WS;
    S.dj = cs - S.di * WS;
    return S;
}

// NP pairs of waves per workgroup (2: the 256-thread kernel of fmmt_window_attn_bwd; 4: the recompute variant).
// RC = 0: q, k, v and d(attention output) are read from memory (qkv, dout).  RC = C (96 / 192), the recompute variant behind
// fmmt_window_block_attn_bwd: nothing qkv-sized is read -- a wave loads LN1(x) and the block-output gradient of its two token tiles
// (C channels each) and forms its q / k / v fragments with the head's 96 rows of Wqkv and its d(attention output) fragments with the
// head's 32 columns of Wproj, both staged once per workgroup in LDS in fragment order: 48 MFMAs per wave and (window, head) at C = 96
// in exchange for 1.5 GB less to read per stage-0 launch, and neither the qkv recomputation GEMM nor the proj input-gradient GEMM run.
template <int NP, int RC>
struct WaBwdLds {
    static constexpr int TILE = 64 * TP * 2;                                            // bytes of one [64][TP] bf16 tile
    static constexpr int TILES = 4 * NP * TILE, STATS = 2 * NP * 64 * 4, BIAS = 64 * BPM * 4;
    static constexpr int WP = RC + 8;                                                   // weight row pitch (bf16)
    static constexpr int WGT = RC ? 128 * WP * 2 + 96 * 4 : 0;
    static constexpr int TOTAL = TILES + STATS + BIAS + WGT;
};

template <int MM, int NP, int RC>
__global__ __launch_bounds__(NP * 128) __attribute__((amdgpu_waves_per_eu(2)))
void wattn_mfma_bwd_kernel(WaArgs p) {
    using L = WaBwdLds<NP, RC>;
    constexpr int NT = NP * 128;
    extern __shared__ __attribute__((aligned(16))) char smem_bwd[];
    bf16 (*Kt)[64 * TP] = reinterpret_cast<bf16 (*)[64 * TP]>(smem_bwd);
    bf16 (*Qt)[64 * TP] = Kt + NP;
    bf16 (*Gt)[64 * TP] = Qt + NP;
    bf16 (*Vt)[64 * TP] = Gt + NP;
    float (*Ls)[64] = reinterpret_cast<float (*)[64]>(smem_bwd + L::TILES);
    float (*Dl)[64] = Ls + NP;
    float* Bs = reinterpret_cast<float*>(smem_bwd + L::TILES + L::STATS);
    bf16* Wh = reinterpret_cast<bf16*>(smem_bwd + L::TILES + L::STATS + L::BIAS);      // RC: 96 rows of Wqkv + 32 "rows" of Wproj^T
    float* bq = reinterpret_cast<float*>(Wh + 128 * L::WP);                              // RC: the head's q | k | v bias (96 values)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = wave >> 1, h = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    int head, grp;
    head_group_of_block(p, head, grp);
    const int nW = (p.H / WS) * (p.W / WS), B_ = p.n_img * nW;
    const int stride = p.groups_per_head * NP;
    const bf16* __restrict__ qkv = reinterpret_cast<const bf16*>(p.qkv);
    const bf16* __restrict__ og = reinterpret_cast<const bf16*>(p.out);
    const bf16* __restrict__ dog = reinterpret_cast<const bf16*>(p.dout);
    bf16* __restrict__ dqkv = reinterpret_cast<bf16*>(p.dqkv);
    const LaneGeom G = lane_geom(li, lg, p.shift);
    bf16 *kt_ = Kt[pair], *qt_ = Qt[pair], *gt_ = Gt[pair], *vt_ = Vt[pair];
    bf16* const sinkp = reinterpret_cast<bf16*>(p.sink) + threadIdx.x * 8;
    const Slot own[2] = {slot_of((2 * h) * 16 + li), slot_of((2 * h + 1) * 16 + li)};

    fill_bias_mfma(p, head, Bs, NT, WA_LOG2E);              // logits in base 2: bias, scale, mask constant and log-sum-exp all carry log2(e)
    const float sc2 = p.scale * WA_LOG2E;
    if constexpr (RC != 0) {
        // rows in FRAGMENT order: (part * 2 + nt) * 16 + i <-> Wqkv row part * C + head * 32 + (i >> 2) * 8 + nt * 4 + (i & 3);
        // 96 + nt * 16 + i <-> COLUMN head * 32 + (i >> 2) * 8 + nt * 4 + (i & 3) of Wproj (row c of Wproj -> LDS column c)
        const bf16* wq = reinterpret_cast<const bf16*>(p.wqkv);
        const bf16* wp = reinterpret_cast<const bf16*>(p.wproj);
        for (int q = threadIdx.x; q < 96 * (RC / 8); q += NT) {
            const int d = q / (RC / 8), ch = q - d * (RC / 8);
            const int part = d >> 5, nt = (d >> 4) & 1, i = d & 15;
            const int sr = part * RC + head * HD + (i >> 2) * 8 + nt * 4 + (i & 3);
            *reinterpret_cast<bf16x8*>(Wh + d * L::WP + ch * 8) = *reinterpret_cast<const bf16x8*>(wq + (size_t)sr * RC + ch * 8);
        }
        for (int q = threadIdx.x; q < 32 * RC; q += NT) {
            const int c = q >> 5, d = q & 31, nt = d >> 4, i = d & 15;
            Wh[(96 + d) * L::WP + c] = wp[(size_t)c * RC + head * HD + (i >> 2) * 8 + nt * 4 + (i & 3)];
        }
        for (int t = threadIdx.x; t < 96; t += NT) bq[t] = p.bqkv ? p.bqkv[(t >> 5) * RC + head * HD + (t & 31)] : 0.f;
    }
    f32x4 dbias[2][4];                               // [own query tile][key tile], lane = query column layout
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) dbias[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int iters = (B_ + stride - 1) / stride;
    // The (window, head) problem of iteration it + 1 is requested while iteration it is being worked on (RC == 0: its q / k / v /
    // d(out) / out fragments and log-sum-exp, 42 registers; RC: the LN1(x) / dy / out rows of the wave's tokens, 58 at C = 96, in
    // flight under the two passes) -- an iteration is otherwise a chain of HBM round trip -> LDS -> barrier -> 56 MFMAs -> stores
    // on two workgroups per CU with nothing to cover the round trip.
    constexpr int FKS = RC ? RC / 32 : 1;
    struct Fetch {
        bf16x8 q[2], k[2], v[2], g[2];                     // RC == 0
        bf16x8 x[2][FKS], y[2][FKS];                       // RC != 0: LN1(x) and dy rows of the two own tokens
        bf16x8 o[2];
        float ls[2], rs;
        int tok[2];                                        // token index (n_img * H * W < 2^31)
    };
    auto fetch = [&](int it, Fetch& F) {
        const int b_raw = it * stride + grp * NP + pair;
        const int b_ = b_raw < B_ ? b_raw : B_ - 1;
        const WinPos P = win_pos(p, b_);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            F.tok[a] = (int)tok_of(p, P, own[a].di, own[a].dj);
            if constexpr (RC == 0) {
                const bf16* row = qkv + (size_t)F.tok[a] * 3 * p.C + head * HD + lg * 8;
                F.q[a] = ld_frag(row);
                F.k[a] = ld_frag(row + p.C);
                F.v[a] = ld_frag(row + 2 * p.C);
                F.g[a] = ld_frag(dog + (size_t)F.tok[a] * p.C + head * HD + lg * 8);
            } else {
                const bf16* xr = reinterpret_cast<const bf16*>(p.xn) + (size_t)F.tok[a] * RC + lg * 8;
                const bf16* yr = dog + (size_t)F.tok[a] * RC + lg * 8;
#pragma unroll
                for (int ks = 0; ks < FKS; ++ks) {
                    F.x[a][ks] = ld_frag(xr + ks * 32);
                    F.y[a][ks] = ld_frag(yr + ks * 32);
                }
            }
            F.o[a] = ld_frag(og + (size_t)F.tok[a] * p.C + head * HD + lg * 8);
            const int slot = (2 * h + a) * 16 + li;
            F.ls[a] = p.lse[((size_t)b_ * p.nH + head) * TOK + (slot < TOK ? slot : TOK - 1)];
        }
        if constexpr (RC != 0) F.rs = p.rowscale ? p.rowscale[P.img] : 1.0f;
    };
    Fetch cur;
    constexpr int NLD = RC ? 2 * (2 * FKS + 1) : 10;       // 16-byte loads of a fetch (plus the scalars)
    {
        // Every path into the loop head carries the same instruction counts behind the fetch (12 loads, then 6 stores): the compiler's
        // vmcnt bookkeeping merges the loop's entry and its back edge, and an entry without the six stores turns the head's
        // "fragments have landed" into vmcnt(0) -- the previous problem's stores waited out on every iteration.  Hence six sink
        // stores here, and an unconditional (clamped: problem B_ - 1 again) fetch in the loop.
        fetch(0, cur);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            *reinterpret_cast<bf16x8*>(sinkp) = zero_frag();
            asm volatile("" ::: "memory");                 // six store instructions, not one
        }
    }
    for (int it = 0; it < iters; ++it) {
        const int b_raw = it * stride + grp * NP + pair;
        const bool wactive = b_raw < B_;
        const int b_ = wactive ? b_raw : B_ - 1;
        const WinPos P = win_pos(p, b_);
        const float* mbase = (MM == 2) ? p.mask + (size_t)(b_ % p.nW_mask) * TOK * TOK : nullptr;
        int tok[2];
        bf16x8 qf[2], kf[2], gf[2];
        float ls[2], dl[2];
        __syncthreads();                                   // previous iteration finished with the LDS tiles
        bf16x8 vv[2];
        if constexpr (RC == 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                tok[a] = cur.tok[a];
                qf[a] = cur.q[a];
                kf[a] = cur.k[a];
                vv[a] = cur.v[a];
                gf[a] = cur.g[a];
            }
        } else {
            constexpr int KS = RC / 32;
            bf16x8 xf[2][KS], yf[2][KS];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                tok[a] = cur.tok[a];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    xf[a][ks] = cur.x[a][ks];
                    yf[a][ks] = cur.y[a][ks];
                }
            }
            const float rs = cur.rs;
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
                        const bf16x8 wf = ld_frag(Wh + ((part * 2 + nt) * 16 + li) * L::WP + ks * 32 + lg * 8);
#pragma unroll
                        for (int a = 0; a < 2; ++a) acc[nt][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, part == 3 ? yf[a][ks] : xf[a][ks], acc[nt][a], 0, 0, 0);
                    }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    bf16x8 f;
                    if (part < 3) {
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bq + part * 32 + lg * 8), b1 = *reinterpret_cast<const f32x4*>(bq + part * 32 + lg * 8 + 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = (bf16)(acc[0][a][r] + b0[r]);
                            f[4 + r] = (bf16)(acc[1][a][r] + b1[r]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = (bf16)(acc[0][a][r] * rs);
                            f[4 + r] = (bf16)(acc[1][a][r] * rs);
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
            const bf16x8 of = cur.o[a];
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += (float)gf[a][e] * (float)of[e];
            // (Round 6, measured and not kept: the row statistics through the MFMAs' C operand -- S' = K.Q^T - lse / scale, dP' = V.dO^T - delta -- instead of
            //  one v_sub per logit each, 2 of ~12 VALU slots: stage 0 / 1 backward 0.965 / 0.993 / 0.484 -> 0.951 / 0.937 / 0.470 ms, stage 2 0.256 -> 0.34,
            //  Swin forward + backward 41.2 -> 41.1 ms, the step 58.3 -> 58.3: nothing.  A two-instruction mask (AND against a sign-extended bit field instead
            //  of test + compare + select) spilled 15-44 registers: the sixteen masks of a tile are formed early.  And P handed from pass 1 to pass 2 through a
            //  64 x 64 bf16 tile of the pair's in LDS -- written as four keys per lane, read back transposed by ds_read_b64_tr_b16 as the four query rows of the lane's key
            //  column -- instead of recomputed (per lane 32 logits, 8 MFMAs, 32 bias reads and the conversions of P^T's fragments less, one barrier more): correct, twin
            //  and goldens green, stage 0 / 1 sizes 0.936 / 0.954 / 0.463 / 0.488 -> 0.889 / 0.881 / 0.447 / 0.461 ms, stage 2 / 3 unchanged (0.251 / 0.266 / 0.180 ->
            //  0.257 / 0.267 / 0.190), Swin 40.1 -> 40.1 ms: a quarter of the VALU slots gone and 3 % of the time -- the kernels are bound by the latency of their
            //  dependent LDS -> MFMA -> exponential chains at two waves per SIMD, not by issue slots.  profiles/r06_wattn_fold.txt)
            dl[a] = xor_sum(d);
            ls[a] = cur.ls[a] * WA_LOG2E;
            const int off = slot * TP + lg * 8;
            // pad slots (49..63) hold zeros, in the LDS tiles AND in the register copies the passes below use for the wave's own tiles
            kf[a] = own[a].valid ? kf[a] : zero_frag();
            qf[a] = own[a].valid ? qf[a] : zero_frag();
            gf[a] = own[a].valid ? gf[a] : zero_frag();
            vv[a] = own[a].valid ? vv[a] : zero_frag();
            *reinterpret_cast<bf16x8*>(kt_ + off) = kf[a];
            *reinterpret_cast<bf16x8*>(qt_ + off) = qf[a];
            *reinterpret_cast<bf16x8*>(gt_ + off) = gf[a];
            *reinterpret_cast<bf16x8*>(vt_ + off) = vv[a];
            if (lg == 0) {
                Ls[pair][slot] = ls[a];
                Dl[pair][slot] = dl[a];
            }
        }
        __syncthreads();
        fetch(it + 1, cur);                                // in flight under both passes; consumed at the head of the next iteration

        // Both passes need the natural-layout fragments of all four token tiles of K / V (pass 1) and Q / d(out) (pass 2).  The wave's own
        // two tiles are in its registers; only the PARTNER's two come from the pair's LDS tiles, read once per pass for both own tiles
        // (round 3 read all four from LDS for each own tile: 34 ds_read_b128 and their waits per problem, now 8).  Register arrays
        // want compile-time indices, so the tiles are walked in the order own, own, partner, partner (t' = 0..3 <-> tile
        // t' < 2 ? 2h + t' : 2(1 - h) + t' - 2): the contraction order of a 64-key (64-query) sum is free, and everything that depends
        // on the actual tile index -- bias / statistics addresses, mask-bit shifts, the transposed fragments' row blocks, where d(bias)
        // goes at the end -- is an LDS address or a shift amount, which may be run-time values.  (RC != 0, the recompute variant: its
        // prefetch already holds 58 registers in flight and the 24 more this takes spill; it keeps reading all four tiles from LDS.)
        // OWNREG: the own-first walk with the own tiles from registers; HOIST: the partner's fragments read once per pass and kept for both own
        // tiles (16 registers more: the recompute variant at C = 96 has no room for them and re-reads them per own tile -- 34 -> 16 reads)
        constexpr bool OWNREG = (RC == 0 || RC == 96) && MM != 2;   // (MM == 2, an arbitrary mask tensor: test-only path, already over the register budget)
        constexpr bool HOIST = OWNREG && RC == 0;
        const int tile_of[4] = {2 * h, 2 * h + 1, 2 * (1 - h), 2 * (1 - h) + 1};
        // ------------------------------------------------ pass 1: own QUERY tiles -> dQ, d bias
        {
            bf16x8 pk[2], pv[2];
            if constexpr (HOIST) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    pk[j] = ld_frag(kt_ + (tile_of[2 + j] * 16 + li) * TP + lg * 8);
                    pv[j] = ld_frag(vt_ + (tile_of[2 + j] * 16 + li) * TP + lg * 8);
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int qt = 2 * h + a;
                const int q = qt * 16 + li;
                float ds[8];
                bf16x8 dpk[2];
                const unsigned mb = (MM == 1) ? std_mask_bits(G, P, qt) : 0u;
#pragma unroll
                for (int ktp = 0; ktp < 4; ++ktp) {
                    const int kt = OWNREG ? tile_of[ktp] : ktp;              // actual key tile (run time where tiles are walked own-first)
                    const bf16x8 kfk = OWNREG && ktp < 2 ? kf[ktp & 1] : HOIST ? pk[ktp & 1] : ld_frag(kt_ + (kt * 16 + li) * TP + lg * 8);
                    const f32x4 sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfk, qf[a], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const bf16x8 vfk = OWNREG && ktp < 2 ? vv[ktp & 1] : HOIST ? pv[ktp & 1] : ld_frag(vt_ + (kt * 16 + li) * TP + lg * 8);
                    const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfk, gf[a], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[q * BPM + kt * 16 + lg * 4]);
                    const unsigned mbk = mb >> (kt * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float s = sa[r] * sc2 + b[r];
                        if constexpr (MM == 2) {
                            const int key = kt * 16 + lg * 4 + r;
                            s += mbase[(q < TOK ? q : TOK - 1) * TOK + (key < TOK ? key : TOK - 1)] * WA_LOG2E;
                        }
                        float pij = __builtin_amdgcn_exp2f(s - ls[a]);
                        // SW-MSA mask (Swin_Transformer.py:222 adds -100 to the logit): exp(s - 100 - lse) < 4e-44 x P is zero in every bf16 operand it
                        // reaches, so the probability is zeroed instead -- a select BEHIND the exponential, which the scheduler cannot hoist
                        // (the added form let it form all 16 addends of a tile early: 25 more live registers)
                        if constexpr (MM == 1) pij = ((mbk >> r) & 1u) ? 0.0f : pij;
                        const float d = pij * (dp[r] - dl[a]);
                        ds[(ktp & 1) * 4 + r] = d;
                        if (wactive) dbias[a][ktp][r] += d;                   // OWNREG: indexed by walk position, un-permuted at the end
                    }
                    if (ktp & 1) dpk[ktp >> 1] = pack8(&ds[0], &ds[4]);      // packed as soon as a k-slot block is complete (registers)
                }
                const bf16x8 d0 = dpk[0], d1 = dpk[1];
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 kT[2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)                      // k-slot block ks of d0 / d1 = walk positions 2 ks, 2 ks + 1 = row block tile_of[2 ks] / 2
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) kT[ks][dt] = tr_fragT(kt_, 32 * (OWNREG ? tile_of[2 * ks] >> 1 : ks) + 4 * lg, dt, li);
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[0][0], d0, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[1][0], d1, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[0][1], d0, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[1][1], d1, a1, 0, 0, 0);
                {
                    bf16x8 ob;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ob[r] = (bf16)(a0[r] * p.scale); ob[4 + r] = (bf16)(a1[r] * p.scale); }
                    *reinterpret_cast<bf16x8*>((wactive && own[a].valid) ? dqkv + (size_t)tok[a] * 3 * p.C + head * HD + lg * 8 : sinkp) = ob;    // WaArgs::sink
                }
                __builtin_amdgcn_sched_barrier(0);          // one query tile at a time: interleaved, the two want 30+ more registers
            }
        }

        // ------------------------------------------------ pass 2: own KEY tiles -> dK, dV
        {
            bf16x8 pq[2], pg[2];
            if constexpr (HOIST) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    pq[j] = ld_frag(qt_ + (tile_of[2 + j] * 16 + li) * TP + lg * 8);
                    pg[j] = ld_frag(gt_ + (tile_of[2 + j] * 16 + li) * TP + lg * 8);
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int kt = 2 * h + a;
                const int key = kt * 16 + li;
                const bf16x8 vfk = OWNREG ? vv[a] : ld_frag(vt_ + (kt * 16 + li) * TP + lg * 8);
                float pp[8], ds[8];
                bf16x8 ppk[2], dpk[2];
                const unsigned mb = (MM == 1) ? std_mask_bits(G, P, kt) : 0u;
#pragma unroll
                for (int qtp = 0; qtp < 4; ++qtp) {
                    const int qt = OWNREG ? tile_of[qtp] : qtp;
                    const bf16x8 qfq = OWNREG && qtp < 2 ? qf[qtp & 1] : HOIST ? pq[qtp & 1] : ld_frag(qt_ + (qt * 16 + li) * TP + lg * 8);
                    const bf16x8 gfq = OWNREG && qtp < 2 ? gf[qtp & 1] : HOIST ? pg[qtp & 1] : ld_frag(gt_ + (qt * 16 + li) * TP + lg * 8);
                    const f32x4 sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfq, kf[a], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gfq, vfk, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const f32x4 lq = *reinterpret_cast<const f32x4*>(&Ls[pair][qt * 16 + lg * 4]);
                    const f32x4 dq = *reinterpret_cast<const f32x4*>(&Dl[pair][qt * 16 + lg * 4]);
                    const unsigned mbq = mb >> (qt * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = qt * 16 + lg * 4 + r;
                        float s = sa[r] * sc2 + Bs[q * BPM + key];
                        if constexpr (MM == 2) s += mbase[(q < TOK ? q : TOK - 1) * TOK + (key < TOK ? key : TOK - 1)] * WA_LOG2E;
                        float pij = __builtin_amdgcn_exp2f(s - lq[r]);
                        if constexpr (MM == 1) pij = ((mbq >> r) & 1u) ? 0.0f : pij;
                        pp[(qtp & 1) * 4 + r] = pij;
                        ds[(qtp & 1) * 4 + r] = pij * (dp[r] - dq[r]);
                    }
                    if (qtp & 1) {
                        ppk[qtp >> 1] = pack8(&pp[0], &pp[4]);
                        dpk[qtp >> 1] = pack8(&ds[0], &ds[4]);
                    }
                }
                const bf16x8 p0 = ppk[0], p1 = ppk[1], d0 = dpk[0], d1 = dpk[1];
                // transposed fragments read per key tile, behind the logits (32 registers that need not be live under them)
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 gT[2][2], qT[2][2];
#pragma unroll
                for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int rb = OWNREG ? tile_of[2 * qs] >> 1 : qs;     // row block of k-slot block qs (see pass 1)
                        gT[qs][dt] = tr_fragT(gt_, 32 * rb + 4 * lg, dt, li);
                        qT[qs][dt] = tr_fragT(qt_, 32 * rb + 4 * lg, dt, li);
                    }
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f}, k0 = {0.f, 0.f, 0.f, 0.f}, k1 = {0.f, 0.f, 0.f, 0.f};
                v0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gT[0][0], p0, v0, 0, 0, 0);
                v0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gT[1][0], p1, v0, 0, 0, 0);
                v1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gT[0][1], p0, v1, 0, 0, 0);
                v1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gT[1][1], p1, v1, 0, 0, 0);
                k0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[0][0], d0, k0, 0, 0, 0);
                k0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[1][0], d1, k0, 0, 0, 0);
                k1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[0][1], d0, k1, 0, 0, 0);
                k1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[1][1], d1, k1, 0, 0, 0);
                {
                    bf16x8 kb, vb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        kb[r] = (bf16)(k0[r] * p.scale); kb[4 + r] = (bf16)(k1[r] * p.scale);
                        vb[r] = (bf16)v0[r]; vb[4 + r] = (bf16)v1[r];
                    }
                    const bool live = wactive && own[a].valid;
                    bf16* dst = dqkv + (size_t)tok[a] * 3 * p.C + head * HD + lg * 8;
                    *reinterpret_cast<bf16x8*>(live ? dst + p.C : sinkp) = kb;
                    *reinterpret_cast<bf16x8*>(live ? dst + 2 * p.C : sinkp) = vb;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // d(bias) of the workgroup: waves add their register accumulators into one LDS tile in wave order (deterministic).
    // Wave w holds query rows of half (w & 1); waves 0 and 1 write their rows first, waves 2 and 3 add to them.
    float* acc = reinterpret_cast<float*>(&Kt[0][0]);           // Kt: 2 x 5 KB >= 49*49 floats
    for (int w = 0; w < 2 * NP; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int q = (2 * h + a) * 16 + li;
                if (q >= TOK) continue;
#pragma unroll
                for (int ktp = 0; ktp < 4; ++ktp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kt = ((RC == 0 || RC == 96) && MM != 2) ? (ktp < 2 ? 2 * h + ktp : 2 * (1 - h) + ktp - 2) : ktp;     // walk position -> key tile (pass 1)
                        const int key = kt * 16 + lg * 4 + r;
                        if (key < TOK) {
                            const float v = dbias[a][ktp][r];
                            acc[q * TOK + key] = (w < 2) ? v : acc[q * TOK + key] + v;
                        }
                    }
            }
        }
    }
    __syncthreads();
    float* part = p.part + ((size_t)head * p.groups_per_head + grp) * TOK * TOK;
    for (int t = threadIdx.x; t < TOK * TOK; t += NT) part[t] = acc[t];
}

}  // namespace

int fmmt_wattn_mfma_fwd_launch(const WaArgs& a, int grid, hipStream_t st) {
    const int mm = !a.mask ? 0 : (a.mask_is_shift ? 1 : 2);
    if (mm == 0) hipLaunchKernelGGL(wattn_mfma_fwd_kernel<0>, dim3(grid), dim3(256), 0, st, a);
    else if (mm == 1) hipLaunchKernelGGL(wattn_mfma_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wattn_mfma_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

template <int MM, int NP, int RC>
static int launch_bwd(const WaArgs& a, int grid, hipStream_t st) {
    constexpr int lds = WaBwdLds<NP, RC>::TOTAL;
    static FmmtLdsOnce lds_once;
    if (lds > 48 * 1024) {
        if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&wattn_mfma_bwd_kernel<MM, NP, RC>), lds)) return rc_;
    }
    hipLaunchKernelGGL((wattn_mfma_bwd_kernel<MM, NP, RC>), dim3(grid), dim3(NP * 128), lds, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

int fmmt_wattn_mfma_bwd_launch(const WaArgs& a, int grid, hipStream_t st) {
    const int mm = !a.mask ? 0 : (a.mask_is_shift ? 1 : 2);
    if (mm == 0) return launch_bwd<0, 2, 0>(a, grid, st);
    if (mm == 1) return launch_bwd<1, 2, 0>(a, grid, st);
    return launch_bwd<2, 2, 0>(a, grid, st);
}

// recompute variant (8-wave workgroups): mask none or the standard SW-MSA mask (a.mask_is_shift with a.shift > 0)
int fmmt_wattn_mfma_bwd_rc_launch(const WaArgs& a, int grid, hipStream_t st) {
    const bool masked = a.shift > 0;
    if (a.C == 96) return masked ? launch_bwd<1, 4, 96>(a, grid, st) : launch_bwd<0, 4, 96>(a, grid, st);
    if (a.C == 192) return masked ? launch_bwd<1, 4, 192>(a, grid, st) : launch_bwd<0, 4, 192>(a, grid, st);
    return FMMT_EINVAL;
}
