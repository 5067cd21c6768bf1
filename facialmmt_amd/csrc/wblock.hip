// The attention half of a Swin block, one launch, one wavefront per window (gfx950, bf16):
//
//     y = x + rowscale * ( proj( W-MSA / SW-MSA( LN1(x) . Wqkv^T + bqkv ) ) + bproj )        Swin_Transformer.py:233-266, :113-144
//
// As separate launches (LayerNorm, qkv Linear, attention core, proj Linear) a stage-0 block moves LN(x) (385 MB at 640 frames),
// qkv (1.16 GB) and the attention output (385 MB) through HBM and back; here a window's 49 tokens are read once and everything
// between x and y lives in the registers of ONE wave:
//
//   x fragments      16-byte loads of (token li, channels 32 ks + 8 lg ..): the B (or A) operand layout of a 32-deep K block;
//                    the window gather with cyclic shift is the address (roll / window_partition never exist)
//   LN1              a token's C channels sit in the 4 lanes li + 16 lg: in-lane sums + two cross-lane steps
//   q^T, k^T         D[channel][token] = W (A side, rows in fragment order) . xn^T (B side): with the head-channel permutation
//                    row 4g + r of tile nt <-> channel 8g + 4nt + r a lane's two accumulator tiles ARE the 8 consecutive head
//                    channels of token li, i.e. the operand fragments of S^T = K . Q^T -- no LDS round trip
//   v                the transposed product D[token][channel] = xn (A side) . Wv^T (B side): two token tiles of accumulators are
//                    the 8 key slots of the A fragment of O^T = V^T . P^T (the k-slot order of that contraction is free and is
//                    chosen to match, as in wattn_mfma.hip)
//   softmax          on S^T: a lane owns one query column; logits in base 2 (scale and the dense LDS bias table carry log2 e: one FMA
//                    per logit, v_exp_f32 directly); the SW-MSA mask from window coordinates, skipped (wave-uniform branch) for the
//                    interior windows, which see none; row maxima by two row / half swaps (v_permlane16_swap / 32_swap, no LDS);
//                    probabilities stay unnormalised in (0, 1] for the second product, whose third A tile of ones returns the row
//                    sums from the matrix cores (the kernel is VALU-bound: 3.2 k VALU against 0.4 k MFMA instructions per window)
//   proj             D[channel][token] = Wproj (A side) . O^T (B side): O's accumulator tiles are the B fragments of K block h
//   epilogue         + bias, DropPath row scale, + x (re-read through L2, requested before the previous tile's stores), 16-byte stores
//
// Weights (Wqkv, Wproj: 73 KB at C = 96), the dense relative-position bias (3 heads: 40 KB) and the LayerNorm / bias vectors
// are staged once per workgroup in LDS (122 KB: one workgroup of eight waves per CU, two waves per SIMD, 256 registers each);
// the waves then run independently -- there is no barrier in the window loop.
//
// Against the four-launch path (fmmt_layernorm_fwd -> fmmt_linear_fwd -> fmmt_window_attn_fwd -> fmmt_linear_fwd) the operand
// roundings are the same (bf16 LN output, q / k / v, probabilities, attention output) and the accumulation orders of the three
// GEMMs too; the softmax differs in the last bits (base-2 exponentials, normalisation after the second product):
// tests/test_gpu_wblock.py holds the two within a bf16 rounding of one another and both to an fp64 restatement.
// For the backward the kernel also emits what the four launches would have left in HBM minus qkv: LN1(x), the attention output,
// the row statistics and the log-sum-exp.  Measured (640 frames, stage 0): 0.62 ms against 1.31 ms for the four launches.
#include "wblock_common.h"

namespace {

// Scheduling fences between the phases of a window: without any, the scheduler hoists the LDS fragment reads of later phases over
// earlier ones and the kernel wants 450-900 registers.  Kept after each q / k product, the v product and each query tile of the
// attention core; none after the LayerNorm tiles and the proj tiles (with those two the allocator spills 6 registers -- and a
// scratch reload behind the window's stores waits for them, see the epilogue).
#define WB_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ f32x2 bf2_to_f2(unsigned u) {                 // two packed bf16 -> two floats
    return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}

template <int C, int NW>
__global__ __launch_bounds__(NW * 64) void wblock_fwd_kernel(WbArgs p) {
    using L = WbLds<C>;
    constexpr int NH = L::NH, KS = C / 32, PITCH = L::PITCH, BP = WB_BP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Wl = reinterpret_cast<bf16*>(smem);
    float* Bs = reinterpret_cast<float*>(smem + L::W_BYTES);
    float* Gm = Bs + NH * TOK * BP;
    float* Bt = Gm + C;
    float* Bq = Bt + C;
    float* Bp = Bq + 3 * C;

    // ---------------------------------------------------------------- stage the operands every window shares
    for (int q = threadIdx.x; q < 4 * C * (C / 8); q += NW * 64) {
        const int d = q / (C / 8), ch = q - d * (C / 8);
        bool is_proj;
        const int sr = wb_src_row<C>(d, is_proj);
        const bf16* src = (is_proj ? p.wproj : p.wqkv) + (size_t)sr * C + ch * 8;
        *reinterpret_cast<bf16x8*>(Wl + d * PITCH + ch * 8) = *reinterpret_cast<const bf16x8*>(src);
    }
    // dense relative-position bias, pre-multiplied by log2(e): the softmax runs on base-2 exponentials (one FMA per logit)
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
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    const int cperm[2] = {(li >> 2) * 8 + (li & 3), (li >> 2) * 8 + 4 + (li & 3)};      // head channel of column li of v tile dt

    auto load_x = [&](int b_, bf16x8 (&dst)[4][KS]) {
        const WinPos P = win_pos(p, b_);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16* row = p.x + tok_of(p, P, G.di[t], G.dj[t]) * C + lg * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dst[t][ks] = ld_frag(row + ks * 32);
        }
    };

    // nrm: this window's x fragments on entry, LN1(x) after the LayerNorm (in place)
    auto body = [&](int b_, bf16x8 (&nrm)[4][KS]) {
        const WinPos P = win_pos(p, b_);
        int tok[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) tok[t] = (int)tok_of(p, P, G.di[t], G.dj[t]);
        const float rs = p.rowscale ? p.rowscale[P.img] : 1.0f;
        const bool masked = p.shift > 0 && (P.lastrow || P.lastcol);          // wave-uniform: interior windows of SW-MSA see no mask

        // ------------------------------------------------------------ LayerNorm (Swin_Transformer.py:239,243), packed fp32 math
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x2 v[KS * 4];
            f32x2 s2 = {0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 u = *reinterpret_cast<const u32x4*>(&nrm[t][ks]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[ks * 4 + j] = bf2_to_f2(u[j]);
                    s2 += v[ks * 4 + j];
                }
            }
            const float mean = swap_sum(s2.x + s2.y) * invC;
            f32x2 q2 = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < KS * 4; ++e) {
                v[e] -= mean;
                q2 += v[e] * v[e];
            }
            const float rstd = rsqrtf(swap_sum(q2.x + q2.y) * invC + p.eps);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gm + ks * 32 + lg * 8), g1 = *reinterpret_cast<const f32x4*>(Gm + ks * 32 + lg * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bt + ks * 32 + lg * 8), b1 = *reinterpret_cast<const f32x4*>(Bt + ks * 32 + lg * 8 + 4);
                const f32x2 y0 = v[ks * 4 + 0] * rstd * f32x2{g0[0], g0[1]} + f32x2{b0[0], b0[1]};
                const f32x2 y1 = v[ks * 4 + 1] * rstd * f32x2{g0[2], g0[3]} + f32x2{b0[2], b0[3]};
                const f32x2 y2 = v[ks * 4 + 2] * rstd * f32x2{g1[0], g1[1]} + f32x2{b1[0], b1[1]};
                const f32x2 y3 = v[ks * 4 + 3] * rstd * f32x2{g1[2], g1[3]} + f32x2{b1[2], b1[3]};
                bf16x8 o;
                o[0] = (bf16)y0.x; o[1] = (bf16)y0.y; o[2] = (bf16)y1.x; o[3] = (bf16)y1.y;
                o[4] = (bf16)y2.x; o[5] = (bf16)y2.y; o[6] = (bf16)y3.x; o[7] = (bf16)y3.y;
                nrm[t][ks] = o;
                if (p.xn && G.valid[t]) *reinterpret_cast<bf16x8*>(p.xn + (size_t)tok[t] * C + ks * 32 + lg * 8) = o;
            }
            if (p.mean && G.valid[t] && lg == 0) {
                p.mean[tok[t]] = mean;
                p.rstd[tok[t]] = rstd;
            }
        }

        bf16x8 of[NH][4];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            // -------------------------------------------------------- q^T, k^T of head h: [32 channels][64 tokens]
            bf16x8 qk[2][4];
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
                        const bf16x8 wf = ld_frag(Wl + (((h * 3 + part) * 2 + nt) * 16 + li) * PITCH + ks * 32 + lg * 8);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, nrm[t][ks], acc[nt][t], 0, 0, 0);
                    }
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bq + part * C + h * 32 + lg * 8);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bq + part * C + h * 32 + lg * 8 + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 a0 = acc[0][t] + b0, a1 = acc[1][t] + b1;
                    bf16x8 f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f[r] = (bf16)a0[r];
                        f[4 + r] = (bf16)a1[r];
                    }
                    qk[part][t] = f;
                }
                WB_FENCE();
            }
            // -------------------------------------------------------- v of head h, transposed product: [64 tokens][32 channels]
            bf16x8 vT[2][2];
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
                        const bf16x8 wf = ld_frag(Wl + (((h * 3 + 2) * 2 + dt) * 16 + li) * PITCH + ks * 32 + lg * 8);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nrm[t][ks], wf, acc[t][dt], 0, 0, 0);
                    }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const float bv = Bq[2 * C + h * 32 + cperm[dt]];
#pragma unroll
                    for (int ks2 = 0; ks2 < 2; ++ks2) {
                        const f32x4 a0 = acc[2 * ks2][dt] + bv, a1 = acc[2 * ks2 + 1][dt] + bv;
                        bf16x8 f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = (bf16)a0[r];
                            f[4 + r] = (bf16)a1[r];
                        }
                        vT[ks2][dt] = f;
                    }
                }
                WB_FENCE();
            }

                // -------------------------------------------------------- attention core of head h.  S^T = K . Q^T: a lane owns query column
            // q = 16 qt + li and the keys 16 kt + 4 lg + r; logits in base 2: s2 = (q.k scale + bias) log2 e; P is left unnormalised
            // (in (0, 1]) for the second product and O is divided by the row sum afterwards
            const float* bh = Bs + h * TOK * BP;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const int q = qt * 16 + li;
                const float* brow = bh + (q < TOK ? q : TOK - 1) * BP + lg * 4;
                f32x4 s[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qk[1][kt], qk[0][qt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
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
                bf16x8 pb0, pb1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pb0[r] = (bf16)s[0][r]; pb0[4 + r] = (bf16)s[1][r];
                    pb1[r] = (bf16)s[2][r]; pb1[4 + r] = (bf16)s[3][r];
                }
                // O^T = V^T . P^T, and the row sums by the same instruction: a third A tile of ones makes every row of its result
                // the sum over the keys of the (bf16-rounded) probabilities the numerator was formed from.  (Summing the unrounded
                // exponentials on the VALU, or normalising before the rounding as the four-launch kernel does, gives the same
                // end-to-end gradient statistics: tests/test_gpu_swin.py::test_bf16_gradients_against_oracle_n32.)
                f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f}, ls = {0.f, 0.f, 0.f, 0.f};
                o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT[0][0], pb0, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT[0][1], pb0, o1, 0, 0, 0);
                ls = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb0, ls, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT[1][0], pb1, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT[1][1], pb1, o1, 0, 0, 0);
                ls = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb1, ls, 0, 0, 0);
                const float l = ls[0];
                const float inv = __builtin_amdgcn_rcpf(l);
                o0 *= inv;
                o1 *= inv;
                bf16x8 ob;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ob[r] = (bf16)o0[r];
                    ob[4 + r] = (bf16)o1[r];
                }
                of[h][qt] = ob;
                if (G.valid[qt]) {
                    if (p.o) *reinterpret_cast<bf16x8*>(p.o + (size_t)tok[qt] * C + h * HD + lg * 8) = ob;
                    if (lg == 0) p.lse[((size_t)b_ * NH + h) * TOK + q] = (m + __builtin_amdgcn_logf(l)) * WB_LN2;     // natural-log lse
                }
                WB_FENCE();
            }
            }

        // ------------------------------------------------------------ proj + bias, DropPath scale, residual (Swin_Transformer.py:142,266)
        auto epi_compute = [&](f32x4 (&acc)[2 * KS], const bf16x8 (&xr)[KS], bf16x8 (&ob)[KS]) {
#pragma unroll
            for (int c = 0; c < KS; ++c) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bp + c * 32 + lg * 8), b1 = *reinterpret_cast<const f32x4*>(Bp + c * 32 + lg * 8 + 4);
                const u32x4 u = *reinterpret_cast<const u32x4*>(&xr[c]);
                const f32x4 a0 = (acc[2 * c] + b0) * rs, a1 = (acc[2 * c + 1] + b1) * rs;
                const f32x2 y0 = f32x2{a0[0], a0[1]} + bf2_to_f2(u[0]), y1 = f32x2{a0[2], a0[3]} + bf2_to_f2(u[1]);
                const f32x2 y2 = f32x2{a1[0], a1[1]} + bf2_to_f2(u[2]), y3 = f32x2{a1[2], a1[3]} + bf2_to_f2(u[3]);
                bf16x8 o;
                o[0] = (bf16)y0.x; o[1] = (bf16)y0.y; o[2] = (bf16)y1.x; o[3] = (bf16)y1.y;
                o[4] = (bf16)y2.x; o[5] = (bf16)y2.y; o[6] = (bf16)y3.x; o[7] = (bf16)y3.y;
                ob[c] = o;
            }
        };
        auto epi_store = [&](int t, const bf16x8 (&ob)[KS]) {
            if (G.valid[t]) {
#pragma unroll
                for (int c = 0; c < KS; ++c) *reinterpret_cast<bf16x8*>(p.y + (size_t)tok[t] * C + c * 32 + lg * 8) = ob[c];
            }
        };
        auto proj_tile = [&](int t, f32x4 (&acc)[2 * KS]) {
#pragma unroll
            for (int cn = 0; cn < 2 * KS; ++cn) acc[cn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int cn = 0; cn < 2 * KS; ++cn) {
                    const bf16x8 wf = ld_frag(Wl + (3 * C + cn * 16 + li) * PITCH + h * 32 + lg * 8);
                    acc[cn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, of[h][t], acc[cn], 0, 0, 0);
                }
        };
        // The residual is x again (L2 / MALL: this CU read it a few microseconds ago).  Tile t + 1's fragments are requested BEFORE
        // tile t's stores are issued: vmcnt retires in order, so a load issued behind a store cannot be waited for without waiting
        // out the store's whole write latency.  (Measured alternatives, all slower because the extra live registers spill and every
        // scratch reload is such a load behind stores: keeping x in registers, prefetching the next window's x behind the last
        // head / at the head of the epilogue / tile by tile into the freed residual registers -- 0.70-0.84 ms against 0.62.)
        bf16x8 xres[2][KS];
        auto load_res = [&](int t, bf16x8 (&dst)[KS]) {
#pragma unroll
            for (int c = 0; c < KS; ++c) dst[c] = ld_frag(p.x + (size_t)tok[t] * C + c * 32 + lg * 8);
        };
        load_res(0, xres[0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t + 1 < 4) load_res(t + 1, xres[(t + 1) & 1]);
            f32x4 acc[2 * KS];
            proj_tile(t, acc);
            bf16x8 ob[KS];
            epi_compute(acc, xres[t & 1], ob);
            epi_store(t, ob);
        }
    };

    bf16x8 buf[4][KS];
    int b = blockIdx.x * NW + wave;
    for (; b < B_; b += nwt) {
        load_x(b, buf);
        body(b, buf);
    }
}

// NW waves per workgroup, one workgroup per CU (the LDS image is 122 KB).  NW = 4: one wave per SIMD with the whole register
// file (the window's x, LN1(x) and O fragments alone are 144 registers).
template <int C, int NW>
int wb_launch(const WbArgs& a, hipStream_t st) {
    constexpr int lds = WbLds<C>::TOTAL;
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&wblock_fwd_kernel<C, NW>), lds)) return rc_;
    const int need = (a.B_ + NW - 1) / NW;
    const int grid = need < 256 ? need : 256;
    hipLaunchKernelGGL((wblock_fwd_kernel<C, NW>), dim3(grid), dim3(NW * 64), lds, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

bool wb_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int wb_fill(WbArgs& a, int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                   const void* x, const float* ln_gamma, const float* ln_beta, float eps,
                   const void* wqkv, const float* bqkv, const void* wproj, const float* bproj,
                   const float* table, const int32_t* index, float scale, const float* rowscale,
                   void* y, void* xn, void* attn_out, float* mean, float* rstd, float* lse) {
    if ((dtype != FMMT_BF16 && dtype != FMMT_F32) || n_img <= 0 || H <= 0 || W <= 0 || H % WS || W % WS || shift < 0 || shift >= WS) return FMMT_EINVAL;
    if (C != 96 || num_heads * HD != C) return FMMT_EINVAL;                       // other widths: the four-launch form
    if (!x || !ln_gamma || !ln_beta || !wqkv || !wproj || !table || !index || !y || !lse) return FMMT_EINVAL;
    if ((mean == nullptr) != (rstd == nullptr)) return FMMT_EINVAL;
    if (!wb_al16(x) || !wb_al16(wqkv) || !wb_al16(wproj) || !wb_al16(y) || (xn && !wb_al16(xn)) || (attn_out && !wb_al16(attn_out))) return FMMT_EALIGN;
    a = WbArgs{};
    a.n_img = n_img; a.H = H; a.W = W; a.shift = shift;
    a.x = (const bf16*)x; a.ln_g = ln_gamma; a.ln_b = ln_beta; a.eps = eps;
    a.wqkv = (const bf16*)wqkv; a.bqkv = bqkv; a.wproj = (const bf16*)wproj; a.bproj = bproj;
    a.table = table; a.index = index; a.scale = scale; a.rowscale = rowscale;
    a.y = (bf16*)y; a.xn = (bf16*)xn; a.o = (bf16*)attn_out; a.mean = mean; a.rstd = rstd; a.lse = lse;
    a.B_ = n_img * (H / WS) * (W / WS);
    return 0;
}

extern "C" int fmmt_window_block_fwd(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                                     const void* x, const float* ln_gamma, const float* ln_beta, float eps,
                                     const void* wqkv, const float* bqkv, const void* wproj, const float* bproj,
                                     const float* table, const int32_t* index, float scale, const float* rowscale,
                                     void* y, void* xn, void* attn_out, float* mean, float* rstd, float* lse, void* stream) {
    WbArgs a;
    if (int rc = wb_fill(a, dtype, n_img, H, W, C, num_heads, shift, x, ln_gamma, ln_beta, eps, wqkv, bqkv, wproj, bproj, table, index, scale, rowscale,
                         y, xn, attn_out, mean, rstd, lse)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FMMT_F32) return fmmt_wblock_ref_fwd_launch(FMMT_F32, &a, st);     // parity: fp32 operands, wblock_ref.hip
    return wb_launch<96, 8>(a, st);
}

// The element-type-generic restatement of the same kernel (wblock_ref.hip): dtype FMMT_F32 = what fmmt_window_block_fwd(FMMT_F32) runs,
// FMMT_BF16 = the bf16 instantiation of the generic template (a test compares it with the production kernel).
extern "C" int fmmt_window_block_fwd_ref(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                                         const void* x, const float* ln_gamma, const float* ln_beta, float eps,
                                         const void* wqkv, const float* bqkv, const void* wproj, const float* bproj,
                                         const float* table, const int32_t* index, float scale, const float* rowscale,
                                         void* y, void* xn, void* attn_out, float* mean, float* rstd, float* lse, void* stream) {
    WbArgs a;
    if (int rc = wb_fill(a, dtype, n_img, H, W, C, num_heads, shift, x, ln_gamma, ln_beta, eps, wqkv, bqkv, wproj, bproj, table, index, scale, rowscale,
                         y, xn, attn_out, mean, rstd, lse)) return rc;
    return fmmt_wblock_ref_fwd_launch(dtype, &a, reinterpret_cast<hipStream_t>(stream));
}
