// Fused Mlp forward of the Swin stages with few channels (stage 0: C = 96, stage 1: C = 192; hidden 4C):
//
//     y = res + rowscale * ( gelu(x . W1^T + b1) . W2^T + b2 )          (Swin_Transformer.py:14-30, :268)
//
// in ONE launch, the 4C-wide hidden activation never leaving the CU.  As two launches (fc1 + GELU epilogue, fc2 + residual
// epilogue) a stage-0 block moves 5.8 GB through HBM for 0.3 TFLOP -- x in, pre-activation AND activation out, activation in
// again, residual in, y out -- and both launches are HBM streams.  Here x is read once, the pre-activation
// is written once (training only: the backward's GELU' and the recomputed activation need it), y once: 2.7 GB.
//
// Decomposition.  A workgroup of 8 waves walks 256-token tiles; wave w owns tokens [32 w, 32 w + 32) of the tile for BOTH
// products.  Hidden channels go by in blocks of 32:
//   product 1   D1[hidden 32][token 32] = W1 block . x^T      (weight on the MFMA A side, tokens on the B side; x fragments
//               stay in registers for the whole tile, K = C)
//   epilogue 1  + b1, optional bf16 store of the pre-activation, erf-GELU (packed fast form, fmmt_common.h)
//   product 2   D2[channel C][token 32] += W2 block . h       where the B operand h is product 1's OWN accumulator: with the
//               hidden-row permutation chan_of<8> a lane's two 16x16 accumulator tiles hold 8 consecutive hidden channels
//               of one token, which is exactly the B-fragment layout of a 32-deep K block -- no LDS round trip, no shuffle
//               (the P.V trick of mha_mfma.hip).
// The weights stream through an LDS ring shared by the 8 waves: a stage = 64 hidden channels = W1 rows [64][C] (stored per
// 32-wide K slice, 64-byte rows), W2 columns as two [C][32] blocks, and the 64 bias values; filled by direct global->LDS DMA,
// two stages in flight behind the one in use, counted vmcnt + one raw s_barrier per stage, and -- as in
// linear_nt_p256_kernel -- the ring keeps running across tile boundaries.  Weights come from L2 (144 / 576 KB per tile,
// re-read by every workgroup); activations are the HBM traffic.
//
// Accumulation orders equal those of the two-launch path (K ascending in both products, bf16 rounding of h before product 2),
// so the result is bit-identical to fmmt_linear_fwd(GELU) followed by fmmt_linear_fwd(residual): tests/support_op_cases.py::t_mlp_fused.
#include "gemm_common.h"
#include "mlp_args.h"

namespace {

// (round 6: the hidden tensors -- derivative / pre-activation, activation, dh -- written non-temporal, as the GEMM epilogues' whole-line stores are:
//  Swin forward + backward 39.47 / 39.53 -> 39.43 / 39.52 ms, whole step 57.52 / 57.25 -> 57.26 / 57.69 ms, same call: nothing; plain stores stay)
__device__ __forceinline__ void st_hidden(bf16* dst, const bf16x8& v) { *reinterpret_cast<bf16x8*>(dst) = v; }

// (Two workgroups per CU at C = 96 -- 76.8 KB of LDS each, registers forced to 128 -- were measured: the inference form gains
//  13 %, the training forms, whose time goes into the hidden-tensor stores, lose 0-19 % to the 20 spilled registers.)
// FULL: M is a multiple of the 256-token tile -- no token guard anywhere, the stage body is branch-free (the stores of a ragged tail
// are conditional: branches, which end the compiler's scheduling regions inside the stage)
// DG (FMMT_SAVE_DG, round 6): h_pre receives gelu'(pre-activation) instead of the pre-activation -- the one thing the backward wants of it -- formed beside
// gelu() from the same exponential (fmmt_common.h, gelu_both_exp_f); the backward kernel then multiplies instead of evaluating a polynomial.
template <int C, bool LN, bool FULL, bool DG = false>
__global__ __launch_bounds__(512) void mlp_fused_fwd_kernel(MlpArgs p) {
    using T = bf16;
    constexpr int H = 4 * C, HS = 64, NS = H / HS;          // hidden channels per ring stage, stages per tile
    constexpr int KS = C / 32;                              // K steps of product 1
    constexpr int NT2 = C / 16, CW2 = 4 * NT2;              // product 2: 16-row output-channel tiles per wave, channels per lane
    constexpr int W1_EL = KS * HS * 32, W2_EL = 2 * C * 32; // elements per stage
    constexpr int STAGE_B = (W1_EL + W2_EL) * 2 + 1024;     // bytes: + a 1 KB slot for the 64 bias values (256 B used)
    constexpr int NBUF = 3;
    constexpr int NI = KS * (HS / 16) + 2 * (C / 16);       // DMA instructions per stage (16 rows x 64 B each), dealt round-robin
    static_assert(NI % 8 == 0, "uniform DMA count per wave");
    constexpr int CNT = NI / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    auto swz = [](int row) { return ((row >> 3) ^ (row >> 2)) & 3; };       // 16-byte chunk swizzle of 64-byte rows (as linear_nt_deep32)
    const int r16 = lane >> 2, c4 = lane & 3;

    // A DMA piece's global address = (uniform: matrix base + the stage's first hidden channel) + (this lane's 32-bit byte offset inside
    // the stage, fixed for the kernel): the SGPR-base form of the instruction, one VGPR per piece (as 64-bit pointers formed per stage
    // the pieces cost ~10 VALU instructions each, or -- hoisted by the compiler -- two registers each, which at C = 192 spilled)
    unsigned loff[CNT];
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int j = i * 8 + wave;                          // wave-uniform
        if (j < KS * 4) {                                    // W1: K slice ks, hidden rows r0 .. r0 + 15 of the stage
            const int ks = j >> 2, r = (j & 3) * 16 + r16;
            loff[i] = (unsigned)(r * C + ks * 32 + ((c4 ^ swz(r)) << 3)) * 2u;
        } else {                                             // W2: hidden block b, output-channel rows r0 .. r0 + 15
            const int q = j - KS * 4, b = q / (C / 16), r = (q % (C / 16)) * 16 + r16;
            loff[i] = (unsigned)(r * H + b * 32 + ((c4 ^ swz(r)) << 3)) * 2u;
        }
    }
    auto issue = [&](int slot, int hs) {
        char* base = smem + slot * STAGE_B;
        const char* u1 = reinterpret_cast<const char*>(p.w1 + (size_t)hs * HS * C);
        const char* u2 = reinterpret_cast<const char*>(p.w2 + (size_t)hs * HS);
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int j = i * 8 + wave;
            T* dst;
            if (j < KS * 4) {
                const int ks = j >> 2, r0 = (j & 3) * 16;
                dst = reinterpret_cast<T*>(base) + ks * (HS * 32) + r0 * 32;
            } else {
                const int q = j - KS * 4, b = q / (C / 16), r0 = (q % (C / 16)) * 16;
                dst = reinterpret_cast<T*>(base) + W1_EL + b * (C * 32) + r0 * 32;
            }
            unsigned lo = loff[i];
            asm volatile("" : "+v"(lo));                     // keep the offset a 32-bit register (hoisted as a zero-extended pair otherwise)
            __builtin_amdgcn_global_load_lds((gptr_t*)((j < KS * 4 ? u1 : u2) + (size_t)lo), (lptr_t*)dst, 16, 0, 0);
        }
        if (wave == 7) {                                     // 64 bias values (lanes 16.. re-read the last 16 bytes into the slot's unused tail)
            unsigned bo = (unsigned)min(lane, 15) * 16u;
            asm volatile("" : "+v"(bo));
            __builtin_amdgcn_global_load_lds((gptr_t*)(reinterpret_cast<const char*>(p.b1 + hs * HS) + (size_t)bo), (lptr_t*)(base + (W1_EL + W2_EL) * 2), 16, 0, 0);
        }
    };
    // Stage s has landed when all but the youngest N of this wave's memory operations are complete, N = what was issued after
    // stage s's DMA and may still fly: the next stage's DMA (CNT, + 1 bias slab for wave 7) and -- CDNA4's vmcnt counts
    // stores too -- the four pre-activation stores of the stage computed in between.  Only a LOWER bound on that number is
    // safe (too large an N would let stage s itself count as "young"): the stores are counted only when every lane of the
    // wave certainly issued them (no ragged tail, pre-activation requested); elsewhere the wait is merely stricter.
    auto wait_landed = [&](bool one_ahead, int stores) {      // stores: 0, 4 or 8 vector stores certainly issued by the previous step
        if (!one_ahead) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (stores == 8) {
            if (wave == 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + 9) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + 8) : "memory");
        } else if (stores == 4) {
            if (wave == 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + 5) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + 4) : "memory");
        } else {
            if (wave == 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
        }
    };

    // fragment offsets (elements) inside a stage
    int w1off[2], w2off[NT2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int r = chan_of<8>(nt, li >> 2, li & 3);       // hidden row inside a 32-block: lg*8 + 4*nt + r after the MFMA
        w1off[nt] = r * 32 + ((lg ^ swz(r)) << 3);
    }
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
        const int r = chan_of<CW2>(nt, li >> 2, li & 3);
        w2off[nt] = W1_EL + r * 32 + ((lg ^ swz(r)) << 3);
    }

    const int G = gridDim.x;
    const int ntile = (int)blockIdx.x < p.tiles ? (p.tiles - (int)blockIdx.x + G - 1) / G : 0;
    const int nsteps = ntile * NS;
    int ihs = 0, islot = 0;
    auto issue_next = [&]() {
        issue(islot, ihs);
        islot = islot + 1 == NBUF ? 0 : islot + 1;
        ihs = ihs + 1 == NS ? 0 : ihs + 1;
    };
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < nsteps) issue_next();

    bf16x8 xf[2][KS];
    // LN, C = 96: the residual of the tile's epilogue is x itself, and an accumulator element of product 2 (token li; channel
    // (nt / 2) * 32 + lg * 8 + (nt % 2) * 4 + r) is the element (nt % 2) * 4 + r of THIS lane's x fragment nt / 2 -- the raw fragments are
    // kept (24 registers) and handed to nt_epilogue as its prefetched operand: no second read of x (385 MB per stage-0 launch), no load --
    // and so no memory round trip -- between the tile's last MFMA and its stores.  The rows' DropPath scales travel with the x loads.
    constexpr bool KEEPX = LN && C == 96;
    EpiPre<2, NT2> pre;
    float rsn[2] = {1.f, 1.f};
    auto load_x = [&](int tile) {
        const int t0 = tile * 256 + wave * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int tok = min(t0 + mt * 16 + li, p.M - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[mt][ks] = *reinterpret_cast<const bf16x8*>(p.x + (size_t)tok * C + ks * 32 + lg * 8);
            if constexpr (KEEPX) rsn[mt] = row_scale(p.rowscale, tok, p.rows_per_scale);
        }
    };
    f32x4 acc2[2][NT2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int tile = blockIdx.x, hs = 0, cslot = 0;
    if (ntile > 0) load_x(tile);
    for (int s = 0; s < nsteps; ++s) {
        // hs > 0: the previous step belonged to this tile and stored its four pre-activation vectors (a tile's first step
        // follows the previous tile's epilogue instead: more, not fewer, younger operations -- the plain count stays safe)
        const bool full = hs > 0 && (FULL || tile * 256 + wave * 32 + 32 <= p.M);
        wait_landed(s + 1 < nsteps, full ? 4 * (int)(p.h_pre != nullptr) + 4 * (int)(p.h_act != nullptr) : 0);
        __builtin_amdgcn_s_barrier();                        // stage s is in LDS for every wave; the stage read in step s - 1 is free
        if (s + NBUF - 1 < nsteps) issue_next();
        const T* sb = reinterpret_cast<const T*>(smem + cslot * STAGE_B);
        const float* bs = reinterpret_cast<const float*>(smem + cslot * STAGE_B + (W1_EL + W2_EL) * 2);
        const int t0 = tile * 256 + wave * 32;
        if constexpr (LN) {
            // norm2 of the block (Swin_Transformer.py:267-268) on the fragments this tile's products consume: a token's C channels sit
            // in the four lanes li + 16 g, so the row statistics are in-lane sums plus two cross-lane steps; LN(x) goes out once
            // (training: fc1's weight gradient contracts with it), the residual is x itself
            if (hs == 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int tok = t0 + mt * 16 + li;
                    if constexpr (KEEPX) {
                        pre.rs[mt] = rsn[mt];
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) pre.full[mt][ks] = xf[mt][ks];
                    }
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
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_g + ks * 32 + lg * 8), g1 = *reinterpret_cast<const f32x4*>(p.ln_g + ks * 32 + lg * 8 + 4);
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln_b + ks * 32 + lg * 8), b1 = *reinterpret_cast<const f32x4*>(p.ln_b + ks * 32 + lg * 8 + 4);
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = (bf16)(v[ks * 8 + e] * rstd * g0[e] + b0[e]);
                            o[4 + e] = (bf16)(v[ks * 8 + 4 + e] * rstd * g1[e] + b1[e]);
                        }
                        xf[mt][ks] = o;
                        if (p.xn && (FULL || tok < p.M)) *reinterpret_cast<bf16x8*>(p.xn + (size_t)tok * C + ks * 32 + lg * 8) = o;
                    }
                    if (p.mean && (FULL || tok < p.M) && lg == 0) {
                        p.mean[tok] = mean;
                        p.rstd[tok] = rstd;
                    }
                }
            }
        }
        if constexpr (C == 96) {
        // ---- C = 96: the stage as ONE software pipeline over its two 32-channel hidden blocks:
        //     RB R1(0) R2(0) | P1(0) | R1(1) | P1(1) + E(0) | P2(0) + E(1) | R2(1) | stores | P2(1)
        // (R = fragment reads, issued by hand a whole block at a time and one phase ahead of their use; P1 / P2 = the products' 12 MFMAs
        // each; E = bias + GELU + conversions): three lgkmcnt waits per stage -- behind the barrier, behind P1(0)'s MFMAs, behind the
        // stores -- instead of the ~13 the compiler places in front of groups of 2-4 MFMAs in the straightforward order (the C = 192 body
        // below).  The W1 / W2 fragment registers are reused by the second block's reads (issued after the MFMAs that consume the first
        // block's have been issued; the data comes back >= 64 cycles later).  Offsets inside a stage are immediates of the reads.
        // What it bought, and why not more (round 4; all same-call): inference form 0.831 -> 0.810 ms, training form 1.043 -> 1.025.  The
        // ablations -- GELU removed 0.60 ms, GELU and product 2 removed 0.49, DMA + barrier removed 0.73, residual / next-x loads of the
        // tile end removed 0.64 -- and the issue-rate probe (tools/probes/valu_rate.hip, profiles/r04_issue_rates.txt) say the stage is
        // ISSUE-bound, not latency-bound: on gfx950 a SIMD's two waves do not overlap VALU with MFMA issue -- 8 v_pk_fma_f32 + 1 MFMA
        // take 67 cycles where the VALU alone takes 40 and the MFMA alone 16.6 --, VOP3 / packed fp32 instructions cost ~5 cycles per
        // wave instruction (VOP2 2.6), so the ~370 VALU + 48 MFMA instructions of a wave-stage cost their sum whatever their order.
        static_assert(HS * 32 * 2 == 4096 && 32 * 32 * 2 == 2048 && C * 32 * 2 == 6144 && (W1_EL + W2_EL) * 2 == 24576, "literal offsets of the reads below");
        const unsigned sbase = (unsigned)(uintptr_t)(lptr_t*)smem + (unsigned)cslot * STAGE_B;
        const unsigned a10 = sbase + (unsigned)w1off[0] * 2u, a11 = sbase + (unsigned)w1off[1] * 2u;
        const unsigned a20 = sbase + (unsigned)w2off[0] * 2u, a21 = sbase + (unsigned)w2off[1] * 2u;     // w2off[nt] = w2off[nt & 1] + (nt >> 1) * 1024
        const unsigned ab = sbase + 24576u + (unsigned)lg * 32u;
        bf16x8 w1f[KS][2], w2f[NT2];
        f32x4 bbf[2][2];
#define FMMT_RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr) : "memory")
#define FMMT_PIN(x) asm volatile("" : "+v"(x))
        FMMT_RD(bbf[0][0], ab, 0); FMMT_RD(bbf[0][1], ab, 16); FMMT_RD(bbf[1][0], ab, 128); FMMT_RD(bbf[1][1], ab, 144);
        FMMT_RD(w1f[0][0], a10, 0); FMMT_RD(w1f[0][1], a11, 0); FMMT_RD(w1f[1][0], a10, 4096); FMMT_RD(w1f[1][1], a11, 4096);
        FMMT_RD(w1f[2][0], a10, 8192); FMMT_RD(w1f[2][1], a11, 8192);
        FMMT_RD(w2f[0], a20, 0); FMMT_RD(w2f[1], a21, 0); FMMT_RD(w2f[2], a20, 2048); FMMT_RD(w2f[3], a21, 2048);
        FMMT_RD(w2f[4], a20, 4096); FMMT_RD(w2f[5], a21, 4096);
        auto pin_w1 = [&]() {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { FMMT_PIN(w1f[ks][0]); FMMT_PIN(w1f[ks][1]); }
        };
        auto pin_w2 = [&]() {
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) FMMT_PIN(w2f[nt]);
        };
        auto prod1 = [&](f32x4 (&acc1)[2][2]) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[ks][nt], xf[mt][ks], acc1[mt][nt], 0, 0, 0);
        };
        auto prod2 = [&](const bf16x8 (&hf)[2]) {
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[nt], hf[mt], acc2[mt][nt], 0, 0, 0);
        };
        // epilogue 1 of a block: this lane holds, per token tile, hidden channels lg * 8 .. + 7 of the block for token li
        auto epi1 = [&](const f32x4 (&acc1)[2][2], const f32x4 (&bb)[2], bf16x8 (&pre)[2], bf16x8 (&hf)[2]) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc1[mt][0][r] + bb[0][r];
                    v[4 + r] = acc1[mt][1][r] + bb[1][r];
                }
                if constexpr (DG) {
                    float d[8];
                    gelu_both_inplace<T>(v, d, 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) pre[mt][e] = (bf16)d[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pre[mt][e] = (bf16)v[e];
                    gelu_inplace<T>(v, 8);                   // fmmt_common.h: FMAs and one v_exp_f32, no table, no wait
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) hf[mt][e] = (bf16)v[e];
            }
        };
        f32x4 acc1a[2][2], acc1b[2][2];
        bf16x8 pre0[2], pre1[2], hf0[2], hf1[2];
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // bias + W1(0) are in
        FMMT_PIN(bbf[0][0]); FMMT_PIN(bbf[0][1]); FMMT_PIN(bbf[1][0]); FMMT_PIN(bbf[1][1]);
        pin_w1();
        prod1(acc1a);
        FMMT_RD(w1f[0][0], a10, 2048); FMMT_RD(w1f[0][1], a11, 2048); FMMT_RD(w1f[1][0], a10, 6144); FMMT_RD(w1f[1][1], a11, 6144);
        FMMT_RD(w1f[2][0], a10, 10240); FMMT_RD(w1f[2][1], a11, 10240);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // W2(0), W1(1)
        pin_w1();
        pin_w2();
        // P1(1) under E(0): one MFMA (16 cycles of the matrix pipe) per group of VALU instructions, in program order -- left to itself the
        // compiler issues the 12 MFMAs back to back (the wave's in-order issue then stands for ~190 cycles) and the ~150 VALU behind them
        constexpr int VPM = 12;                              // VALU instructions per MFMA slot (E = ~150 VALU, 12 MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        prod1(acc1b);
        epi1(acc1a, bbf[0], pre0, hf0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        prod2(hf0);                                          // P2(0) under E(1), the same way
        epi1(acc1b, bbf[1], pre1, hf1);
        FMMT_PIN(hf1[0]); FMMT_PIN(hf1[1]); FMMT_PIN(pre1[0]); FMMT_PIN(pre1[1]);   // (keeps E(1) here: the IR-level sinking pass moves it to its first use otherwise)
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        FMMT_RD(w2f[0], a20, 6144); FMMT_RD(w2f[1], a21, 6144); FMMT_RD(w2f[2], a20, 8192); FMMT_RD(w2f[3], a21, 8192);
        FMMT_RD(w2f[4], a20, 10240); FMMT_RD(w2f[5], a21, 10240);
        // a token's 64 hidden values of this stage are one 128-byte line of h_pre / h_act: the two 64-byte halves go out back to back
        // (the stores' address arithmetic covers the latency of the W2(1) reads)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int tok = t0 + mt * 16 + li;
            if (FULL || tok < p.M) {
                const size_t off = (size_t)tok * H + hs * HS + lg * 8;
                if (p.h_pre) {
                    st_hidden(p.h_pre + off, pre0[mt]);
                    st_hidden(p.h_pre + off + 32, pre1[mt]);
                }
                if (p.h_act) {
                    st_hidden(p.h_act + off, hf0[mt]);
                    st_hidden(p.h_act + off + 32, hf1[mt]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // W2(1)
        pin_w2();
        prod2(hf1);
#undef FMMT_RD
#undef FMMT_PIN
        } else {
        // a token's 64 hidden values of this stage are one 128-byte line of h_pre / h_act: block 0's half is held back and
        // written together with block 1's, so that the two 64-byte halves reach L2 back to back
        bf16x8 keep_pre[2], keep_act[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x4 acc1[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 wf[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(sb + ks * (HS * 32) + blk * (32 * 32) + w1off[nt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt][ks], acc1[mt][nt], 0, 0, 0);
            }
            // epilogue 1: this lane holds, per token tile, hidden channels hbase .. hbase + 7 of token li
            const f32x4 bb0 = *reinterpret_cast<const f32x4*>(bs + blk * 32 + lg * 8);
            const f32x4 bb1 = *reinterpret_cast<const f32x4*>(bs + blk * 32 + lg * 8 + 4);
            bf16x8 hf[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc1[mt][0][r] + bb0[r];
                    v[4 + r] = acc1[mt][1][r] + bb1[r];
                }
                const int tok = t0 + mt * 16 + li;
                bf16x8 pre8;
                if constexpr (DG) {
                    // four values at a time, the derivative rounded at once (C = 192 sits at the register limit; the DG instantiation spills 34 registers either way and
                    // its extra ~10 VALU instructions per element cost 0.72 -> 1.12 ms on a kernel whose VALU, LDS reads and MFMAs are co-limiting: not used, ops._MLP_FUSED_DG_WIDTHS)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float d[4];
                        gelu_both_inplace<T>(v + 4 * q, d, 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) pre8[4 * q + e] = (bf16)d[e];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pre8[e] = (bf16)v[e];
                    gelu_inplace<T, C == 96 ? 4 : 2>(v, 8);  // fmmt_common.h: FMAs and one v_exp_f32, no table, no wait
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) hf[mt][e] = (bf16)v[e];
                if (blk == 0) {
                    keep_pre[mt] = pre8;
                    keep_act[mt] = hf[mt];
                } else if (FULL || tok < p.M) {
                    const size_t off = (size_t)tok * H + hs * HS + lg * 8;
                    if (p.h_pre) {
                        st_hidden(p.h_pre + off, keep_pre[mt]);
                        st_hidden(p.h_pre + off + 32, pre8);
                    }
                    if (p.h_act) {
                        st_hidden(p.h_act + off, keep_act[mt]);
                        st_hidden(p.h_act + off + 32, hf[mt]);
                    }
                }
            }
            // product 2: K block = these 32 hidden channels
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                const bf16x8 wf2 = *reinterpret_cast<const bf16x8*>(sb + blk * (C * 32) + w2off[nt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2, hf[mt], acc2[mt][nt], 0, 0, 0);
            }
        }
        }
        cslot = cslot + 1 == NBUF ? 0 : cslot + 1;
        if (++hs == NS) {
            // tile done: the x fragments are dead -- fetch the next tile's while this one's result is finished and stored
            const int next = tile + G;
            if (next < p.tiles) load_x(next);
            LinArgs e{};
            e.M = p.M;
            e.N = C;
            e.bias = p.b2;
            e.y = p.y;
            e.ldy = C;
            e.res = LN ? p.x : p.res;
            e.ldres = C;
            e.rowscale = p.rowscale;
            e.rows_per_scale = p.rows_per_scale;
            if constexpr (KEEPX) nt_epilogue<T, 2, NT2, false, true>(e, acc2, t0, 0, li, lg, &pre);
            else nt_epilogue<T, 2, NT2>(e, acc2, t0, 0, li, lg);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            hs = 0;
            tile = next;
        }
    }
}

// =============================================================================================
// Backward of the same Mlp with respect to its input, one launch:
//     dh = rowscale * (dy . W2) * gelu'(h_pre)          [M, 4C]   (stored: the two weight-gradient launches contract with it)
//     dx = dh . W1                                       [M, C]
// i.e. the forward kernel with the roles turned: product 1 contracts dy (in registers for the tile) with 64-row stages of W2^T,
// epilogue 1 multiplies by GELU' of the saved pre-activation and the DropPath scale, and its accumulator tiles are the B operand of
// product 2 against W1^T.  As two launches (fmmt_linear_fwd with FMMT_EPI_GELU_BWD, fmmt_linear_fwd) dh is written and read back:
// 1.5 GB of the 3.9 GB the pair moves at stage 0.
//
// The pre-activation tile of a stage (a token's 64 values = one 128-byte line, four 16-byte loads per lane) is the one operand that is
// neither DMA-able into the shared ring (32 KB per stage) nor free to be an ordinary load: beside LDS-DMA the compiler waits vmcnt(0)
// for every ordinary load, which would drain the ring once per stage.  It is therefore loaded by inline assembly one stage ahead and
// counted by hand: per wave and step the memory operations are, in program order,
//     [wait stage s] [barrier] DMA(s+2): CNT   AUX(s+1): 4   ... dh stores of step s: 0..4   (+ the tile end's loads and stores)
// so stage s has landed when at most CNT + 8 younger operations are outstanding (AUX(s-1), DMA(s+1), AUX(s); step 0: CNT + 4), and
// AUX(s) has landed when at most CNT + 4 are (DMA(s+2), AUX(s+1); fewer at the tail, where the waits fall back to vmcnt(0)).
// Lower bounds only: stores and tile-end operations make a wait stricter, never wrong.  The two register sets of the AUX loads
// alternate statically (the step loop is unrolled by two: a register copy of data still in flight would copy garbage).
__device__ __forceinline__ void gload16_asm(bf16x8& dst, const void* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}

__device__ __forceinline__ void gload4_asm(float& dst, const void* ptr) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
// sum over the 16 lanes of a DPP row (= the 16 tokens li of one lane group lg), result in every lane; fixed order
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

// LNB (fmmt_mlp_ln_bwd_input, C = 96): the backward of the block's norm2 as this kernel's tile epilogue.  Product 2's accumulator tile
// IS d(LN out) of the wave's 32 tokens in the layout of the LayerNorm input's fragments (token li; channels c * 32 + lg * 8 + e), the
// whole 96-channel row in the four lanes li + 16 g: the two row sums are in-lane sums + two swaps, dx = rstd (g - mean(g) - xhat
// mean(g xhat)) + dy is stored instead of d(LN out) -- the separate LayerNorm-backward launch (read d(LN out), x, dy; write dx: 1.5 GB
// at stage 0) is gone.  x / mean / rstd of the tile are requested by inline assembly at the END of the tile's last-but-one step: the
// last step's AUX wait (vmcnt(CNT + 4), or 0 at the tail) then covers them, they are older than everything it lets fly.
// d(gamma) / d(beta): in-lane products, summed over the row's 16 tokens by DPP (fixed order), each of the 48 values kept by the lane
// li == v % 16 (3 registers); per-wave slots -> fixed-order sum over the waves -> one row of partial sums per workgroup.
// DG: p.h_pre holds gelu'(pre-activation) (the forward's FMMT_SAVE_DG form): epilogue 1 is a product.
template <int C, bool LNB = false, bool DG = false>
__global__ __launch_bounds__(512) void mlp_fused_bwd_kernel(MlpArgs p) {
    using T = bf16;
    constexpr int H = 4 * C, HS = 64, NS = H / HS;
    constexpr int KS = C / 32;
    // token tiles (16 tokens) per wave: two at C = 96; one at C = 192, where two would need 96 + 48 + 32 registers for product 2's
    // accumulators, the dy fragments and the pre-activation tiles alone (it spilled 23 registers under loads counted by hand)
    constexpr int MTW = C == 96 ? 2 : 1, TW = 16 * MTW, TT = 8 * TW;
    constexpr int NT2 = C / 16, CW2 = 4 * NT2;
    constexpr int W1_EL = KS * HS * 32, W2_EL = 2 * C * 32;
    constexpr int STAGE_B = (W1_EL + W2_EL) * 2;
    constexpr int NBUF = 3;
    constexpr int NI = KS * (HS / 16) + 2 * (C / 16);
    static_assert(NI % 8 == 0 && NS % 2 == 0, "uniform DMA count per wave; an even number of steps per tile");
    constexpr int CNT = NI / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    float* gam_s = reinterpret_cast<float*>(smem + NBUF * STAGE_B);                                  // LNB: gamma [C]
    float* slot_s = reinterpret_cast<float*>(smem);                                                  // LNB, after the last step: [8 waves][4 lg][2 NV] over the ring
    if constexpr (LNB) {
        if (tid < C) gam_s[tid] = p.ln_g[tid];
    }
    __syncthreads();
    auto swz = [](int row) { return ((row >> 3) ^ (row >> 2)) & 3; };
    const int r16 = lane >> 2, c4 = lane & 3;

    // p.w1 = W2^T [4C][C] (rows: hidden), p.w2 = W1^T [C][4C] (rows: input channels): the stage layout of the forward kernel
    unsigned loff[CNT];                                      // per-piece lane offsets: see the forward kernel
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int j = i * 8 + wave;
        if (j < KS * 4) {
            const int ks = j >> 2, r = (j & 3) * 16 + r16;
            loff[i] = (unsigned)(r * C + ks * 32 + ((c4 ^ swz(r)) << 3)) * 2u;
        } else {
            const int q = j - KS * 4, b = q / (C / 16), r = (q % (C / 16)) * 16 + r16;
            loff[i] = (unsigned)(r * H + b * 32 + ((c4 ^ swz(r)) << 3)) * 2u;
        }
    }
    auto issue = [&](int slot, int hs) {
        char* base = smem + slot * STAGE_B;
        const char* u1 = reinterpret_cast<const char*>(p.w1 + (size_t)hs * HS * C);
        const char* u2 = reinterpret_cast<const char*>(p.w2 + (size_t)hs * HS);
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int j = i * 8 + wave;
            T* dst;
            if (j < KS * 4) {
                const int ks = j >> 2, r0 = (j & 3) * 16;
                dst = reinterpret_cast<T*>(base) + ks * (HS * 32) + r0 * 32;
            } else {
                const int q = j - KS * 4, b = q / (C / 16), r0 = (q % (C / 16)) * 16;
                dst = reinterpret_cast<T*>(base) + W1_EL + b * (C * 32) + r0 * 32;
            }
            unsigned lo = loff[i];
            asm volatile("" : "+v"(lo));                     // keep the offset a 32-bit register (hoisted as a zero-extended pair otherwise)
            __builtin_amdgcn_global_load_lds((gptr_t*)((j < KS * 4 ? u1 : u2) + (size_t)lo), (lptr_t*)dst, 16, 0, 0);
        }
    };

    int w1off[2], w2off[NT2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int r = chan_of<8>(nt, li >> 2, li & 3);
        w1off[nt] = r * 32 + ((lg ^ swz(r)) << 3);
    }
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
        const int r = chan_of<CW2>(nt, li >> 2, li & 3);
        w2off[nt] = W1_EL + r * 32 + ((lg ^ swz(r)) << 3);
    }

    const int G = gridDim.x;
    const int ntile = (int)blockIdx.x < p.tiles ? (p.tiles - (int)blockIdx.x + G - 1) / G : 0;
    const int nsteps = ntile * NS;
    int ihs = 0, islot = 0;
    auto issue_next = [&]() {
        issue(islot, ihs);
        islot = islot + 1 == NBUF ? 0 : islot + 1;
        ihs = ihs + 1 == NS ? 0 : ihs + 1;
    };
    // the saved pre-activation of (tile, stage hs): lane = token li of m-tile mt, hidden channels hs * 64 + blk * 32 + lg * 8 ..
    auto issue_aux = [&](int tile, int hs, bf16x8 (&dst)[2 * MTW]) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int tok = min(tile * TT + wave * TW + mt * 16 + li, p.M - 1);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) gload16_asm(dst[mt * 2 + blk], p.h_pre + (size_t)tok * H + hs * HS + blk * 32 + lg * 8);
        }
    };
    // LNB: the LayerNorm input and statistics of this wave's 32 tokens (see the kernel's header)
    bf16x8 lx[MTW][LNB ? KS : 1];
    constexpr int NV = KS * 8, NOWN = 2 * NV / 16;          // LNB: d(gamma) | d(beta) values per lane group, kept per lane
    float lmean[MTW], lrstd[MTW], own[NOWN];
#pragma unroll
    for (int i = 0; i < NOWN; ++i) own[i] = 0.f;
    auto issue_ln = [&](int tile) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int tok = min(tile * TT + wave * TW + mt * 16 + li, p.M - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) gload16_asm(lx[mt][ks], p.ln_x + (size_t)tok * C + ks * 32 + lg * 8);
            gload4_asm(lmean[mt], p.mean + tok);
            gload4_asm(lrstd[mt], p.rstd + tok);
        }
    };
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < nsteps) issue_next();

    bf16x8 xf[MTW][KS];
    float rsv[MTW];
    auto load_x = [&](int tile) {
        const int t0 = tile * TT + wave * TW;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int tok = min(t0 + mt * 16 + li, p.M - 1);
            rsv[mt] = row_scale(p.rowscale, tok, p.rows_per_scale);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[mt][ks] = *reinterpret_cast<const bf16x8*>(p.x + (size_t)tok * C + ks * 32 + lg * 8);
        }
    };
    f32x4 acc2[MTW][NT2];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int tile = blockIdx.x, hs = 0, cslot = 0, s = 0;
    bf16x8 auxA[2 * MTW], auxB[2 * MTW];
    if (ntile > 0) {
        issue_aux(tile, 0, auxA);                            // AUX(0), behind DMA(0) and DMA(1)
        load_x(tile);
    }
    auto step = [&](bf16x8 (&cur)[2 * MTW], bf16x8 (&nxt)[2 * MTW]) {
        // -- stage s landed?
        if (s + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (s == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + 2 * MTW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + 4 * MTW) : "memory");
        __builtin_amdgcn_s_barrier();
        if (s + NBUF - 1 < nsteps) issue_next();
        if (s + 1 < nsteps) {
            const int nhs = hs + 1 == NS ? 0 : hs + 1;
            issue_aux(hs + 1 == NS ? tile + G : tile, nhs, nxt);
        }
        const T* sb = reinterpret_cast<const T*>(smem + cslot * STAGE_B);
        const int t0 = tile * TT + wave * TW;
        bf16x8 keep[MTW];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x4 acc1[MTW][2];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 wf[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(sb + ks * (HS * 32) + blk * (32 * 32) + w1off[nt]);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt][ks], acc1[mt][nt], 0, 0, 0);
            }
            if (blk == 0) {
                // -- AUX(s) landed?  (younger: DMA(s+2), AUX(s+1), where issued)
                if (s + NBUF - 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + 2 * MTW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 2 * MTW; ++i) asm volatile("" : "+v"(cur[i]));
            }
            bf16x8 hf[MTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                float v[8], ax[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc1[mt][0][r];
                    v[4 + r] = acc1[mt][1][r];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) ax[e] = (float)cur[mt * 2 + blk][e];
                if constexpr (DG) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= ax[e];
                } else {
                    gelu_grad_mul_inplace<T, C == 96 ? 4 : 2>(v, ax, 8);   // v *= gelu'(pre) (fmmt_common.h)
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) hf[mt][e] = (bf16)(v[e] * rsv[mt]);
                const int tok = t0 + mt * 16 + li;
                if (blk == 0) keep[mt] = hf[mt];
                else if (tok < p.M) {
                    const size_t off = (size_t)tok * H + hs * HS + lg * 8;
                    st_hidden(p.h_act + off, keep[mt]);
                    st_hidden(p.h_act + off + 32, hf[mt]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                const bf16x8 wf2 = *reinterpret_cast<const bf16x8*>(sb + blk * (C * 32) + w2off[nt]);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2, hf[mt], acc2[mt][nt], 0, 0, 0);
            }
        }
        if constexpr (LNB) {
            if (hs == NS - 2) issue_ln(tile);
        }
        cslot = cslot + 1 == NBUF ? 0 : cslot + 1;
        ++s;
        if (++hs == NS) {
            const int next = tile + G;
            if constexpr (LNB) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    // one token tile at a time, the row's values formed twice (sums, then outputs) rather than kept: registers
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(lx[mt][ks]));
                    asm volatile("" : "+v"(lmean[mt]), "+v"(lrstd[mt]));
                    const int tok = t0 + mt * 16 + li;
                    const bool valid = tok < p.M;
                    const float mean = lmean[mt], rstd = lrstd[mt];
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int c = 0; c < KS; ++c) {
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam_s + c * 32 + lg * 8), g1 = *reinterpret_cast<const f32x4*>(gam_s + c * 32 + lg * 8 + 4);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float d = acc2[mt][2 * c + (e >> 2)][e & 3];
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
                            const float d = acc2[mt][2 * c + (e >> 2)][e & 3];
                            const float xh = ((float)lx[mt][c][e] - mean) * rstd;
                            const float gm = d * (e < 4 ? g0[e & 3] : g1[e & 3]);
                            o[e] = (bf16)(rstd * (gm - s1 - xh * s2) + (float)xf[mt][c][e]);
                            // d(gamma), d(beta): summed over the row's 16 tokens now (DPP, fixed order), kept by lane li == v % 16
                            const float dv = valid ? d : 0.f;
                            const float sg = row16_sum(dv * xh), sb = row16_sum(dv);
                            if (li == (v & 15)) own[v >> 4] += sg;
                            if (li == ((v + NV) & 15)) own[(v + NV) >> 4] += sb;
                        }
                        if (valid) *reinterpret_cast<bf16x8*>(p.y + (size_t)tok * C + c * 32 + lg * 8) = o;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (next < p.tiles) load_x(next);
            } else {
            if (next < p.tiles) load_x(next);
            LinArgs e{};
            e.M = p.M;
            e.N = C;
            e.y = p.y;
            e.ldy = C;
            nt_epilogue<T, MTW, NT2>(e, acc2, t0, 0, li, lg);
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            hs = 0;
            tile = next;
        }
    };
    while (s < nsteps) {                                     // nsteps is even (NS is)
        step(auxA, auxB);
        step(auxB, auxA);
    }
    if constexpr (LNB) {
        __syncthreads();                                     // every wave is done with the ring: the slots alias its first bytes
#pragma unroll
        for (int k = 0; k < NOWN; ++k) slot_s[(wave * 4 + lg) * (2 * NV) + k * 16 + li] = own[k];
        __syncthreads();
        if (tid < 2 * C) {
            const int kind = tid / C, ch = tid % C;
            const int v = kind * NV + (ch >> 5) * 8 + (ch & 7), lgc = (ch >> 3) & 3;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += slot_s[(w * 4 + lgc) * (2 * NV) + v];
            p.ln_part[(size_t)blockIdx.x * 2 * C + tid] = a;
        }
    }
}

// rows of per-workgroup partial sums [nblocks][2 C] -> d(gamma) [C], d(beta) [C]; fixed order: 16 row groups x 64 columns per workgroup
// (one thread per column walking all rows: 62 us for 256 x 192 values)
__global__ __launch_bounds__(1024) void mlp_ln_part_reduce_kernel(const float* __restrict__ part, int nblocks, int C, float* dgamma, float* dbeta) {
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

template <int C, bool LNB = false, bool DG = false>
int launch_mlp_bwd(const MlpArgs& a0, hipStream_t st) {
    constexpr size_t lds = (size_t)3 * (((C / 32) * 64 * 32 + 2 * C * 32) * 2) + (LNB ? C * sizeof(float) : 0);
    static_assert(lds <= 160 * 1024, "LDS");
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&mlp_fused_bwd_kernel<C, LNB, DG>), (int)lds)) return rc_;
    MlpArgs a = a0;
    constexpr int TT = C == 96 ? 256 : 128;                  // tokens per tile (the kernel's MTW)
    a.tiles = (a.M + TT - 1) / TT;
    const int grid = a.tiles < 256 ? a.tiles : 256;
    hipLaunchKernelGGL((mlp_fused_bwd_kernel<C, LNB, DG>), dim3(grid), dim3(512), lds, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

template <int C, bool LN, bool FULL, bool DG>
int launch_mlp_f(const MlpArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)3 * (((C / 32) * 64 * 32 + 2 * C * 32) * 2 + 1024);
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&mlp_fused_fwd_kernel<C, LN, FULL, DG>), (int)lds)) return rc_;
    const int grid = a.tiles < 256 ? a.tiles : 256;
    hipLaunchKernelGGL((mlp_fused_fwd_kernel<C, LN, FULL, DG>), dim3(grid), dim3(512), lds, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}
template <int C, bool LN>
int launch_mlp(const MlpArgs& a, hipStream_t st) {
    if (a.dg) return a.M % 256 == 0 ? launch_mlp_f<C, LN, true, true>(a, st) : launch_mlp_f<C, LN, false, true>(a, st);
    return a.M % 256 == 0 ? launch_mlp_f<C, LN, true, false>(a, st) : launch_mlp_f<C, LN, false, false>(a, st);
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int fmmt_mlp_fwd(int dtype, int M, int C, const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                            const void* res, const float* rowscale, int rows_per_scale, void* y, void* h_pre, void* h_act, void* stream) {
    const int el = dtype & 0xff;
    const bool generic = (dtype & FMMT_GENERIC) || el == FMMT_F32;                       // parity instantiations: csrc/mlp_ref.hip
    if ((el != FMMT_BF16 && el != FMMT_F32) || M <= 0 || (C != 96 && C != 192)) return FMMT_EINVAL;      // other widths: fmmt_linear_fwd twice
    if (!x || !w1 || !b1 || !w2 || !b2 || !y) return FMMT_EINVAL;
    if (rowscale && rows_per_scale <= 0) return FMMT_EINVAL;
    if (!al16(x) || !al16(w1) || !al16(b1) || !al16(w2) || !al16(b2) || !al16(y) || (res && !al16(res)) || (h_pre && !al16(h_pre)) || (h_act && !al16(h_act))) return FMMT_EALIGN;
    MlpArgs a{M, (const bf16*)x, (const bf16*)w1, b1, (const bf16*)w2, b2, (const bf16*)res, rowscale, rows_per_scale, (bf16*)y, (bf16*)h_pre, (bf16*)h_act, (M + 255) / 256};
    a.dg = (dtype & FMMT_SAVE_DG) ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (generic) return fmmt_mlp_ref_fwd_launch(el, C, false, a, st);
    return C == 96 ? launch_mlp<96, false>(a, st) : launch_mlp<192, false>(a, st);
}

extern "C" int fmmt_mlp_ln_fwd(int dtype, int M, int C, const void* x, const float* ln_gamma, const float* ln_beta, float eps,
                               const void* w1, const float* b1, const void* w2, const float* b2, const float* rowscale, int rows_per_scale,
                               void* y, void* xn, float* mean, float* rstd, void* h_pre, void* h_act, void* stream) {
    const int el = dtype & 0xff;
    const bool generic = (dtype & FMMT_GENERIC) || el == FMMT_F32;
    if ((el != FMMT_BF16 && el != FMMT_F32) || M <= 0 || (C != 96 && C != 192)) return FMMT_EINVAL;
    if (!x || !ln_gamma || !ln_beta || !w1 || !b1 || !w2 || !b2 || !y) return FMMT_EINVAL;
    if ((mean == nullptr) != (rstd == nullptr) || (rowscale && rows_per_scale <= 0)) return FMMT_EINVAL;
    if (!al16(x) || !al16(ln_gamma) || !al16(ln_beta) || !al16(w1) || !al16(b1) || !al16(w2) || !al16(b2) || !al16(y) || (xn && !al16(xn)) ||
        (h_pre && !al16(h_pre)) || (h_act && !al16(h_act))) return FMMT_EALIGN;
    MlpArgs a{M, (const bf16*)x, (const bf16*)w1, b1, (const bf16*)w2, b2, nullptr, rowscale, rows_per_scale, (bf16*)y, (bf16*)h_pre, (bf16*)h_act, (M + 255) / 256,
              ln_gamma, ln_beta, eps, (bf16*)xn, mean, rstd};
    a.dg = (dtype & FMMT_SAVE_DG) ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (generic) return fmmt_mlp_ref_fwd_launch(el, C, true, a, st);
    return C == 96 ? launch_mlp<96, true>(a, st) : launch_mlp<192, true>(a, st);
}

extern "C" int fmmt_mlp_bwd_input(int dtype, int M, int C, const void* dy, const void* h_pre, const void* w2t, const void* w1t,
                                  const float* rowscale, int rows_per_scale, void* dh, void* dx, void* stream) {
    const int el = dtype & 0xff;
    const bool generic = (dtype & FMMT_GENERIC) || el == FMMT_F32;
    if ((el != FMMT_BF16 && el != FMMT_F32) || M <= 0 || (C != 96 && C != 192)) return FMMT_EINVAL;
    if (!dy || !h_pre || !w2t || !w1t || !dh || !dx || (rowscale && rows_per_scale <= 0)) return FMMT_EINVAL;
    if (!al16(dy) || !al16(h_pre) || !al16(w2t) || !al16(w1t) || !al16(dh) || !al16(dx)) return FMMT_EALIGN;
    MlpArgs a{};
    a.M = M; a.x = (const bf16*)dy; a.w1 = (const bf16*)w2t; a.w2 = (const bf16*)w1t; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale;
    a.y = (bf16*)dx; a.h_pre = (bf16*)const_cast<void*>(h_pre); a.h_act = (bf16*)dh; a.tiles = (M + 255) / 256;
    a.dg = (dtype & FMMT_SAVE_DG) ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (generic) return fmmt_mlp_ref_bwd_launch(el, C, false, a, st);
    if (a.dg) return C == 96 ? launch_mlp_bwd<96, false, true>(a, st) : launch_mlp_bwd<192, false, true>(a, st);
    return C == 96 ? launch_mlp_bwd<96>(a, st) : launch_mlp_bwd<192>(a, st);
}

extern "C" size_t fmmt_mlp_ln_bwd_input_workspace(int C) { return (size_t)256 * 2 * C * sizeof(float); }

extern "C" int fmmt_mlp_ln_bwd_input(int dtype, int M, int C, const void* dy, const void* h_pre, const void* w2t, const void* w1t,
                                     const float* rowscale, int rows_per_scale, const void* x, const float* mean, const float* rstd,
                                     const float* ln_gamma, void* dh, void* dx, float* dgamma, float* dbeta, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    const int el = dtype & 0xff;
    const bool generic = (dtype & FMMT_GENERIC) || el == FMMT_F32;
    if ((el != FMMT_BF16 && el != FMMT_F32) || M <= 0 || (C != 96 && C != 192)) return FMMT_EINVAL;       // other widths: fmmt_mlp_bwd_input + fmmt_layernorm_bwd
    if (!dy || !h_pre || !w2t || !w1t || !dh || !dx || !x || !mean || !rstd || !ln_gamma || !dgamma || !dbeta || !workspace) return FMMT_EINVAL;
    if (rowscale && rows_per_scale <= 0) return FMMT_EINVAL;
    if (workspace_bytes < fmmt_mlp_ln_bwd_input_workspace(C)) return FMMT_EWORKSPACE;
    if (!al16(dy) || !al16(h_pre) || !al16(w2t) || !al16(w1t) || !al16(dh) || !al16(dx) || !al16(x) || !al16(workspace)) return FMMT_EALIGN;
    MlpArgs a{};
    a.M = M; a.x = (const bf16*)dy; a.w1 = (const bf16*)w2t; a.w2 = (const bf16*)w1t; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale;
    a.y = (bf16*)dx; a.h_pre = (bf16*)const_cast<void*>(h_pre); a.h_act = (bf16*)dh; a.tiles = (M + 255) / 256;
    a.ln_g = ln_gamma; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.ln_x = (const bf16*)x; a.ln_part = (float*)workspace;
    a.dg = (dtype & FMMT_SAVE_DG) ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (generic) {
        if (int rc = fmmt_mlp_ref_bwd_launch(el, C, true, a, st)) return rc;
        const int g = a.tiles < 256 ? a.tiles : 256;                                    // the generic kernel's tiles are 256 tokens at either width
        hipLaunchKernelGGL(mlp_ln_part_reduce_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, st, (const float*)workspace, g, C, dgamma, dbeta);
        FMMT_CHECK_LAUNCH();
        return 0;
    }
    if (int rc = a.dg ? (C == 96 ? launch_mlp_bwd<96, true, true>(a, st) : launch_mlp_bwd<192, true, true>(a, st))
                      : (C == 96 ? launch_mlp_bwd<96, true>(a, st) : launch_mlp_bwd<192, true>(a, st))) return rc;
    const int tiles = (M + (C == 96 ? 256 : 128) - 1) / (C == 96 ? 256 : 128), grid = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL(mlp_ln_part_reduce_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, st, (const float*)workspace, grid, C, dgamma, dbeta);
    FMMT_CHECK_LAUNCH();
    return 0;
}
