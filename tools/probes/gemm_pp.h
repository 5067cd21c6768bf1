// Persistent NT GEMM, round 5: two-phase wave ping-pong on v_mfma_f32_32x32x16_bf16 with DMA issue and the previous tile's epilogue INSIDE the MFMA segments.
//
//   y[M,N] = epi( x[M,K] . w[N,K]^T + bias )            bf16 operands, fp32 accumulate; Swin stages 1-3 (Swin_Transformer.py:19-28,105-107,142,304)
//
// What round 4's kernel (linear_nt_p256_kernel) left on the table, from its own s_memtime stamps: its eight waves pass ONE barrier per K step together, so
// they all issue DMA, then all read fragments, then all feed the matrix cores -- the three phases add (~3600 cycles per 64-deep step for 1536 cycles of
// MFMA work), and a finished tile's conversion + stores (a quarter of a K = 384 tile's life) run with the matrix cores idle.  Here:
//   * one workgroup of 8 waves per CU walks 256-token x BN-channel tiles; waves 0-3 (one per SIMD) own token rows 0-127, waves 4-7 (their SIMD partners)
//     rows 128-255; wave tile 64 tokens x BN/2 channels = 2 x (BN/64) accumulator tiles of 32 x 32.
//   * a K step is 32 deep.  Every wave alternates a LOAD segment -- nothing but the fragment reads of this step (2 x (2 + BN/64) ds_read_b128) and the counted
//     wait for the next stage -- with an MFMA segment: 2 x 2 x BN/64 MFMAs at priority 1 and, in the issue slots the matrix pipe leaves free (profiles/
//     r05_issue_rates.txt: <= 5 plain single-issue instructions per 32x32x16 MFMA are hidden on this SIMD), the DMA instructions of the stage NBUF - 1 steps
//     ahead and one slice of the PREVIOUS tile's epilogue.  One raw s_barrier between segments; waves 4-7 run ONE segment behind waves 0-3, so on every
//     SIMD one wave feeds the matrix pipe while its partner reads fragments.
//     (First form of this kernel, same round: DMA issue and epilogue slices in the LOAD segment -- LOAD 1290 cycles against MFMA 530 per K step, 0.8x the
//     round-4 kernel: the LOAD segment is the critical path, so everything that can run beside MFMAs has to.)
//   * ring of NBUF stages of (BN + 256) rows x 64 B filled by global_load_lds (swizzled on the source side, 16-byte chunk ^ ((row >> 2) & 3): every
//     16-lane group of a fragment read meets 16 distinct 16-byte slots); NBUF - 1 stages in flight (86 KB at BN = 192 against 57 KB before), counted vmcnt.
//   * weight rows sit in LDS in MFMA-row order such that a lane's 16 accumulator registers are 16 CONSECUTIVE output channels (32 B of bf16) and the
//     two half-waves hold adjacent halves: chan(a) = 16 ((a >> 2) & 1) + 4 (a >> 3) + (a & 3) for MFMA row a.
//   * epilogue: when a tile's last MFMA segment is done the wave adds the bias (from an LDS slab that travels with the ring), rounds to bf16 and keeps
//     the tile in 8 x BN/64 x 2 packed registers (next LOAD segment); over the next five MFMA segments 32-token-row passes go through three rotating LDS
//     slabs -- every wave writes 16 of its rows per pass, all four waves of the group read the pass back row-major -- and leave as whole 128-byte lines.
//     The MFMA segment is ONE basic block whatever the state of the epilogue: slab reads, stores and slab writes are issued in EVERY segment, steered by
//     wave-uniform values -- lanes / segments with nothing to write go to a trash strip of LDS, stores with nothing to store carry a byte offset beyond the
//     buffer descriptor's size and are dropped by the hardware (rows past M of a ragged last panel likewise) -- so the MFMAs exist once in the loop: with
//     them duplicated over the arms of a switch the compiler left the accumulators un-coalesced (192 registers + 48 moves, 150-240 spilled registers).
//     GELU (+ the bf16 pre-activation as a second tensor) is applied on the read side, i.e. on the bf16-rounded pre-activation -- what the backward
//     differentiates and what a bf16 autocast of the reference computes.
// Requirements: M % 8 == 0, N % BN == 0, K % 32 == 0, K >= 192 (six K steps: five to drain an epilogue under), 32-bit byte offsets into x, w and y.
#pragma once
#include <type_traits>
#include "gemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int V> using pp_ic = std::integral_constant<int, V>;

template <int BN, int NBUF>
struct PpGeom {
    static constexpr int BM = 256, BK = 32, ROWB = 64;         // bytes per stage row
    static constexpr int ROWS = BN + BM, STAGE = ROWS * ROWB;  // bytes per ring stage: BN weight rows, then 256 token rows
    static constexpr int NT = BN / 64, MT = 2, HB = BN / 2;    // accumulator tiles per wave (channels x tokens), channels per wave column
    static constexpr int SP = BN * 2 + 16, SLAB = 32 * SP;     // epilogue slab: 32 token rows, pitch = row + 16 B
    static constexpr int RING = NBUF * STAGE, BIAS_OFF = RING, SLAB_OFF = RING + 2 * 1024, TRASH_OFF = SLAB_OFF + 3 * SLAB, LDS = TRASH_OFF + 16 * SP;
    static constexpr int D = NBUF - 1;                         // stages in flight
    static_assert(LDS <= 160 * 1024, "LDS");
    static_assert(BN % 64 == 0 && BN <= 256, "channel tile");
    static_assert(D == 3, "the mark queue of the kernel is written for three stages in flight");
};

__device__ __forceinline__ void pp_wait_vm_dyn(int n) {      // s_waitcnt vmcnt(n), n wave-uniform; larger n than the table holds: the table's maximum (stricter)
    switch (n) {
#define FMMT_PW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        FMMT_PW(0) FMMT_PW(1) FMMT_PW(2) FMMT_PW(3) FMMT_PW(4) FMMT_PW(5) FMMT_PW(6) FMMT_PW(7) FMMT_PW(8) FMMT_PW(9) FMMT_PW(10) FMMT_PW(11)
        FMMT_PW(12) FMMT_PW(13) FMMT_PW(14) FMMT_PW(15) FMMT_PW(16) FMMT_PW(17) FMMT_PW(18) FMMT_PW(19) FMMT_PW(20) FMMT_PW(21) FMMT_PW(22) FMMT_PW(23)
        FMMT_PW(24) FMMT_PW(25) FMMT_PW(26) FMMT_PW(27) FMMT_PW(28)
#undef FMMT_PW
        default: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    }
}

template <int OFF> __device__ __forceinline__ void pp_rd(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N - 1 (inline-asm immediates must be constants, not unrolled loop variables)
template <int I, int N, typename F> __device__ __forceinline__ void pp_for(F&& f) {
    if constexpr (I < N) {
        f(pp_ic<I>{});
        pp_for<I + 1, N>(f);
    }
}

// GELU of 8 values with PLAIN single-issue FMAs (one element per instruction): beside MFMAs a packed v_pk_fma_f32 costs ~12 cycles of matrix-pipe time each,
// a plain v_fma_f32 ~0.5 (profiles/r05_issue_rates.txt) -- same polynomial, same coefficients as fmmt_common.h's packed form
__device__ __forceinline__ void pp_gelu8(float (&v)[8]) {
    constexpr int N = sizeof(fmmt_gelu_phi_poly) / sizeof(float);
    constexpr float R = FMMT_GELU_PHI_R;
    float xc[8], u[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xc[j] = __builtin_amdgcn_fmed3f(v[j], -R, R);
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = __builtin_fmaf(xc[j] * xc[j], 2.0f / (R * R), -1.0f);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = __builtin_fmaf(u[j], fmmt_gelu_phi_poly[N - 1], fmmt_gelu_phi_poly[N - 2]);
#pragma unroll
    for (int k = N - 3; k >= 0; --k)
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = __builtin_fmaf(q[j], u[j], fmmt_gelu_phi_poly[k]);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * __builtin_fmaf(q[j], xc[j], 0.5f);
}

// EPI: 0 = no output (probe of the main loop alone), 1 = plain / bias, 2 = GELU with the pre-activation stored to y_pre (both tensors ld = ldy)
// SGB: sched_group_barrier pipeline inside the MFMA segments (A/B switch of the probe); PRIO: s_setprio 1 around them
template <int BN, int NBUF, int EPI = 1, bool TRACE = false, bool SGB = true, bool PRIO = true, bool NODMA = false>
__global__ __launch_bounds__(512) void linear_nt_pp_kernel(LinArgs p) {
    using Gm = PpGeom<BN, NBUF>;
    constexpr int NT = Gm::NT, MT = Gm::MT, HB = Gm::HB, SP = Gm::SP, STAGE = Gm::STAGE, D = Gm::D;
    constexpr int WP = BN / 16;                                // weight pieces (16 rows x 64 B = 1 KB per DMA instruction) per stage
    constexpr int NSL = BN / 64;                               // 128-byte lines per output row = store instructions per lane and slab pass
    constexpr int NSTORE = EPI == 2 ? 2 * NSL : (EPI == 1 ? NSL : 0);
    // weight pieces per wave and stage: NWC for every wave (piece `wave`, + 8) inside the MFMA segment and, at BN = 192, one more for the waves of
    // group 0 (piece wave + 8), issued at the head of their NEXT LOAD segment (the MFMA segment must be the same code for both groups)
    constexpr int NWC = WP >= 16 ? 2 : 1;
    constexpr bool XTRA = WP > 8 * NWC;
    static_assert(WP == 8 || WP == 12 || WP == 16, "weight pieces per stage");
    constexpr unsigned OOB = 0xC0000000u;                      // store offset of "nothing to store": beyond any descriptor size this kernel accepts
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wml = (wave >> 1) & 1, wn = wave & 1, w4 = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const char* __restrict__ xg = reinterpret_cast<const char*>(p.x);
    const char* __restrict__ wg = reinterpret_cast<const char*>(p.w);
    const int nk = p.K / 32;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)smem;
    float* bias_s = reinterpret_cast<float*>(smem + Gm::BIAS_OFF);             // [2][256]

    // ---- tile schedule: workgroup b (observed on XCD b % 8) takes, in round i, tile i * G + (b % 8) * (G / 8) + b / 8 ----
    const int G = gridDim.x;
    const int total = p.tiles_m * p.tiles_n;
    const int first = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    const int ntile = first < total ? (total - first + G - 1) / G : 0;
    const int nsteps = ntile * nk;
    if (nsteps == 0) return;

    // ---- DMA pieces of this wave: weight pieces wave, wave + 8 (if they exist); token pieces (its group's half) 2 (wave & 3), + 1 ----
    const int rl = lane >> 2, cl = lane & 3;
    unsigned poff[4], boff = 0;
    auto tile_offsets = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (wave + 8 * i) * 16 + rl;                                   // LDS weight row = MFMA row a = r & 31 of 32-channel tile r >> 5
            const int a = r & 31;
            const int ch = n0 + (r & ~31) + 16 * ((a >> 2) & 1) + 4 * (a >> 3) + (a & 3);
            poff[i] = ((unsigned)min(ch, p.N - 1) * (unsigned)p.ldw + (unsigned)((cl ^ ((r >> 2) & 3)) << 3)) * 2u;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int xr = grp * 128 + (w4 * 2 + i) * 16 + rl;
            const int gm = min(m0 + xr, p.M - 1);                                     // ragged last panel: a valid row (never stored)
            poff[2 + i] = ((unsigned)gm * (unsigned)p.ldx + (unsigned)((cl ^ ((xr >> 2) & 3)) << 3)) * 2u;
        }
        boff = p.bias ? (unsigned)(n0 + min(lane * 4, BN - 4)) * 4u : (unsigned)lane * 16u;
    };
    // issue-side cursor (wave-uniform).  It stops ON the last stage: the MFMA segments of the job's last D steps re-issue that stage into its own slot
    // (the same bytes over themselves: harmless to a concurrent fragment read) rather than carry a branch.
    int it = first, ik = 0, ipar = 0, islot = 0, istage = 0;
    bool new_tile = true;                                      // offsets (and the bias slab) of tile `it` still to be set up
    int vm_total = 0, mq0 = 0, mq1 = 0;                        // vector-memory instructions issued so far; its value right after the last DMA piece of stage s + 1 (mq1: prologue only)
    auto setup_tile = [&]() {                                  // LOAD segment / prologue: offsets of tile `it`, its bias slab on the way (wave 7)
        tile_offsets((it / p.tiles_n) * 256, (it % p.tiles_n) * BN);
        if (wave == 7) {
            const char* b = p.bias ? reinterpret_cast<const char*>(p.bias) : wg;
            __builtin_amdgcn_global_load_lds((gptr_t*)(b + (size_t)boff), (lptr_t*)(bias_s + ipar * 256), 16, 0, 0);
            ++vm_total;
        }
        new_tile = false;
    };
    unsigned xkbyte = 0;                                       // extra piece pending (group 0, BN = 192): its k offset and stage base
    char* xsb = smem;
    bool xpend = false;
    auto issue_pieces = [&]() {                                // the common DMA pieces of stage `istage`: branch-free
        const unsigned kbyte = (unsigned)ik * 64u;
        char* sb = smem + islot * STAGE;
        if constexpr (!NODMA) {                                // (NODMA: probe of the loop without its operand traffic -- the product never instantiates it)
#pragma unroll
            for (int i = 0; i < NWC; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t*)(wg + (size_t)(poff[i] + kbyte)), (lptr_t*)(sb + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t*)(xg + (size_t)(poff[2 + i] + kbyte)), (lptr_t*)(sb + (BN + grp * 128 + (w4 * 2 + i) * 16) * 64), 16, 0, 0);
            vm_total += NWC + 2;
        } else {
            asm volatile("" ::"v"(poff[0]), "v"(poff[2]), "v"(poff[3]), "s"(kbyte), "s"(sb));
        }
        if constexpr (XTRA && !NODMA) {
            xkbyte = kbyte;
            xsb = sb;
            xpend = true;
        }
        const bool adv = istage + 1 < nsteps;                  // the cursor stays on the last stage
        const bool wrap = adv && ik + 1 == nk;
        islot = adv ? (islot + 1 == NBUF ? 0 : islot + 1) : islot;
        istage += adv ? 1 : 0;
        ik = wrap ? 0 : (adv ? ik + 1 : ik);
        it += wrap ? G : 0;
        ipar ^= wrap ? 1 : 0;
        new_tile = new_tile || wrap;
    };
    auto issue_extra = [&]() {                                 // LOAD segment / prologue, BEFORE setup_tile() moves poff to the next tile
        if constexpr (XTRA) {
            if (xpend && wave + 8 * NWC < WP) {
                __builtin_amdgcn_global_load_lds((gptr_t*)(wg + (size_t)(poff[NWC] + xkbyte)), (lptr_t*)(xsb + (wave + 8 * NWC) * 1024), 16, 0, 0);
                ++vm_total;
            }
            xpend = false;
        }
    };

    // ---- fragment addresses (byte offsets inside a stage): row (lane & 31) of a 32-row tile, 16-byte chunk 2 kk + (lane >> 5), swizzled ----
    const unsigned fsw = (unsigned)(l31 * 64 + ((lh ^ ((l31 >> 2) & 3)) << 4));      // kk = 0; kk = 1 flips chunk bit 1: ^ 32
    const unsigned wad = lds0 + (unsigned)(wn * HB * 64) + fsw;
    const unsigned xad = lds0 + (unsigned)((BN + grp * 128 + wml * 64) * 64) + fsw;

    // ---- epilogue addressing ----
    // write side, pass q = 2 tm + half: lanes with ((lane & 31) >> 4) == half park rows 16 half + (lane & 15) of their token tile tm in slab row 16 wml + (lane & 15);
    // the other lanes (every lane when no pass is due) write the same shape into the trash strip.  Read side: slab row er = 8 (wave & 3) + lane / 8,
    // 16-byte chunk lane % 8 of line i of that row.
    const unsigned wrow = (unsigned)((wml * 16 + (l31 & 15)) * SP + (wn * HB + 16 * lh) * 2);
    const unsigned trow = (unsigned)Gm::TRASH_OFF + (unsigned)((l31 & 15) * SP + (wn * HB + 16 * lh) * 2);
    const bool hi16 = (l31 >> 4) != 0;
    const int er = w4 * 8 + (lane >> 3);
    const int erow = (er >> 4) * 64 + (er & 15);              // row of the group's 128 this slab row holds, before the pass offset 32 tm + 16 half
    const unsigned esl = (unsigned)(er * SP + (lane & 7) * 16);
    const unsigned eyl = ((unsigned)erow * (unsigned)p.ldy + (unsigned)(lane & 7) * 8u) * 2u;
    const unsigned ybytes = (unsigned)p.M * (unsigned)p.ldy * 2u;
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, ybytes, 0x00020000);
    const auto prs = __builtin_amdgcn_make_buffer_rsrc(p.y_pre ? p.y_pre : p.y, 0, p.y_pre ? ybytes : 0u, 0x00020000);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    bf16x8 wf[2][NT], xf[2][MT];
    u32x4 pk[MT][NT][2];                                       // the finished tile, bias added, bf16: [token tile][channel tile][16-byte half]
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) pk[a][b][0] = pk[a][b][1] = u32x4{0u, 0u, 0u, 0u};

    // compute-side cursor
    int ct = first, ck = 0, cpar = 0, cslot = 0;
    int ph3 = grp;                                             // (global phase of this wave's MFMA segment) mod 3: slab it writes; it reads (ph3 + 1) mod 3
    int ep_step = 5, em0 = 0, en0 = 0;                         // epilogue in progress: slice index (5 = none), origin of the parked tile (this group's 128 rows)
    float tr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- the epilogue slice of one MFMA segment, steered by wave-uniform values (no branch) ----
    auto slab_read = [&](bf16x8 (&v)[NSL]) {
        if constexpr (EPI != 0) {
            const int pr = ph3 == 2 ? 0 : ph3 + 1;
            const char* slab_r = smem + Gm::SLAB_OFF + pr * Gm::SLAB + esl;
#pragma unroll
            for (int i = 0; i < NSL; ++i) v[i] = *reinterpret_cast<const bf16x8*>(slab_r + i * 128);
        }
    };
    auto slab_store = [&](bf16x8 (&v)[NSL], unsigned st_off) {   // st_off: this lane's byte offset of line 0 (>= the descriptor's size: dropped)
        if constexpr (EPI == 2) {
#pragma unroll
            for (int i = 0; i < NSL; ++i) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), prs, st_off + i * 128, 0, 0);
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (float)v[i][e];
                pp_gelu8(t);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = (bf16)t[e];
            }
        }
        if constexpr (EPI != 0) {
#pragma unroll
            for (int i = 0; i < NSL; ++i) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), yrs, st_off + i * 128, 0, 0);
        }
    };
    auto slab_write = [&](bool tm1, unsigned wr_hi, unsigned wr_lo) {   // wr_hi / wr_lo: LDS byte base for the lanes of the upper / lower 16 rows
        if constexpr (EPI != 0) {
            char* row = smem + (hi16 ? wr_hi : wr_lo);
#pragma unroll
            for (int b = 0; b < NT; ++b) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 d;
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = tm1 ? pk[1][b][h][e] : pk[0][b][h][e];
                    *reinterpret_cast<u32x4*>(row + b * 64 + h * 16) = d;
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < NT; ++b) asm volatile("" ::"v"(pk[0][b][0]), "v"(pk[0][b][1]), "v"(pk[1][b][0]), "v"(pk[1][b][1]));
        }
    };
    auto park = [&]() {                                        // the tile this wave's last MFMA segment completed: bias, round to bf16, keep; restart the accumulators
        const float* bs = bias_s + cpar * 256 + wn * HB + 16 * lh;
        auto go = [&](auto has_bias) {
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                f32x4 bb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) bb[q] = decltype(has_bias)::value ? *reinterpret_cast<const f32x4*>(bs + b * 32 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < MT; ++a) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        bf16x8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (bf16)(acc[a][b][h * 8 + e] + bb[h * 2 + (e >> 2)][e & 3]);
                        pk[a][b][h] = __builtin_bit_cast(u32x4, v);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
                }
            }
        };
        if (p.bias) go(std::true_type{});
        else go(std::false_type{});
        const int pt = ct - G;
        em0 = (pt / p.tiles_n) * 256 + grp * 128;
        en0 = (pt % p.tiles_n) * BN;
        cpar ^= 1;
        ep_step = 0;
    };
    // the values that steer the slice of this wave's next MFMA segment, from ep_step
    unsigned st_off = OOB, wr_hi = trow, wr_lo = trow;
    bool tm1 = false;
    auto steer = [&]() {
        const int q = ep_step - 1;                             // pass to read back and store (0 .. 3)
        const bool rd = ep_step >= 1 && ep_step <= 4;
        const unsigned base = ((unsigned)(em0 + (q >> 1) * 32 + (q & 1) * 16) * (unsigned)p.ldy + (unsigned)en0) * 2u;
        st_off = rd ? base + eyl : OOB;
        const bool wr = ep_step <= 3;                          // pass ep_step to write: token tile ep_step >> 1, row half ep_step & 1
        const unsigned sw = (unsigned)(Gm::SLAB_OFF + ph3 * Gm::SLAB) + wrow;
        wr_hi = (wr && (ep_step & 1) == 1) ? sw : trow;
        wr_lo = (wr && (ep_step & 1) == 0) ? sw : trow;
        tm1 = (ep_step >> 1) == 1;
    };

    // ---- prologue: D stages in flight, stage 0 landed for everybody ----
    {
        int mk0 = 0;
#pragma unroll
        for (int s = 0; s < D; ++s) {
            if (s < nsteps) {
                if (new_tile) setup_tile();
                issue_pieces();
                issue_extra();
            }
            if (s == 0) mk0 = vm_total;
            if (s == 1) mq0 = vm_total;
            if (s == 2) mq1 = vm_total;
        }
        pp_wait_vm_dyn(vm_total - mk0);
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();                // waves 4-7 run one segment behind
    __builtin_amdgcn_sched_barrier(0);

    for (int s = 0; s < nsteps; ++s) {
        // ======================= LOAD segment =======================
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
        if constexpr (TRACE) t0 = __builtin_amdgcn_s_memtime();
        issue_extra();                                         // (the offsets still belong to the tile of the stage MFMA segment s - 1 issued)
        const int mark_new = vm_total;                         // everything up to here is not younger than the pieces of stage s - 1 + D
        if (new_tile) setup_tile();
        if (ck == 0 && s > 0) park();
        steer();
        if constexpr (TRACE) t1 = __builtin_amdgcn_s_memtime();
        {
            const unsigned sb = (unsigned)cslot * (unsigned)STAGE;
            const unsigned w0 = wad + sb, w1 = (wad ^ 32u) + sb, x0 = xad + sb, x1 = (xad ^ 32u) + sb;
            pp_for<0, NT>([&](auto I) { pp_rd<I.value * 2048>(wf[0][I.value], w0); });
            pp_for<0, MT>([&](auto I) { pp_rd<I.value * 2048>(xf[0][I.value], x0); });
            pp_for<0, NT>([&](auto I) { pp_rd<I.value * 2048>(wf[1][I.value], w1); });
            pp_for<0, MT>([&](auto I) { pp_rd<I.value * 2048>(xf[1][I.value], x1); });
        }
        if constexpr (TRACE) t2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TRACE) t3 = __builtin_amdgcn_s_memtime();
        if (s + 1 < nsteps) pp_wait_vm_dyn(vm_total - mq0);    // this wave's pieces of stage s + 1 have landed
        mq0 = s == 0 ? mq1 : mark_new;                         // (stage 2's mark comes from the prologue)
#pragma unroll
        for (int b = 0; b < NT; ++b) asm volatile("" : "+v"(wf[0][b]), "+v"(wf[1][b]));
#pragma unroll
        for (int a = 0; a < MT; ++a) asm volatile("" : "+v"(xf[0][a]), "+v"(xf[1][a]));
        if constexpr (TRACE) t4 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ======================= MFMA segment: one basic block =======================
        if constexpr (TRACE) t5 = __builtin_amdgcn_s_memtime();
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
        {
            bf16x8 v[NSL];
            slab_read(v);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][b], xf[kk][a], acc[a][b], 0, 0, 0);
            issue_pieces();
            slab_write(tm1, wr_hi, wr_lo);
            slab_store(v, st_off);
            if constexpr (SGB) {
#pragma unroll
                for (int i = 0; i < 2 * MT * NT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x296, 5, 0);       // up to five of: VALU, SALU, VMEM, DS
                }
            }
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        vm_total += NSTORE;                                    // (counted as older than this segment's DMA: the scheduler may have put them there)
        cslot = cslot + 1 == NBUF ? 0 : cslot + 1;
        {
            const bool wrap = ck + 1 == nk;
            ck = wrap ? 0 : ck + 1;
            ct += wrap ? G : 0;
        }
        ep_step = ep_step < 5 ? ep_step + 1 : 5;
        ph3 = ph3 == 0 ? 2 : ph3 - 1;                           // + 2 mod 3
        if constexpr (TRACE) t6 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TRACE) {
            const unsigned long long t7 = __builtin_amdgcn_s_memtime();
            tr[0] += (float)(t1 - t0); tr[1] += (float)(t2 - t1); tr[2] += (float)(t3 - t2); tr[3] += (float)(t4 - t3);
            tr[4] += (float)(t5 - t4); tr[5] += (float)(t6 - t5); tr[6] += (float)(t7 - t6); tr[7] += 1.0f;
        }
    }
    // ---- drain: the last tile's epilogue, five more segment pairs without MFMAs (the two groups keep alternating: the slab rotation relies on it) ----
    for (int s = 0; s < 5; ++s) {
        if (s == 0) park();
        steer();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {
            bf16x8 v[NSL];
            slab_read(v);
            slab_store(v, st_off);
            slab_write(tm1, wr_hi, wr_lo);
        }
        ep_step = ep_step < 5 ? ep_step + 1 : 5;
        ph3 = ph3 == 0 ? 2 : ph3 - 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the re-issued last stage must not land in a successor workgroup's LDS
    if constexpr (TRACE) {
        if (lane == 0 && p.part) {
            float* o = p.part + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = tr[i];
        }
    }
}

template <int BN, int NBUF, int EPI = 1, bool TRACE = false, bool SGB = true, bool PRIO = true, bool NODMA = false>
int launch_pp(const LinArgs& a, hipStream_t st, int grid = 256) {
    using Gm = PpGeom<BN, NBUF>;
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&linear_nt_pp_kernel<BN, NBUF, EPI, TRACE, SGB, PRIO, NODMA>), Gm::LDS)) return rc_;
    LinArgs p = a;
    p.tiles_m = (a.M + 255) / 256;
    p.tiles_n = a.N / BN;
    hipLaunchKernelGGL((linear_nt_pp_kernel<BN, NBUF, EPI, TRACE, SGB, PRIO, NODMA>), dim3(grid), dim3(512), Gm::LDS, st, p);
    FMMT_CHECK_LAUNCH();
    return 0;
}

}  // namespace
