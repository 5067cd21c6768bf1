// Persistent NT GEMM, round 5 (second form): 256 x 256 tiles, K step 64 cut into FOUR phases of 16 MFMAs, two wave groups half a phase apart.
//
//   y[M,N] = x[M,K] . w[N,K]^T + bias            bf16 operands, fp32 accumulate; Swin stages 1-3 (Swin_Transformer.py:19-28,105-107,142,304)
//
// Why another main loop.  linear_nt_p256_kernel passes ONE barrier per K step with all eight waves in the same phase (DMA issue, fragment reads, MFMAs
// add up: ~3600 cycles per step for 1536 of matrix-pipe work); the first ping-pong form (tools/probes/gemm_pp.h: 32-deep steps on 32x32x16 MFMAs, every fragment of
// a step read in one LOAD segment) measured no better -- its LOAD segment, not the MFMA segment, was the critical path.  This form follows the phase
// structure the CDNA4 guide reports at 1.3-1.5 PF/s on square problems:
//   * one workgroup of 8 waves per CU, tile 256 tokens x 256 channels, waves 2 (tokens) x 4 (channels), wave tile 128 x 64 = 8 x 4 accumulator tiles of
//     v_mfma_f32_16x16x32_bf16 (128 accumulator registers);
//   * a K step (64 deep) is four phases, each the 16 MFMAs of one 64-token x 32-channel QUADRANT of the wave tile over the whole step:
//         P1: tokens 0-63 x channels 0-31    reads 8 token + 4 channel fragments        P3: tokens 64-127 x channels 32-63   reads 8 token fragments
//         P2: tokens 0-63 x channels 32-63   reads 4 channel fragments                  P4: tokens 64-127 x channels 0-31    reads nothing
//     a phase = [ds_read_b128 of the fragments it needs; 2 DMA instructions] s_barrier [lgkmcnt(0); 16 MFMAs at priority 1] s_barrier;
//   * waves 4-7 (token rows 128-255) run ONE BARRIER behind waves 0-3: on every SIMD one wave feeds the matrix pipe while its partner issues LDS reads
//     and DMA -- the phases of the two groups interleave instead of adding;
//   * LDS: two buffers of four 16 KB half-tiles (channel rows 0-127 / 128-255, token rows 0-127 / 128-255; 128-byte rows, 16-byte chunk XOR ((row >> 1) & 7)
//     applied on the DMA's source side), one half-tile staged per phase by global_load_lds (2 instructions per wave), in the order that the write-after-read
//     distance allows (a slot is re-staged two phases after its last fragment read):  P4(s-1): W0(s+1) | P1(s): X0(s+1) | P2(s): X1(s+1) | P3(s): W1(s+1);
//     ONE counted vmcnt per K step (in P4: everything but P4's own two instructions has landed);
//   * the pipeline is flat over (tile, K step): the staging cursor runs into the workgroup's next tile while the current one is still being multiplied.
// EPI 0: no output (main loop alone); 1: bias + bf16, whole 128-byte lines through wave-private LDS slabs, all at the tile boundary;
// 2: the same rows spread over the phases around the tile boundary (drip), non-temporal.
// Requirements: N % 256 == 0, K % 64 == 0, K >= 128, M % 8 == 0, 32-bit byte offsets into x and w.
#pragma once
#include <type_traits>
#include "gemm_common.h"

namespace {

template <int OFF> __device__ __forceinline__ void ph_rd(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

template <int EPI = 1, bool STAGGER = true, bool PRIO = true, int DBG = 0>
__global__ __launch_bounds__(512) void linear_nt_ph_kernel(LinArgs p) {
    constexpr int HALF = 128 * 128;                            // bytes per half-tile: 128 rows x 128 B
    constexpr int BUF = 4 * HALF;                              // W0 | W1 | X0 | X1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    const char* __restrict__ xg = reinterpret_cast<const char*>(p.x);
    const char* __restrict__ wg = reinterpret_cast<const char*>(p.w);
    const int nk = p.K >> 6;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)smem;

    // ---- tile schedule (as linear_nt_p256_kernel): workgroup b, observed on XCD b % 8, takes tiles first, first + G, ... ----
    const int G = gridDim.x;
    const int total = p.tiles_m * p.tiles_n;
    const int first = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    const int ntile = first < total ? (total - first + G - 1) / G : 0;
    const int nsteps = ntile * nk;
    if (nsteps == 0) return;

    // ---- staging: wave w owns pieces 2w, 2w + 1 (8 rows x 128 B each) of every half-tile ----
    const int r8 = lane >> 3, cp = lane & 7;
    unsigned xo[2][2], wo[2][2];                               // byte offsets at k = 0 of this lane's 16 bytes: [half][piece]
    int st_tile = first, st_k = 0;                             // staging cursor: the K step whose half-tiles are being staged
    auto tile_offsets = [&](int t) {
        const int m0 = (t / p.tiles_n) * 256, n0 = (t % p.tiles_n) * 256;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (2 * wave + i) * 8 + r8;        // row inside the half-tile
                const unsigned c = (unsigned)((cp ^ ((r >> 1) & 7)) << 4);
                xo[h][i] = (unsigned)min(m0 + h * 128 + r, p.M - 1) * (unsigned)p.ldx * 2u + c;
                // channel rows sit in LDS in FRAGMENT order: row wq * 64 + b * 16 + i of the half holds channel wq * 64 + chan_of<16>(b, i >> 2, i & 3), so that
                // the accumulator tiles (b, b + 1) of a lane are 8 consecutive output channels (one 16-byte chunk of the epilogue)
                const int q = r & 63;
                const int ch = (r & 64) + chan_of<16>(q >> 4, (q >> 2) & 3, q & 3);
                wo[h][i] = (unsigned)(n0 + h * 128 + ch) * (unsigned)p.ldw * 2u + c;
            }
    };
    // which: 0 W0, 1 W1, 2 X0, 3 X1; buffer = parity of the staged step
    auto stage = [&](int which, int buf) {
        const unsigned kb = (unsigned)st_k * 128u;
        char* dst = smem + buf * BUF + which * HALF + (2 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* src = which < 2 ? wg + (size_t)(wo[which & 1][i] + kb) : xg + (size_t)(xo[which & 1][i] + kb);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + i * 1024), 16, 0, 0);
        }
    };
    int st_step = 0;                                           // index of the step at the cursor
    auto advance = [&]() {                                     // cursor to the next K step; it stays ON the job's last step: the steps behind it re-stage
        if (st_step + 1 >= nsteps) return;                     // that step's half-tiles into slots nobody reads any more -- every step then issues the
        ++st_step;                                             // same number of DMA instructions and the counted waits are constants
        if (++st_k == nk) {
            st_k = 0;
            st_tile += G;
            tile_offsets(st_tile);
        }
    };

    // ---- fragment addresses inside a buffer ----
    // token fragment a (0-7), K block kk: row a * 16 + li of X half wm, chunk kk * 4 + lg; channel fragment b (0-3): row (wn & 1) * 64 + b * 16 + li of W half wn >> 1
    const unsigned xrow = (unsigned)li, wrow = (unsigned)((wn & 1) * 64 + li);
    // rows a * 16 + li: (row >> 1) & 7 = (li >> 1) for every a (16 a is a multiple of 16), so one swizzled chunk offset per kk serves all fragments
    const unsigned sw0 = (unsigned)(((0 + lg) ^ ((li >> 1) & 7)) << 4), sw1 = (unsigned)(((4 + lg) ^ ((li >> 1) & 7)) << 4);
    const unsigned xa0 = lds0 + (2 + wm) * HALF + xrow * 128u + sw0, xa1 = lds0 + (2 + wm) * HALF + xrow * 128u + sw1;
    const unsigned wa0 = lds0 + (wn >> 1) * HALF + wrow * 128u + sw0, wa1 = lds0 + (wn >> 1) * HALF + wrow * 128u + sw1;

    f32x4 acc[8][4];                                          // (never zeroed: a tile's first K step starts every quadrant from a zero C operand)
    bf16x8 tf[4][2], cf[4][2];                                 // token fragments of the current half (4 tiles x 2 K blocks), channel fragments 0-3

    auto read_tokens = [&](unsigned base, int half) {          // 8 reads: tiles 4 half .. 4 half + 3
        const unsigned a0 = xa0 + base + (unsigned)half * (64u * 128u), a1 = xa1 + base + (unsigned)half * (64u * 128u);
        ph_rd<0 * 2048>(tf[0][0], a0); ph_rd<0 * 2048>(tf[0][1], a1);
        ph_rd<1 * 2048>(tf[1][0], a0); ph_rd<1 * 2048>(tf[1][1], a1);
        ph_rd<2 * 2048>(tf[2][0], a0); ph_rd<2 * 2048>(tf[2][1], a1);
        ph_rd<3 * 2048>(tf[3][0], a0); ph_rd<3 * 2048>(tf[3][1], a1);
    };
    auto read_chans = [&](unsigned base, int half) {           // 4 reads: tiles 2 half, 2 half + 1
        const unsigned a0 = wa0 + base + (unsigned)half * (32u * 128u), a1 = wa1 + base + (unsigned)half * (32u * 128u);
        if (half == 0) {
            ph_rd<0 * 2048>(cf[0][0], a0); ph_rd<0 * 2048>(cf[0][1], a1);
            ph_rd<1 * 2048>(cf[1][0], a0); ph_rd<1 * 2048>(cf[1][1], a1);
        } else {
            ph_rd<0 * 2048>(cf[2][0], a0); ph_rd<0 * 2048>(cf[2][1], a1);
            ph_rd<1 * 2048>(cf[3][0], a0); ph_rd<1 * 2048>(cf[3][1], a1);
        }
    };
    auto landed = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(tf[i][0]), "+v"(tf[i][1]), "+v"(cf[i][0]), "+v"(cf[i][1]));
    };
    // 16 MFMAs: token tiles 4 th .. + 3 (held in tf), channel tiles 2 ch, 2 ch + 1.  ZERO (first K step of a tile): the first K block starts from a
    // zero C operand instead of the accumulator, so that nothing has to re-zero 128 registers per tile and a converted accumulator row is DEAD until
    // the next tile writes it (the epilogue's registers come out of that)
    auto quadrant = [&](int th, int ch, auto ZERO_) {
        constexpr bool ZERO = decltype(ZERO_)::value;
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (ZERO && kk == 0)
                        acc[4 * th + a][2 * ch + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[2 * ch + b][kk], tf[a][kk], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else
                        acc[4 * th + a][2 * ch + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[2 * ch + b][kk], tf[a][kk], acc[4 * th + a][2 * ch + b], 0, 0, 0);
                }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- drip epilogue (EPI 2) ----
    // Quadrant j of the wave tile is final after phase j of a tile's LAST K step and is not accumulated into again before phase j of the next
    // tile's FIRST step: a window of four phases in which its registers can be converted and re-zeroed without a second register set.  The
    // tile leaves in four pairs of 16-token accumulator rows -- A (rows 0-31), B (32-63) after P2 / P3 of the last step, C (64-95) after its P4,
    // D (96-127) after P1 of the next step -- each through this wave's 2.3 KB slab (bf16 chunks in, whole 128-byte rows out: the LDS executes a
    // wave's instructions in order, so write / read / write / read on the one slab need no wait) into 16 registers, stored one phase later.
    // The stores are spread over four phases, and every counted vmcnt lets the youngest of them stay in flight: a burst of all 16 store
    // instructions at the tile boundary stalled the next step's DMA wait until the whole chip's tiles had drained to HBM (vmcnt retires in order).
    bf16x8 rb[4];
    int em0 = 0, en0 = 0;
    char* const wslab = smem + 2 * BUF + wave * (16 * 144);
    // bias: this wave's 64 channels travel into a 256-byte LDS slot by DMA (4 bytes per lane) at the head of the tile's first K step -- a plain global load
    // in the epilogue would have to wait, in vmcnt's issue order, for every DMA and store in front of it; two slots by tile parity (the previous tile's
    // pair D is converted after the next tile's slot has been requested)
    float* const bslab = reinterpret_cast<float*>(smem + 2 * BUF + 8 * (16 * 144)) + wave * 128;
    int bpar = 0;                                              // slot of the tile being multiplied
    auto stage_bias = [&](int t, int par) {
        if (p.bias) {
            const int n0w = (t % p.tiles_n) * 256 + wn * 64;
            __builtin_amdgcn_global_load_lds((gptr_t*)(p.bias + n0w + lane), (lptr_t*)(bslab + par * 64), 4, 0, 0);
        }
    };
    const int rr = lane >> 3, rc = lane & 7;
    auto convert_pair = [&](auto A0, int par) {                // accumulator rows 16 A0 .. 16 A0 + 31 (+ bias) -> rb
        constexpr int a0 = decltype(A0)::value;
        f32x4 bb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) bb[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int b = 0; b < 4; ++b) bb[b] = *reinterpret_cast<const f32x4*>(bslab + par * 64 + chan_of<16>(b, lg, 0));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)(acc[a0 + t][2 * c + (e >> 2)][e & 3] + bb[2 * c + (e >> 2)][e & 3]);
                *reinterpret_cast<bf16x8*>(wslab + li * 144 + (c * 32 + lg * 8) * 2) = v;
            }
            // The registers of rb still hold the rows store_pair() has just sent: keep them reserved up to here.  The compiler re-uses a store's data
            // registers two wait states behind the instruction (the documented hazard), but with the vector-memory queue full of DMA the data was
            // read LATER than that: whole registers of stored rows came out as the next store's address offset (race screen of the probe: ~1000 of
            // 1e8 elements per launch, none with the burst epilogue).
            // (DBG 1: reserve them explicitly up to here -- clean as well, but the 16 extra live registers spill; the product form relies on the pause
            //  behind the stores in store_pair)
            if constexpr (DBG & 1) asm volatile("" ::"v"(rb[2 * t]), "v"(rb[2 * t + 1]));
#pragma unroll
            for (int h = 0; h < 2; ++h) rb[2 * t + h] = *reinterpret_cast<const bf16x8*>(wslab + (h * 8 + rr) * 144 + rc * 16);
        }
    };
    // buffer stores: rows past M of a ragged last panel carry an offset beyond the descriptor's size and are dropped by the hardware -- the
    // instruction is issued (and counted by vmcnt) whatever the row, which the counted waits rely on
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (unsigned)p.M * (unsigned)p.ldy * 2u, 0x00020000);
    const auto nullrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, 0u, 0x00020000);
    const unsigned lane_off = ((unsigned)rr * (unsigned)p.ldy + (unsigned)(rc * 8)) * 2u;     // this lane's 16 bytes inside an 8-row block
    auto store_pair = [&](int a0) {
        // address = descriptor base + SCALAR offset of the 8-row block + lane_off: one address register for the whole kernel (per-row-block
        // vector offsets were hoisted into 15 registers by the compiler and spilled the kernel).  The range check looks at the vector offset alone.
        // Rows past M of a ragged last panel: M % 8 == 0, so an 8-row block is valid or invalid as a whole -- a SCALAR choice between the output's
        // descriptor and an empty one (every access out of range, dropped by the hardware).  The instruction is issued either way, which the counted
        // waits rely on, and the only address register is lane_off: per-store vector offsets cost registers the kernel does not have (spills), and a
        // store's registers must not be re-used right behind it (see convert_pair).
        // non-temporal: the output streams to memory once; as ordinary write-allocating stores each tile round fills the XCD's L2 with dirty lines and
        // evicts the operand panels the next K steps re-read (125440 x 1536 x 384: 190 -> 135 us, 8192^3: 842 -> 769; `nt sc1` the same, `sc0 sc1` no gain)
        // two 8-row blocks at a time, then a pause before anything may overwrite the stores' data registers (see convert_pair).
        // (Round 5 carried a GELU form here -- EPI 3, GELU of the bf16-ROUNDED pre-activation on the read-back side; it moved the bf16 gradient
        // statistics past their bar (0.089 -> 0.116 against 0.10) and was never selected; removed in round 6.  GELU epilogues evaluate on the fp32
        // accumulator: gemm_ph3.h, gemm.hip.)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m_blk = em0 + (a0 + j) * 16 + i * 8;                                   // wave-uniform
                const unsigned soff = ((unsigned)m_blk * (unsigned)p.ldy + (unsigned)en0) * 2u;
                const bool ok = m_blk < p.M;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rb[2 * j + i]), ok ? yrs : nullrs, lane_off, ok ? soff : 0u, 2);
            }
            if (j == 1) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_sleep(2);                   // ~128 cycles
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- prologue: step 0 complete, W0 of step 1 on the way ----
    tile_offsets(first);
    stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
    advance();
    stage(0, 1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    bar();
    if (STAGGER && wm == 1) bar();                             // waves 4-7 run one barrier behind

    // One K step.  FIRST: the previous tile's pairs C, D leave during this step (first step of a tile that has a predecessor); LAST: this step
    // completes tile ct.  Compile-time, so that the steps in between carry no epilogue code and no branch: at 16 MFMAs per phase every scalar
    // branch behind the MFMAs is on the workgroup's critical path (measured: the same epilogue behind run-time conditions cost the main loop 10 %).
    int ct = first;
    auto step = [&](auto ZERO_, auto FIRST_, auto LAST_, int s) {
        constexpr bool FIRST = decltype(FIRST_)::value && EPI >= 2, LAST = decltype(LAST_)::value && EPI >= 2;
        const unsigned base = (unsigned)(s & 1) * (unsigned)BUF;
        const int nb = (s + 1) & 1;                            // buffer of the step being staged (cursor = s + 1 during P1-P3)
        // ---- P1 ----
        read_chans(base, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_tokens(base, 0);
        if constexpr (decltype(ZERO_)::value && EPI >= 2) stage_bias(ct, bpar);
        stage(2, nb);
        bar();
        landed();
        quadrant(0, 0, ZERO_);
        if constexpr (FIRST) {
            store_pair(4);                                     // C (converted after the previous step's P4)
            convert_pair(std::integral_constant<int, 6>{}, bpar ^ 1);
        }
        bar();
        // ---- P2 ----
        read_chans(base, 1);
        stage(3, nb);
        bar();
        landed();
        quadrant(0, 1, ZERO_);
        if constexpr (LAST) {
            em0 = (ct / p.tiles_n) * 256 + wm * 128;
            en0 = (ct % p.tiles_n) * 256 + wn * 64;
            convert_pair(std::integral_constant<int, 0>{}, bpar);
        }
        bar();
        // ---- P3 ----
        read_tokens(base, 1);
        stage(1, nb);
        bar();
        landed();
        quadrant(1, 1, ZERO_);
        if constexpr (LAST) {
            store_pair(0);
            convert_pair(std::integral_constant<int, 2>{}, bpar);
        } else if constexpr (FIRST) {
            store_pair(6);                                     // D: the previous tile is out
        }
        bar();
        // ---- P4 ----
        advance();                                             // cursor -> step s + 2
        stage(0, s & 1);                                       // W0 of step s + 2 into the buffer whose channel half-tiles were last read in P2
        // everything of step s + 1 has landed (this wave's part); younger and allowed in flight: P4's own two DMA instructions and, in the
        // steps around a tile boundary, the four stores of this step's P3
        if constexpr (FIRST || LAST) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        bar();
        quadrant(1, 0, ZERO_);
        if constexpr (LAST) {
            store_pair(2);
            convert_pair(std::integral_constant<int, 4>{}, bpar);
        }
        bar();
    };
    auto tile_done_burst = [&]() {                             // EPI 0 / 1: everything at the tile boundary
        if constexpr (EPI == 1) {
            const int m0 = (ct / p.tiles_n) * 256 + wm * 128, n0 = (ct % p.tiles_n) * 256 + wn * 64;
            bf16* __restrict__ yg = reinterpret_cast<bf16*>(p.y);
            char* ws = smem + 2 * BUF + wave * (16 * 144);
            f32x4 bb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bb[b] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + chan_of<16>(b, lg, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 8; ++a) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)(acc[a][2 * c + (e >> 2)][e & 3] + bb[2 * c + (e >> 2)][e & 3]);
                    *reinterpret_cast<bf16x8*>(ws + li * 144 + (c * 32 + lg * 8) * 2) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m = m0 + a * 16 + h * 8 + rr;
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(ws + (h * 8 + rr) * 144 + rc * 16);
                    if (m < p.M) *reinterpret_cast<bf16x8*>(yg + (size_t)m * p.ldy + n0 + rc * 8) = v;
                }
            }
        }
        if constexpr (EPI == 0) {
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(acc[a][b]));
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    int s = 0;
    for (int t = 0; t < ntile; ++t) {
        if (t == 0) step(T_{}, F_{}, F_{}, s);
        else step(T_{}, T_{}, F_{}, s);
        ++s;
        for (int k = 1; k < nk - 1; ++k, ++s) step(F_{}, F_{}, F_{}, s);
        step(F_{}, F_{}, T_{}, s);
        ++s;
        tile_done_burst();
        ct += G;
        bpar ^= 1;
    }
    if constexpr (EPI >= 2) {                                  // the job's last tile: pairs C, D
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        store_pair(4);
        convert_pair(std::integral_constant<int, 6>{}, bpar ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        store_pair(6);
    }
    if (STAGGER && wm == 0) bar();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI = 1, bool STAGGER = true, bool PRIO = true, int DBG = 0>
int launch_ph(const LinArgs& a, hipStream_t st, int grid = 256) {
    constexpr int lds = 2 * 4 * 128 * 128 + 8 * 16 * 144 + 8 * 512;   // ring + wave-private epilogue slabs + bias slots
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&linear_nt_ph_kernel<EPI, STAGGER, PRIO, DBG>), lds)) return rc_;
    if (a.N % 256 || a.K % 64 || a.K < 128) return FMMT_EINVAL;
    LinArgs p = a;
    p.tiles_m = (a.M + 255) / 256;
    p.tiles_n = a.N / 256;
    hipLaunchKernelGGL((linear_nt_ph_kernel<EPI, STAGGER, PRIO, DBG>), dim3(grid), dim3(512), lds, st, p);
    FMMT_CHECK_LAUNCH();
    return 0;
}

}  // namespace
