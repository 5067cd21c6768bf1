// Persistent NT GEMM, round 5 (third form): 192-token x 256-channel tiles, K step 64 in THREE phases of 16 MFMAs, two wave groups half a phase apart.
//
//   y[M,N] = epi( x[M,K] . w[N,K]^T + bias )       bf16 operands, fp32 accumulate; Swin stages 1-3 (Swin_Transformer.py:19-28,105-107,142,304)
//
// Same phase structure as gemm_ph.h (which see for why), re-cut for the shapes this model has:
//   * 192-token tiles: 31360 tokens x 768 channels are 369 tiles of 256 x 256 (1.44 rounds of 256 workgroups: the second round 44 % full) but 492 tiles of
//     192 x 256 (1.92 rounds); 7840 tokens x 1536 channels 186 -> 246 tiles.  Every token count of Swin stages 1-3 (125440, 31360, 7840) fills its last
//     round to >= 95 % with 192-row tiles.
//   * waves 2 (tokens) x 4 (channels), wave tile 96 tokens x 64 channels = 6 x 4 accumulator tiles (96 registers instead of 128): the epilogue's operands
//     (rows read back from the slab, residual / GELU' operand rows, bias) have registers of their own -- the 256 x 256 form sat at the 256-register limit.
//   * a K step = three phases, phase j = token tiles 2j, 2j + 1 x all four channel tiles x both K blocks (16 MFMAs):
//         P1 reads 8 channel + 4 token fragments | P2 reads the other 8 token fragments | P3 reads nothing
//     so a token pair is FINAL after its own phase of the tile's last step and untouched until the same phase of the next tile: the epilogue of pair j
//     (convert, through the wave's slab, whole 128-byte rows out) sits behind phase j of the last step, its stores one phase later.
//   * LDS: two buffers of W0 | W1 (128 channel rows each) | X (192 token rows), 56 KB; staging by write-after-read distance (a slot is re-staged two
//     phases after its last fragment read):  P1(s): X(s+1) (3 DMA instructions per wave) | P3(s): W0, W1(s+2) (4) and the step's ONE counted vmcnt.
// EPI 0: no output (main loop alone); 2: bias, bf16, non-temporal whole rows; 3: GELU of the fp32 value (+ the bf16 pre-activation to y_pre);
// 4: y = value * gelu'(aux) (fc2's input gradient); 5: y = res + rowscale[row / rows_per_scale] * (value + bias) (proj / fc2 forward, res or rowscale may be
// absent); 6: y = value * aux * rowscale (FMMT_EPI_MUL_AUX: fc2's input gradient against the STORED derivative -- EPI 5's read-back path with a product instead
// of a sum; round 6: the GELU' launches of stages 2 / 3 had been the step's slowest GEMMs, 0.19 of their roofline, on the round-2 kernels).
// 4, 5 and 6 read their M x N operand as WHOLE ROWS on the epilogue's read-back side (16 bytes per lane, 8 lanes per 128-byte line), requested three
// phases before they are used; they need K >= 192.
// Requirements: N % 256 == 0, K % 64 == 0, K >= 128, M % 8 == 0, 32-bit byte offsets into x, w, y.
#pragma once
#include <type_traits>
#include "gemm_common.h"

namespace {

template <int OFF> __device__ __forceinline__ void q3_rd(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// GELU / GELU' of NE values: the forms of fmmt_common.h (one v_exp_f32 + plain FMAs whose coefficients are literals -- nothing to hoist into registers;
// round 5 carried its own copy of the packed odd polynomials here because those kept 2 x 10 coefficient registers alive for the whole kernel)
template <int NE> __device__ __forceinline__ void q3_gelu(float (&v)[NE]) {
#pragma unroll
    for (int j = 0; j < NE; ++j) v[j] = gelu_exp_f(v[j]);
}
template <int NE> __device__ __forceinline__ void q3_gelu_grad_mul(float (&v)[NE], const float (&pre)[NE]) {
#pragma unroll
    for (int j = 0; j < NE; ++j) v[j] *= gelu_grad_exp_f(pre[j]);
}

template <int EPI = 2, bool STAGGER = true, int WT = 2>
__global__ __launch_bounds__(512) void linear_nt_ph3_kernel(LinArgs p) {
    // the argument's fields as locals, read once: referenced through the lambdas below the struct itself was kept in scratch memory by the larger instantiations
    // (GELU / GELU' epilogues) and every use re-loaded from there
    const int p_M = p.M, p_K = p.K, p_ldx = p.ldx, p_ldw = p.ldw, p_ldy = p.ldy, p_ldaux = p.ldaux, p_ldres = p.ldres, p_tiles_m = p.tiles_m, p_tiles_n = p.tiles_n,
              p_rows_per_scale = p.rows_per_scale;
    const void* const p_x = p.x; const void* const p_w = p.w; const void* const p_aux = p.aux; const void* const p_res = p.res;
    void* const p_y = p.y; void* const p_y_pre = p.y_pre;
    const float* const p_bias = p.bias; const float* const p_rowscale = p.rowscale;
    // WT = waves along the tokens: 2 -> tile 192 tokens x 256 channels (waves 2 x 4), 4 -> 384 tokens x 128 channels (waves 4 x 2: the channel counts that are
    // multiples of 128 but not of 256 -- Swin stage 2's 384 and 1152).  The wave tile is 96 x 64 either way.
    static_assert(WT == 2 || WT == 4, "wave layout");
    constexpr int BM = WT * 96, BN = (8 / WT) * 64;
    constexpr int WP = BN / 64, XP = BM / 64;                  // DMA instructions (8 rows x 128 B) per wave and K step: channel rows, token rows
    constexpr int XOFF = BN * 128, XB = BM * 128;              // token rows behind the channel rows
    constexpr int BUF = XOFF + XB;                             // 56 KB / 64 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WT == 2 ? wave >> 2 : wave >> 1, wn = WT == 2 ? wave & 3 : wave & 1;
    const int grp = wave >> 2;                                 // waves w and w + 4 share a SIMD: the two groups are the token halves
    const int li = lane & 15, lg = lane >> 4;
    const char* __restrict__ xg = reinterpret_cast<const char*>(p_x);
    const char* __restrict__ wg = reinterpret_cast<const char*>(p_w);
    const int nk = p_K >> 6;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)smem;

    // ---- tile schedule: workgroup b, observed on XCD b % 8, takes tiles first, first + G, ...; channel tiles fastest ----
    const int G = gridDim.x;
    const int total = p_tiles_m * p_tiles_n;
    const int first = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    const int ntile = first < total ? (total - first + G - 1) / G : 0;
    const int nsteps = ntile * nk;
    if (nsteps == 0) return;

    // ---- staging: wave w owns pieces WP w .. of the channel rows and XP w .. of the token rows (8 rows x 128 B a piece) ----
    const int r8 = lane >> 3, cp = lane & 7;
    unsigned xo[XP], wo[WP];                                   // byte offsets at k = 0 of this lane's 16 bytes
    int st_tile = first, st_k = 0, st_step = 0;                // staging cursor
    auto tile_offsets = [&](int t) __attribute__((always_inline)) {
        const int m0 = (t / p_tiles_n) * BM, n0 = (t % p_tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int r = (XP * wave + i) * 8 + r8;           // token row inside the tile
            xo[i] = (unsigned)min(m0 + r, p_M - 1) * (unsigned)p_ldx * 2u + (unsigned)((cp ^ ((r >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int r = (WP * wave + i) * 8 + r8;           // channel row inside the tile
            // fragment order: row wq * 64 + b * 16 + i holds channel wq * 64 + chan_of<16>(b, i >> 2, i & 3): the accumulator tiles (b, b + 1) of a lane are 8
            // consecutive output channels
            const int q = r & 63;
            const int ch = (r & ~63) + chan_of<16>(q >> 4, (q >> 2) & 3, q & 3);
            wo[i] = (unsigned)(n0 + ch) * (unsigned)p_ldw * 2u + (unsigned)((cp ^ ((r >> 1) & 7)) << 4);
        }
    };
    auto stage_x = [&](int buf) __attribute__((always_inline)) {
        const unsigned kb = (unsigned)st_k * 128u;
        char* dst = smem + buf * BUF + XOFF + (XP * wave) * 1024;
#pragma unroll
        for (int i = 0; i < XP; ++i) __builtin_amdgcn_global_load_lds((gptr_t*)(xg + (size_t)(xo[i] + kb)), (lptr_t*)(dst + i * 1024), 16, 0, 0);
    };
    auto stage_w = [&](int buf) __attribute__((always_inline)) {
        const unsigned kb = (unsigned)st_k * 128u;
        char* dst = smem + buf * BUF + (WP * wave) * 1024;
#pragma unroll
        for (int i = 0; i < WP; ++i) __builtin_amdgcn_global_load_lds((gptr_t*)(wg + (size_t)(wo[i] + kb)), (lptr_t*)(dst + i * 1024), 16, 0, 0);
    };
    auto advance = [&]() __attribute__((always_inline)) {                                     // cursor to the next K step; it stays ON the job's last step (harmless re-staging, constant DMA counts)
        if (st_step + 1 >= nsteps) return;
        ++st_step;
        if (++st_k == nk) {
            st_k = 0;
            st_tile += G;
            tile_offsets(st_tile);
        }
    };

    // ---- fragment addresses inside a buffer: token fragment a (0-5): row wm * 96 + a * 16 + li; channel fragment b (0-3): row wn * 64 + b * 16 + li ----
    const unsigned sw0 = (unsigned)(((0 + lg) ^ ((li >> 1) & 7)) << 4), sw1 = (unsigned)(((4 + lg) ^ ((li >> 1) & 7)) << 4);
    const unsigned xrow = (unsigned)(wm * 96 + li) * 128u, wrow = (unsigned)(wn * 64 + li) * 128u;
    const unsigned xa0 = lds0 + XOFF + xrow + sw0, xa1 = lds0 + XOFF + xrow + sw1;
    const unsigned wa0 = lds0 + wrow + sw0, wa1 = lds0 + wrow + sw1;

    f32x4 acc[6][4];                                           // (never zeroed: a tile's first K step starts from a zero C operand)
    bf16x8 cf[4][2], tfa[2][2], tfb[4][2];

    auto landed = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto bar = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // 16 MFMAs: token tiles A0, A0 + 1 (fragments t[0], t[1]) x the four channel tiles x both K blocks
    auto mma16 = [&](auto A0_, bf16x8 (&t0)[2], bf16x8 (&t1)[2], auto ZERO_) {
        constexpr int a0 = decltype(A0_)::value;
        constexpr bool ZERO = decltype(ZERO_)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(t0[i]), "+v"(t1[i]));
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(cf[b][0]), "+v"(cf[b][1]));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const bf16x8& tf = t == 0 ? t0[kk] : t1[kk];
                    if (ZERO && kk == 0) acc[a0 + t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[b][kk], tf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else acc[a0 + t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[b][kk], tf, acc[a0 + t][b], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- drip epilogue ----
    bf16x8 rb[4];
    bf16x8 rg[4];                                              // EPI 3: the GELU rows beside the pre-activation rows in rb; EPI 6: the rounding residuals of rb
    float rsv[4];                                              // EPI 5 / 6: DropPath scale of the four 8-row blocks' rows
    float* const rslab = reinterpret_cast<float*>(smem + 2 * BUF + 8 * (16 * 144) + 8 * 512) + wave * 256;   // two 128-float slots by tile parity: rowscale of this wave's 96 rows
    int em0 = 0, en0 = 0;
    char* const wslab = smem + 2 * BUF + wave * (16 * 144);
    float* const bslab = reinterpret_cast<float*>(smem + 2 * BUF + 8 * (16 * 144)) + wave * 128;      // two 64-float slots by tile parity
    int bpar = 0;
    const int rr = lane >> 3, rc = lane & 7;
    auto stage_bias = [&](int t, int par) __attribute__((always_inline)) {                    // issued whether or not there is a bias (from the weights then: any readable address): constant DMA counts
        const int n0w = (t % p_tiles_n) * BN + wn * 64;
        const float* src = p_bias ? p_bias + n0w + lane : reinterpret_cast<const float*>(wg) + lane;
        __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(bslab + par * 64), 4, 0, 0);
    };
    auto convert_pair = [&](auto A0_, int par) __attribute__((always_inline)) {               // accumulator rows 16 A0 .. 16 A0 + 31 (+ bias) -> rb (whole rows: lane = row rr (+ 8), channels 8 rc .. + 7)
        constexpr int a0 = decltype(A0_)::value;
        f32x4 bb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) bb[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p_bias) {
#pragma unroll
            for (int b = 0; b < 4; ++b) bb[b] = *reinterpret_cast<const f32x4*>(bslab + par * 64 + chan_of<16>(b, lg, 0));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float tv[2][8];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    tv[c][e] = acc[a0 + t][2 * c + (e >> 2)][e & 3] + bb[2 * c + (e >> 2)][e & 3];
                    v[e] = (bf16)tv[c][e];
                }
                *reinterpret_cast<bf16x8*>(wslab + li * 144 + (c * 32 + lg * 8) * 2) = v;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) rb[2 * t + h] = *reinterpret_cast<const bf16x8*>(wslab + (h * 8 + rr) * 144 + rc * 16);
            if constexpr (EPI == 6) {
                // the product with the stored derivative is formed on the read-back side, where the value has been rounded to bf16 once already: a second pass
                // carries the rounding residual (value - bf16(value), itself to bf16: 2^-17 relative together), so that y = (hi + lo) * aux * scale rounds ONCE.
                // (Without it the whole-Swin bf16 gradient statistics went from 0.089 to 0.10-0.12 relative L2 against the fp32 oracle: every Mlp backward of
                // stages 1-3 rounded d(hidden) twice.)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)(tv[c][e] - (float)(bf16)tv[c][e]);
                    *reinterpret_cast<bf16x8*>(wslab + li * 144 + (c * 32 + lg * 8) * 2) = v;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) rg[2 * t + h] = *reinterpret_cast<const bf16x8*>(wslab + (h * 8 + rr) * 144 + rc * 16);
            }
            if constexpr (EPI == 3) {                          // GELU of the fp32 value (as the persistent kernel's epilogue does), a second pass through the slab
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bf16x8 v;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        float g4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) g4[e] = tv[c][4 * hf + e];
                        __builtin_amdgcn_sched_barrier(0);
                        q3_gelu<4>(g4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * hf + e] = (bf16)g4[e];
                    }
                    *reinterpret_cast<bf16x8*>(wslab + li * 144 + (c * 32 + lg * 8) * 2) = v;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) rg[2 * t + h] = *reinterpret_cast<const bf16x8*>(wslab + (h * 8 + rr) * 144 + rc * 16);
            }
        }
        if constexpr (EPI == 5 || EPI == 6) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rsv[i] = p_rowscale ? rslab[par * 128 + (a0 + (i >> 1)) * 16 + (i & 1) * 8 + rr] : 1.0f;
        }
    };
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(p_y, 0, (unsigned)p_M * (unsigned)p_ldy * 2u, 0x00020000);
    const auto nullrs = __builtin_amdgcn_make_buffer_rsrc(p_y, 0, 0u, 0x00020000);
    const auto prs = __builtin_amdgcn_make_buffer_rsrc(p_y_pre ? p_y_pre : p_y, 0, p_y_pre ? (unsigned)p_M * (unsigned)p_ldy * 2u : 0u, 0x00020000);
    const unsigned lane_off = ((unsigned)rr * (unsigned)p_ldy + (unsigned)(rc * 8)) * 2u;
    // operand of the read-back side (EPI 4: aux, EPI 5: res; an absent one: empty descriptor, loads return zero -- the instruction count stays fixed)
    constexpr bool OPS = EPI == 4 || EPI == 5 || EPI == 6;
    const void* const opp = (EPI == 4 || EPI == 6) ? p_aux : p_res;
    const int ldop = (EPI == 4 || EPI == 6) ? p_ldaux : p_ldres;
    const auto ors = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(opp ? opp : p_y), 0, opp ? (unsigned)p_M * (unsigned)ldop * 2u : 0u, 0x00020000);
    const unsigned lane_off_op = ((unsigned)rr * (unsigned)ldop + (unsigned)(rc * 8)) * 2u;
    u32x4 op0[4], op1[4];                                      // operand rows: op0 = pair A, then pair C; op1 = pair B
    auto load_ops = [&](u32x4 (&dst)[4], int a0, int m0w, int n0w) {   // rows 16 a0 .. + 31 of the wave tile at (m0w, n0w)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m_blk = m0w + (a0 + (i >> 1)) * 16 + (i & 1) * 8;
            const unsigned soff = ((unsigned)m_blk * (unsigned)ldop + (unsigned)n0w) * 2u;
            const bool ok = m_blk < p_M;
            dst[i] = __builtin_amdgcn_raw_buffer_load_b128(ok ? ors : nullrs, lane_off_op, ok ? soff : 0u, 0);
        }
    };
    auto stage_rowscale = [&](int t, int par) __attribute__((always_inline)) {                // EPI 5: rowscale[(m0 + row) / rows_per_scale] for this wave's 96 rows -> LDS (two 4-byte-per-lane DMA instructions)
        const int m0w = (t / p_tiles_n) * BM + wm * 96;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = min(m0w + i * 64 + lane, p_M - 1);
            const float* src = p_rowscale ? p_rowscale + m / p_rows_per_scale : reinterpret_cast<const float*>(wg);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(rslab + par * 128 + i * 64), 4, 0, 0);
        }
    };
    // see gemm_ph.h: scalar block offsets, empty descriptor for rows past M, pause behind the stores.  opv: the pair's operand rows (EPI 4 / 5)
    auto store_pair = [&](int a0, u32x4 (&opv)[4]) {
        if constexpr (OPS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 o = __builtin_bit_cast(bf16x8, opv[i]);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {               // four elements at a time, one row block after the other: the polynomial's temporaries are what spills
                    float v[4], a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = (float)rb[i][4 * hf + e]; a[e] = (float)o[4 * hf + e]; }
                    if constexpr (EPI == 4) {
                        q3_gelu_grad_mul<4>(v, a);
                    } else if constexpr (EPI == 6) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = a[e] * (rsv[i] * (v[e] + (float)rg[i][4 * hf + e]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = a[e] + rsv[i] * v[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[i][4 * hf + e] = (bf16)v[e];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m_blk = em0 + (a0 + (i >> 1)) * 16 + (i & 1) * 8;
            const unsigned soff = ((unsigned)m_blk * (unsigned)p_ldy + (unsigned)en0) * 2u;
            const bool ok = m_blk < p_M;
            if constexpr (EPI == 3) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rb[i]), ok ? prs : nullrs, lane_off, ok ? soff : 0u, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rg[i]), ok ? yrs : nullrs, lane_off, ok ? soff : 0u, 2);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rb[i]), ok ? yrs : nullrs, lane_off, ok ? soff : 0u, 2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: step 0 complete, the channel halves of step 1 on the way ----
    tile_offsets(first);
    stage_w(0);
    stage_x(0);
    advance();
    stage_w(1);
    if constexpr (WP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    bar();
    if (STAGGER && grp == 1) bar();                            // waves 4-7 run one barrier behind

    int ct = first;
    auto step = [&](auto ZERO_, auto FIRST_, auto PRELAST_, auto LAST_, int s) __attribute__((always_inline)) {
        constexpr bool ZERO = decltype(ZERO_)::value;
        constexpr bool FIRST = decltype(FIRST_)::value && EPI >= 2, LAST = decltype(LAST_)::value && EPI >= 2;
        constexpr bool PRELAST = decltype(PRELAST_)::value && OPS;
        constexpr int NST = EPI == 3 ? 8 : 4;                  // store instructions per pair
        const unsigned base = (unsigned)(s & 1) * (unsigned)BUF;
        if constexpr (LAST) {
            em0 = (ct / p_tiles_n) * BM + wm * 96;
            en0 = (ct % p_tiles_n) * BN + wn * 64;
        }
        // ---- P1: token tiles 0, 1 ----
        {
            const unsigned w0 = wa0 + base, w1 = wa1 + base, x0 = xa0 + base, x1 = xa1 + base;
            q3_rd<0 * 2048>(cf[0][0], w0); q3_rd<0 * 2048>(cf[0][1], w1);
            q3_rd<1 * 2048>(cf[1][0], w0); q3_rd<1 * 2048>(cf[1][1], w1);
            q3_rd<0 * 2048>(tfa[0][0], x0); q3_rd<0 * 2048>(tfa[0][1], x1);
            q3_rd<1 * 2048>(tfa[1][0], x0); q3_rd<1 * 2048>(tfa[1][1], x1);
            q3_rd<2 * 2048>(cf[2][0], w0); q3_rd<2 * 2048>(cf[2][1], w1);
            q3_rd<3 * 2048>(cf[3][0], w0); q3_rd<3 * 2048>(cf[3][1], w1);
        }
        if constexpr (ZERO && EPI >= 2) stage_bias(ct, bpar);
        if constexpr (ZERO && (EPI == 5 || EPI == 6)) stage_rowscale(ct, bpar);
        stage_x((s + 1) & 1);                                  // token rows of step s + 1 (cursor) into the other buffer: last read in P2 of step s - 1
        bar();
        landed();
        mma16(std::integral_constant<int, 0>{}, tfa[0], tfa[1], ZERO_);
        if constexpr (FIRST) store_pair(4, op0);               // the previous tile's last pair (converted behind its P3)
        if constexpr (LAST) convert_pair(std::integral_constant<int, 0>{}, bpar);
        bar();
        // ---- P2: token tiles 2, 3 (and the fragments of 4, 5) ----
        {
            const unsigned x0 = xa0 + base, x1 = xa1 + base;
            q3_rd<2 * 2048>(tfb[0][0], x0); q3_rd<2 * 2048>(tfb[0][1], x1);
            q3_rd<3 * 2048>(tfb[1][0], x0); q3_rd<3 * 2048>(tfb[1][1], x1);
            q3_rd<4 * 2048>(tfb[2][0], x0); q3_rd<4 * 2048>(tfb[2][1], x1);
            q3_rd<5 * 2048>(tfb[3][0], x0); q3_rd<5 * 2048>(tfb[3][1], x1);
        }
        bar();
        landed();
        mma16(std::integral_constant<int, 2>{}, tfb[0], tfb[1], ZERO_);
        if constexpr (LAST) {
            store_pair(0, op0);
            if constexpr (OPS) load_ops(op0, 4, em0, en0);     // pair C's operand rows: used behind P1 of the next step
            convert_pair(std::integral_constant<int, 2>{}, bpar);
        }
        bar();
        // ---- P3: token tiles 4, 5 ----
        advance();                                             // cursor -> step s + 2
        stage_w(s & 1);                                        // its channel halves into THIS buffer: last read in P1
        // everything of step s + 1 has landed (this wave's part); younger and allowed in flight: the four DMA instructions just issued and, around a tile
        // boundary, the four stores behind this step's P1 (FIRST) / P2 (LAST)
        // (LAST with operands: the pair-C operand loads behind P2 are younger as well)
        {
            constexpr int NW = WP + ((FIRST || LAST) ? NST : 0) + ((LAST && OPS) ? 4 : 0);
            static_assert(NW == 2 || NW == 4 || NW == 6 || NW == 8 || NW == 10 || NW == 12, "counted wait");
            if constexpr (NW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (NW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (NW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NW == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
        bar();
        mma16(std::integral_constant<int, 4>{}, tfb[2], tfb[3], ZERO_);
        if constexpr (LAST) {
            store_pair(2, op1);
            convert_pair(std::integral_constant<int, 4>{}, bpar);
        }
        if constexpr (PRELAST) {                               // the operand rows of pairs A and B of the tile the NEXT step completes: three phases ahead of their use
            em0 = (ct / p_tiles_n) * BM + wm * 96;
            en0 = (ct % p_tiles_n) * BN + wn * 64;
            load_ops(op0, 0, em0, en0);
            load_ops(op1, 2, em0, en0);
        }
        bar();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    int s = 0;
    for (int t = 0; t < ntile; ++t) {
        if (t == 0) step(T_{}, F_{}, F_{}, F_{}, s);
        else step(T_{}, T_{}, F_{}, F_{}, s);
        ++s;
        if constexpr (OPS) {                                   // K >= 192: the step in front of the last one is neither the first nor the last
            for (int k = 1; k < nk - 2; ++k, ++s) step(F_{}, F_{}, F_{}, F_{}, s);
            step(F_{}, F_{}, T_{}, F_{}, s);
            ++s;
        } else {
            for (int k = 1; k < nk - 1; ++k, ++s) step(F_{}, F_{}, F_{}, F_{}, s);
        }
        step(F_{}, F_{}, F_{}, T_{}, s);
        ++s;
        if constexpr (EPI == 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(acc[a][b]));
        }
        ct += G;
        bpar ^= 1;
    }
    if constexpr (EPI >= 2) {                                  // the job's last pair
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        store_pair(4, op0);
    }
    if (STAGGER && grp == 0) bar();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the re-staged last step must not land in a successor workgroup's LDS
}

template <int EPI = 2, bool STAGGER = true, int WT = 2>
int launch_ph3(const LinArgs& a, hipStream_t st, int grid = 256) {
    constexpr int BM = WT * 96, BN = (8 / WT) * 64;
    constexpr int lds = 2 * (BN + BM) * 128 + 8 * 16 * 144 + 8 * 512 + 8 * 1024;     // ring + wave-private epilogue slabs + bias slots + rowscale slots
    static_assert(lds <= 160 * 1024, "LDS");
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&linear_nt_ph3_kernel<EPI, STAGGER, WT>), lds)) return rc_;
    if (a.N % BN || a.K % 64 || a.K < 128 || ((EPI == 4 || EPI == 5 || EPI == 6) && a.K < 192)) return FMMT_EINVAL;
    LinArgs p = a;
    p.tiles_m = (a.M + BM - 1) / BM;
    p.tiles_n = a.N / BN;
    hipLaunchKernelGGL((linear_nt_ph3_kernel<EPI, STAGGER, WT>), dim3(grid), dim3(512), lds, st, p);
    FMMT_CHECK_LAUNCH();
    return 0;
}

}  // namespace
