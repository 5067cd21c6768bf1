// Linear layers of the hot path as hand-written MFMA kernels for gfx950.
//
//   linear_nt_kernel : y = epi(x . w^T + b) [+ residual]        (forward and input-gradient GEMMs)
//   linear_tn_kernel : dw = dy^T . x, db = colsum(dy)           (weight-gradient GEMM, split over M)
//
// Both keep the *weight* on the MFMA "A" side and the *activation* on the "B" side, so the
// accumulator fragment of a lane is a run of consecutive output channels of ONE token: the
// epilogue (bias, GELU, residual, DropPath scale) works on contiguous vectors and stores rows.
#include "gemm_common.h"
#include "gemm_ph.h"
#include "gemm_ph3.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static constexpr int KM = 32;  // K per MFMA
    static constexpr int KP = 8;   // K elements per lane
    using frag = bf16x8;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag load(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
};
template <> struct Mma<float> {
    static constexpr int KM = 4;
    static constexpr int KP = 1;
    using frag = float;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag load(const float* p) { return *p; }
};

// XCD-aware, bijective block remap: consecutive logical tiles share an XCD's L2
// (cdna_hip_programming.md T1; the dispatcher places block b on XCD b % 8).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
// (Identity and banded tile orders were measured on the write-heavy stage-0/1 shapes: within +-3 % of this one.)

// ---------------------------------------------------------------------------------------------
// NT kernel: 256 threads = 4 waves (2 along M x 2 along N); block tile BM x BN, K step BK.
//  * K pipeline: while tile k is multiplied out of LDS, tile k+1 is in flight into registers; two LDS
//    buffers, one barrier per K step.  BK = 64 makes every global load instruction fetch whole 128-byte
//    rows.  K == 96 (Swin stage 0) is a single step (NBUF = 1).  (A second register stage was measured
//    and lost: occupancy beats prefetch depth on this chip for these shapes -- tools/probes/gemm_bench.py.)
//  * one output tile per workgroup.  (A persistent variant that issues the next tile's first K step under
//    the epilogue was measured and lost as well: +60..100 VGPRs -> one wave per SIMD.)
// ---------------------------------------------------------------------------------------------
// Register budget: the K = 96 single-step kernels are latency-bound (load -> LDS -> MFMA -> store with nothing to
// overlap inside a workgroup), so they need the three workgroups per CU that their 43-53 KB of LDS allows, i.e.
// <= 168 VGPRs+AGPRs; the double-buffered 128-row kernels are LDS-limited to two per CU (<= 256 registers).
template <int BM, int NBUF> struct NtWaves { static constexpr int value = (BM == 128) ? (NBUF == 1 ? 3 : 2) : 4; };
template <typename T, int BM, int BN, int BK, int NBUF, bool GLDS, bool SEG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NtWaves<BM, NBUF>::value)))
void linear_nt_kernel(LinArgs p) {
    constexpr int VEC = Vec<T>::N;
    static_assert(!SEG || GLDS, "segmented weights: direct-to-LDS kernels only");
    // GLDS: tiles are filled by direct global->LDS DMA (global_load_lds_dwordx4): the LDS image of a wave
    // instruction is lane-linear, so rows are unpadded (128 B) and bank conflicts are removed by an XOR
    // swizzle applied to the *source* chunk index and again on the fragment read (key = (row >> 1) & 7).
    constexpr int PITCH = GLDS ? BK : BK + VEC;
    constexpr int KM = Mma<T>::KM, KP = Mma<T>::KP;
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 16, NT = WN / 16;
    constexpr int KV = BK / VEC;                        // 16-byte vectors per tile row
    constexpr int W_VECS = BN * KV, X_VECS = BM * KV;
    constexpr int WV = (W_VECS + 255) / 256, XV = (X_VECS + 255) / 256;
    constexpr int CW = 4 * NT;                          // consecutive channels owned by a lane in the epilogue

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ws = reinterpret_cast<T*>(smem);                 // [NBUF][BN][PITCH]
    T* Xs = Ws + NBUF * BN * PITCH;                     // [NBUF][BM][PITCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;

    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const float* bias_seg = nullptr;
    if constexpr (SEG) {
        // (SEG is its own instantiation: the kernel argument stays untouched -- a modified copy of LinArgs ends up in scratch memory and made
        //  every launch of this kernel 2-3x slower, measured)
        if (p.wseg_mode == 1) {
            // output channels in segments of p.wseg, one weight / bias each: re-base this workgroup's pointers so that the global channel
            // index n addresses row n - s * wseg of its segment (a tile never straddles two: wseg % BN == 0)
            const int s_ = ((xcd_remap(blockIdx.x, gridDim.x) % p.tiles_n) * BN) / p.wseg;
            wg = reinterpret_cast<const T*>(s_ == 0 ? p.w : (s_ == 1 ? p.w1 : p.w2)) - (ptrdiff_t)s_ * p.wseg * p.ldw;
            const float* bs_ = s_ == 0 ? p.bias : (s_ == 1 ? p.bias1 : p.bias2);
            bias_seg = bs_ ? bs_ - (ptrdiff_t)s_ * p.wseg : nullptr;
        }
    }

    const int kbeg = p.ksplit ? blockIdx.y * p.ksplit : 0;
    const int kend = p.ksplit ? min(p.K, kbeg + p.ksplit) : p.K;
    const int nk = (kend - kbeg + BK - 1) / BK;

    struct Stage { Vec<T> w[WV], x[XV]; };
    Stage R0;

    auto gload = [&](Stage& R, int k0, int m0, int n0) {
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * 256;
            if (v < W_VECS) {
                const int row = v / KV, kc = (v % KV) * VEC;
                const int n = min(n0 + row, p.N - 1);
                R.w[i] = (k0 + kc < kend) ? ldvec<T>(wg + (size_t)n * p.ldw + k0 + kc) : zerovec<T>();
            }
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * 256;
            if (v < X_VECS) {
                const int row = v / KV, kc = (v % KV) * VEC;
                const int m = min(m0 + row, p.M - 1);
                R.x[i] = (k0 + kc < kend) ? ldvec<T>(xg + (size_t)m * p.ldx + k0 + kc) : zerovec<T>();
            }
        }
    };
    auto lstore = [&](const Stage& R, int buf) {
        T* wsb = Ws + buf * BN * PITCH;
        T* xsb = Xs + buf * BM * PITCH;
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * 256;
            if (v < W_VECS) stvec<T>(wsb + (v / KV) * PITCH + (v % KV) * VEC, R.w[i]);
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * 256;
            if (v < X_VECS) stvec<T>(xsb + (v / KV) * PITCH + (v % KV) * VEC, R.x[i]);
        }
    };

    f32x4 acc[MT][NT];

    // weight-row permutation inside a wave tile: see chan_of()
    int wrow[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) wrow[b] = wn * WN + chan_of<CW>(b, li >> 2, li & 3);
    const int xrow_base = wm * WM + li;
    const int koff = lg * KP;

    auto compute = [&](int buf) {
        const T* wsb = Ws + buf * BN * PITCH;
        const T* xsb = Xs + buf * BM * PITCH;
#pragma unroll
        for (int kk = 0; kk < BK / KM; ++kk) {
            typename Mma<T>::frag wf[NT], xf[MT];
            if constexpr (GLDS) {
#pragma unroll
                for (int b = 0; b < NT; ++b) wf[b] = Mma<T>::load(wsb + wrow[b] * PITCH + (((kk * 4 + lg) ^ ((wrow[b] >> 1) & 7)) << 3));
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    const int r = xrow_base + a * 16;
                    xf[a] = Mma<T>::load(xsb + r * PITCH + (((kk * 4 + lg) ^ ((r >> 1) & 7)) << 3));
                }
            } else {
#pragma unroll
                for (int b = 0; b < NT; ++b) wf[b] = Mma<T>::load(wsb + wrow[b] * PITCH + kk * KM + koff);
#pragma unroll
                for (int a = 0; a < MT; ++a) xf[a] = Mma<T>::load(xsb + (xrow_base + a * 16) * PITCH + kk * KM + koff);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = Mma<T>::mma(wf[b], xf[a], acc[a][b]);
        }
    };

    auto epilogue = [&](int m0, int n0) {
        if constexpr (SEG) {
            if (bias_seg) {
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_seg + n0 + wn * WN + chan_of<CW>(b, lg, 0));
#pragma unroll
                    for (int a = 0; a < MT; ++a) acc[a][b] += bb;
                }
            }
            nt_epilogue<T, MT, NT, true>(p, acc, m0 + wm * WM, n0 + wn * WN, li, lg);
        } else {
            nt_epilogue<T, MT, NT>(p, acc, m0 + wm * WM, n0 + wn * WN, li, lg);
        }
    };

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (logical / p.tiles_n) * BM, n0 = (logical % p.tiles_n) * BN;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (GLDS) {
        static_assert(BK == 64 && (NBUF == 2 || NBUF == 4) && sizeof(T) == 2, "direct-to-LDS path: bf16, BK = 64, two or four buffers");
        typedef __attribute__((address_space(1))) const void gptr_t;
        typedef __attribute__((address_space(3))) void lptr_t;
        auto issue = [&](int buf, int k0) {
            const int r8 = lane >> 3, c = lane & 7;
            // K-segmented weight (wseg_mode 2): this K step's columns live in weight k0 / wseg, at column k0 % wseg
            const T* wk = wg;
            int kw = k0;
            if constexpr (SEG) {
                if (p.wseg_mode == 2) {
                    const int s_ = k0 / p.wseg;
                    wk = reinterpret_cast<const T*>(s_ == 0 ? p.w : (s_ == 1 ? p.w1 : p.w2));
                    kw = k0 - s_ * p.wseg;
                }
            }
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) {
                const int grp = i * 4 + wave, row = grp * 8 + r8;
                const T* src = wk + (size_t)min(n0 + row, p.N - 1) * p.ldw + kw + ((c ^ ((row >> 1) & 7)) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(Ws + (buf * BN + grp * 8) * PITCH), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) {
                const int grp = i * 4 + wave, row = grp * 8 + r8;
                const T* src = xg + (size_t)min(m0 + row, p.M - 1) * p.ldx + k0 + ((c ^ ((row >> 1) & 7)) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(Xs + (buf * BM + grp * 8) * PITCH), 16, 0, 0);
            }
        };
        if constexpr (NBUF == 4) {
            // Few-token problems (one workgroup per CU at most, K loop of 12-48 steps that is pure latency): a ring of
            // four buffers keeps three K steps in flight; each wave waits only for its own DI DMA instructions of the
            // step about to be used (counted vmcnt), a raw s_barrier publishes it and frees the buffer read last step.
            constexpr int DI = BN / 32 + BM / 32;
            static_assert(2 * DI <= 63, "vmcnt immediate");
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (s < nk) issue(s, kbeg + s * BK);
            for (int kt = 0; kt < nk; ++kt) {
                const int ahead = min(nk - 1 - kt, 2);     // steps issued after step kt that may still be in flight
                if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DI) : "memory");
                else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DI) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kt + 3 < nk) issue((kt + 3) & 3, kbeg + (kt + 3) * BK);
                compute(kt & 3);
            }
            epilogue(m0, n0);
            return;
        }
        issue(0, kbeg);
        __syncthreads();                                   // the compiler drains vmcnt(0) in front of the barrier
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) issue(cur ^ 1, kbeg + (kt + 1) * BK);
            compute(cur);
            __syncthreads();
        }
        epilogue(m0, n0);
        return;
    }
    gload(R0, kbeg, m0, n0);
    lstore(R0, 0);
    __syncthreads();
    if constexpr (NBUF == 1) {
        compute(0);
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) gload(R0, kbeg + (kt + 1) * BK, m0, n0);
            compute(cur);
            if (kt + 1 < nk) lstore(R0, cur ^ 1);
            __syncthreads();
        }
    }
    epilogue(m0, n0);
}

// ---------------------------------------------------------------------------------------------
// Deep-pipelined NT kernel for the compute-heavy shapes (bf16, K % 64 == 0, N % 128 == 0):
// 512 threads = 8 waves (4 along M x 2 along N), block tile 256 x 128, K step 64, THREE LDS buffers
// filled by direct global->LDS DMA.  Two tiles are in flight while one is multiplied: each wave waits
// only for its own six DMA instructions of the tile about to be used (counted s_waitcnt vmcnt(6)), then
// a raw s_barrier publishes the tile; the buffer freed by that barrier is refilled immediately.
// LDS 3 x 48 KB = 144 KB: one workgroup (8 waves, 2 per SIMD) per CU.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void linear_nt_deep_kernel(LinArgs p) {
    using T = bf16;
    constexpr int BM = 256, BN = 128, BK = 64, PITCH = 64, MT = 4, NT = 4, CW = 16;
    constexpr int STAGE = (BM + BN) * PITCH;               // elements per buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* S = reinterpret_cast<T*>(smem);                     // [3][BN rows of W | BM rows of X][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (logical / p.tiles_n) * BM, n0 = (logical % p.tiles_n) * BN;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const int nk = p.K / BK;

    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int r8 = lane >> 3, c = lane & 7;
    auto issue = [&](int buf, int k0) {
        T* base = S + buf * STAGE;
#pragma unroll
        for (int i = 0; i < BN / 64; ++i) {                // 2 DMA instructions per wave for W
            const int grp = i * 8 + wave, row = grp * 8 + r8;
            const T* src = wg + (size_t)min(n0 + row, p.N - 1) * p.ldw + k0 + ((c ^ ((row >> 1) & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(base + grp * 8 * PITCH), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BM / 64; ++i) {                // 4 DMA instructions per wave for X
            const int grp = i * 8 + wave, row = grp * 8 + r8;
            const T* src = xg + (size_t)min(m0 + row, p.M - 1) * p.ldx + k0 + ((c ^ ((row >> 1) & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(base + (BN + grp * 8) * PITCH), 16, 0, 0);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int wrow[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) wrow[b] = wn * 64 + chan_of<CW>(b, li >> 2, li & 3);
    const int xrow_base = wm * 64 + li;

    auto compute = [&](int buf) {
        const T* wsb = S + buf * STAGE;
        const T* xsb = wsb + BN * PITCH;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 wf[NT], xf[MT];
#pragma unroll
            for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8*>(wsb + wrow[b] * PITCH + (((kk * 4 + lg) ^ ((wrow[b] >> 1) & 7)) << 3));
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int r = xrow_base + a * 16;
                xf[a] = *reinterpret_cast<const bf16x8*>(xsb + r * PITCH + (((kk * 4 + lg) ^ ((r >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
        }
    };

    issue(0, 0);
    if (nk > 1) issue(1, BK);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // tile kt landed (this wave's part); tile kt+1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                         // every wave's part of tile kt is in LDS
        if (kt + 2 < nk) issue(cur == 0 ? 2 : cur - 1, (kt + 2) * BK);        // refill the buffer last read in step kt-1
        compute(cur);
        cur = cur == 2 ? 0 : cur + 1;
    }
    // epilogue through LDS, as in linear_nt_deep32_kernel: plain / bias through the block-wide slab, everything else through
    // wave-private slabs (FMMT_NT_SLAB=0 / FMMT_NT_WSLAB=0: straight from the accumulator layout)
    const bool plain = p.epi == 0 && !p.y_pre && !p.res && !p.aux && !p.rowscale;
    if (p.part || (p.reserved & (plain ? 16 : 32))) {
        nt_epilogue<T, MT, NT>(p, acc, m0 + wm * 64, n0 + wn * 64, li, lg);
    } else if (plain) {
        nt_epilogue_slab<T, MT, NT, 2, BN, 512>(p, acc, smem, wm, wn, li, lg, tid, m0, n0);
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // every wave is done with the K loop's stages
        nt_epilogue_wslab<T>(p, acc, smem + wave * (16 * 272), lane, li, lg, m0 + wm * 64, n0 + wn * 64);
    }
}

// ---------------------------------------------------------------------------------------------
// Same 256 x 128 tile with K step 32: three 24 KB buffers = 72 KB, so TWO workgroups (16 waves) share a CU and
// one workgroup's epilogue (GELU, stores) overlaps the other's MFMA loop.  A DMA instruction now covers 16 rows
// x 64 B; one chunk swizzle serves token rows and channel-permuted weight rows (see swz below).  BN = 96 is the
// same kernel for N = 96 / 192 / 288 / 576 (48-channel wave slabs, 67.5 KB of LDS).
// ---------------------------------------------------------------------------------------------
// NK: compile-time number of K steps (3 / 6 for the K = 96 / 192 stage-0 / stage-1 problems, which are HBM streams and
// get their own fully unrolled instantiations), 0 = run-time.
template <int NK, int BN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4)))
void linear_nt_deep32_kernel(LinArgs p) {
    using T = bf16;
    constexpr int BM = 256, BK = 32, PITCH = 32, MT = 4, NT = BN / 32, CW = 4 * NT, WN = BN / 2;
    static_assert(BN == 128 || BN == 96, "block tile is 256 x 128 or 256 x 96");
    constexpr int STAGE = (BM + BN) * PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* S = reinterpret_cast<T*>(smem);                     // [3][BN rows of W | BM rows of X][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (logical / p.tiles_n) * BM, n0 = (logical % p.tiles_n) * BN;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const int nk = NK ? NK : p.K / BK;

    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int r16 = lane >> 2, c = lane & 3;
    // bank swizzle of the four 16-byte chunks of a 64-byte row: ((row >> 3) ^ (row >> 2)) & 3 gives 16 distinct slots of
    // the 256-byte bank window both for 16 consecutive rows (token rows, the 4-channel tail of a 48-channel wave slab)
    // and for the channel-permuted weight rows {8g + q + c0}
    auto swz = [](int row) { return ((row >> 3) ^ (row >> 2)) & 3; };
    // DMA instructions this wave issues per stage (the counted vmcnt below): 2 for X, +1 if it owns a group of W rows
    const bool w_owner = BN == 128 || wave < BN / 16;
    auto wait_stage = [&](bool next_in_flight) {
        if (!next_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (w_owner) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    };
    auto issue = [&](int buf, int k0) {
        T* base = S + buf * STAGE;
        if (BN == 128 || wave < BN / 16) {                 // W: BN / 16 groups of 16 rows, one DMA instruction each
            const int row = wave * 16 + r16;
            const T* src = wg + (size_t)min(n0 + row, p.N - 1) * p.ldw + k0 + ((c ^ swz(row)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(base + wave * 16 * PITCH), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                      // 2 DMA instructions per wave for X (16 x 16 rows)
            const int grp = i * 8 + wave, row = grp * 16 + r16;
            const T* src = xg + (size_t)min(m0 + row, p.M - 1) * p.ldx + k0 + ((c ^ swz(row)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(base + (BN + grp * 16) * PITCH), 16, 0, 0);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int woff[NT], xoff[MT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int r = wn * WN + chan_of<CW>(b, li >> 2, li & 3);
        woff[b] = r * PITCH + ((lg ^ swz(r)) << 3);
    }
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int r = wm * 64 + a * 16 + li;
        xoff[a] = (BN + r) * PITCH + ((lg ^ swz(r)) << 3);
    }

    auto compute = [&](int buf) {
        const T* sb = S + buf * STAGE;
        bf16x8 wf[NT], xf[MT];
#pragma unroll
        for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8*>(sb + woff[b]);
#pragma unroll
        for (int a = 0; a < MT; ++a) xf[a] = *reinterpret_cast<const bf16x8*>(sb + xoff[a]);
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    };

    issue(0, 0);
    if (nk > 1) issue(1, BK);
    if constexpr (NK > 0) {
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            wait_stage(kt + 1 < NK);
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < NK) issue((kt + 2) % 3, (kt + 2) * BK);
            compute(kt % 3);
        }
    } else {
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            wait_stage(kt + 1 < nk);
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) issue(cur == 0 ? 2 : cur - 1, (kt + 2) * BK);
            compute(cur);
            cur = cur == 2 ? 0 : cur + 1;
        }
    }
    // FMMT_NT_SLAB=0 (p.reserved bit 4): the older epilogue, straight from the accumulator layout (A/B switch)
    const bool plain = p.epi == 0 && !p.y_pre && !p.res && !p.aux && !p.rowscale;
    if (p.part || (p.reserved & 16) || (!plain && (BN != 128 || (p.reserved & 32)))) {
        nt_epilogue<T, MT, NT>(p, acc, m0 + wm * 64, n0 + wn * WN, li, lg);
    } else if (plain) {
        nt_epilogue_slab<T, MT, NT, 2, BN, 512>(p, acc, smem, wm, wn, li, lg, tid, m0, n0);
    } else {
        if constexpr (BN == 128) {                         // operand / GELU epilogues: wave-private slab, whole 128-byte lines
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // every wave is done with the K loop's stages
            // (The GELU' table of fmmt_common.h staged in the freed stages was tried here: 532 -> 501 us at stage 1 -- and one launch in
            //  ~40 wrote a garbage 16 x 64 fragment, always a wave of the third token quarter, wherever the table sat; staging it without
            //  using it was clean.  Not understood, not kept: this epilogue stays on the polynomial.)
            nt_epilogue_wslab<T>(p, acc, smem + wave * (16 * 272), lane, li, lg, m0 + wm * 64, n0 + wn * WN);
        }
    }
}

// (A "wide-wave" variant -- the same 256 x 128 block computed by four waves with 128 x 64 wave tiles, 0.375 KB of
//  LDS reads per MFMA instead of 0.5 KB -- was measured on the 125440-token shapes: 0.98-1.02x, i.e. LDS bandwidth is
//  not what bounds these kernels.  What does: bytes in flight.  A K step's MFMA work (~0.1 us per wave) is far
//  shorter than the ~2 us L2/HBM latency of the DMA that feeds it, and 160 KB of LDS holds ~100 KB of in-flight
//  tiles per CU, so a CU can pull ~50 GB/s = 12.8 TB/s chip-wide; at 85 FLOP per loaded byte (256 x 128 tile)
//  that caps the kernel near 1.1 PFLOP/s, and the measured 0.8 PFLOP/s on K = 1536 is 72 % of that cap.  The next
//  step up is a 256 x 256 tile (128 FLOP/B) with persistent scheduling for the tile-count quantisation.)

// ---------------------------------------------------------------------------------------------
// Persistent NT kernel for the compute-heavy shapes (bf16; M % 8 == 0; N % BN == 0; K % BK == 0, K >= 192):
// one workgroup of 8 waves per CU walks 256-token x BN-channel output tiles (BN = 256 / 192 / 128: 128 / 110 / 85 FLOP
// per loaded byte against 85 for the 256 x 128 kernels above).  Waves: 2 along tokens x 4 along channels, wave tile
// 128 x BN/4, i.e. 0.375 KB of LDS fragment reads per MFMA (0.5 KB with 64 x 64 wave tiles).
//  * ONE flat pipeline over (tile, K step) pairs: a ring of NBUF LDS stages filled by direct global->LDS DMA, NBUF - 1
//    stages in flight, counted s_waitcnt vmcnt + one raw s_barrier per K step.  The ring does not drain at a tile
//    boundary: while a tile's accumulators are converted and stored, the first stages of the workgroup's next tile are
//    already landing (a one-tile-per-workgroup kernel pays the DMA latency and an idle epilogue once per tile -- at
//    K = 384 that is a third of the tile's life -- and two co-resident workgroups do not fit: 128 accumulator
//    registers per lane).
//  * the bias slab travels through the ring as well (one extra 1 KB DMA per stage by wave 7, two slots by tile parity):
//    an ordinary global load in the epilogue would make the in-order vmcnt wait for every DMA issued before it.
//    Residual / GELU' operand / DropPath scale are M x N data and do load in the epilogue (one partial drain per tile).
//  * tile order: channel tiles fastest within a token panel; workgroup b (observed on XCD b % 8) takes, in round i,
//    tile i * 256 + (b % 8) * 32 + b / 8 -- each XCD works on a contiguous run of 32 tiles, so a token panel is fetched
//    into one XCD's L2 once per round.
// ---------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// All fragment reads of one 32-deep K block (NT weight + 8 token fragments, ds_read_b128) and the wait for them as ONE inline-asm
// statement.  Left to itself hipcc loads a token fragment, waits lgkmcnt(0), issues its four MFMAs, loads the next ... -- the
// LDS latency is exposed once per four MFMAs; batched, once per 32-48.  (Same device as tn_read12 below.)
#define FMMT_RD(o, ad) "ds_read_b128 %" #o ", %" #ad "\n\t"
template <int NF>
__device__ __forceinline__ void nt_read_frags(bf16x8 (&o)[NF], const unsigned (&ad)[NF]) {
    static_assert(NF >= 10 && NF <= 12, "fragments per K block");
    if constexpr (NF == 12) {
        asm volatile(FMMT_RD(0, 12) FMMT_RD(1, 13) FMMT_RD(2, 14) FMMT_RD(3, 15) FMMT_RD(4, 16) FMMT_RD(5, 17) FMMT_RD(6, 18) FMMT_RD(7, 19)
                     FMMT_RD(8, 20) FMMT_RD(9, 21) FMMT_RD(10, 22) FMMT_RD(11, 23) "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                       "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
                     : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]),
                       "v"(ad[8]), "v"(ad[9]), "v"(ad[10]), "v"(ad[11])
                     : "memory");
    } else if constexpr (NF == 11) {
        asm volatile(FMMT_RD(0, 11) FMMT_RD(1, 12) FMMT_RD(2, 13) FMMT_RD(3, 14) FMMT_RD(4, 15) FMMT_RD(5, 16) FMMT_RD(6, 17) FMMT_RD(7, 18)
                     FMMT_RD(8, 19) FMMT_RD(9, 20) FMMT_RD(10, 21) "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                       "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10])
                     : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]),
                       "v"(ad[8]), "v"(ad[9]), "v"(ad[10])
                     : "memory");
    } else {
        asm volatile(FMMT_RD(0, 10) FMMT_RD(1, 11) FMMT_RD(2, 12) FMMT_RD(3, 13) FMMT_RD(4, 14) FMMT_RD(5, 15) FMMT_RD(6, 16) FMMT_RD(7, 17)
                     FMMT_RD(8, 18) FMMT_RD(9, 19) "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                       "=&v"(o[8]), "=&v"(o[9])
                     : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]),
                       "v"(ad[8]), "v"(ad[9])
                     : "memory");
    }
}

template <int BN, int BK, int NBUF, bool BATCH = true, bool HASOP = false, bool PIPE = false>
__global__ __launch_bounds__(512) void linear_nt_p256_kernel(LinArgs p) {

    using T = bf16;
    constexpr int BM = 256, PITCH = BK, MT = 8, NT = BN / 64, WN = BN / 4, CW = 4 * NT;
    constexpr int RPI = BK == 64 ? 8 : 16;                 // tile rows per DMA instruction (64 lanes x 16 B = 1 KB)
    constexpr int STAGE = (BN + BM) * PITCH;               // elements per ring stage: BN weight rows, then 256 token rows
    constexpr int NI = (BN + BM) / RPI;                    // DMA instructions per stage, dealt round-robin to the 8 waves
    constexpr int CNT_LO = NI / 8;
    static_assert(BK == 32 || BK == 64, "K step");
    static_assert(BN % 64 == 0 && (NBUF - 2) * (CNT_LO + 2) <= 63, "tile / vmcnt immediate");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* S = reinterpret_cast<T*>(smem);                                              // [NBUF][BN + 256][BK]
    float* bias_s = reinterpret_cast<float*>(smem + (size_t)NBUF * STAGE * sizeof(T));   // [2][256]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const int nk = p.K / BK;

    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    auto key = [](int row) { return BK == 64 ? ((row >> 1) & 7) : (((row >> 3) ^ (row >> 2)) & 3); };   // 16-byte chunk swizzle
    const int rl = BK == 64 ? (lane >> 3) : (lane >> 2), cl = BK == 64 ? (lane & 7) : (lane & 3);
    // Weight rows sit in LDS in FRAGMENT order (K step 64): row wn * WN + b * 16 + i of the stage holds the output channel that
    // lane i of n-tile b accumulates (chan_of), so that a weight fragment read touches 16 consecutive 128-byte rows like a token
    // fragment does -- with the rows in channel order the lanes of one ds_read_b128 phase met rows {0-3, 24-27} and {8-11, 16-19},
    // whose swizzle keys coincide pairwise: 2-way bank conflicts on every 8-channel-chunk fragment (SQ_LDS_BANK_CONFLICT 19 % of
    // the LDS cycles).  FMMT_NT_P256_WROWS=0 (reserved bit 64): channel order (A/B switch).
    const bool wperm = BK == 64 && !(p.reserved & 64);
    auto wchan = [&](int r) {
        const int w = r / WN, q = r - w * WN;
        return wperm ? w * WN + chan_of<CW>(q >> 4, (q >> 2) & 3, q & 3) : r;
    };

    // DMA instructions this wave issues per stage (wave 7 carries the bias slab on top)
    const int my_cnt = (NI - wave + 7) / 8 + (wave == 7 ? 1 : 0);
    const bool cnt_hi = my_cnt > CNT_LO;                                            // CNT_LO or CNT_LO + 1 (+ 2 never: see below)
    // Addresses of the DMA pieces.  A piece's global address is (uniform base) + (32-bit byte offset of this lane's 16 bytes at
    // k = 0, recomputed when the issue cursor enters a tile) + (k offset): one VALU add per piece and K step, the SGPR-base form
    // of the instruction.  (Formed from scratch per piece -- row, channel permutation, 64-bit row x pitch product -- the address
    // arithmetic was ~12 VALU instructions per piece, and the issue phase 1370 cycles of a 4100-cycle K step, s_memtime stamps.)
    constexpr int NP = (NI + 7) / 8;                       // pieces per wave and stage (the last may be absent: j >= NI)
    unsigned poff[NP], boff = 0;
    auto tile_offsets = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row0 = (i * 8 + wave) * RPI;                                    // wave-uniform
            if (row0 < BN) {
                const int r = row0 + rl;
                poff[i] = ((unsigned)(n0 + wchan(r)) * (unsigned)p.ldw + (unsigned)((cl ^ key(r)) << 3)) * 2u;
            } else {
                const int rx = row0 - BN;
                const int rb = min(m0 + rx, p.M - RPI);                             // ragged last panel: re-read valid rows (never stored)
                poff[i] = ((unsigned)(rb + rl) * (unsigned)p.ldx + (unsigned)((cl ^ key(rx + rl)) << 3)) * 2u;
            }
        }
        boff = p.bias ? (unsigned)(n0 + min(lane * 4, BN - 4)) * 4u : (unsigned)lane * 16u;
    };
    auto issue_piece = [&](int i, int slot, unsigned kbyte) {
        const int j = i * 8 + wave;
        if (j < NI) {
            const int row0 = j * RPI;
            const char* b = row0 < BN ? reinterpret_cast<const char*>(wg) : reinterpret_cast<const char*>(xg);
            __builtin_amdgcn_global_load_lds((gptr_t*)(b + (size_t)(poff[i] + kbyte)), (lptr_t*)(S + slot * STAGE + row0 * PITCH), 16, 0, 0);
        }
    };
    auto issue_bias = [&](int par) {
        if (wave == 7) {
            const char* b = p.bias ? reinterpret_cast<const char*>(p.bias) : reinterpret_cast<const char*>(wg);
            __builtin_amdgcn_global_load_lds((gptr_t*)(b + (size_t)boff), (lptr_t*)(bias_s + par * 256), 16, 0, 0);
        }
    };
    // The oldest stage in flight has landed when at most `ahead` younger stages -- and the `stores` store instructions this wave
    // issued AFTER the newest of them -- are still outstanding (vmcnt counts stores too and retires in issue order; an
    // under-count only makes the wait stricter).
    auto wait_landed = [&](int ahead, int stores) {
        const int n = (ahead <= 0 ? 0 : (ahead == 1 ? 1 : 2)) * (CNT_LO + (cnt_hi ? 1 : 0)) + stores;
        switch (n) {
#define FMMT_W(N) case N: wait_vm<N>(); break;
            FMMT_W(1) FMMT_W(2) FMMT_W(3) FMMT_W(4) FMMT_W(5) FMMT_W(6) FMMT_W(7) FMMT_W(8) FMMT_W(9) FMMT_W(10) FMMT_W(11) FMMT_W(12)
            FMMT_W(13) FMMT_W(14) FMMT_W(15) FMMT_W(16) FMMT_W(17) FMMT_W(18) FMMT_W(19) FMMT_W(20) FMMT_W(21) FMMT_W(22) FMMT_W(23) FMMT_W(24)
#undef FMMT_W
            default: wait_vm<0>(); break;                  // 0, or more than the table holds: wait for everything
        }
    };
    static_assert(NBUF >= 2 && NBUF <= 4, "ring depth");
    static_assert(NI % 8 == 0 || NI % 8 <= 7, "wave 7 never owns a high count plus the bias slab");

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int woff[NT], xoff[MT];                                // element offsets of this lane's fragments inside a stage (K chunk lg)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int r = wn * WN + (wperm ? b * 16 + li : chan_of<CW>(b, li >> 2, li & 3));
        woff[b] = r * PITCH + ((lg ^ key(r)) << 3);
    }
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int r = wm * 128 + a * 16 + li;
        xoff[a] = (BN + r) * PITCH + ((lg ^ key(r)) << 3);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)S;
    auto compute = [&](int slot) {
        if constexpr (BATCH) {
            const unsigned sbase = lds0 + (unsigned)slot * (unsigned)(STAGE * 2);
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                // second half of a 64-wide K step: chunk index + 4 -> the swizzled chunk offset flips bit 2 (32 elements = 64 bytes)
                unsigned ad[NT + MT];
#pragma unroll
                for (int b = 0; b < NT; ++b) ad[b] = (((unsigned)woff[b] * 2u) ^ (kk ? 64u : 0u)) + sbase;
#pragma unroll
                for (int a = 0; a < MT; ++a) ad[NT + a] = (((unsigned)xoff[a] * 2u) ^ (kk ? 64u : 0u)) + sbase;
                bf16x8 fr[NT + MT];
                nt_read_frags<NT + MT>(fr, ad);
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[b], fr[NT + a], acc[a][b], 0, 0, 0);
            }
        } else {
        const T* sb = S + slot * STAGE;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            // second half of a 64-wide K step: chunk index + 4 -> the swizzled chunk offset flips bit 2 (32 elements)
            bf16x8 wf[NT], xf[MT];
#pragma unroll
            for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8*>(sb + (kk ? (woff[b] ^ 32) : woff[b]));
#pragma unroll
            for (int a = 0; a < MT; ++a) xf[a] = *reinterpret_cast<const bf16x8*>(sb + (kk ? (xoff[a] ^ 32) : xoff[a]));
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
        }
        }
    };

    const int G = gridDim.x;                               // multiple of 8
    const int total = p.tiles_m * p.tiles_n;
    // (Tile order: a contiguous run of tiles per workgroup -- a token panel fetched from HBM by its first tile, re-read from L2 /
    //  MALL by the same workgroup's next tiles_n - 1 -- was measured against this round-robin order: 0-10 % slower, every shape.)
    const int first = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    const int tstep = G;
    const int ntile = first < total ? (total - first + G - 1) / G : 0;
    const int nsteps = ntile * nk;
    // (A start stagger -- workgroups delayed by a hashed fraction of a tile period, on the theory that 256 persistent workgroups in
    //  lockstep all write their tiles in the same 3.6 us window and meet the HBM write ceiling there -- was measured, round 4, same call:
    //  only the workgroups with one tile fewer delayed: +-1 % on every shape; everyone by up to half / a whole period: -2...-40 %, in
    //  proportion to the delay.  The epilogue's cost is per CU, not a chip-wide write queue.)
    // issue-side cursor
    int it = first, ik = 0, ipar = 0, islot = 0;
    tile_offsets((first / p.tiles_n) * BM, (first % p.tiles_n) * BN);
    auto issue_advance = [&]() {
        islot = islot + 1 == NBUF ? 0 : islot + 1;
        if (++ik == nk) {
            ik = 0;
            it += tstep;
            ipar ^= 1;
            tile_offsets((it / p.tiles_n) * BM, (it % p.tiles_n) * BN);
        }
    };
    auto issue_next = [&]() {
        const unsigned kbyte = (unsigned)ik * (BK * 2);
#pragma unroll
        for (int i = 0; i < NP; ++i) issue_piece(i, islot, kbyte);
        issue_bias(ipar);
        issue_advance();
    };
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < nsteps) issue_next();
    // compute-side cursor
    int ct = first, ck = 0, cpar = 0, cslot = 0, st_prev = 0;
    EpiPre<MT, NT> pre;
    // Epilogue through LDS.  The accumulator layout gives a store instruction 16 token rows x 64 contiguous bytes; the CU's store
    // path takes such stores at ~7 B per cycle and, being the vector-memory path the DMA loads use as well, it cannot be hidden:
    // 125440 x 1152 x 384 took 190 us with the epilogue and 125 us without, spread over the next tile's K steps or not (both
    // measured).  So a finished tile goes through a small LDS scratch, PR token rows per token-wave group at a time: the waves
    // write their 16-byte accumulator chunks (bias added, bf16) into row-major [PR][BN] slabs, and read whole rows back -- every
    // store instruction then writes 1 KB of complete 128-byte lines.  Plain / bias epilogues; GELU + pre-activation, operand and
    // split-K epilogues keep nt_epilogue.
    constexpr int PR = BN == 192 ? 32 : 16;                // rows per pass and wave group (LDS budget: 25.6 / 16.9 / 8.7 KB)
    constexpr int SP = BN * 2 + 16;                        // slab row pitch in bytes
    constexpr int CPR = BN / 8;                            // 16-byte chunks per row
    constexpr int NSL = PR * CPR / 256;                    // store instructions per lane and pass (256 lanes per wave group)
    static_assert((PR * CPR) % 256 == 0, "whole store instructions");
    char* scratch = smem + (size_t)NBUF * STAGE * sizeof(T) + 2 * 256 * sizeof(float) + (size_t)wm * PR * SP;
    // (K > 1536: the tile's 48+ K steps dwarf the epilogue and the passes' sixteen barriers cost more than the stores save:
    //  31360 x 768 x 3072 measured 153 us direct, 165 us through LDS)
    const bool lds_gelu = p.epi == FMMT_EPI_GELU && !(p.reserved & 8);      // GELU (+ pre-activation): two tensors through the slab
    const bool lds_epi = !HASOP && (p.epi == 0 || lds_gelu) && (!p.y_pre || lds_gelu) && !p.part && !(p.reserved & 4) && p.K <= 1536;
    T* __restrict__ yg = reinterpret_cast<T*>(p.y);
    // PIPE: fragment reads software-pipelined against the MFMAs.  The eight waves of the workgroup pass the K step's barrier
    // together, so with "read a K block's fragments, wait, issue its MFMAs" they all queue on the LDS at once (88 KB per K block
    // at 128 B per cycle: ~700 cycles) and then all on the matrix cores (2 waves x 24 MFMAs x 16 cycles per SIMD): the two
    // phases alternate instead of overlapping, and a K step takes ~4100 cycles for 1536 cycles of MFMA work.  Here a K step is
    // cut into micro-batches of XB token fragments (+ the weight fragments at the head of a K block); the reads of micro-batch
    // u + 1 are issued in front of the MFMAs of micro-batch u and waited for behind them, two register buffers alternating.
    // The step's barrier moves in front of the LAST micro-batch's MFMAs: by then every read of stage s has returned (the slot
    // is free for the DMA of stage s + NBUF) and the first reads of stage s + 1 can go out under those MFMAs.
    constexpr int XB = NT >= 4 ? 2 : 4;                    // token fragments per micro-batch (register budget: 128 accumulators at NT = 4)
    constexpr int UB = MT / XB, U = (BK / 32) * UB;        // micro-batches per 32-deep K block / per K step
    bf16x8 pw[2][NT], px[2][XB];
    auto request = [&](int slot, int u) {                  // issue (do not wait for) the fragment reads of micro-batch u
        const unsigned sbase = lds0 + (unsigned)slot * (unsigned)(STAGE * 2);
        const int kk = u / UB, h = u % UB;
        const unsigned flip = kk ? 64u : 0u;
        if (h == 0) {
#pragma unroll
            for (int b = 0; b < NT; ++b)
                asm volatile("ds_read_b128 %0, %1" : "=v"(pw[kk & 1][b]) : "v"((((unsigned)woff[b] * 2u) ^ flip) + sbase) : "memory");
        }
#pragma unroll
        for (int a = 0; a < XB; ++a)
            asm volatile("ds_read_b128 %0, %1" : "=v"(px[u & 1][a]) : "v"((((unsigned)xoff[h * XB + a] * 2u) ^ flip) + sbase) : "memory");
    };
    auto landed = [&](int u) {                             // the reads request(.., u) issued have returned
        const int kk = u / UB, h = u % UB;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int a = 0; a < XB; ++a) asm volatile("" : "+v"(px[u & 1][a]));
        if (h == 0) {
#pragma unroll
            for (int b = 0; b < NT; ++b) asm volatile("" : "+v"(pw[kk & 1][b]));
        }
    };
    bool pend = false;
    // sync point of stage k: its DMA has landed for every wave, and every wave is done reading the stage before it
    auto sync_stage = [&](int k, int ck_k, int ct_k) {
        wait_landed(min(nsteps - 1 - k, NBUF - 2), st_prev);
        st_prev = 0;
        __builtin_amdgcn_s_barrier();
        if constexpr (HASOP) {
            // residual / GELU' operand / DropPath scale of the tile that step k completes: loaded now, in front of the DMA, used
            // after the step's MFMAs
            if (ck_k == nk - 1) nt_epilogue_prefetch<MT, NT>(p, pre, (ct_k / p.tiles_n) * BM + wm * 128, (ct_k % p.tiles_n) * BN + wn * WN, li, lg);
        }
        if constexpr (PIPE) {
            pend = k + NBUF - 1 < nsteps;                  // the stage at the issue cursor goes out piecewise: issue_group
        } else {
            if (k + NBUF - 1 < nsteps) issue_next();
        }
    };
    // PIPE: the DMA pieces of the stage that sync point P(k) releases a slot for go out in NG groups, one in front of the MFMAs
    // of each of the next NG micro-batches (the one P(k) sits in, then the first ones of step k).  What the s_memtime stamps say
    // about a K step of the plain loop (256 x 192 tile, cycles per wave): DMA issue 940-1370, fragment reads + MFMAs 1040-1300,
    // barrier + DMA wait 950-1150 -- and the issue phase is not address arithmetic (12 -> 2 VALU instructions per piece took it
    // from 1370 to 940): a piece occupies the CU's one vector-memory path for ~16 cycles (64 B per cycle), the eight waves issue
    // their 7 pieces together, each piece queues behind the other waves' (~125 cycles per piece).  The path's 900 cycles, the
    // LDS's ~1150 and the matrix cores' 1630 per K step add up instead of overlapping because the barrier keeps all waves in the
    // same phase.  Moving groups of pieces under later micro-batches trades issue-phase cycles for DMA-wait cycles (a stage has
    // one K step to land): worth 5-10 % at K >= 768, nothing at K = 384, a loss at K = 192; turns taken by wave class within a
    // micro-batch (SIMD partners in different classes): 0-5 % slower.
    constexpr int NG = BN == 256 ? 1 : 2;                  // same-call A/B of 1 / 2 / 3 groups, see launch_p256
    auto issue_group = [&](int g) {
        if (!pend) return;
        const unsigned kbyte = (unsigned)ik * (BK * 2);
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if ((i * NG) / NP == g) issue_piece(i, islot, kbyte);
        if (g == NG - 1) {
            issue_bias(ipar);
            issue_advance();
        }
    };
    if constexpr (PIPE) {
        static_assert(BK == 64 && BATCH, "pipelined reads: 64-deep K steps (the weight buffers alternate by K block)");
        if (nsteps > 0) {
            sync_stage(0, 0, first);
            issue_group(0);
            request(0, 0);
            landed(0);
        }
    }
    for (int s = 0; s < nsteps; ++s) {
        if constexpr (PIPE) {
            const int nslot = cslot + 1 == NBUF ? 0 : cslot + 1;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int grp_now = -1;                          // the DMA group that goes out under this micro-batch's MFMAs
                if (u + 1 < U) {
                    request(cslot, u + 1);
                    if (u + 1 < NG) grp_now = u + 1;
                } else if (s + 1 < nsteps) {
                    const bool wrap = ck + 1 == nk;
                    sync_stage(s + 1, wrap ? 0 : ck + 1, wrap ? ct + tstep : ct);
                    request(nslot, 0);
                    grp_now = 0;
                }
                __builtin_amdgcn_sched_barrier(0);
                if (grp_now >= 0) issue_group(grp_now);
                {
                    const int kk = u / UB, h = u % UB;
#pragma unroll
                    for (int a = 0; a < XB; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b)
                            acc[h * XB + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw[kk & 1][b], px[u & 1][a], acc[h * XB + a][b], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (u + 1 < U) landed(u + 1);
                else if (s + 1 < nsteps) landed(0);
            }
        } else {
            sync_stage(s, ck, ct);
            compute(cslot);
        }
        cslot = cslot + 1 == NBUF ? 0 : cslot + 1;
        if (++ck == nk) {
            const int m0 = (ct / p.tiles_n) * BM, n0 = (ct % p.tiles_n) * BN;
            if (p.bias) {
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_s + cpar * 256 + wn * WN + chan_of<CW>(b, lg, 0));
#pragma unroll
                    for (int a = 0; a < MT; ++a) acc[a][b] += bb;
                }
            }
            // (GELU + pre-activation through the block-wide slabs of the narrower tiles: FMMT_NT_P256_LDSGELU=1; measured 281 -> 292 us, off.)
            // Round 6: the GELU launches (+ pre-activation or derivative: TWO output tensors) of the 256-wide tile take the wave-private slabs as well --
            // straight from the accumulator layout they were 16-row x 64-byte partial-line stores, twice (FMMT_NT_P256_WAVEGELU=0: as before).
            const bool gelu_any = p.epi == FMMT_EPI_GELU || p.epi == FMMT_EPI_GELU_DG;
            if (BN == 256 && !HASOP && !p.part && !(p.reserved & 4) && ((p.epi == 0 && !p.y_pre) || (gelu_any && !(p.reserved & 128)))) {
                // 256-wide tile: a wave's 64 channels are one 128-byte line, so the transposition is wave-private -- 16 token rows
                // at a time through this wave's own 2.3 KB slab, no barrier -- and serves GELU + pre-activation (two tensors) too
                if constexpr (BN == 256) {
                    char* ws = smem + (size_t)NBUF * STAGE * sizeof(T) + 2 * 256 * sizeof(float) + wave * (16 * 144);
                    T* __restrict__ ypre = reinterpret_cast<T*>(p.y_pre);
                    const int mw = m0 + wm * 128, nw = n0 + wn * WN;
                    const int rr = lane >> 3, rc = lane & 7;           // read side: 8 rows x 8 chunks of 16 bytes per instruction
                    auto emit = [&](int a, T* dst, bool gelu) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            float t[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) t[e] = acc[a][2 * c + (e >> 2)][e & 3];
                            if (gelu) gelu_inplace<T>(t, 8);
                            bf16x8 v;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (bf16)t[e];
                            *reinterpret_cast<bf16x8*>(ws + li * 144 + chan_of<CW>(2 * c, lg, 0) * 2) = v;
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int m = mw + a * 16 + h * 8 + rr;
                            const bf16x8 v = *reinterpret_cast<const bf16x8*>(ws + (h * 8 + rr) * 144 + rc * 16);
                            // non-temporal: whole lines of an output that streams to memory once -- as write-allocating stores a tile round's output fills the
                            // XCD's L2 with dirty lines and evicts the operand panels (round 5, same call: 125440 x 1152 x 384 165 -> 131 us, x 1536 x 384
                            // 182 -> 158; not for the partial-line stores of nt_epilogue: 263 -> 316 us there)
                            if (m < p.M) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst + (size_t)m * p.ldy + nw + rc * 8));
                        }
                    };
                    // FMMT_EPI_GELU_DG: activation and derivative of a 16-row tile from ONE evaluation (shared exponential), through the slab one after the other
                    // (a wave's LDS operations execute in order: write, read, write, read without waits between them)
                    auto emit_dg = [&](int a) {
                        bf16x8 dv[2];
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            float t[8], d[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) t[e] = acc[a][2 * c + (e >> 2)][e & 3];
                            gelu_both_inplace<T>(t, d, 8);
                            bf16x8 v;
#pragma unroll
                            for (int e = 0; e < 8; ++e) { v[e] = (bf16)t[e]; dv[c][e] = (bf16)d[e]; }
                            *reinterpret_cast<bf16x8*>(ws + li * 144 + chan_of<CW>(2 * c, lg, 0) * 2) = v;
                        }
#pragma unroll
                        for (int pass = 0; pass < 2; ++pass) {
                            T* dst = pass == 0 ? yg : ypre;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int m = mw + a * 16 + h * 8 + rr;
                                const bf16x8 v = *reinterpret_cast<const bf16x8*>(ws + (h * 8 + rr) * 144 + rc * 16);
                                if (m < p.M && dst) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst + (size_t)m * p.ldy + nw + rc * 8));
                            }
                            if (pass == 0) {
#pragma unroll
                                for (int c = 0; c < 2; ++c) *reinterpret_cast<bf16x8*>(ws + li * 144 + chan_of<CW>(2 * c, lg, 0) * 2) = dv[c];
                            }
                        }
                    };
                    const bool two = gelu_any && ypre != nullptr;
#pragma unroll
                    for (int a = 0; a < MT; ++a) {
                        if (p.epi == FMMT_EPI_GELU_DG) {
                            emit_dg(a);
                        } else {
                            if (two) emit(a, ypre, false);
                            emit(a, yg, p.epi == FMMT_EPI_GELU);
                        }
                    }
                    if (!PIPE && mw + 128 <= p.M) st_prev = MT * 2 * (two ? 2 : 1);   // (PIPE: DMA pieces follow the stores, see issue_group)
                }
            } else if (lds_epi) {
                const int rows_left = p.M - (m0 + wm * 128);             // token rows of this wave group that exist
                T* __restrict__ ypre = reinterpret_cast<T*>(p.y_pre);
                // one slab round: the waves park `which` (0: the values as they are, 1: their GELU) of PR rows, then store whole rows
                auto round = [&](int pass, T* dst, bool gelu) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                          // the slab's previous rows have been read
#pragma unroll
                    for (int f = 0; f < PR / 16; ++f) {
                        const int a = pass * (PR / 16) + f;
                        char* row = scratch + (f * 16 + li) * SP + (wn * WN) * 2;
#pragma unroll
                        for (int c = 0; c < NT / 2; ++c) {
                            float t[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) t[e] = acc[a][2 * c + (e >> 2)][e & 3];
                            if (gelu) gelu_inplace<T>(t, 8);
                            bf16x8 v;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (bf16)t[e];
                            *reinterpret_cast<bf16x8*>(row + chan_of<CW>(2 * c, lg, 0) * 2) = v;
                        }
                        if constexpr (NT % 2) {
                            float t[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) t[e] = acc[a][NT - 1][e];
                            if (gelu) gelu_inplace<T>(t, 4);
                            bf16x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (bf16)t[e];
                            *reinterpret_cast<bf16x4*>(row + chan_of<CW>(NT - 1, lg, 0) * 2) = v;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's chunks are in the slab ...
                    __builtin_amdgcn_s_barrier();                          // ... and so is everybody else's
#pragma unroll
                    for (int i = 0; i < NSL; ++i) {
                        const int q = i * 256 + wn * 64 + lane, r = q / CPR, cc = q - r * CPR;
                        const int row = pass * PR + r;
                        if (row < rows_left) {
                            const bf16x8 v = *reinterpret_cast<const bf16x8*>(scratch + r * SP + cc * 16);
                            __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst + (size_t)(m0 + wm * 128 + row) * p.ldy + n0 + cc * 8));
                        }
                    }
                };
                const bool two = lds_gelu && ypre != nullptr;
#pragma unroll
                for (int pass = 0; pass < 128 / PR; ++pass) {
                    if (two) round(pass, ypre, false);
                    round(pass, yg, lds_gelu);
                }
                // a wave group with all its 128 rows inside M issued exactly this many stores per wave (ragged panel: unknown -> 0)
                if (!PIPE && rows_left >= 128) st_prev = (128 / PR) * NSL * (two ? 2 : 1);
            } else if constexpr (HASOP) {
                nt_epilogue<T, MT, NT, true, true>(p, acc, m0 + wm * 128, n0 + wn * WN, li, lg, &pre);
            } else {
                nt_epilogue<T, MT, NT, true, false, true>(p, acc, m0 + wm * 128, n0 + wn * WN, li, lg);      // no operands: see p256_plan
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            ck = 0;
            ct += tstep;
            cpar ^= 1;
        }
    }
}

template <int BN, int BK, int NBUF, bool BATCH, bool HASOP, bool PIPE = false>
int launch_p256_b(const LinArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)NBUF * (BN + 256) * BK * 2 + 2 * 256 * sizeof(float) + (BN == 256 ? 8 * 16 * 144 : 2 * (BN == 192 ? 32 : 16) * (BN * 2 + 16));   // ring, bias slabs, epilogue scratch (BN = 256: wave-private slabs)
    static_assert(lds <= 160 * 1024, "LDS");
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&linear_nt_p256_kernel<BN, BK, NBUF, BATCH, HASOP, PIPE>), (int)lds)) return rc_;
    LinArgs p = a;
    p.tiles_m = (a.M + 255) / 256;
    p.tiles_n = a.N / BN;
    // FMMT_NT_P256_LDSEPI=0: epilogue stores straight from the accumulator layout (A/B switch)
    static const int lds_epi = fmmt_const("FMMT_NT_P256_LDSEPI", 1);
    static const int lds_gelu = fmmt_const("FMMT_NT_P256_LDSGELU", 0);   // measured slower (two tensors, 16-row passes: 32 barriers per tile): 273 -> 296 us
    static const int wrows = fmmt_const("FMMT_NT_P256_WROWS", 1);
    static const int wave_gelu = fmmt_const("FMMT_NT_P256_WAVEGELU", 1);   // round 6: the two-tensor GELU epilogues of the 256-wide tile through the wave-private slabs
    p.reserved = (lds_epi ? 0 : 4) | (lds_gelu ? 0 : 8) | (wrows ? 0 : 64) | (wave_gelu ? 0 : 128);
    hipLaunchKernelGGL((linear_nt_p256_kernel<BN, BK, NBUF, BATCH, HASOP, PIPE>), dim3(256), dim3(512), lds, st, p);
    FMMT_CHECK_LAUNCH();
    return 0;
}
template <int BN, int BK, int NBUF>
int launch_p256(const LinArgs& a, hipStream_t st) {
    // FMMT_NT_P256_BATCH=0: fragment reads left to the compiler's schedule (A/B switch)
    static const int batch = fmmt_const("FMMT_NT_P256_BATCH", 1);
    if constexpr (BN != 256) {
        // launches with an M x N epilogue operand or a DropPath scale: operand prefetched into registers (48 / 32 of them:
        // no room beside the 128 accumulators of the 256-wide tile)
        if (a.res || a.aux || a.rowscale) return launch_p256_b<BN, BK, NBUF, true, true>(a, st);
        // FMMT_NT_P256_PLAINOP: 1 = operand-free launches whose epilogue stores directly (K > 1536) take the operand-prefetch
        // instantiation as well (round 2: 31360 x 768 x 3072 160 -> 153 us); against the pipelined loop below it loses (135 -> 127 us,
        // same call): 0
        static const int plainop = fmmt_const("FMMT_NT_P256_PLAINOP", 0);
        if (plainop && !a.part && (a.K > 1536 || plainop > 1)) return launch_p256_b<BN, BK, NBUF, true, true>(a, st);
    }
    if constexpr (BK == 64) {
        // Pipelined fragment reads + DMA pieces in groups (PIPE), for K >= 384.  Same-call A/B against the plain loop, us per
        // launch: 31360 x 2304 x 768 124 -> 115, 31360 x 3072 x 768 150 -> 141, 31360 x 768 x 768 44.4 -> 41.2, 125440 x 1536 x 384
        // (256-wide tile, one group) 190 -> 178, 125440 x 384 x 384 52.7 -> 50.2, 125440 x 1152 x 384 / 384 x 1536 / 384 x 1152 +-1 %;
        // K = 192 (three K steps per tile: the late pieces are waited for) 210 -> 220, stays on the plain loop.
        static const int pipe = fmmt_const("FMMT_NT_P256_PIPE", 1);
        if (pipe && a.K >= 384) return launch_p256_b<BN, BK, NBUF, true, false, true>(a, st);
    }
    return batch ? launch_p256_b<BN, BK, NBUF, true, false>(a, st) : launch_p256_b<BN, BK, NBUF, false, false>(a, st);
}

// Tile choice for the persistent kernel: the widest channel tile that divides N, unless a narrower one fills the last
// round of 256 workgroups better (31360 tokens x 768 channels: 369 tiles of 256 x 256 = 2 rounds at 72 %, 492 tiles of
// 256 x 192 = 2 rounds at 96 %).  Returns 0 if the shape is not for this kernel.
int p256_plan(const LinArgs& a) {
    static const int mode = fmmt_const("FMMT_NT_P256", 1);       // 0: off; 256 / 192 / 128: force that tile
    // K % 64 != 0 (Swin stage 0: K = 96, three K steps of 32): only with FMMT_NT_P256_K32=1 (A/B switch)
    static const int k32 = fmmt_const("FMMT_NT_P256_K32", 0);
    static const int minm = fmmt_const("FMMT_NT_P256_MINM", 16384);      // fewest tokens for this kernel
    if (!mode || a.ksplit || a.M < minm || a.M % 16 || a.K % 32 || a.K < 96 || a.ldx % 8 || a.ldw % 8) return 0;
    if (a.K % 64 && !k32) return 0;
    if ((unsigned long long)a.M * a.ldx >= (1ull << 31) || (unsigned long long)a.N * a.ldw >= (1ull << 31)) return 0;   // 32-bit byte offsets of the DMA pieces
    // One workgroup per CU has nothing to hide an epilogue's own M x N loads behind (residual, GELU' operand, DropPath
    // scale: the in-order vmcnt also makes them wait for the DMA stages in flight).  Measured on MI355X
    // (profiles/r02_gemm_shapes.txt): plain / bias / GELU + pre-activation launches gain 5-50 % over the two-workgroup
    // 256 x 128 kernels (125440 x 1536 x 384 GELU: 0.439 -> 0.295 ms), launches with such loads lose 0-25 %: those stay
    // on the older kernels.  FMMT_NT_P256=2 sends them here as well (A/B switch).
    // FMMT_NT_P256_OPS: 1 (default) = launches with a residual and / or DropPath scale come here too, on the 192- / 128-wide
    // tiles, with the operand prefetched into registers one K step before the epilogue (HASOP): measured, same call, 321 -> 262 us
    // (501760 x 192 x 768), 210 -> 201 (125440 x 384 x 1536); 2 = GELU' launches as well: 338 -> 391 / 530 -> 614 us -- that
    // epilogue is ~11 k VALU cycles per wave tile against 4.6 k MFMA cycles of a K = 384 tile, and one workgroup per CU has no
    // second workgroup whose K loop could run under it; 0 = none.
    // (round 4: mode 2 re-measured with gelu' from an LDS table in this kernel's unused epilogue scratch instead of the polynomial: 125440 x 1536 x 384
    //  328 -> 364 us, 501760 x 768 x 192 522 -> 561, 31360 x 3072 x 768 267 -> 253 -- still a loss where it matters; stays 1.
    //  And once more with the polynomial GELU' that replaced the tables: 314 -> 340, 491 -> 529, 258 -> 241 us.)
    static const int ops_mode = fmmt_const("FMMT_NT_P256_OPS", 1);
    const bool has_op = a.res || a.aux || a.rowscale;
    if (has_op && (!ops_mode || (a.aux && (ops_mode < 2 || a.res)) || a.ldres % 8 || a.ldaux % 8)) return 0;
    const int tm = (a.M + 255) / 256;
    int best = 0;
    double best_cost = 0;
    const int cand[3] = {256, 192, 128};
    const double pen[3] = {1.0, 1.04, 1.10};
    for (int i = 0; i < 3; ++i) {
        const int bn = cand[i];
        if (a.N % bn) continue;
        if (has_op && bn == 256) continue;
        if (mode > 2 && mode != bn) continue;
        const int tiles = tm * (a.N / bn);
        if (tiles < 256) continue;
        const double cost = (double)((tiles + 255) / 256) * bn * pen[i];
        if (!best || cost < best_cost) {
            best = bn;
            best_cost = cost;
        }
    }
    return best;
}

template <typename T, int BM, int BN, int BK, int NBUF, bool GLDS = false, bool SEG = false>
int launch_nt(const LinArgs& a, hipStream_t st) {
    constexpr int VEC = Vec<T>::N;
    constexpr size_t lds = (size_t)NBUF * (BM + BN) * (GLDS ? BK : BK + VEC) * sizeof(T);
    static FmmtLdsOnce lds_once;
    if (lds > 65536) {
        if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&linear_nt_kernel<T, BM, BN, BK, NBUF, GLDS, SEG>), (int)lds)) return rc_;
    }
    LinArgs p = a;
    p.tiles_n = (a.N + BN - 1) / BN;
    p.tiles_m = (a.M + BM - 1) / BM;
    const int grid = p.tiles_m * p.tiles_n;
    const int splits = a.ksplit ? (a.K + a.ksplit - 1) / a.ksplit : 1;
    hipLaunchKernelGGL((linear_nt_kernel<T, BM, BN, BK, NBUF, GLDS, SEG>), dim3(grid, splits), dim3(256), lds, st, p);
    FMMT_CHECK_LAUNCH();
    return 0;
}

template <typename T, int BM, int BN>
int dispatch_nt_bk(const LinArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        // Swin stage 0: one K step.  (64- and 256-row tiles measured: no gain / -35 %.)  These launches are HBM streams
        // (40-70 FLOP/B); epilogue-shaped stores alone reach 5.5 TB/s on this part (profiles/r01_hbm_calibration.txt), what
        // holds a single-step workgroup at ~3 TB/s is its load -> LDS -> MFMA -> store chain with nothing to overlap, so
        // occupancy is what counts (NtWaves above); the fc1 + GELU + pre-activation launch is the only one still on it.
        if (!a.ksplit && a.K == 96) return launch_nt<T, BM, BN, 96, 1>(a, st);
        if (!a.ksplit && a.K <= 64) return launch_nt<T, BM, BN, 64, 1>(a, st);     // PatchEmbed (K = 48)
        // measured on MI355X (tools/probes/gemm_bench.py, sum over the bench shapes): BK=64 5.33 ms vs BK=32 5.88 ms
        // direct global->LDS DMA staging: measured +7 % over register staging summed over the bench shapes, up to
        // +25 % on the K >= 768 ones (971 TF/s on 31360x768x3072); FMMT_NT_GLDS=0 selects the register-staged kernel
        static const int glds = fmmt_const("FMMT_NT_GLDS", 1);
        if (glds && a.K % 64 == 0 && !a.ksplit && a.ldx % 8 == 0 && a.ldw % 8 == 0) {
            // four-buffer ring for few-token problems with a long K loop and at most one workgroup per CU (fc2 of the
            // encoder FFNs, 512-1328 tokens x 768 x 3072: 32 -> 25 us); with more tiles than CUs the 96 KB ring costs
            // co-residency (1328 x 3072 x 768: 15 -> 21 us), and at K = 768 it is a wash.  FMMT_NT_GLDS=2: never.
            if constexpr (BM == 64) {
                const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
                if (glds != 2 && a.K >= 2048 && tiles <= 256) return launch_nt<T, BM, BN, 64, 4, true>(a, st);
            }
            return launch_nt<T, BM, BN, 64, 2, true>(a, st);
        }
        if (a.K % 64 == 0 && (!a.ksplit || a.ksplit % 64 == 0)) return launch_nt<T, BM, BN, 64, 2>(a, st);
        return launch_nt<T, BM, BN, 32, 2>(a, st);
    } else {
        return launch_nt<T, BM, BN, 16, 2>(a, st);                                 // fp32 parity path
    }
}

// The phase-structured kernels (gemm_ph.h: 256 x 256 tiles, four phases; gemm_ph3.h: 192 x 256 tiles, three phases, all epilogues but GELU').
// Returns 0 (none), 1 (gemm_ph.h, plain / bias), 3 (gemm_ph3.h, 192 x 256 tiles) or 4 (gemm_ph3.h, 384 x 128 tiles); *epi3 = gemm_ph3.h's EPI template value:
// 2 plain / bias, 3 GELU + pre-activation, 5 residual / row scale, 6 product with the stored GELU derivative (+ row scale).
// Measured against the kernels below and each other (tools/probes/nt_ph_probe.hip, same call, us per launch, production -> ph / ph3):
//   plain / bias     31360 x 768 x 3072 136 -> 139 / 110 (369 tiles of 256 x 256 = 1.44 rounds of 256 workgroups, 492 of 192 x 256 = 1.92), x 768 x 768 42 -> 44 / 35,
//                    x 2304 x 768 113 -> 101 / 100, x 3072 x 768 143 -> 129 / 132, 125440 x 1536 x 384 190 -> 154 / 167, 7840 x 1536 x 6144 168 -> 134 / 106,
//                    7840 x 6144 x 1536 186 -> 116 / 123, 7840 x 1536 x 1536 38.5 (128-row kernels) -> 38 / 31: the tile whose rounds waste less wins;
//   res + row scale  31360 x 768 x 3072 156 -> 117, x 768 x 768 56 -> 42, x 3072 x 768 235 -> 161, 125440 x 1536 x 384 318 -> 242, 7840 x 1536 x 6144 169 -> 110 (ph3 only);
//   GELU + pre       wins from K = 1536 (7840 x 6144 x 1536 211 -> 184, 31360 x 768 x 3072 159 -> 141), loses below (31360 x 3072 x 768 195 -> 254: the polynomial runs
//                    behind the MFMAs of a phase, on the workgroup's critical path, and a short K loop has few phases to amortise it);
//   GELU'            loses everywhere (262 -> 828): stays on the persistent kernel.
int ph_plan(const LinArgs& a, int* epi3) {
    static const int on = fmmt_const("FMMT_NT_PH", 1);
    const bool mul_aux = a.epi == FMMT_EPI_MUL_AUX && a.aux && !a.res && !a.y_pre && !a.bias;
    if (!on || a.ksplit || a.part || (a.aux && !mul_aux)) return 0;
    const bool gelu_pre = a.epi == FMMT_EPI_GELU && a.y_pre, has_op = a.res || a.rowscale;
    if (!gelu_pre && !mul_aux && (a.epi != 0 || a.y_pre)) return 0;
    if (gelu_pre && (has_op || a.K < 1536)) return 0;
    if (mul_aux && (a.ldaux % 8 || (unsigned long long)a.M * a.ldaux >= (1ull << 31))) return 0;
    if (a.M < 4096 || a.M % 8 || a.N % 128 || a.K % 64 || a.K < 192 || a.ldx % 8 || a.ldw % 8 || a.ldy % 8) return 0;
    if (a.res && a.ldres % 8) return 0;
    if (a.rowscale && a.rows_per_scale <= 0) return 0;
    if ((unsigned long long)a.M * a.ldx >= (1ull << 31) || (unsigned long long)a.N * a.ldw >= (1ull << 31) || (unsigned long long)a.M * a.ldy >= (1ull << 31) ||
        (a.res && (unsigned long long)a.M * a.ldres >= (1ull << 31)))
        return 0;
    if (a.N % 256) {
        // channel counts that are multiples of 128 only (Swin stage 2: 384, 1152): gemm_ph3.h's 384-token x 128-channel layout (returns 4).  Measured at 125440 tokens
        // (production -> ph3): x 1152 x 384 134 -> 129, x 384 x 384 45.5 -> 43.6, with residual + row scale 240 -> 204 / 79 -> 58, row scale alone 59 -> 54.5;
        // at K = 1152 / 1536 it loses (164 -> 166, with residual 181 -> 196: three channel tiles per token panel, 96 FLOP per staged byte): K <= 512 only.
        if (gelu_pre || a.K > 512) return 0;
        const long long t384 = (long long)((a.M + 383) / 384) * (a.N / 128), r384 = (t384 + 255) / 256;
        if (t384 < 256 || t384 * 100 < r384 * 256 * 85) return 0;
        *epi3 = mul_aux ? 6 : has_op ? 5 : 2;
        return 4;
    }
    const long long t256 = (long long)((a.M + 255) / 256) * (a.N / 256), t192 = (long long)((a.M + 191) / 192) * (a.N / 256);
    const long long r256 = (t256 + 255) / 256, r192 = (t192 + 255) / 256;
    const long long min_tiles = a.M < 16384 ? 150 : 256;        // 16384+ tokens: the persistent 256-row kernel's ground, taken only with (nearly) full rounds
    const bool ok256 = t256 >= min_tiles && (a.M < 16384 || t256 * 100 >= r256 * 256 * 85);
    const bool ok192 = t192 >= min_tiles && (a.M < 16384 || t192 * 100 >= r192 * 256 * 85);
    *epi3 = gelu_pre ? 3 : mul_aux ? 6 : has_op ? 5 : 2;
    if (gelu_pre || has_op || mul_aux) return ok192 ? 3 : 0;
    // plain / bias: rows of MFMA work per workgroup over the launch; the 192-row tile is charged 3 % (shorter phases per byte staged)
    const long long c256 = r256 * 256 * 100, c192 = r192 * 192 * 103;
    if (ok192 && (!ok256 || c192 < c256)) return 3;
    return ok256 ? 1 : 0;
}

template <typename T>
int dispatch_nt(const LinArgs& a, hipStream_t st) {
    // BN = 96 when it tiles N exactly and 128 would not (C = 96, 288, 192, 576 ...)
    const bool n96 = (a.N % 96 == 0) && (a.N % 128 != 0);
    // few-token problems (cross-modal encoder: 152..1280 rows; embedding head: 640 rows) use 64-row tiles so
    // that twice as many workgroups share the work; the multi-million-token Swin GEMMs use 128-row tiles
    // (few rows but tens of thousands of output channels -- the input gradient of the 37632 -> 512 embedding head: there are
    //  workgroups enough, 128-row tiles read each weight slab half as often; FMMT_NT_WIDE64=1 keeps the 64-row tiles for it)
    static const int wide64 = fmmt_const("FMMT_NT_WIDE64", 0);
    if (a.M <= 4096 && (a.N < 16384 || a.M < 256 || wide64)) {
        if constexpr (sizeof(T) == 2) {
            // Fewer 64 x 128 tiles than half the CUs (the fusion stack: 152-1328 tokens x 768 channels = 18-126 tiles): such a
            // launch is a chain of K / 64 steps whose cost is the step's DMA issue (six 1-KB pieces per wave) plus a barrier, on a
            // mostly idle GPU.  Quarter tiles (32 x 64) put four times the workgroups on the chip, each with half the pieces per
            // wave and step.  Measured per launch inside a graph (tools/probes/few_probe.py, same call): 152-640 x 768 x 768 8.5 -> 5.5 us
            // (hipBLASLt 7.5-8.1), 1328 x 768 x 768 9.2 -> 6.7, 664 x 1536 x 768 9.0 -> 6.5, 512 x 3072 x 768 12.9 -> 9.8,
            // 512 x 768 x 3072 24.4 -> 16, 1328 x 768 x 3072 25.5 -> 19.5; with 256+ tiles of 64 x 128 the quarter tiles lose
            // (1328 x 3072 x 768: 14 -> 22 us), and the four-buffer ring does nothing for K = 768.
            // FMMT_NT_SMALL: 1 (default) = 32 x 64, 2 = 64 x 64 (6.4 us on the first group), 0 = off.
            static const int small = fmmt_const("FMMT_NT_SMALL", 1);
            const int tiles = ((a.M + 63) / 64) * ((a.N + 127) / 128);
            static const int small_tiles = fmmt_const("FMMT_NT_SMALL_TILES", 256);
            static const int small_r4k = fmmt_const("FMMT_NT_SMALL_R4K", 2048);
            if (small && tiles < small_tiles && a.N % 64 == 0 && a.K % 64 == 0 && a.K >= 128 && !a.ksplit && a.ldx % 8 == 0 && a.ldw % 8 == 0) {
                if (small == 2) {
                    if (a.K >= 2048 && 2 * tiles <= 256) return launch_nt<T, 64, 64, 64, 4, true>(a, st);
                    return launch_nt<T, 64, 64, 64, 2, true>(a, st);
                }
                if (a.K >= small_r4k) return launch_nt<T, 32, 64, 64, 4, true>(a, st);
                return launch_nt<T, 32, 64, 64, 2, true>(a, st);
            }
        }
        return n96 ? dispatch_nt_bk<T, 64, 96>(a, st) : dispatch_nt_bk<T, 64, 128>(a, st);
    }
    if constexpr (sizeof(T) == 2) {
        int epi3 = 2;
        if (const int ph = ph_plan(a, &epi3)) {
            if (ph == 1) return launch_ph<2>(a, st);
            if (ph == 4) return epi3 == 6 ? launch_ph3<6, true, 4>(a, st) : epi3 == 5 ? launch_ph3<5, true, 4>(a, st) : launch_ph3<2, true, 4>(a, st);
            return epi3 == 3 ? launch_ph3<3>(a, st) : epi3 == 6 ? launch_ph3<6>(a, st) : epi3 == 5 ? launch_ph3<5>(a, st) : launch_ph3<2>(a, st);
        }
        if (const int bn = p256_plan(a)) {
            // FMMT_NT_P256_RING: 1 (default) = K step 64, ring of 2 (3 for 128-wide tiles); 0 = K step 32, ring of 4.
            // Measured (profiles/r02_gemm_shapes.txt): the K-step-64 form wins on every shape by 3-17 % (half the barriers)
            static const int ring = fmmt_const("FMMT_NT_P256_RING", 1);
            if (ring == 0 || a.K % 64) return bn == 256 ? launch_p256<256, 32, 4>(a, st) : bn == 192 ? launch_p256<192, 32, 4>(a, st) : launch_p256<128, 32, 4>(a, st);
            return bn == 256 ? launch_p256<256, 64, 2>(a, st) : bn == 192 ? launch_p256<192, 64, 2>(a, st) : launch_p256<128, 64, 3>(a, st);
        }
        // measured (tools/probes/gemm_bench.py): +8 % on the 125440-token stage-2 shapes (K = 384: 496 -> 540 TF/s),
        // -4 % on the 31360-token stage-3 shapes (tile quantisation at one workgroup per CU) -> only for M >= 65536
        static const int deep = fmmt_const("FMMT_NT_DEEP", 1);
        static const int deep_mink = fmmt_const("FMMT_NT_DEEP_MINK", 96);
        static const int deep96 = fmmt_const("FMMT_NT_DEEP96", 1);
        // K = 96 (stage 0, three K steps): -5..7 % with the deep kernel except for the GELU + pre-activation launch
        // (two output streams; measured +2 %), which keeps the single-step kernel
        static const int deep_minm = fmmt_const("FMMT_NT_DEEP_MINM", 65536);
        const bool big = deep && a.M >= deep_minm && !a.ksplit && a.K % 32 == 0 && a.K >= deep_mink && a.ldx % 8 == 0 && a.ldw % 8 == 0 &&
                         !(a.K == 96 && a.epi == FMMT_EPI_GELU);
        if (big && (a.N % 128 == 0 || (n96 && deep96)) && (a.K % 64 == 0 || deep == 1 || deep == 2)) {
            constexpr size_t lds = (size_t)3 * (256 + 128) * 64 * 2;
            constexpr size_t lds96 = (size_t)3 * (256 + 96) * 32 * 2;
            static FmmtLdsOnce lds_once[7];
            {
                const void* fns[] = {reinterpret_cast<const void*>(&linear_nt_deep_kernel),
                                     reinterpret_cast<const void*>(&linear_nt_deep32_kernel<0, 128>),
                                     reinterpret_cast<const void*>(&linear_nt_deep32_kernel<6, 128>),
                                     reinterpret_cast<const void*>(&linear_nt_deep32_kernel<3, 128>),
                                     reinterpret_cast<const void*>(&linear_nt_deep32_kernel<0, 96>),
                                     reinterpret_cast<const void*>(&linear_nt_deep32_kernel<6, 96>),
                                     reinterpret_cast<const void*>(&linear_nt_deep32_kernel<3, 96>)};
                const int sizes[] = {(int)lds, (int)lds / 2, (int)lds / 2, (int)lds / 2, (int)lds96, (int)lds96, (int)lds96};
                for (int i = 0; i < 7; ++i)
                    if (int rc_ = lds_once[i].set(fns[i], sizes[i])) return rc_;
            }
            LinArgs p = a;
            p.tiles_m = (a.M + 255) / 256;
            static const int slab = fmmt_const("FMMT_NT_SLAB", 1);
            static const int wslab = fmmt_const("FMMT_NT_WSLAB", 1);
            p.reserved = (slab ? 0 : 16) | (wslab ? 0 : 32);
            if (n96) {
                p.tiles_n = a.N / 96;
                if (a.K == 192) hipLaunchKernelGGL((linear_nt_deep32_kernel<6, 96>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds96, st, p);
                else if (a.K == 96) hipLaunchKernelGGL((linear_nt_deep32_kernel<3, 96>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds96, st, p);
                else hipLaunchKernelGGL((linear_nt_deep32_kernel<0, 96>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds96, st, p);
                FMMT_CHECK_LAUNCH();
                return 0;
            }
            p.tiles_n = a.N / 128;
            // measured (tools/probes/gemm_bench.py, 125440 tokens): the K-step-32 kernel with two workgroups per CU wins
            // where the epilogue is a large share of a tile's life -- GELU / GELU' launches (0.388 -> 0.340 ms) and
            // K = 384 (384x384: 0.069 -> 0.057 ms); the K-step-64 kernel keeps a 2-4 % edge on plain K >= 1152.
            // FMMT_NT_DEEP: 1 = this policy, 2 = always K step 32, 4 = always K step 64, 0 = 128-row kernels only
            if (deep == 2 || (deep == 1 && (a.epi != 0 || a.K <= 512)) || a.K % 64 != 0) {
                if (a.K == 192) hipLaunchKernelGGL((linear_nt_deep32_kernel<6, 128>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds / 2, st, p);
                else if (a.K == 96) hipLaunchKernelGGL((linear_nt_deep32_kernel<3, 128>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds / 2, st, p);
                else hipLaunchKernelGGL((linear_nt_deep32_kernel<0, 128>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds / 2, st, p);
            } else {
                hipLaunchKernelGGL(linear_nt_deep_kernel, dim3(p.tiles_m * p.tiles_n), dim3(512), lds, st, p);
            }
            FMMT_CHECK_LAUNCH();
            return 0;
        }
    }
    // (256-row tiles with 4 waves measured: -30 % on the stage-2/3 shapes -- one workgroup per CU)
    return n96 ? dispatch_nt_bk<T, 128, 96>(a, st) : dispatch_nt_bk<T, 128, 128>(a, st);
}

// y = T(sum_s part[s] + bias) for the split-K path
template <typename T>
__global__ void splitk_finish_kernel(const float* __restrict__ part, int splits, int M, int N, const float* __restrict__ bias,
                                     T* __restrict__ y, int ldy) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
    float t = bias ? bias[n] : 0.f;
    for (int s = 0; s < splits; ++s) t += part[(size_t)s * M * N + i];
    y[(size_t)m * ldy + n] = from_f32<T>(t);
}

// split-K plan for skinny problems (the 37632 -> 512 head): few output tiles, very long K
int splitk_plan(int M, int N, int K, int* ksplit) {
    const int bm = M <= 4096 ? 64 : 128;
    const int tiles = ((M + bm - 1) / bm) * ((N + 127) / 128);
    if (tiles >= 128 || K < 4096) { *ksplit = 0; return 1; }
    int splits = 512 / tiles;
    if (splits > 32) splits = 32;
    int ks = (K + splits - 1) / splits;
    ks = (ks + 63) / 64 * 64;
    *ksplit = ks;
    return (K + ks - 1) / ks;
}

// ---------------------------------------------------------------------------------------------
// TN kernel (weight gradient).  Output tile 128 (n) x 128 (k); the contraction runs over tokens m,
// which is the *row* index of both operands, so MFMA fragments need 8 consecutive m for one channel:
// bf16 uses the gfx950 LDS transpose read (ds_read_b64_tr_b16) on tiles kept in natural
// [m][channel] layout; f32 reads single dwords.  A and B use the same read pattern, so the order of
// m inside a fragment cancels in the contraction.
// ---------------------------------------------------------------------------------------------
struct TnArgs {
    int M, N, K;
    const void* dy; int lddy;
    const void* x; int ldx;
    float* part_w;      // [splits][N][K]
    float* part_b;      // [splits][N] or nullptr
    const float* rowscale; int rows_per_scale;
    int tiles_k;
    int chunk;          // rows of m per split (multiple of the m step)
    int tiles_n;
    int xcd;            // 1: chunked XCD remap of the (split, tile) work list
    int x_gelu;         // 1: the x operand holds a pre-activation; contract with gelu(x) (recomputed activation of the fused Mlp)
    int* hdr;           // workspace header: hdr[0] receives the number of splits written (read by the finish pass); may be NULL
    int splits;
    int npad;           // rows of one split's partials (stride of part_w / part_b); 0 = N.  > N: linear_tn_dma_kernel<384,192> on an N that is 192 short of a tile
};

__device__ __forceinline__ bf16x8 lds_tr_frag(const bf16* s, int pitch, int c0, int li, int lg) {
    // 16-lane group lg covers token rows lg*8 .. lg*8+7; lane li supplies row (li>>2) (+4), column
    // chunk (li&3)*4 and receives column li of the 4x16 block (4 tokens).
    const bf16* a0 = s + (lg * 8 + (li >> 2)) * pitch + c0 + (li & 3) * 4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * pitch));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo;
    u.s.hi = hi;
    return u.v;
}

// bf16 tiles of the TN kernel: unpadded 256-byte rows (128 channels) whose eight 32-byte column blocks are XOR-permuted by
// g(row) = (row & 3) | ((row >> 3) & 1) << 2.  A ds_read_b64_tr_b16 serves lanes 0-31 in one LDS cycle: two 16-lane groups,
// i.e. token rows {0..3, 8..11} (or {4..7, 12..15}) x one 32-byte column block each -- with 256-byte rows every row starts on
// bank 0, and g() sends the eight rows of a cycle to eight different 8-bank slots: conflict-free, where the padded
// layout (272-byte pitch) was 2-way on every fragment read.  It is also 6 % smaller.
__device__ __forceinline__ int tn_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ bf16x8 lds_tr_frag_swz(const bf16* s, int row0, int c0, int li, int lg) {
    // rows row0 + lg*8 + (li>>2) and + 4; column block c0 / 16, this lane's 8 bytes at (li & 3) * 4 elements inside it
    const int r = row0 + lg * 8 + (li >> 2);
    const bf16* a0 = s + r * 128 + (((c0 >> 4) ^ tn_swz(r)) << 4) + (li & 3) * 4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 128));        // row + 4: same g()
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo;
    u.s.hi = hi;
    return u.v;
}

// FEW: instantiation used for few-token problems (cross-modal encoder, embedding head) -- same code, its own
// symbol, so that profiles keep the multi-million-token Swin launches and the tiny ones apart.
// PF: depth of the register prefetch (token steps in flight).  The 64-token-step instantiation is LDS-limited to two
// workgroups per CU whatever it does with registers, and with one step in flight it sits exactly on its bytes-in-flight
// bound (2 x 32 KB per CU / ~2 us = 8 TB/s x 64 FLOP/B = 0.52 PFLOP/s, measured 0.54): PF = 2 keeps two steps in flight.
template <int BMS> struct TnWaves { static constexpr int value = BMS >= 64 ? 2 : 3; };   // workgroups per CU that the LDS tiles allow
template <typename T, int BMS, bool FEW = false, int PF = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TnWaves<BMS>::value)))
void linear_tn_kernel(TnArgs p) {
    constexpr int VEC = Vec<T>::N;
    constexpr bool SWZ = sizeof(T) == 2;                // bf16: unpadded, XOR-swizzled rows (see tn_swz)
    constexpr int PITCH = SWZ ? 128 : 128 + VEC;
    constexpr int CV = 128 / VEC;                       // vectors per tile row
    constexpr int NV = BMS * CV / 256;                  // vectors per thread per operand
    constexpr int KM = sizeof(T) == 2 ? 32 : 4;         // token rows consumed per MFMA

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);                 // [2][BMS][PITCH]
    T* Bs = As + 2 * BMS * PITCH;                       // [2][BMS][PITCH]
    float* bsum = reinterpret_cast<float*>(Bs + 2 * BMS * PITCH);   // [256/CV][128]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    // 1-D grid of tiles x splits.  All output tiles of one split read the same token rows of dy and x, so a split's
    // tiles are kept on one XCD (chunked XCD remap over the split-major work list): with the head-major order each
    // XCD's L2 fetched every operand slab again (FETCH_SIZE 1.9x the algorithmic bytes on the stage-2 launches).
    const int tiles = p.tiles_n * p.tiles_k;
    const int logical = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int split = logical / tiles, tile = logical - split * tiles;
    const int tile_n = tile / p.tiles_k, tile_k = tile % p.tiles_k;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int mbeg = split * p.chunk, mend = min(p.M, mbeg + p.chunk);
    if (p.hdr && blockIdx.x == 0 && tid == 0) { p.hdr[0] = p.splits; p.hdr[1] = 0; p.hdr[2] = p.N; }

    const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const bool do_bias = (p.part_b != nullptr) && (tile_k == 0);

    struct Regs { Vec<T> a[NV], b[NV]; float s[NV]; };
    Regs R0, R1;
    float colsum[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) colsum[e] = 0.f;

    // (PMC, profiles/r02: 3.7 VALU instructions per MFMA.  Dropping the ones that guard -- unconditional loads for interior
    //  steps, one wave-uniform division per step for the DropPath sample index -- was measured in-step, same call, twice:
    //  7.29 / 7.30 ms against 6.92 / 6.72 ms for the stage-2/3 launches, 5.4 against 4.6 for stage 0/1: slower, not kept.)
    // DropPath sample index of this thread's token rows, advanced by BMS per gload() (the calls walk the split's steps in order):
    // one division per thread instead of one per vector and step
    int sidx[NV], srem[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int m = mbeg + (tid + i * 256) / CV;
        sidx[i] = p.rowscale ? m / p.rows_per_scale : 0;
        srem[i] = p.rowscale ? m - sidx[i] * p.rows_per_scale : 0;
    }
    auto gload = [&](Regs& R, int mb) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * 256;
            const int row = v / CV, c = (v % CV) * VEC;
            const int m = mb + row;
            const bool mv = m < mend;
            // the DropPath scale is only *fetched* here and applied in lstore(), after the MFMA block: applying it
            // right away makes every tile load wait for two dependent HBM round trips (measured: 2x slower launches)
            R.a[i] = (mv && n0 + c < p.N) ? ldvec<T>(dyg + (size_t)m * p.lddy + n0 + c) : zerovec<T>();
            if (p.rowscale) {
                R.s[i] = mv ? p.rowscale[sidx[i]] : 0.f;
                srem[i] += BMS;
                while (srem[i] >= p.rows_per_scale) {
                    srem[i] -= p.rows_per_scale;
                    ++sidx[i];
                }
            }
            R.b[i] = (mv && k0 + c < p.K) ? ldvec<T>(xg + (size_t)m * p.ldx + k0 + c) : zerovec<T>();
        }
    };
    auto lstore = [&](Regs& R, int buf) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * 256;
            const int row = v / CV, c = (v % CV) * VEC;
            if (p.rowscale) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) R.a[i].set(e, R.a[i].get(e) * R.s[i]);
            }
            const int cs = SWZ ? ((((c >> 4) ^ tn_swz(row)) << 4) | (c & 8)) : c;          // 16-byte vector inside its permuted 32-byte block
            if (p.x_gelu) {
                float g[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] = R.b[i].get(e);
                gelu_inplace<T>(g, VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) R.b[i].set(e, g[e]);
            }
            stvec<T>(As + (buf * BMS + row) * PITCH + cs, R.a[i]);
            stvec<T>(Bs + (buf * BMS + row) * PITCH + cs, R.b[i]);
            if (do_bias) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) colsum[e] += R.a[i].get(e);
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int cur) {
        const T* asb = As + cur * BMS * PITCH;
        const T* bsb = Bs + cur * BMS * PITCH;
#pragma unroll
        for (int kk = 0; kk < BMS / KM; ++kk) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 af[4], bf_[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) af[a] = lds_tr_frag_swz(reinterpret_cast<const bf16*>(asb), kk * 32, wn * 64 + a * 16, li, lg);
#pragma unroll
                for (int b = 0; b < 4; ++b) bf_[b] = lds_tr_frag_swz(reinterpret_cast<const bf16*>(bsb), kk * 32, wk * 64 + b * 16, li, lg);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf_[b], acc[a][b], 0, 0, 0);
            } else {
                float af[4], bf_[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) af[a] = asb[(kk * 4 + lg) * PITCH + wn * 64 + a * 16 + li];
#pragma unroll
                for (int b = 0; b < 4; ++b) bf_[b] = bsb[(kk * 4 + lg) * PITCH + wk * 64 + b * 16 + li];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf_[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    const int nsteps = (mend - mbeg + BMS - 1) / BMS;
    if (nsteps > 0) {
        gload(R0, mbeg);
        lstore(R0, 0);
    }
    if constexpr (PF == 2) {
        // step k (k >= 1) travels in register set (k - 1) & 1; at iteration s the set s & 1 delivers step s + 1 to LDS and is
        // refilled with step s + 3, so two steps are always in flight behind the one being multiplied
        if (nsteps > 1) gload(R0, mbeg + BMS);
        if (nsteps > 2) gload(R1, mbeg + 2 * BMS);
        __syncthreads();
        auto iter = [&](int s, Regs& R) {
            compute(s & 1);
            if (s + 1 < nsteps) lstore(R, (s & 1) ^ 1);
            if (s + 3 < nsteps) gload(R, mbeg + (s + 3) * BMS);
            __syncthreads();
        };
        for (int s = 0; s < nsteps; s += 2) {
            iter(s, R0);
            if (s + 1 < nsteps) iter(s + 1, R1);
        }
    } else {
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int cur = s & 1;
            if (s + 1 < nsteps) gload(R0, mbeg + (s + 1) * BMS);
            compute(cur);
            if (s + 1 < nsteps) lstore(R0, cur ^ 1);
            __syncthreads();
        }
    }

    float* pw = p.part_w + (size_t)split * p.N * p.K;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = k0 + wk * 64 + b * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + a * 16 + lg * 4 + r;
                if (n < p.N && k < p.K) pw[(size_t)n * p.K + k] = acc[a][b][r];
            }
        }

    if (do_bias) {
        // thread's column chunk is (tid % CV) for every vector it staged; 256/CV threads share it
        const int cchunk = tid % CV, rowgrp = tid / CV;   // rowgrp < 256/CV (16 for bf16, 8 for f32)
#pragma unroll
        for (int e = 0; e < VEC; ++e) bsum[rowgrp * 128 + cchunk * VEC + e] = colsum[e];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
            for (int g = 0; g < 256 / CV; ++g) t += bsum[g * 128 + tid];
            if (n0 + tid < p.N) p.part_b[(size_t)split * p.N + n0 + tid] = t;
        }
    }
}

template <typename T, int BMS, bool FEW = false, int PF = 1>
int launch_tn(const TnArgs& a, dim3 grid, hipStream_t st) {
    constexpr int VEC = Vec<T>::N;
    constexpr size_t lds = (size_t)4 * BMS * (sizeof(T) == 2 ? 128 : 128 + VEC) * sizeof(T) + (256 / (128 / VEC)) * 128 * sizeof(float);
    static FmmtLdsOnce lds_once;
    if (lds > 65536) {
        if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&linear_tn_kernel<T, BMS, FEW, PF>), (int)lds)) return rc_;
    }
    hipLaunchKernelGGL((linear_tn_kernel<T, BMS, FEW, PF>), grid, dim3(256), lds, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Few-token TN kernel (bf16; the fusion stack's and the embedding heads' weight gradients: 150-2048 tokens, N and K
// multiples of 64).  With so few tokens one (output tile, all tokens) workgroup is a CHAIN of dependent load -> MFMA steps
// (640 tokens = 20 steps of 32: 17 us for 0.75 GFLOP), and splitting the tokens over workgroups costs a second launch.
// Here the eight waves of a workgroup split the tokens of one 64 x 64 output tile among themselves -- each wave stages its
// own 32-token slabs in a wave-private LDS region (no barriers: a wave's LDS operations execute in order), 80 tokens per
// wave for 640 -- and add their accumulators in a fixed-order tree through LDS at the end.  One launch, deterministic.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void linear_tn_few_kernel(TnArgs p) {
    using T = bf16;
    constexpr int WS = 2 * 32 * 64;                        // elements of a wave's staging region: dy slab, x slab (32 tokens x 64 channels)
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 80 KB: 8 staging regions (64 KB); the reduction tree reuses it
    T* stage = reinterpret_cast<T*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int tile_n = blockIdx.x / p.tiles_k, tile_k = blockIdx.x % p.tiles_k;
    const int n0 = tile_n * 64, k0 = tile_k * 64;
    if (p.hdr && blockIdx.x == 0 && tid == 0) { p.hdr[0] = 1; p.hdr[1] = 0; p.hdr[2] = p.N; }
    const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const bool do_bias = p.part_b != nullptr && tile_k == 0;
    const int mbeg = wave * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const int nsteps = mend > mbeg ? (mend - mbeg + 31) / 32 : 0;
    T* my = stage + wave * WS;

    // a lane moves 16 bytes: token row lane / 8 (+ 8 i), 16-byte chunk lane % 8 of the 128-byte slab row
    const int lr = lane >> 3, lc = lane & 7;
    auto f = [](int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); };   // 32-byte block permutation: conflict-free transpose reads
    bf16x8 ra[4], rb[4];
    auto load = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lr + 8 * i;
            if (m < mend) {
                ra[i] = *reinterpret_cast<const bf16x8*>(dyg + (size_t)m * p.lddy + n0 + lc * 8);
                rb[i] = *reinterpret_cast<const bf16x8*>(xg + (size_t)m * p.ldx + k0 + lc * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { ra[i][e] = (bf16)0.f; rb[i][e] = (bf16)0.f; }
            }
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lr + 8 * i;
            const int o = r * 64 + ((((lc >> 1) ^ f(r)) << 1) | (lc & 1)) * 8;
            *reinterpret_cast<bf16x8*>(my + o) = ra[i];
            *reinterpret_cast<bf16x8*>(my + 32 * 64 + o) = rb[i];
        }
    };
    auto frag = [&](const T* base, int c0) {
        const int r = lg * 8 + (li >> 2);
        const T* a0 = base + r * 64 + (((c0 >> 4) ^ f(r)) << 4) + (li & 3) * 4;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 64));          // row + 4: same permutation
        union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
        u.s.lo = lo;
        u.s.hi = hi;
        return u.v;
    };

    f32x4 acc[4][4], accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    if (nsteps > 0) load(mbeg);
    for (int s = 0; s < nsteps; ++s) {
        put();                                             // after the previous step's fragment reads: same wave, in order
        if (s + 1 < nsteps) load(mbeg + (s + 1) * 32);     // in flight under this step's MFMAs
        bf16x8 af[4], bf_[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = frag(my, a * 16);
#pragma unroll
        for (int b = 0; b < 4; ++b) bf_[b] = frag(my + 32 * 64, b * 16);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf_[b], acc[a][b], 0, 0, 0);
            if (do_bias) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], ones, accb[a], 0, 0, 0);
        }
    }

    // fixed-order tree over the eight waves: (0+4, 1+5, 2+6, 3+7), then (0+2, 1+3), then 0+1.  A sender parks its 20 fragments
    // (16 of the tile + 4 bias sums: 20 KB) in LDS; the staging bytes are free once every wave has left the loop.
    float* red = reinterpret_cast<float*>(smem);           // 4 senders x 20 x 256 floats = 80 KB
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        __syncthreads();
        if (wave >= half && wave < 2 * half) {
            float* dst = red + (size_t)(wave - half) * (20 * 256) + lane * 4;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(dst + (a * 4 + b) * 256) = acc[a][b];
                *reinterpret_cast<f32x4*>(dst + (16 + a) * 256) = accb[a];
            }
        }
        __syncthreads();
        if (wave < half) {
            const float* src = red + (size_t)wave * (20 * 256) + lane * 4;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += *reinterpret_cast<const f32x4*>(src + (a * 4 + b) * 256);
                accb[a] += *reinterpret_cast<const f32x4*>(src + (16 + a) * 256);
            }
        }
    }
    if (wave != 0) return;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = k0 + b * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) p.part_w[(size_t)(n0 + a * 16 + lg * 4 + r) * p.K + k] = acc[a][b][r];
        }
    if (do_bias && li == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) p.part_b[n0 + a * 16 + lg * 4 + r] = accb[a][r];
    }
}

int launch_tn_few(int M, int N, int K, const void* dy, int lddy, const void* x, int ldx, float* dw, float* db, int* hdr, hipStream_t st) {
    constexpr int lds = 80 * 1024;
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&linear_tn_few_kernel), lds)) return rc_;
    TnArgs a{M, N, K, dy, lddy, x, ldx, dw, db, nullptr, 1, K / 64, (M + 7) / 8, N / 64, 0, 0, hdr, 1};
    hipLaunchKernelGGL(linear_tn_few_kernel, dim3((N / 64) * (K / 64)), dim3(512), lds, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

// eligibility of linear_tn_few_kernel: bf16, 129..2048 tokens, whole 64 x 64 tiles, at least 8 of them, 16-byte rows
bool tn_few_ok(int M, int N, int K, int dtype) {
    static const int on = fmmt_const("FMMT_TN_FEW", 1);
    // (at most 1024 tiles: the embedding head's 512 x 37632 gradient -- 4704 tiles of 640 tokens -- took 767 us here against 292 us
    //  with the 128 x 128-tile kernel, whose workgroups reuse each staged token slab four times as often)
    return on && dtype == FMMT_BF16 && M > 128 && M <= 2048 && N % 64 == 0 && K % 64 == 0 && (N / 64) * (K / 64) >= 8 && (N / 64) * (K / 64) <= 1024;
}

// ---------------------------------------------------------------------------------------------
// DMA-staged TN kernel (bf16 weight gradients of the many-token stage-2/3 launches): the ring of the persistent NT kernel applied
// to the token contraction.  One 8-wave workgroup per CU owns one (output tile, token split): 256 x 256 or 192 x 384 output
// channels (dy channels x x channels), waves 2 x 4, twelve 16-wide fragments per wave.
//  * a stage = 32 tokens of both operands in their natural [token][channel] layout, filled by direct global->LDS DMA; four
//    stages, three in flight; counted vmcnt + one raw s_barrier per stage; no register pass, no ds_write at all;
//  * fragments by ds_read_b64_tr_b16, issued from ONE inline-asm statement per stage (tn_read12: why).  The DMA image of a
//    stage is lane-linear, so the conflict-free layout is made on the SOURCE side: the 32-byte column blocks of a token row
//    are permuted by sigma_row (block ^ g(row) for blocks whose low three bits it can flip, 8 + ((block & 3) ^ g2(row)) for
//    blocks 8-11 of a 384-byte row), an involution the fragment read applies again.  With 256-, 512- and 768-byte rows every
//    row starts on bank 0; with 384-byte rows on bank 0 / 32 alternately, which the same g() absorbs;
//  * bias gradient without a register pass: one extra MFMA against a B fragment of ones at fixed fragment positions;
//  * partial sums leave in fragment order (1 KB per store); reduce_partials_kernel undoes the permutation;
//  * DropPath scale: the SCALED instantiation zeroes a dropped image's rows of the dy stage in LDS where the stage is consumed and applies the
//    common factor of a two-valued scale vector to the accumulators (any other vector: every stage rescaled in LDS);
//  * token counts are multiples of 64 (every Swin launch is); anything else, a recomputed activation, fewer than four tiles:
//    the register-staged kernel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_swz2(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
template <int CH>                                           // CH channels per row -> CH / 16 blocks of 32 bytes
__device__ __forceinline__ int tn_sigma(int row, int blk) {
    if constexpr (CH == 192) return blk < 8 ? (blk ^ tn_swz(row)) : 8 + ((blk & 3) ^ tn_swz2(row));
    else return blk ^ tn_swz(row);                          // 8 or 16 blocks: low three bits
}


// The twelve transposing fragment reads of one 32-token stage (24 ds_read_b64_tr_b16) and the wait for them, as ONE inline-asm
// statement.  Why not the builtin: hipcc's wait-count insertion treats a ds_read_tr intrinsic as possibly aliasing every LDS-DMA
// write in flight and puts `s_waitcnt vmcnt(0)` in front of the first fragment read of every stage -- i.e. it waits for the
// stage issued a few instructions earlier, and the ring never has more than the current stage in flight (found in the ISA of
// the first version of this kernel: 68 % of its wave cycles parked in waits).  Inline asm is opaque to that pass; the statement
// carries its own lgkmcnt(0), so its outputs are valid wherever the compiler uses them.
#define FMMT_TR2(o0, o1, ad, im) "ds_read_b64_tr_b16 %" #o0 ", %" #ad "\n\tds_read_b64_tr_b16 %" #o1 ", %" #ad " offset:%" #im "\n\t"
template <int FA, int OFFA, int OFFB>
__device__ __forceinline__ void tn_read12(s16x4 (&o)[24], const unsigned (&ad)[12]) {
    static_assert(FA == 8 || FA == 6, "fragment split");
    if constexpr (FA == 8) {
        asm volatile(FMMT_TR2(0, 1, 24, 36) FMMT_TR2(2, 3, 25, 36) FMMT_TR2(4, 5, 26, 36) FMMT_TR2(6, 7, 27, 36)
                     FMMT_TR2(8, 9, 28, 36) FMMT_TR2(10, 11, 29, 36) FMMT_TR2(12, 13, 30, 36) FMMT_TR2(14, 15, 31, 36)
                     FMMT_TR2(16, 17, 32, 37) FMMT_TR2(18, 19, 33, 37) FMMT_TR2(20, 21, 34, 37) FMMT_TR2(22, 23, 35, 37)
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                       "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]),
                       "=&v"(o[16]), "=&v"(o[17]), "=&v"(o[18]), "=&v"(o[19]), "=&v"(o[20]), "=&v"(o[21]), "=&v"(o[22]), "=&v"(o[23])
                     : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]),
                       "v"(ad[8]), "v"(ad[9]), "v"(ad[10]), "v"(ad[11]), "n"(OFFA), "n"(OFFB)
                     : "memory");
    } else {
        asm volatile(FMMT_TR2(0, 1, 24, 36) FMMT_TR2(2, 3, 25, 36) FMMT_TR2(4, 5, 26, 36) FMMT_TR2(6, 7, 27, 36)
                     FMMT_TR2(8, 9, 28, 36) FMMT_TR2(10, 11, 29, 36) FMMT_TR2(12, 13, 30, 37) FMMT_TR2(14, 15, 31, 37)
                     FMMT_TR2(16, 17, 32, 37) FMMT_TR2(18, 19, 33, 37) FMMT_TR2(20, 21, 34, 37) FMMT_TR2(22, 23, 35, 37)
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                       "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]),
                       "=&v"(o[16]), "=&v"(o[17]), "=&v"(o[18]), "=&v"(o[19]), "=&v"(o[20]), "=&v"(o[21]), "=&v"(o[22]), "=&v"(o[23])
                     : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]),
                       "v"(ad[8]), "v"(ad[9]), "v"(ad[10]), "v"(ad[11]), "n"(OFFA), "n"(OFFB)
                     : "memory");
    }
}

// TNn x TKk output tile, 32 tokens per stage, NBUF stages, waves WN x (8 / WN), 12 fragments per wave.
// SCALED: DropPath row scale s[token / rows_per_scale] on dy (rows_per_scale >= 32: a 32-token stage touches at most two images).  The DMA image of a
// dy stage is rescaled IN LDS (each element once per workgroup; on the fragments it would be once per wave that reads it) where the stage is consumed,
// between two barriers of its own -- 512 threads x 16 (12) elements, fp32 multiply, RNE back to bf16.  DropPath's vectors are two-valued (0 / 1 / keep):
// then only the stages holding a dropped image's tokens are touched (rows zeroed) and the common factor multiplies the accumulators once, exactly.
// The slice of the scale vector a token split needs sits in LDS behind the ring (host side: at most 1024 samples per split, else the register-staged kernel).
template <int TNn, int TKk, int NBUF, int WN, bool SCALED = false>
__global__ __launch_bounds__(512) void linear_tn_dma_kernel(TnArgs p) {
    using T = bf16;
    constexpr int BT = 32, WK = 8 / WN;
    constexpr int CPRA = TNn / 8, CPRB = TKk / 8;           // 16-byte chunks per token row
    constexpr int A_EL = BT * TNn, B_EL = BT * TKk, STAGE = A_EL + B_EL;
    constexpr int NIA = BT * CPRA / 64, NIB = BT * CPRB / 64, NI = NIA + NIB;   // DMA instructions per stage
    static_assert((BT * CPRA) % 64 == 0 && (BT * CPRB) % 64 == 0, "whole DMA instructions per operand");
    constexpr int CNT = NI / 8, REM = NI % 8;               // waves below REM issue one more
    constexpr int FA = TNn / (16 * WN), FB = TKk / (16 * WK);   // 16-wide fragments per wave
    static_assert(TNn % (16 * WN) == 0 && TKk % (16 * WK) == 0 && FA + FB == 12 && NBUF >= 2 && NBUF <= 5, "tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* S = reinterpret_cast<T*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int li = lane & 15, lg = lane >> 4;
    const int tiles = p.tiles_n * p.tiles_k;
    const int logical = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int split = logical / tiles, tile = logical - split * tiles;
    const int tile_n = tile / p.tiles_k, tile_k = tile % p.tiles_k;
    const int n0 = tile_n * TNn, k0 = tile_k * TKk;
    const int mbeg = split * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const int NP = p.npad ? p.npad : p.N;                    // rows of a split's partials (> N: the last tile's upper 192 channels do not exist)
    if (p.hdr && blockIdx.x == 0 && tid == 0) { p.hdr[0] = p.splits; p.hdr[1] = TNn; p.hdr[2] = NP; }     // TNn != 0: fragment-order partials
    const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const bool extra = wave < REM;

    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    // source address of a DMA instruction = a workgroup-uniform 64-bit base that moves with the stage (scalar registers) + this lane's loop-invariant
    // 32-bit byte offset (one VGPR per instruction; as 64-bit per-lane pointers the five of them were the registers the scaled kernel ran out of)
    constexpr int NISS = CNT + (REM != 0 ? 1 : 0);
    unsigned goff[NISS];
#pragma unroll
    for (int i = 0; i < NISS; ++i) {
        const int j = i * 8 + wave;
        if (j < NIA) {
            const int q = j * 64 + lane, row = q / CPRA, pos = q - row * CPRA;          // LDS slot (row, pos) of this lane
            const int gpos = (tn_sigma<TNn>(row, pos >> 1) << 1) | (pos & 1);           // the global chunk that belongs there
            int col = n0 + gpos * 8;
            if (col >= p.N) col -= NP - p.N;                                            // padded tile: a valid column instead (its products are never read)
            goff[i] = (unsigned)(row * p.lddy + col) * 2u;
        } else {
            const int q = (j - NIA) * 64 + lane, row = q / CPRB, pos = q - row * CPRB;
            const int gpos = (tn_sigma<TKk>(row, pos >> 1) << 1) | (pos & 1);
            goff[i] = (unsigned)(row * p.ldx + k0 + gpos * 8) * 2u;
        }
    }
    auto issue_one = [&](T* base, int i, int mb) {
        const int j = i * 8 + wave;                                                     // wave-uniform instruction index
        const char* sb = j < NIA ? reinterpret_cast<const char*>(dyg) + (size_t)mb * p.lddy * 2 : reinterpret_cast<const char*>(xg) + (size_t)mb * p.ldx * 2;
        T* dst = j < NIA ? base + j * 512 : base + A_EL + (j - NIA) * 512;
        __builtin_amdgcn_global_load_lds((gptr_t*)(sb + goff[i]), (lptr_t*)dst, 16, 0, 0);
    };
    auto issue = [&](int slot, int mb) {
        T* base = S + slot * STAGE;
#pragma unroll
        for (int i = 0; i < CNT; ++i) issue_one(base, i, mb);
        if constexpr (REM != 0)
            if (extra) issue_one(base, CNT, mb);
    };
    // the stage about to be consumed has landed when at most (NBUF - 2) later stages of this wave are still in flight
    auto wait_landed = [&](bool steady) {
        if (!steady) { wait_vm<0>(); return; }
        if constexpr (REM != 0) {
            if (extra) wait_vm<(NBUF - 2) * (CNT + 1)>();
            else wait_vm<(NBUF - 2) * CNT>();
        } else {
            wait_vm<(NBUF - 2) * CNT>();
        }
    };

    // byte offsets of this lane's twelve fragments inside a stage: token row lg*8 + (li>>2) (+4 through the instruction offset),
    // 32-byte column block permuted by the row's key (tn_sigma), 8 bytes at (li & 3) inside it
    //
    // bias gradient: 16-channel block a of this wave row's dy channels is summed by exactly one (k-tile, k-wave) pair, the one
    // with a % (tiles_k * WK) == wk * tiles_k + tile_k: at most two blocks per wave (FA <= 2 WK).  The wave ROTATES its fragment
    // order so that its first block sits at position 0 (and the second, which exists only for tiles_k == 1, at position WK):
    // the extra MFMA against a ones fragment then hangs off fixed positions behind loop-invariant flags -- selecting the
    // position at run time made the compiler copy accumulators around every MFMA row.
    // (384 x 192 tiles, waves 4 x 2, round 6: FA = 6 = 3 WK -- a third block at position 2 WK)
    static_assert(FA > WK && FA <= 3 * WK, "positions 0, WK and 2 WK");
    const int o0 = wk * p.tiles_k + tile_k;
    const bool has0 = p.part_b != nullptr && o0 < FA;
    const bool has1 = has0 && p.tiles_k == 1 && o0 + WK < FA;
    const bool has2 = has1 && FA > 2 * WK && o0 + 2 * WK < FA;
    const int rot = has0 ? o0 : 0;
    auto block_of = [&](int j) { const int a = j + rot; return a >= FA ? a - FA : a; };
    unsigned foff[12];
    {
        const int r0 = lg * 8 + (li >> 2);
#pragma unroll
        for (int a = 0; a < FA; ++a) {
            const int c0 = wn * (TNn / WN) + block_of(a) * 16;
            foff[a] = (unsigned)(r0 * TNn + (tn_sigma<TNn>(r0, c0 >> 4) << 4) + (li & 3) * 4) * 2u;
        }
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int c0 = wk * (TKk / WK) + b * 16;
            foff[FA + b] = (unsigned)(A_EL + r0 * TKk + (tn_sigma<TKk>(r0, c0 >> 4) << 4) + (li & 3) * 4) * 2u;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)S;

    f32x4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb0 = f32x4{0.f, 0.f, 0.f, 0.f}, accb1 = f32x4{0.f, 0.f, 0.f, 0.f}, accb2 = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    auto compute = [&](int slot) {
        const unsigned sbase = lds0 + (unsigned)slot * (unsigned)(STAGE * 2);
        unsigned ad[12];
#pragma unroll
        for (int f = 0; f < 12; ++f) ad[f] = foff[f] + sbase;
        s16x4 o[24];
        tn_read12<FA, 8 * TNn, 8 * TKk>(o, ad);
        bf16x8 fr[12];
#pragma unroll
        for (int f = 0; f < 12; ++f) {
            union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
            u.s.lo = o[2 * f];
            u.s.hi = o[2 * f + 1];
            fr[f] = u.v;
        }
#pragma unroll
        for (int a = 0; a < FA; ++a) {
#pragma unroll
            for (int b = 0; b < FB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[a], fr[FA + b], acc[a][b], 0, 0, 0);
            if (a == 0 && has0) accb0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0], ones, accb0, 0, 0, 0);
            if (a == WK && has1) accb1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[WK], ones, accb1, 0, 0, 0);
            if constexpr (FA > 2 * WK) {
                if (a == 2 * WK && has2) accb2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[2 * WK], ones, accb2, 0, 0, 0);
            }
        }
    };

    const int nsteps = (mend - mbeg) / BT;                   // whole steps only (M % 64 == 0, chunk % 64 == 0)
    // SCALED: this thread's (at most two) entries of the split's scale slice are requested BEFORE the first stages, so that they return first and the
    // DMA prologue stays in flight while the slice is examined (requested behind it, their wait was a wait for the whole prologue: +4-8 us per workgroup)
    // (inline asm: the compiler's own wait for a global load it tracks is vmcnt(0) once DMA instructions follow; here the wait is counted, below)
    float rsv0 = 0.f, rsv1 = 0.f;
    if constexpr (SCALED) {
        const int samp0_ = mbeg / p.rows_per_scale, nsamp_ = (mend - 1) / p.rows_per_scale - samp0_ + 1;
        const float* a0 = p.rowscale + samp0_ + min(tid, nsamp_ - 1);
        const float* a1 = p.rowscale + samp0_ + min(tid + 512, nsamp_ - 1);
        asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(rsv0), "=&v"(rsv1) : "v"(a0), "v"(a1) : "memory");
    }
    int islot = 0, im = mbeg;
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < nsteps) {
            issue(islot, im);
            islot = islot + 1 == NBUF ? 0 : islot + 1;
            im += BT;
        }
    int cslot = 0;
    float post = 1.0f;                                       // SCALED, two-valued scale vector: the common non-zero value, applied to the accumulators
    if constexpr (!SCALED) {
        for (int s = 0; s < nsteps; ++s) {
            wait_landed(s + NBUF - 2 < nsteps);              // steady: NBUF - 2 later stages have been issued
            __builtin_amdgcn_s_barrier();
            if (s + NBUF - 1 < nsteps) {
                issue(islot, im);
                islot = islot + 1 == NBUF ? 0 : islot + 1;
                im += BT;
            }
            compute(cslot);
            cslot = cslot + 1 == NBUF ? 0 : cslot + 1;
        }
    } else {
        static_assert(NBUF >= 3, "scaled ring");
        constexpr int EPT = TNn / 16;                        // elements per thread and stage: 32 rows x 16 threads per row
        static_assert(EPT == 16 || EPT == 12, "scale pass");
        const int rps = p.rows_per_scale;
        float* sc = reinterpret_cast<float*>(smem + (size_t)NBUF * STAGE * sizeof(T));
        const int samp0 = mbeg / rps, nsamp = (mend - 1) / rps - samp0 + 1;
        // DropPath hands over a TWO-VALUED vector (0 for a dropped image, 1 / keep for the others).  Then the non-zero value is a factor of the whole
        // sum: it goes onto the accumulators once, and a stage is touched in LDS only if one of its (at most two) images is dropped -- zeroing its
        // rows; the other ~90 % of the stages skip the pass.  Any other vector takes the general pass (every stage rescaled).
        // the two slice loads were issued in front of the prologue's stages: wait for them by count (everything issued since stays in flight)
        if (nsteps >= NBUF - 1) {
            if constexpr (REM != 0) {
                if (extra) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rsv0), "+v"(rsv1) : "n"((NBUF - 1) * (CNT + 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rsv0), "+v"(rsv1) : "n"((NBUF - 1) * CNT) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rsv0), "+v"(rsv1) : "n"((NBUF - 1) * CNT) : "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(rsv0), "+v"(rsv1) : : "memory");
        }
        // extrema over the NON-ZERO entries only (lanes without an entry, and zeros, contribute neither): "two-valued" = zeros and ONE other value of
        // either sign (round-5 ADVICE: with the maximum taken over all entries and 0 for idle lanes, a slice of zeros and negatives looked all-zero)
        float vmx = -3.0e38f, vmn = 3.0e38f;
        if (tid < nsamp) {
            sc[tid] = rsv0;
            if (rsv0 != 0.f) vmx = vmn = rsv0;
        }
        if (tid + 512 < nsamp) {
            sc[tid + 512] = rsv1;
            if (rsv1 != 0.f) {
                vmx = fmaxf(vmx, rsv1);
                vmn = fminf(vmn, rsv1);
            }
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            vmx = fmaxf(vmx, __shfl_xor(vmx, o, 64));
            vmn = fminf(vmn, __shfl_xor(vmn, o, 64));
        }
        float* red = sc + 1024;
        if ((tid & 63) == 0) {
            red[tid >> 6] = vmx;
            red[8 + (tid >> 6)] = vmn;
        }
        const int srow = tid >> 4;                          // (rows_per_scale >= 32, host-checked: a 32-token stage touches at most two images)
        const unsigned sc0 = (unsigned)(uintptr_t)(lptr_t*)sc;
        const unsigned soff = lds0 + (unsigned)((srow * TNn + (tid & 15) * EPT) * 2);
        auto scale2 = [](unsigned w, float sv) -> unsigned {  // two bf16 in a dword -> scaled, rounded to nearest even
            const float lo = __uint_as_float(w << 16) * sv, hi = __uint_as_float(w & 0xffff0000u) * sv;
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            union { bf16x2_t v; unsigned u; } r;
            r.v = bf16x2_t{(__bf16)lo, (__bf16)hi};
            return r.u;
        };
        bool binary = false;                                 // set behind the prologue barrier below
        auto scale_stage = [&](int slot, int step) {        // (runs on the few touched stages: the division is off the common path)
            const int sidx = min((mbeg + step * BT + srow) / rps - samp0, nsamp - 1);
            const unsigned a = soff + (unsigned)slot * (unsigned)(STAGE * 2), sa = sc0 + (unsigned)sidx * 4u;
            u32x4 v0;
            float sv;
            if constexpr (EPT == 16) {
                u32x4 v1;
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b32 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(sv) : "v"(a), "v"(sa) : "memory");
                if (binary) sv = sv != 0.f ? 1.0f : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = scale2(v0[e], sv); v1[e] = scale2(v1[e], sv); }
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16" ::"v"(a), "v"(v0), "v"(v1) : "memory");
            } else {
                u32x2 v1;
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b64 %1, %3 offset:16\n\tds_read_b32 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(sv) : "v"(a), "v"(sa) : "memory");
                if (binary) sv = sv != 0.f ? 1.0f : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) v0[e] = scale2(v0[e], sv);
                v1[0] = scale2(v1[0], sv);
                v1[1] = scale2(v1[1], sv);
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b64 %0, %2 offset:16" ::"v"(a), "v"(v0), "v"(v1) : "memory");
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slice and the per-wave extrema are in LDS (the first stages may still be in flight)
        __builtin_amdgcn_s_barrier();
        {
            float mx = red[0], mn = red[8];
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                mx = fmaxf(mx, red[w]);
                mn = fminf(mn, red[8 + w]);
            }
            // no non-zero entry at all (mx still at its start value): the sum is zero; one non-zero value: mn == mx.  Both workgroup-uniform: scalar registers
            const bool none = mx == -3.0e38f;
            binary = __builtin_amdgcn_readfirstlane((rps >= BT && (none || mn == mx)) ? 1 : 0) != 0;
            if (binary) post = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, none ? 0.f : mx)));
        }
        // ONE schedule -- the unscaled kernel's (NBUF - 1 stages in flight).  A stage is rescaled in LDS where it is consumed, between two barriers of
        // its own: with a two-valued vector only the stages that hold a dropped image's tokens (one in ten at DropPath's largest rate; the rows are
        // zeroed, the common factor goes onto the accumulators), with any other vector every stage (the pass is then exposed: that form is for
        // callers outside the model -- round 4 scaled one stage ahead inside a shallower ring and paid +30-40 % on every launch).
        // "does the stage hold a token of a dropped image?" is looked up one step ahead (the LDS read sits behind the previous step's MFMAs)
        // which stages hold a token of a dropped image: one bit per stage in four 64-bit scalar masks, built once (lane l looks at stages l, l + 64, ...);
        // the loop tests a bit -- no LDS access, no wait, no per-step image arithmetic.  More than 256 stages per split: every stage is touched.
        unsigned long long tm[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        if (binary && nsteps <= 256) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int st = k * 64 + lane, m0 = mbeg + st * BT;
                const int i0 = min(m0 / rps - samp0, nsamp - 1), i1 = min((m0 + BT - 1) / rps - samp0, nsamp - 1);
                tm[k] = __ballot(st < nsteps && (sc[max(i0, 0)] == 0.f || sc[max(i1, 0)] == 0.f));
            }
        }
        for (int s = 0; s < nsteps; ++s) {
            wait_landed(s + NBUF - 2 < nsteps);
            __builtin_amdgcn_s_barrier();
            if (s + NBUF - 1 < nsteps) {
                issue(islot, im);
                islot = islot + 1 == NBUF ? 0 : islot + 1;
                im += BT;
            }
            const unsigned long long word = s < 64 ? tm[0] : s < 128 ? tm[1] : s < 192 ? tm[2] : tm[3];
            if ((s >= 256) || ((word >> (s & 63)) & 1ull)) {
                scale_stage(cslot, s);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            compute(cslot);
            cslot = cslot + 1 == NBUF ? 0 : cslot + 1;
        }
    }

    // partial sums in FRAGMENT order: [split][tile][wave][block a][b][lane] x 4 floats -- one 1 KB store per accumulator tile
    // (row-major order would be 4-byte stores to four rows per instruction); the finish pass undoes the permutation while it
    // sums (reduce_partials_kernel, layout id in the workspace header)
    float* pw = p.part_w + (size_t)split * NP * p.K + (size_t)tile * (TNn * TKk) + (size_t)wave * (FA * FB * 256) + lane * 4;
    if constexpr (SCALED) {
#pragma unroll
        for (int a = 0; a < FA; ++a)
#pragma unroll
            for (int b = 0; b < FB; ++b) acc[a][b] *= post;
        accb0 *= post;
        accb1 *= post;
        accb2 *= post;
    }
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) *reinterpret_cast<f32x4*>(pw + (block_of(a) * FB + b) * 256) = acc[a][b];
    if (li == 0 && has0) {                                   // every column of the ones-product holds the row sum
        float* pb = p.part_b + (size_t)split * NP + n0 + wn * (TNn / WN) + lg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[block_of(0) * 16 + r] = accb0[r];
        if (has1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pb[block_of(WK) * 16 + r] = accb1[r];
        }
        if constexpr (FA > 2 * WK) {
            if (has2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[block_of(2 * WK) * 16 + r] = accb2[r];
            }
        }
    }
}

template <int TNn, int TKk, int NBUF, int WN, bool SCALED = false>
int launch_tn_dma(const TnArgs& a, int grid, hipStream_t st) {
    constexpr size_t lds = (size_t)NBUF * 32 * (TNn + TKk) * 2 + (SCALED ? 4096 + 64 : 0);
    static_assert(lds <= 160 * 1024, "LDS");
    static FmmtLdsOnce lds_once;
    if (int rc_ = lds_once.set(reinterpret_cast<const void*>(&linear_tn_dma_kernel<TNn, TKk, NBUF, WN, SCALED>), (int)lds)) return rc_;
    hipLaunchKernelGGL((linear_tn_dma_kernel<TNn, TKk, NBUF, WN, SCALED>), dim3(grid), dim3(512), lds, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

// out[i] = sum_s part[s][i] : 256 threads = 64 float4 outputs x 4 split groups, fixed-order tree (deterministic).
// One launch finishes both the weight gradient (blocks [0, wblocks)) and, if present, the bias gradient.
// hdr[0] = number of splits, hdr[1] = layout of the weight partials as written by the contraction kernel: 0 = row-major
// [N][K]; 256 / 192 / 384 = fragment order of linear_tn_dma_kernel<256,256> / <192,384> / <384,192> (undone here, K = row length of dW).
__device__ __forceinline__ size_t tn_frag_dest(int layout, size_t q, int K, int r) {
    // q = float4 index inside one split's partials: ((((tile * 8 + wave) * FA + a) * FB + b) * 64 + lane)
    const int TNn = layout, TKk = layout == 256 ? 256 : layout == 192 ? 384 : 192, FA = layout == 256 ? 8 : 6, FB = layout == 256 ? 4 : 6;
    const int WK = layout == 384 ? 2 : 4, WN = 8 / WK;          // 384: linear_tn_dma_kernel<384,192>, waves 4 x 2
    const int lane = (int)(q & 63);
    size_t t = q >> 6;
    const int b = (int)(t % FB); t /= FB;
    const int a = (int)(t % FA); t /= FA;
    const int wave = (int)(t & 7);
    const int tile = (int)(t >> 3);
    const int tiles_k = K / TKk, tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;      // (row n may be >= N for a padded last tile: the caller drops it)
    const int wn = wave / WK, wk = wave % WK;
    const int n = tile_n * TNn + wn * (TNn / WN) + a * 16 + (lane >> 4) * 4 + r;
    const int k = tile_k * TKk + wk * (TKk / WK) + b * 16 + (lane & 15);
    return (size_t)n * K + k;
}
// 16 split groups x 64 float4 columns per workgroup (4 groups until round 3: a 96 x 96 weight gradient with 768 token splits is 36
// workgroups, each thread a chain of 192 loads -- 58 us for 28 MB)
constexpr int RP_GROUPS = 16;
// n / n2: the true sizes N K and N of dW / db; a split's partials hold hdr[2] >= N rows (0: N) -- the stride, and for fragment-order partials the range walked.
__global__ __launch_bounds__(64 * RP_GROUPS) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n, const int* __restrict__ hdr,
                                                              int wblocks, const float* __restrict__ part2, float* __restrict__ out2, size_t n2, int K) {
    __shared__ f32x4 red[RP_GROUPS][64];
    const int splits = hdr[0];                               // written by the contraction kernel that filled the partials
    int layout = hdr[1];
    const size_t rows = hdr[2] > 0 ? (size_t)hdr[2] : n2;
    const size_t n_true = n, n2_true = n2;
    n = rows * (size_t)K;                                    // per-split stride / walked range of the weight partials
    n2 = rows;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int blk = blockIdx.x;
    size_t limit = n_true;                                   // row-major partials: elements past the true size are padding
    if (blk >= wblocks) {                                  // bias gradient blocks
        blk -= wblocks;
        part = part2;
        out = out2;
        n = n2;
        limit = n2_true;
        layout = 0;
    }
    const size_t q = (size_t)blk * 64 + tx;                  // float4 index; n % 4 == 0
    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q * 4 < n)
        for (int s = ty; s < splits; s += RP_GROUPS) t += *reinterpret_cast<const f32x4*>(part + (size_t)s * n + q * 4);
    red[ty][tx] = t;
    __syncthreads();
    if (ty == 0 && q * 4 < n) {
        f32x4 v = red[0][tx];
#pragma unroll
        for (int g = 1; g < RP_GROUPS; ++g) v += red[g][tx];
        if (layout == 0) {
            if (q * 4 < limit) *reinterpret_cast<f32x4*>(out + q * 4) = v;      // (limit % 4 == 0)
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t d = tn_frag_dest(layout, q, K, r);
                if (d < n_true) out[d] = v[r];                                   // rows >= N of a padded last tile: dropped
            }
        }
    }
}

struct TnPlan { int tiles_n, tiles_k, splits, chunk; size_t bytes; int tn, tk; int npad; };     // tn != 0: linear_tn_dma_kernel<tn>; npad: rows of a split's partials (0 = N)

// (An XCD-local decomposition -- tile shapes 192 x 96 / 96 x 192 / 128 x 128 chosen so that all tiles of a token split fill
//  whole XCDs, every dy / x slab fetched into exactly one L2 -- was written, tested and measured: the stage-2/3 launches of the
//  step took 7.16 ms against 6.57 ms with this split-major order (same call).  The second fetch of a slab is served by the
//  256 MB Infinity Cache; what the XCD-local form paid was 6-25 % idle workgroup slots and 244-252 registers.  Not kept.)
TnPlan tn_plan(int M, int N, int K, int dtype) {
    TnPlan pl;
    pl.tn = pl.tk = 0;
    pl.tiles_n = (N + 127) / 128;
    pl.tiles_k = (K + 127) / 128;
    const int tiles = pl.tiles_n * pl.tiles_k;
    // one round of co-resident workgroups: 2 per CU for the 64-token-step kernels (64-78 KB LDS), 3 per CU for
    // the 32-token-step kernel; more would run as a second, partly empty round
    // (leaving head-room for the concurrent text-encoder stream -- 448/672, 384/576 -- measured: no gain)
    const int target = (M <= 262144) ? 512 : 768;
    int splits = target / tiles;
    // few-token problems (fusion stack: 152-664 rows): one pass over the tokens per output tile, written straight to
    // dW / db by fmmt_linear_wgrad -- a second launch to add two or three partials costs as much as the contraction
    if (M <= 768) splits = 1;
    const int max_by_rows = (M + 255) / 256;
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits < 1) splits = 1;
    // keep the partial buffer under 512 MiB
    while (splits > 1 && (size_t)splits * N * K * 4 > ((size_t)512 << 20)) --splits;
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + 63) / 64 * 64;
    pl.splits = (M + chunk - 1) / chunk;
    pl.chunk = chunk;
    pl.bytes = (size_t)pl.splits * ((size_t)N * K + N) * sizeof(float);
    return pl;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int fmmt_linear_fwd(int dtype, int M, int N, int K,
                               const void* x, int ldx, const void* w, int ldw, const float* bias,
                               void* y, int ldy, void* y_pre,
                               int epi, const void* aux, int ldaux,
                               const void* res, int ldres, const float* rowscale, int rows_per_scale,
                               void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return FMMT_EINVAL;
    const int vec = dtype == FMMT_BF16 ? 8 : 4;
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    if (K % vec || N % 4 || ldx % vec || ldw % vec || ldy % 4) return FMMT_EINVAL;
    if (epi < 0 || epi > FMMT_EPI_MUL_AUX) return FMMT_EINVAL;
    if ((epi == FMMT_EPI_GELU_BWD || epi == FMMT_EPI_MUL_AUX) && (!aux || ldaux % 4)) return FMMT_EINVAL;
    if (res && ldres % 4) return FMMT_EINVAL;
    if (rowscale && rows_per_scale <= 0) return FMMT_EINVAL;
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (bias && !aligned16(bias)) ||
        (res && !aligned16(res)) || (aux && !aligned16(aux)) || (y_pre && !aligned16(y_pre)))
        return FMMT_EALIGN;
    LinArgs a{M, N, K, x, ldx, w, ldw, bias, y, ldy, y_pre, epi, aux, ldaux, res, ldres, rowscale, rows_per_scale, 0, 0, 1, 0, nullptr};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == FMMT_BF16 ? dispatch_nt<bf16>(a, st) : dispatch_nt<float>(a, st);
}

// Few-token Linear over THREE weights in one launch (MELDTransEncoder's query / key / value, modules/Transformer.py:64-103: three nn.Linear over
// the same hidden states, and their input gradient dx = dq Wq + dk Wk + dv Wv).
//   seg_mode 1 (along N): y[:, s * N/3 : (s + 1) * N/3] = x @ w_s^T + bias_s        w_s [N/3][K]
//   seg_mode 2 (along K): y = sum_s x[:, s * K/3 : (s + 1) * K/3] @ w_s^T            w_s [N][K/3]   (no bias)
// bf16, the direct-to-LDS quarter-tile kernels only (the shapes dispatch_nt sends there: tokens <= 4096, few tiles); anything else: FMMT_EINVAL and
// the caller issues three fmmt_linear_fwd calls.
extern "C" int fmmt_linear_fwd_seg3(int dtype, int M, int N, int K, const void* x, int ldx, const void* w0, const void* w1, const void* w2, int ldw,
                                    int seg_mode, const float* bias0, const float* bias1, const float* bias2, void* y, int ldy, void* stream) {
    if (dtype != FMMT_BF16 || M <= 0 || N <= 0 || K <= 0 || (seg_mode != 1 && seg_mode != 2)) return FMMT_EINVAL;
    if (!x || !w0 || !w1 || !w2 || !y) return FMMT_EINVAL;
    const int seg = seg_mode == 1 ? N / 3 : K / 3;
    if ((seg_mode == 1 ? N : K) % 3 || seg % 64 || N % 64 || K % 64 || K < 128 || ldx % 8 || ldw % 8 || ldy % 4 || M > 4096) return FMMT_EINVAL;
    if (seg_mode == 2 && (bias0 || bias1 || bias2)) return FMMT_EINVAL;
    if (!aligned16(x) || !aligned16(w0) || !aligned16(w1) || !aligned16(w2) || !aligned16(y)) return FMMT_EALIGN;
    if (((M + 63) / 64) * ((N + 127) / 128) >= 256) return FMMT_EINVAL;                 // dispatch_nt's few-tile rule: larger problems do not come here
    LinArgs a{M, N, K, x, ldx, w0, ldw, bias0, y, ldy, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 1, 0, 0, 1, 0, nullptr};
    a.w1 = w1;
    a.w2 = w2;
    a.bias1 = bias1;
    a.bias2 = bias2;
    a.wseg = seg;
    a.wseg_mode = seg_mode;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (K >= 2048) return launch_nt<bf16, 32, 64, 64, 4, true, true>(a, st);
    return launch_nt<bf16, 32, 64, 64, 2, true, true>(a, st);
}

extern "C" size_t fmmt_linear_splitk_workspace(int M, int N, int K) {
    int ks;
    const int splits = splitk_plan(M, N, K, &ks);
    return ks ? (size_t)splits * M * N * sizeof(float) : 0;
}

extern "C" int fmmt_linear_fwd_splitk(int dtype, int M, int N, int K, const void* x, int ldx, const void* w, int ldw,
                                      const float* bias, void* y, int ldy, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return FMMT_EINVAL;
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    const int vec = dtype == FMMT_BF16 ? 8 : 4;
    if (K % vec || N % 4 || ldx % vec || ldw % vec) return FMMT_EINVAL;
    if (!aligned16(x) || !aligned16(w) || !aligned16(workspace)) return FMMT_EALIGN;
    int ks;
    const int splits = splitk_plan(M, N, K, &ks);
    if (!ks) return FMMT_EINVAL;                            // not a split-K shape: use fmmt_linear_fwd
    if (workspace_bytes < (size_t)splits * M * N * sizeof(float)) return FMMT_EWORKSPACE;
    LinArgs a{M, N, K, x, ldx, w, ldw, nullptr, nullptr, N, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 1, 0, 0, 1, ks,
              reinterpret_cast<float*>(workspace)};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (int rc = (dtype == FMMT_BF16 ? dispatch_nt<bf16>(a, st) : dispatch_nt<float>(a, st))) return rc;
    const size_t total = (size_t)M * N;
    if (dtype == FMMT_BF16)
        hipLaunchKernelGGL(splitk_finish_kernel<bf16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.part, splits, M, N, bias, (bf16*)y, ldy);
    else
        hipLaunchKernelGGL(splitk_finish_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.part, splits, M, N, bias, (float*)y, ldy);
    FMMT_CHECK_LAUNCH();
    return 0;
}

namespace {
// DMA-staged plan (linear_tn_dma_kernel): tile TNn x 128 with TNn = 256 / 192, one workgroup per CU, splits = 256 / tiles
TnPlan tn_plan_dma(int M, int N, int K) {
    TnPlan pl{0, 0, 0, 0, 0, 0, 0, 0};
    // Default on (FMMT_TN_DMA=0: everything register-staged).  History: the first version of this kernel (256 / 192 x 128 tiles,
    // the builtin ds_read_tr) measured 3-9 % SLOWER than the register-staged kernel with 68 % of its wave cycles parked in
    // waits -- hipcc had put s_waitcnt vmcnt(0) in front of every stage's first fragment read, so the ring never had a second
    // stage in flight.  With the reads in inline asm, 128 FLOP per staged byte and four 32-token stages: 860-1105 TF/s against
    // 550-630.
    static const int mode = fmmt_const("FMMT_TN_DMA", 1);
    // FMMT_TN_DMA_MINM: fewest tokens for this kernel, exclusive.  8192 since the 320-frame legs (configs[4]: 15680 stage-3 tokens)
    // measured 56.2 -> 55.7 ms per step with it, three alternating pairs in one call; 16384 before.
    static const int minm = fmmt_const("FMMT_TN_DMA_MINM", 8192);
    if (!mode || M <= minm || M % 64) return pl;
    int tn = 0, tk = 0;
    if (N % 256 == 0 && K % 256 == 0) tn = 256, tk = 256;
    else if (N % 192 == 0 && K % 384 == 0) tn = 192, tk = 384;
    // round 6: the transposed tile for K = 192 (Swin stage 1's fc1 weight gradient, 768 x 192: it fell to the 128 x 128 register-staged kernel at 3.3 TB/s
    // where its mirror image 192 x 768 runs at 4.7 on <192,384>); one k-tile only (the bias blocks' positions assume it), unscaled launches only
    // N = 576 (stage 1's qkv weight gradient) is 192 short of two tiles: the last tile's upper half re-reads valid columns and its products go to partial rows
    // the finish pass drops (npad)
    else if ((N % 384 == 0 || (N % 384 == 192 && N >= 576)) && K == 192) tn = 384, tk = 192;
    // (two or three tiles used to be refused -- > 64 splits, "the partial sums outweigh the operands" --; re-measured in round 4, same call:
    //  501760 x 192 x 768 with the DropPath scale 324 -> 233 us, 125440 x 384 x 384 70 -> 64 us with the finish pass, no shape slower.  One tile stays out.)
    const int npad = tn ? (N + tn - 1) / tn * tn : 0;
    if (tn && (npad / tn) * (K / tk) < 2) tn = 0;
    if (!tn) return pl;
    const int tiles = (npad / tn) * (K / tk);
    if (tiles > 128) return pl;
    int splits = 256 / tiles;
    while (splits > 1 && (size_t)splits * npad * K * 4 > ((size_t)512 << 20)) --splits;
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + 63) / 64 * 64;
    splits = (M + chunk - 1) / chunk;
    if (chunk < 512) return pl;
    pl = TnPlan{npad / tn, K / tk, splits, chunk, 0, tn, tk, npad == N ? 0 : npad};
    return pl;
}

// Workspace: [256-byte header | bias partials smax x N | weight partials smax x N x K], smax = the larger split count of the
// plans a (dtype, M, N, K) launch may take.  Which plan a launch takes also depends on its operands (DropPath scale ...), so
// the contraction kernel records the split count it wrote in the header and the finish pass reads it from there.
constexpr size_t TN_HDR = 256;
int tn_smax(int M, int N, int K, int dtype) {
    int smax = tn_plan(M, N, K, dtype).splits;
    if (dtype == FMMT_BF16) {
        const TnPlan pd = tn_plan_dma(M, N, K);
        if (pd.tn && pd.splits > smax) smax = pd.splits;
    }
    return smax;
}
// rows of a split's partials the workspace is carved for: N, or the padded count of a DMA plan that may take the launch
int tn_rows(int M, int N, int K, int dtype) {
    if (dtype == FMMT_BF16) {
        const TnPlan pd = tn_plan_dma(M, N, K);
        if (pd.tn && pd.npad > N) return pd.npad;
    }
    return N;
}
size_t tn_ws_bytes(int M, int N, int K, int dtype) {
    const size_t rows = (size_t)tn_rows(M, N, K, dtype);
    return TN_HDR + (size_t)tn_smax(M, N, K, dtype) * (rows * K + rows) * sizeof(float);
}

}  // namespace

extern "C" size_t fmmt_linear_wgrad_workspace(int dtype, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || (dtype != FMMT_BF16 && dtype != FMMT_F32)) return 0;
    return tn_ws_bytes(M, N, K, dtype);
}

namespace {
// the split contraction: part_w [splits][N][K], part_b [splits][N] or nullptr (splits == 1: these may be dw / db themselves)
int launch_tn_plan(int dtype, int M, int N, int K, const void* dy, int lddy, const void* x, int ldx, float* part_w, float* part_b,
                   const float* rowscale, int rows_per_scale, int x_epi, int* hdr, hipStream_t st) {
    static const int tn_xcd = fmmt_const("FMMT_TN_XCD", 1);
    if (x_epi != 0 && x_epi != FMMT_EPI_GELU) return FMMT_EINVAL;
    if (tn_few_ok(M, N, K, dtype) && !rowscale && !x_epi && lddy % 8 == 0 && ldx % 8 == 0)
        return launch_tn_few(M, N, K, dy, lddy, x, ldx, part_w, part_b, hdr, st);     // one split: part_w / part_b may be dw / db themselves
    if (dtype == FMMT_BF16 && hdr && !x_epi && lddy % 8 == 0 && ldx % 8 == 0) {
        const TnPlan pd = tn_plan_dma(M, N, K);
        // scaled launches: the split's slice of the scale vector has to fit the 4 KB behind the ring
        static const int dma_scaled = fmmt_const("FMMT_TN_DMA_SCALED", 1);
        const bool scaled_ok = !rowscale || (dma_scaled && rows_per_scale >= 32 && pd.tn && pd.tk != 192 && pd.chunk / rows_per_scale + 2 <= 1024);
        if (pd.tn && scaled_ok) {
            TnArgs a{M, N, K, dy, lddy, x, ldx, part_w, part_b, rowscale, rows_per_scale, pd.tiles_k, pd.chunk, pd.tiles_n, tn_xcd, 0, hdr, pd.splits, pd.npad};
            const int grid = pd.tiles_n * pd.tiles_k * pd.splits;
            if (rowscale) return pd.tk == 256 ? launch_tn_dma<256, 256, 4, 2, true>(a, grid, st) : launch_tn_dma<192, 384, 4, 2, true>(a, grid, st);
            if (pd.tk == 192) return launch_tn_dma<384, 192, 4, 4>(a, grid, st);
            return pd.tk == 256 ? launch_tn_dma<256, 256, 4, 2>(a, grid, st) : launch_tn_dma<192, 384, 4, 2>(a, grid, st);
        }
    }
    const TnPlan pl = tn_plan(M, N, K, dtype);
    TnArgs a{M, N, K, dy, lddy, x, ldx, part_w, part_b, rowscale, rows_per_scale, pl.tiles_k, pl.chunk, pl.tiles_n, tn_xcd, x_epi == FMMT_EPI_GELU, hdr, pl.splits};
    dim3 grid(pl.tiles_n * pl.tiles_k * pl.splits);
    static const int tn_cfg = fmmt_const("FMMT_TN_CFG", 0);
    // measured (tools/probes/gemm_bench.py): 64-token steps win for the compute-heavy stage-2/3 shapes (+15-25 %),
    // 32-token steps (3 workgroups per CU) win for the HBM-bound multi-million-token stage-0/1 shapes
    const bool bms64 = tn_cfg == 2 || (tn_cfg == 0 && M <= 262144);
    // two token steps in flight (register sets R0/R1): measured +3..5 % on the stage-2/3 shapes, +3..11 % on the stage-0/1
    // ones, at unchanged occupancy (200 / 160 registers); FMMT_TN_PF=1 selects the single-step prefetch
    static const int tn_pf = fmmt_const("FMMT_TN_PF", 3);
    if (dtype == FMMT_BF16) {
        // few-token problems: FMMT_TN_FEW64=1 (A/B switch) takes 64-token steps with two register sets in flight
        static const int few64 = fmmt_const("FMMT_TN_FEW64", 0);
        if (M <= 4096) return few64 ? launch_tn<bf16, 64, true, 2>(a, grid, st) : launch_tn<bf16, 32, true>(a, grid, st);
        if (bms64) return tn_pf >= 2 ? launch_tn<bf16, 64, false, 2>(a, grid, st) : launch_tn<bf16, 64>(a, grid, st);
        return tn_pf >= 3 ? launch_tn<bf16, 32, false, 2>(a, grid, st) : launch_tn<bf16, 32>(a, grid, st);
    }
    return launch_tn<float, 16>(a, grid, st);
}
}  // namespace

extern "C" int fmmt_linear_wgrad_partials(int dtype, int M, int N, int K,
                                          const void* dy, int lddy, const void* x, int ldx, int want_bias,
                                          const float* rowscale, int rows_per_scale, int x_epi,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return FMMT_EINVAL;
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    const int vec = dtype == FMMT_BF16 ? 8 : 4;
    if (N % vec || K % vec || lddy % vec || ldx % vec) return FMMT_EINVAL;
    if (rowscale && rows_per_scale <= 0) return FMMT_EINVAL;
    if (!aligned16(dy) || !aligned16(x) || !aligned16(workspace)) return FMMT_EALIGN;
    if (workspace_bytes < tn_ws_bytes(M, N, K, dtype)) return FMMT_EWORKSPACE;
    const int smax = tn_smax(M, N, K, dtype);
    int* hdr = reinterpret_cast<int*>(workspace);
    float* part_b = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + TN_HDR);
    float* part_w = part_b + (size_t)smax * tn_rows(M, N, K, dtype);
    return launch_tn_plan(dtype, M, N, K, dy, lddy, x, ldx, part_w, want_bias ? part_b : nullptr, rowscale, rows_per_scale, x_epi, hdr,
                          reinterpret_cast<hipStream_t>(stream));
}

extern "C" int fmmt_linear_wgrad_finish(int dtype, int M, int N, int K, float* dw, float* db,
                                        const void* workspace, size_t workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !dw) return FMMT_EINVAL;
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    if (!aligned16(dw) || !aligned16(workspace)) return FMMT_EALIGN;
    if (workspace_bytes < tn_ws_bytes(M, N, K, dtype)) return FMMT_EWORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int smax = tn_smax(M, N, K, dtype);
    const int* hdr = reinterpret_cast<const int*>(workspace);
    const float* part_b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(workspace) + TN_HDR);
    const int rows = tn_rows(M, N, K, dtype);                // >= N: the grid covers the padded partials, the kernel reads the row count actually written from the header
    const float* part_w = part_b + (size_t)smax * rows;
    const size_t nw = (size_t)N * K;
    if (nw % 4 || N % 4) return FMMT_EINVAL;
    const int wblocks = (int)(((size_t)rows * K / 4 + 63) / 64), bblocks = db ? (rows / 4 + 63) / 64 : 0;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(wblocks + bblocks), dim3(64 * RP_GROUPS), 0, st, part_w, dw, nw, hdr, wblocks, part_b, db, (size_t)N, K);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_linear_wgrad(int dtype, int M, int N, int K,
                                 const void* dy, int lddy, const void* x, int ldx,
                                 float* dw, float* db, const float* rowscale, int rows_per_scale, int x_epi,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!aligned16(dw)) return FMMT_EALIGN;
    const bool few = M > 0 && N > 0 && K > 0 && tn_few_ok(M, N, K, dtype) && !rowscale && !x_epi && lddy % 8 == 0 && ldx % 8 == 0;
    if (M > 0 && N > 0 && K > 0 && (dtype == FMMT_BF16 || dtype == FMMT_F32) && (few || tn_smax(M, N, K, dtype) == 1)) {
        // single split: the "partials" ARE the result -- let the contraction kernel write dw / db directly
        const int vec = dtype == FMMT_BF16 ? 8 : 4;
        if (N % vec || K % vec || lddy % vec || ldx % vec) return FMMT_EINVAL;
        if (rowscale && rows_per_scale <= 0) return FMMT_EINVAL;
        if (!aligned16(dy) || !aligned16(x)) return FMMT_EALIGN;
        return launch_tn_plan(dtype, M, N, K, dy, lddy, x, ldx, dw, db, rowscale, rows_per_scale, x_epi, nullptr, reinterpret_cast<hipStream_t>(stream));
    }
    if (int rc = fmmt_linear_wgrad_partials(dtype, M, N, K, dy, lddy, x, ldx, db != nullptr, rowscale, rows_per_scale, x_epi,
                                            workspace, workspace_bytes, stream)) return rc;
    return fmmt_linear_wgrad_finish(dtype, M, N, K, dw, db, workspace, workspace_bytes, stream);
}
