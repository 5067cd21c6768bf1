// Multi-head attention core on the matrix cores -- bf16 throughput path, head_dim 64 (gfx950): the cross-modal encoder's attention
// (multihead_attention.py:85-128), the self-attention encoders' and -- batch-major -- the text encoder's (16 heads x 64, 4 x 512 tokens).
// Flash-style with an online softmax in the log2 domain.  A workgroup of FOUR waves owns 64 query rows of one (batch, head) (forward, dQ) or 64
// keys (dK / dV); each wave owns 16 of them and the four share the streamed 64-row tiles (K and V, or Q and dO) through LDS, staged
// cooperatively with the next tile's global loads in flight under the current tile's arithmetic.  (Until round 6 a workgroup was ONE wave that
// owned all 64 rows: 512 waves on the 1024 SIMDs for the text shape, and the softmax's VALU work never overlapped anyone's MFMAs -- 75 / 178 us
// forward / backward there against 47 / 103 for the library kernels torch dispatches to.)
// S^T = K.Q^T puts a query column in each lane (softmax needs two cross-lane steps), P^T feeds O^T = V^T.P^T from the registers it was produced
// in; V^T / K^T / Q^T / dO^T come from natural-layout LDS tiles via ds_read_b64_tr_b16 with the channel permutation that makes every lane store
// 16-byte row pieces.  Same building blocks as wattn_mfma.hip.
//
// Backward recomputes P from the saved log-sum-exp: a query-owner kernel (dQ) and a key-owner kernel (dK, dV).  Dropout keep-masks are a pure
// function of (seed, query row, key) -- fmmt_common.h: one pair of 32-bit mixes per four consecutive keys -- and are replayed bit-identically;
// the key-owner kernel, whose lanes hold four QUERIES of one key, computes one group per lane and passes the words around its lane quad (DPP).
#include "fmmt_common.h"
#include "../../include/fmmt.h"
#include "mha_args.h"

namespace {

constexpr int D = 64;
constexpr int VP = 72;            // LDS tile pitch (bf16): 144-byte rows
constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 ldg8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 lds8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
    return z;
}
__device__ __forceinline__ bf16x8 pack8(const float* lo4, const float* hi4) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (bf16)lo4[e]; v[4 + e] = (bf16)hi4[e]; }
    return v;
}
__device__ __forceinline__ float xor_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// A-operand fragment of X^T for X stored [row][D] (pitch VP) in LDS.  MFMA row i = 4*g' + r' <-> channel
// (dt>>1)*32 + g'*8 + (dt&1)*4 + r'; k-slot (g, e) <-> row r0 + e (e < 4) / r0 + 16 + e - 4, r0 = 32*ks + 4*g.
__device__ __forceinline__ bf16x8 tr_fragT(const bf16* tile, int r0, int dt, int li) {
    const bf16* a0 = tile + (r0 + (li >> 2)) * VP + (dt >> 1) * 32 + (li & 3) * 8 + (dt & 1) * 4;
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * VP));
    return u.v;
}

// two 16-byte row pieces per lane: channels [lg*8, lg*8+8) and [32 + lg*8, ...) from accumulators acc[dt][r]
__device__ __forceinline__ void store_row(bf16* dst, const f32x4* acc, float mul, int lg) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r] = (bf16)(acc[2 * c][r] * mul); o[4 + r] = (bf16)(acc[2 * c + 1][r] * mul); }
        *reinterpret_cast<bf16x8*>(dst + c * 32 + lg * 8) = o;
    }
}

// Workgroups that share a (batch, head) -- the blocks along x -- on ONE XCD (its L2 then holds that head's K / V, or Q / dO, once): hardware
// deals linear workgroup ids round-robin over the 8 XCDs, so XCD x is given the x-th eighth of the (head-major) logical order.
__device__ __forceinline__ void block_of(int& bx, int& bh) {
    const int nx = gridDim.x, n = nx * gridDim.y;
    int w = blockIdx.y * nx + blockIdx.x;
    if ((n & 7) == 0) w = (w & 7) * (n >> 3) + (w >> 3);
    bh = w / nx;
    bx = w - bh * nx;
}

// a wave's 16 rows of a 64-row tile: row r0 + wave*16 + li, 16-byte pieces lg*8 and 32 + lg*8; rows >= L read as zero
struct RowRegs { bf16x8 v[2]; };
__device__ __forceinline__ RowRegs fetch_row(const bf16* base, size_t rowoff, bool ok) {
    RowRegs r;
    r.v[0] = ok ? ldg8(base + rowoff) : zero8();
    r.v[1] = ok ? ldg8(base + rowoff + 32) : zero8();
    return r;
}
__device__ __forceinline__ void put_row(bf16* tile, int sr, int lg, const RowRegs& r) {
    *reinterpret_cast<bf16x8*>(tile + sr * VP + lg * 8) = r.v[0];
    *reinterpret_cast<bf16x8*>(tile + sr * VP + 32 + lg * 8) = r.v[1];
}

// the four keep multipliers of keys 4g .. 4g+3 of query row `row`
__device__ __forceinline__ void keep4(const AttnDrop& d, uint32_t row, uint32_t g, float* k4) {
    uint32_t a, b;
    attn_drop_words(d, row, g, a, b);
    k4[0] = (a & 0xFFFFu) >= d.thresh ? d.inv : 0.f;
    k4[1] = (a >> 16) >= d.thresh ? d.inv : 0.f;
    k4[2] = (b & 0xFFFFu) >= d.thresh ? d.inv : 0.f;
    k4[3] = (b >> 16) >= d.thresh ? d.inv : 0.f;
}

// =============================================================================================
// forward.  Kb: the tile's per-key logit terms in the log2 domain (key_bias, 0 without one, -inf for keys >= Lk), staged with the tile
template <bool DROP>
__device__ __forceinline__ void fwd_tile(const AttnDrop& dr, const bf16* Kt, const bf16* Vt, const float* Kb, const bf16x8* qf, f32x4* o, float& m, float& l,
                                         float c2, uint32_t drow, int j0, int li, int lg) {
    float s[16];
    float tmax = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const bf16* krow = Kt + (kt * 16 + li) * VP + lg * 8;
        f32x4 a = mfma(lds8(krow), qf[0], f32x4{0.f, 0.f, 0.f, 0.f});
        a = mfma(lds8(krow + 32), qf[1], a);
        const f32x4 kb = *reinterpret_cast<const f32x4*>(Kb + kt * 16 + lg * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = __builtin_fmaf(a[r], c2, kb[r]);
            s[kt * 4 + r] = v;
            tmax = fmaxf(tmax, v);
        }
    }
    const float mnew = fmaxf(m, xor_max(tmax));
    const float alpha = ex2(m - mnew);
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        s[e] = ex2(s[e] - mnew);
        ls += s[e];
    }
    l = l * alpha + xor_sum(ls);
    m = mnew;
    if constexpr (DROP) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            float k4[4];
            keep4(dr, drow, (uint32_t)(j0 + kt * 16 + lg * 4) >> 2, k4);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt * 4 + r] *= k4[r];
        }
    }
    const bf16x8 pb0 = pack8(&s[0], &s[4]), pb1 = pack8(&s[8], &s[12]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        f32x4 acc = o[dt];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= alpha;
        acc = mfma(tr_fragT(Vt, 4 * lg, dt, li), pb0, acc);
        o[dt] = mfma(tr_fragT(Vt, 32 + 4 * lg, dt, li), pb1, acc);
    }
}

// the per-key term of key `key` (log2 domain); the first 64 threads of a workgroup fetch one each with the tile
__device__ __forceinline__ float key_term(const MhaArgs& p, int b, int key) {
    return key < p.Lk ? (p.key_bias ? p.key_bias[(size_t)b * p.Lk + key] * LOG2E : 0.f) : NEG_BIG;
}

template <bool DROP>
__global__ __launch_bounds__(256) void mha_mfma_fwd_kernel(MhaArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 Kt[64 * VP];
    __shared__ __attribute__((aligned(16))) bf16 Vt[64 * VP];
    __shared__ __attribute__((aligned(16))) float Kb[64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    int bx, bh;
    block_of(bx, bh);
    const int b = bh / p.nH, h = bh - b * p.nH;
    const int q0 = bx * 64 + wave * 16, sr = wave * 16 + li;
    const bool active = q0 < p.Lq;                          // wave-uniform: a wave whose 16 queries are all past the end only helps staging
    const bf16* kg = reinterpret_cast<const bf16*>(p.k) + h * D + lg * 8;
    const bf16* vg = reinterpret_cast<const bf16*>(p.v) + h * D + lg * 8;
    const int q = min(q0 + li, p.Lq - 1);
    bf16x8 qf[2];
    {
        const bf16* src = reinterpret_cast<const bf16*>(p.q) + p.rq(q, b) * p.ldq + h * D + lg * 8;
        qf[0] = ldg8(src);
        qf[1] = ldg8(src + 32);
    }
    AttnDrop dr{};
    if constexpr (DROP) dr = attn_drop_setup(p.drop_p, p.seed_dev ? *p.seed_dev : p.seed, p.Lk);
    const uint32_t drow = (uint32_t)bh * (uint32_t)p.Lq + (uint32_t)q;
    const float c2 = p.scale * LOG2E;
    f32x4 o[4];
    float m = NEG_BIG, l = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    RowRegs kr = fetch_row(kg, p.rk(min(sr, p.Lk - 1), b) * p.ldkv, sr < p.Lk);
    RowRegs vr = fetch_row(vg, p.rk(min(sr, p.Lk - 1), b) * p.ldkv, sr < p.Lk);
    float kbr = wave == 0 ? key_term(p, b, lane) : 0.f;
    for (int j0 = 0; j0 < p.Lk; j0 += 64) {
        __syncthreads();
        put_row(Kt, sr, lg, kr);
        put_row(Vt, sr, lg, vr);
        if (wave == 0) Kb[lane] = kbr;
        __syncthreads();
        if (j0 + 64 < p.Lk) {
            const int row = j0 + 64 + sr;
            const size_t off = p.rk(min(row, p.Lk - 1), b) * p.ldkv;
            kr = fetch_row(kg, off, row < p.Lk);
            vr = fetch_row(vg, off, row < p.Lk);
            if (wave == 0) kbr = key_term(p, b, j0 + 64 + lane);
        }
        if (active) fwd_tile<DROP>(dr, Kt, Vt, Kb, qf, o, m, l, c2, drow, j0, li, lg);
    }
    if (q0 + li < p.Lq) {
        store_row(reinterpret_cast<bf16*>(p.out) + p.rq(q, b) * p.ldo + h * D, o, 1.0f / l, lg);
        if (lg == 0) p.lse[(size_t)bh * p.Lq + q] = (m + __log2f(l)) * LN2;
    }
}

// =============================================================================================
// dQ: wave owns 16 queries, the workgroup streams key / value tiles
template <bool DROP>
__device__ __forceinline__ void dq_tile(const AttnDrop& dr, const bf16* Kt, const bf16* Vt, const float* Kb, const bf16x8* qf, const bf16x8* gf, f32x4* dq,
                                        float ls2, float dl, float c2, uint32_t drow, int j0, int li, int lg) {
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {                       // 32 keys at a time: key tiles 2*ks2, 2*ks2+1
        float ds[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kt = 2 * ks2 + u;
            const bf16* krow = Kt + (kt * 16 + li) * VP + lg * 8;
            const bf16* vrow = Vt + (kt * 16 + li) * VP + lg * 8;
            f32x4 a = mfma(lds8(krow), qf[0], f32x4{0.f, 0.f, 0.f, 0.f});
            a = mfma(lds8(krow + 32), qf[1], a);
            f32x4 dp = mfma(lds8(vrow), gf[0], f32x4{0.f, 0.f, 0.f, 0.f});
            dp = mfma(lds8(vrow + 32), gf[1], dp);
            const f32x4 kb = *reinterpret_cast<const f32x4*>(Kb + kt * 16 + lg * 4);
            float k4[4];
            if constexpr (DROP) keep4(dr, drow, (uint32_t)(j0 + kt * 16 + lg * 4) >> 2, k4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pij = ex2(__builtin_fmaf(a[r], c2, kb[r] - ls2));
                ds[u * 4 + r] = pij * ((DROP ? dp[r] * k4[r] : dp[r]) - dl);
            }
        }
        const bf16x8 dsb = pack8(&ds[0], &ds[4]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma(tr_fragT(Kt, 32 * ks2 + 4 * lg, dt, li), dsb, dq[dt]);
    }
}

template <bool DROP>
__global__ __launch_bounds__(256) void mha_mfma_bwd_dq_kernel(MhaArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 Kt[64 * VP];
    __shared__ __attribute__((aligned(16))) bf16 Vt[64 * VP];
    __shared__ __attribute__((aligned(16))) float Kb[64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    int bx, bh;
    block_of(bx, bh);
    const int b = bh / p.nH, h = bh - b * p.nH;
    const int q0 = bx * 64 + wave * 16, sr = wave * 16 + li;
    const bool active = q0 < p.Lq;
    const bf16* kg = reinterpret_cast<const bf16*>(p.k) + h * D + lg * 8;
    const bf16* vg = reinterpret_cast<const bf16*>(p.v) + h * D + lg * 8;
    const int q = min(q0 + li, p.Lq - 1);
    bf16x8 qf[2], gf[2];
    float dl;
    {
        const bf16* src = reinterpret_cast<const bf16*>(p.q) + p.rq(q, b) * p.ldq + h * D + lg * 8;
        qf[0] = ldg8(src);
        qf[1] = ldg8(src + 32);
        const size_t orow = p.rq(q, b) * p.ldo + h * D + lg * 8;
        gf[0] = ldg8(reinterpret_cast<const bf16*>(p.dout) + orow);
        gf[1] = ldg8(reinterpret_cast<const bf16*>(p.dout) + orow + 32);
        const bf16x8 o0 = ldg8(reinterpret_cast<const bf16*>(p.out) + orow), o1 = ldg8(reinterpret_cast<const bf16*>(p.out) + orow + 32);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)gf[0][e] * (float)o0[e] + (float)gf[1][e] * (float)o1[e];
        dl = xor_sum(d);
    }
    const float ls2 = p.lse[(size_t)bh * p.Lq + q] * LOG2E;
    AttnDrop dr{};
    if constexpr (DROP) dr = attn_drop_setup(p.drop_p, p.seed_dev ? *p.seed_dev : p.seed, p.Lk);
    const uint32_t drow = (uint32_t)bh * (uint32_t)p.Lq + (uint32_t)q;
    const float c2 = p.scale * LOG2E;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    RowRegs kr = fetch_row(kg, p.rk(min(sr, p.Lk - 1), b) * p.ldkv, sr < p.Lk);
    RowRegs vr = fetch_row(vg, p.rk(min(sr, p.Lk - 1), b) * p.ldkv, sr < p.Lk);
    float kbr = wave == 0 ? key_term(p, b, lane) : 0.f;
    for (int j0 = 0; j0 < p.Lk; j0 += 64) {
        __syncthreads();
        put_row(Kt, sr, lg, kr);
        put_row(Vt, sr, lg, vr);
        if (wave == 0) Kb[lane] = kbr;
        __syncthreads();
        if (j0 + 64 < p.Lk) {
            const int row = j0 + 64 + sr;
            const size_t off = p.rk(min(row, p.Lk - 1), b) * p.ldkv;
            kr = fetch_row(kg, off, row < p.Lk);
            vr = fetch_row(vg, off, row < p.Lk);
            if (wave == 0) kbr = key_term(p, b, j0 + 64 + lane);
        }
        if (active) dq_tile<DROP>(dr, Kt, Vt, Kb, qf, gf, dq, ls2, dl, c2, drow, j0, li, lg);
    }
    if (q0 + li < p.Lq) store_row(reinterpret_cast<bf16*>(p.dq) + p.rq(q, b) * p.lddq + h * D, dq, p.scale, lg);
}

// =============================================================================================
// dK, dV: wave owns 16 keys, the workgroup streams query tiles of 64 (Q, dO, and per query the log-sum-exp and delta = rowsum(dO * O))
template <bool DROP>
__global__ __launch_bounds__(256) void mha_mfma_bwd_dkv_kernel(MhaArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 Qt[64 * VP];
    __shared__ __attribute__((aligned(16))) bf16 Gt[64 * VP];
    __shared__ __attribute__((aligned(16))) float Ls[64];       // log2-domain log-sum-exp; +big for rows >= Lq (their probabilities become 0)
    __shared__ __attribute__((aligned(16))) float Dl[64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    int bx, bh;
    block_of(bx, bh);
    const int b = bh / p.nH, h = bh - b * p.nH;
    const int k0 = bx * 64 + wave * 16, sr = wave * 16 + li;
    const bool active = k0 < p.Lk;
    const bf16* qg = reinterpret_cast<const bf16*>(p.q) + h * D + lg * 8;
    const bf16* og = reinterpret_cast<const bf16*>(p.out) + h * D + lg * 8;
    const bf16* gg = reinterpret_cast<const bf16*>(p.dout) + h * D + lg * 8;
    const int key = min(k0 + li, p.Lk - 1);
    bf16x8 kf[2], vf[2];
    {
        const size_t off = p.rk(key, b) * p.ldkv + h * D + lg * 8;
        kf[0] = ldg8(reinterpret_cast<const bf16*>(p.k) + off);
        kf[1] = ldg8(reinterpret_cast<const bf16*>(p.k) + off + 32);
        vf[0] = ldg8(reinterpret_cast<const bf16*>(p.v) + off);
        vf[1] = ldg8(reinterpret_cast<const bf16*>(p.v) + off + 32);
    }
    const float kb2 = p.key_bias ? p.key_bias[(size_t)b * p.Lk + key] * LOG2E : 0.f;
    AttnDrop dr{};
    if constexpr (DROP) dr = attn_drop_setup(p.drop_p, p.seed_dev ? *p.seed_dev : p.seed, p.Lk);
    const uint32_t kgroup = (uint32_t)(k0 + li) >> 2;        // the same for the four lanes of a quad (k0 is a multiple of 16)
    // which field of its group this lane's key is: a or b, low or high half
    const bool f_b = (key & 2) != 0;
    const uint32_t f_sh = (key & 1) ? 16u : 0u;
    const float c2 = p.scale * LOG2E;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    RowRegs qr, gr;
    float dpre, lpre;
    auto fetch = [&](int i0) {
        const int row = i0 + sr;
        const bool ok = row < p.Lq;
        const int rc = min(row, p.Lq - 1);
        qr = fetch_row(qg, p.rq(rc, b) * p.ldq, ok);
        const size_t off = p.rq(rc, b) * p.ldo;
        gr = fetch_row(gg, off, ok);
        const bf16x8 o0 = ldg8(og + off), o1 = ldg8(og + off + 32);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)gr.v[0][e] * (float)o0[e] + (float)gr.v[1][e] * (float)o1[e];
        dpre = xor_sum(d);
        lpre = ok ? p.lse[(size_t)bh * p.Lq + rc] * LOG2E : -NEG_BIG;
    };
    fetch(0);
    for (int i0 = 0; i0 < p.Lq; i0 += 64) {
        __syncthreads();
        put_row(Qt, sr, lg, qr);
        put_row(Gt, sr, lg, gr);
        if (lg == 0) {
            Dl[sr] = dpre;
            Ls[sr] = lpre;
        }
        __syncthreads();
        if (i0 + 64 < p.Lq) fetch(i0 + 64);
        if (!active) continue;
        float pp[16], ds[16];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const bf16* qrow = Qt + (qt * 16 + li) * VP + lg * 8;
            const bf16* grow = Gt + (qt * 16 + li) * VP + lg * 8;
            f32x4 a = mfma(lds8(qrow), kf[0], f32x4{0.f, 0.f, 0.f, 0.f});
            a = mfma(lds8(qrow + 32), kf[1], a);
            f32x4 dp = mfma(lds8(grow), vf[0], f32x4{0.f, 0.f, 0.f, 0.f});
            dp = mfma(lds8(grow + 32), vf[1], dp);
            const f32x4 lq = *reinterpret_cast<const f32x4*>(&Ls[qt * 16 + lg * 4]);
            const f32x4 dd = *reinterpret_cast<const f32x4*>(&Dl[qt * 16 + lg * 4]);
            float ksc[4];
            if constexpr (DROP) {
                // lane (quad position u = li & 3) hashes the group of query lg*4 + u; query r's words come from quad lane r
                const uint32_t qrow_u = (uint32_t)bh * (uint32_t)p.Lq + (uint32_t)min(i0 + qt * 16 + lg * 4 + (li & 3), p.Lq - 1);
                uint32_t wa, wb;
                attn_drop_words(dr, qrow_u, kgroup, wa, wb);
                // which WORD a lane needs is decided by its own key, whose words by the source lane: move both (quad_perm broadcasts), then select
                const int ia = (int)wa, ib = (int)wb;
                const uint32_t a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, ia, 0x00, 0xF, 0xF, false), b0 = (uint32_t)__builtin_amdgcn_update_dpp(0, ib, 0x00, 0xF, 0xF, false);
                const uint32_t a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, ia, 0x55, 0xF, 0xF, false), b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, ib, 0x55, 0xF, 0xF, false);
                const uint32_t a2 = (uint32_t)__builtin_amdgcn_update_dpp(0, ia, 0xAA, 0xF, 0xF, false), b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, ib, 0xAA, 0xF, 0xF, false);
                const uint32_t a3 = (uint32_t)__builtin_amdgcn_update_dpp(0, ia, 0xFF, 0xF, 0xF, false), b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, ib, 0xFF, 0xF, 0xF, false);
                const uint32_t ws[4] = {f_b ? b0 : a0, f_b ? b1 : a1, f_b ? b2 : a2, f_b ? b3 : a3};
#pragma unroll
                for (int r = 0; r < 4; ++r) ksc[r] = ((ws[r] >> f_sh) & 0xFFFFu) >= dr.thresh ? dr.inv : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pij = ex2(__builtin_fmaf(a[r], c2, kb2) - lq[r]);
                if constexpr (DROP) {
                    pp[qt * 4 + r] = pij * ksc[r];
                    ds[qt * 4 + r] = pij * (dp[r] * ksc[r] - dd[r]);
                } else {
                    pp[qt * 4 + r] = pij;
                    ds[qt * 4 + r] = pij * (dp[r] - dd[r]);
                }
            }
        }
        const bf16x8 p0 = pack8(&pp[0], &pp[4]), p1 = pack8(&pp[8], &pp[12]);
        const bf16x8 d0 = pack8(&ds[0], &ds[4]), d1 = pack8(&ds[8], &ds[12]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = mfma(tr_fragT(Gt, 4 * lg, dt, li), p0, dv[dt]);
            dv[dt] = mfma(tr_fragT(Gt, 32 + 4 * lg, dt, li), p1, dv[dt]);
            dk[dt] = mfma(tr_fragT(Qt, 4 * lg, dt, li), d0, dk[dt]);
            dk[dt] = mfma(tr_fragT(Qt, 32 + 4 * lg, dt, li), d1, dk[dt]);
        }
    }
    if (k0 + li < p.Lk) {
        const size_t off = p.rk(key, b) * p.lddkv + h * D;
        store_row(reinterpret_cast<bf16*>(p.dk) + off, dk, p.scale, lg);
        store_row(reinterpret_cast<bf16*>(p.dv) + off, dv, 1.0f, lg);
    }
}

}  // namespace

int fmmt_mha_mfma_fwd_launch(const MhaArgs& a, hipStream_t st) {
    const dim3 grid((a.Lq + 63) / 64, a.B * a.nH);
    if (a.drop_p > 0.f) hipLaunchKernelGGL(mha_mfma_fwd_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(mha_mfma_fwd_kernel<false>, grid, dim3(256), 0, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

int fmmt_mha_mfma_bwd_launch(const MhaArgs& a, hipStream_t st) {
    const dim3 gq((a.Lq + 63) / 64, a.B * a.nH), gk((a.Lk + 63) / 64, a.B * a.nH);
    if (a.drop_p > 0.f) hipLaunchKernelGGL(mha_mfma_bwd_dq_kernel<true>, gq, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(mha_mfma_bwd_dq_kernel<false>, gq, dim3(256), 0, st, a);
    FMMT_CHECK_LAUNCH();
    if (a.drop_p > 0.f) hipLaunchKernelGGL(mha_mfma_bwd_dkv_kernel<true>, gk, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(mha_mfma_bwd_dkv_kernel<false>, gk, dim3(256), 0, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}
