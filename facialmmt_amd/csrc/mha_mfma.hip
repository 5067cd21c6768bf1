// Cross-modal multi-head attention core on the matrix cores -- bf16 throughput path, head_dim 64 (gfx950).
// Flash-style: one wave owns 64 query rows of one (batch, head) and streams keys/values in tiles of 64
// with an online softmax; S^T = K.Q^T puts a query column in each lane (softmax needs two cross-lane
// steps), P^T feeds O^T = V^T.P^T from the registers it was produced in, V^T / K^T / Q^T / dO^T come from
// natural-layout LDS tiles via ds_read_b64_tr_b16 with the channel permutation that makes every lane
// store 16-byte row pieces.  Same building blocks as wattn_mfma.hip; time-major operands
// (row t of a (L, B, ld) tensor lives at (t*B + b)*ld + h*64).
//
// Backward recomputes P from the saved log-sum-exp: a query-owner kernel (dQ) and a key-owner kernel
// (dK, dV; one wave per 32 keys).  Dropout keep-masks are a pure function of (seed, element index) and
// are replayed bit-identically.
#include "fmmt_common.h"
#include "../../include/fmmt.h"
#include "mha_args.h"

namespace {

constexpr int D = 64;
constexpr int VP = 72;            // LDS tile pitch (bf16): 144-byte rows
constexpr float NEG_BIG = -1.0e30f;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 ldg8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
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

// A-operand fragment of X^T for X stored [row][D] (pitch VP) in LDS.  MFMA row i = 4*g' + r' <-> channel
// (dt>>1)*32 + g'*8 + (dt&1)*4 + r'; k-slot (g, e) <-> row r0 + e (e < 4) / r0 + 16 + e - 4, r0 = 32*ks + 4*g.
__device__ __forceinline__ bf16x8 tr_fragT(const bf16* tile, int r0, int dt, int li) {
    const bf16* a0 = tile + (r0 + (li >> 2)) * VP + (dt >> 1) * 32 + (li & 3) * 8 + (dt & 1) * 4;
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * VP));
    return u.v;
}

__device__ __forceinline__ float keep_scale(const MhaArgs& p, int bh, int i, int j) {
    if (p.drop_p <= 0.f) return 1.f;
    const uint64_t idx = ((uint64_t)bh * p.Lq + i) * p.Lk + j;
    const uint64_t seed = p.seed_dev ? *p.seed_dev : p.seed;
    return hash_uniform(seed, idx) >= p.drop_p ? 1.0f / (1.0f - p.drop_p) : 0.f;
}

// additive logit bias of key j ("(1 - mask) * -10000" extended attention mask of the self-attention encoders); 0 if absent
__device__ __forceinline__ float key_bias(const MhaArgs& p, int b, int j) {
    return p.key_bias ? p.key_bias[(size_t)b * p.Lk + min(j, p.Lk - 1)] : 0.f;
}

// stage 64 rows [r0, r0+64) of a time-major tensor as a natural-layout bf16 tile; rows >= L are zero
__device__ __forceinline__ void stage_tile(const bf16* base, int ld, int B, int b, int h, int r0, int L, bf16* tile, int li, int lg) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = r0 + t * 16 + li;
        const bf16* src = base + ((size_t)min(row, L - 1) * B + b) * ld + h * D + lg * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 v = row < L ? ldg8(src + ks * 32) : zero8();
            *reinterpret_cast<bf16x8*>(tile + (t * 16 + li) * VP + ks * 32 + lg * 8) = v;
        }
    }
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

// =============================================================================================
__global__ __launch_bounds__(64) void mha_mfma_fwd_kernel(MhaArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 Vt[64 * VP];
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.nH, h = bh - b * p.nH;
    const int q0 = blockIdx.x * 64;
    const bf16* qg = reinterpret_cast<const bf16*>(p.q);
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);

    bf16x8 qf[4][2];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = min(q0 + qt * 16 + li, p.Lq - 1);
        const bf16* src = qg + ((size_t)q * p.B + b) * p.ldq + h * D + lg * 8;
        qf[qt][0] = ldg8(src);
        qf[qt][1] = ldg8(src + 32);
    }
    f32x4 o[4][4];
    float m[4], l[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        m[qt] = NEG_BIG;
        l[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int j0 = 0; j0 < p.Lk; j0 += 64) {
        bf16x8 kf[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int key = min(j0 + kt * 16 + li, p.Lk - 1);
            const bf16* src = kg + ((size_t)key * p.B + b) * p.ldkv + h * D + lg * 8;
            kf[kt][0] = ldg8(src);
            kf[kt][1] = ldg8(src + 32);
        }
        __syncthreads();
        stage_tile(vg, p.ldkv, p.B, b, h, j0, p.Lk, Vt, li, lg);
        __syncthreads();
        bf16x8 vT[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vT[ks][dt] = tr_fragT(Vt, 32 * ks + 4 * lg, dt, li);
        float kb[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) kb[e] = key_bias(p, b, j0 + (e >> 2) * 16 + lg * 4 + (e & 3));
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const int q = q0 + qt * 16 + li;
            float s[16];
            float tmax = NEG_BIG;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x4 a = mfma(kf[kt][0], qf[qt][0], f32x4{0.f, 0.f, 0.f, 0.f});
                a = mfma(kf[kt][1], qf[qt][1], a);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = j0 + kt * 16 + lg * 4 + r;
                    const float v = key < p.Lk ? a[r] * p.scale + kb[kt * 4 + r] : NEG_BIG;
                    s[kt * 4 + r] = v;
                    tmax = fmaxf(tmax, v);
                }
            }
            const float mnew = fmaxf(m[qt], xor_max(tmax));
            const float alpha = __expf(m[qt] - mnew);
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = __expf(s[e] - mnew);
                ls += s[e];
            }
            l[qt] = l[qt] * alpha + xor_sum(ls);
            m[qt] = mnew;
            if (p.drop_p > 0.f) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] *= keep_scale(p, bh, min(q, p.Lq - 1), min(j0 + (e >> 2) * 16 + lg * 4 + (e & 3), p.Lk - 1));
            }
            const bf16x8 pb0 = pack8(&s[0], &s[4]), pb1 = pack8(&s[8], &s[12]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 acc = o[qt][dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] *= alpha;
                acc = mfma(vT[0][dt], pb0, acc);
                o[qt][dt] = mfma(vT[1][dt], pb1, acc);
            }
        }
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = q0 + qt * 16 + li;
        if (q < p.Lq) {
            store_row(reinterpret_cast<bf16*>(p.out) + ((size_t)q * p.B + b) * p.ldo + h * D, o[qt], 1.0f / l[qt], lg);
            if (lg == 0) p.lse[(size_t)bh * p.Lq + q] = m[qt] + __logf(l[qt]);
        }
    }
}

// =============================================================================================
// dQ: wave owns 64 queries, streams key tiles
__global__ __launch_bounds__(64) void mha_mfma_bwd_dq_kernel(MhaArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 Kt[64 * VP];
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.nH, h = bh - b * p.nH;
    const int q0 = blockIdx.x * 64;
    const bf16* qg = reinterpret_cast<const bf16*>(p.q);
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    const bf16* og = reinterpret_cast<const bf16*>(p.out);
    const bf16* gg = reinterpret_cast<const bf16*>(p.dout);

    bf16x8 qf[4][2], gf[4][2];
    float ls[4], dl[4];
    f32x4 dq[4][4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = min(q0 + qt * 16 + li, p.Lq - 1);
        const bf16* src = qg + ((size_t)q * p.B + b) * p.ldq + h * D + lg * 8;
        qf[qt][0] = ldg8(src);
        qf[qt][1] = ldg8(src + 32);
        const size_t orow = ((size_t)q * p.B + b) * p.ldo + h * D + lg * 8;
        gf[qt][0] = ldg8(gg + orow);
        gf[qt][1] = ldg8(gg + orow + 32);
        const bf16x8 o0 = ldg8(og + orow), o1 = ldg8(og + orow + 32);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)gf[qt][0][e] * (float)o0[e] + (float)gf[qt][1][e] * (float)o1[e];
        dl[qt] = xor_sum(d);
        ls[qt] = p.lse[(size_t)bh * p.Lq + q];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int j0 = 0; j0 < p.Lk; j0 += 64) {
        __syncthreads();
        stage_tile(kg, p.ldkv, p.B, b, h, j0, p.Lk, Kt, li, lg);
        __syncthreads();
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {                       // 32 keys at a time: key tiles 2*ks2, 2*ks2+1
            bf16x8 kf[2][2], vf[2][2], kT[4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = (2 * ks2 + u) * 16 + li;
                kf[u][0] = *reinterpret_cast<const bf16x8*>(Kt + row * VP + lg * 8);
                kf[u][1] = *reinterpret_cast<const bf16x8*>(Kt + row * VP + 32 + lg * 8);
                const int key = min(j0 + row, p.Lk - 1);
                const bf16* src = vg + ((size_t)key * p.B + b) * p.ldkv + h * D + lg * 8;
                vf[u][0] = ldg8(src);
                vf[u][1] = ldg8(src + 32);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) kT[dt] = tr_fragT(Kt, 32 * ks2 + 4 * lg, dt, li);
            float kb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) kb[e] = key_bias(p, b, j0 + (2 * ks2 + (e >> 2)) * 16 + lg * 4 + (e & 3));
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const int q = min(q0 + qt * 16 + li, p.Lq - 1);
                float ds[8];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 a = mfma(kf[u][0], qf[qt][0], f32x4{0.f, 0.f, 0.f, 0.f});
                    a = mfma(kf[u][1], qf[qt][1], a);
                    f32x4 dp = mfma(vf[u][0], gf[qt][0], f32x4{0.f, 0.f, 0.f, 0.f});
                    dp = mfma(vf[u][1], gf[qt][1], dp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = j0 + (2 * ks2 + u) * 16 + lg * 4 + r;
                        const float pij = key < p.Lk ? __expf(a[r] * p.scale + kb[u * 4 + r] - ls[qt]) : 0.f;
                        const float ksc = p.drop_p > 0.f ? keep_scale(p, bh, q, min(key, p.Lk - 1)) : 1.f;
                        ds[u * 4 + r] = pij * (dp[r] * ksc - dl[qt]);
                    }
                }
                const bf16x8 dsb = pack8(&ds[0], &ds[4]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[qt][dt] = mfma(kT[dt], dsb, dq[qt][dt]);
            }
        }
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = q0 + qt * 16 + li;
        if (q < p.Lq) store_row(reinterpret_cast<bf16*>(p.dq) + ((size_t)q * p.B + b) * p.lddq + h * D, dq[qt], p.scale, lg);
    }
}

// dK, dV: wave owns 32 keys, streams query tiles of 64
__global__ __launch_bounds__(64) void mha_mfma_bwd_dkv_kernel(MhaArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 Qt[64 * VP];
    __shared__ __attribute__((aligned(16))) bf16 Gt[64 * VP];
    __shared__ __attribute__((aligned(16))) float Ls[64];
    __shared__ __attribute__((aligned(16))) float Dl[64];
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.nH, h = bh - b * p.nH;
    const int k0 = blockIdx.x * 32;
    const bf16* qg = reinterpret_cast<const bf16*>(p.q);
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    const bf16* og = reinterpret_cast<const bf16*>(p.out);
    const bf16* gg = reinterpret_cast<const bf16*>(p.dout);

    bf16x8 kf[2][2], vf[2][2];
    f32x4 dk[2][4], dv[2][4];
    float kbk[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = min(k0 + kt * 16 + li, p.Lk - 1);
        kbk[kt] = key_bias(p, b, key);
        const size_t off = ((size_t)key * p.B + b) * p.ldkv + h * D + lg * 8;
        kf[kt][0] = ldg8(kg + off);
        kf[kt][1] = ldg8(kg + off + 32);
        vf[kt][0] = ldg8(vg + off);
        vf[kt][1] = ldg8(vg + off + 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[kt][dt] = dv[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i0 = 0; i0 < p.Lq; i0 += 64) {
        __syncthreads();
        stage_tile(qg, p.ldq, p.B, b, h, i0, p.Lq, Qt, li, lg);
        // dO tile + delta = rowsum(dO * O) + lse for the 64 queries of this tile
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = i0 + t * 16 + li;
            const size_t off = ((size_t)min(row, p.Lq - 1) * p.B + b) * p.ldo + h * D + lg * 8;
            float d = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 g = row < p.Lq ? ldg8(gg + off + ks * 32) : zero8();
                const bf16x8 ov = ldg8(og + off + ks * 32);
                *reinterpret_cast<bf16x8*>(Gt + (t * 16 + li) * VP + ks * 32 + lg * 8) = g;
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (float)g[e] * (float)ov[e];
            }
            d = xor_sum(d);
            if (lg == 0) {
                Dl[t * 16 + li] = d;
                Ls[t * 16 + li] = p.lse[(size_t)bh * p.Lq + min(row, p.Lq - 1)];
            }
        }
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = k0 + kt * 16 + li;
            float pp[16], ds[16];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const bf16* qrow = Qt + (qt * 16 + li) * VP + lg * 8;
                const bf16* grow = Gt + (qt * 16 + li) * VP + lg * 8;
                f32x4 a = mfma(*reinterpret_cast<const bf16x8*>(qrow), kf[kt][0], f32x4{0.f, 0.f, 0.f, 0.f});
                a = mfma(*reinterpret_cast<const bf16x8*>(qrow + 32), kf[kt][1], a);
                f32x4 dp = mfma(*reinterpret_cast<const bf16x8*>(grow), vf[kt][0], f32x4{0.f, 0.f, 0.f, 0.f});
                dp = mfma(*reinterpret_cast<const bf16x8*>(grow + 32), vf[kt][1], dp);
                const f32x4 lq = *reinterpret_cast<const f32x4*>(&Ls[qt * 16 + lg * 4]);
                const f32x4 dq = *reinterpret_cast<const f32x4*>(&Dl[qt * 16 + lg * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = i0 + qt * 16 + lg * 4 + r;
                    const float pij = q < p.Lq ? __expf(a[r] * p.scale + kbk[kt] - lq[r]) : 0.f;
                    const float ksc = p.drop_p > 0.f ? keep_scale(p, bh, min(q, p.Lq - 1), min(key, p.Lk - 1)) : 1.f;
                    pp[qt * 4 + r] = pij * ksc;
                    ds[qt * 4 + r] = pij * (dp[r] * ksc - dq[r]);
                }
            }
            const bf16x8 p0 = pack8(&pp[0], &pp[4]), p1 = pack8(&pp[8], &pp[12]);
            const bf16x8 d0 = pack8(&ds[0], &ds[4]), d1 = pack8(&ds[8], &ds[12]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[kt][dt] = mfma(tr_fragT(Gt, 4 * lg, dt, li), p0, dv[kt][dt]);
                dv[kt][dt] = mfma(tr_fragT(Gt, 32 + 4 * lg, dt, li), p1, dv[kt][dt]);
                dk[kt][dt] = mfma(tr_fragT(Qt, 4 * lg, dt, li), d0, dk[kt][dt]);
                dk[kt][dt] = mfma(tr_fragT(Qt, 32 + 4 * lg, dt, li), d1, dk[kt][dt]);
            }
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = k0 + kt * 16 + li;
        if (key < p.Lk) {
            const size_t off = ((size_t)key * p.B + b) * p.lddkv + h * D;
            store_row(reinterpret_cast<bf16*>(p.dk) + off, dk[kt], p.scale, lg);
            store_row(reinterpret_cast<bf16*>(p.dv) + off, dv[kt], 1.0f, lg);
        }
    }
}

}  // namespace

int fmmt_mha_mfma_fwd_launch(const MhaArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(mha_mfma_fwd_kernel, dim3((a.Lq + 63) / 64, a.B * a.nH), dim3(64), 0, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}

int fmmt_mha_mfma_bwd_launch(const MhaArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(mha_mfma_bwd_dq_kernel, dim3((a.Lq + 63) / 64, a.B * a.nH), dim3(64), 0, st, a);
    FMMT_CHECK_LAUNCH();
    hipLaunchKernelGGL(mha_mfma_bwd_dkv_kernel, dim3((a.Lk + 31) / 32, a.B * a.nH), dim3(64), 0, st, a);
    FMMT_CHECK_LAUNCH();
    return 0;
}
