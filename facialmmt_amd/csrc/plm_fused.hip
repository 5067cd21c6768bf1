// The tail of a BERT / RoBERTa sublayer as one launch per direction (bf16 module, bf16 affine parameters):
//     y = LayerNorm( dropout(h) + res )            transformers' RobertaSelfOutput / RobertaOutput (BertSelfOutput / BertOutput), src/models.py:75-91
// h = the sublayer's dense output (its GEMM stays with the vendor library, R4.3: hipBLASLt owns the 2048-token shapes).
// Stock PyTorch runs this as fused_dropout + add + layer_norm forward (3 launches) and, backward, LayerNorm' (2-3), masked_scale (1) and the dense
// bias' column sum (1): 48 sublayers x ~8 dependent launches of 4-25 us in a chain that runs largely alone on the GPU (the text encoder's forward in
// front of Swin's, the tail of its backward behind it).  Here: one forward launch, one backward launch + one reduction.
//   * the dropout mask is not stored: keep(e) = a 16-bit field of a pair of 32-bit mixes of (seed, salt, e / 4) >= round(p * 2^16), replayed bit for bit in the
//     backward (the counter-based generator of the attention dropout, fmmt_common.h: splitmix64 per element cost more issue slots than the rest of the kernel);
//     kept values are scaled by the exact inverse of the realised keep rate; `seed` is a device int64 word (graph-replay safe), `salt` separates the call sites;
//   * rounding follows the bf16 module op by op: t = bf16(h * keep / (1 - p)), x = bf16(t + res), LayerNorm on x in fp32, y = bf16(...);
//   * backward: dx = LayerNorm'(dy) rounded to bf16 IS the residual branch's gradient; dh = bf16(dx * keep / (1 - p)); partial sums of d gamma, d beta
//     and of the dense bias' gradient colsum(dh) per block, finished in a fixed order by the reduction.
// One wave per token row, rows grid-strided (forward: 4 waves per workgroup, up to 1024 workgroups; backward: 8 waves and at most 256 workgroups -- the count of
// partial rows the reduction reads -- so that 2048 tokens are one row per wave: with 4 waves every wave walked two rows' dependent reduction chains alone on its
// SIMD, 15.6 us for 17 MB); C % 8 == 0, C <= 2048.
#include "fmmt_common.h"
#include "../../include/fmmt.h"

namespace {

constexpr int PF_MAXV = 4;                                  // 8-element vectors per lane: C <= 2048
constexpr int PF_BW = 8;                                    // waves (= rows in flight) per workgroup of the backward

// Affine parameters (and their gradients) in bf16 (the text encoder's bf16 module) or fp32 (MELDTransEncoder's fp32 master LayerNorms,
// modules/Transformer.py:109-137: same op sequence dense -> dropout -> + input -> LayerNorm)
template <typename P> __device__ __forceinline__ void pf_load8(const P* __restrict__ p, float (&o)[8]);
template <> __device__ __forceinline__ void pf_load8<bf16>(const bf16* __restrict__ p, float (&o)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}
template <> __device__ __forceinline__ void pf_load8<float>(const float* __restrict__ p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
}


template <typename P>
__global__ __launch_bounds__(256) void plm_dropadd_ln_fwd_kernel(int M, int C, float eps, const bf16* __restrict__ h, const bf16* __restrict__ res,
                                                                 const P* __restrict__ gamma, const P* __restrict__ beta, float p,
                                                                 unsigned long long seed_i, const unsigned long long* __restrict__ seed_ptr, unsigned long long salt,
                                                                 bf16* __restrict__ xsum, bf16* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C / 8;
    const float invC = 1.0f / (float)C;
    const ElemDrop dr = elem_drop_setup(p, seed_ptr ? seed_ptr[0] : seed_i, salt);
    const float scale = p > 0.f ? dr.inv : 1.0f;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        bf16x8 xv[PF_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PF_MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                const size_t o = (size_t)row * C + v * 8;
                const bf16x8 hv = *reinterpret_cast<const bf16x8*>(h + o), rv = *reinterpret_cast<const bf16x8*>(res + o);
                uint32_t keep = 0xFFu;
                if (p > 0.f) keep = elem_keep4(dr, (uint32_t)(o >> 2)) | (elem_keep4(dr, (uint32_t)(o >> 2) + 1u) << 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = (float)hv[e];
                    if (p > 0.f) t = (keep >> e) & 1u ? (float)(bf16)(t * scale) : 0.f;
                    xv[i][e] = (bf16)(t + (float)rv[e]);
                    s += (float)xv[i][e];
                }
                *reinterpret_cast<bf16x8*>(xsum + o) = xv[i];
            }
        }
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < PF_MAXV; ++i)
            if (lane + 64 * i < nv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)xv[i][e] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(q) * invC + eps);
#pragma unroll
        for (int i = 0; i < PF_MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                float gm[8], bt[8];
                pf_load8<P>(gamma + v * 8, gm);
                pf_load8<P>(beta + v * 8, bt);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)(((float)xv[i][e] - mean) * rstd * gm[e] + bt[e]);
                *reinterpret_cast<bf16x8*>(y + (size_t)row * C + v * 8) = o;
            }
        }
    }
}

// part: [block][3][C] fp32 (d gamma, d beta, d dense-bias)
template <typename P>
__global__ __launch_bounds__(512) void plm_dropadd_ln_bwd_kernel(int M, int C, float eps, const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                                 const P* __restrict__ gamma, float p, unsigned long long seed_i,
                                                                 const unsigned long long* __restrict__ seed_ptr, unsigned long long salt, bf16* __restrict__ dx, bf16* __restrict__ dh,
                                                                 float* __restrict__ part) {
    __shared__ float red[PF_BW][3][64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C / 8;
    float dg[PF_MAXV][8], db[PF_MAXV][8], dc[PF_MAXV][8];
#pragma unroll
    for (int i = 0; i < PF_MAXV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dg[i][e] = db[i][e] = dc[i][e] = 0.f;
    const float invC = 1.0f / (float)C;
    const ElemDrop dr = elem_drop_setup(p, seed_ptr ? seed_ptr[0] : seed_i, salt);
    const float scale = p > 0.f ? dr.inv : 1.0f;
    for (int row = blockIdx.x * PF_BW + wave; row < M; row += gridDim.x * PF_BW) {
        bf16x8 xv[PF_MAXV], gv[PF_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PF_MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                xv[i] = *reinterpret_cast<const bf16x8*>(x + (size_t)row * C + v * 8);
                gv[i] = *reinterpret_cast<const bf16x8*>(dy + (size_t)row * C + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)xv[i][e];
            }
        }
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < PF_MAXV; ++i)
            if (lane + 64 * i < nv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)xv[i][e] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(q) * invC + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PF_MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                float gm[8];
                pf_load8<P>(gamma + v * 8, gm);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)xv[i][e] - mean) * rstd, dyv = (float)gv[i][e], g = dyv * gm[e];
                    s1 += g;
                    s2 += g * xh;
                    dg[i][e] += dyv * xh;
                    db[i][e] += dyv;
                }
            }
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
#pragma unroll
        for (int i = 0; i < PF_MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                const size_t o = (size_t)row * C + v * 8;
                float gm[8];
                pf_load8<P>(gamma + v * 8, gm);
                bf16x8 ox, oh;
                uint32_t keep = 0xFFu;
                if (p > 0.f) keep = elem_keep4(dr, (uint32_t)(o >> 2)) | (elem_keep4(dr, (uint32_t)(o >> 2) + 1u) << 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)xv[i][e] - mean) * rstd, g = (float)gv[i][e] * gm[e];
                    ox[e] = (bf16)(rstd * (g - s1 - xh * s2));
                    float t = (float)ox[e];
                    if (p > 0.f) t = (keep >> e) & 1u ? t * scale : 0.f;
                    oh[e] = (bf16)t;
                    dc[i][e] += (float)oh[e];
                }
                *reinterpret_cast<bf16x8*>(dx + o) = ox;
                *reinterpret_cast<bf16x8*>(dh + o) = oh;
            }
        }
    }
    float* pb = part + (size_t)blockIdx.x * 3 * C;
#pragma unroll
    for (int i = 0; i < PF_MAXV; ++i) {
        if (64 * i >= nv) break;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[wave][0][lane * 8 + e] = dg[i][e];
            red[wave][1][lane * 8 + e] = db[i][e];
            red[wave][2][lane * 8 + e] = dc[i][e];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 3 * 512; t += 64 * PF_BW) {
            const int which = t >> 9, col = t & 511, ch = i * 512 + col;
            if (ch < C) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < PF_BW; ++w) a += red[w][which][col];      // fixed order
                pb[which * C + ch] = a;
            }
        }
    }
}

// 1024 threads = 32 columns x 32 partial groups, fixed-order tree over the blocks' [3][C] partials
template <typename P>
__global__ __launch_bounds__(1024) void plm_dropadd_ln_reduce_kernel(const float* __restrict__ part, int nblocks, int C, P* __restrict__ dgamma,
                                                                    P* __restrict__ dbeta, P* __restrict__ dbias) {
    __shared__ float red[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;                     // column of the [3][C] triple
    float a = 0.f;
    if (i < 3 * C) {
        float v[8];                                         // nblocks <= 256: every load of the thread in flight at once (a rolled loop waited for each, ~2 us apiece)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = ty + 32 * u;
            v[u] = b < nblocks ? part[(size_t)b * 3 * C + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
        for (int b = ty + 256; b < nblocks; b += 32) a += part[(size_t)b * 3 * C + i];
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && i < 3 * C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += red[g][tx];
        if (i < C) dgamma[i] = (P)t;
        else if (i < 2 * C) dbeta[i - C] = (P)t;
        else if (dbias) dbias[i - 2 * C] = (P)t;
    }
}

// Backward of the text encoder's intermediate activation together with the bias gradient of the Linear in front of it (transformers' *Intermediate:
// dense -> GELU, src/models.py:75-91):  d(pre) = d(act) * gelu'(pre),  d(bias) = colsum(d(pre)) -- stock PyTorch runs GeluBackward and a column sum (fmmt_colsum
// here) as two passes over the (tokens x 4096) matrix.  A thread owns columns: its sums over the block's rows need no cross-thread step; per-block partial rows,
// then a fixed-order reduction over the blocks.  H % 8 == 0, H <= 8192.
constexpr int GB_MAXV = 4;                                  // 8-element vectors per thread: H <= 256 * 8 * 4
constexpr int GB_BLOCKS = 256;                              // partial rows the reduction reads
constexpr int GB_PH = 4;                                    // row phases per workgroup: 1024 threads = four waves per SIMD, phase p takes the block's rows p, p + 4, ...
                                                            // (one phase, 256 threads: 20 us for 50 MB at 2048 x 4096 -- one wave per SIMD walking eight rows)
__global__ __launch_bounds__(256 * GB_PH) void plm_gelu_bwd_colsum_kernel(int M, int H, const bf16* __restrict__ dact, const bf16* __restrict__ pre, bf16* __restrict__ dpre,
                                                                         float* __restrict__ part) {
    __shared__ float red[GB_PH - 1][256][8];
    const int nv = H / 8, tc = threadIdx.x & 255, ph = threadIdx.x >> 8;
    float acc[GB_MAXV][8];
#pragma unroll
    for (int j = 0; j < GB_MAXV; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    for (int row = blockIdx.x + gridDim.x * ph; row < M; row += gridDim.x * GB_PH) {
        bf16x8 g[GB_MAXV], x[GB_MAXV];
#pragma unroll
        for (int j = 0; j < GB_MAXV; ++j) {                 // the row's loads first
            const int v = tc + 256 * j;
            if (v < nv) {
                const size_t o = (size_t)row * H + v * 8;
                g[j] = *reinterpret_cast<const bf16x8*>(dact + o);
                x[j] = *reinterpret_cast<const bf16x8*>(pre + o);
            }
        }
#pragma unroll
        for (int j = 0; j < GB_MAXV; ++j) {
            const int v = tc + 256 * j;
            if (v < nv) {
                bf16x8 out;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    out[e] = (bf16)((float)g[j][e] * gelu_grad_exp_f((float)x[j][e]));
                    acc[j][e] += (float)out[e];             // the sum of what is stored (what a column sum over d(pre) would see)
                }
                *reinterpret_cast<bf16x8*>(dpre + (size_t)row * H + v * 8) = out;
            }
        }
    }
    // the phases' sums meet in LDS, one vector slot at a time, in phase order; phase 0 writes the block's partial row
    float* pb = part + (size_t)blockIdx.x * H;
#pragma unroll
    for (int j = 0; j < GB_MAXV; ++j) {
        if (256 * j >= nv) break;                           // uniform
        __syncthreads();
        if (ph > 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[ph - 1][tc][e] = acc[j][e];
        }
        __syncthreads();
        const int v = tc + 256 * j;
        if (ph == 0 && v < nv) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                t[e] = acc[j][e];
#pragma unroll
                for (int q = 0; q < GB_PH - 1; ++q) t[e] += red[q][tc][e];
            }
            *reinterpret_cast<f32x4*>(pb + v * 8) = f32x4{t[0], t[1], t[2], t[3]};
            *reinterpret_cast<f32x4*>(pb + v * 8 + 4) = f32x4{t[4], t[5], t[6], t[7]};
        }
    }
}
// 1024 threads = 32 columns x 32 row groups over the blocks' partial rows (<= GB_BLOCKS), fixed order
__global__ __launch_bounds__(1024) void plm_colpart_reduce_kernel(const float* __restrict__ part, int nblocks, int H, bf16* __restrict__ out) {
    __shared__ float red[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    float a = 0.f;
    if (c < H) {
        float v[GB_BLOCKS / 32];
#pragma unroll
        for (int u = 0; u < GB_BLOCKS / 32; ++u) {
            const int b = ty + 32 * u;
            v[u] = b < nblocks ? part[(size_t)b * H + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < GB_BLOCKS / 32; ++u) a += v[u];
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && c < H) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += red[g][tx];
        out[c] = (bf16)t;
    }
}
int gb_blocks(int M) { const int b = (M + GB_PH - 1) / GB_PH; return b < GB_BLOCKS ? b : GB_BLOCKS; }

// Weight gradient of an nn.Embedding of the text encoder (transformers' *Embeddings: word / position / token-type tables, src/models.py:75-91) for a few
// thousand tokens: dW[id] = sum over the tokens t with ids[t] == id of dy[t], tokens of padding_idx skipped.  torch's kernel for <= 3072 indices
// (embedding_backward_feature_kernel) walks the indices in ballot-matched batches per feature block: 110-170 us per table at 2048 tokens x 1024, alone at
// the very end of the text encoder's backward.  Here, deterministic (no atomics on the output, token order):
//   * tables of more than EB_SMALL rows: one workgroup per token.  It marks the tokens that share its id in an LDS bitmap; the FIRST of them owns the row and
//     adds the others' rows in token order (fp32, one rounding), four loads in flight.  dW is zeroed first (the rows no token names).
//   * tables of at most EB_SMALL rows (token types: one or two ids for every token): per row a predicated column sum over all tokens (the layout of
//     colsum_kernel), every row written.
// C % 8 == 0, C <= 2048; dy bf16 [T][C]; ids int64 [T].
constexpr int EB_SMALL = 8;
__global__ __launch_bounds__(256) void embedding_bwd_kernel(int T, int C, const long long* __restrict__ ids, long long padding_idx, const bf16* __restrict__ dy,
                                                           bf16* __restrict__ dw) {
    extern __shared__ unsigned eb_lds[];                    // [nwords] bitmap, then the ordered list of the duplicates (T ints)
    const int nwords = (T + 31) >> 5;
    unsigned* bits = eb_lds;
    int* list = reinterpret_cast<int*>(eb_lds + nwords);
    __shared__ int count;
    const int t = blockIdx.x;
    const long long my = ids[t];
    if (my == padding_idx) return;                          // uniform
    for (int w = threadIdx.x; w < nwords; w += 256) bits[w] = 0u;
    __syncthreads();
    int earlier = 0;
    for (int j = threadIdx.x; j < T; j += 256)
        if (ids[j] == my) {
            if (j < t) earlier = 1;
            atomicOr(&bits[j >> 5], 1u << (j & 31));
        }
    if (__syncthreads_or(earlier)) return;                  // an earlier token owns the row
    if (threadIdx.x == 0) {                                 // ascending token order
        int n = 0;
        for (int w = t >> 5; w < nwords; ++w) {
            unsigned word = bits[w];
            while (word) {
                const int b = __ffs(word) - 1;
                word &= word - 1;
                list[n++] = w * 32 + b;
            }
        }
        count = n;
    }
    __syncthreads();
    const int n = count, v = threadIdx.x;
    if (v * 8 >= C) return;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    int k = 0;
    for (; k + 3 < n; k += 4) {
        bf16x8 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const bf16x8*>(dy + (size_t)list[k + u] * C + v * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)r[u][e];
    }
    for (; k < n; ++k) {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(dy + (size_t)list[k] * C + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)r[e];
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
    *reinterpret_cast<bf16x8*>(dw + (size_t)my * C + v * 8) = o;
}
// grid (C / 32, rows): 4 column chunks (8 columns) x 64 token lanes; a lane strides over the tokens and adds the rows whose id is this row
__global__ __launch_bounds__(256) void embedding_bwd_small_kernel(int T, int C, const long long* __restrict__ ids, long long padding_idx, const bf16* __restrict__ dy,
                                                                 bf16* __restrict__ dw) {
    __shared__ float red[64][33];
    const int c = threadIdx.x & 3, r = threadIdx.x >> 2;
    const int col = (blockIdx.x * 4 + c) * 8;
    const long long row = blockIdx.y;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (col < C && row != padding_idx)
        for (int m = r; m < T; m += 64)
            if (ids[m] == row) {
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(dy + (size_t)m * C + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)x[e];
            }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[r][c * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 32) {
        const int oc = blockIdx.x * 32 + threadIdx.x;
        if (oc < C) {
            float t = 0.f;
#pragma unroll 8
            for (int i = 0; i < 64; ++i) t += red[i][threadIdx.x];
            dw[(size_t)row * C + oc] = (bf16)t;
        }
    }
}

int pf_blocks(int M) { return M / PF_BW < 1 ? 1 : (M / PF_BW > 256 ? 256 : M / PF_BW); }      // backward: workgroups = partial rows
int pf_blocks_fwd(int M) { return M / 4 < 1 ? 1 : (M / 4 > 1024 ? 1024 : M / 4); }
bool pf_misaligned(const void* a, const void* b, const void* c, const void* d) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d)) & 15) != 0;
}

}  // namespace

extern "C" int fmmt_dropadd_ln_fwd(int param_dtype, int M, int C, float eps, const void* h, const void* res, const void* gamma, const void* beta, float p,
                                   uint64_t seed, const uint64_t* seed_dev, uint64_t salt, void* xsum, void* y, void* stream) {
    if (M <= 0 || C <= 0 || C % 8 || C > 2048 || !(p >= 0.f && p < 1.f)) return FMMT_EINVAL;
    if (param_dtype != FMMT_BF16 && param_dtype != FMMT_F32) return FMMT_EINVAL;
    if (!h || !res || !gamma || !beta || !xsum || !y) return FMMT_EINVAL;
    if (pf_misaligned(h, res, xsum, y) || pf_misaligned(gamma, beta, gamma, beta)) return FMMT_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (param_dtype == FMMT_BF16)
        hipLaunchKernelGGL(plm_dropadd_ln_fwd_kernel<bf16>, dim3(pf_blocks_fwd(M)), dim3(256), 0, st, M, C, eps, (const bf16*)h, (const bf16*)res, (const bf16*)gamma,
                           (const bf16*)beta, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, (unsigned long long)salt, (bf16*)xsum, (bf16*)y);
    else
        hipLaunchKernelGGL(plm_dropadd_ln_fwd_kernel<float>, dim3(pf_blocks_fwd(M)), dim3(256), 0, st, M, C, eps, (const bf16*)h, (const bf16*)res, (const float*)gamma,
                           (const float*)beta, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, (unsigned long long)salt, (bf16*)xsum, (bf16*)y);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t fmmt_dropadd_ln_bwd_workspace(int M, int C) {
    if (M <= 0 || C <= 0) return 0;
    return (size_t)pf_blocks(M) * 3 * (size_t)C * sizeof(float);
}

extern "C" int fmmt_dropadd_ln_bwd(int param_dtype, int M, int C, float eps, const void* dy, const void* xsum, const void* gamma, float p, uint64_t seed,
                                   const uint64_t* seed_dev, uint64_t salt, void* dx, void* dh, void* dgamma, void* dbeta, void* dbias, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (M <= 0 || C <= 0 || C % 8 || C > 2048 || !(p >= 0.f && p < 1.f)) return FMMT_EINVAL;
    if (param_dtype != FMMT_BF16 && param_dtype != FMMT_F32) return FMMT_EINVAL;
    if (!dy || !xsum || !gamma || !dx || !dh || !dgamma || !dbeta || !workspace) return FMMT_EINVAL;
    if (workspace_bytes < fmmt_dropadd_ln_bwd_workspace(M, C)) return FMMT_EWORKSPACE;
    if (pf_misaligned(dy, xsum, dx, dh) || pf_misaligned(gamma, gamma, gamma, gamma)) return FMMT_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int blocks = pf_blocks(M);
    float* part = reinterpret_cast<float*>(workspace);
    if (param_dtype == FMMT_BF16) {
        hipLaunchKernelGGL(plm_dropadd_ln_bwd_kernel<bf16>, dim3(blocks), dim3(64 * PF_BW), 0, st, M, C, eps, (const bf16*)dy, (const bf16*)xsum, (const bf16*)gamma, p,
                           (unsigned long long)seed, (const unsigned long long*)seed_dev, (unsigned long long)salt, (bf16*)dx, (bf16*)dh, part);
        FMMT_CHECK_LAUNCH();
        hipLaunchKernelGGL(plm_dropadd_ln_reduce_kernel<bf16>, dim3((3 * C + 31) / 32), dim3(1024), 0, st, (const float*)part, blocks, C, (bf16*)dgamma, (bf16*)dbeta,
                           (bf16*)dbias);
    } else {
        hipLaunchKernelGGL(plm_dropadd_ln_bwd_kernel<float>, dim3(blocks), dim3(64 * PF_BW), 0, st, M, C, eps, (const bf16*)dy, (const bf16*)xsum, (const float*)gamma, p,
                           (unsigned long long)seed, (const unsigned long long*)seed_dev, (unsigned long long)salt, (bf16*)dx, (bf16*)dh, part);
        FMMT_CHECK_LAUNCH();
        hipLaunchKernelGGL(plm_dropadd_ln_reduce_kernel<float>, dim3((3 * C + 31) / 32), dim3(1024), 0, st, (const float*)part, blocks, C, (float*)dgamma,
                           (float*)dbeta, (float*)dbias);
    }
    FMMT_CHECK_LAUNCH();
    return 0;
}

// the text encoder's form: bf16 affine parameters
extern "C" int fmmt_plm_dropadd_ln_fwd(int M, int C, float eps, const void* h, const void* res, const void* gamma, const void* beta, float p,
                                       uint64_t seed, const uint64_t* seed_dev, uint64_t salt, void* xsum, void* y, void* stream) {
    return fmmt_dropadd_ln_fwd(FMMT_BF16, M, C, eps, h, res, gamma, beta, p, seed, seed_dev, salt, xsum, y, stream);
}
extern "C" size_t fmmt_plm_dropadd_ln_bwd_workspace(int M, int C) { return fmmt_dropadd_ln_bwd_workspace(M, C); }
extern "C" int fmmt_plm_dropadd_ln_bwd(int M, int C, float eps, const void* dy, const void* xsum, const void* gamma, float p, uint64_t seed,
                                       const uint64_t* seed_dev, uint64_t salt, void* dx, void* dh, void* dgamma, void* dbeta, void* dbias, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return fmmt_dropadd_ln_bwd(FMMT_BF16, M, C, eps, dy, xsum, gamma, p, seed, seed_dev, salt, dx, dh, dgamma, dbeta, dbias, workspace, workspace_bytes, stream);
}

extern "C" size_t fmmt_plm_gelu_bwd_colsum_workspace(int M, int H) {
    if (M <= 0 || H <= 0) return 0;
    return (size_t)gb_blocks(M) * (size_t)H * sizeof(float);
}
extern "C" int fmmt_plm_gelu_bwd_colsum(int M, int H, const void* dact, const void* pre, void* dpre, void* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    if (M <= 0 || H <= 0 || H % 8 || H > 256 * 8 * GB_MAXV) return FMMT_EINVAL;
    if (!dact || !pre || !dpre || !dbias || !workspace) return FMMT_EINVAL;
    if (workspace_bytes < fmmt_plm_gelu_bwd_colsum_workspace(M, H)) return FMMT_EWORKSPACE;
    if (pf_misaligned(dact, pre, dpre, workspace)) return FMMT_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int blocks = gb_blocks(M);
    hipLaunchKernelGGL(plm_gelu_bwd_colsum_kernel, dim3(blocks), dim3(256 * GB_PH), 0, st, M, H, (const bf16*)dact, (const bf16*)pre, (bf16*)dpre, (float*)workspace);
    FMMT_CHECK_LAUNCH();
    hipLaunchKernelGGL(plm_colpart_reduce_kernel, dim3((H + 31) / 32), dim3(1024), 0, st, (const float*)workspace, blocks, H, (bf16*)dbias);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_embedding_bwd(int T, int C, int V, const int64_t* ids, int64_t padding_idx, const void* dy, void* dweight, void* stream) {
    if (T <= 0 || C <= 0 || V <= 0 || C % 8 || C > 2048 || T > 32768 || !ids || !dy || !dweight) return FMMT_EINVAL;
    if (pf_misaligned(dy, dweight, dy, dweight)) return FMMT_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (V <= EB_SMALL) {
        hipLaunchKernelGGL(embedding_bwd_small_kernel, dim3((C + 31) / 32, V), dim3(256), 0, st, T, C, (const long long*)ids, (long long)padding_idx, (const bf16*)dy,
                           (bf16*)dweight);
        FMMT_CHECK_LAUNCH();
        return 0;
    }
    if (hipError_t e = hipMemsetAsync(dweight, 0, (size_t)V * C * sizeof(bf16), st); e != hipSuccess) return (int)e;
    const size_t lds = (size_t)((T + 31) / 32) * 4 + (size_t)T * 4;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(T), dim3(256), lds, st, T, C, (const long long*)ids, (long long)padding_idx, (const bf16*)dy, (bf16*)dweight);
    FMMT_CHECK_LAUNCH();
    return 0;
}
