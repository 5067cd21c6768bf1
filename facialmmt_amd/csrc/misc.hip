// Small streaming kernels of the hot path: patch im2col / col2im (PatchEmbed's 4x4 stride-4 conv is
// run as a GEMM on the gathered patches), BatchNorm1d of the embedding head, the cross-modal input
// embedding (scale + sinusoidal position with the zero-is-padding rule) and an elementwise scale.
#include "fmmt_common.h"
#include "../../include/fmmt.h"

namespace {

constexpr int IMG = 224, PS = 4, GRID = 56, KPATCH = 48;

// thread <-> (n, c, y, px): reads 4 contiguous pixels of one image row, moves them to/from
// cols[(n*3136 + (y/4)*56 + px)][c*16 + (y%4)*4 .. +4]
template <typename T, bool TO_COLS>
__global__ void patch_cols_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int px = (int)(t % GRID);
    size_t r = t / GRID;
    const int y = (int)(r % IMG);
    r /= IMG;
    const int c = (int)(r % 3);
    const size_t n = r / 3;
    const size_t img_off = ((n * 3 + c) * IMG + y) * IMG + (size_t)px * PS;
    const size_t col_off = ((n * GRID + (y >> 2)) * GRID + px) * KPATCH + c * 16 + (y & 3) * PS;
    const T* s = TO_COLS ? src + img_off : src + col_off;
    T* d = TO_COLS ? dst + col_off : dst + img_off;
    if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(d) = *reinterpret_cast<const uint2*>(s);
    else *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
}

// BatchNorm1d over (n, C) rows: a workgroup of 1024 threads = 64 columns x 16 row groups; every column sum is sixteen per-group partial
// sums (rows g, g + 16, ...) added in group order through LDS -- fixed order, two passes (mean, then centred squares) as before.
// (One thread per column walking all n rows serially, the round-1 form, took 295 us forward and 250 us backward for the 640 x 512 head
//  of the bench step: three dependent passes of 640 loads.)
constexpr int BN_COLS = 64, BN_GROUPS = 16;

__device__ __forceinline__ float bn_colsum(float v, float (*red)[BN_COLS], int tc, int tg) {
    __syncthreads();                                        // the previous use of `red` is over
    red[tg][tc] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < BN_GROUPS; ++g) s += red[g][tc];
    return s;
}

template <typename T>
__global__ __launch_bounds__(BN_COLS * BN_GROUPS) void bn1d_fwd_kernel(int n, int C, const T* __restrict__ x, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* running_mean, float* running_var,
                                float momentum, float eps, int training, T* __restrict__ y,
                                float* save_mean, float* save_invstd) {
    __shared__ float red[BN_GROUPS][BN_COLS];
    const int tc = threadIdx.x % BN_COLS, tg = threadIdx.x / BN_COLS;
    const int c = blockIdx.x * BN_COLS + tc;
    const bool ok = c < C;
    float mean, invstd;
    if (training) {
        float s = 0.f;
        if (ok)
            for (int r = tg; r < n; r += BN_GROUPS) s += to_f32(x[(size_t)r * C + c]);
        mean = bn_colsum(s, red, tc, tg) / n;
        float q = 0.f;
        if (ok)
            for (int r = tg; r < n; r += BN_GROUPS) {
                const float d = to_f32(x[(size_t)r * C + c]) - mean;
                q += d * d;
            }
        q = bn_colsum(q, red, tc, tg);
        const float var = q / n;
        invstd = rsqrtf(var + eps);
        if (ok && tg == 0) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1 ? q / (n - 1) : var);
        }
    } else {
        mean = ok ? running_mean[c] : 0.f;
        invstd = ok ? rsqrtf(running_var[c] + eps) : 0.f;
    }
    if (!ok) return;
    if (tg == 0) {
        if (save_mean) save_mean[c] = mean;
        if (save_invstd) save_invstd[c] = invstd;
    }
    const float g = gamma[c] * invstd, b = beta[c] - mean * g;
    for (int r = tg; r < n; r += BN_GROUPS) y[(size_t)r * C + c] = from_f32<T>(to_f32(x[(size_t)r * C + c]) * g + b);
}

template <typename T>
__global__ __launch_bounds__(BN_COLS * BN_GROUPS) void bn1d_bwd_kernel(int n, int C, const T* __restrict__ dy, const T* __restrict__ x,
                                const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                const float* __restrict__ save_invstd, int training, T* __restrict__ dx,
                                float* dgamma, float* dbeta) {
    __shared__ float red[BN_GROUPS][BN_COLS];
    const int tc = threadIdx.x % BN_COLS, tg = threadIdx.x / BN_COLS;
    const int c = blockIdx.x * BN_COLS + tc;
    const bool ok = c < C;
    const float mean = ok ? save_mean[c] : 0.f, invstd = ok ? save_invstd[c] : 0.f;
    float sb = 0.f, sg = 0.f;
    if (ok)
        for (int r = tg; r < n; r += BN_GROUPS) {
            const float g = to_f32(dy[(size_t)r * C + c]);
            sb += g;
            sg += g * (to_f32(x[(size_t)r * C + c]) - mean) * invstd;
        }
    sb = bn_colsum(sb, red, tc, tg);
    sg = bn_colsum(sg, red, tc, tg);
    if (!ok) return;
    if (tg == 0) {
        if (dgamma) dgamma[c] = sg;
        if (dbeta) dbeta[c] = sb;
    }
    const float k = gamma[c] * invstd;
    if (training) {
        const float inv_n = 1.f / n;
        for (int r = tg; r < n; r += BN_GROUPS) {
            const float xh = (to_f32(x[(size_t)r * C + c]) - mean) * invstd;
            dx[(size_t)r * C + c] = from_f32<T>(k * (to_f32(dy[(size_t)r * C + c]) - sb * inv_n - xh * sg * inv_n));
        }
    } else {
        for (int r = tg; r < n; r += BN_GROUPS) dx[(size_t)r * C + c] = from_f32<T>(k * to_f32(dy[(size_t)r * C + c]));
    }
}

template <typename T>
__global__ void posemb_scale_kernel(int L, int B, int E, const T* __restrict__ x, const float* __restrict__ table,
                                    float scale, T* __restrict__ y) {
    constexpr int VEC = Vec<T>::N;
    const int chunks = E / VEC;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)L * B * chunks) return;
    const int c = (int)(t % chunks);
    const size_t row = t / chunks;                  // = time * B + b
    const int time = (int)(row / B);
    const int pos = (to_f32(x[row * E]) != 0.f) ? time + 1 : 0;
    const Vec<T> v = ldvec<T>(x + row * E + c * VEC);
    Vec<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, scale * v.get(e) + table[(size_t)pos * E + c * VEC + e]);
    stvec<T>(y + row * E + c * VEC, o);
}

template <typename T>
__global__ void scale_kernel(size_t nvec, const T* __restrict__ x, float alpha, T* __restrict__ y) {
    constexpr int VEC = Vec<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const Vec<T> v = ldvec<T>(x + i * VEC);
        Vec<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, alpha * v.get(e));
        stvec<T>(y + i * VEC, o);
    }
}

bool dt_ok(int dtype) { return dtype == FMMT_BF16 || dtype == FMMT_F32; }


// Column sums of a [M][N] matrix (bias gradient of a Linear whose GEMMs are the vendor library's: the text encoder).
// block = 4 column chunks (8 columns, 16 bytes each) x 64 row lanes: a lane strides over the rows with 16-byte loads, the 64
// partial sums of a column meet in LDS.  One launch, no atomics, fixed summation order.
template <typename T, typename TO>
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const T* __restrict__ x, int ldx, TO* __restrict__ out) {
    constexpr int V = 16 / sizeof(T);                       // columns per chunk
    __shared__ float red[64][4 * V + 1];
    const int c = threadIdx.x & 3, r = threadIdx.x >> 2;
    const int col = (blockIdx.x * 4 + c) * V;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    if (col < N) {
        const T* px = x + col;
        int m = r;
#pragma unroll 1
        for (; m + 192 < M; m += 256) {                      // four loads in flight per lane
            T v[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(v[u]) = *reinterpret_cast<const uint4*>(px + (size_t)(m + 64 * u) * ldx);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += (float)v[u][e];
        }
        for (; m < M; m += 64) {
            T v[V];
            *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(px + (size_t)m * ldx);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) red[r][c * V + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 4 * V) {
        const int oc = blockIdx.x * 4 * V + threadIdx.x;
        if (oc < N) {
            float t = 0.f;
#pragma unroll 8
            for (int i = 0; i < 64; ++i) t += red[i][threadIdx.x];
            out[oc] = (TO)t;
        }
    }
}

// Batched cast (+ transpose) of parameter matrices into their bf16 shadows: ONE launch refreshes every weight shadow a training
// step uses (the per-weight casts and transposes were ~360 launches of 4-9 us inside the step's graph).  One 64 x 64 tile per
// block; the block finds its matrix by bisection over the tile offsets.
struct CastDesc {
    const void* src;    // [rows][src_ld] fp32 or bf16
    void* dst;          // bf16: [rows][cols], or [cols][rows] when transposing
    int rows, cols, src_ld, flags;      // flags: bit 0 transpose, bit 1 source is fp32
    int tile_begin, tiles_c;
};
__global__ __launch_bounds__(256) void cast_batch_kernel(const CastDesc* __restrict__ desc, int n) {
    __shared__ bf16 tile[64][66];
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].tile_begin <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const CastDesc d = desc[lo];
    const int t = (int)blockIdx.x - d.tile_begin, tr = t / d.tiles_c, tc = t - tr * d.tiles_c;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const bool f32 = d.flags & 2, tp = d.flags & 1;
    const float* sf = reinterpret_cast<const float*>(d.src);
    const bf16* sb = reinterpret_cast<const bf16*>(d.src);
    bf16* dst = reinterpret_cast<bf16*>(d.dst);
    // Whole fp32 tiles of 16-byte-aligned matrices (every Linear weight of the models: rows and columns are multiples of 4): 16-byte loads,
    // a wave reads four rows x 256 contiguous bytes, and 8-byte stores -- row-contiguous for the plain shadow, 4 consecutive source rows of one
    // source column for the transposed one.  (The element-wise form below moved the step's 127 M shadowed weights at ~1.5 TB/s: 1 ms at the head
    // of every step.)  Ragged edge tiles, bf16 sources and unaligned views keep the element-wise form.
    const int r0 = tr * 64, c0 = tc * 64;
    const bool vec = f32 && r0 + 64 <= d.rows && c0 + 64 <= d.cols && (d.src_ld & 3) == 0 && (d.cols & 3) == 0 && (d.rows & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(d.src) | reinterpret_cast<uintptr_t>(d.dst)) & 15) == 0;      // block-uniform
    if (vec) {
        const int rr = threadIdx.x >> 4, cg = (threadIdx.x & 15) * 4;
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(sf + (size_t)(r0 + rr + 16 * i) * d.src_ld + c0 + cg);
        if (!tp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)v[i][e];
                *reinterpret_cast<bf16x4*>(dst + (size_t)(r0 + rr + 16 * i) * d.cols + c0 + cg) = o;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[rr + 16 * i][cg + e] = (bf16)v[i][e];
        __syncthreads();
        // destination row = source column c0 + oc, destination columns = source rows r0 + og .. og + 3
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oc = (threadIdx.x >> 4) + 16 * i, og = (threadIdx.x & 15) * 4;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = tile[og + e][oc];
            *reinterpret_cast<bf16x4*>(dst + (size_t)(c0 + oc) * d.rows + r0 + og) = o;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = tr * 64 + ty + 16 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = tc * 64 + tx + 16 * j;
            if (r < d.rows && c < d.cols) {
                const size_t o = (size_t)r * d.src_ld + c;
                const bf16 v = f32 ? (bf16)sf[o] : sb[o];
                if (tp) tile[ty + 16 * i][tx + 16 * j] = v;
                else dst[(size_t)r * d.cols + c] = v;
            }
        }
    }
    if (!tp) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tc * 64 + ty + 16 * i;                 // source column = destination row
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = tr * 64 + tx + 16 * j;             // source row = destination column
            if (r < d.rows && c < d.cols) dst[(size_t)c * d.rows + r] = tile[tx + 16 * j][ty + 16 * i];
        }
    }
}

// Clip + AdamW + low-precision shadow in ONE launch over every tensor the step's optimizer updates (train.py:135-143:
// clip_grad_norm_ -> optimizer.step()).  Two flavours of the update, selected by `hf`:
//   hf = 1  transformers.AdamW, the optimizer the reference constructs (train.py:307,333; defaults eps 1e-6, weight_decay 0):
//           m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;  p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)
//           -- eps is added BEFORE the bias correction -- and the decoupled decay is applied AFTER the update: p += p * (-lr wd)
//   hf = 0  torch.optim.AdamW: p *= 1 - lr wd first, then p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//   coef = min(1, max_norm / (total_norm + 1e-6)), g' = coef g   (torch.nn.utils.clip_grad_norm_);  low = bf16(p) for bf16 twins
// lr, t and total_norm are device scalars (graph replays), the gradients are read once and NOT written back: against
// clip_grad_norm_ + fused AdamW + the multi-tensor re-rounding of the bf16 text encoder this saves the scaling pass over the
// gradients (read + write) and one pass over the fp32 parameters.
// Gradient hand-over + the clip norm in one pass: every fresh gradient of the step (fp32 or bf16, wherever autograd allocated it) is
// written -- or, with accumulation, added -- into its slot of the flat fp32 buffers the optimizer and the all-reduce work on, and the sum
// of squares of what the slot then holds is formed on the way (per-block partial sums, finished in fixed order by ONE block).  Replaces
// a multi-tensor copy (23 launches) and a multi-tensor L2 norm (22 launches + 8 small ones) over the same 435 M values: 1.3 ms of kernel
// time in the step's serial tail, where nothing else runs.
struct HoDesc {
    const void* src; float* dst;
    long long n;
    int blk_begin, flags;                                       // flags bit 0: src is bf16 (else fp32); bit 1: dst += src (else dst = src)
};
__global__ __launch_bounds__(256) void grad_handover_kernel(const HoDesc* __restrict__ desc, int n_desc, float* __restrict__ partial) {
    __shared__ float red[4];
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].blk_begin <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const HoDesc d = desc[lo];
    const long long base = (long long)((int)blockIdx.x - d.blk_begin) * 4096;
    const bool bf = d.flags & 1, acc = d.flags & 2;
    const float* sf = reinterpret_cast<const float*>(d.src);
    const bf16* sb = reinterpret_cast<const bf16*>(d.src);
    float ss = 0.f;
    const bool vec = base + 4096 <= d.n && ((reinterpret_cast<uintptr_t>(d.src) | reinterpret_cast<uintptr_t>(d.dst)) & 15) == 0;   // block-uniform
    if (vec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long o = base + (long long)(i * 256 + threadIdx.x) * 4;
            f32x4 v;
            if (bf) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(sb + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (float)t[e];
            } else {
                v = *reinterpret_cast<const f32x4*>(sf + o);
            }
            if (acc) v += *reinterpret_cast<const f32x4*>(d.dst + o);
            *reinterpret_cast<f32x4*>(d.dst + o) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += v[e] * v[e];
        }
    } else {
        for (long long o = base + threadIdx.x; o < base + 4096 && o < d.n; o += 256) {
            float v = bf ? (float)sb[o] : sf[o];
            if (acc) v += d.dst[o];
            d.dst[o] = v;
            ss += v * v;
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// norm = sqrt(sum of the per-block partial sums): one block, fixed order (strided per-thread sums, then a tree over the 1024 threads)
__global__ __launch_bounds__(1024) void sumsq_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float red[1024];
    float a = 0.f;
    int i = threadIdx.x;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {               // eight loads in flight, added in the rolled loop's order (435 M gradients: 104 partials per thread)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; i < n; i += 1024) a += partial[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sqrtf(red[0]);
}

struct AdamDesc {
    float* p; const void* g; float* m; float* v; bf16* low;     // low may be null
    long long n;
    int blk_begin, g_bf16;                                      // g_bf16: the gradient is bf16 (the twin's own .grad), else fp32
};
__global__ __launch_bounds__(256) void adamw_batch_kernel(const AdamDesc* __restrict__ desc, int n_desc, const float* __restrict__ lr_p,
                                                          const float* __restrict__ step_p, const float* __restrict__ norm_p,
                                                          float beta1, float beta2, float eps, float wd, float max_norm, int hf) {
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].blk_begin <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const AdamDesc d = desc[lo];
    const float lr = *lr_p, t = *step_p;
    const float coef = norm_p ? fminf(1.0f, max_norm / (*norm_p + 1e-6f)) : 1.0f;
    const float bc1 = 1.0f - powf(beta1, t), bc2s = sqrtf(1.0f - powf(beta2, t));
    const float step_size = hf ? lr * bc2s / bc1 : lr / bc1, decay = hf ? 1.0f : 1.0f - lr * wd;
    const float den_div = hf ? 1.0f : bc2s, post = hf ? -lr * wd : 0.0f;
    const long long base = (long long)((int)blockIdx.x - d.blk_begin) * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long o = base + (long long)(i * 256 + threadIdx.x) * 4;
        if (o >= d.n) break;
        const float* gf = reinterpret_cast<const float*>(d.g);
        const bf16* gb = reinterpret_cast<const bf16*>(d.g);
        const bool g_ok = d.g_bf16 ? (reinterpret_cast<uintptr_t>(gb + o) & 7) == 0 : (reinterpret_cast<uintptr_t>(gf + o) & 15) == 0;
        if (o + 4 <= d.n && g_ok && ((reinterpret_cast<uintptr_t>(d.p + o) | reinterpret_cast<uintptr_t>(d.m + o) |
                                       reinterpret_cast<uintptr_t>(d.v + o)) & 15) == 0) {
            f32x4 p = *reinterpret_cast<const f32x4*>(d.p + o), g;
            if (d.g_bf16) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(gb + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = (float)t[e];
            } else {
                g = *reinterpret_cast<const f32x4*>(gf + o);
            }
            f32x4 m = *reinterpret_cast<const f32x4*>(d.m + o), v = *reinterpret_cast<const f32x4*>(d.v + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = g[e] * coef;
                p[e] *= decay;
                m[e] = beta1 * m[e] + (1.0f - beta1) * ge;
                v[e] = beta2 * v[e] + (1.0f - beta2) * ge * ge;
                p[e] -= step_size * (m[e] / (sqrtf(v[e]) / den_div + eps));
                p[e] = fmaf(p[e], post, p[e]);
            }
            *reinterpret_cast<f32x4*>(d.p + o) = p;
            *reinterpret_cast<f32x4*>(d.m + o) = m;
            *reinterpret_cast<f32x4*>(d.v + o) = v;
            if (d.low) {
                if ((reinterpret_cast<uintptr_t>(d.low + o) & 7) == 0) {
                    bf16x4 l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) l[e] = (bf16)p[e];
                    *reinterpret_cast<bf16x4*>(d.low + o) = l;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) d.low[o + e] = (bf16)p[e];
                }
            }
        } else {
            for (long long j = o; j < d.n && j < o + 4; ++j) {
                const float ge = (d.g_bf16 ? (float)gb[j] : gf[j]) * coef;
                float p = d.p[j] * decay;
                const float m = beta1 * d.m[j] + (1.0f - beta1) * ge;
                const float v = beta2 * d.v[j] + (1.0f - beta2) * ge * ge;
                p -= step_size * (m / (sqrtf(v) / den_div + eps));
                p = fmaf(p, post, p);
                d.p[j] = p;
                d.m[j] = m;
                d.v[j] = v;
                if (d.low) d.low[j] = (bf16)p;
            }
        }
    }
}

// LayerNorm backward for a bf16 module with bf16 affine parameters (the text encoder's 49 LayerNorms: torch's own backward is
// three launches -- input gradient, partial and final gamma / beta sums: 39 us in the step).  One wave per token row, rows
// grid-strided; the row statistics are recomputed from x in fp32 (the forward is torch's layer_norm, nothing of it is saved but
// x); per-block partial sums of d gamma / d beta, finished by lnp_reduce_kernel in a fixed order, written in bf16.
__global__ __launch_bounds__(256) void lnp_bwd_kernel(int M, int C, float eps, const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                      const bf16* __restrict__ gamma, bf16* __restrict__ dx, float* __restrict__ part) {
    constexpr int MAXV = 4;                                 // 8-element vectors per lane: C <= 2048
    __shared__ float red[4][2][64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C / 8;                                   // vectors per row
    float dg[MAXV][8], db[MAXV][8];
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dg[i][e] = db[i][e] = 0.f;
    const float invC = 1.0f / (float)C;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        bf16x8 xv[MAXV], gv[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
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
        for (int i = 0; i < MAXV; ++i)
            if (lane + 64 * i < nv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)xv[i][e] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(q) * invC + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                const bf16x8 gm = *reinterpret_cast<const bf16x8*>(gamma + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)xv[i][e] - mean) * rstd, dyv = (float)gv[i][e], g = dyv * (float)gm[e];
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
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                const bf16x8 gm = *reinterpret_cast<const bf16x8*>(gamma + v * 8);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)xv[i][e] - mean) * rstd, g = (float)gv[i][e] * (float)gm[e];
                    o[e] = (bf16)(rstd * (g - s1 - xh * s2));
                }
                *reinterpret_cast<bf16x8*>(dx + (size_t)row * C + v * 8) = o;
            }
        }
    }
    // the four waves of the block, one 64-vector slab at a time: partial [block][2][C]
    float* pb = part + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (64 * i >= nv) break;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[wave][0][lane * 8 + e] = dg[i][e];
            red[wave][1][lane * 8 + e] = db[i][e];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * 512; t += 256) {
            const int which = t >> 9, col = t & 511, ch = i * 512 + col;
            if (ch < C) pb[which * C + ch] = (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
        }
    }
}
__global__ __launch_bounds__(1024) void lnp_reduce_kernel(const float* __restrict__ part, int nblocks, int C, bf16* __restrict__ dgamma, bf16* __restrict__ dbeta) {
    // 1024 threads = 32 columns x 32 partial groups, fixed-order tree (256 threads x one column each was 21 us: eight workgroups)
    __shared__ float red[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;                     // column of the [2][C] pair
    float a = 0.f;
    if (i < 2 * C)
        for (int b = ty; b < nblocks; b += 32) a += part[(size_t)b * 2 * C + i];
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && i < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += red[g][tx];
        if (i < C) dgamma[i] = (bf16)t;
        else dbeta[i - C] = (bf16)t;
    }
}

}  // namespace

extern "C" int fmmt_version(void) { return 3; }     // = the round whose ABI this is (round 3: fmmt_window_block_fwd, hf_semantics of fmmt_adamw_batch)

extern "C" int fmmt_patch_im2col(int dtype, int n_img, const void* img, void* cols, void* stream) {
    if (!dt_ok(dtype) || n_img <= 0) return FMMT_EINVAL;
    const size_t total = (size_t)n_img * 3 * IMG * GRID;
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FMMT_BF16) hipLaunchKernelGGL((patch_cols_kernel<bf16, true>), grid, dim3(256), 0, st, (const bf16*)img, (bf16*)cols, total);
    else hipLaunchKernelGGL((patch_cols_kernel<float, true>), grid, dim3(256), 0, st, (const float*)img, (float*)cols, total);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_patch_col2im(int dtype, int n_img, const void* cols, void* dimg, void* stream) {
    if (!dt_ok(dtype) || n_img <= 0) return FMMT_EINVAL;
    const size_t total = (size_t)n_img * 3 * IMG * GRID;
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FMMT_BF16) hipLaunchKernelGGL((patch_cols_kernel<bf16, false>), grid, dim3(256), 0, st, (const bf16*)cols, (bf16*)dimg, total);
    else hipLaunchKernelGGL((patch_cols_kernel<float, false>), grid, dim3(256), 0, st, (const float*)cols, (float*)dimg, total);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_batchnorm1d_fwd(int dtype, int n, int C, const void* x, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps,
                                    int training, void* y, float* save_mean, float* save_invstd, void* stream) {
    if (!dt_ok(dtype) || n <= 0 || C <= 0) return FMMT_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((C + BN_COLS - 1) / BN_COLS);
    if (dtype == FMMT_BF16)
        hipLaunchKernelGGL(bn1d_fwd_kernel<bf16>, grid, dim3(BN_COLS * BN_GROUPS), 0, st, n, C, (const bf16*)x, gamma, beta, running_mean,
                           running_var, momentum, eps, training, (bf16*)y, save_mean, save_invstd);
    else
        hipLaunchKernelGGL(bn1d_fwd_kernel<float>, grid, dim3(BN_COLS * BN_GROUPS), 0, st, n, C, (const float*)x, gamma, beta, running_mean,
                           running_var, momentum, eps, training, (float*)y, save_mean, save_invstd);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_batchnorm1d_bwd(int dtype, int n, int C, const void* dy, const void* x, const float* gamma,
                                    const float* save_mean, const float* save_invstd, int training,
                                    void* dx, float* dgamma, float* dbeta, void* stream) {
    if (!dt_ok(dtype) || n <= 0 || C <= 0) return FMMT_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((C + BN_COLS - 1) / BN_COLS);
    if (dtype == FMMT_BF16)
        hipLaunchKernelGGL(bn1d_bwd_kernel<bf16>, grid, dim3(BN_COLS * BN_GROUPS), 0, st, n, C, (const bf16*)dy, (const bf16*)x, gamma,
                           save_mean, save_invstd, training, (bf16*)dx, dgamma, dbeta);
    else
        hipLaunchKernelGGL(bn1d_bwd_kernel<float>, grid, dim3(BN_COLS * BN_GROUPS), 0, st, n, C, (const float*)dy, (const float*)x, gamma,
                           save_mean, save_invstd, training, (float*)dx, dgamma, dbeta);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_posemb_scale_fwd(int dtype, int L, int B, int E, const void* x, const float* table,
                                     float scale, void* y, void* stream) {
    if (!dt_ok(dtype) || L <= 0 || B <= 0 || E <= 0) return FMMT_EINVAL;
    const int vec = dtype == FMMT_BF16 ? 8 : 4;
    if (E % vec) return FMMT_EINVAL;
    const size_t total = (size_t)L * B * (E / vec);
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FMMT_BF16) hipLaunchKernelGGL(posemb_scale_kernel<bf16>, grid, dim3(256), 0, st, L, B, E, (const bf16*)x, table, scale, (bf16*)y);
    else hipLaunchKernelGGL(posemb_scale_kernel<float>, grid, dim3(256), 0, st, L, B, E, (const float*)x, table, scale, (float*)y);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_scale(int dtype, size_t n, const void* x, float alpha, void* y, void* stream) {
    if (!dt_ok(dtype) || n == 0) return FMMT_EINVAL;
    const int vec = dtype == FMMT_BF16 ? 8 : 4;
    if (n % vec) return FMMT_EINVAL;
    const size_t nvec = n / vec;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FMMT_BF16) hipLaunchKernelGGL(scale_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, st, nvec, (const bf16*)x, alpha, (bf16*)y);
    else hipLaunchKernelGGL(scale_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, nvec, (const float*)x, alpha, (float*)y);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_colsum(int dtype, int out_dtype, int M, int N, const void* x, int ldx, void* out, void* stream) {
    if (M <= 0 || N <= 0) return FMMT_EINVAL;
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    if (out_dtype != dtype && out_dtype != FMMT_F32) return FMMT_EINVAL;
    const int vec = dtype == FMMT_BF16 ? 8 : 4;
    if (N % vec || ldx % vec) return FMMT_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return FMMT_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((N + 4 * vec - 1) / (4 * vec)));
    if (dtype == FMMT_BF16) {
        if (out_dtype == FMMT_BF16) hipLaunchKernelGGL((colsum_kernel<bf16, bf16>), grid, dim3(256), 0, st, M, N, (const bf16*)x, ldx, (bf16*)out);
        else hipLaunchKernelGGL((colsum_kernel<bf16, float>), grid, dim3(256), 0, st, M, N, (const bf16*)x, ldx, (float*)out);
    } else {
        hipLaunchKernelGGL((colsum_kernel<float, float>), grid, dim3(256), 0, st, M, N, (const float*)x, ldx, (float*)out);
    }
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_cast_batch(int n_desc, int n_tiles, const void* desc, void* stream) {
    if (n_desc <= 0 || n_tiles <= 0 || !desc) return FMMT_EINVAL;
    hipLaunchKernelGGL(cast_batch_kernel, dim3((unsigned)n_tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const CastDesc*>(desc), n_desc);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_grad_handover(int n_desc, int n_blocks, const void* desc, float* partial, float* norm_out, void* stream) {
    if (n_desc <= 0 || n_blocks <= 0 || !desc || !partial || !norm_out) return FMMT_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(grad_handover_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, reinterpret_cast<const HoDesc*>(desc), n_desc, partial);
    FMMT_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(1024), 0, st, (const float*)partial, n_blocks, norm_out);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_adamw_batch(int n_desc, int n_blocks, const void* desc, const float* lr, const float* step, const float* total_norm,
                                float beta1, float beta2, float eps, float weight_decay, float max_norm, int hf_semantics, void* stream) {
    if (n_desc <= 0 || n_blocks <= 0 || !desc || !lr || !step) return FMMT_EINVAL;
    hipLaunchKernelGGL(adamw_batch_kernel, dim3((unsigned)n_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const AdamDesc*>(desc), n_desc, lr, step, total_norm, beta1, beta2, eps, weight_decay, max_norm, hf_semantics);
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t fmmt_layernorm_bwd_bf16_workspace(int M, int C) {
    if (M <= 0 || C <= 0) return 0;
    const int blocks = M / 4 < 1 ? 1 : (M / 4 > 256 ? 256 : M / 4);
    return (size_t)blocks * 2 * (size_t)C * sizeof(float);
}

extern "C" int fmmt_layernorm_bwd_bf16(int M, int C, float eps, const void* dy, const void* x, const void* gamma, void* dx,
                                       void* dgamma, void* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    if (M <= 0 || C <= 0 || C % 8 || C > 2048) return FMMT_EINVAL;
    if (!dy || !x || !gamma || !dx || !dgamma || !dbeta || !workspace) return FMMT_EINVAL;
    if (workspace_bytes < fmmt_layernorm_bwd_bf16_workspace(M, C)) return FMMT_EWORKSPACE;
    if (((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dx)) & 15) != 0)
        return FMMT_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int blocks = M / 4 < 1 ? 1 : (M / 4 > 256 ? 256 : M / 4);
    hipLaunchKernelGGL(lnp_bwd_kernel, dim3(blocks), dim3(256), 0, st, M, C, eps, (const bf16*)dy, (const bf16*)x, (const bf16*)gamma, (bf16*)dx,
                       reinterpret_cast<float*>(workspace));
    FMMT_CHECK_LAUNCH();
    hipLaunchKernelGGL(lnp_reduce_kernel, dim3((2 * C + 31) / 32), dim3(1024), 0, st, reinterpret_cast<const float*>(workspace), blocks, C,
                       (bf16*)dgamma, (bf16*)dbeta);
    FMMT_CHECK_LAUNCH();
    return 0;
}
