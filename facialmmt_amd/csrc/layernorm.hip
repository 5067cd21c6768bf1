// LayerNorm forward / backward for gfx950: HBM-bound streaming kernels.
//
// A row (one token) is owned by a group of G lanes (16 for narrow rows, a whole 64-lane wave for wide
// ones); each lane keeps up to NCH 16-byte chunks of the row in registers (one global read per
// element), statistics are fp32 and reduced with cross-lane shuffles.  A 256-thread workgroup streams
// 256/G rows per pass with fully coalesced 16-byte accesses.  MERGE = the PatchMerging 2x2 gather
// folded into the address computation.
#include "fmmt_common.h"
#include "../../include/fmmt.h"

namespace {

constexpr int LN_BWD_MAX_BLOCKS = 1024;  // four partial blocks per CU: the kernel is HBM-bound and needs the occupancy

struct LnArgs {
    int M, C;
    const void* x;
    const float* gamma;
    const float* beta;
    float eps;
    void* y;
    float* mean;
    float* rstd;
    int merge_hw;
    // backward
    const void* dy;
    const void* add;
    void* dx;
    float* part;  // [gridDim][2][C]
};

// element offset of logical (row, channel ch) in the source tensor
template <bool MERGE>
__device__ __forceinline__ size_t src_offset(int row, int ch, int C, int hw) {
    if constexpr (!MERGE) {
        return (size_t)row * C + ch;
    } else {
        const int Cq = C >> 2, hh = hw >> 1;
        const int q = ch / Cq, cc = ch - q * Cq;
        const int n = row / (hh * hh), rem = row - n * hh * hh;
        const int h2 = rem / hh, w2 = rem - h2 * hh;
        const size_t tok = (size_t)n * hw * hw + (size_t)(2 * h2 + (q & 1)) * hw + 2 * w2 + (q >> 1);
        return tok * Cq + cc;
    }
}

template <int G> __device__ __forceinline__ float group_sum(float v) {
    if constexpr (G == 16) return group16_sum(v);
    else return wave_sum(v);
}

template <typename T, int G, int NCH, bool MERGE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnArgs p) {
    constexpr int VEC = Vec<T>::N;
    constexpr int ROWS = 256 / G;
    const int sub = threadIdx.x % G, rib = threadIdx.x / G;
    const int chunks = p.C / VEC;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    T* __restrict__ yg = reinterpret_cast<T*>(p.y);
    const float invC = 1.0f / (float)p.C;
    for (int row = blockIdx.x * ROWS + rib; row < p.M; row += gridDim.x * ROWS) {
        Vec<T> v[NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = sub + i * G;
            if (c < chunks) {
                v[i] = ldvec<T>(xg + src_offset<MERGE>(row, c * VEC, p.C, p.merge_hw));
#pragma unroll
                for (int e = 0; e < VEC; ++e) s += v[i].get(e);
            }
        }
        const float mean = group_sum<G>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = sub + i * G;
            if (c < chunks) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float d = v[i].get(e) - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(group_sum<G>(q) * invC + p.eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = sub + i * G;
            if (c < chunks) {
                Vec<T> o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int ch = c * VEC + e;
                    o.set(e, (v[i].get(e) - mean) * rstd * p.gamma[ch] + p.beta[ch]);
                }
                stvec<T>(yg + (size_t)row * p.C + c * VEC, o);
            }
        }
        if (sub == 0) {
            if (p.mean) p.mean[row] = mean;
            if (p.rstd) p.rstd[row] = rstd;
        }
    }
}

template <typename T, int G, int NCH, bool MERGE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnArgs p) {
    constexpr int VEC = Vec<T>::N;
    constexpr int ROWS = 256 / G;
    __shared__ float red[2][ROWS][G * VEC];
    const int sub = threadIdx.x % G, rib = threadIdx.x / G;
    const int chunks = p.C / VEC;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);
    const T* __restrict__ addg = reinterpret_cast<const T*>(p.add);
    T* __restrict__ dxg = reinterpret_cast<T*>(p.dx);
    const float invC = 1.0f / (float)p.C;

    float dg[NCH][VEC], db[NCH][VEC];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) dg[i][e] = db[i][e] = 0.f;

    for (int row = blockIdx.x * ROWS + rib; row < p.M; row += gridDim.x * ROWS) {
        const float mean = p.mean[row], rstd = p.rstd[row];
        // the residual gradient is requested with the row's other loads where that measured faster (us per launch, one profile: three
        // chunks per lane 87 -> 75, the PatchMerging forms 395 -> 247 / 255 -> 191 / 109 -> 72) and behind the row sums, as before,
        // where it did not (one and two chunks per lane)
        constexpr bool EARLY_ADD = MERGE || NCH >= 3;
        Vec<T> xv[NCH], gv[NCH], avp[EARLY_ADD ? NCH : 1];
        float s1 = 0.f, s2 = 0.f;
        if constexpr (EARLY_ADD) {                          // ... and all of the row's loads in front of the arithmetic
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = sub + i * G;
                if (c < chunks) {
                    const size_t off = src_offset<MERGE>(row, c * VEC, p.C, p.merge_hw);
                    xv[i] = ldvec<T>(xg + off);
                    gv[i] = ldvec<T>(dyg + (size_t)row * p.C + c * VEC);
                    if (addg) avp[i] = ldvec<T>(addg + off);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = sub + i * G;
            if (c < chunks) {
                if constexpr (!EARLY_ADD) {
                    xv[i] = ldvec<T>(xg + src_offset<MERGE>(row, c * VEC, p.C, p.merge_hw));
                    gv[i] = ldvec<T>(dyg + (size_t)row * p.C + c * VEC);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xh = (xv[i].get(e) - mean) * rstd;
                    const float dyv = gv[i].get(e);
                    const float g = dyv * p.gamma[c * VEC + e];
                    s1 += g;
                    s2 += g * xh;
                    dg[i][e] += dyv * xh;
                    db[i][e] += dyv;
                }
            }
        }
        s1 = group_sum<G>(s1) * invC;
        s2 = group_sum<G>(s2) * invC;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = sub + i * G;
            if (c < chunks) {
                const size_t off = src_offset<MERGE>(row, c * VEC, p.C, p.merge_hw);
                Vec<T> o, av;
                if constexpr (EARLY_ADD) {
                    if (addg) av = avp[i];
                } else {
                    if (addg) av = ldvec<T>(addg + off);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xh = (xv[i].get(e) - mean) * rstd;
                    const float g = gv[i].get(e) * p.gamma[c * VEC + e];
                    float d = rstd * (g - s1 - xh * s2);
                    if (addg) d += av.get(e);
                    o.set(e, d);
                }
                stvec<T>(dxg + off, o);
            }
        }
    }

    // cross-row reduction inside the workgroup, one 16-chunk slab at a time, then one partial per block
    float* part = p.part + (size_t)blockIdx.x * 2 * p.C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            red[0][rib][sub * VEC + e] = dg[i][e];
            red[1][rib][sub * VEC + e] = db[i][e];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * G * VEC; t += 256) {
            const int which = t / (G * VEC), col = t % (G * VEC);
            const int ch = i * G * VEC + col;
            if (ch < p.C) {
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) a += red[which][r][col];
                part[which * p.C + ch] = a;
            }
        }
    }
}

// [nblocks][2][C] partials -> dgamma/dbeta.  1024 threads = 16 columns x 64 partial groups; fixed-order tree.  (64 columns x 16
// groups left C = 96 with three workgroups walking up to 1024 partial rows: 234 us for the stage-0 LayerNorm, 14.5 us on average.)
__global__ __launch_bounds__(1024) void ln_reduce_kernel(const float* __restrict__ part, int nblocks, int C, float* dgamma, float* dbeta) {
    __shared__ float red[64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + tx;
    float a = 0.f;
    if (i < 2 * C)
        for (int b = ty; b < nblocks; b += 64) a += part[(size_t)b * 2 * C + i];
    red[ty][tx] = a;
    __syncthreads();
    if (ty < 16) {                                          // 16 x 16 threads: four partial groups each, then a 16-lane tree in LDS order
        float t = (red[ty][tx] + red[ty + 16][tx]) + (red[ty + 32][tx] + red[ty + 48][tx]);
        red[ty][tx] = t;
    }
    __syncthreads();
    if (ty == 0 && i < 2 * C) {
        a = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) a += red[g][tx];
        if (i < C) { if (dgamma) dgamma[i] = a; }
        else { if (dbeta) dbeta[i - C] = a; }
    }
}

template <typename T, bool MERGE, bool BWD>
int launch_ln(const LnArgs& a, int g, int nch, int grid, hipStream_t st) {
#define FMMT_LN_CASE(GG, N)                                                                                  \
    if (g == GG && nch == N) {                                                                               \
        if constexpr (BWD) hipLaunchKernelGGL((ln_bwd_kernel<T, GG, N, MERGE>), dim3(grid), dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((ln_fwd_kernel<T, GG, N, MERGE>), dim3(grid), dim3(256), 0, st, a);          \
        FMMT_CHECK_LAUNCH();                                                                                 \
        return 0;                                                                                            \
    }
    FMMT_LN_CASE(16, 1)
    FMMT_LN_CASE(16, 2)
    FMMT_LN_CASE(16, 3)
    FMMT_LN_CASE(64, 2)
    FMMT_LN_CASE(64, 3)
    FMMT_LN_CASE(64, 6)
#undef FMMT_LN_CASE
    return FMMT_EINVAL;
}

// (group width, chunks per lane) for a row of C elements: 16-lane groups up to 48 chunks, else a wave
bool pick_shape(int C, int vec, int* g, int* nch) {
    const int chunks = C / vec;
    if (chunks <= 48) {
        *g = 16;
        *nch = (chunks + 15) / 16;
        return true;
    }
    *g = 64;
    const int need = (chunks + 63) / 64;
    *nch = need <= 2 ? 2 : need <= 3 ? 3 : need <= 6 ? 6 : -1;
    return *nch > 0;
}

int check_ln(int dtype, int M, int C, int merge_hw) {
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    const int vec = dtype == FMMT_BF16 ? 8 : 4;
    if (M <= 0 || C <= 0 || C % vec) return FMMT_EINVAL;
    if (merge_hw) {
        if (merge_hw % 2 || C % 4 || (C / 4) % vec) return FMMT_EINVAL;
        if (M % ((merge_hw / 2) * (merge_hw / 2))) return FMMT_EINVAL;
    }
    int g, nch;
    return pick_shape(C, vec, &g, &nch) ? 0 : FMMT_EINVAL;
}

}  // namespace

extern "C" int fmmt_layernorm_fwd(int dtype, int M, int C, const void* x, const float* gamma, const float* beta,
                                  float eps, void* y, float* mean, float* rstd, int merge_hw, void* stream) {
    if (int e = check_ln(dtype, M, C, merge_hw)) return e;
    LnArgs a{};
    a.M = M; a.C = C; a.x = x; a.gamma = gamma; a.beta = beta; a.eps = eps; a.y = y; a.mean = mean; a.rstd = rstd;
    a.merge_hw = merge_hw;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int vec = dtype == FMMT_BF16 ? 8 : 4;
    int g, nch;
    pick_shape(C, vec, &g, &nch);
    const int rows = 256 / g;
    int grid = (M + rows - 1) / rows;
    if (grid > 8192) grid = 8192;
    if (dtype == FMMT_BF16) return merge_hw ? launch_ln<bf16, true, false>(a, g, nch, grid, st) : launch_ln<bf16, false, false>(a, g, nch, grid, st);
    return merge_hw ? launch_ln<float, true, false>(a, g, nch, grid, st) : launch_ln<float, false, false>(a, g, nch, grid, st);
}

extern "C" size_t fmmt_layernorm_bwd_workspace(int C) { return (size_t)LN_BWD_MAX_BLOCKS * 2 * (size_t)C * sizeof(float); }

extern "C" int fmmt_layernorm_bwd(int dtype, int M, int C, const void* dy, const void* x, const float* mean,
                                  const float* rstd, const float* gamma, const void* add, void* dx,
                                  float* dgamma, float* dbeta, int merge_hw,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = check_ln(dtype, M, C, merge_hw)) return e;
    if (workspace_bytes < fmmt_layernorm_bwd_workspace(C)) return FMMT_EWORKSPACE;
    LnArgs a{};
    a.M = M; a.C = C; a.x = x; a.gamma = gamma; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
    a.merge_hw = merge_hw; a.dy = dy; a.add = add; a.dx = dx; a.part = reinterpret_cast<float*>(workspace);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int vec = dtype == FMMT_BF16 ? 8 : 4;
    int g, nch;
    pick_shape(C, vec, &g, &nch);
    const int rows = 256 / g;
    int grid = (M + rows - 1) / rows;
    if (grid > LN_BWD_MAX_BLOCKS) grid = LN_BWD_MAX_BLOCKS;
    int rc;
    if (dtype == FMMT_BF16) rc = merge_hw ? launch_ln<bf16, true, true>(a, g, nch, grid, st) : launch_ln<bf16, false, true>(a, g, nch, grid, st);
    else rc = merge_hw ? launch_ln<float, true, true>(a, g, nch, grid, st) : launch_ln<float, false, true>(a, g, nch, grid, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ln_reduce_kernel, dim3((2 * C + 15) / 16), dim3(1024), 0, st, a.part, grid, C, dgamma, dbeta);
    FMMT_CHECK_LAUNCH();
    return 0;
}
