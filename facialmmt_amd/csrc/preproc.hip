// Input pre-step fused into PatchEmbed's gather: uint8 face crops go straight to the
// (patches x 48) operand of the patch-embedding GEMM -- bicubic resize to 224x224 (uint8, integer arithmetic, bit-exact
// with the library the reference calls), ToTensor (/255), Normalize(.5,.5), 4x4 patch gather -- without ever
// materialising the 224x224 float tensor the reference caches per frame (utils/dataset.py:47-69,275; utils/util.py:43-52).
//
// Two resize flavours (oracle/preproc.py restates both on the CPU and says which one is pinned):
//   FMMT_RESIZE_PIL  Pillow Image.resize(BICUBIC) -- the Aff-Wild2 path (transforms.Resize on a PIL image, util.py:45):
//                    a = -0.5, truncated + renormalised window at the border, 22-bit coefficients, uint8 after EACH pass;
//   FMMT_RESIZE_CV2  cv2.resize(INTER_CUBIC) on uint8 -- the MELD path (dataset.py:57): a = -0.75 in float, replicate
//                    border, 11-bit coefficients, un-rounded int32 row sums, one 22-bit rounding after the vertical pass.
// Both reduce to: per output coordinate four source indices and four integer weights (one table, used for x and for y --
// the crops are square), built on the HOST by fmmt_resize_table() so that the double / float coefficient arithmetic is
// the library's own, and uploaded once per (flavour, size) by the caller.
#include "fmmt_common.h"
#include "../../include/fmmt.h"
#include "patch_ln_core.h"
#include <math.h>

namespace {

constexpr int OUT = 224, GRID = 56, KPATCH = 48;
constexpr int RMAX = 8;              // source rows a band of 4 output rows may touch (checked on the host for the table)
constexpr int SMAX = 224;            // largest source edge (up-scaling only)

double pil_cubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// block = (patch row p, image n); 256 threads.  Round 6 form (the round-5 kernel kept the band's horizontal pass in a 21 KB LDS array, re-read its tap
// tables from LDS for every element and ran at 0.19 of the HBM rate its 0.6 GB would allow -- instruction- and LDS-issue-bound):
//   1. stage the source rows [ymin, ymax] of the band as bytes            (coalesced 4-byte loads)
//   2. a thread OWNS output columns (x, c) = tid, tid + 256, tid + 512 of the band: its four horizontal taps (source offsets, weights) are two 16-byte
//      table loads per column, its horizontal results for the band's <= 8 source rows stay in REGISTERS (PIL: uint8-rounded; cv2: raw int32 sums) --
//   3. -- and feed the vertical pass directly: the band's 4 x 4 vertical taps are workgroup-uniform (scalar registers), folded once into a 4 x 8 matrix
//      W[dy][r] of weights per staged row, so an output value is 8 integer multiply-adds on registers; rounding, byte -> float through the ToTensor /
//      Normalize table, into the LDS patch tile (patch-column order).  Same integer sums as before (integer addition is exact: the order is free).
//   4. the patch tile goes out as whole 16-byte chunks (training: the projection's weight gradient reads it);
//   FUSE (fmmt_patch_embed_u8_ln_fwd): 5. the four waves run PatchEmbed's projection + bias + LayerNorm on the tile (patch_ln_core.h: 16 patches per wave,
//      the tile arithmetic of patch_embed_ln_kernel) -- the patch matrix is written for the backward or not at all (inference), never read back.
template <typename T>
struct PeTail {                                             // the projection + LayerNorm behind the gather (FUSE)
    const T* w;
    const float* bias;
    const float* gamma;
    const float* beta;
    float eps;
    T* x_pre;
    T* y;
    float* mean;
    float* rstd;
};
constexpr int CPITCH = 56;                                  // LDS pitch of a patch row (48 values): 112 / 224 bytes, 16-byte aligned fragments

template <typename T, bool PIL, bool FUSE>
__global__ __launch_bounds__(256) void patch_embed_u8_kernel(const uint8_t* __restrict__ img, int S, const int32_t* __restrict__ tab,
                                                            const float* __restrict__ lut, T* __restrict__ cols, PeTail<T> tail) {
    __shared__ __attribute__((aligned(16))) uint8_t src[RMAX * SMAX * 3];
    __shared__ __attribute__((aligned(16))) T colsl[64 * CPITCH];
    __shared__ float slut[256];
    const int p = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    slut[tid] = lut[tid];
    // the band's vertical taps: the same for every thread (uniform addresses: scalar loads)
    int tyi[4][4], tyw[4][4];
    int ymin = S, ymax = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tyi[j][k] = tab[(4 * p + j) * 8 + k];
            tyw[j][k] = tab[(4 * p + j) * 8 + 4 + k];
            ymin = min(ymin, tyi[j][k]);
            ymax = max(ymax, tyi[j][k]);
        }
    const int nr = ymax - ymin + 1;                          // <= RMAX (host-checked)
    int W[4][RMAX];                                          // weight of staged row r in output row dy of the band
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            int w = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) w += (tyi[j][k] - ymin == r) ? tyw[j][k] : 0;
            W[j][r] = w;
        }
    const int rowb = S * 3;
    const uint8_t* base = img + ((size_t)n * S + ymin) * rowb;
    const int nbytes = nr * rowb;                            // the rows are contiguous in the image
    // (n * S + ymin) * S * 3 is a multiple of 4 only for some rows: stage bytes with a 4-byte main loop on the aligned part
    const int mis = (int)((uintptr_t)base & 3);
    const int head = mis ? min(4 - mis, nbytes) : 0;
    if (tid < head) src[tid] = base[tid];
    const int words = (nbytes - head) >> 2;
    const uint32_t* b4 = reinterpret_cast<const uint32_t*>(base + head);
    for (int i = tid; i < words; i += 256) {
        const uint32_t v = b4[i];
        const int o = head + 4 * i;
        src[o] = (uint8_t)v;
        src[o + 1] = (uint8_t)(v >> 8);
        src[o + 2] = (uint8_t)(v >> 16);
        src[o + 3] = (uint8_t)(v >> 24);
    }
    for (int i = head + 4 * words + tid; i < nbytes; i += 256) src[i] = base[i];
    if constexpr (FUSE) {
        for (int e = tid; e < 8 * KPATCH; e += 256) colsl[(GRID + e / KPATCH) * CPITCH + e % KPATCH] = from_f32<T>(0.f);   // rows 56-63 of the last wave's tile
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int rem = tid + 256 * i;                       // output column: x = rem / 3, channel c = rem % 3
        if (rem >= OUT * 3) break;
        const int x = rem / 3, c = rem - 3 * x;
        const int4 ti = *reinterpret_cast<const int4*>(tab + x * 8), tw = *reinterpret_cast<const int4*>(tab + x * 8 + 4);
        const int o0 = ti.x * 3 + c, o1 = ti.y * 3 + c, o2 = ti.z * 3 + c, o3 = ti.w * 3 + c;
        int h[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            if (r < nr) {                                    // uniform
                const uint8_t* s = src + r * rowb;
                int32_t acc = PIL ? (1 << 21) : 0;
                acc += (int32_t)s[o0] * tw.x + (int32_t)s[o1] * tw.y + (int32_t)s[o2] * tw.z + (int32_t)s[o3] * tw.w;
                h[r] = PIL ? min(max(acc >> 22, 0), 255) : acc;
            } else {
                h[r] = 0;
            }
        }
        const int patch = x >> 2, dx = x & 3;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            int32_t acc = 1 << 21;
#pragma unroll
            for (int r = 0; r < RMAX; ++r) acc += h[r] * W[dy][r];
            const int v = min(max(acc >> 22, 0), 255);
            colsl[patch * CPITCH + c * 16 + dy * 4 + dx] = from_f32<T>(slut[v]);
        }
    }
    __syncthreads();
    if (!FUSE || cols) {                                     // the patch matrix of the band: 56 rows of 48 values, whole 16-byte chunks
        constexpr int CH = 16 / (int)sizeof(T), CPR = KPATCH / CH;
        T* out = cols + ((size_t)n * GRID + p) * GRID * KPATCH;
        for (int q = tid; q < GRID * CPR; q += 256) {
            const int patch = q / CPR, ch = q - patch * CPR;
            *reinterpret_cast<uint4*>(out + patch * KPATCH + ch * CH) = *reinterpret_cast<const uint4*>(colsl + patch * CPITCH + ch * CH);
        }
    }
    if constexpr (FUSE) {
        using E = ElemTrait<T>;
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
        const int patch = wave * 16 + li;
        const typename E::frag c0 = E::ld(colsl + patch * CPITCH + lg * 8);
        const typename E::frag c1 = lg < 2 ? E::ld(colsl + patch * CPITCH + 32 + lg * 8) : E::zero();
        // (the projection's 9 KB of weights and the per-channel constants are loaded where they are used: held from the kernel's start -- round 5 -- they
        //  made this a 172-register kernel, two workgroups per CU under the latency-bound gather above)
        patch_ln_tile_lean<T>(tail.w, tail.bias, tail.gamma, tail.beta, c0, c1, tail.eps, ((size_t)n * GRID + p) * GRID + patch, patch < GRID, li, lg,
                              tail.x_pre, tail.y, tail.mean, tail.rstd);
    }
}

}  // namespace

extern "C" int fmmt_resize_table(int mode, int in_size, int out_size, int32_t* table, float* lut) {
    if ((mode != FMMT_RESIZE_PIL && mode != FMMT_RESIZE_CV2) || in_size < 4 || out_size < in_size || !table) return FMMT_EINVAL;
    if (mode == FMMT_RESIZE_PIL) {
        // Pillow src/libImaging/Resample.c precompute_coeffs / normalize_coeffs_8bpc, bicubic filter (support 2)
        const double scale = (double)in_size / out_size;
        const double fscale = scale < 1.0 ? 1.0 : scale, support = 2.0 * fscale;
        for (int xx = 0; xx < out_size; ++xx) {
            const double center = (xx + 0.5) * scale;
            int xmin = (int)(center - support + 0.5);
            if (xmin < 0) xmin = 0;
            int xmax = (int)(center + support + 0.5);
            if (xmax > in_size) xmax = in_size;
            const int n = xmax - xmin;
            if (n < 1 || n > 4) return FMMT_EINVAL;
            double k[4], ww = 0.0;
            for (int x = 0; x < n; ++x) {
                k[x] = pil_cubic((x + xmin - center + 0.5) * (1.0 / fscale));
                ww += k[x];
            }
            for (int x = 0; x < 4; ++x) {
                table[xx * 8 + x] = xmin + (x < n ? x : n - 1);
                int32_t w = 0;
                if (x < n) {
                    const double v = k[x] / ww;                 // Pillow: k[x] /= ww (ww != 0 always holds for the cubic window)
                    w = v < 0 ? (int32_t)(-0.5 + v * (1 << 22)) : (int32_t)(0.5 + v * (1 << 22));
                }
                table[xx * 8 + 4 + x] = w;
            }
        }
    } else {
        // OpenCV modules/imgproc/src/resize.cpp (8U, INTER_CUBIC): float coordinate and interpolateCubic, A = -0.75
        const double scale = 1.0 / ((double)out_size / in_size);
        const float A = -0.75f;
        for (int dx = 0; dx < out_size; ++dx) {
            float fx = (float)((dx + 0.5) * scale - 0.5);
            const int sx = (int)floorf(fx);
            fx -= sx;
            float c[4];
            c[0] = ((A * (fx + 1) - 5 * A) * (fx + 1) + 8 * A) * (fx + 1) - 4 * A;
            c[1] = ((A + 2) * fx - (A + 3)) * fx * fx + 1;
            c[2] = ((A + 2) * (1 - fx) - (A + 3)) * (1 - fx) * (1 - fx) + 1;
            c[3] = 1.f - c[0] - c[1] - c[2];
            for (int k = 0; k < 4; ++k) {
                int s = sx - 1 + k;
                s = s < 0 ? 0 : (s > in_size - 1 ? in_size - 1 : s);
                table[dx * 8 + k] = s;
                long r = lrintf(c[k] * 2048.f);                  // cvRound: to nearest even (default rounding mode)
                r = r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
                table[dx * 8 + 4 + k] = (int32_t)r;
            }
        }
    }
    if (lut)
        for (int v = 0; v < 256; ++v) lut[v] = ((float)v / 255.0f - 0.5f) / 0.5f;     // ToTensor, then Normalize(.5, .5), in float32
    return 0;
}

extern "C" int fmmt_resize_band_rows(const int32_t* table, int out_size) {
    // largest number of distinct source rows that four consecutive output rows (one patch row) touch
    if (!table || out_size < 4 || out_size % 4) return FMMT_EINVAL;
    int worst = 0;
    for (int p = 0; p < out_size / 4; ++p) {
        int lo = 1 << 30, hi = -1;
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 4; ++k) {
                const int v = table[(4 * p + j) * 8 + k];
                lo = v < lo ? v : lo;
                hi = v > hi ? v : hi;
            }
        worst = hi - lo + 1 > worst ? hi - lo + 1 : worst;
    }
    return worst;
}

extern "C" int fmmt_patch_embed_u8(int dtype, int mode, int n_img, int in_size, const void* img_u8, const int32_t* table_dev,
                                   const float* lut_dev, void* cols, void* stream) {
    if ((dtype != FMMT_BF16 && dtype != FMMT_F32) || (mode != FMMT_RESIZE_PIL && mode != FMMT_RESIZE_CV2)) return FMMT_EINVAL;
    if (n_img <= 0 || n_img > 65535 || in_size < 4 || in_size > SMAX || !img_u8 || !table_dev || !lut_dev || !cols) return FMMT_EINVAL;
    if ((reinterpret_cast<uintptr_t>(table_dev) | reinterpret_cast<uintptr_t>(cols)) & 15) return FMMT_EALIGN;     // 16-byte table loads, 16-byte stores of the patch rows
    // a band of four output rows of an up-scaling (in_size <= 224) table touches at most 4 + 3 + 1 = 8 source rows
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(GRID, n_img);
    const uint8_t* img = reinterpret_cast<const uint8_t*>(img_u8);
    if (dtype == FMMT_BF16) {
        if (mode == FMMT_RESIZE_PIL) hipLaunchKernelGGL((patch_embed_u8_kernel<bf16, true, false>), grid, dim3(256), 0, st, img, in_size, table_dev, lut_dev, (bf16*)cols, PeTail<bf16>{});
        else hipLaunchKernelGGL((patch_embed_u8_kernel<bf16, false, false>), grid, dim3(256), 0, st, img, in_size, table_dev, lut_dev, (bf16*)cols, PeTail<bf16>{});
    } else {
        if (mode == FMMT_RESIZE_PIL) hipLaunchKernelGGL((patch_embed_u8_kernel<float, true, false>), grid, dim3(256), 0, st, img, in_size, table_dev, lut_dev, (float*)cols, PeTail<float>{});
        else hipLaunchKernelGGL((patch_embed_u8_kernel<float, false, false>), grid, dim3(256), 0, st, img, in_size, table_dev, lut_dev, (float*)cols, PeTail<float>{});
    }
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_patch_embed_u8_ln_fwd(int dtype, int mode, int n_img, int in_size, const void* img_u8, const int32_t* table_dev, const float* lut_dev,
                                          const void* w, const float* bias, const float* ln_gamma, const float* ln_beta, float eps,
                                          void* cols, void* x_pre, void* y, float* mean, float* rstd, void* stream) {
    if ((dtype != FMMT_BF16 && dtype != FMMT_F32) || (mode != FMMT_RESIZE_PIL && mode != FMMT_RESIZE_CV2)) return FMMT_EINVAL;
    if (n_img <= 0 || n_img > 65535 || in_size < 4 || in_size > SMAX || !img_u8 || !table_dev || !lut_dev) return FMMT_EINVAL;
    if (!w || !ln_gamma || !ln_beta || !y || (mean == nullptr) != (rstd == nullptr)) return FMMT_EINVAL;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(w) || !al16(y) || (x_pre && !al16(x_pre)) || (cols && !al16(cols)) || !al16(table_dev) || (bias && !al16(bias)) || !al16(ln_gamma) || !al16(ln_beta))
        return FMMT_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(GRID, n_img);
    const uint8_t* img = reinterpret_cast<const uint8_t*>(img_u8);
    if (dtype == FMMT_BF16) {
        const PeTail<bf16> t{(const bf16*)w, bias, ln_gamma, ln_beta, eps, (bf16*)x_pre, (bf16*)y, mean, rstd};
        if (mode == FMMT_RESIZE_PIL) hipLaunchKernelGGL((patch_embed_u8_kernel<bf16, true, true>), grid, dim3(256), 0, st, img, in_size, table_dev, lut_dev, (bf16*)cols, t);
        else hipLaunchKernelGGL((patch_embed_u8_kernel<bf16, false, true>), grid, dim3(256), 0, st, img, in_size, table_dev, lut_dev, (bf16*)cols, t);
    } else {
        const PeTail<float> t{(const float*)w, bias, ln_gamma, ln_beta, eps, (float*)x_pre, (float*)y, mean, rstd};
        if (mode == FMMT_RESIZE_PIL) hipLaunchKernelGGL((patch_embed_u8_kernel<float, true, true>), grid, dim3(256), 0, st, img, in_size, table_dev, lut_dev, (float*)cols, t);
        else hipLaunchKernelGGL((patch_embed_u8_kernel<float, false, true>), grid, dim3(256), 0, st, img, in_size, table_dev, lut_dev, (float*)cols, t);
    }
    FMMT_CHECK_LAUNCH();
    return 0;
}
