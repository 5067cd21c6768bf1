// The facial-emotion frame filter of a target-task step as ONE launch per direction (train.py:75-114).
//
// The reference walks Python loops over the batch: faces whose emotion distribution p (Gumbel-softmax of the Swin logits) has sum(p^2) above a
// threshold are kept, every utterance packs its kept faces to the front (new mask = how many were kept) together with the matching rows of its
// pre-extracted vision features, and the 7 probabilities are appended to those rows; if no face of the whole batch passes, everything real is kept.
// train_step.select_frames restates that without host synchronisation as ~55 torch launches (cumsum / searchsorted / index_put / where ...), each a
// 5 us link in the chain between Swin's forward and the fusion stack.  This file is the same index arithmetic in one kernel:
//   * every workgroup recomputes the (tiny) index maps in LDS -- selection flags, their inclusive prefix sum, the utterance boundaries with the
//     reference's (n - 1) margin quirk: utterance u owns the faces [b_{u-1}, b_u) with b_u = sum_{i<=u} n_i - u -- and then writes its share of the
//     (B, Lv, D + NL) output rows;
//   * `src_face` (B, Lv) records which face fed the emotion columns of a slot (-1: none): the backward is a gather of d(out)[..., D:] through it.
// Requirements (FMMT_EINVAL otherwise): nF <= 8192 faces, B <= 256, B * Lv <= 8192 slots; num_imgs[u] <= Lv is the caller's contract (the torch
// restatement clamps and accumulates colliding faces for such invalid input; here the last one wins).
#include "fmmt_common.h"
#include "../../include/fmmt.h"

namespace {

constexpr int FF_THREADS = 256;

template <typename T>
__global__ __launch_bounds__(FF_THREADS) void select_frames_fwd_kernel(int nF, int NL, int B, int Lv, int D, const float* __restrict__ preds,
                                                                       const T* __restrict__ vin, const float* __restrict__ vmask,
                                                                       const long long* __restrict__ num_imgs, float threshold, T* __restrict__ out,
                                                                       float* __restrict__ new_mask, int* __restrict__ src_face) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* cum = reinterpret_cast<int*>(smem);                  // [nF]   inclusive prefix sum of `owned`
    int* slot = cum + nF;                                     // [B*Lv] face feeding the slot (selection branch), -1 = empty
    int* upper = slot + B * Lv;                               // [B]    exclusive upper face index of utterance u
    int* nimg = upper + B;                                    // [B]
    int* before = nimg + B;                                   // [B]    owned faces in front of utterance u
    int* run = before + B;                                    // [B]    leading run of ones of vision_mask[u] (fallback branch)
    int* offs = run + B;                                      // [B]    faces in front of utterance u in the fallback enumeration
    int* part = offs + B;                                     // [FF_THREADS] scan partials
    __shared__ int n_sel;
    const int tid = threadIdx.x;
    if (tid == 0) {
        n_sel = 0;
        long long c = 0;
        for (int u = 0; u < B; ++u) {
            const long long n = num_imgs[u];
            c += n;
            nimg[u] = (int)n;
            upper[u] = (int)(c - u);
        }
    }
    for (int i = tid; i < B * Lv; i += FF_THREADS) slot[i] = -1;
    __syncthreads();
    // owned(g) = importance above the threshold AND g below the last boundary; chunked block scan
    const int CH = (nF + FF_THREADS - 1) / FF_THREADS;
    const int g0 = tid * CH, g1 = min(g0 + CH, nF);
    int local = 0, sel_local = 0;
    const int last_upper = upper[B - 1];
    for (int g = g0; g < g1; ++g) {
        float imp = 0.f;
        for (int j = 0; j < NL; ++j) { const float p = preds[(size_t)g * NL + j]; imp += p * p; }
        const bool sel = imp > threshold;
        sel_local += sel ? 1 : 0;
        local += (sel && g < last_upper) ? 1 : 0;
        cum[g] = (sel && g < last_upper) ? 1 : 0;
    }
    part[tid] = local;
    if (sel_local) atomicAdd(&n_sel, sel_local);
    __syncthreads();
    if (tid == 0) {                                            // 256 partials: a serial exclusive scan is a few hundred cycles
        int a = 0;
        for (int i = 0; i < FF_THREADS; ++i) { const int v = part[i]; part[i] = a; a += v; }
    }
    __syncthreads();
    {
        int a = part[tid];
        for (int g = g0; g < g1; ++g) { a += cum[g]; cum[g] = a; }
    }
    __syncthreads();
    if (tid < B) {
        const int u = tid;
        const int first = min(u == 0 ? 0 : upper[u - 1], nF);
        before[u] = first > 0 ? cum[first - 1] : 0;
        int r = 0;                                             // leading run of ones: cumsum(mask)[k] == k + 1
        float cs = 0.f;
        for (int k = 0; k < Lv; ++k) {
            cs += vmask[(size_t)u * Lv + k];
            if (cs == (float)(k + 1)) r = k + 1; else break;
        }
        run[u] = r;
    }
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        for (int u = 0; u < B; ++u) { offs[u] = a; a += run[u]; }
    }
    // slot map of the selection branch
    for (int g = tid; g < nF; g += FF_THREADS) {
        const int c = cum[g], prev = g > 0 ? cum[g - 1] : 0;
        if (c == prev) continue;                               // not owned
        int u = 0;
        while (u < B && upper[u] <= g) ++u;                    // searchsorted(upper, g, right=True); owned => u < B
        const int k = min(max(c - 1 - before[u], 0), Lv - 1);
        slot[u * Lv + k] = g;
    }
    __syncthreads();
    const bool any_sel = n_sel > 0;
    const int W = D + NL;
    // output rows, grid-strided over the B * Lv slots; one row per wave-sized group of threads would waste lanes at D = 512: the whole block per row
    for (int r = blockIdx.x; r < B * Lv; r += gridDim.x) {
        const int u = r / Lv, k = r - u * Lv;
        int face, vrow;
        float m;
        bool keep_in;
        if (any_sel) {
            face = slot[r];
            const int margin = upper[u] - nimg[u];
            vrow = face >= 0 ? min(max(face - margin, 0), Lv - 1) : -1;
            keep_in = face >= 0;
            const int first = min(u == 0 ? 0 : upper[u - 1], nF), lastf = min(upper[u], nF);
            const int cnt = (lastf > 0 ? cum[lastf - 1] : 0) - (first > 0 ? cum[first - 1] : 0);
            m = k < cnt ? 1.f : 0.f;
        } else {
            const bool real = k < run[u];
            face = real ? min(offs[u] + k, nF - 1) : -1;
            vrow = k;
            keep_in = true;
            m = vmask[r];
        }
        T* o = out + (size_t)r * W;
        const T* vi = vin + ((size_t)u * Lv + (vrow >= 0 ? vrow : 0)) * D;
        for (int c = tid; c < D; c += FF_THREADS) o[c] = keep_in ? vi[c] : (T)0.f;
        if (tid < NL) o[D + tid] = face >= 0 ? (T)preds[(size_t)face * NL + tid] : (T)0.f;
        if (tid == 0) {
            new_mask[r] = m;
            src_face[r] = face;
        }
    }
}

template <typename T>
__global__ void select_frames_bwd_kernel(int NL, int slots, int W, int D, const T* __restrict__ dout, const int* __restrict__ src_face,
                                         float* __restrict__ dpreds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots * NL) return;
    const int r = i / NL, j = i - r * NL;
    const int f = src_face[r];
    if (f >= 0) atomicAdd(dpreds + (size_t)f * NL + j, (float)dout[(size_t)r * W + D + j]);   // one slot per face except for clamped (invalid) input
}

size_t ff_lds(int nF, int B, int Lv) { return ((size_t)nF + (size_t)B * Lv + 5 * (size_t)B + FF_THREADS) * sizeof(int); }

}  // namespace

extern "C" int fmmt_select_frames_fwd(int dtype, int nF, int NL, int B, int Lv, int D, const float* preds, const void* vision_inputs,
                                      const float* vision_mask, const int64_t* num_imgs, float threshold, void* out, float* new_mask,
                                      int32_t* src_face, void* stream) {
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    if (nF <= 0 || nF > 8192 || NL <= 0 || NL > FF_THREADS || B <= 0 || B > 256 || Lv <= 0 || B * Lv > 8192 || D <= 0) return FMMT_EINVAL;
    if (!preds || !vision_inputs || !vision_mask || !num_imgs || !out || !new_mask || !src_face) return FMMT_EINVAL;
    const size_t lds = ff_lds(nF, B, Lv);
    if (lds > 96 * 1024) return FMMT_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = B * Lv < 256 ? B * Lv : 256;
    if (dtype == FMMT_BF16) {
        static FmmtLdsOnce once;
        if (int rc = once.set(reinterpret_cast<const void*>(&select_frames_fwd_kernel<bf16>), 96 * 1024)) return rc;
        hipLaunchKernelGGL(select_frames_fwd_kernel<bf16>, dim3(grid), dim3(FF_THREADS), lds, st, nF, NL, B, Lv, D, preds, (const bf16*)vision_inputs, vision_mask,
                           (const long long*)num_imgs, threshold, (bf16*)out, new_mask, src_face);
    } else {
        static FmmtLdsOnce once;
        if (int rc = once.set(reinterpret_cast<const void*>(&select_frames_fwd_kernel<float>), 96 * 1024)) return rc;
        hipLaunchKernelGGL(select_frames_fwd_kernel<float>, dim3(grid), dim3(FF_THREADS), lds, st, nF, NL, B, Lv, D, preds, (const float*)vision_inputs, vision_mask,
                           (const long long*)num_imgs, threshold, (float*)out, new_mask, src_face);
    }
    FMMT_CHECK_LAUNCH();
    return 0;
}

extern "C" int fmmt_select_frames_bwd(int dtype, int nF, int NL, int B, int Lv, int D, const void* dout, const int32_t* src_face, float* dpreds,
                                      void* stream) {
    if (dtype != FMMT_BF16 && dtype != FMMT_F32) return FMMT_EINVAL;
    if (nF <= 0 || NL <= 0 || B <= 0 || Lv <= 0 || D <= 0 || !dout || !src_face || !dpreds) return FMMT_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipError_t e = hipMemsetAsync(dpreds, 0, (size_t)nF * NL * sizeof(float), st)) return (int)e;
    const int total = B * Lv * NL;
    if (dtype == FMMT_BF16)
        hipLaunchKernelGGL(select_frames_bwd_kernel<bf16>, dim3((total + 255) / 256), dim3(256), 0, st, NL, B * Lv, D + NL, D, (const bf16*)dout, src_face, dpreds);
    else
        hipLaunchKernelGGL(select_frames_bwd_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, st, NL, B * Lv, D + NL, D, (const float*)dout, src_face, dpreds);
    FMMT_CHECK_LAUNCH();
    return 0;
}
