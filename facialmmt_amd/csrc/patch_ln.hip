// fmmt_patch_embed_ln_fwd: PatchEmbed's projection, bias and LayerNorm in one launch (Swin_Transformer.py:392-422:
// proj = Conv2d(3, 96, kernel 4, stride 4) on the patch matrix, then norm_layer(embed_dim)).
//
//   x_pre[m][c] = bf16( sum_k cols[m][k] w[c][k] + bias[c] )        k < 48 = 3 x 4 x 4, c < 96
//   y[m][c]     = bf16( (x_pre[m][c] - mean_m) rstd_m gamma[c] + beta[c] ),   statistics over the 96 ROUNDED x_pre values of the row
//
// (the statistics are taken from the rounded values so that they are the ones fmmt_layernorm_bwd finds when it re-reads x_pre).
// One wave = 32 patches; D = W (A side, 16-channel tiles in the chan_of<24> order) . cols^T, so that a lane ends up with the channels
// c * 32 + lg * 8 + e of patch li -- the whole row in the four lanes li + 16 g: the LayerNorm is in-lane sums + two swaps, as in the
// fused Mlp's prologue.  K = 48 is one 32-deep MFMA step plus one whose upper half is zero.  The weights are 9 KB: every lane keeps its
// twelve fragments in registers, no LDS.  HBM: cols in (96 B per patch), y out (192 B), in training x_pre out (192 B) + statistics.
//
// The kernel is written over the element-type trait of elem_trait.h: T = bf16 is the production instantiation (the trait's members are the
// instructions the round-4 kernel spelled out), T = float the parity instantiation -- 8 x mfma_f32_16x16x4 per 32-deep block, nothing
// rounded -- that the fp32 module runs and the reference's `patch_embed` golden is held to at 1e-3.
#include "patch_ln_core.h"

namespace {

template <typename T>
struct PlArgs {
    int M;
    const T* cols;
    const T* w;
    const float* bias;
    const float* gamma;
    const float* beta;
    float eps;
    T* x_pre;
    T* y;
    float* mean;
    float* rstd;
    int tiles;
};

template <typename T>
__global__ __launch_bounds__(512) void patch_embed_ln_kernel(PlArgs<T> p) {
    using E = ElemTrait<T>;
    using F = typename E::frag;
    constexpr int K = 48;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    PatchLnParams<T> P;
    patch_ln_load<T>(P, p.w, p.bias, p.gamma, p.beta, li, lg);
    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int t0 = tile * 256 + wave * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int tok = t0 + mt * 16 + li;
            const int tk = min(tok, p.M - 1);
            const F c0 = E::ld(p.cols + (size_t)tk * K + lg * 8);
            const F c1 = lg < 2 ? E::ld(p.cols + (size_t)tk * K + 32 + lg * 8) : E::zero();
            patch_ln_tile<T>(P, c0, c1, p.eps, (size_t)tok, tok < p.M, lg, p.x_pre, p.y, p.mean, p.rstd);
        }
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int fmmt_patch_embed_ln_fwd(int dtype, int M, int C, int K, const void* cols, const void* w, const float* bias, const float* ln_gamma,
                                       const float* ln_beta, float eps, void* x_pre, void* y, float* mean, float* rstd, void* stream) {
    if ((dtype != FMMT_BF16 && dtype != FMMT_F32) || M <= 0 || C != 96 || K != 48) return FMMT_EINVAL;   // other geometries: fmmt_linear_fwd + fmmt_layernorm_fwd
    if (!cols || !w || !ln_gamma || !ln_beta || !y || (mean == nullptr) != (rstd == nullptr)) return FMMT_EINVAL;
    if (!al16(cols) || !al16(w) || !al16(y) || (x_pre && !al16(x_pre))) return FMMT_EALIGN;
    const int tiles = (M + 255) / 256, grid = tiles < 2048 ? tiles : 2048;
    if (dtype == FMMT_BF16) {
        PlArgs<bf16> a{M, (const bf16*)cols, (const bf16*)w, bias, ln_gamma, ln_beta, eps, (bf16*)x_pre, (bf16*)y, mean, rstd, tiles};
        hipLaunchKernelGGL(patch_embed_ln_kernel<bf16>, dim3(grid), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), a);
    } else {
        PlArgs<float> a{M, (const float*)cols, (const float*)w, bias, ln_gamma, ln_beta, eps, (float*)x_pre, (float*)y, mean, rstd, tiles};
        hipLaunchKernelGGL(patch_embed_ln_kernel<float>, dim3(grid), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), a);
    }
    FMMT_CHECK_LAUNCH();
    return 0;
}
