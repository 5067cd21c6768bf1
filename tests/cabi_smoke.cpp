// C ABI smoke test WITHOUT Python (level 1 of the test pyramid: "C++/HIP unit tests per kernel vs a CPU
// restatement").  A plain hipcc program that links libfmmt_hip.so, calls three entry points of include/fmmt.h with raw
// hipMalloc'ed buffers in parity mode (FMMT_F32) and compares with straightforward CPU loops written here:
//   fmmt_linear_fwd        y = res + s * (x w^T + b)                            (Swin_Transformer.py:19-28,142)
//   fmmt_layernorm_fwd     nn.LayerNorm, eps 1e-5                               (Swin_Transformer.py:239,243)
//   fmmt_window_attn_fwd   shifted-window attention on token-order qkv          (Swin_Transformer.py:113-144, :33-62, :208-227, :244,261)
// Build + run (tests/test_gpu_cabi.py does exactly this on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tests/cabi_smoke.cpp -L facialmmt_amd -lfmmt_hip -Wl,-rpath,$PWD/facialmmt_amd -o /tmp/cabi_smoke
// Exit code 0 and a line "CABI_SMOKE_OK" on success; 1 and the failing check otherwise.  `--symbols-only` (no GPU needed) stops
// after resolving the entry points.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fmmt.h"

#define HIP_OK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::printf("HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static std::vector<float> rnd(size_t n, uint64_t seed, float scale) {
    std::vector<float> v(n);
    uint64_t s = seed;
    for (auto& x : v) x = ((float)(splitmix(s) >> 40) * (1.0f / 8388608.0f) - 1.0f) * scale;      // uniform [-scale, scale)
    return v;
}
template <typename T> static T* to_dev(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}
static bool close_to(const char* what, const std::vector<float>& got, const std::vector<double>& ref, double tol) {
    double worst = 0, scale = 1e-6;
    for (size_t i = 0; i < ref.size(); ++i) {
        scale = std::fmax(scale, std::fabs(ref[i]));
        worst = std::fmax(worst, std::fabs((double)got[i] - ref[i]));
        if (!std::isfinite(got[i])) worst = INFINITY;
    }
    const bool ok = worst <= tol * scale;
    std::printf("%s %-24s max|err| = %.3e  (scale %.3e, tol %.1e)\n", ok ? "OK  " : "FAIL", what, worst, scale, tol);
    return ok;
}

int main(int argc, char** argv) {
    if (fmmt_version() <= 0) {
        std::printf("fmmt_version() = %d\n", fmmt_version());
        return 1;
    }
    void* fns[] = {(void*)&fmmt_linear_fwd, (void*)&fmmt_layernorm_fwd, (void*)&fmmt_window_attn_fwd, (void*)&fmmt_window_block_fwd};
    for (void* f : fns)
        if (!f) return 1;
    if (argc > 1 && !std::strcmp(argv[1], "--symbols-only")) {
        std::printf("CABI_SYMBOLS_OK version %d\n", fmmt_version());
        return 0;
    }
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    bool ok = true;

    // ---------------------------------------------------------------- fmmt_linear_fwd: ragged M, bias, residual, per-sample scale
    {
        const int M = 203, N = 96, K = 96, RPS = 50;
        auto x = rnd((size_t)M * K, 1, 1.0f), w = rnd((size_t)N * K, 2, 0.1f), b = rnd(N, 3, 0.1f), res = rnd((size_t)M * N, 4, 1.0f);
        std::vector<float> rs((M + RPS - 1) / RPS);
        for (size_t i = 0; i < rs.size(); ++i) rs[i] = 0.5f + 0.25f * (float)i;
        float *dx = to_dev(x), *dw = to_dev(w), *db = to_dev(b), *dres = to_dev(res), *drs = to_dev(rs), *dy = nullptr;
        HIP_OK(hipMalloc(&dy, (size_t)M * N * 4));
        const int rc = fmmt_linear_fwd(FMMT_F32, M, N, K, dx, K, dw, K, db, dy, N, nullptr, 0, nullptr, N, dres, N, drs, RPS, st);
        if (rc) { std::printf("fmmt_linear_fwd rc = %d\n", rc); return 1; }
        HIP_OK(hipStreamSynchronize(st));
        std::vector<float> y((size_t)M * N);
        HIP_OK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> ref((size_t)M * N);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double a = b[n];
                for (int k = 0; k < K; ++k) a += (double)x[(size_t)m * K + k] * w[(size_t)n * K + k];
                ref[(size_t)m * N + n] = res[(size_t)m * N + n] + rs[m / RPS] * a;
            }
        ok &= close_to("fmmt_linear_fwd", y, ref, 2e-5);
    }

    // ---------------------------------------------------------------- fmmt_layernorm_fwd
    {
        const int M = 101, C = 96;
        auto x = rnd((size_t)M * C, 5, 2.0f), g = rnd(C, 6, 1.0f), b = rnd(C, 7, 0.5f);
        float *dx = to_dev(x), *dg = to_dev(g), *dbt = to_dev(b), *dy = nullptr, *dmean = nullptr, *drstd = nullptr;
        HIP_OK(hipMalloc(&dy, (size_t)M * C * 4));
        HIP_OK(hipMalloc(&dmean, M * 4));
        HIP_OK(hipMalloc(&drstd, M * 4));
        const int rc = fmmt_layernorm_fwd(FMMT_F32, M, C, dx, dg, dbt, 1e-5f, dy, dmean, drstd, 0, st);
        if (rc) { std::printf("fmmt_layernorm_fwd rc = %d\n", rc); return 1; }
        HIP_OK(hipStreamSynchronize(st));
        std::vector<float> y((size_t)M * C);
        HIP_OK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> ref((size_t)M * C);
        for (int m = 0; m < M; ++m) {
            double mean = 0, var = 0;
            for (int c = 0; c < C; ++c) mean += x[(size_t)m * C + c];
            mean /= C;
            for (int c = 0; c < C; ++c) var += (x[(size_t)m * C + c] - mean) * (x[(size_t)m * C + c] - mean);
            var /= C;
            for (int c = 0; c < C; ++c) ref[(size_t)m * C + c] = (x[(size_t)m * C + c] - mean) / std::sqrt(var + 1e-5) * g[c] + b[c];
        }
        ok &= close_to("fmmt_layernorm_fwd", y, ref, 2e-5);
    }

    // ---------------------------------------------------------------- fmmt_window_attn_fwd: 2 images of 14 x 14 tokens, C = 96, 3 heads, shift 3,
    // against the literal roll -> window_partition -> attention -> window_reverse -> roll sequence
    {
        const int n_img = 2, H = 14, W = 14, C = 96, nH = 3, hd = 32, shift = 3, WS = 7, T = 49, nWx = W / WS, nW = (H / WS) * nWx;
        auto qkv = rnd((size_t)n_img * H * W * 3 * C, 8, 1.0f), table = rnd((size_t)169 * nH, 9, 0.5f);
        std::vector<int32_t> index(T * T);
        for (int a = 0; a < T; ++a)
            for (int b = 0; b < T; ++b) index[a * T + b] = (a / WS - b / WS + WS - 1) * (2 * WS - 1) + (a % WS - b % WS + WS - 1);
        // SW-MSA mask (Swin_Transformer.py:208-227): region ids on the shifted grid
        auto region = [&](int v, int n) { return v < n - WS ? 0 : (v < n - shift ? 1 : 2); };
        std::vector<float> mask((size_t)nW * T * T);
        for (int w = 0; w < nW; ++w)
            for (int a = 0; a < T; ++a)
                for (int b = 0; b < T; ++b) {
                    const int ha = (w / nWx) * WS + a / WS, wa = (w % nWx) * WS + a % WS, hb = (w / nWx) * WS + b / WS, wb = (w % nWx) * WS + b % WS;
                    const bool same = region(ha, H) * 3 + region(wa, W) == region(hb, H) * 3 + region(wb, W);
                    mask[((size_t)w * T + a) * T + b] = same ? 0.f : -100.f;
                }
        float *dq = to_dev(qkv), *dt = to_dev(table), *dm = to_dev(mask), *dout = nullptr, *dlse = nullptr;
        int32_t* di = to_dev(index);
        HIP_OK(hipMalloc(&dout, (size_t)n_img * H * W * C * 4));
        HIP_OK(hipMalloc(&dlse, (size_t)n_img * nW * nH * T * 4));
        const float scale = 1.0f / std::sqrt((float)hd);
        const int rc = fmmt_window_attn_fwd(FMMT_F32, n_img, H, W, C, nH, shift, dq, dt, di, dm, nW, 1, scale, dout, dlse, st);
        if (rc) { std::printf("fmmt_window_attn_fwd rc = %d\n", rc); return 1; }
        HIP_OK(hipStreamSynchronize(st));
        std::vector<float> out((size_t)n_img * H * W * C);
        HIP_OK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> ref(out.size());
        for (int img = 0; img < n_img; ++img)
            for (int w = 0; w < nW; ++w)
                for (int h = 0; h < nH; ++h) {
                    // token of window slot a on the ROLLED grid -> original grid position
                    auto tok = [&](int a) {
                        const int hh = ((w / nWx) * WS + a / WS + shift) % H, ww = ((w % nWx) * WS + a % WS + shift) % W;
                        return (size_t)img * H * W + (size_t)hh * W + ww;
                    };
                    for (int a = 0; a < T; ++a) {
                        double s[49], mx = -1e300, den = 0;
                        for (int b = 0; b < T; ++b) {
                            double d = 0;
                            for (int e = 0; e < hd; ++e) d += (double)qkv[tok(a) * 3 * C + h * hd + e] * scale * qkv[tok(b) * 3 * C + C + h * hd + e];
                            s[b] = d + table[(size_t)index[a * T + b] * nH + h] + mask[((size_t)w * T + a) * T + b];
                            mx = std::fmax(mx, s[b]);
                        }
                        for (int b = 0; b < T; ++b) den += (s[b] = std::exp(s[b] - mx));
                        for (int e = 0; e < hd; ++e) {
                            double o = 0;
                            for (int b = 0; b < T; ++b) o += s[b] / den * qkv[tok(b) * 3 * C + 2 * C + h * hd + e];
                            ref[tok(a) * C + h * hd + e] = o;
                        }
                    }
                }
        ok &= close_to("fmmt_window_attn_fwd", out, ref, 2e-5);
    }
    std::printf(ok ? "CABI_SMOKE_OK\n" : "CABI_SMOKE_FAILED\n");
    return ok ? 0 : 1;
}
