// Development probe for tools/probes/gemm_pp.h (round 5; a rejected main loop, kept beside its probe -- not part of the library): the ping-pong NT kernel alone, checked against an fp32-accumulating reference kernel on the
// same bf16 operands, and timed with HIP events beside the production kernel (fmmt_linear_fwd of the in-tree libfmmt_hip.so, dlopen-ed).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I facialmmt_amd/csrc -I tools/probes -o /tmp/nt_pp_probe tools/probes/nt_pp_probe.hip -ldl && /tmp/nt_pp_probe
#include "gemm_pp.h"
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(bf16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = (bf16)(((float)(z >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f) * scale);
    }
}
__global__ void fillf_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        p[i] = (float)(z >> 40) * (1.0f / 16777216.0f) - 0.5f;
    }
}
// reference on sampled rows: out[r][n] = sum_k x[row_r][k] w[n][k] + bias[n] in fp32 (one thread per output)
__global__ void ref_kernel(const bf16* x, const bf16* w, const float* bias, const int* rows, int nrows, int N, int K, float* out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (n >= N || r >= nrows) return;
    const bf16* xr = x + (size_t)rows[r] * K;
    const bf16* wr = w + (size_t)n * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)xr[k] * (float)wr[k];
    out[(size_t)r * N + n] = s + (bias ? bias[n] : 0.f);
}

typedef int (*linear_fwd_t)(int, int, int, int, const void*, int, const void*, int, const float*, void*, int, void*, int, const void*, int, const void*, int,
                            const float*, int, void*);

template <typename F> float time_ms(F&& f, int n = 10, int reps = 3) {
    f(); CK(hipDeviceSynchronize());
    float best = 1e9f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < n; ++i) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / n);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

static double gelu_ref(double x) { return 0.5 * x * (1.0 + erf(x * 0.70710678118654752)); }

int main(int argc, char** argv) {
    const char* libpath = argc > 1 ? argv[1] : "facialmmt_amd/libfmmt_hip.so";
    void* h = dlopen(libpath, RTLD_NOW);
    linear_fwd_t prod = h ? (linear_fwd_t)dlsym(h, "fmmt_linear_fwd") : nullptr;
    if (!prod) printf("(production library not loaded: %s)\n", dlerror());
    struct Shape { int M, N, K; };
    const Shape shapes[] = {{125440, 1152, 384}, {125440, 384, 1536}, {31360, 768, 3072}};
    const size_t maxMK = 125440ull * 1536, maxNK = 3072ull * 768, maxMN = 501760ull * 768;
    bf16 *x, *w, *y, *y2, *ypre; float *bias, *ref, *trace; int* rows;
    CK(hipMalloc(&x, maxMK * 2)); CK(hipMalloc(&w, maxNK * 2 + 4096)); CK(hipMalloc(&y, maxMN * 2)); CK(hipMalloc(&y2, maxMN * 2)); CK(hipMalloc(&ypre, maxMN * 2));
    CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&trace, 256 * 8 * 8 * 4));
    const int NR = 64;
    CK(hipMalloc(&ref, (size_t)NR * 4096 * 4)); CK(hipMalloc(&rows, NR * 4));
    fillf_kernel<<<16, 256>>>(bias, 4096, 7u);
    for (const Shape& s : shapes) {
        const int M = s.M, N = s.N, K = s.K;
        fill_kernel<<<2048, 256>>>(x, (size_t)M * K, 1u, 1.0f);
        fill_kernel<<<256, 256>>>(w, (size_t)N * K, 2u, 1.0f / sqrtf((float)K));
        std::vector<int> hr(NR);
        for (int i = 0; i < NR; ++i) hr[i] = (int)(((long long)i * 2654435761ll) % M);
        hr[0] = 0; hr[1] = M - 1; hr[2] = 255; hr[3] = 256; hr[4] = M - 257 > 0 ? M - 257 : 0; hr[5] = 127; hr[6] = 128; hr[7] = 31; hr[8] = 32; hr[9] = 95;
        CK(hipMemcpy(rows, hr.data(), NR * 4, hipMemcpyHostToDevice));
        ref_kernel<<<dim3((N + 255) / 256, NR), 256>>>(x, w, bias, rows, NR, N, K, ref);
        std::vector<float> href((size_t)NR * N);
        CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
        LinArgs a{};
        a.M = M; a.N = N; a.K = K; a.x = x; a.ldx = K; a.w = w; a.ldw = K; a.bias = bias; a.y = y; a.ldy = N; a.y_pre = nullptr; a.epi = 0;
        a.part = trace;
        auto check = [&](const bf16* out, bool gelu, const char* tag) {
            std::vector<bf16> hy((size_t)N);
            double worst = 0; int bad = 0;
            for (int i = 0; i < NR; ++i) {
                CK(hipMemcpy(hy.data(), out + (size_t)hr[i] * N, (size_t)N * 2, hipMemcpyDeviceToHost));
                for (int n = 0; n < N; ++n) {
                    double r = href[(size_t)i * N + n];
                    if (gelu) r = gelu_ref((double)(float)(bf16)(float)r);
                    const double d = fabs((double)(float)hy[n] - r), tol = 0.01 * fabs(r) + (gelu ? 2e-3 : 4e-3);
                    if (d > tol) { if (bad < 3) printf("    %s mismatch row %d col %d: %f vs %f\n", tag, hr[i], n, (float)hy[n], r); ++bad; }
                    if (d > worst) worst = d;
                }
            }
            printf("    %-22s check: worst |diff| %.4g, %d bad of %d\n", tag, worst, bad, NR * N);
        };
        auto run_pp = [&](auto fn, const char* tag, bool gelu, bool stores) {
            CK(hipMemset(y, 0xff, (size_t)M * N * 2));
            int rc = fn(a);
            if (rc) { printf("  %s: launch rc %d\n", tag, rc); return; }
            CK(hipDeviceSynchronize());
            if (stores) check(y, gelu, tag);
            const float ms = time_ms([&] { fn(a); });
            printf("  pp  %7dx%5dx%5d %-26s %8.1f us %7.1f TF/s\n", M, N, K, tag, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
        };
        printf("shape %d x %d x %d\n", M, N, K);
        if (prod) {
            auto pf = [&](bool gelu) { return prod(1, M, N, K, x, K, w, K, bias, y2, N, gelu ? ypre : nullptr, gelu ? 1 : 0, nullptr, 0, nullptr, 0, nullptr, 1, nullptr); };
            int rc = pf(false); CK(hipDeviceSynchronize());
            if (rc) printf("  production rc %d\n", rc);
            check(y2, false, "production");
            const float ms = time_ms([&] { pf(false); });
            printf("  production plain                                   %8.1f us %7.1f TF/s\n", ms * 1e3, 2.0 * M * N * K / ms / 1e9);
            if (N == 4 * K) {
                const float msg = time_ms([&] { pf(true); });
                printf("  production gelu+pre                                %8.1f us\n", msg * 1e3);
            }
        }
        auto stamps = [&](auto fn) {                            // segment stamps: cycles per K step (32 deep), mean over the waves of each group
            CK(hipMemset(trace, 0, 256 * 8 * 8 * 4));
            fn(a); CK(hipDeviceSynchronize());
            std::vector<float> ht(256 * 8 * 8);
            CK(hipMemcpy(ht.data(), trace, ht.size() * 4, hipMemcpyDeviceToHost));
            for (int g = 0; g < 2; ++g) {
                double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int b = 0; b < 256; ++b) for (int wv = g * 4; wv < g * 4 + 4; ++wv) for (int i = 0; i < 8; ++i) t[i] += ht[(b * 8 + wv) * 8 + i];
                const double ns = t[7];
                printf("    stamps group %d, cycles per K step: LOAD [setup+park %.0f | read issue %.0f | lgkm wait %.0f | vm wait %.0f] barrier %.0f | MFMA segment %.0f | barrier %.0f   (%.0f steps per wave)\n",
                       g, t[0] / ns, t[1] / ns, t[2] / ns, t[3] / ns, t[4] / ns, t[5] / ns, t[6] / ns, ns / 1024);
            }
        };
        if (N % 192 == 0 && K >= 192) {
            run_pp([&](const LinArgs& q) { return launch_pp<192, 4, 1>(q, 0); }, "BN192", false, true);
            run_pp([&](const LinArgs& q) { return launch_pp<192, 4, 0>(q, 0); }, "BN192 nostore", false, false);
            run_pp([&](const LinArgs& q) { return launch_pp<192, 4, 0, false, true, true, true>(q, 0); }, "BN192 nostore nodma", false, false);
            run_pp([&](const LinArgs& q) { return launch_pp<192, 4, 1, false, true, true, true>(q, 0); }, "BN192 nodma", false, false);
            stamps([&](const LinArgs& q) { return launch_pp<192, 4, 0, true>(q, 0); });
            stamps([&](const LinArgs& q) { return launch_pp<192, 4, 0, true, true, true, true>(q, 0); });
            stamps([&](const LinArgs& q) { return launch_pp<192, 4, 1, true>(q, 0); });
            if (N == 4 * K) {
                a.epi = FMMT_EPI_GELU; a.y_pre = ypre;
                run_pp([&](const LinArgs& q) { return launch_pp<192, 4, 2>(q, 0); }, "BN192 gelu+pre", true, true);
                stamps([&](const LinArgs& q) { return launch_pp<192, 4, 2, true>(q, 0); });
                a.epi = 0; a.y_pre = nullptr;
            }
        }
        if (N % 128 == 0 && K >= 192) {
            run_pp([&](const LinArgs& q) { return launch_pp<128, 4, 1>(q, 0); }, "BN128", false, true);
        }
        fflush(stdout);
    }
    return 0;
}
