// Development probe for csrc/gemm_ph.h (round 5): the four-phase NT kernel alone, checked against an fp32-accumulating reference kernel on the
// same bf16 operands, and timed with HIP events beside the production kernel (fmmt_linear_fwd of the in-tree libfmmt_hip.so, dlopen-ed).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I facialmmt_amd/csrc -o /tmp/nt_ph_probe tools/probes/nt_ph_probe.hip -ldl && /tmp/nt_ph_probe
#include "gemm_ph.h"
#include "gemm_ph3.h"
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

__global__ void cmp_kernel(const bf16* a, const bf16* b, size_t n, unsigned long long* bad) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = (float)a[i], y = (float)b[i];
        if (!(fabsf(x - y) <= 0.02f * fabsf(y) + 8e-3f)) ++c;
    }
    if (c) atomicAdd(bad, c);
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
    const Shape shapes[] = {{125440, 1152, 384}, {125440, 384, 1536}, {125440, 384, 384}, {125440, 384, 1152}, {125440, 1536, 384}, {31360, 768, 3072}, {31360, 2304, 768}};
    const size_t maxMK = 125440ull * 1536, maxNK = 8192ull * 8192, maxMN = 501760ull * 768;
    bf16 *x, *w, *y, *y2; float *bias, *ref; int* rows;
    CK(hipMalloc(&x, maxMK * 2)); CK(hipMalloc(&w, maxNK * 2 + 4096)); CK(hipMalloc(&y, maxMN * 2)); CK(hipMalloc(&y2, maxMN * 2));
    CK(hipMalloc(&bias, 8192 * 4));
    const int NR = 64;
    CK(hipMalloc(&ref, (size_t)NR * 8192 * 4)); CK(hipMalloc(&rows, NR * 4));
    fillf_kernel<<<16, 256>>>(bias, 8192, 7u);
    for (const Shape& s : shapes) {
        const int M = s.M, N = s.N, K = s.K;
        fill_kernel<<<2048, 256>>>(x, (size_t)M * K, 1u, 1.0f);
        fill_kernel<<<256, 256>>>(w, (size_t)N * K, 2u, 1.0f / sqrtf((float)K));
        std::vector<int> hr(NR);
        for (int i = 0; i < NR; ++i) hr[i] = (int)(((long long)i * 2654435761ll) % M);
        hr[0] = 0; hr[1] = M - 1; hr[2] = 255; hr[3] = 256; hr[4] = M - 257 > 0 ? M - 257 : 0; hr[5] = 127; hr[6] = 128; hr[7] = 31; hr[8] = 32; hr[9] = 95; hr[10] = 63; hr[11] = 64; hr[12] = 191; hr[13] = 192;
        CK(hipMemcpy(rows, hr.data(), NR * 4, hipMemcpyHostToDevice));
        ref_kernel<<<dim3((N + 255) / 256, NR), 256>>>(x, w, bias, rows, NR, N, K, ref);
        std::vector<float> href((size_t)NR * N);
        CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
        LinArgs a{};
        a.M = M; a.N = N; a.K = K; a.x = x; a.ldx = K; a.w = w; a.ldw = K; a.bias = bias; a.y = y; a.ldy = N; a.y_pre = nullptr; a.epi = 0;
        auto check = [&](const bf16* out, const char* tag) {
            std::vector<bf16> hy((size_t)N);
            double worst = 0; int bad = 0;
            for (int i = 0; i < NR; ++i) {
                CK(hipMemcpy(hy.data(), out + (size_t)hr[i] * N, (size_t)N * 2, hipMemcpyDeviceToHost));
                for (int n = 0; n < N; ++n) {
                    const double r = href[(size_t)i * N + n];
                    const double d = fabs((double)(float)hy[n] - r), tol = 0.01 * fabs(r) + 4e-3;
                    if (d > tol) { if (bad < 3) printf("    %s mismatch row %d col %d: %f vs %f\n", tag, hr[i], n, (float)hy[n], r); ++bad; }
                    if (d > worst) worst = d;
                }
            }
            printf("    %-22s check: worst |diff| %.4g, %d bad of %d\n", tag, worst, bad, NR * N);
        };
        auto run = [&](auto fn, const char* tag, bool stores) {
            CK(hipMemset(y, 0xff, (size_t)M * N * 2));
            int rc = fn(a);
            if (rc) { printf("  %s: launch rc %d\n", tag, rc); return; }
            CK(hipDeviceSynchronize());
            if (stores) for (int rep = 0; rep < 2; ++rep) { check(y, tag); if (rep == 0) { CK(hipMemset(y, 0xff, (size_t)M * N * 2)); fn(a); CK(hipDeviceSynchronize()); } }
            const float ms = time_ms([&] { fn(a); });
            printf("  ph  %7dx%5dx%5d %-26s %8.1f us %7.1f TF/s\n", M, N, K, tag, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
        };
        printf("shape %d x %d x %d\n", M, N, K);
        if (prod) {
            auto pf = [&]() { return prod(1, M, N, K, x, K, w, K, bias, y2, N, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 1, nullptr); };
            int rc = pf(); CK(hipDeviceSynchronize());
            if (rc) printf("  production rc %d\n", rc);
            check(y2, "production");
            const float ms = time_ms([&] { pf(); });
            printf("  production plain                                   %8.1f us %7.1f TF/s\n", ms * 1e3, 2.0 * M * N * K / ms / 1e9);
        }
        const bool n256 = N % 256 == 0;
        if (n256) run([&](const LinArgs& q) { return launch_ph<2>(q, 0); }, "ph drip nt bias", true);
        if (n256) run([&](const LinArgs& q) { return launch_ph3<2>(q, 0); }, "ph3 (192 x 256) drip bias", true);
        if (n256) run([&](const LinArgs& q) { return launch_ph3<0>(q, 0); }, "ph3 (192 x 256) nostore", false);
        run([&](const LinArgs& q) { return launch_ph3<2, true, 4>(q, 0); }, "ph3 (384 x 128) drip bias", true);
        run([&](const LinArgs& q) { return launch_ph3<0, true, 4>(q, 0); }, "ph3 (384 x 128) nostore", false);
        if (prod) {
            unsigned long long* dbad; CK(hipMalloc(&dbad, 8));
            { auto pf = [&]() { return prod(1, M, N, K, x, K, w, K, bias, y2, N, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 1, nullptr); }; pf(); CK(hipDeviceSynchronize()); }
            unsigned long long tot = 0;
            for (int rep = 0; rep < 8; ++rep) {
                CK(hipMemset(y, 0xff, (size_t)M * N * 2)); CK(hipMemset(dbad, 0, 8));
                launch_ph3<2, true, 4>(a, 0);
                cmp_kernel<<<1024, 256>>>(y, y2, (size_t)M * N, dbad);
                unsigned long long hb; CK(hipMemcpy(&hb, dbad, 8, hipMemcpyDeviceToHost));
                tot += hb;
            }
            printf("    screen ph3 (384 x 128) drip bias      : %llu elements differ over 8 launches\n", tot);
            CK(hipFree(dbad));
        }
        if (prod) {                                             // ph3's other epilogues against the production kernels: whole outputs, 4 launches each
            bf16 *t1, *t2, *opd; float* rsd;
            CK(hipMalloc(&t1, (size_t)M * N * 2)); CK(hipMalloc(&t2, (size_t)M * N * 2)); CK(hipMalloc(&opd, (size_t)M * N * 2)); CK(hipMalloc(&rsd, (size_t)(M / 49 + 2) * 4));
            fill_kernel<<<2048, 256>>>(opd, (size_t)M * N, 5u, 1.0f);
            fillf_kernel<<<64, 256>>>(rsd, (size_t)(M / 49 + 2), 9u);
            unsigned long long* dbad; CK(hipMalloc(&dbad, 8));
            auto cmp = [&](const bf16* u, const bf16* v) { unsigned long long hb; CK(hipMemset(dbad, 0, 8)); cmp_kernel<<<1024, 256>>>(u, v, (size_t)M * N, dbad); CK(hipMemcpy(&hb, dbad, 8, hipMemcpyDeviceToHost)); return hb; };
            auto variant = [&](const char* tag, auto pf, auto qf, bool two) {
                pf(); CK(hipDeviceSynchronize());
                const float msp = time_ms([&] { pf(); });
                unsigned long long tot = 0, totp = 0;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipMemset(y, 0xff, (size_t)M * N * 2)); CK(hipMemset(t1, 0xff, (size_t)M * N * 2));
                    int rc = qf(); if (rc) { printf("  %s: rc %d\n", tag, rc); return; }
                    tot += cmp(y, y2);
                    if (two) totp += cmp(t1, t2);
                }
                const float msq = time_ms([&] { qf(); });
                printf("  ph3 %-18s production %8.1f us, ph3 %8.1f us; over 4 launches: y %llu%s elements differ\n", tag, msp * 1e3, msq * 1e3, tot, two ? (std::string(", y_pre ") + std::to_string(totp)).c_str() : "");
            };
            LinArgs q = a;
            q.epi = FMMT_EPI_GELU; q.y_pre = t1;
            variant("gelu+pre", [&] { return prod(1, M, N, K, x, K, w, K, bias, y2, N, t2, 1, nullptr, 0, nullptr, 0, nullptr, 1, nullptr); }, [&] { return launch_ph3<3, true, 4>(q, 0); }, true);
            q = a; q.bias = nullptr; q.epi = FMMT_EPI_GELU_BWD; q.aux = opd; q.ldaux = N;
            variant("gelu'", [&] { return prod(1, M, N, K, x, K, w, K, nullptr, y2, N, nullptr, 2, opd, N, nullptr, 0, nullptr, 1, nullptr); }, [&] { return launch_ph3<4, true, 4>(q, 0); }, false);
            q = a; q.res = opd; q.ldres = N; q.rowscale = rsd; q.rows_per_scale = 49;
            variant("res+rowscale", [&] { return prod(1, M, N, K, x, K, w, K, bias, y2, N, nullptr, 0, nullptr, 0, opd, N, rsd, 49, nullptr); }, [&] { return launch_ph3<5, true, 4>(q, 0); }, false);
            q = a; q.bias = nullptr; q.rowscale = rsd; q.rows_per_scale = 49;
            variant("rowscale only", [&] { return prod(1, M, N, K, x, K, w, K, nullptr, y2, N, nullptr, 0, nullptr, 0, nullptr, 0, rsd, 49, nullptr); }, [&] { return launch_ph3<5, true, 4>(q, 0); }, false);
            CK(hipFree(dbad)); CK(hipFree(t1)); CK(hipFree(t2)); CK(hipFree(opd)); CK(hipFree(rsd));
        }
        fflush(stdout);
    }
    return 0;
}
