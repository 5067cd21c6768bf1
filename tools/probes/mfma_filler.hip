// Issue-slot probe, round 5: how many PLAIN single-issue VALU instructions fit into the gap of one MFMA on a gfx950 SIMD?
//   k fillers (v_fma_f32 / v_mul_f32 / v_cvt_pk_bf16_f32 / v_exp_f32 / v_pk_fma_f32 for reference) hand-placed behind every MFMA
//   (v_mfma_f32_32x32x16_bf16 or v_mfma_f32_16x16x32_bf16), k = 0, 2..8, at 1 and 2 waves per SIMD, independent register chains.
//   Cycles are SHADER cycles from s_memtime inside the kernel (not wall time at a nominal clock: the round-4 table could not tell a slower
//   clock from a longer stream), averaged over the waves of 256 workgroups; wall time printed beside it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_filler tools/probes/mfma_filler.hip && /tmp/mfma_filler
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int ITER = 1024, NM = 8;                              // MFMAs per loop body

enum { F_FMA = 0, F_MUL = 1, F_CVT = 2, F_EXP = 3, F_PKFMA = 4 };

template <int KIND>
__device__ __forceinline__ void filler(float& r, float& r2, float u) {
    if constexpr (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r) : "v"(u));
    if constexpr (KIND == F_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(u));
    if constexpr (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(r2), "v"(u));
    if constexpr (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
    if constexpr (KIND == F_PKFMA) {
        f32x2 t = {r, r2};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(t) : "v"(f32x2{u, u}));
        r = t.x; r2 = t.y;
    }
}

// MF: 0 = no MFMA (fillers alone: K * NM of them per body), 1 = 16x16x32, 2 = 32x32x16
template <int MF, int KIND, int K>
__global__ __launch_bounds__(512) void probe(unsigned long long* cyc, float* out, float c0) {
    float a[32], b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { a[i] = c0 + i; b[i] = c0 * i; }
    const float u = c0 + threadIdx.x * 1e-9f;
    f32x4 acc4[4] = {};
    f32x16 acc16[4] = {};
    bf16x8 bx = {}, by = {};
    asm volatile("" : "+v"(bx), "+v"(by));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long s0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if constexpr (MF == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[m & 3]) : "v"(bx), "v"(by));
            if constexpr (MF == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16[m & 3]) : "v"(bx), "v"(by));
#pragma unroll
            for (int j = 0; j < K; ++j) filler<KIND>(a[(m * K + j) & 31], b[(m * K + j) & 31], u);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long s1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i] + b[i];
    for (int j = 0; j < 4; ++j) s += acc4[j][0] + acc16[j][5];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s1 - s0;
}

struct Res { double cyc, ms; };
template <int MF, int KIND, int K>
Res run(int wps) {
    static unsigned long long* cyc = nullptr; static float* out = nullptr;
    if (!cyc) { hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&out, 4); }
    const int threads = 64 * 4 * wps, nw = 256 * 4 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MF, KIND, K>), dim3(256), dim3(threads), 0, 0, cyc, out, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MF, KIND, K>), dim3(256), dim3(threads), 0, 0, cyc, out, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return {sum / nw / ((double)ITER * NM), (double)ms};          // cycles per (MFMA + its K fillers) per WAVE
}

template <int MF, int KIND>
void sweep(const char* mname, const char* fname) {
    for (int wps = 1; wps <= 2; ++wps) {
        Res r[9];
        r[0] = run<MF, KIND, 0>(wps); r[2] = run<MF, KIND, 2>(wps); r[3] = run<MF, KIND, 3>(wps); r[4] = run<MF, KIND, 4>(wps);
        r[5] = run<MF, KIND, 5>(wps); r[6] = run<MF, KIND, 6>(wps); r[7] = run<MF, KIND, 7>(wps); r[8] = run<MF, KIND, 8>(wps);
        // per SIMD: wps waves share it, so cycles per MFMA-slot on the SIMD = wave cycles / wps
        printf("%-10s + k x %-18s %d wave/SIMD | wave-cycles per group, k=0,2..8: %6.1f |", mname, fname, wps, r[0].cyc);
        for (int k = 2; k <= 8; ++k) printf(" %6.1f", r[k].cyc);
        printf(" | SIMD-cycles per MFMA:");
        printf(" %5.1f |", r[0].cyc / wps);
        for (int k = 2; k <= 8; ++k) printf(" %5.1f", r[k].cyc / wps);
        printf(" | wall ms k=0/4/8: %.3f %.3f %.3f\n", r[0].ms, r[4].ms, r[8].ms);
    }
}

template <int KIND>
void alone(const char* fname) {                                  // the fillers without MFMAs: cost of k fillers per group of the same loop
    for (int wps = 1; wps <= 2; ++wps) {
        Res r4 = run<0, KIND, 4>(wps), r8 = run<0, KIND, 8>(wps);
        printf("alone      k x %-18s %d wave/SIMD | wave-cycles per instruction: k=4 %5.2f  k=8 %5.2f | SIMD-cycles per instruction: %5.2f %5.2f\n", fname, wps,
               r4.cyc / 4, r8.cyc / 8, r4.cyc / 4 / wps, r8.cyc / 8 / wps);
    }
}

int main() {
    alone<F_FMA>("v_fma_f32"); alone<F_MUL>("v_mul_f32"); alone<F_CVT>("v_cvt_pk_bf16_f32"); alone<F_EXP>("v_exp_f32"); alone<F_PKFMA>("v_pk_fma_f32");
    sweep<2, F_FMA>("32x32x16", "v_fma_f32"); sweep<2, F_MUL>("32x32x16", "v_mul_f32"); sweep<2, F_CVT>("32x32x16", "v_cvt_pk_bf16_f32");
    sweep<2, F_EXP>("32x32x16", "v_exp_f32"); sweep<2, F_PKFMA>("32x32x16", "v_pk_fma_f32");
    sweep<1, F_FMA>("16x16x32", "v_fma_f32"); sweep<1, F_MUL>("16x16x32", "v_mul_f32"); sweep<1, F_CVT>("16x16x32", "v_cvt_pk_bf16_f32");
    sweep<1, F_EXP>("16x16x32", "v_exp_f32"); sweep<1, F_PKFMA>("16x16x32", "v_pk_fma_f32");
    return 0;
}
