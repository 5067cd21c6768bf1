// VALU issue-rate probe (gfx950): how many cycles a wave64 instruction of each kind occupies a SIMD, at 1 / 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/probes/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int ITER = 2048, UNR = 32;

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, float c0, float c1) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = c0;
    __syncthreads();
    f32x16 big[2] = {};
    f32x4 ld4 = {};
    const unsigned laddr = (unsigned)(threadIdx.x & 63) * 16u;
    f32x2 a[UNR], u = {c0 + threadIdx.x * 1e-9f, c0};
    f32x4 acc[4] = {};
    bf16x8 bx = {}, by = {};
    const f32x2 cc = {__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c1))), 0.f};   // an SGPR pair
#pragma unroll
    for (int i = 0; i < UNR; ++i) a[i] = f32x2{(float)i, c1};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(u));                 // packed, VGPR operands
            if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(u), "s"(cc));   // packed, broadcast SGPR addend
            if constexpr (KIND == 2) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i].x) : "v"(u.x)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i].y) : "v"(u.y)); }   // two scalar FMAs
            if constexpr (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
            if constexpr (KIND == 4) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(u.x), "v"(u.y));
            if constexpr (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a[i].x) : "v"(a[i].y), "v"(u.y));
            if constexpr (KIND == 6) {                                                                                         // 1 MFMA : 8 packed FMAs
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(u));
                if ((i & 7) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i >> 3) & 3]) : "v"(bx), "v"(by));
            }
            if constexpr (KIND == 13) {                                                                                        // 1 MFMA : 4 packed FMAs
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(u));
                if ((i & 3) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i >> 2) & 3]) : "v"(bx), "v"(by));
            }
            if constexpr (KIND == 8) {                                                                                         // 1 MFMA 32x32x16 : 8 packed FMAs
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(u));
                if ((i & 7) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[(i >> 3) & 1]) : "v"(bx), "v"(by));
            }
            if constexpr (KIND == 9) { if ((i & 3) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[(i >> 2) & 1]) : "v"(bx), "v"(by)); }
            if constexpr (KIND == 10) {                                                                                        // 1 ds_read_b128 : 4 packed FMAs
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(u));
                if ((i & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld4) : "v"(laddr) : "memory");
            }
            if constexpr (KIND == 11) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(u.x), "v"(u.y));
            if constexpr (KIND == 12) {                                                                                        // 1 MFMA : 8 scalar VOP2 FMAs
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(u.x), "v"(u.y));
                if ((i & 7) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i >> 3) & 3]) : "v"(bx), "v"(by));
            }
            if constexpr (KIND == 7) { if ((i & 3) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i >> 2) & 3]) : "v"(bx), "v"(by)); }   // MFMAs alone
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < UNR; ++i) s += a[i].x + a[i].y;
    for (int j = 0; j < 4; ++j) s += acc[j][0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    s += big[0][0] + big[1][3] + ld4[0];
    if (s == 123.456f) out[0] = s;
}

template <int KIND>
void run(const char* name, int per_iter, int waves_per_simd) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 64 * 4 * waves_per_simd;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, 0.5f, 0.25f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9;                                     // at the nominal 2.4 GHz
    const double inst = (double)ITER * per_iter * waves_per_simd;            // instructions per SIMD
    printf("%-44s %d wave(s)/SIMD: %7.3f ms  %6.2f cycles per wave-instruction (per SIMD)\n", name, waves_per_simd, ms, cyc / inst);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_pk_fma_f32 (VGPR operands)", UNR, w);
        run<1>("v_pk_fma_f32 (SGPR addend, op_sel_hi)", UNR, w);
        run<2>("v_fma_f32 x2", 2 * UNR, w);
        run<3>("v_pk_mul_f32", UNR, w);
        run<4>("v_med3_f32", UNR, w);
        run<5>("v_cvt_pk_bf16_f32", UNR, w);
        run<6>("8 v_pk_fma_f32 + 1 MFMA 16x16x32 (count: pk_fma)", UNR, w);
        run<7>("MFMA 16x16x32 bf16 alone (count: MFMA)", UNR / 4, w);
        run<8>("8 v_pk_fma_f32 + 1 MFMA 32x32x16 (count: pk_fma)", UNR, w);
        run<9>("MFMA 32x32x16 bf16 alone (count: MFMA)", UNR / 4, w);
        run<10>("4 v_pk_fma_f32 + 1 ds_read_b128 (count: pk_fma)", UNR, w);
        run<11>("v_fmac_f32 (VOP2)", UNR, w);
        run<12>("8 v_fmac_f32 + 1 MFMA 16x16x32 (count: fmac)", UNR, w);
        run<13>("4 v_pk_fma_f32 + 1 MFMA 16x16x32 (count: pk_fma)", UNR, w);
    }
    return 0;
}
