// How fast can one CU fill LDS from L2 with global_load_lds (gfx950)?  The persistent NT GEMMs of this repo (256-token x 192-channel tiles) are all
// bound near 30 GB/s per CU of operand traffic; this probe issues exactly a GEMM's DMA stream -- same tile walk, same rows, a ring of stages, counted
// vmcnt, one barrier per stage -- and NOTHING else (no fragment reads, no MFMAs), for
//   ROWB = 64 : stage rows of 64 B (32-deep K steps: a 1 KB piece = 16 rows x half a cache line)
//   ROWB = 128: stage rows of 128 B (64-deep K steps: a 1 KB piece = 8 rows x one full line)
// and D = stages in flight.  Prints GB/s per CU and the chip total.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_rate tools/probes/dma_rate.hip && /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// x [M][K] bf16, w [N][K] bf16; tile = 256 tokens x BN channels; stage = (BN + 256) rows x ROWB bytes
template <int ROWB, int D, int BN, int NWAVES, bool BARRIER>
__global__ __launch_bounds__(512) void dma_kernel(const char* x, const char* w, int M, int N, int K, int tiles_n, int total, unsigned* sink) {
    constexpr int ROWS = BN + 256, RPP = 1024 / ROWB, NP = ROWS / RPP, STAGE = ROWS * ROWB, NBUF = (D + 1) * STAGE <= 160 * 1024 ? D + 1 : 160 * 1024 / STAGE;   // (a rate probe: slots may be overwritten while in flight)
    constexpr int PPW = (NP + NWAVES - 1) / NWAVES;            // pieces per issuing wave and stage (the last may be absent)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x;
    const int first = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    const int nk = K * 2 / ROWB;
    const int rl = lane / (ROWB / 16), cl = lane % (ROWB / 16);
    int slot = 0;
    for (int t = first; t < total; t += G) {
        const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * BN;
        for (int k = 0; k < nk; ++k) {
            if (wave < NWAVES) {
#pragma unroll
                for (int i = 0; i < PPW; ++i) {
                    const int j = i * NWAVES + wave;
                    if (j < NP) {
                        const int row0 = j * RPP;
                        const char* src = row0 < BN ? w + ((size_t)(n0 + row0 + rl) * K) * 2 : x + ((size_t)min(m0 + row0 - BN + rl, M - 1) * K) * 2;
                        __builtin_amdgcn_global_load_lds((gptr_t*)(src + (size_t)k * ROWB + cl * 16), (lptr_t*)(smem + slot * STAGE + row0 * ROWB), 16, 0, 0);
                    }
                }
                // at most D - 1 younger stages stay in flight (absent last pieces: the count is an upper bound, i.e. a laxer wait by one -- fine for a rate probe)
                wait_vm<(D - 1) * PPW>();
            }
            if (BARRIER) __builtin_amdgcn_s_barrier();
            slot = slot + 1 == NBUF ? 0 : slot + 1;
        }
    }
    wait_vm<0>();
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = (unsigned)smem[0];
}

template <int ROWB, int D, int BN, int NWAVES, bool BARRIER>
void run(const char* x, const char* w, int M, int N, int K, unsigned* sink) {
    constexpr int STAGE = (BN + 256) * ROWB, LDS = ((D + 1) * STAGE <= 160 * 1024 ? D + 1 : 160 * 1024 / STAGE) * STAGE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<ROWB, D, BN, NWAVES, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int tiles_n = N / BN, total = ((M + 255) / 256) * tiles_n;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((dma_kernel<ROWB, D, BN, NWAVES, BARRIER>), dim3(256), dim3(512), LDS, 0, x, w, M, N, K, tiles_n, total, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms / 5 < best) best = ms / 5;
    }
    const double bytes = (double)total * (BN + 256) * K * 2;
    printf("  %7d x %5d x %5d  rows of %3d B, %d stage(s) of %5.1f KB in flight, %d issuing waves, barrier %d: %8.1f us  %6.2f TB/s chip  %6.1f GB/s per CU  (a GEMM at this rate: %6.0f TF/s)\n",
           M, N, K, ROWB, D, STAGE / 1024.0, NWAVES, (int)BARRIER, best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / 256, 2.0 * M * N * K / best / 1e9);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    const size_t xb = 125440ull * 1536 * 2, wb = 3072ull * 1536 * 2;
    char *x, *w; unsigned* sink;
    CK(hipMalloc(&x, xb)); CK(hipMalloc(&w, wb)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(x, 1, xb)); CK(hipMemset(w, 1, wb));
    struct S { int M, N, K; } shapes[] = {{125440, 1152, 384}, {125440, 384, 1536}, {31360, 768, 3072}};
    for (auto s : shapes) {
        run<64, 3, 192, 8, true>(x, w, s.M, s.N, s.K, sink);
        run<64, 4, 192, 8, true>(x, w, s.M, s.N, s.K, sink);
        run<64, 3, 192, 8, false>(x, w, s.M, s.N, s.K, sink);
        run<64, 3, 192, 4, true>(x, w, s.M, s.N, s.K, sink);
        run<64, 6, 192, 8, true>(x, w, s.M, s.N, s.K, sink);
        run<128, 1, 192, 8, true>(x, w, s.M, s.N, s.K, sink);
        run<128, 2, 192, 8, true>(x, w, s.M, s.N, s.K, sink);
        run<128, 3, 192, 8, true>(x, w, s.M, s.N, s.K, sink);
        run<128, 2, 192, 8, false>(x, w, s.M, s.N, s.K, sink);
        run<128, 2, 192, 4, true>(x, w, s.M, s.N, s.K, sink);
        run<128, 2, 256, 8, true>(x, w, s.M, s.N / 256 * 256, s.K, sink);
        run<64, 3, 256, 8, true>(x, w, s.M, s.N / 256 * 256, s.K, sink);
    }
    return 0;
}
