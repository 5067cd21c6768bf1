// HBM bandwidth calibration on gfx950 (development aid, not part of the library):
// fill / read / copy / 3:1 read:write / row-span writes like the GEMM epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_fill(f4* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  f4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n; i += st) p[i] = v;
}
__global__ void k_fill_nt(f4* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  f4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n; i += st) __builtin_nontemporal_store(v, p + i);
}
__global__ void k_read(const f4* __restrict__ p, size_t n, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  f4 a = {0, 0, 0, 0};
  for (; i < n; i += st) a += p[i];
  if (a.x + a.y + a.z + a.w == 12345.f) *out = 1.f;
}
__global__ void k_copy(const f4* __restrict__ s, f4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) d[i] = s[i];
}
__global__ void k_r3w1(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, f4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) d[i] = a[i] + b[i] * c[i];
}
// each lane writes SPAN bytes contiguous (SPAN/16 stores); a group of 16 lanes covers 16 consecutive rows of
// pitch `pitch` bytes; 4 groups of the wave cover 4 column spans (like the NT epilogue: token rows, 32-channel spans)
template <int SPAN>
__global__ void k_rowspan(char* __restrict__ p, size_t rows, int pitch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int spans_per_row = pitch / SPAN;
  size_t tile = (size_t)blockIdx.x * 4 + wave;             // tile = 16 rows x 4 spans
  size_t tiles_per_rowblock = spans_per_row / 4;
  size_t ntiles = rows / 16 * tiles_per_rowblock;
  f4 v = {1.f, 2.f, 3.f, 4.f};
  for (; tile < ntiles; tile += (size_t)gridDim.x * 4) {
    size_t rb = tile / tiles_per_rowblock, cb = tile % tiles_per_rowblock;
    char* q = p + (rb * 16 + (lane & 15)) * (size_t)pitch + (cb * 4 + (lane >> 4)) * SPAN;
#pragma unroll
    for (int j = 0; j < SPAN / 16; ++j) *(f4*)(q + j * 16) = v;
  }
}

// GEMM-epilogue-shaped stores: one workgroup per 128-row x 128-channel bf16 tile (256 B of a `pitch`-byte row)
// PAT 0: what nt_epilogue does (per instruction 16 rows x 64 B)   PAT 1: via-LDS shape (per instruction 4 rows x 256 B)
// PAT 2: workgroup owns 128 full rows (per instruction 1 KB contiguous)
template <int PAT>
__global__ void k_tile(char* __restrict__ p, int tiles_n, int pitch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  f4 v = {1.f, 2.f, 3.f, 4.f};
  if (PAT == 2) {
    char* base = p + (size_t)blockIdx.x * 128 * pitch;
    for (int o = tid * 16; o < 128 * pitch; o += 4096) *(f4*)(base + o) = v;
    return;
  }
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  char* base = p + (size_t)tm * 128 * pitch + tn * 256;
  if (PAT == 0) {
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) *(f4*)(base + (size_t)(wm * 64 + a * 16 + li) * pitch + wn * 128 + c * 64 + lg * 16) = v;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) *(f4*)(base + (size_t)(i * 16 + (tid >> 4)) * pitch + (tid & 15) * 16) = v;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <class F> float timeit(F f, int it = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}
int main() {
  const size_t bytes = (size_t)3 << 30;   // 3 GiB per buffer
  const size_t n = bytes / 16;
  f4 *a, *b, *c, *d; float* o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 4));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes));
  for (int grid : {2048, 8192, 65536}) {
    float t;
    t = timeit([&] { k_fill<<<grid, 256>>>(d, n); });            printf("grid %6d fill      %7.0f GB/s\n", grid, bytes / t / 1e6);
    t = timeit([&] { k_fill_nt<<<grid, 256>>>(d, n); });         printf("grid %6d fill_nt   %7.0f GB/s\n", grid, bytes / t / 1e6);
    t = timeit([&] { k_read<<<grid, 256>>>(a, n, o); });         printf("grid %6d read      %7.0f GB/s\n", grid, bytes / t / 1e6);
    t = timeit([&] { k_copy<<<grid, 256>>>(a, d, n); });         printf("grid %6d copy      %7.0f GB/s (r+w)\n", grid, 2.0 * bytes / t / 1e6);
    t = timeit([&] { k_r3w1<<<grid, 256>>>(a, b, c, d, n); });   printf("grid %6d r3w1      %7.0f GB/s (r+w)\n", grid, 4.0 * bytes / t / 1e6);
  }
  for (int pitch : {192, 768, 1536}) {
    size_t rows = bytes / pitch / 16 * 16;
    float t;
    if (pitch % 256 == 0) { t = timeit([&] { k_rowspan<64><<<8192, 256>>>((char*)d, rows, pitch); }); printf("rowspan64  pitch %5d  %7.0f GB/s\n", pitch, rows * (double)pitch / t / 1e6); }
    if (pitch % 128 == 0) { t = timeit([&] { k_rowspan<32><<<8192, 256>>>((char*)d, rows, pitch); }); printf("rowspan32  pitch %5d  %7.0f GB/s\n", pitch, rows * (double)pitch / t / 1e6); }
    t = timeit([&] { k_rowspan<16><<<8192, 256>>>((char*)d, rows, pitch); }); printf("rowspan16  pitch %5d  %7.0f GB/s\n", pitch, rows * (double)pitch / t / 1e6);
  }
  for (int pitch : {768, 192 * 3, 1536, 3072}) {
    int rows = (int)(((size_t)1540 << 20) / pitch / 128 * 128);
    int tiles_n = pitch / 256;
    double by = (double)rows * pitch;
    float t;
    t = timeit([&] { k_tile<0><<<rows / 128 * tiles_n, 256>>>((char*)d, tiles_n, pitch); }); printf("tile pat0 (16 rows x 64B)  pitch %5d  %7.0f GB/s\n", pitch, by / t / 1e6);
    t = timeit([&] { k_tile<1><<<rows / 128 * tiles_n, 256>>>((char*)d, tiles_n, pitch); }); printf("tile pat1 (4 rows x 256B)  pitch %5d  %7.0f GB/s\n", pitch, by / t / 1e6);
    t = timeit([&] { k_tile<2><<<rows / 128, 256>>>((char*)d, tiles_n, pitch); });           printf("tile pat2 (full rows)      pitch %5d  %7.0f GB/s\n", pitch, by / t / 1e6);
  }
  hipMemsetAsync(d, 0, bytes, 0);
  float t = timeit([&] { hipMemsetAsync(d, 0, bytes, 0); }); printf("hipMemsetAsync %7.0f GB/s\n", bytes / t / 1e6);
  t = timeit([&] { hipMemcpyAsync(d, a, bytes, hipMemcpyDeviceToDevice, 0); }); printf("hipMemcpy D2D %7.0f GB/s (r+w)\n", 2.0 * bytes / t / 1e6);
  return 0;
}
