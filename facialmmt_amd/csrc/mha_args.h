// Launch arguments shared by the two multi-head-attention formulations (attn.hip: fp32-exact VALU path and
// head_dim 32; mha_mfma.hip: bf16 MFMA path for head_dim 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct MhaArgs {
    int Lq, Lk, B, E, nH;
    const void* q; int ldq;
    const void* k; const void* v; int ldkv;
    float scale, drop_p; uint64_t seed; const uint64_t* seed_dev;
    const float* key_bias;          // optional additive logit bias per (batch, key): [B][Lk] fp32, nullptr = none
    void* out; int ldo; float* lse;
    const void* dout;
    void* dq; int lddq; void* dk; void* dv; int lddkv;
    int bm;                         // 0: time-major operands (row t of batch b at (t * B + b) * ld); 1: batch-major ((b * L + t) * ld; dtype | FMMT_BATCH_MAJOR)
    // row index (in units of the row pitch) of query / key t of batch b
    __host__ __device__ size_t rq(int t, int b) const { return bm ? (size_t)b * Lq + t : (size_t)t * B + b; }
    __host__ __device__ size_t rk(int t, int b) const { return bm ? (size_t)b * Lk + t : (size_t)t * B + b; }
};

int fmmt_mha_mfma_fwd_launch(const MhaArgs& a, hipStream_t st);
int fmmt_mha_mfma_bwd_launch(const MhaArgs& a, hipStream_t st);
