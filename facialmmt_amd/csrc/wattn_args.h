// Launch arguments shared by the two window-attention formulations (attn.hip: fp32-exact VALU path used
// for parity; wattn_mfma.hip: bf16 MFMA path used for throughput).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct WaArgs {
    int n_img, H, W, C, nH, shift;
    const void* qkv;
    const float* table;
    const int32_t* index;
    const float* mask;
    int nW_mask;
    int mask_is_shift;     // mask == the standard SW-MSA mask of (H, W, shift): may be computed arithmetically
    float scale;
    void* out;
    float* lse;
    // backward
    const void* dout;
    void* dqkv;
    float* part;           // [nH][waves_per_head][49*49]
    int groups_per_head;   // workgroups per head
    int xcd_grouped;       // block numbering keeps the heads of a window group on one XCD (groups_per_head % 8 == 0)
};

int fmmt_wattn_mfma_fwd_launch(const WaArgs& a, int grid, hipStream_t st);
int fmmt_wattn_mfma_bwd_launch(const WaArgs& a, int grid, hipStream_t st);
