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
    // backward with recomputation (fmmt_window_block_attn_bwd): q, k, v are formed inside the kernel from LN1(x) and the head's rows of
    // Wqkv, d(attention output) from the block's output gradient `dout` [tokens, C] and the head's columns of Wproj
    const void* xn;
    const void* wqkv;
    const float* bqkv;
    const void* wproj;
    const float* rowscale;
    // 16 KB behind the d(bias) partials of the workspace: where the backward's lanes without a token (pad slots 49..63, idle tail
    // iterations) put their 16-byte stores.  Unconditional stores keep the kernel's store COUNT static, and with it the compiler's
    // vmcnt arithmetic: behind conditional stores every wait for a prefetched load degrades to vmcnt(0), i.e. to waiting out the
    // previous problem's write latency as well.
    void* sink;
};
constexpr size_t WA_SINK_BYTES = 16 * 1024;

int fmmt_wattn_mfma_fwd_launch(const WaArgs& a, int grid, hipStream_t st);
int fmmt_wattn_mfma_bwd_launch(const WaArgs& a, int grid, hipStream_t st);
int fmmt_wattn_mfma_bwd_rc_launch(const WaArgs& a, int grid, hipStream_t st);      // recompute variant: 8-wave workgroups, C = 96 / 192
int fmmt_wattn_bwd_ref_launch(int dtype, const WaArgs& a, int grid, hipStream_t st);   // wattn_bwd_ref.hip: its element-type-generic restatement, C = 96

// attn.hip: fixed-order reduction of the per-workgroup dense d(bias) partials [num_heads][parts_per_head][49 * 49] that sit at the head
// of a fmmt_window_attn_bwd_workspace(num_heads) workspace into dtable [169][num_heads]; workgroups per head for an 8-wave backward
int fmmt_wattn_dtable_finish(float* part, int num_heads, int parts_per_head, const int32_t* index, float* dtable, hipStream_t st);
int fmmt_wattn_bwd_groups(int B_, int num_heads, int windows_per_iteration);
