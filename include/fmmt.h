/* libfmmt_hip -- C ABI of the MI355X (gfx950) hot path of FacialMMT.
 *
 * Drop-in boundary.  The reference has no native code: the interface each entry
 * point replaces is a piece of `torch.nn` module maths, cited per function as file:line relative to
 * the reference tree.  Rules of the ABI:
 *   - plain C: device pointers, sizes, a hipStream_t passed as void*; no torch / C++ types;
 *   - nothing is allocated or freed inside: outputs, saved-for-backward tensors and workspaces are
 *     caller-owned device buffers (the Python host allocates them with torch's caching allocator);
 *   - every launch is asynchronous on `stream`; no global mutable state (re-entrant per stream);
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative FMMT_E* for
 *     argument errors (shape/alignment) -- never a silent fallback;
 *   - `dtype` selects the activation element type: FMMT_F32 (parity mode, exact-fp32 MFMA/VALU) or
 *     FMMT_BF16 (throughput mode: bf16 storage, fp32 accumulate / softmax / statistics).
 *     Parameters that are *reduced into* (weight/bias/affine gradients, statistics) are always fp32.
 *   - all matrices are row-major; "ld" = leading dimension in elements; pointers 16-byte aligned,
 *     channel counts multiples of 8 (bf16) / 4 (f32).
 */
#ifndef FMMT_H
#define FMMT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMMT_F32 0
#define FMMT_BF16 1
/* dtype flag (OR-ed into FMMT_BF16) understood by the fused Mlp entry points (fmmt_mlp_fwd, fmmt_mlp_ln_fwd, fmmt_mlp_bwd_input,
 * fmmt_mlp_ln_bwd_input): run the kernel's element-type-generic restatement (csrc/mlp_ref.hip) instead of the tuned kernel.  The
 * same restatement with fp32 fragments is what dtype = FMMT_F32 selects on those entry points (all activation / weight operands fp32,
 * nothing rounded: the parity instantiation, held to the reference's goldens at 1e-3); FMMT_BF16 | FMMT_GENERIC exists so that tests
 * can hold the generic template against the kernel the benchmark runs.  (The block-half kernels have _ref entry points instead.) */
#define FMMT_GENERIC 0x100
/* fmmt_mlp_fwd / fmmt_mlp_ln_fwd / fmmt_mlp_bwd_input / fmmt_mlp_ln_bwd_input, dtype | FMMT_SAVE_DG: the h_pre tensor holds gelu'(pre-activation) instead
 * of the pre-activation -- the only thing the backward needs of it (Swin_Transformer.py:19-28) -- written by the forward beside gelu() (one shared
 * exponential), multiplied in by the backward (no polynomial there).  The fused counterpart of FMMT_EPI_GELU_DG / FMMT_EPI_MUL_AUX; round 6. */
#define FMMT_SAVE_DG 0x200
/* fmmt_mha_fwd / fmmt_mha_bwd, dtype | FMMT_BATCH_MAJOR: the operands are batch-major -- row t of batch b of q / out / dout / dq at
 * (b * Lq + t) * ld, of k / v / dk / dv at (b * Lk + t) * ld -- instead of time-major ((t * B + b) * ld): the layout of the text encoder's
 * (batch, tokens, hidden) activations (transformers' *SelfAttention, src/models.py:75-91), whose attention then needs no transposes.
 * lse, key_bias and the dropout stream do not depend on the layout. */
#define FMMT_BATCH_MAJOR 0x400

#define FMMT_EINVAL (-1)   /* bad shape / unsupported size */
#define FMMT_EALIGN (-2)   /* pointer or leading dimension not 16-byte aligned */
#define FMMT_EWORKSPACE (-3) /* workspace too small */

/* epilogue flags of fmmt_linear_fwd */
#define FMMT_EPI_GELU 1       /* y = gelu(acc + bias)          (nn.GELU / F.gelu, exact erf form) */
#define FMMT_EPI_GELU_BWD 2   /* y = acc * gelu'(aux)          (backward of the line above)        */
#define FMMT_EPI_GELU_DG 3    /* y = gelu(acc + bias), y_pre = gelu'(acc + bias): the DERIVATIVE is what the backward needs of the pre-activation
                                 (Swin_Transformer.py:19-28: fc1 -> GELU), so it is formed where gelu() is -- one shared exponential -- and stored
                                 in its place; round 6 */
#define FMMT_EPI_MUL_AUX 4    /* y = acc * aux                 (backward of the line above: aux = the stored derivative; no polynomial) */

int fmmt_version(void);

/* ------------------------------------------------------------------------------------------------
 * Linear layers.  Replaces nn.Linear / F.linear at every site of the hot path:
 *   Swin_Transformer.py:19-21,25-28 (Mlp fc1/fc2), :105-107,120,142 (qkv / proj), :304,326
 *   (PatchMerging.reduction), :493 (head Linear(37632,512)); multihead_attention.py:152-158
 *   (_in_proj slices), :130 (out_proj); CrossmodalTransformer.py:129-130,156-158 (fc1/fc2).
 *
 * y[M,N] = epi( x[M,K] . w[N,K]^T + bias[N] )  then  y = res[M,N] + rowscale[m / rows_per_scale] * y
 *   - w is the nn.Linear weight layout (out_features, in_features), element type = dtype;
 *   - bias (fp32) may be NULL; res (dtype) may be NULL; rowscale (fp32, DropPath per-sample multiplier,
 *     Swin_Transformer.py:267-268) may be NULL;
 *   - FMMT_EPI_GELU: if y_pre != NULL the pre-activation (acc + bias) is stored there as well;
 *   - FMMT_EPI_GELU_BWD: aux[M,N] (dtype) holds the saved pre-activation;
 *   - FMMT_EPI_GELU_DG: y_pre (may be NULL) receives gelu'(acc + bias); FMMT_EPI_MUL_AUX: aux[M,N] (dtype) holds that derivative.
 * The same entry point computes input gradients: dx[M,K] = dy[M,N] . (w^T)[K,N]^T with a transposed
 * copy of the weight.  K % 8 == 0 (bf16) / K % 4 == 0 (f32), N % 4 == 0.
 */
int fmmt_linear_fwd(int dtype, int M, int N, int K,
                    const void* x, int ldx, const void* w, int ldw, const float* bias,
                    void* y, int ldy, void* y_pre,
                    int epi, const void* aux, int ldaux,
                    const void* res, int ldres, const float* rowscale, int rows_per_scale,
                    void* stream);

/* Few-token Linear over three weights in one launch: the query / key / value projections of MELDTransEncoder's SelfAttention
 * (modules/Transformer.py:64-103: three nn.Linear applied to the same hidden states) and their input gradient.
 *   seg_mode 1: y[:, s*N/3:(s+1)*N/3] = x @ w_s^T + bias_s, w_s [N/3][K] (bias_s may be NULL);
 *   seg_mode 2: y = sum_s x[:, s*K/3:(s+1)*K/3] @ w_s^T, w_s [N][K/3], no bias.
 * bf16; M <= 4096, N, K, N/3 resp. K/3 multiples of 64, few tiles (the fusion stack's shapes).  FMMT_EINVAL for anything else: issue three
 * fmmt_linear_fwd calls instead. */
int fmmt_linear_fwd_seg3(int dtype, int M, int N, int K, const void* x, int ldx, const void* w0, const void* w1, const void* w2, int ldw,
                         int seg_mode, const float* bias0, const float* bias1, const float* bias2, void* y, int ldy, void* stream);

/* Split-K variant for skinny problems (few output tiles, very long K: the 49*768 -> 512 embedding head,
 * Swin_Transformer.py:493).  y = x . w^T + bias only (no activation / residual).  The K range is cut
 * across workgroups into fp32 partials in `workspace` and summed in a fixed order.
 * fmmt_linear_splitk_workspace returns 0 when the shape is not a split-K shape (use fmmt_linear_fwd). */
size_t fmmt_linear_splitk_workspace(int M, int N, int K);
int fmmt_linear_fwd_splitk(int dtype, int M, int N, int K, const void* x, int ldx, const void* w, int ldw,
                           const float* bias, void* y, int ldy, void* workspace, size_t workspace_bytes,
                           void* stream);

/* Weight/bias gradients of the same layers (autograd of F.linear):
 *   dw[N,K] (fp32) = sum_m s_m * dy[m,N]^T x[m,K],   db[N] (fp32) = sum_m s_m * dy[m,N]   (db may be NULL)
 * with s_m = rowscale[m / rows_per_scale] (NULL -> 1).  The contraction over M is split across
 * workgroups into fp32 partials in `workspace` and combined in a fixed order (deterministic).
 * x_epi = FMMT_EPI_GELU contracts with gelu(x) instead of x: the weight gradient of the second Linear of an Mlp whose
 * activation was not stored (fmmt_mlp_fwd keeps only the pre-activation); 0 = plain.
 * Query the workspace size first (it depends on dtype: the bf16 and fp32 kernels split the tokens differently).
 * N % 8 == 0 and K % 8 == 0 (bf16) / % 4 (f32). */
size_t fmmt_linear_wgrad_workspace(int dtype, int M, int N, int K);
int fmmt_linear_wgrad(int dtype, int M, int N, int K,
                      const void* dy, int lddy, const void* x, int ldx,
                      float* dw, float* db, const float* rowscale, int rows_per_scale, int x_epi,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The same operation as its two launches, for callers that want to time or overlap them separately:
 *   _partials: the split contraction (MFMA kernel) -> fp32 partials in `workspace` (want_bias != 0: also the bias partials);
 *   _finish  : fixed-order sum of the partials into dw (and db, which requires the partials call to have had want_bias).
 * Both must be given the same (dtype, M, N, K) and workspace. */
int fmmt_linear_wgrad_partials(int dtype, int M, int N, int K,
                               const void* dy, int lddy, const void* x, int ldx, int want_bias,
                               const float* rowscale, int rows_per_scale, int x_epi,
                               void* workspace, size_t workspace_bytes, void* stream);
int fmmt_linear_wgrad_finish(int dtype, int M, int N, int K, float* dw, float* db,
                             const void* workspace, size_t workspace_bytes, void* stream);

/* Fused Mlp forward for the narrow Swin stages (C = 96 / 192, hidden 4C; bf16 only):
 *   y[M,C] = res[M,C] + rowscale[m / rows_per_scale] * ( gelu(x[M,C] . w1[4C,C]^T + b1) . w2[C,4C]^T + b2 )
 * replaces Mlp.forward (Swin_Transformer.py:14-30) together with the residual add and DropPath of the block (:268) in ONE
 * launch; the hidden activation stays on chip.  h_pre (bf16 [M,4C], may be NULL in inference) receives the pre-activation
 * x . w1^T + b1, which is all the backward needs (GELU' via FMMT_EPI_GELU_BWD, the activation recomputed inside
 * fmmt_linear_wgrad with x_epi = FMMT_EPI_GELU); h_act (bf16 [M,4C], may be NULL) additionally receives the activation
 * gelu(.) for callers that prefer reading it back to recomputing it.  Result bit-identical to fmmt_linear_fwd(FMMT_EPI_GELU) followed by
 * fmmt_linear_fwd(res, rowscale).  Other widths return FMMT_EINVAL (use the two-launch form). */
int fmmt_mlp_fwd(int dtype, int M, int C, const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                 const void* res, const float* rowscale, int rows_per_scale, void* y, void* h_pre, void* h_act, void* stream);

/* The Mlp half of a SwinTransformerBlock in ONE launch, LayerNorm included (bf16; C = 96 / 192):
 *   y[M,C] = x + rowscale[m / rows_per_scale] * ( gelu(LayerNorm(x) . w1^T + b1) . w2^T + b2 )
 * replaces norm2 -> Mlp -> DropPath -> residual of SwinTransformerBlock.forward (Swin_Transformer.py:267-268, Mlp :14-30): the
 * row statistics are formed on the fragments the first product consumes, so LayerNorm(x) never makes a round trip and the
 * residual is the input itself.  For the backward: xn [M,C] = LayerNorm(x) (fc1's weight gradient contracts with it; may be NULL),
 * mean / rstd [M] fp32 (both or neither), h_pre / h_act as fmmt_mlp_fwd. */
int fmmt_mlp_ln_fwd(int dtype, int M, int C, const void* x, const float* ln_gamma, const float* ln_beta, float eps,
                    const void* w1, const float* b1, const void* w2, const float* b2, const float* rowscale, int rows_per_scale,
                    void* y, void* xn, float* mean, float* rstd, void* h_pre, void* h_act, void* stream);

/* Input gradient of the same Mlp in ONE launch (bf16; C = 96 / 192), autograd of Swin_Transformer.py:14-30 with respect to x:
 *   dh[M,4C] = rowscale[m / rows_per_scale] * (dy[M,C] . w2) * gelu'(h_pre)      (stored: both weight gradients contract with it)
 *   dx[M,C]  = dh . w1
 * w2t = w2^T [4C, C] and w1t = w1^T [C, 4C] (the transposed bf16 copies the two-launch form also reads).  Replaces
 * fmmt_linear_fwd(FMMT_EPI_GELU_BWD) followed by fmmt_linear_fwd: dh is written once and never read back by this pair. */
int fmmt_mlp_bwd_input(int dtype, int M, int C, const void* dy, const void* h_pre, const void* w2t, const void* w1t,
                       const float* rowscale, int rows_per_scale, void* dh, void* dx, void* stream);

/* The same launch with the backward of the block's norm2 (Swin_Transformer.py:267-268: x = x + drop_path(mlp(norm2(x)))) as its tile
 * epilogue: instead of d(LN out) it writes dx = LayerNorm'(d(LN out); x, mean, rstd, gamma) + dy, the gradient of the block's residual
 * stream below the Mlp half, and d(gamma) / d(beta) of norm2 (per-workgroup partial sums in `workspace`, summed in fixed order by a
 * second small launch).  x: the LayerNorm's input (M, C); mean / rstd: the statistics fmmt_mlp_ln_fwd saved.  bf16, C = 96
 * (FMMT_EINVAL otherwise: fmmt_mlp_bwd_input + fmmt_layernorm_bwd).  Replaces autograd through nn.LayerNorm + Mlp. */
size_t fmmt_mlp_ln_bwd_input_workspace(int C);
int fmmt_mlp_ln_bwd_input(int dtype, int M, int C, const void* dy, const void* h_pre, const void* w2t, const void* w1t,
                          const float* rowscale, int rows_per_scale, const void* x, const float* mean, const float* rstd,
                          const float* ln_gamma, void* dh, void* dx, float* dgamma, float* dbeta, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm.  Replaces nn.LayerNorm (eps 1e-5) at Swin_Transformer.py:198,204,239,268 (norm1/2),
 * :305,325 (PatchMerging.norm over the 2x2 concat), :410,421 (PatchEmbed.norm), :491 (head norm);
 * CrossmodalTransformer.py:131,144-150,155,87-88.
 *
 * y[r,:] = (x_r - mean_r) * rstd_r * gamma + beta; mean/rstd (fp32, [M]) are saved for backward.
 * merge_hw > 0 selects the PatchMerging gather (Swin_Transformer.py:316-323): x is the
 * (n, H=W=merge_hw, C/4) token grid, logical row r = (n, h2, w2) and logical channel block q of
 * width C/4 reads token (2*h2 + (q&1), 2*w2 + (q>>1)); y is the dense [M, C] matrix.
 */
int fmmt_layernorm_fwd(int dtype, int M, int C, const void* x, const float* gamma, const float* beta,
                       float eps, void* y, float* mean, float* rstd, int merge_hw, void* stream);

/* dx = add + LN'(dy) ; dgamma/dbeta (fp32 [C]) are reduced deterministically through `workspace`
 * (fmmt_layernorm_bwd_workspace(C) bytes).  `add` (dtype, same layout as dx) may be NULL.  With merge_hw > 0, dy is
 * the dense [M, C] gradient and dx / add use the (n, H, W, C/4) token-grid layout (the scatter is a
 * permutation, every element is written once). */
size_t fmmt_layernorm_bwd_workspace(int C);
int fmmt_layernorm_bwd(int dtype, int M, int C, const void* dy, const void* x, const float* mean,
                       const float* rstd, const float* gamma, const void* add, void* dx,
                       float* dgamma, float* dbeta, int merge_hw,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (Shifted-)window attention core.  Replaces, in one kernel and without materialising anything:
 * torch.roll (Swin_Transformer.py:244,261), window_partition (:33-45,249-250), the per-head
 * q*scale @ k^T + relative_position_bias_table[relative_position_index] + attn_mask -> softmax ->
 * @ v of WindowAttention.forward (:120-141) and window_reverse (:48-62,256-257).
 *
 * qkv : [n_img * H * W, 3C] token order (output of the qkv Linear applied per token), dtype
 * out : [n_img * H * W, C]  token order, head h occupies channels [h*hd, (h+1)*hd), hd = C/num_heads (== 32)
 * lse : fp32 [n_img * nW * num_heads * 49] log-sum-exp per query row (saved for backward)
 * table: fp32 [(2*7-1)^2, num_heads]; index: int32 [49*49]; mask: fp32 [nW_mask,49,49] or NULL,
 *        window b_ uses mask[b_ % nW_mask] (WindowAttention.forward's broadcast, :131-134);
 *        mask_is_shift != 0 asserts that `mask` is exactly the SW-MSA mask SwinTransformerBlock builds for
 *        (H, W, shift) (:208-227, values {0,-100}): the bf16 kernel then derives it from window coordinates
 *        instead of reading 9.6 KB per window (the fp32 kernel always reads the tensor).
 * window size is 7 (swin_conf.yaml:20); H, W multiples of 7; shift in [0,7).
 */
int fmmt_window_attn_fwd(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                         const void* qkv, const float* table, const int32_t* index,
                         const float* mask, int nW_mask, int mask_is_shift, float scale,
                         void* out, float* lse, void* stream);

/* dqkv [n_img*H*W, 3C] (dtype, every element written); dtable fp32 [(169), num_heads] (overwritten).
 * workspace: fmmt_window_attn_bwd_workspace(num_heads) bytes. */
size_t fmmt_window_attn_bwd_workspace(int num_heads);
int fmmt_window_attn_bwd(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                         const void* qkv, const void* out, const void* dout, const float* lse,
                         const float* table, const int32_t* index, const float* mask, int nW_mask,
                         int mask_is_shift, float scale, void* dqkv, float* dtable,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The attention half of a SwinTransformerBlock in ONE launch, one wavefront per window (bf16; C = 96, head_dim 32):
 *   y = x + rowscale[img] * ( proj( W-MSA( LayerNorm(x) . wqkv^T + bqkv ) ) + bproj )
 * replaces norm1 -> roll -> window_partition -> WindowAttention.forward -> window_reverse -> roll -> residual + DropPath of
 * SwinTransformerBlock.forward (Swin_Transformer.py:233-266; WindowAttention :113-144; window helpers :33-62): the
 * normalised tokens, qkv (three times the activation) and the attention output never round-trip HBM between four launches.
 *   x, y      : [n_img*H*W, C] token order (dtype = FMMT_BF16);   wqkv [3C, C], wproj [C, C] (bf16, nn.Linear layout)
 *   ln_gamma/ln_beta [C], bqkv [3C] (may be NULL), bproj [C] (may be NULL), table [169, num_heads], rowscale [n_img] (may be NULL): fp32
 *   shift > 0 : the SW-MSA mask SwinTransformerBlock builds for (H, W, shift) (:208-227), derived from window coordinates
 *               (a caller holding any other mask tensor uses the four-launch form: fmmt_window_attn_fwd takes the tensor)
 *   saved for the backward (what the four launches would have left behind, minus qkv):
 *     xn [tokens, C] = LayerNorm(x) (may be NULL), attn_out [tokens, C] = attention output before proj (may be NULL),
 *     mean / rstd [tokens] fp32 (both or neither), lse [n_img*nW*num_heads*49] fp32 (required).
 * Operand roundings and GEMM accumulation orders are those of fmmt_layernorm_fwd -> fmmt_linear_fwd -> fmmt_window_attn_fwd ->
 * fmmt_linear_fwd; the softmax differs in the last bits (base-2 exponentials, normalisation after the second product).
 * dtype = FMMT_F32 (x, y, xn, attn_out, wqkv, wproj fp32): the PARITY instantiation -- the same kernel written over an element-type
 * trait (csrc/wblock_ref.hip: fp32 fragments, 8 x v_mfma_f32_16x16x4_f32 per 32-deep block, nothing rounded), slow, held to the
 * reference's block goldens at 1e-3.  Other widths return FMMT_EINVAL (use the four-launch form). */
int fmmt_window_block_fwd(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                          const void* x, const float* ln_gamma, const float* ln_beta, float eps,
                          const void* wqkv, const float* bqkv, const void* wproj, const float* bproj,
                          const float* table, const int32_t* index, float scale, const float* rowscale,
                          void* y, void* xn, void* attn_out, float* mean, float* rstd, float* lse, void* stream);
/* The element-type-generic restatement itself (same arguments): FMMT_F32 = what fmmt_window_block_fwd(FMMT_F32) runs; FMMT_BF16 = the
 * bf16 instantiation of the generic template, which tests compare with the production kernel -- the link between the kernel the
 * benchmark runs and the instantiation the 1e-3 goldens reach (Swin_Transformer.py:233-266, :113-144). */
int fmmt_window_block_fwd_ref(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                              const void* x, const float* ln_gamma, const float* ln_beta, float eps,
                              const void* wqkv, const float* bqkv, const void* wproj, const float* bproj,
                              const float* table, const int32_t* index, float scale, const float* rowscale,
                              void* y, void* xn, void* attn_out, float* mean, float* rstd, float* lse, void* stream);

/* Backward of the attention core for the fused block half, WITHOUT a materialised qkv (bf16; C = 96 or 192):
 *   dqkv [tokens, 3C] = d(loss) / d(qkv) of WindowAttention (Swin_Transformer.py:120-141) and dtable [169, num_heads],
 * (fp32 parity form: see fmmt_window_block_attn_bwd_ref below)
 * from xn = LayerNorm(x) [tokens, C], dy = the gradient of the block half's OUTPUT [tokens, C] (not of the attention output), the saved
 * attention output and log-sum-exp, wqkv / bqkv / wproj and the DropPath scale.  Each wave re-forms the q, k, v fragments of its
 * (window, head, token tiles) with the head's rows of wqkv and d(attention output) = rowscale * dy . wproj[:, head] with the head's
 * columns of wproj on the matrix cores: replaces the qkv re-computation GEMM, the proj input-gradient GEMM and the reads of qkv and
 * d(attention output) of fmmt_window_attn_bwd.  shift > 0 = the standard SW-MSA mask.  workspace: fmmt_window_attn_bwd_workspace(num_heads).
 * The weight and LayerNorm gradients of the block half remain separate launches (fmmt_linear_wgrad on dqkv / xn and on dy / attn_out,
 * fmmt_linear_fwd for d(xn), fmmt_layernorm_bwd). */
int fmmt_window_block_attn_bwd(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                               const void* xn, const void* dy, const void* attn_out, const float* lse,
                               const void* wqkv, const float* bqkv, const void* wproj,
                               const float* table, const int32_t* index, float scale, const float* rowscale,
                               void* dqkv, float* dtable, void* workspace, size_t workspace_bytes, void* stream);
/* Input gradient of "LayerNorm -> Linear" in one launch: the tail of the fused block half's backward (Swin_Transformer.py:239-243,
 * attn(norm1(x)); replaces autograd through nn.LayerNorm + the qkv nn.Linear, i.e. fmmt_linear_fwd on W^T followed by fmmt_layernorm_bwd):
 *   dx[M,C] = LayerNorm'( dz[M,K] . w ; x, mean, rstd, ln_gamma ) + dres[M,C]         (dres: the residual stream's gradient, or NULL)
 * plus d(gamma) / d(beta) of the LayerNorm (per-workgroup partial sums in `workspace`, summed in fixed order by a second small launch).
 * wt = w^T [C, K] (the transposed bf16 copy the two-launch form also reads); mean / rstd: the statistics the forward saved.
 * C = 96, K = 288 (FMMT_EINVAL otherwise: the two launches).  d(LN out) is never written.  dtype FMMT_BF16: the production kernel;
 * FMMT_F32 (all five matrices fp32) and FMMT_BF16 | FMMT_GENERIC: the same kernel restated over an element-type trait
 * (lin_lnbwd_ref_kernel; fp32 = 8 x mfma_f32_16x16x4 per 32-deep block, nothing rounded) -- the parity instantiations. */
size_t fmmt_linear_ln_bwd_workspace(int C);
int fmmt_linear_ln_bwd(int dtype, int M, int C, int K, const void* dz, const void* wt, const void* x, const float* mean, const float* rstd,
                       const float* ln_gamma, const void* dres, void* dx, float* dgamma, float* dbeta, void* workspace,
                       size_t workspace_bytes, void* stream);

/* The same backward restated over an element-type trait (csrc/wattn_bwd_ref.hip; C = 96): dtype FMMT_F32 -- also what
 * fmmt_window_block_attn_bwd(FMMT_F32) runs -- takes fp32 xn / dy / attn_out / wqkv / wproj / dqkv and rounds nothing (parity: the reference's
 * gradient goldens at 1e-3); FMMT_BF16 is the generic template's bf16 instantiation, which tests compare with the production kernel. */
int fmmt_window_block_attn_bwd_ref(int dtype, int n_img, int H, int W, int C, int num_heads, int shift,
                                   const void* xn, const void* dy, const void* attn_out, const float* lse,
                                   const void* wqkv, const float* bqkv, const void* wproj,
                                   const float* table, const int32_t* index, float scale, const float* rowscale,
                                   void* dqkv, float* dtable, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-head attention core of the cross-modal encoder.  Replaces multihead_attention.py:85
 * (q *= scaling), :94-98 (head split), :109 (bmm), :121 (fp32 softmax), :124 (dropout), :126 (bmm),
 * :128 (head merge).  Time-major operands: q [Lq, B, E], k / v [Lk, B, ldkv] (k and v may be column
 * slices of one packed projection: pass the slice pointers and the shared row pitch ldkv).
 * head_dim = E / num_heads must be 64 or 32.  Dropout: keep-mask = 16-bit hash field(seed, query row, key) >= round(p * 2^16), kept
 * probabilities scaled by the inverse of the realised keep rate (|it - 1/(1-p)| < 1e-5 / (1-p)^2); p == 0 disables.  If seed_dev != NULL the seed is read from that device
 * word at kernel run time (so a captured hipGraph draws a fresh mask on every replay) and `seed` is ignored.  The head-averaged weights the reference also
 * returns (:133-134) are discarded by every caller (CrossmodalTransformer.py:147,151) and are not produced.
 * key_bias (fp32 [B, Lk], may be NULL): added to the scaled logits of key j of batch b before the softmax.  The
 * cross-modal encoder passes NULL (it has no key-padding mask at all); the per-modality self-attention encoders
 * (modules/Transformer.py:93-95 `attention_scores + attention_mask`, mask built at src/models.py:157,164 as
 * (1 - m) * -10000) pass their extended attention mask squeezed to [B, Lk].  No gradient is produced for it.
 */
int fmmt_mha_fwd(int dtype, int Lq, int Lk, int B, int E, int num_heads,
                 const void* q, int ldq, const void* k, const void* v, int ldkv, float scale, const float* key_bias,
                 float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* out, int ldo, float* lse, void* stream);
int fmmt_mha_bwd(int dtype, int Lq, int Lk, int B, int E, int num_heads,
                 const void* q, int ldq, const void* k, const void* v, int ldkv, float scale, const float* key_bias,
                 float dropout_p, uint64_t seed, const uint64_t* seed_dev, const void* out, const void* dout, int ldo,
                 const float* lse, void* dq, int lddq, void* dk, void* dv, int lddkv, void* stream);

/* The second return value of MultiheadAttention.forward (multihead_attention.py:133-134): the attention probabilities AFTER dropout,
 * averaged over the heads, weights fp32 [B, Lq, Lk].  Every caller in the reference discards it, so fmmt_mha_fwd does not form it;
 * this entry point recomputes it on request from q, k, the log-sum-exp fmmt_mha_fwd saved and the same dropout seed. */
int fmmt_mha_avg_weights(int dtype, int Lq, int Lk, int B, int E, int num_heads, const void* q, int ldq, const void* k, int ldkv,
                         float scale, const float* key_bias, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                         const float* lse, float* weights, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PatchEmbed.  Conv2d(3,96,k=4,s=4) + flatten + transpose (Swin_Transformer.py:407,419) has
 * non-overlapping 4x4 patches, so it is exactly a Linear over the gathered patches:
 *   cols = im2col(img)  [n*3136, 48], column order (c, ky, kx) == proj.weight.view(96, 48);
 *   fmmt_linear_fwd(cols, proj.weight.view(96,48), proj.bias) ; fmmt_layernorm_fwd (norm, :420-421).
 * img: [n, 3, 224, 224] NCHW (dtype).  col2im is the exact inverse permutation (every pixel belongs
 * to one patch) and turns d(cols) into d(img).
 */
int fmmt_patch_im2col(int dtype, int n_img, const void* img, void* cols, void* stream);
int fmmt_patch_col2im(int dtype, int n_img, const void* cols, void* dimg, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm1d(512) of the embedding head (Swin_Transformer.py:494).  x,y: [n, C] (dtype).
 * training != 0: batch statistics (biased variance for normalisation), running_mean/var updated
 * in place with `momentum` (unbiased variance), save_mean/save_invstd written for backward.
 * training == 0: running statistics.
 */
int fmmt_batchnorm1d_fwd(int dtype, int n, int C, const void* x, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps,
                         int training, void* y, float* save_mean, float* save_invstd, void* stream);
int fmmt_batchnorm1d_bwd(int dtype, int n, int C, const void* dy, const void* x, const float* gamma,
                         const float* save_mean, const float* save_invstd, int training,
                         void* dx, float* dgamma, float* dbeta, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Cross-modal input embedding.  Replaces `embed_scale * x_in + embed_positions(x_in[..., 0])`
 * (CrossmodalTransformer.py:63-66,71-74; position_embedding.py:8-27,63-76): position of (t, b) is
 * t+1 where x[t,b,0] != 0 and 0 (the zeroed padding row of the table) otherwise.
 * x,y: [L, B, E] time-major (dtype); table: fp32 [>= L+1, E].
 */
int fmmt_posemb_scale_fwd(int dtype, int L, int B, int E, const void* x, const float* table,
                          float scale, void* y, void* stream);

/* y = alpha * x elementwise (backward of the embedding scale); n elements, n % 8 == 0. */
int fmmt_scale(int dtype, size_t n, const void* x, float alpha, void* y, void* stream);

/* out[n] = sum_m x[m][n]: the bias gradient `grad_output.sum(0)` of a Linear whose GEMMs stay with the vendor library -- the
 * text encoder's 146 Linear layers (src/models.py:75-91: RobertaModel / BertModel; torch's own reduction takes a memset and a
 * multi-block reduce_kernel per layer, 25 us against 4).  x: [M][N] (dtype), row pitch ldx; out: N values of out_dtype
 * (dtype, or FMMT_F32); N % (16 / sizeof(dtype)) == 0; one launch, fixed summation order. */
int fmmt_colsum(int dtype, int out_dtype, int M, int N, const void* x, int ldx, void* out, void* stream);

/* LayerNorm backward of a bf16 module with bf16 affine parameters: the text encoder's LayerNorms (transformers'
 * RobertaSelfOutput / RobertaOutput / embeddings LayerNorm, src/models.py:75-91), whose forward stays torch's layer_norm.
 * Replaces torch's three backward launches by two; the row statistics are recomputed from x (nothing but x is saved).
 * dy, x, dx: bf16 [M][C]; gamma, dgamma, dbeta: bf16 [C]; C % 8 == 0, C <= 2048; workspace: fmmt_layernorm_bwd_bf16_workspace. */
size_t fmmt_layernorm_bwd_bf16_workspace(int M, int C);
int fmmt_layernorm_bwd_bf16(int M, int C, float eps, const void* dy, const void* x, const void* gamma, void* dx,
                            void* dgamma, void* dbeta, void* workspace, size_t workspace_bytes, void* stream);

/* The tail of a BERT / RoBERTa sublayer, y = LayerNorm(dropout(h) + res), as ONE launch per direction (+ the fixed-order reduction of the
 * backward's parameter gradients): transformers' RobertaSelfOutput / RobertaOutput and BertSelfOutput / BertOutput (the text encoder the
 * reference builds at src/models.py:75-91) after their dense layer, whose GEMM stays with the vendor library.  Replaces fused_dropout + add +
 * layer_norm (forward) and LayerNorm' + masked_scale + the dense bias' column sum (backward).  bf16 activations and bf16 affine parameters.
 *   forward : t = bf16(h * keep / (1 - p)); xsum = bf16(t + res) (saved for the backward); y = LayerNorm(xsum) * gamma + beta
 *   backward: dx = LayerNorm'(dy) (= the residual branch's gradient); dh = bf16(dx * keep / (1 - p)); dgamma, dbeta; dbias = colsum(dh) (may be NULL)
 * keep(e) = a 16-bit hash field of (seed, salt, e) >= round(p * 2^16) -- the counter-based generator of fmmt_mha_fwd, four elements per hash pair; "1 - p" above is
 * the realised keep rate 1 - round(p * 2^16) / 2^16 --: no mask is stored, the backward replays it from the same
 * (seed | *seed_dev, salt).  h, res, xsum, y, dy, dx, dh: bf16 [M][C]; gamma, beta, dgamma, dbeta, dbias: bf16 [C]; C % 8 == 0, C <= 2048, 0 <= p < 1. */
int fmmt_plm_dropadd_ln_fwd(int M, int C, float eps, const void* h, const void* res, const void* gamma, const void* beta, float p,
                            uint64_t seed, const uint64_t* seed_dev, uint64_t salt, void* xsum, void* y, void* stream);
size_t fmmt_plm_dropadd_ln_bwd_workspace(int M, int C);
int fmmt_plm_dropadd_ln_bwd(int M, int C, float eps, const void* dy, const void* xsum, const void* gamma, float p, uint64_t seed,
                            const uint64_t* seed_dev, uint64_t salt, void* dx, void* dh, void* dgamma, void* dbeta, void* dbias, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Backward of the text encoder's intermediate activation with the bias gradient of the Linear in front of it (transformers' *Intermediate: dense -> exact GELU,
 * src/models.py:75-91): dpre = dact * gelu'(pre) (bf16, the erf form to 7e-4 relative like every GELU' of this library) and dbias = colsum(dpre) over the
 * values as stored, in one pass + a fixed-order reduction (stock: GeluBackward, then a column sum -- two passes over the tokens x 4096 matrix).
 * dact, pre, dpre: bf16 [M][H]; dbias: bf16 [H]; H % 8 == 0, H <= 8192; workspace: fmmt_plm_gelu_bwd_colsum_workspace(M, H). */
size_t fmmt_plm_gelu_bwd_colsum_workspace(int M, int H);
int fmmt_plm_gelu_bwd_colsum(int M, int H, const void* dact, const void* pre, void* dpre, void* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* Weight gradient of an nn.Embedding with a few thousand indices (the text encoder's word / position / token-type tables, transformers' *Embeddings,
 * src/models.py:75-91): dweight[id] = sum of dy[t] over the tokens t with ids[t] == id in token order (fp32 sums, one rounding), every other row zero,
 * tokens of padding_idx (pass -1 for none) skipped -- torch.nn.functional.embedding's backward for a dense bf16 weight without max_norm /
 * scale_grad_by_freq.  Deterministic: no atomics on the output.  ids: int64 [T] (values in [0, V)); dy: bf16 [T][C]; dweight: bf16 [V][C], written
 * entirely; C % 8 == 0, C <= 2048, T <= 32768. */
int fmmt_embedding_bwd(int T, int C, int V, const int64_t* ids, int64_t padding_idx, const void* dy, void* dweight, void* stream);

/* The same pair with the affine parameters' type as an argument (param_dtype FMMT_BF16 | FMMT_F32: gamma, beta, dgamma, dbeta, dbias in that type;
 * activations bf16): also serves MELDTransEncoder's sublayer tails, LayerNorm(dropout(dense(h)) + input) with fp32 master parameters
 * (modules/Transformer.py:109-137).  fmmt_plm_dropadd_ln_* = these with FMMT_BF16. */
int fmmt_dropadd_ln_fwd(int param_dtype, int M, int C, float eps, const void* h, const void* res, const void* gamma, const void* beta, float p,
                        uint64_t seed, const uint64_t* seed_dev, uint64_t salt, void* xsum, void* y, void* stream);
size_t fmmt_dropadd_ln_bwd_workspace(int M, int C);
int fmmt_dropadd_ln_bwd(int param_dtype, int M, int C, float eps, const void* dy, const void* xsum, const void* gamma, float p, uint64_t seed,
                        const uint64_t* seed_dev, uint64_t salt, void* dx, void* dh, void* dgamma, void* dbeta, void* dbias, void* workspace,
                        size_t workspace_bytes, void* stream);

/* Facial-emotion frame filter of a target-task step (train.py:75-114), one launch per direction.  preds [nF][NL] fp32: per-face emotion
 * distribution (Gumbel-softmax of the Swin logits); vision_inputs [B][Lv][D] and out [B][Lv][D + NL] in `dtype`; vision_mask, new_mask [B][Lv] fp32;
 * num_imgs [B] int64 on the DEVICE (real frames per utterance, <= Lv).  Faces with sum(p^2) > threshold are kept; utterance u owns the faces
 * [b_{u-1}, b_u), b_u = sum_{i<=u} n_i - u (the reference's margin arithmetic), packs its kept faces to the front of its Lv slots with the matching
 * vision rows, the NL probabilities appended, new_mask = 1 on the filled slots; if no face of the batch passes: out = [vision_inputs | preds of the
 * real frames in order], new_mask = vision_mask.  src_face [B][Lv] int32 receives the face that fed each slot's probability columns (-1: none).
 * _bwd: dpreds [nF][NL] fp32 = gather of dout[..., D:] through src_face (vision_inputs are data: no gradient).
 * nF <= 8192, B <= 256, B * Lv <= 8192, else FMMT_EINVAL. */
int fmmt_select_frames_fwd(int dtype, int nF, int NL, int B, int Lv, int D, const float* preds, const void* vision_inputs,
                           const float* vision_mask, const int64_t* num_imgs, float threshold, void* out, float* new_mask,
                           int32_t* src_face, void* stream);
int fmmt_select_frames_bwd(int dtype, int nF, int NL, int B, int Lv, int D, const void* dout, const int32_t* src_face, float* dpreds,
                           void* stream);

/* Batched refresh of bf16 weight shadows: one launch casts (fp32 -> bf16, or copies bf16) and optionally transposes n_desc
 * parameter matrices.  The reference keeps fp32 nn.Parameters (train.py:336-349 builds the optimizer over them); the bf16 GEMMs
 * read bf16 shadows W and W^T of them, which a training step has to rebuild after every optimizer step -- per weight that
 * was ~360 launches per step.  desc: DEVICE array of n_desc records
 *   { const void* src; void* dst; int32 rows, cols, src_ld, flags, tile_begin, tiles_c; }   (40 bytes, natural alignment)
 * flags bit 0: dst is [cols][rows] (transpose), bit 1: src is fp32 (else bf16); 64 x 64 tiles, tiles_c = ceil(cols / 64),
 * tile_begin = tiles of all earlier records; n_tiles = their total.  The caller builds the table once. */
int fmmt_cast_batch(int n_desc, int n_tiles, const void* desc, void* stream);

/* Gradient hand-over + the norm of `clip_grad_norm_` in one pass (train.py:135-140: the backward's gradients -> the buffers the optimizer
 * reads; the reference's clip_grad_norm_ computes this norm over the same values): every record's gradient is written (flags bit 1: added)
 * into its fp32 slot and norm_out[0] = sqrt(sum of squares of everything the slots then hold).  desc: DEVICE array of n_desc records
 *   { const void* src; float* dst; int64 n; int32 blk_begin, flags; }     (32 bytes; flags bit 0: src is bf16, else fp32)
 * one block per 4096 elements, blk_begin = blocks of all earlier records, n_blocks = their total; partial: n_blocks floats of scratch.
 * Fixed summation order (per-block sums, finished by one block): replays are bit-identical. */
int fmmt_grad_handover(int n_desc, int n_blocks, const void* desc, float* partial, float* norm_out, void* stream);

/* Gradient clipping + AdamW + bf16 re-rounding of the parameters in one launch over every tensor of the step's optimizer.
 * Replaces `clip_grad_norm_(model.parameters(), clip)` (its scaling pass; the norm itself is the caller's, a device scalar) and
 * `optimizer.step()` (train.py:135-143), and the re-rounding of bf16 parameters that are stepped through fp32 masters.
 * hf_semantics != 0: the update of transformers.AdamW, the class the reference constructs (train.py:307,333: eps added to
 * sqrt(v) before the bias correction, decoupled weight decay applied after the update; its defaults are eps 1e-6, weight_decay 0);
 * hf_semantics == 0: torch.optim.AdamW (decay first, bias-corrected denominator).  desc: DEVICE array of n_desc records
 *   { float* p; const void* g; float* m; float* v; bf16* low_or_null; int64 n; int32 blk_begin, g_is_bf16; }  (56 bytes)
 * one block per 4096 elements, blk_begin = blocks of all earlier records, n_blocks = their total.  lr, step (the 1-based
 * step count t as a float) and total_norm (may be NULL: no clipping) are DEVICE scalars, so the call can sit in a HIP graph. */
int fmmt_adamw_batch(int n_desc, int n_blocks, const void* desc, const float* lr, const float* step, const float* total_norm,
                     float beta1, float beta2, float eps, float weight_decay, float max_norm, int hf_semantics, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input pre-step fused into PatchEmbed's gather.  Replaces, for one batch of square uint8 face
 * crops (n, S, S, 3) in image (HWC) layout, the chain the reference runs per frame on the host and caches as a
 * 224x224 float tensor:
 *   Aff-Wild2: transforms.Resize(224, BICUBIC) on the PIL image -> ToTensor -> Normalize(.5,.5)   utils/util.py:43-52
 *   MELD:      cv2.resize(im, (224,224), INTER_CUBIC) on the uint8 array -> ToTensor -> Normalize  utils/dataset.py:47-69
 * followed by PatchEmbed's 4x4 / stride-4 patch gather (Swin_Transformer.py:407,419): the output IS the
 * (n*3136, 48) operand of the patch-embedding GEMM (same layout as fmmt_patch_im2col), element type `dtype`.
 * Augmentations (ColorJitter, RandomErasing, ...) are host-side data pipeline and not part of this entry point.
 *
 * fmmt_resize_table   HOST function: fills table[out_size][8] = {4 source indices, 4 integer weights} per output
 *                     coordinate (the same table serves x and y) and lut[256] = float32 Normalize(ToTensor(byte)), with the
 *                     coefficient arithmetic of the library the reference calls (FMMT_RESIZE_PIL: Pillow, 22-bit, bit-exact,
 *                     pinned by tests/golden/preproc.npz; FMMT_RESIZE_CV2: OpenCV 8U INTER_CUBIC, 11-bit, restated from the
 *                     published algorithm, parity unpinned).  Up-scaling only (in_size <= out_size).
 * fmmt_resize_band_rows  HOST: the largest number of source rows one band of 4 output rows touches (must be <= 8).
 * fmmt_patch_embed_u8 device launch; table_dev / lut_dev are the two host tables copied to the device by the caller.
 */
#define FMMT_RESIZE_PIL 0
#define FMMT_RESIZE_CV2 1
int fmmt_resize_table(int mode, int in_size, int out_size, int32_t* table, float* lut);
int fmmt_resize_band_rows(const int32_t* table, int out_size);
int fmmt_patch_embed_u8(int dtype, int mode, int n_img, int in_size, const void* img_u8, const int32_t* table_dev,
                        const float* lut_dev, void* cols, void* stream);

/* PatchEmbed's projection + bias + LayerNorm in one launch (Swin_Transformer.py:392-422: proj, flatten, norm) on the patch matrix
 * fmmt_patch_embed_u8 / fmmt_patch_im2col produced: cols (M, K = 48) bf16, w (C = 96, 48) bf16 (Conv2d weight viewed as a matrix), bias /
 * ln_gamma / ln_beta fp32.  y (M, 96) = LayerNorm(x_pre), x_pre = bf16(cols . w^T + bias); x_pre, mean, rstd (or NULL at inference) are
 * what fmmt_layernorm_bwd needs.  C = 96, K = 48 only (FMMT_EINVAL otherwise: fmmt_linear_fwd + fmmt_layernorm_fwd).  FMMT_F32 (cols, w,
 * x_pre, y fp32): the same kernel template with fp32 fragments, nothing rounded -- the parity instantiation. */
int fmmt_patch_embed_ln_fwd(int dtype, int M, int C, int K, const void* cols, const void* w, const float* bias, const float* ln_gamma,
                            const float* ln_beta, float eps, void* x_pre, void* y, float* mean, float* rstd, void* stream);

/* The two launches above as one (SURVEY 8f rank 3 "fused into PatchEmbed's load"; utils/dataset.py:47-69, utils/util.py:43-52 ->
 * Swin_Transformer.py:392-422): uint8 crops (n_img, in_size, in_size, 3) -> bicubic resize / ToTensor / Normalize / 4x4 gather of one patch
 * row per workgroup in LDS -> projection + bias + LayerNorm from there.  y (n_img * 3136, 96); x_pre / mean / rstd as in
 * fmmt_patch_embed_ln_fwd (NULL at inference); cols (n_img * 3136, 48): the patch matrix, written for the projection's weight gradient, or
 * NULL (inference: it is never materialised).  table_dev / lut_dev: fmmt_resize_table's tables on the device.  FMMT_BF16 / FMMT_F32 (the
 * parity instantiation: same template, fp32 fragments). */
int fmmt_patch_embed_u8_ln_fwd(int dtype, int mode, int n_img, int in_size, const void* img_u8, const int32_t* table_dev, const float* lut_dev,
                               const void* w, const float* bias, const float* ln_gamma, const float* ln_beta, float eps,
                               void* cols, void* x_pre, void* y, float* mean, float* rstd, void* stream);

#ifdef __cplusplus
}
#endif
#endif
