/* aki_mi355x.h - C ABI of the MI355X-native AKI modality-mutual-attention (MMA) path: forward, KV-cache decode, the
 * pre-training step (backward, loss, optimizer) and the fp8 (e4m3) projection variants.
 *
 * The reference (sony/aki) has no FFI layer: its boundary is Python object protocol
 * (SURVEY.md section 8(b)).  Each entry point below names the reference function whose work it
 * replaces (paths relative to /root/reference/codes/open_flamingo/, "HF:" = the transformers
 * dependency the reference delegates to).  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *  - Plain C: raw DEVICE pointers, sizes and strides in ELEMENTS, no torch/HIP types.
 *    `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - Ownership: the caller owns every buffer, including workspaces.  Kernels allocate nothing,
 *    keep no state between calls (one documented exception: the decode-attention workspace holds arrival counters
 *    that the kernel re-arms itself), never synchronise the host and are graph-capturable.
 *  - Errors: int return, 0 = AKI_OK, < 0 = aki_status code; never throws or aborts.  Shapes and
 *    alignment are validated on the host before anything is launched.
 *  - Threading: re-entrant; ordering only through `stream`.
 *  - dtype: AKI_DT_BF16 (MFMA path, fp32 accumulate/softmax) or AKI_DT_F32 (exact-f32 parity path, forward only);
 *    AKI_DT_FP8_E4M3 / AKI_DT_W8A16 where an entry point says so.  The backward / optimizer entry points are bf16.
 */
#ifndef AKI_MI355X_H
#define AKI_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AKI_ABI_VERSION 17

typedef enum {
  AKI_OK = 0,
  AKI_ERR_INVALID_ARG = -1,  /* null pointer, non-positive size, inconsistent shapes           */
  AKI_ERR_UNSUPPORTED = -2,  /* dtype / head_dim / size outside what the kernels are built for  */
  AKI_ERR_ALIGNMENT = -3,    /* pointer or leading dimension not 16-byte aligned                */
  AKI_ERR_WORKSPACE = -4,    /* workspace missing or too small                                  */
  AKI_ERR_LAUNCH = -5        /* hipGetLastError() after launch was not hipSuccess               */
} aki_status;

typedef enum {
  AKI_DT_BF16 = 0,
  AKI_DT_F32 = 1,
  AKI_DT_FP8_E4M3 = 2, /* e4m3 x and w with per-row scales (aki_linear_args / aki_mma_attn_args) */
  AKI_DT_W8A16 = 3     /* weight-only fp8: e4m3 w + w_scale, bf16 x; decode rows (aki_linear_fwd: M <= 16, aki_decode_linear_fwd: M <= 8) */
} aki_dtype;

/* Activation fused into aki_linear_fwd. */
typedef enum {
  AKI_ACT_NONE = 0,
  AKI_ACT_GELU_ERF = 1,  /* torch.nn.GELU()            - src/helpers.py:37                    */
  AKI_ACT_GELU_TANH = 2, /* gelu_pytorch_tanh          - HF:siglip MLP                        */
  AKI_ACT_SWIGLU = 3     /* out = up * silu(gate), W = [gate; up] - HF:phi3/modeling_phi3.py:49-64 */
} aki_act;

const char* aki_strerror(int status);
int aki_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Mask description.  The reference materialises a dense (B,1,L,L) int64 0/1 tensor
 * (src/vlm.py:410-443 `_make_modality_mutual_mask`, stacked by src/utils.py:99-108).  Here the
 * same information travels as a tiny table:
 *   visible(r,c) = valid(c) && ( c <= r || exists k: row_lo_k <= r < row_hi_k && col_lo_k <= c < col_hi_k )
 * `rects` is [B][max_rects]; unused entries are all-zero.  Row ranges of one sample's rectangles
 * must not overlap.  One rectangle per sample reproduces the reference; more are build-defined.
 * `col_valid_bits` is [B][ceil(L/64)] uint64, bit (c & 63) of word (c >> 6) = attention mask of
 * column c (bits at c >= L must be 0); NULL = every column < L valid.
 * `seq_lens` is [B] int32: rows r >= seq_lens[b] are the all-zero rows that batch stacking adds at
 * the bottom of a shorter sample's mask (src/utils.py:99-108) and are treated as rows with no
 * visible column; NULL = L for every sample.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t row_lo, row_hi, col_lo, col_hi;
} aki_mma_rect;

#define AKI_MAX_RECTS 8
#define AKI_PLAN_STRIDE 12 /* int32 per sample in the splice plan */

/* Rows with no visible column: the reference's additive finfo.min mask (transformers==4.41.2
 * `_prepare_4d_causal_attention_mask`, implicit at src/aki.py:125) yields a uniform softmax over
 * all L columns.  AKI_DEAD_ROWS_UNIFORM reproduces that; AKI_DEAD_ROWS_ZERO writes zeros. */
typedef enum { AKI_DEAD_ROWS_ZERO = 0, AKI_DEAD_ROWS_UNIFORM = 1 } aki_dead_rows;

/* ------------------------------------------------------------------------------------------------
 * aki_mma_attn_core_fwd - span-driven block-sparse softmax(QK^T * scale + mask) V.
 * Replaces: HF:phi3/modeling_phi3.py:145-167 `eager_attention_forward` as called from
 * `Phi3Attention.forward` (:218-263) under the reference's mask (src/vlm.py:556-564), plus the
 * 4-D mask inversion of transformers==4.41.2.  No L x L tensor is ever formed.
 *   q,k,v : [B,H,L,Dh] contiguous (k already rotated);  o : [B,L,H*Dh];  lse : [B,H,L] f32 or NULL
 *   workspace: aki_mma_attn_core_workspace_bytes() bytes, 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  const aki_mma_rect* rects;
  const uint64_t* col_valid_bits;
  const int32_t* seq_lens;
  int32_t max_rects;
  int32_t B, H, L, Dh;
  float scale;
  int32_t dtype;     /* aki_dtype */
  int32_t dead_rows; /* aki_dead_rows */
  int32_t kv_capacity; /* rows allocated per (batch, head) in k and v ([B,H,kv_capacity,Dh], a KV cache); 0 = L */
} aki_mma_attn_core_args;

size_t aki_mma_attn_core_workspace_bytes(int32_t B, int32_t H, int32_t L, int32_t Dh, int32_t dtype);
int aki_mma_attn_core_fwd(const aki_mma_attn_core_args* args, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * aki_mma_attn_fwd - the fused MMA op: QKV projection + RoPE + span-driven attention.
 * Replaces: `Phi3Attention.forward` up to (not including) o_proj, HF:phi3/modeling_phi3.py:218-258:
 *   qkv = x W_qkv^T ; split ; rotate-half RoPE with (cos,sin) ; attention core as above.
 *   x      : [B*L, d_model] (the RMS-normed hidden states), ldx elements between rows
 *   w_qkv  : [3*H*Dh, d_model] = `self_attn.qkv_proj.weight`, row-major, ldw
 *   cos,sin: f32 [pos_rows, Dh] tables (HF `Phi3RotaryEmbedding`; default or LongRoPE computed by the host)
 *   position_ids: int32 [B*L] row index into cos/sin per token, or NULL for (token index mod L)
 *   o      : [B, L, H*Dh]
 * Two stream-ordered launches inside one call (projection+RoPE epilogue, then attention); the
 * rotated Q/K and V live in the caller-owned workspace in head-major [B,H,L,Dh] order.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x;
  const void* w_qkv;
  const float* cos;
  const float* sin;
  const int32_t* position_ids;
  void* o;
  float* lse;
  const aki_mma_rect* rects;
  const uint64_t* col_valid_bits;
  const int32_t* seq_lens;
  int32_t max_rects;
  int32_t B, H, L, Dh, d_model;
  int32_t ldx, ldw;
  int32_t pos_rows;
  float scale;
  int32_t dtype;
  int32_t dead_rows;
  int32_t kv_capacity; /* aki_qkv_rope_fwd: rows per (batch, head) of k_out / v_out (prefill straight into a KV cache); 0 = L */
  /* dtype == AKI_DT_FP8_E4M3: x / w_qkv are e4m3 bytes with per-row scales (see aki_linear_args); q, k, v, o are bf16 and
   * the attention core runs exactly as in the bf16 path. */
  const float* x_scale;
  const float* w_scale;
  const float* row_scale;  /* folded input RMSNorm (see aki_linear_args): x is the RAW hidden state, w_qkv = W diag(gain); NULL = none */
} aki_mma_attn_args;

size_t aki_mma_attn_workspace_bytes(int32_t B, int32_t H, int32_t L, int32_t Dh, int32_t dtype);
int aki_mma_attn_fwd(const aki_mma_attn_args* args, void* workspace, size_t workspace_bytes, void* stream);

/* Stage 1 of the fused op on its own (KV-cache prefill, tests): q_out [B,H,L,Dh]; k_out / v_out [B,H,kv_capacity,Dh]
 * (kv_capacity = 0 means L).  AKI_DT_F32 needs a workspace of B*L*3*H*Dh floats; bf16 needs none. */
int aki_qkv_rope_fwd(const aki_mma_attn_args* args, void* q_out, void* k_out, void* v_out, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * aki_attn_fwd - plain (unmasked) multi-head attention softmax(Q K^T * scale) V for the vision side.
 * Replaces: SigLIP `eager_attention_forward` (HF:siglip/modeling_siglip.py:226-247, 16 heads x 72; call site
 * src/vlm.py:202-203) and the softmax attention of `PerceiverAttention.forward` (src/helpers.py:93-100, 8 x 64,
 * 144 latent queries over 729+144 keys).
 *   q/k/v are read in place through ELEMENT strides (batch, head, token; channel stride 1), so the fused QKV / KV
 *   projection outputs need no transposition;  o : [B, Lq, H*Dh] contiguous.  Dh in {32, 64, 72, 96}.
 *   AKI_DT_F32 needs contiguous head-major [B,H,L,Dh] inputs and a workspace of B*H*Dh floats.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  int64_t q_stride_b, q_stride_h, q_stride_t;
  int64_t k_stride_b, k_stride_h, k_stride_t;
  int64_t v_stride_b, v_stride_h, v_stride_t;
  int32_t B, H, Lq, Lk, Dh;
  float scale;
  int32_t dtype;
  float* lse;          /* optional [B,H,Lq] f32 log-sum-exp of the scaled scores (bf16 path; consumed by aki_attn_bwd) */
} aki_attn_args;

int aki_attn_fwd(const aki_attn_args* args, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * aki_linear_fwd - y = act(x W^T + bias) [+ residual]   (torch.nn.functional.linear semantics)
 * Replaces: every nn.Linear on the path - o_proj / gate_up_proj / down_proj
 * (HF:phi3/modeling_phi3.py:49-64,215-216), Perceiver to_q/to_kv/to_out and FeedForward linears
 * (src/helpers.py:32-39,72-74), projection (src/helpers.py:147), DecoupledLinear pieces (src/helpers.py:594-603).
 *   x [M,K] ldx ; w [N,K] ldw ; y [M,N_out] ldy (N_out = N/2 for AKI_ACT_SWIGLU, else N)
 *   bias [N] or NULL ; residual [res_rows? , N_out] ldr or NULL, row index = m % res_row_mod when res_row_mod > 0
 *   K must be a multiple of 64 (bf16) ; N_out a multiple of 4.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x;
  const void* w;
  const void* bias;
  const void* residual;
  void* y;
  int32_t M, N, K;
  int32_t ldx, ldw, ldy, ldr;
  int32_t res_row_mod;
  int32_t act;   /* aki_act */
  int32_t dtype; /* aki_dtype */
  /* AKI_DT_FP8_E4M3 (BASELINE configs[4]): x and w hold OCP e4m3 bytes (ldx / ldw in bytes, multiples of 16; K a multiple
   * of 128) produced by aki_quant_rows_fp8, with one f32 dequantisation scale per x row / per w row; bias, residual and
   * y stay bf16.  act: NONE or SWIGLU.  Ignored for the other dtypes. */
  const float* x_scale;
  const float* w_scale;
  /* Two-segment weight (bf16 MFMA GEMM, act NONE): logical weight row r comes from w[r] for r < w2_row0 and from
   * w2[min(r - w2_row0, w2_rows - 1)] beyond it (same ldw) - `DecoupledLinear` (src/helpers.py:594-603: the original
   * vocabulary rows ++ the rows of the added tokens) as ONE GEMM without a concatenated copy of the 197 MB head weight.
   * N counts logical rows and may exceed w2_row0 + w2_rows (padding columns repeat the last w2 row).  w2 = NULL: plain. */
  const void* w2;
  int32_t w2_row0;
  int32_t w2_rows;
  /* Normalisation folded into the GEMMs (bf16 MFMA path; HF:phi3/modeling_phi3.py:266-284 RMSNorm, torch.nn.LayerNorm of
   * HF:siglip/modeling_siglip.py:329-354) - the pre-norm of a block costs no launch and no pass over the activations:
   *   consumer   y = act( row_scale[m] * (x W'^T - row_shift[m] * col_shift[n]) + bias ) [+ residual]
   *              RMSNorm: W' = W diag(gain), row_scale = 1/rms(x_m), row_shift = NULL.
   *              LayerNorm: W' = W diag(gain), row_scale = 1/std(x_m), row_shift = mean(x_m), col_shift[n] = sum_k W'[n][k] (f32),
   *              bias = W beta + b.  (W', col_shift and that bias are prepared once per weight by the caller.)
   *   producer   stats_rstd != NULL: while writing y, the GEMM also computes the row statistics of y AS STORED (bf16):
   *              stats_rstd[m] = 1/sqrt(mean(y_m^2) + eps)                      (stats_mean == NULL: RMSNorm of the next block)
   *              stats_mean[m] = mean(y_m), stats_rstd[m] = 1/sqrt(var(y_m) + eps)   (stats_mean != NULL: LayerNorm)
   *              stats_workspace: aki_linear_stats_workspace_bytes(M, N_out) bytes, 16-byte aligned, ZERO-FILLED ONCE by the
   *              caller before its first use (arrival counters at its front, which every launch puts back to zero; launches
   *              sharing a workspace must be stream-ordered).  The counter area has a fixed size -
   *              aki_linear_stats_counter_bytes(M) is the same 1 MiB for every M - so one buffer sized for the largest
   *              (M, N_out) serves any sequence of smaller and larger launches without being zero-filled again.
   *              M <= 16 777 216 rows; act must not be SWIGLU.
   * All NULL: plain GEMM. */
  const float* row_scale;
  const float* row_shift;
  const float* col_shift;
  float* stats_rstd;
  float* stats_mean;
  float stats_eps;
  void* stats_workspace;
  size_t stats_workspace_bytes;
  /* Split-K workspace (optional; bf16, act NONE).  Launches with few rows (a one-sample prefill: M = 655 / 207 against N = 3072) have
   * too few output tiles to fill the chip; given this buffer the library may split the K range of a tile over several workgroups
   * whose f32 partial sums meet here and are added in a FIXED order by the last one to arrive (bit-reproducible).  256-byte aligned,
   * aki_linear_splitk_workspace_bytes(M, N, K) bytes or more (0: this shape is never split), ZERO-FILLED ONCE by the caller (tickets
   * at its front, put back to zero by every launch; launches sharing it must be stream-ordered).  NULL / too small: no split. */
  void* splitk_workspace;
  size_t splitk_workspace_bytes;
  /* Training forward of the gated MLP (bf16, act SWIGLU; HF:phi3/modeling_phi3.py:49-64 under autograd): besides y = silu(g) * u the
   * launch leaves the pre-activations [g | u] as bf16 [M, N] (row stride ld_preact elements, a multiple of 4, >= N; 8-byte aligned) for the
   * backward pass, and takes the activation OF those bf16 values - the same numbers aki_linear_fwd (act NONE) followed by aki_swiglu_fwd
   * produce, without the second pass over [M, N].  NULL: not kept (inference). */
  void* preact_out;
  int64_t ld_preact;
} aki_linear_args;

int aki_linear_fwd(const aki_linear_args* args, void* stream);
size_t aki_linear_stats_workspace_bytes(int32_t M, int32_t N_out);
size_t aki_linear_splitk_workspace_bytes(int32_t M, int32_t N, int32_t K);
size_t aki_linear_stats_counter_bytes(int32_t M);
/* aki_row_stats - the same statistics for a tensor no GEMM of this library produced (the first block's input): rstd[m] (and,
 * when mean != NULL, mean[m]) of x [rows, cols] bf16. */
int aki_row_stats(const void* x, int32_t rows, int32_t cols, int32_t ldx, float eps, float* rstd, float* mean, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation.  rmsnorm: HF:phi3/modeling_phi3.py:266-284 (fp32 statistics, result cast to the
 * input dtype BEFORE the gain multiply).  layernorm: torch.nn.LayerNorm (src/helpers.py:35,69-70,168).
 * ---------------------------------------------------------------------------------------------- */
int aki_rmsnorm_fwd(const void* x, const void* weight, void* y, int32_t rows, int32_t cols, int32_t ldx, int32_t ldy,
                    float eps, int32_t dtype, void* stream);
int aki_layernorm_fwd(const void* x, const void* weight, const void* bias, void* y, int32_t rows, int32_t cols,
                      int32_t ldx, int32_t ldy, float eps, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * aki_patch_embed_fwd - SigLIP patch embedding: Conv2d(3->E, k=P, s=P, valid) + flatten + pos-emb.
 * Replaces: `SiglipVisionEmbeddings.forward`, HF:siglip/modeling_siglip.py:175-185 (call site src/vlm.py:202-203).
 *   pixels [N,3,S,S] ; w [E, Kp] = conv weight flattened to [E,3*P*P] and zero-padded to Kp = roundup(3*P*P,64)
 *   bias [E] ; pos [G*G,E] (G = S/P) ; out [N,G*G,E] ; workspace = aki_patch_embed_workspace_bytes().
 * ---------------------------------------------------------------------------------------------- */
size_t aki_patch_embed_workspace_bytes(int32_t N, int32_t S, int32_t P, int32_t dtype);
int aki_patch_embed_fwd(const void* pixels, const void* w, const void* bias, const void* pos, void* out, int32_t N,
                        int32_t S, int32_t P, int32_t E, int32_t Kp, int32_t dtype, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Connector MLP (the Perceiver FeedForward block and the final projection).
 * aki_connector_mlp_fwd : out = x + W2 gelu(W1 LN(x))          - src/helpers.py:32-39,194
 * aki_connector_proj_fwd: out = Wp LN(x) + bp                   - src/helpers.py:196-197
 *   x [rows,d] ; w1 [4d,d] ; w2 [d,4d] ; workspace rows*(d+4d) elements.
 * ---------------------------------------------------------------------------------------------- */
size_t aki_connector_mlp_workspace_bytes(int32_t rows, int32_t d, int32_t d_inner, int32_t dtype);
int aki_connector_mlp_fwd(const void* x, const void* ln_w, const void* ln_b, const void* w1, const void* w2, void* out,
                          int32_t rows, int32_t d, int32_t d_inner, float eps, int32_t dtype, void* workspace,
                          size_t workspace_bytes, void* stream);
int aki_connector_proj_fwd(const void* x, const void* ln_w, const void* ln_b, const void* w, const void* b, void* out,
                           int32_t rows, int32_t d, int32_t d_out, float eps, int32_t dtype, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Language-stream fusion (integer / byte work).
 * aki_splice_plan : per sample, count <image> placeholders, find the first id == assistant_token_id
 *                   and the expanded length L_i = T - n_img + Nv*n_img   (src/vlm.py:488-496)
 *                   plan [B][AKI_PLAN_STRIDE] int32 = {n_img, q_idx, L_i, 0, t_0 .. t_7} with t_k the
 *                   index of the k-th placeholder (more than AKI_MAX_RECTS images per sample: unsupported)
 * aki_splice_fwd  : DecoupledEmbedding gather (src/helpers.py:445-484) + vision-token splice
 *                   (src/vlm.py:539-577) + padding with the scalar pad_token_id / -100
 *                   (src/vlm.py:584-598, src/utils.py:62-96) + the mask table (rects, valid bits).
 *   lang_x/attention_mask/labels int64 [B,T] ; embed_weight [V,d] ; embed_additional [n_add,d] or NULL
 *   vision_tokens [B,T_img,Nv,d] ; outputs sized for L_out = max_i L_i (right or left padding).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const int64_t* lang_x;
  const int64_t* attention_mask; /* NULL = all ones */
  const int64_t* labels;         /* NULL = no labels */
  const void* embed_weight;
  const void* embed_additional;
  const void* vision_tokens;
  const int32_t* plan; /* from aki_splice_plan */
  void* inputs_embeds; /* [B,L_out,d] */
  int64_t* labels_out; /* [B,L_out] or NULL */
  int64_t* mask_1d_out;/* [B,L_out] spliced 1-D mask, zero padded (always right-aligned like the reference's 2-D masks) */
  aki_mma_rect* rects; /* [B][max_rects] */
  uint64_t* col_valid_bits; /* [B][ceil(L_out/64)] */
  int32_t* seq_lens;        /* [B] = L_i */
  int64_t max_original_id;
  int64_t media_token_id;
  int64_t pad_token_id;
  int32_t B, T, T_img, Nv, d, L_out;
  int32_t max_rects;
  int32_t padding_side; /* 0 = right, 1 = left (embeds/labels only, src/utils.py:88-92) */
  int32_t dtype;
} aki_splice_args;

int aki_splice_plan(const int64_t* lang_x, int32_t B, int32_t T, int64_t media_token_id, int64_t assistant_token_id,
                    int32_t Nv, int32_t* plan, void* stream);
int aki_splice_fwd(const aki_splice_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Decode path (after the MMA prefill): one new token per sequence.
 * Replaces the per-token steps of `lang_model.generate` as driven by src/aki.py:136-209 and the patched
 * `_update_model_kwargs_for_generation` (src/aki_generation.py:36-86): after the prefill the mask is all ones, i.e.
 * the new token attends to everything cached.  aki_linear_fwd streams weights with a GEMV kernel when M <= 8.
 * aki_rope_append_fwd : qkv [B,3*H*Dh] of the new tokens -> rotated q [B,H,Dh]; rotated k and v appended to the caches
 *                       [B,H,capacity,Dh] at index cache_len[b] (device int32); pos[b] = row into cos/sin.
 * aki_decode_attn_fwd : o [B,H*Dh] = softmax(q K^T * scale) V over the first n_keys[b] cache rows (device int32),
 *                       optional valid bits [B][nwords] of the prefill columns (padding), NULL = all valid.
 *                       bf16/Dh=96 runs split over the keys (flash-decoding) and needs the workspace below; max_keys is
 *                       a HOST upper bound of max_b n_keys[b] (<= capacity; 0 = capacity) that sizes the grid - keys
 *                       beyond it are not visited.
 * aki_decode_attn_fused_fwd : the two calls above in one launch (bf16, Dh=96): qkv rows in, rotated k / v appended at
 *                       cache_len[b] (which is also the RoPE position), attention over cache_len[b]+1 keys.
 *                       Returns AKI_ERR_UNSUPPORTED for f32 / other head sizes (use the two separate calls).
 * Workspace: aki_decode_attn_workspace_bytes(B,H,Dh,capacity) bytes, caller-owned, ZERO-FILLED ONCE before its first
 *                       use and then passed unchanged from call to call (it holds per-row arrival counters that the
 *                       kernel re-arms itself); calls sharing one workspace must be stream-ordered.
 * aki_decode_linear_fwd : aki_linear_fwd for M <= 8 bf16 rows with the layer's RMSNorm applied to x on the way in:
 *                       y = act(rmsnorm(x; rms_weight, eps) W^T + bias) [+ residual]  (HF:phi3/modeling_phi3.py:266-284
 *                       pre-norm followed by qkv_proj / gate_up_proj).  AKI_ERR_UNSUPPORTED if M > 8 or M*K > 64 Ki.
 * ---------------------------------------------------------------------------------------------- */
int aki_rope_append_fwd(const void* qkv, const float* cos, const float* sin, const int32_t* pos, const int32_t* cache_len,
                        void* q_out, void* k_cache, void* v_cache, int32_t B, int32_t H, int32_t Dh, int32_t capacity,
                        int32_t dtype, void* stream);
size_t aki_decode_attn_workspace_bytes(int32_t B, int32_t H, int32_t Dh, int32_t capacity);
int aki_decode_attn_fwd(const void* q, const void* k_cache, const void* v_cache, void* o, const int32_t* n_keys,
                        const uint64_t* col_valid_bits, int32_t nwords, int32_t B, int32_t H, int32_t Dh, int32_t capacity,
                        int32_t max_keys, float scale, int32_t dtype, void* ws, size_t ws_bytes, void* stream);
int aki_decode_attn_fused_fwd(const void* qkv, const float* cos, const float* sin, const int32_t* cache_len, void* k_cache,
                              void* v_cache, void* o, const uint64_t* col_valid_bits, int32_t nwords, int32_t B, int32_t H,
                              int32_t Dh, int32_t capacity, int32_t max_keys, float scale, int32_t dtype, void* ws,
                              size_t ws_bytes, void* stream);
int aki_decode_linear_fwd(const aki_linear_args* args, const void* rms_weight, float rms_eps, void* stream);

/* aki_decode_chain_fwd - ALL decoder layers of one decode step for ONE sequence (batch 1) in one launch (bf16 weights, or
 * AKI_DT_W8A16: e4m3 weights with one f32 scale per weight row).  Same arithmetic as the per-layer calls above
 * (aki_decode_linear_fwd -> aki_decode_attn_fused_fwd -> aki_linear_fwd + residual -> aki_decode_linear_fwd SwiGLU -> aki_linear_fwd
 * + residual), bit for bit; what changes is the schedule: every workgroup issues its weight loads before it waits for its
 * input vector, so the weight stream runs ahead of the dependency chain across phase and layer seams (decode_chain.hip).
 *   layers        DEVICE array of n_layers descriptors (weights as stored by nn.Linear, [N, K] row-major, K contiguous;
 *                 k_cache / v_cache [1, H, capacity, 96] bf16; s_* = per-row scales for AKI_DT_W8A16, else NULL)
 *   h_in / h_out  [d] bf16: the new token's embedding in, the residual stream after the last layer out (PRE final norm)
 *   cache_len     device int32 [1]: tokens cached so far = the new token's position and append index (not advanced here)
 *   max_keys      host upper bound of cache_len + 1 that sizes the attention split (0 = capacity)
 *   workspace     aki_decode_chain_workspace_bytes(...) bytes, 256-byte aligned, ZERO-FILLED ONCE by the caller before its
 *                 first use and owned by the calls from then on (one call in flight per workspace): it holds two sets of arrival
 *                 counters and the count of completed calls - a call uses the set of its parity and leaves the other one zeroed
 *                 for its successor, so a decode step is ONE launch.  The 32-bit word at aki_decode_chain_error_offset(...) is
 *                 sticky: 0 = every wait of every call so far was satisfied; otherwise (layer << 8 | phase) of a wait that gave
 *                 up after its bounded spin (the launch then drains and h_out is garbage) - read it after synchronising, and
 *                 zero-fill the workspace again before any further call on it.
 * Supported: Dh = 96, d = H * 96 = 3072, F = 8192 (Phi-3.5-mini); AKI_ERR_UNSUPPORTED otherwise (use the per-layer calls). */
typedef struct {
  const void* w_qkv;
  const void* w_o;
  const void* w_gate_up;
  const void* w_down;
  const void* norm1;
  const void* norm2;
  void* k_cache;
  void* v_cache;
  const float* s_qkv;
  const float* s_o;
  const float* s_gate_up;
  const float* s_down;
} aki_decode_chain_layer;

typedef struct {
  const aki_decode_chain_layer* layers;
  const void* h_in;
  void* h_out;
  const float* cos;
  const float* sin;
  const int32_t* cache_len;
  const uint64_t* col_valid_bits;
  void* workspace;
  size_t workspace_bytes;
  int32_t n_layers;
  int32_t nwords;
  int32_t d;
  int32_t H;
  int32_t Dh;
  int32_t F;
  int32_t capacity;
  int32_t max_keys;
  float scale;
  float rms_eps;
  int32_t dtype;
  int32_t batch; /* 0 or 1: one sequence.  2..8 (bf16): that many sequences per step - h_in / h_out are [batch, d] rows, cache_len [batch],
                    col_valid_bits [batch][nwords], k_cache / v_cache [batch, H, capacity, 96]; the GEMV phases become 16-feature MFMA tiles and
                    the step reproduces the per-layer batched calls (aki_decode_linear_fwd / aki_linear_fwd at 2-8 rows,
                    aki_decode_attn_fused_fwd) bit for bit.  Workspace: the *_batch_* functions below. */
} aki_decode_chain_args;

size_t aki_decode_chain_workspace_bytes(int32_t n_layers, int32_t d, int32_t H, int32_t F, int32_t capacity);
size_t aki_decode_chain_error_offset(int32_t n_layers, int32_t H);
size_t aki_decode_chain_batch_workspace_bytes(int32_t n_layers, int32_t d, int32_t H, int32_t F, int32_t capacity, int32_t batch);
size_t aki_decode_chain_batch_error_offset(int32_t n_layers, int32_t H, int32_t batch);
int aki_decode_chain_fwd(const aki_decode_chain_args* args, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Training step (BASELINE configs[2]).  The reference's backward is torch autograd over its
 * eager forward under bf16 autocast, followed by clip_grad_norm_(1.0) and AdamW (train/train_utils.py:242-266,
 * train/train.py:330-337, train/losses.py:83-116).  These entry points are the kernels an autograd wrapper of the
 * forward ops needs; all of them are bf16 (dtype must be AKI_DT_BF16), caller-owned buffers, stream-ordered.
 * GEMM-shaped gradients: dX = dY W runs through aki_linear_fwd on W^T (aki_transpose, once per optimizer step), dW = dY^T X through aki_gemm_tn.
 *
 * aki_attn_bwd      backward of aki_mma_attn_core_fwd (masked = 1; needs Lq == Lk) or of aki_attn_fwd (masked = 0):
 *                   q,k,v [B,H,L,Dh] (k rotated) as the forward saw them, o and d_o [B,Lq,H*Dh], lse [B,H,Lq] from the
 *                   forward -> dq,dk,dv [B,H,L,Dh].  Dh 96 or 64.  Rows >= seq_lens[b] get zero gradient.
 * aki_transpose     y[C][ldy] = x[R][C]^T with columns R..Rpad-1 of y zero-filled (Rpad <= ldy).
 * aki_gemm_tn       c[I][ldc] = sum over the Kc rows of a[Kc][lda]^T b[Kc][ldb]: the weight gradient dW = dY^T X on dY [M,N] and X [M,K] AS THEY LIE
 *                   (contraction over the row index of both; the operands are staged row-major and read transposed from LDS - no
 *                   aki_transpose pass).  I, J, lda, ldb multiples of 8, ldc of 4; a, b 16-byte aligned, c 8-byte aligned.
 * aki_norm_bwd      RMSNorm (rms=1) / LayerNorm backward: dx [rows,cols] (+ dres when given: the gradient arriving through
 *                   the residual branch of a pre-norm block, so the two are summed without a separate pass); dw (and db
 *                   for LayerNorm) [cols], written or accumulated (accumulate=1).  cols % 8 == 0, cols <= 4096.
 * aki_colsum        out[c] (+)= sum_r x[r][c]   (bias gradients)
 * aki_swiglu_fwd/_bwd   a = up * silu(gate) on gate_up [rows, 2F] (gate first), and its backward to d(gate_up)
 * aki_gelu_fwd/_bwd     erf GELU (src/helpers.py:32-39), n elements, n % 8 == 0
 * aki_rope_bwd_merge    dq,dk,dv [B,H,L,Dh] -> d(qkv) [B*L, 3*H*Dh] through the transpose of the rotate-half RoPE
 * aki_ce_loss_fwd_bwd   HF shifted cross-entropy: row (b,t) against labels[b][t+1], ignore_index -100;
 *                       loss_rows [B*L] f32 (sum / *n_valid = loss), n_valid: device int32 (written),
 *                       dlogits (may alias logits, may be NULL) = d(mean loss)/d(logits) * gscale
 * aki_ce_rows_fwd_bwd   the same arithmetic for an arbitrary CHUNK of rows: targets[r] is the (already shifted) class of row r
 *                       (< 0 or >= V: ignored row), *n_valid (device int32) is an INPUT - the number of scored rows of the
 *                       whole batch.  With it the lm_head and the loss run chunk by chunk and the [B, L, V] logits tensor
 *                       is never formed (src/helpers.py:594-603 + train/losses.py:83-116).
 * aki_grad_sqnorm       *out (+)= sum g^2 over a gradient buffer (n % 8 == 0; dtype AKI_DT_BF16, or AKI_DT_F32 for the fp32 exchange)
 * aki_adamw_step        fp32 master weights p, moments m, v; bf16 gradients g -> updated p/m/v and bf16 weights w16.
 *                       g is scaled by gscale (1/world, 1/grad_accum) and, when sqnorm != NULL and max_norm > 0, clipped by
 *                       min(1, max_norm / (sqrt(*sqnorm) * gscale + 1e-6))  (torch.nn.utils.clip_grad_norm_ semantics).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q; const void* k; const void* v;
  const void* o; const void* d_o;
  const float* lse;
  void* dq; void* dk; void* dv;
  const aki_mma_rect* rects; int32_t max_rects;
  const uint64_t* col_valid_bits;
  const int32_t* seq_lens;
  int32_t masked;
  int32_t B, H, Lq, Lk, Dh;
  float scale;
  int32_t dtype;
} aki_attn_bwd_args;

size_t aki_attn_bwd_workspace_bytes(int32_t B, int32_t H, int32_t Lq);
int aki_attn_bwd(const aki_attn_bwd_args* args, void* workspace, size_t workspace_bytes, void* stream);
int aki_transpose(const void* x, void* y, int32_t R, int32_t C, int32_t ldx, int32_t ldy, int32_t Rpad, int32_t dtype, void* stream);
int aki_gemm_tn(const void* a, const void* b, void* c, int32_t Kc, int32_t I, int32_t J, int64_t lda, int64_t ldb, int64_t ldc, int32_t dtype,
                void* stream);
size_t aki_norm_bwd_workspace_bytes(int32_t cols);
int aki_norm_bwd(int32_t rms, const void* x, const void* w, const void* dy, const void* dres, void* dx, void* dw, void* db,
                 int32_t rows, int32_t cols, int32_t ldx, int32_t lddy, int32_t lddr, int32_t lddx, float eps, int32_t accumulate,
                 int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);
size_t aki_colsum_workspace_bytes(int32_t cols);
int aki_colsum(const void* x, void* out, int32_t rows, int32_t cols, int32_t ldx, int32_t accumulate, int32_t dtype, void* workspace,
               size_t workspace_bytes, void* stream);
int aki_swiglu_fwd(const void* gate_up, void* a, int32_t rows, int32_t F, int32_t ldg, int32_t lda, int32_t dtype, void* stream);
int aki_swiglu_bwd(const void* gate_up, const void* da, void* dgate_up, int32_t rows, int32_t F, int32_t ldg, int32_t ldda,
                   int32_t lddg, int32_t dtype, void* stream);
int aki_gelu_fwd(const void* x, void* y, size_t n, int32_t dtype, void* stream);
int aki_gelu_bwd(const void* x, const void* dy, void* dx, size_t n, int32_t dtype, void* stream);
int aki_rope_bwd_merge(const void* dq, const void* dk, const void* dv, const float* cos, const float* sin, const int32_t* position_ids,
                       void* dqkv, int32_t B, int32_t H, int32_t L, int32_t Dh, int32_t dtype, void* stream);
int aki_ce_loss_fwd_bwd(const void* logits, const int64_t* labels, int32_t* n_valid, float* loss_rows, void* dlogits, int32_t B,
                        int32_t L, int32_t V, int32_t ldl, int32_t lddl, float gscale, int32_t dtype, void* stream);
int aki_ce_rows_fwd_bwd(const void* logits, const int64_t* targets, const int32_t* n_valid, float* loss_rows, void* dlogits,
                        int32_t rows, int32_t V, int32_t ldl, int32_t lddl, float gscale, int32_t dtype, void* stream);
size_t aki_grad_sqnorm_workspace_bytes(void);
int aki_grad_sqnorm(const void* g, size_t n, float* out, int32_t accumulate, int32_t dtype, void* workspace, size_t workspace_bytes,
                    void* stream);
int aki_adamw_step(float* p, float* m, float* v, const void* g, void* w16, size_t n, const float* sqnorm, float max_norm, float gscale,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);
/* aki_adamw_step for ONE 2-D weight [N, K] (K % 4 == 0) that also writes W^T [K, ldT] (ldT = N rounded up to x64, padding columns zero) -
 * what aki_transpose would make of the updated bf16 weight, in the same pass (the backward's input-gradient GEMMs read W^T).
 * grad_dtype: AKI_DT_BF16 or AKI_DT_F32. */
int aki_adamw_step_t(float* p, float* m, float* v, const void* g, void* w16, void* wT, int32_t N, int32_t K, int32_t ldT, const float* sqnorm,
                     float max_norm, float gscale, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                     int32_t grad_dtype, void* stream);
/* the same update from FP32 gradients: the reference's DDP path under `--precision amp_bf16` keeps fp32 parameters, so the gradients
 * it all-reduces and its optimizer consumes are fp32 (train/train.py:311-312, train/train_utils.py:56-65) - AkiTrainer(reduce_dtype=float32) */
int aki_adamw_step_g32(float* p, float* m, float* v, const float* g, void* w16, size_t n, const float* sqnorm, float max_norm, float gscale,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);

/* aki_quant_rows_fp8 - q[r][c] = e4m3(y[r][c] / scale[r]), scale[r] = max_c |y[r][c]| / 448, with y = x (bf16 [rows, cols])
 * or, when rms_weight != NULL, y = Phi3RMSNorm(x; rms_weight, rms_eps) (HF:phi3/modeling_phi3.py:266-284) - the input side
 * of the fp8 projections of BASELINE configs[4].  Also used once per nn.Linear weight (one scale per output feature).
 * cols % 8 == 0, cols <= 8192; q is [rows, ldq] bytes. */
int aki_quant_rows_fp8(const void* x, const void* rms_weight, float rms_eps, void* q, float* scale, int32_t rows, int32_t cols,
                       int32_t ldx, int32_t ldq, void* stream);

/* aki_mma_mask_dense - materialise the reference's (B,1,L,L) int64 0/1 mask from the table, for
 * callers that still want it (bit-exact vs src/vlm.py:410-443 + src/utils.py:99-108). */
int aki_mma_mask_dense(const aki_mma_rect* rects, int32_t max_rects, const uint64_t* col_valid_bits,
                       const int32_t* seq_lens, int32_t B, int32_t L, int64_t* out, void* stream);

/* aki_mma_mask_to_table - the inverse: the reference's LM hand-off type, `attention_mask` (B,1,L,L) int64 0/1 as returned
 * by `_prepare_inputs_for_forward` (src/vlm.py:589-603) and passed to `lang_model(...)` (src/aki.py:125-130), converted on
 * the device into the table the attention kernels consume.
 *   mask [B,1,L,L] int64 (non-zero = visible) -> rects [B][max_rects], col_valid_bits [B][ceil(L/64)], seq_lens [B],
 *   status [B] device int32: 0 = ok, n > max_rects = the sample needs n row groups (its rects are then incomplete).
 *   workspace: aki_mma_mask_to_table_workspace_bytes(B, L) bytes.
 * The result is a CANDIDATE: only masks of the family {causal triangle + row-interval rectangles + invalid columns +
 * empty bottom rows} are representable.  The caller proves it by materialising the table again with aki_mma_mask_dense
 * and comparing with the input bit for bit (aki_amd/phi3.py does; anything else is refused, never approximated). */
size_t aki_mma_mask_to_table_workspace_bytes(int32_t B, int32_t L);
int aki_mma_mask_to_table(const int64_t* mask, int32_t B, int32_t L, int32_t max_rects, aki_mma_rect* rects,
                          uint64_t* col_valid_bits, int32_t* seq_lens, int32_t* status, void* workspace, size_t workspace_bytes,
                          void* stream);

/* aki_sft_collate_pad - the SFT collate (train/sft_data_utils/loader_utils.py:11-91 `_pad_trunc` + `batch_collate_pad`) on the
 * device: B ragged samples, concatenated (sample b = elements offsets[b] .. offsets[b+1]-1 of ids / labels / attention_mask),
 * -> [B, T_out] int64 arrays.  A sample of T_out tokens or more keeps its FIRST T_out tokens; a shorter one is padded on the
 * right (padding_side 0) or left (1) with pad_token_id / ignore_index / 0.  T_out is the caller's: `max_length + 1` for
 * padding="max_length" (loader_utils.py:78-82), the longest sample for "longest" (:30-31).  labels / attention_mask and their
 * outputs may be NULL together. */
int aki_sft_collate_pad(const int64_t* ids, const int64_t* labels, const int64_t* attention_mask, const int32_t* offsets, int32_t B,
                        int32_t T_out, int64_t pad_token_id, int64_t ignore_index, int32_t padding_side, int64_t* out_ids,
                        int64_t* out_labels, int64_t* out_mask, void* stream);

/* aki_greedy_pick - the step between two decode steps of a greedy `generate` (HF GenerationMixin's greedy branch, which
 * src/aki.py:136-209 drives through src/aki_generation.py:36-86): per row b of bf16 logits [B, ld], V columns used,
 *   next = done[b] ? pad_token_id : argmax_v logits[b, v]     (ties: the lower index; a NaN outranks every number - torch.argmax)
 *   t = cache_len[b] + advance - start_len[b]                 (0 when cache_len is NULL; start_len NULL = zeros)
 *   tokens[b, t] = next (when tokens != NULL and 0 <= t < tokens_ld);  next_ids[b] = next
 *   a row that was not done and whose next is one of eos_ids becomes done (done[b] = 1, done_at[b] = t)
 *   advance = 1: cache_len[b] += 1 (the decode step's own advance, folded into this launch)
 * done / tokens / cache_len / start_len / done_at may be NULL.  One launch, no host value: it sits inside the replayed hipGraph
 * of a decode step, so a greedy token is one replay and the host looks at `done` every few tokens. */
int aki_greedy_pick(const void* logits, int32_t B, int32_t V, int64_t ld, const int64_t* eos_ids, int32_t n_eos, int64_t pad_token_id,
                    uint8_t* done, int64_t* next_ids, int64_t* tokens, int32_t tokens_ld, int32_t* cache_len, const int32_t* start_len,
                    int32_t advance, int32_t* done_at, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layer loops issued from ONE call (host code; csrc/stack.hip).  A one-sample prefill - the shape the reference's callers run
 * (local_demo.py:75-87, eval_cv_bench/eval.py:92-104) - is ~400 launches of 10-70 us: issued one by one from Python the HOST was 9 ms
 * of a 15 ms first token.  These two functions issue, on `stream`, exactly the launches of the per-layer entry points above, in
 * the order of HF:phi3/modeling_phi3.py `Phi3Model.forward` (:287-328 per layer) and HF:siglip/modeling_siglip.py:329-354 -
 * bf16 inference with the normalisations folded into the GEMMs (aki_linear_args: row_scale / row_shift / stats_*).
 *
 * aki_decoder_stack_fwd: per layer  q,k,v = RoPE(rstd (h W_qkv'^T)) -> span attention -> h1 = h + o W_o^T -> act = SwiGLU(rstd1 (h1 W_gu'^T))
 * -> h2 = h1 + act W_down^T, every h-writing GEMM leaving 1/rms of its rows for the next one.  W' = W diag(RMSNorm gain), prepared by the
 * caller.  k_cache / v_cache ([B,H,kv_capacity,Dh], all layers or none): rotated keys and values are written there (KV-cache prefill).
 *   h_in, h_out [B*L, d] bf16 (may not alias); rstd_out [B*L] f32 or NULL: 1/rms(h_out) for the final norm folded into the head;
 *   workspace: aki_decoder_stack_workspace_bytes() bytes, 256-byte aligned, scratch; stats_workspace / splitk_workspace as in
 *   aki_linear_args (zero-filled once by the caller; splitk optional).  `layers` is a HOST array. */
typedef struct {
  const void* w_qkv;     /* [3*H*Dh, d] */
  const void* w_o;       /* [d, H*Dh] */
  const void* w_gate_up; /* [2F, d] */
  const void* w_down;    /* [d, F] */
  void* k_cache;
  void* v_cache;
} aki_decoder_layer;

typedef struct {
  const aki_decoder_layer* layers;
  int32_t n_layers;
  const void* h_in;
  void* h_out;
  float* rstd_out;
  const float* cos;
  const float* sin;
  const int32_t* position_ids;
  int32_t pos_rows;
  const aki_mma_rect* rects;
  const uint64_t* col_valid_bits;
  const int32_t* seq_lens;
  int32_t max_rects;
  int32_t B, H, L, Dh, d, F;
  int32_t kv_capacity;   /* rows per (batch, head) of the caches; 0 with no caches */
  float scale;
  float rms_eps;
  int32_t dead_rows;
  void* workspace;
  size_t workspace_bytes;
  void* stats_workspace;
  size_t stats_workspace_bytes;
  void* splitk_workspace;
  size_t splitk_workspace_bytes;
} aki_decoder_stack_args;

size_t aki_decoder_stack_workspace_bytes(int32_t B, int32_t H, int32_t L, int32_t Dh, int32_t d, int32_t F, int32_t keep_kv);
int aki_decoder_stack_fwd(const aki_decoder_stack_args* args, void* stream);

/* aki_siglip_stack_fwd: per layer  qkv = LN1(h) W_qkv^T + b (folded: rstd (h W'^T - mean c) + b') -> 16-head attention read in place out
 * of qkv -> h1 = h + out_proj -> fc1 + GELU with LN2 folded, into fc1_out -> h2 = h1 + fc2.  W', b', c = fold of the LayerNorm gain /
 * bias into the weight (see aki_linear_args), prepared by the caller; w_fc2 is K-padded to Ip (a multiple of 64) with zero columns.
 *   h_in, h_out [N*L, E] bf16; fc1_out [N*L, Ip] bf16, caller-owned, its pad columns [I, Ip) ZERO (nothing here writes them);
 *   workspace: aki_siglip_stack_workspace_bytes() bytes, 256-byte aligned, scratch.  `layers` is a HOST array. */
typedef struct {
  const void* w_qkv;  /* [3E, E] */
  const void* b_qkv;  /* [3E] */
  const float* c_qkv; /* [3E] f32 */
  const void* w_out;  /* [E, E] */
  const void* b_out;  /* [E] or NULL */
  const void* w_fc1;  /* [I, E] */
  const void* b_fc1;  /* [I] */
  const float* c_fc1; /* [roundup(I, 4)] f32 */
  const void* w_fc2;  /* [E, Ip] */
  const void* b_fc2;  /* [E] or NULL */
} aki_siglip_layer;

typedef struct {
  const aki_siglip_layer* layers;
  int32_t n_layers;
  const void* h_in;
  void* h_out;
  void* fc1_out;
  int32_t N, L, E, heads, I, Ip;
  int32_t act;       /* AKI_ACT_GELU_TANH (SigLIP) or AKI_ACT_GELU_ERF */
  float ln_eps;
  float scale;       /* softmax scale, (E / heads)^-0.5 as the caller's module computed it */
  void* workspace;
  size_t workspace_bytes;
  void* stats_workspace;
  size_t stats_workspace_bytes;
  void* splitk_workspace;
  size_t splitk_workspace_bytes;
} aki_siglip_stack_args;

size_t aki_siglip_stack_workspace_bytes(int32_t N, int32_t L, int32_t E, int32_t heads);
int aki_siglip_stack_fwd(const aki_siglip_stack_args* args, void* stream);

/* aki_perceiver_stack_fwd: the Perceiver connector (src/helpers.py:170-199: `depth` x [PerceiverAttention + residual, FeedForward + residual],
 * final LayerNorm, projection) for ONE (sample, image) pair - the one-sample case, where issuing its ~55 launches from Python takes longer than
 * they run.  x [n1, D] media features, latents [n2, D] (the learned latents), out [n2, D_out] (D_out = D without a projection).  Every
 * LayerNorm uses `eps`; the attention is softmax(scale q k^T) over the n1 + n2 keys LN_media(x) ++ LN_latents(latents).  Same launches as
 * aki_layernorm_fwd / aki_linear_fwd / aki_attn_fwd / aki_connector_mlp_fwd / aki_connector_proj_fwd issued one by one. */
typedef struct {
  const void* norm_media_w;
  const void* norm_media_b;
  const void* norm_latents_w;
  const void* norm_latents_b;
  const void* w_q;   /* [heads*dim_head, D] */
  const void* w_kv;  /* [2*heads*dim_head, D] */
  const void* w_out; /* [D, heads*dim_head] */
  const void* ff_ln_w;
  const void* ff_ln_b;
  const void* ff_w1; /* [d_ff, D] */
  const void* ff_w2; /* [D, d_ff] */
} aki_perceiver_layer;

typedef struct {
  const aki_perceiver_layer* layers;
  int32_t n_layers;
  const void* x;
  const void* latents;
  const void* norm_w;
  const void* norm_b;
  const void* proj_w; /* [D_out, D] or NULL */
  const void* proj_b;
  void* out;
  int32_t n1, n2, D, heads, dim_head, d_ff, D_out;
  float scale;
  float eps;
  void* workspace;
  size_t workspace_bytes;
} aki_perceiver_stack_args;

size_t aki_perceiver_stack_workspace_bytes(int32_t n1, int32_t n2, int32_t D, int32_t heads, int32_t dim_head, int32_t d_ff);
int aki_perceiver_stack_fwd(const aki_perceiver_stack_args* args, void* stream);

/* aki_greedy_pick_embed - aki_greedy_pick + the embedding lookup of the token it picked (`DecoupledEmbedding.forward`,
 * src/helpers.py:440-492, which HF's generate loop runs at the top of the next step): next_embeds[b, :] = bf16 row `next` of embed_weight
 * [max_original_id + 1 or more rows, d], or row `next - max_original_id - 1` of additional_weight [num_additional, d] when
 * next > max_original_id (additional_weight NULL: one table).  d % 8 == 0, 16-byte aligned tables; V may not exceed the number of rows.
 * next_embeds is what the following decode step takes as h_in: a greedy token is decode chain + head + this launch. */
int aki_greedy_pick_embed(const void* logits, int32_t B, int32_t V, int64_t ld, const int64_t* eos_ids, int32_t n_eos, int64_t pad_token_id,
                          uint8_t* done, int64_t* next_ids, int64_t* tokens, int32_t tokens_ld, int32_t* cache_len, const int32_t* start_len,
                          int32_t advance, int32_t* done_at, const void* embed_weight, const void* additional_weight, int64_t max_original_id,
                          int64_t num_additional, int32_t d, void* next_embeds, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AKI_MI355X_H */
