// api.hip - extern "C" entry points of libaki_mi355x.so (include/aki_mi355x.h).
// Host-side validation only; every launch goes to the stream the caller passes.
#include "aki_device.h"

namespace aki {
int linear_bf16(const aki_linear_args* a, hipStream_t stream);
size_t linear_stats_ws_bytes(int M, int n_out);
size_t linear_stats_cnt_bytes(int M);
int row_stats_launch(const void* x, int rows, int cols, int ldx, float eps, float* rstd, float* mean, hipStream_t s);
int linear_f32(const aki_linear_args* a, hipStream_t stream);
int qkv_rope_bf16(const aki_mma_attn_args* a, void* q, void* k, void* v, hipStream_t stream);
int qkv_rope_fp8(const aki_mma_attn_args* a, void* q, void* k, void* v, hipStream_t stream);
int linear_fp8(const aki_linear_args* a, hipStream_t stream);
int quant_rows_fp8_launch(const void* x, const void* rms_w, float eps, void* q, float* scale, int rows, int cols, int ldx, int ldq,
                          hipStream_t s);
int qkv_rope_f32(const aki_mma_attn_args* a, void* q, void* k, void* v, float* tmp, hipStream_t stream);
int attn_core_bf16(const aki_mma_attn_core_args* a, void* ws, size_t ws_bytes, hipStream_t stream);
int attn_core_f32(const aki_mma_attn_core_args* a, void* ws, size_t ws_bytes, hipStream_t stream, int causal);
int attn_nc_bf16(const aki_attn_args* a, hipStream_t stream);
int gemv_bf16(const aki_linear_args* a, const void* rms_w, float eps, hipStream_t stream);
int skinny_gemm_bf16(const aki_linear_args* a, const void* rms_w, float eps, hipStream_t stream);
int skinny_gemm_w8(const aki_linear_args* a, const void* rms_w, float eps, hipStream_t stream);
size_t decode_attn_ws_bytes(int B, int H, int Dh, int cap);
int decode_attn_split_launch(const void* q_or_qkv, const float* cos, const float* sin, const int* len, void* kc, void* vc, void* o,
                             const uint64_t* vbits, int nwords, int B, int H, int cap, int max_keys, float scale, bool fused,
                             void* ws, size_t ws_bytes, hipStream_t s);
int rope_append_launch(const void* qkv, const float* cos, const float* sin, const int* pos, const int* cache_len, void* q_out,
                       void* k_cache, void* v_cache, int B, int H, int Dh, int cap, int dtype, hipStream_t s);
size_t decode_chain_ws_bytes(int n_layers, int d, int H, int F, int cap);
size_t decode_chain_err_offset(int n_layers, int H);
size_t decode_chain_b_ws_bytes(int n_layers, int d, int H, int F, int cap, int B);
size_t decode_chain_b_err_offset(int n_layers, int H, int B);
int decode_chain_launch(const aki_decode_chain_args* a, hipStream_t stream);
int decode_attn_launch(const void* q, const void* kc, const void* vc, void* o, const int* n_keys, const uint64_t* vbits, int nwords,
                       int B, int H, int Dh, int cap, float scale, int dtype, hipStream_t s);
int norm_launch(bool rms, const void* x, const void* w, const void* b, void* y, int rows, int cols, int ldx, int ldy,
                float eps, int dtype, hipStream_t stream);
int splice_plan_launch(const int64_t* lang_x, int B, int T, int64_t media, int64_t assistant, int Nv, int* plan, hipStream_t s);
int splice_launch(const aki_splice_args* a, hipStream_t s);
int mask_dense_launch(const aki_mma_rect* rects, int max_rects, const uint64_t* vbits, const int* seq_lens, int B, int L,
                      int64_t* out, hipStream_t s);
int im2col_launch(const void* pix, void* out, int N, int S, int P, int Kp, int dtype, hipStream_t s);
int greedy_pick_launch(const void* logits, int B, int V, int ld, const int64_t* eos, int n_eos, int64_t pad, unsigned char* done, int64_t* ids,
                       int64_t* tokens, int tokens_ld, int* cache_len, const int* start_len, int advance, int* done_at, const void* emb_main,
                       const void* emb_extra, int64_t max_original_id, int d, void* emb_out, hipStream_t s);
int sft_collate_launch(const int64_t* ids, const int64_t* labels, const int64_t* mask, const int* offsets, int B, int T_out,
                       int64_t pad_id, int64_t ignore_index, int left, int64_t* out_ids, int64_t* out_labels, int64_t* out_mask,
                       hipStream_t s);
size_t mask_to_table_ws_bytes(int B, int L);
int mask_to_table_launch(const int64_t* mask, int B, int L, int max_rects, aki_mma_rect* rects, uint64_t* vbits, int* seq_lens,
                         int* status, void* ws, hipStream_t s);
size_t linear_splitk_plan_ws_bytes(int M, int N, int K);
#ifdef AKI_LAB_HOOKS
extern int g_sm_variant;
extern int g_sk_slice_major;
extern int g_sm_ksplit;
extern int g_force_tile;
extern int g_deep_ring;
extern int g_pipe;
extern int g_deepx;
extern int g_attn_variant;
extern long long* g_clock_probe;
extern int g_probe_block;
#endif
size_t attn_bwd_ws_bytes(int B, int H, int Lq);
int attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq, void* dk,
                  void* dv, const aki_mma_rect* rects, int max_rects, const uint64_t* vbits, const int* seq_lens, int masked, int B,
                  int H, int Lq, int Lk, int Dh, float scale, void* ws, size_t ws_bytes, hipStream_t s);
int transpose_bf16_launch(const void* x, void* y, int R, int C, int ldx, int ldy, int Rpad, hipStream_t s);
int gemm_tn_launch(const void* a, const void* b, void* c, int Kc, int I, int J, long lda, long ldb, long ldc, hipStream_t stream);
size_t norm_bwd_ws_bytes(int cols);
int norm_bwd_launch(bool rms, const void* x, const void* w, const void* dy, const void* dres, void* dx, void* dw, void* db, int rows,
                    int cols, int ldx, int lddy, int lddx, int lddr, float eps, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
size_t colsum_ws_bytes(int cols);
int colsum_launch(const void* x, void* out, int rows, int cols, int ldx, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
int swiglu_launch(int bwd, const void* gu, const void* da, void* out, int rows, int F, int ldg, int ldda, int ldo, hipStream_t s);
int gelu_launch(int bwd, const void* x, const void* dy, void* out, size_t n, hipStream_t s);
int rope_bwd_merge_launch(const void* dq, const void* dk, const void* dv, const float* cos, const float* sin, const int* pos, void* dqkv,
                          int B, int H, int L, int Dh, hipStream_t s);
int ce_launch(const void* logits, const int64_t* labels, int* n_valid, float* loss_rows, void* dlogits, int B, int L, int V, int ldl,
              int lddl, float gscale, hipStream_t s);
int ce_rows_launch(const void* logits, const int64_t* targets, const int* n_valid, float* loss_rows, void* dlogits, int rows, int V,
                   int ldl, int lddl, float gscale, hipStream_t s);
size_t grad_sqnorm_ws_bytes();
int grad_sqnorm_launch(const void* g, size_t n, float* out, int accumulate, void* ws, size_t ws_bytes, bool g32, hipStream_t s);
int adamw_t_launch(float* p, float* m, float* v, const void* g, void* w16, void* wT, int N, int K, int ldT, const float* sqnorm, float max_norm,
                   float gscale, float lr, float beta1, float beta2, float eps, float wd, int step, bool g32, hipStream_t s);
int adamw_launch(float* p, float* m, float* v, const void* g, void* w16, size_t n, const float* sqnorm, float max_norm, float gscale,
                 float lr, float beta1, float beta2, float eps, float wd, int step, bool g32, hipStream_t s);
}  // namespace aki

using namespace aki;

static inline bool dtype_ok(int dt) { return dt == AKI_DT_BF16 || dt == AKI_DT_F32; }

extern "C" {

const char* aki_strerror(int status) {
  switch (status) {
    case AKI_OK: return "ok";
    case AKI_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive size or inconsistent shapes)";
    case AKI_ERR_UNSUPPORTED: return "unsupported dtype / head_dim / size for the gfx950 kernels";
    case AKI_ERR_ALIGNMENT: return "pointer or leading dimension is not 16-byte aligned";
    case AKI_ERR_WORKSPACE: return "workspace missing or too small";
    case AKI_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown aki status";
  }
}

int aki_abi_version(void) { return AKI_ABI_VERSION; }

#ifdef AKI_LAB_HOOKS
// Lab build only (libaki_mi355x_lab.so, `python -m aki_amd.build --lab`): force the bf16 GEMM tile configuration
// (0 = heuristic, 1 = 256x256, 2 = 128x128, 3 = 128 features x 96 tokens where that tile exists, else 128x128); +256 switches
// off the 4-stage / 64-feature variant of single-row launches, +512 the mid-step pipeline of the 256x256 tile.  Process-global,
// not thread safe - which is why the product library does not carry it.
void aki_lab_set_gemm_tile(int mode) {
  aki::g_deep_ring = (mode & 256) ? 0 : 1;
  aki::g_deepx = (mode & 4096) ? 1 : ((mode & 8192) ? 2 : 0);   // residual GEMMs: three-deep ring on the tokens (forced / forbidden)
  aki::g_pipe = (mode & 512) ? 0 : ((mode & 1024) ? 2 : ((mode & 2048) ? 3 : 1));     // +1024: pipeline without the residual prefetch, +2048: with the two-deep weight ring
  mode &= 255;
  aki::g_force_tile = (mode >= 1 && mode <= 5) ? mode : 0;
}
// split-K workgroup order: 0 = tile-major (product), 1 = slice-major
void aki_lab_set_slice_major(int on) { aki::g_sk_slice_major = on ? 1 : 0; }
// small-M tile variant (gemm_bf16.hip: launch_variant; -1 = the planner) and its K split, for every bf16 GEMM launch
void aki_lab_set_small_m(int variant, int ksplit) { aki::g_sm_variant = variant; aki::g_sm_ksplit = ksplit < 1 ? 1 : ksplit; }
// 0 = product choice, 1 = 32-row attention core (two waves per SIMD), 2 = 64-row core (one wave per SIMD)
void aki_lab_set_attn_variant(int v) { aki::g_attn_variant = v; }
// device pointer to two int64: every bf16 GEMM launch then leaves {shader cycles, 100 MHz wall ticks} of its workgroup 0 there
void aki_lab_set_probe_block(int b) { aki::g_probe_block = b; }   // which workgroup of a GEMM launch stamps
void aki_lab_set_clock_probe(void* int64x32) { aki::g_clock_probe = (long long*)int64x32; }   // 32 int64 (layout: GemmParams::clock_probe)
#endif

// ---- attention core --------------------------------------------------------------------------------
size_t aki_mma_attn_core_workspace_bytes(int32_t B, int32_t H, int32_t L, int32_t Dh, int32_t dtype) {
  AKI_CLEAR_ERR();
  (void)L; (void)dtype;
  return aki_align_up((size_t)B * H * Dh * sizeof(float), 256);
}

int aki_mma_attn_core_fwd(const aki_mma_attn_core_args* a, void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->q && a->k && a->v && a->o);
  AKI_CHECK_ARG(a->B > 0 && a->H > 0 && a->L > 0 && a->Dh > 0 && a->scale > 0.f);
  AKI_CHECK_ARG(dtype_ok(a->dtype));
  AKI_CHECK_ARG(a->max_rects >= 0 && (a->max_rects == 0 || a->rects));
  if (a->dtype == AKI_DT_BF16) return attn_core_bf16(a, ws, ws_bytes, (hipStream_t)stream);
  return attn_core_f32(a, ws, ws_bytes, (hipStream_t)stream, 1);
}

// ---- plain attention (vision side) ---------------------------------------------------------------------
int aki_attn_fwd(const aki_attn_args* a, void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->q && a->k && a->v && a->o);
  AKI_CHECK_ARG(a->B > 0 && a->H > 0 && a->Lq > 0 && a->Lk > 0 && a->Dh > 0 && a->scale > 0.f && dtype_ok(a->dtype));
  if (a->dtype == AKI_DT_BF16) return attn_nc_bf16(a, (hipStream_t)stream);
  // f32 parity path: the simple kernel wants contiguous head-major tensors and equal q/kv lengths are not required
  const int64_t Dh = a->Dh;
  if (a->q_stride_t != Dh || a->k_stride_t != Dh || a->v_stride_t != Dh || a->q_stride_h != (int64_t)a->Lq * Dh ||
      a->k_stride_h != (int64_t)a->Lk * Dh || a->v_stride_h != (int64_t)a->Lk * Dh || a->Lq != a->Lk)
    return AKI_ERR_UNSUPPORTED;
  aki_mma_attn_core_args c = {};
  c.q = a->q; c.k = a->k; c.v = a->v; c.o = a->o; c.B = a->B; c.H = a->H; c.L = a->Lq; c.Dh = a->Dh; c.scale = a->scale;
  c.dtype = AKI_DT_F32; c.dead_rows = AKI_DEAD_ROWS_ZERO;
  return attn_core_f32(&c, ws, ws_bytes, (hipStream_t)stream, 0);
}

// ---- fused MMA op ------------------------------------------------------------------------------------
static size_t qkv_bytes(int B, int H, int L, int Dh, int dtype) {
  return aki_align_up((size_t)B * H * L * Dh * aki_elt_size(dtype), 256);
}

size_t aki_mma_attn_workspace_bytes(int32_t B, int32_t H, int32_t L, int32_t Dh, int32_t dtype) {
  size_t n = 3 * qkv_bytes(B, H, L, Dh, dtype) + aki_mma_attn_core_workspace_bytes(B, H, L, Dh, dtype);
  if (dtype == AKI_DT_F32) n += 3 * qkv_bytes(B, H, L, Dh, dtype);  // un-rotated qkv scratch
  return n;
}

static int check_fused(const aki_mma_attn_args* a) {
  AKI_CHECK_ARG(a && a->x && a->w_qkv && a->cos && a->sin);
  AKI_CHECK_ARG(a->B > 0 && a->H > 0 && a->L > 0 && a->Dh > 0 && a->d_model > 0);
  AKI_CHECK_ARG(a->ldx >= a->d_model && a->ldw >= a->d_model && a->pos_rows > 0);
  AKI_CHECK_ARG(a->position_ids || a->pos_rows >= a->L);
  AKI_CHECK_ARG(dtype_ok(a->dtype) || (a->dtype == AKI_DT_FP8_E4M3 && a->x_scale && a->w_scale));
  AKI_CHECK_ARG(!a->row_scale || a->dtype == AKI_DT_BF16);
  return AKI_OK;
}

int aki_qkv_rope_fwd(const aki_mma_attn_args* a, void* q, void* k, void* v, void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  int rc = check_fused(a);
  if (rc) return rc;
  AKI_CHECK_ARG(q && k && v);
  AKI_CHECK_ARG(a->kv_capacity == 0 || a->kv_capacity >= a->L);
  if (a->dtype == AKI_DT_BF16) return qkv_rope_bf16(a, q, k, v, (hipStream_t)stream);
  if (a->dtype == AKI_DT_FP8_E4M3) return qkv_rope_fp8(a, q, k, v, (hipStream_t)stream);
  // f32 parity path: scratch for the un-rotated projection
  if (!ws || ws_bytes < (size_t)a->B * a->L * 3 * a->H * a->Dh * sizeof(float)) return AKI_ERR_WORKSPACE;
  return qkv_rope_f32(a, q, k, v, (float*)ws, (hipStream_t)stream);
}

int aki_mma_attn_fwd(const aki_mma_attn_args* a, void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  int rc = check_fused(a);
  if (rc) return rc;
  AKI_CHECK_ARG(a->o && a->scale > 0.f);
  AKI_CHECK_ARG(a->max_rects >= 0 && (a->max_rects == 0 || a->rects));
  const int cdt = a->dtype == AKI_DT_FP8_E4M3 ? AKI_DT_BF16 : a->dtype;     // q, k, v, o and the core are bf16 on the fp8 path
  if (!ws || ws_bytes < aki_mma_attn_workspace_bytes(a->B, a->H, a->L, a->Dh, cdt)) return AKI_ERR_WORKSPACE;
  AKI_CHECK_ALIGN16(ws);
  const size_t qb = qkv_bytes(a->B, a->H, a->L, a->Dh, cdt);
  char* w = (char*)ws;
  void* q = w; void* k = w + qb; void* v = w + 2 * qb;
  char* rest = w + 3 * qb;
  if (a->dtype == AKI_DT_BF16) {
    rc = qkv_rope_bf16(a, q, k, v, (hipStream_t)stream);
  } else if (a->dtype == AKI_DT_FP8_E4M3) {
    rc = qkv_rope_fp8(a, q, k, v, (hipStream_t)stream);
  } else {
    rc = qkv_rope_f32(a, q, k, v, (float*)rest, (hipStream_t)stream);
    rest += 3 * qb;
  }
  if (rc) return rc;
  aki_mma_attn_core_args c = {};
  c.q = q; c.k = k; c.v = v; c.o = a->o; c.lse = a->lse; c.rects = a->rects; c.col_valid_bits = a->col_valid_bits;
  c.seq_lens = a->seq_lens; c.max_rects = a->max_rects; c.B = a->B; c.H = a->H; c.L = a->L; c.Dh = a->Dh;
  c.scale = a->scale; c.dtype = cdt; c.dead_rows = a->dead_rows; c.kv_capacity = 0;
  return aki_mma_attn_core_fwd(&c, rest, aki_mma_attn_core_workspace_bytes(a->B, a->H, a->L, a->Dh, cdt), stream);
}

// ---- linear ------------------------------------------------------------------------------------------
size_t aki_linear_stats_counter_bytes(int32_t M) { return M > 0 ? linear_stats_cnt_bytes(M) : 0; }
size_t aki_linear_stats_workspace_bytes(int32_t M, int32_t N_out) { return (M > 0 && N_out > 0) ? aki_align_up(linear_stats_ws_bytes(M, N_out), 256) : 0; }
size_t aki_linear_splitk_workspace_bytes(int32_t M, int32_t N, int32_t K) { return (M > 0 && N > 0 && K > 0) ? aki_align_up(linear_splitk_plan_ws_bytes(M, N, K), 256) : 0; }

int aki_row_stats(const void* x, int32_t rows, int32_t cols, int32_t ldx, float eps, float* rstd, float* mean, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && rstd && rows > 0 && cols > 0 && ldx >= cols && eps > 0.f);
  if (dtype != AKI_DT_BF16) return AKI_ERR_UNSUPPORTED;
  return row_stats_launch(x, rows, cols, ldx, eps, rstd, mean, (hipStream_t)stream);
}

int aki_linear_fwd(const aki_linear_args* a, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->x && a->w && a->y);
  AKI_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0);
  AKI_CHECK_ARG(dtype_ok(a->dtype) || (a->dtype == AKI_DT_FP8_E4M3 && a->x_scale && a->w_scale) || (a->dtype == AKI_DT_W8A16 && a->w_scale));
  AKI_CHECK_ARG(a->act >= AKI_ACT_NONE && a->act <= AKI_ACT_SWIGLU);
  const int n_out = a->act == AKI_ACT_SWIGLU ? a->N / 2 : a->N;
  AKI_CHECK_ARG(a->ldx >= a->K && a->ldw >= a->K && a->ldy >= n_out && (!a->residual || a->ldr >= n_out));
  if (a->row_scale || a->row_shift || a->stats_rstd) {   // folded normalisation: the bf16 MFMA GEMM only
    if (a->dtype != AKI_DT_BF16) return AKI_ERR_UNSUPPORTED;
    return linear_bf16(a, (hipStream_t)stream);
  }
  if (a->preact_out) {   // training forward of the gated MLP: the bf16 MFMA GEMM only (whatever M is)
    if (a->dtype != AKI_DT_BF16 || a->act != AKI_ACT_SWIGLU) return AKI_ERR_UNSUPPORTED;
    return linear_bf16(a, (hipStream_t)stream);
  }
  if (a->w2) {   // two-segment weight: the bf16 MFMA GEMM only
    AKI_CHECK_ARG(a->w2_row0 > 0 && a->w2_rows > 0 && a->w2_row0 < a->N);
    if (a->dtype != AKI_DT_BF16 || a->act != AKI_ACT_NONE) return AKI_ERR_UNSUPPORTED;
    return linear_bf16(a, (hipStream_t)stream);
  }
  if (a->dtype == AKI_DT_FP8_E4M3) return linear_fp8(a, (hipStream_t)stream);
  if (a->dtype == AKI_DT_W8A16) {
    if (a->M >= 2 && a->M <= 16) {   // batched decode in the fp8 configuration: the skinny MFMA GEMM on e4m3 weights
      const int rc = skinny_gemm_w8(a, nullptr, 0.f, (hipStream_t)stream);
      if (rc != AKI_ERR_UNSUPPORTED) return rc;
    }
    return gemv_bf16(a, nullptr, 0.f, (hipStream_t)stream);
  }
  if (a->dtype == AKI_DT_BF16) {
    if (a->M >= 2 && a->M <= 16) {  // batched decode: weight-streaming skinny MFMA GEMM
      const int rc = skinny_gemm_bf16(a, nullptr, 0.f, (hipStream_t)stream);
      if (rc != AKI_ERR_UNSUPPORTED) return rc;
    }
    if (a->M <= 8) {  // decode regime: weight-streaming GEMV (falls through when the shape does not qualify)
      const int rc = gemv_bf16(a, nullptr, 0.f, (hipStream_t)stream);
      if (rc != AKI_ERR_UNSUPPORTED) return rc;
    }
    return linear_bf16(a, (hipStream_t)stream);
  }
  return linear_f32(a, (hipStream_t)stream);
}

// ---- fp8 quantisation --------------------------------------------------------------------------------------
int aki_quant_rows_fp8(const void* x, const void* rms_weight, float rms_eps, void* q, float* scale, int32_t rows, int32_t cols,
                       int32_t ldx, int32_t ldq, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && q && scale && rows > 0 && cols > 0 && ldx >= cols && ldq >= cols && (!rms_weight || rms_eps > 0.f));
  return quant_rows_fp8_launch(x, rms_weight, rms_eps, q, scale, rows, cols, ldx, ldq, (hipStream_t)stream);
}

// ---- norms -------------------------------------------------------------------------------------------
int aki_rmsnorm_fwd(const void* x, const void* w, void* y, int32_t rows, int32_t cols, int32_t ldx, int32_t ldy, float eps,
                    int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && w && y && dtype_ok(dtype) && ldx >= cols && ldy >= cols);
  return norm_launch(true, x, w, nullptr, y, rows, cols, ldx, ldy, eps, dtype, (hipStream_t)stream);
}

int aki_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int32_t rows, int32_t cols, int32_t ldx,
                      int32_t ldy, float eps, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && w && y && dtype_ok(dtype) && ldx >= cols && ldy >= cols);
  return norm_launch(false, x, w, b, y, rows, cols, ldx, ldy, eps, dtype, (hipStream_t)stream);
}

// ---- patch embed -------------------------------------------------------------------------------------
size_t aki_patch_embed_workspace_bytes(int32_t N, int32_t S, int32_t P, int32_t dtype) {
  const int G = S / P;
  const size_t Kp = aki_align_up((size_t)3 * P * P, 64);
  return aki_align_up((size_t)N * G * G * Kp * aki_elt_size(dtype), 256);
}

int aki_patch_embed_fwd(const void* pixels, const void* w, const void* bias, const void* pos, void* out, int32_t N, int32_t S,
                        int32_t P, int32_t E, int32_t Kp, int32_t dtype, void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(pixels && w && out && N > 0 && S > 0 && P > 0 && E > 0 && dtype_ok(dtype));
  AKI_CHECK_ARG(S >= P && Kp == (int)aki_align_up((size_t)3 * P * P, 64));
  if (!ws || ws_bytes < aki_patch_embed_workspace_bytes(N, S, P, dtype)) return AKI_ERR_WORKSPACE;
  const int G = S / P;
  int rc = im2col_launch(pixels, ws, N, S, P, Kp, dtype, (hipStream_t)stream);
  if (rc) return rc;
  aki_linear_args g = {};
  g.x = ws; g.w = w; g.bias = bias; g.residual = pos; g.y = out;
  g.M = N * G * G; g.N = E; g.K = Kp; g.ldx = Kp; g.ldw = Kp; g.ldy = E; g.ldr = E;
  g.res_row_mod = pos ? G * G : 0; g.act = AKI_ACT_NONE; g.dtype = dtype;
  return aki_linear_fwd(&g, stream);
}

// ---- connector ---------------------------------------------------------------------------------------
size_t aki_connector_mlp_workspace_bytes(int32_t rows, int32_t d, int32_t d_inner, int32_t dtype) {
  return aki_align_up((size_t)rows * d * aki_elt_size(dtype), 256) + aki_align_up((size_t)rows * d_inner * aki_elt_size(dtype), 256);
}

int aki_connector_mlp_fwd(const void* x, const void* ln_w, const void* ln_b, const void* w1, const void* w2, void* out,
                          int32_t rows, int32_t d, int32_t d_inner, float eps, int32_t dtype, void* ws, size_t ws_bytes,
                          void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && ln_w && w1 && w2 && out && rows > 0 && d > 0 && d_inner > 0 && dtype_ok(dtype));
  if (!ws || ws_bytes < aki_connector_mlp_workspace_bytes(rows, d, d_inner, dtype)) return AKI_ERR_WORKSPACE;
  char* normed = (char*)ws;
  char* hidden = normed + aki_align_up((size_t)rows * d * aki_elt_size(dtype), 256);
  int rc = aki_layernorm_fwd(x, ln_w, ln_b, normed, rows, d, d, d, eps, dtype, stream);
  if (rc) return rc;
  aki_linear_args g = {};
  g.x = normed; g.w = w1; g.y = hidden; g.M = rows; g.N = d_inner; g.K = d; g.ldx = d; g.ldw = d; g.ldy = d_inner;
  g.act = AKI_ACT_GELU_ERF; g.dtype = dtype;
  rc = aki_linear_fwd(&g, stream);
  if (rc) return rc;
  aki_linear_args g2 = {};
  g2.x = hidden; g2.w = w2; g2.y = out; g2.residual = x; g2.M = rows; g2.N = d; g2.K = d_inner; g2.ldx = d_inner; g2.ldw = d_inner;
  g2.ldy = d; g2.ldr = d; g2.act = AKI_ACT_NONE; g2.dtype = dtype;
  return aki_linear_fwd(&g2, stream);
}

int aki_connector_proj_fwd(const void* x, const void* ln_w, const void* ln_b, const void* w, const void* b, void* out,
                           int32_t rows, int32_t d, int32_t d_out, float eps, int32_t dtype, void* ws, size_t ws_bytes,
                           void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && ln_w && w && out && rows > 0 && d > 0 && d_out > 0 && dtype_ok(dtype));
  if (!ws || ws_bytes < aki_align_up((size_t)rows * d * aki_elt_size(dtype), 256)) return AKI_ERR_WORKSPACE;
  int rc = aki_layernorm_fwd(x, ln_w, ln_b, ws, rows, d, d, d, eps, dtype, stream);
  if (rc) return rc;
  aki_linear_args g = {};
  g.x = ws; g.w = w; g.bias = b; g.y = out; g.M = rows; g.N = d_out; g.K = d; g.ldx = d; g.ldw = d; g.ldy = d_out;
  g.act = AKI_ACT_NONE; g.dtype = dtype;
  return aki_linear_fwd(&g, stream);
}

// ---- decode ------------------------------------------------------------------------------------------
int aki_rope_append_fwd(const void* qkv, const float* cos, const float* sin, const int32_t* pos, const int32_t* cache_len, void* q_out,
                        void* k_cache, void* v_cache, int32_t B, int32_t H, int32_t Dh, int32_t capacity, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(qkv && cos && sin && pos && cache_len && q_out && k_cache && v_cache);
  AKI_CHECK_ARG(B > 0 && H > 0 && Dh > 0 && (Dh % 2) == 0 && capacity > 0 && dtype_ok(dtype));
  return rope_append_launch(qkv, cos, sin, pos, cache_len, q_out, k_cache, v_cache, B, H, Dh, capacity, dtype, (hipStream_t)stream);
}

size_t aki_decode_attn_workspace_bytes(int32_t B, int32_t H, int32_t Dh, int32_t capacity) {
  if (B <= 0 || H <= 0 || Dh <= 0 || capacity <= 0) return 0;
  return decode_attn_ws_bytes(B, H, Dh, capacity);
}

int aki_decode_attn_fwd(const void* q, const void* k_cache, const void* v_cache, void* o, const int32_t* n_keys,
                        const uint64_t* col_valid_bits, int32_t nwords, int32_t B, int32_t H, int32_t Dh, int32_t capacity,
                        int32_t max_keys, float scale, int32_t dtype, void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(q && k_cache && v_cache && o && n_keys && B > 0 && H > 0 && Dh > 0 && capacity > 0 && scale > 0.f && dtype_ok(dtype));
  AKI_CHECK_ARG(!col_valid_bits || nwords > 0);
  if (dtype == AKI_DT_BF16 && Dh == 96)
    return decode_attn_split_launch(q, nullptr, nullptr, n_keys, (void*)k_cache, (void*)v_cache, o, col_valid_bits, nwords, B, H,
                                    capacity, max_keys, scale, false, ws, ws_bytes, (hipStream_t)stream);
  return decode_attn_launch(q, k_cache, v_cache, o, n_keys, col_valid_bits, nwords, B, H, Dh, capacity, scale, dtype, (hipStream_t)stream);
}

int aki_decode_attn_fused_fwd(const void* qkv, const float* cos, const float* sin, const int32_t* cache_len, void* k_cache,
                              void* v_cache, void* o, const uint64_t* col_valid_bits, int32_t nwords, int32_t B, int32_t H,
                              int32_t Dh, int32_t capacity, int32_t max_keys, float scale, int32_t dtype, void* ws, size_t ws_bytes,
                              void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(qkv && cos && sin && cache_len && k_cache && v_cache && o);
  AKI_CHECK_ARG(B > 0 && H > 0 && Dh > 0 && (Dh % 2) == 0 && capacity > 0 && scale > 0.f && dtype_ok(dtype));
  AKI_CHECK_ARG(!col_valid_bits || nwords > 0);
  if (dtype == AKI_DT_BF16 && Dh == 96)
    return decode_attn_split_launch(qkv, cos, sin, cache_len, k_cache, v_cache, o, col_valid_bits, nwords, B, H, capacity, max_keys,
                                    scale, true, ws, ws_bytes, (hipStream_t)stream);
  return AKI_ERR_UNSUPPORTED;   // f32 / other head sizes: call aki_rope_append_fwd + aki_decode_attn_fwd
}

int aki_decode_linear_fwd(const aki_linear_args* a, const void* rms_weight, float rms_eps, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->x && a->w && a->y && rms_weight && rms_eps > 0.f);
  AKI_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && (a->dtype == AKI_DT_BF16 || (a->dtype == AKI_DT_W8A16 && a->w_scale)));
  AKI_CHECK_ARG(a->act >= AKI_ACT_NONE && a->act <= AKI_ACT_SWIGLU);
  const int n_out = a->act == AKI_ACT_SWIGLU ? a->N / 2 : a->N;
  AKI_CHECK_ARG(a->ldx >= a->K && a->ldw >= a->K && a->ldy >= n_out && (!a->residual || a->ldr >= n_out));
  if (a->M >= 2 && a->M <= 8) {      // batched decode: the skinny MFMA GEMM with the norm in its prologue
    const int rc = a->dtype == AKI_DT_BF16 ? skinny_gemm_bf16(a, rms_weight, rms_eps, (hipStream_t)stream)
                                           : skinny_gemm_w8(a, rms_weight, rms_eps, (hipStream_t)stream);
    if (rc != AKI_ERR_UNSUPPORTED) return rc;
  }
  return gemv_bf16(a, rms_weight, rms_eps, (hipStream_t)stream);
}

size_t aki_decode_chain_workspace_bytes(int32_t n_layers, int32_t d, int32_t H, int32_t F, int32_t capacity) {
  if (n_layers <= 0 || d <= 0 || H <= 0 || F <= 0 || capacity <= 0) return 0;
  return decode_chain_ws_bytes(n_layers, d, H, F, capacity);
}

size_t aki_decode_chain_error_offset(int32_t n_layers, int32_t H) { return (n_layers > 0 && H > 0) ? decode_chain_err_offset(n_layers, H) : 0; }

size_t aki_decode_chain_batch_workspace_bytes(int32_t n_layers, int32_t d, int32_t H, int32_t F, int32_t capacity, int32_t batch) {
  if (batch <= 1) return aki_decode_chain_workspace_bytes(n_layers, d, H, F, capacity);
  if (n_layers <= 0 || d <= 0 || H <= 0 || F <= 0 || capacity <= 0 || batch > 8) return 0;
  return decode_chain_b_ws_bytes(n_layers, d, H, F, capacity, batch);
}

size_t aki_decode_chain_batch_error_offset(int32_t n_layers, int32_t H, int32_t batch) {
  if (batch <= 1) return aki_decode_chain_error_offset(n_layers, H);
  return (n_layers > 0 && H > 0 && batch <= 8) ? decode_chain_b_err_offset(n_layers, H, batch) : 0;
}

int aki_decode_chain_fwd(const aki_decode_chain_args* a, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->layers && a->h_in && a->h_out && a->cos && a->sin && a->cache_len && a->workspace);
  AKI_CHECK_ARG(a->n_layers > 0 && a->d > 0 && a->H > 0 && a->Dh > 0 && a->F > 0 && a->capacity > 0 && a->scale > 0.f && a->rms_eps > 0.f);
  AKI_CHECK_ARG(!a->col_valid_bits || a->nwords > 0);
  return decode_chain_launch(a, (hipStream_t)stream);
}

// ---- training step ---------------------------------------------------------------------------------------
#define AKI_BF16_ONLY(dt) do { if ((dt) != AKI_DT_BF16) return AKI_ERR_UNSUPPORTED; } while (0)

size_t aki_attn_bwd_workspace_bytes(int32_t B, int32_t H, int32_t Lq) { return (B > 0 && H > 0 && Lq > 0) ? attn_bwd_ws_bytes(B, H, Lq) : 0; }

int aki_attn_bwd(const aki_attn_bwd_args* a, void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->q && a->k && a->v && a->o && a->d_o && a->lse && a->dq && a->dk && a->dv);
  AKI_CHECK_ARG(a->B > 0 && a->H > 0 && a->Lq > 0 && a->Lk > 0 && a->scale > 0.f);
  AKI_CHECK_ARG(a->max_rects >= 0 && a->max_rects <= AKI_MAX_RECTS);
  AKI_BF16_ONLY(a->dtype);
  return attn_bwd_bf16(a->q, a->k, a->v, a->o, a->d_o, a->lse, a->dq, a->dk, a->dv, a->rects, a->max_rects, a->col_valid_bits,
                       a->seq_lens, a->masked, a->B, a->H, a->Lq, a->Lk, a->Dh, a->scale, ws, ws_bytes, (hipStream_t)stream);
}

int aki_transpose(const void* x, void* y, int32_t R, int32_t C, int32_t ldx, int32_t ldy, int32_t Rpad, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && y && R > 0 && C > 0 && ldx >= C && Rpad >= R && ldy >= Rpad);
  AKI_BF16_ONLY(dtype);
  return transpose_bf16_launch(x, y, R, C, ldx, ldy, Rpad, (hipStream_t)stream);
}

int aki_gemm_tn(const void* a, const void* b, void* c, int32_t Kc, int32_t I, int32_t J, int64_t lda, int64_t ldb, int64_t ldc, int32_t dtype,
                void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && b && c && Kc > 0 && I > 0 && J > 0 && lda >= I && ldb >= J && ldc >= J);
  AKI_BF16_ONLY(dtype);
  return gemm_tn_launch(a, b, c, Kc, I, J, (long)lda, (long)ldb, (long)ldc, (hipStream_t)stream);
}

size_t aki_norm_bwd_workspace_bytes(int32_t cols) { return cols > 0 ? norm_bwd_ws_bytes(cols) : 0; }

int aki_norm_bwd(int32_t rms, const void* x, const void* w, const void* dy, const void* dres, void* dx, void* dw, void* db, int32_t rows,
                 int32_t cols, int32_t ldx, int32_t lddy, int32_t lddr, int32_t lddx, float eps, int32_t accumulate, int32_t dtype,
                 void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && w && dy && dx && dw && rows > 0 && cols > 0 && ldx >= cols && lddy >= cols && lddx >= cols);
  AKI_CHECK_ARG((ldx % 8) == 0 && (lddy % 8) == 0 && (lddx % 8) == 0 && (!dres || (lddr >= cols && (lddr % 8) == 0)));
  AKI_BF16_ONLY(dtype);
  return norm_bwd_launch(rms != 0, x, w, dy, dres, dx, dw, db, rows, cols, ldx, lddy, lddx, lddr, eps, accumulate, ws, ws_bytes,
                         (hipStream_t)stream);
}

size_t aki_colsum_workspace_bytes(int32_t cols) { return cols > 0 ? colsum_ws_bytes(cols) : 0; }

int aki_colsum(const void* x, void* out, int32_t rows, int32_t cols, int32_t ldx, int32_t accumulate, int32_t dtype, void* ws,
               size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && out && rows > 0 && cols > 0 && ldx >= cols);
  AKI_BF16_ONLY(dtype);
  return colsum_launch(x, out, rows, cols, ldx, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

int aki_swiglu_fwd(const void* gate_up, void* a, int32_t rows, int32_t F, int32_t ldg, int32_t lda, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(gate_up && a && rows > 0 && F > 0 && ldg >= 2 * F && lda >= F);
  AKI_BF16_ONLY(dtype);
  return swiglu_launch(0, gate_up, nullptr, a, rows, F, ldg, 0, lda, (hipStream_t)stream);
}

int aki_swiglu_bwd(const void* gate_up, const void* da, void* dgate_up, int32_t rows, int32_t F, int32_t ldg, int32_t ldda, int32_t lddg,
                   int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(gate_up && da && dgate_up && rows > 0 && F > 0 && ldg >= 2 * F && ldda >= F && lddg >= 2 * F);
  AKI_BF16_ONLY(dtype);
  return swiglu_launch(1, gate_up, da, dgate_up, rows, F, ldg, ldda, lddg, (hipStream_t)stream);
}

int aki_gelu_fwd(const void* x, void* y, size_t n, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && y && n > 0);
  AKI_BF16_ONLY(dtype);
  return gelu_launch(0, x, nullptr, y, n, (hipStream_t)stream);
}

int aki_gelu_bwd(const void* x, const void* dy, void* dx, size_t n, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(x && dy && dx && n > 0);
  AKI_BF16_ONLY(dtype);
  return gelu_launch(1, x, dy, dx, n, (hipStream_t)stream);
}

int aki_rope_bwd_merge(const void* dq, const void* dk, const void* dv, const float* cos, const float* sin, const int32_t* position_ids,
                       void* dqkv, int32_t B, int32_t H, int32_t L, int32_t Dh, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(dq && dk && dv && cos && sin && dqkv && B > 0 && H > 0 && L > 0 && Dh > 0);
  AKI_BF16_ONLY(dtype);
  return rope_bwd_merge_launch(dq, dk, dv, cos, sin, position_ids, dqkv, B, H, L, Dh, (hipStream_t)stream);
}

int aki_ce_loss_fwd_bwd(const void* logits, const int64_t* labels, int32_t* n_valid, float* loss_rows, void* dlogits, int32_t B, int32_t L,
                        int32_t V, int32_t ldl, int32_t lddl, float gscale, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(logits && labels && n_valid && loss_rows && B > 0 && L > 0 && V > 0 && ldl >= V && (!dlogits || lddl >= V));
  AKI_BF16_ONLY(dtype);
  return ce_launch(logits, labels, n_valid, loss_rows, dlogits, B, L, V, ldl, lddl, gscale, (hipStream_t)stream);
}

int aki_ce_rows_fwd_bwd(const void* logits, const int64_t* targets, const int32_t* n_valid, float* loss_rows, void* dlogits, int32_t rows,
                        int32_t V, int32_t ldl, int32_t lddl, float gscale, int32_t dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(logits && targets && n_valid && loss_rows && rows > 0 && V > 0 && ldl >= V && (!dlogits || lddl >= V));
  AKI_BF16_ONLY(dtype);
  return ce_rows_launch(logits, targets, n_valid, loss_rows, dlogits, rows, V, ldl, lddl, gscale, (hipStream_t)stream);
}

size_t aki_grad_sqnorm_workspace_bytes(void) { return grad_sqnorm_ws_bytes(); }

int aki_grad_sqnorm(const void* g, size_t n, float* out, int32_t accumulate, int32_t dtype, void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(g && out && n > 0);
  if (dtype != AKI_DT_BF16 && dtype != AKI_DT_F32) return AKI_ERR_UNSUPPORTED;
  return grad_sqnorm_launch(g, n, out, accumulate, ws, ws_bytes, dtype == AKI_DT_F32, (hipStream_t)stream);
}

int aki_adamw_step(float* p, float* m, float* v, const void* g, void* w16, size_t n, const float* sqnorm, float max_norm, float gscale,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(p && m && v && g && w16 && n > 0 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f);
  return adamw_launch(p, m, v, g, w16, n, sqnorm, max_norm, gscale, lr, beta1, beta2, eps, weight_decay, step, false, (hipStream_t)stream);
}

int aki_adamw_step_t(float* p, float* m, float* v, const void* g, void* w16, void* wT, int32_t N, int32_t K, int32_t ldT, const float* sqnorm,
                     float max_norm, float gscale, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                     int32_t grad_dtype, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(p && m && v && g && w16 && wT && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f);
  if (grad_dtype != AKI_DT_BF16 && grad_dtype != AKI_DT_F32) return AKI_ERR_UNSUPPORTED;
  return adamw_t_launch(p, m, v, g, w16, wT, N, K, ldT, sqnorm, max_norm, gscale, lr, beta1, beta2, eps, weight_decay, step, grad_dtype == AKI_DT_F32,
                        (hipStream_t)stream);
}

int aki_adamw_step_g32(float* p, float* m, float* v, const float* g, void* w16, size_t n, const float* sqnorm, float max_norm, float gscale,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(p && m && v && g && w16 && n > 0 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f);
  return adamw_launch(p, m, v, g, w16, n, sqnorm, max_norm, gscale, lr, beta1, beta2, eps, weight_decay, step, true, (hipStream_t)stream);
}

// ---- splice / mask -------------------------------------------------------------------------------------
int aki_splice_plan(const int64_t* lang_x, int32_t B, int32_t T, int64_t media_token_id, int64_t assistant_token_id,
                    int32_t Nv, int32_t* plan, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(lang_x && plan && B > 0 && T > 0 && Nv > 0);
  return splice_plan_launch(lang_x, B, T, media_token_id, assistant_token_id, Nv, plan, (hipStream_t)stream);
}

int aki_splice_fwd(const aki_splice_args* a, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->lang_x && a->embed_weight && a->plan && a->inputs_embeds);
  AKI_CHECK_ARG(a->B > 0 && a->T > 0 && a->Nv > 0 && a->d > 0 && a->L_out > 0 && dtype_ok(a->dtype));
  AKI_CHECK_ARG(a->T_img == 0 || a->vision_tokens);
  AKI_CHECK_ARG(a->max_rects >= 0 && a->max_rects <= AKI_MAX_RECTS);
  AKI_CHECK_ARG(a->padding_side == 0 || a->padding_side == 1);
  if (((size_t)a->d * aki_elt_size(a->dtype)) % 16) return AKI_ERR_ALIGNMENT;
  AKI_CHECK_ALIGN16(a->embed_weight); AKI_CHECK_ALIGN16(a->embed_additional); AKI_CHECK_ALIGN16(a->vision_tokens);
  AKI_CHECK_ALIGN16(a->inputs_embeds);
  return splice_launch(a, (hipStream_t)stream);
}

int aki_mma_mask_dense(const aki_mma_rect* rects, int32_t max_rects, const uint64_t* col_valid_bits, const int32_t* seq_lens,
                       int32_t B, int32_t L, int64_t* out, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(out && B > 0 && L > 0 && max_rects >= 0 && max_rects <= AKI_MAX_RECTS && (max_rects == 0 || rects));
  return mask_dense_launch(rects, max_rects, col_valid_bits, seq_lens, B, L, out, (hipStream_t)stream);
}

int aki_sft_collate_pad(const int64_t* ids, const int64_t* labels, const int64_t* attention_mask, const int32_t* offsets, int32_t B,
                        int32_t T_out, int64_t pad_token_id, int64_t ignore_index, int32_t padding_side, int64_t* out_ids,
                        int64_t* out_labels, int64_t* out_mask, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(ids && offsets && out_ids && B > 0 && T_out > 0 && (padding_side == 0 || padding_side == 1));
  AKI_CHECK_ARG((!out_labels || labels) && (!out_mask || attention_mask));
  return sft_collate_launch(ids, labels, attention_mask, offsets, B, T_out, pad_token_id, ignore_index, padding_side, out_ids, out_labels,
                            out_mask, (hipStream_t)stream);
}

int aki_greedy_pick(const void* logits, int32_t B, int32_t V, int64_t ld, const int64_t* eos_ids, int32_t n_eos, int64_t pad_token_id,
                    uint8_t* done, int64_t* next_ids, int64_t* tokens, int32_t tokens_ld, int32_t* cache_len, const int32_t* start_len,
                    int32_t advance, int32_t* done_at, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(logits && next_ids && B > 0 && V > 0 && ld >= V && ld < (1ll << 31) && n_eos >= 0 && (n_eos == 0 || eos_ids));
  AKI_CHECK_ARG((!tokens || tokens_ld > 0) && (!advance || cache_len) && (advance == 0 || advance == 1));
  return greedy_pick_launch(logits, B, V, (int)ld, eos_ids, n_eos, pad_token_id, done, next_ids, tokens, tokens_ld, cache_len, start_len, advance,
                            done_at, nullptr, nullptr, 0, 0, nullptr, (hipStream_t)stream);
}

int aki_greedy_pick_embed(const void* logits, int32_t B, int32_t V, int64_t ld, const int64_t* eos_ids, int32_t n_eos, int64_t pad_token_id,
                          uint8_t* done, int64_t* next_ids, int64_t* tokens, int32_t tokens_ld, int32_t* cache_len, const int32_t* start_len,
                          int32_t advance, int32_t* done_at, const void* embed_weight, const void* additional_weight, int64_t max_original_id,
                          int64_t num_additional, int32_t d, void* next_embeds, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(logits && next_ids && B > 0 && V > 0 && ld >= V && ld < (1ll << 31) && n_eos >= 0 && (n_eos == 0 || eos_ids));
  AKI_CHECK_ARG((!tokens || tokens_ld > 0) && (!advance || cache_len) && (advance == 0 || advance == 1));
  AKI_CHECK_ARG(embed_weight && next_embeds && d > 0 && d % 8 == 0 && max_original_id >= 0 && num_additional >= 0);
  AKI_CHECK_ARG((((uintptr_t)embed_weight | (uintptr_t)additional_weight | (uintptr_t)next_embeds) & 15) == 0);
  // every id the pick can produce has a row: V' = original rows + additional rows (DecoupledLinear's output width), pad included
  AKI_CHECK_ARG((int64_t)V <= max_original_id + 1 + (additional_weight ? num_additional : 0) && pad_token_id >= 0 &&
                pad_token_id <= max_original_id + (additional_weight ? num_additional : 0));
  return greedy_pick_launch(logits, B, V, (int)ld, eos_ids, n_eos, pad_token_id, done, next_ids, tokens, tokens_ld, cache_len, start_len, advance,
                            done_at, embed_weight, additional_weight, max_original_id, d, next_embeds, (hipStream_t)stream);
}

size_t aki_mma_mask_to_table_workspace_bytes(int32_t B, int32_t L) {
  return (B > 0 && L > 0) ? aki_align_up(mask_to_table_ws_bytes(B, L), 256) : 0;
}

int aki_mma_mask_to_table(const int64_t* mask, int32_t B, int32_t L, int32_t max_rects, aki_mma_rect* rects, uint64_t* col_valid_bits,
                          int32_t* seq_lens, int32_t* status, void* ws, size_t ws_bytes, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(mask && rects && col_valid_bits && seq_lens && status && B > 0 && L > 0);
  AKI_CHECK_ARG(max_rects >= 1 && max_rects <= AKI_MAX_RECTS);
  if (!ws || ws_bytes < aki_mma_mask_to_table_workspace_bytes(B, L)) return AKI_ERR_WORKSPACE;
  AKI_CHECK_ALIGN16(ws);
  return mask_to_table_launch(mask, B, L, max_rects, rects, col_valid_bits, seq_lens, status, ws, (hipStream_t)stream);
}

}  // extern "C"
