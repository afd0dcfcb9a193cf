// stack.hip - the layer loops of an inference forward issued from ONE C call each (host code only).
//
// Why.  A one-sample prefill (the shape every caller of the reference runs: local_demo.py:75-87, eval_cv_bench/eval.py:92-104) is ~400
// launches of 10-70 us; issued from Python through ctypes + torch.empty they cost ~23 us of host time apiece - 9 ms of a 15 ms first
// token were the HOST (profiles/r05_generate_bench_before.json).  The loops over the 32 decoder layers (HF:phi3/modeling_phi3.py
// Phi3Model.forward, :287-328 per layer) and the 27 SigLIP encoder layers (HF:siglip/modeling_siglip.py:329-354) are plain
// sequences of this library's own launches on caller-owned buffers: here they run in C++, ~3 us per launch.
//
// Both functions are the bf16 inference form with the normalisations folded into the GEMMs (aki_linear_args: row_scale /
// row_shift / stats_*), exactly the launches aki_amd/phi3.py::forward_folded and aki_amd/siglip.py::forward_folded issue one by
// one - same kernels, same arguments, bit-identical results (tests/test_stack_gpu.py).
#include <hip/hip_runtime.h>

#include "aki_device.h"

namespace {

inline char* carve(char*& p, size_t bytes) {
  char* r = p;
  p += aki_align_up(bytes, 256);
  return r;
}

}  // namespace

extern "C" {

size_t aki_decoder_stack_workspace_bytes(int32_t B, int32_t H, int32_t L, int32_t Dh, int32_t d, int32_t F, int32_t keep_kv) {
  if (B <= 0 || H <= 0 || L <= 0 || Dh <= 0 || d <= 0 || F <= 0) return 0;
  const size_t M = (size_t)B * L, head = aki_align_up(M * H * Dh * 2, 256);
  return head * (keep_kv ? 1 : 3)                                  // q (+ k, v when no KV cache takes them)
         + aki_align_up(M * (size_t)H * Dh * 2, 256)               // attention output
         + 2 * aki_align_up(M * (size_t)d * 2, 256)                // residual stream, two buffers
         + aki_align_up(M * (size_t)F * 2, 256)                    // SwiGLU output
         + 3 * aki_align_up(M * 4, 256)                            // 1/rms of the three streams in flight
         + aki_mma_attn_core_workspace_bytes(B, H, L, Dh, AKI_DT_BF16);
}

int aki_decoder_stack_fwd(const aki_decoder_stack_args* a, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->layers && a->n_layers > 0 && a->h_in && a->h_out && a->cos && a->sin && a->workspace && a->stats_workspace);
  AKI_CHECK_ARG(a->B > 0 && a->H > 0 && a->L > 0 && a->Dh > 0 && a->d > 0 && a->F > 0 && a->rms_eps > 0.f && a->scale > 0.f);
  AKI_CHECK_ARG(a->kv_capacity == 0 || a->kv_capacity >= a->L);
  const int M = a->B * a->L, HD = a->H * a->Dh;
  bool keep = a->layers[0].k_cache != nullptr;
  for (int i = 0; i < a->n_layers; ++i) {
    const aki_decoder_layer& ly = a->layers[i];
    AKI_CHECK_ARG(ly.w_qkv && ly.w_o && ly.w_gate_up && ly.w_down && (ly.k_cache != nullptr) == keep && (ly.v_cache != nullptr) == keep);
  }
  if (a->workspace_bytes < aki_decoder_stack_workspace_bytes(a->B, a->H, a->L, a->Dh, a->d, a->F, keep ? 1 : 0) || ((uintptr_t)a->workspace & 255))
    return AKI_ERR_WORKSPACE;
  char* w = (char*)a->workspace;
  void* q = carve(w, (size_t)M * HD * 2);
  void* k_tmp = keep ? nullptr : carve(w, (size_t)M * HD * 2);
  void* v_tmp = keep ? nullptr : carve(w, (size_t)M * HD * 2);
  void* o = carve(w, (size_t)M * HD * 2);
  void* hbuf[2] = {carve(w, (size_t)M * a->d * 2), carve(w, (size_t)M * a->d * 2)};
  void* act = carve(w, (size_t)M * a->F * 2);
  float* st[3] = {(float*)carve(w, (size_t)M * 4), (float*)carve(w, (size_t)M * 4), (float*)carve(w, (size_t)M * 4)};
  void* core_ws = w;
  const size_t core_ws_bytes = aki_mma_attn_core_workspace_bytes(a->B, a->H, a->L, a->Dh, AKI_DT_BF16);

  // statistics of the embeddings: the one pass over the residual stream that is not a GEMM
  int rc = aki_row_stats(a->h_in, M, a->d, a->d, a->rms_eps, st[0], nullptr, AKI_DT_BF16, stream);
  if (rc) return rc;
  const void* h = a->h_in;
  float* st_h = st[0];
  for (int i = 0; i < a->n_layers; ++i) {
    const aki_decoder_layer& ly = a->layers[i];
    const bool last = i + 1 == a->n_layers;
    // (1) q, k, v = RoPE(rstd * (h W_qkv'^T)): keys / values straight into the KV cache when there is one
    aki_mma_attn_args qa = {};
    qa.x = h; qa.w_qkv = ly.w_qkv; qa.cos = a->cos; qa.sin = a->sin; qa.position_ids = a->position_ids;
    qa.B = a->B; qa.H = a->H; qa.L = a->L; qa.Dh = a->Dh; qa.d_model = a->d; qa.ldx = a->d; qa.ldw = a->d; qa.pos_rows = a->pos_rows;
    qa.scale = a->scale; qa.dtype = AKI_DT_BF16; qa.dead_rows = a->dead_rows; qa.kv_capacity = keep ? a->kv_capacity : 0;
    qa.row_scale = st_h;
    void* kk = keep ? ly.k_cache : k_tmp;
    void* vv = keep ? ly.v_cache : v_tmp;
    rc = aki_qkv_rope_fwd(&qa, q, kk, vv, nullptr, 0, stream);
    if (rc) return rc;
    // (2) span-driven attention
    aki_mma_attn_core_args ca = {};
    ca.q = q; ca.k = kk; ca.v = vv; ca.o = o; ca.rects = a->rects; ca.col_valid_bits = a->col_valid_bits; ca.seq_lens = a->seq_lens;
    ca.max_rects = a->max_rects; ca.B = a->B; ca.H = a->H; ca.L = a->L; ca.Dh = a->Dh; ca.scale = a->scale; ca.dtype = AKI_DT_BF16;
    ca.dead_rows = a->dead_rows; ca.kv_capacity = keep ? a->kv_capacity : 0;
    rc = aki_mma_attn_core_fwd(&ca, core_ws, core_ws_bytes, stream);
    if (rc) return rc;
    // (3) h1 = h + o W_o^T, leaving 1/rms(h1)
    aki_linear_args la = {};
    la.x = o; la.w = ly.w_o; la.residual = h; la.y = hbuf[0]; la.M = M; la.N = a->d; la.K = HD; la.ldx = HD; la.ldw = HD; la.ldy = a->d; la.ldr = a->d;
    la.act = AKI_ACT_NONE; la.dtype = AKI_DT_BF16; la.stats_rstd = st[1]; la.stats_eps = a->rms_eps;
    la.stats_workspace = a->stats_workspace; la.stats_workspace_bytes = a->stats_workspace_bytes;
    la.splitk_workspace = a->splitk_workspace; la.splitk_workspace_bytes = a->splitk_workspace_bytes;
    rc = aki_linear_fwd(&la, stream);
    if (rc) return rc;
    // (4) act = up * silu(gate) of rstd1 * (h1 W_gate_up'^T)
    aki_linear_args ga = {};
    ga.x = hbuf[0]; ga.w = ly.w_gate_up; ga.y = act; ga.M = M; ga.N = 2 * a->F; ga.K = a->d; ga.ldx = a->d; ga.ldw = a->d; ga.ldy = a->F;
    ga.act = AKI_ACT_SWIGLU; ga.dtype = AKI_DT_BF16; ga.row_scale = st[1];
    rc = aki_linear_fwd(&ga, stream);
    if (rc) return rc;
    // (5) h2 = h1 + act W_down^T, leaving 1/rms(h2) for the next layer (or the head)
    void* h2 = last ? a->h_out : hbuf[1];
    float* st_next = last ? (a->rstd_out ? a->rstd_out : st[2]) : (st_h == st[0] ? st[2] : st[0]);
    aki_linear_args da = {};
    da.x = act; da.w = ly.w_down; da.residual = hbuf[0]; da.y = h2; da.M = M; da.N = a->d; da.K = a->F; da.ldx = a->F; da.ldw = a->F; da.ldy = a->d; da.ldr = a->d;
    da.act = AKI_ACT_NONE; da.dtype = AKI_DT_BF16; da.stats_rstd = st_next; da.stats_eps = a->rms_eps;
    da.stats_workspace = a->stats_workspace; da.stats_workspace_bytes = a->stats_workspace_bytes;
    da.splitk_workspace = a->splitk_workspace; da.splitk_workspace_bytes = a->splitk_workspace_bytes;
    rc = aki_linear_fwd(&da, stream);
    if (rc) return rc;
    h = h2;
    st_h = st_next;
  }
  return AKI_OK;
}

size_t aki_siglip_stack_workspace_bytes(int32_t N, int32_t L, int32_t E, int32_t heads) {
  if (N <= 0 || L <= 0 || E <= 0 || heads <= 0) return 0;
  const size_t M = (size_t)N * L;
  return aki_align_up(M * 3 * E * 2, 256) + aki_align_up(M * (size_t)E * 2, 256) + 2 * aki_align_up(M * (size_t)E * 2, 256) + 6 * aki_align_up(M * 4, 256) +
         aki_align_up((size_t)N * heads * (E / heads) * 4 + 256, 256);
}

int aki_siglip_stack_fwd(const aki_siglip_stack_args* a, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->layers && a->n_layers > 0 && a->h_in && a->h_out && a->fc1_out && a->workspace && a->stats_workspace);
  AKI_CHECK_ARG(a->N > 0 && a->L > 0 && a->E > 0 && a->heads > 0 && a->E % a->heads == 0 && a->I > 0 && a->Ip >= a->I && a->Ip % 64 == 0);
  AKI_CHECK_ARG(a->ln_eps > 0.f && a->scale > 0.f && (a->act == AKI_ACT_GELU_TANH || a->act == AKI_ACT_GELU_ERF));
  for (int i = 0; i < a->n_layers; ++i) {
    const aki_siglip_layer& ly = a->layers[i];
    AKI_CHECK_ARG(ly.w_qkv && ly.b_qkv && ly.c_qkv && ly.w_out && ly.w_fc1 && ly.b_fc1 && ly.c_fc1 && ly.w_fc2);
  }
  if (a->workspace_bytes < aki_siglip_stack_workspace_bytes(a->N, a->L, a->E, a->heads) || ((uintptr_t)a->workspace & 255)) return AKI_ERR_WORKSPACE;
  const int M = a->N * a->L, E = a->E, Dh = E / a->heads;
  char* w = (char*)a->workspace;
  char* qkv = carve(w, (size_t)M * 3 * E * 2);
  void* ao = carve(w, (size_t)M * E * 2);
  void* hbuf[2] = {carve(w, (size_t)M * E * 2), carve(w, (size_t)M * E * 2)};
  float* sr[3]; float* sm[3];
  for (int i = 0; i < 3; ++i) { sr[i] = (float*)carve(w, (size_t)M * 4); sm[i] = (float*)carve(w, (size_t)M * 4); }
  void* attn_ws = w;
  const size_t attn_ws_bytes = (size_t)a->N * a->heads * Dh * 4 + 256;
  const float scale = a->scale;

  int rc = aki_row_stats(a->h_in, M, E, E, a->ln_eps, sr[0], sm[0], AKI_DT_BF16, stream);      // the embeddings' statistics
  if (rc) return rc;
  const void* h = a->h_in;
  int s_in = 0;
  for (int i = 0; i < a->n_layers; ++i) {
    const aki_siglip_layer& ly = a->layers[i];
    const bool last = i + 1 == a->n_layers;
    // (1) qkv = LayerNorm1(h) W_qkv^T + b, the norm folded: rstd * (h W'^T - mean * c) + b'
    aki_linear_args qa = {};
    qa.x = h; qa.w = ly.w_qkv; qa.bias = ly.b_qkv; qa.y = qkv; qa.M = M; qa.N = 3 * E; qa.K = E; qa.ldx = E; qa.ldw = E; qa.ldy = 3 * E;
    qa.act = AKI_ACT_NONE; qa.dtype = AKI_DT_BF16; qa.row_scale = sr[s_in]; qa.row_shift = sm[s_in]; qa.col_shift = ly.c_qkv;
    rc = aki_linear_fwd(&qa, stream);
    if (rc) return rc;
    // (2) 16 x 72 attention, q / k / v read in place out of the fused projection
    aki_attn_args at = {};
    at.q = qkv; at.k = qkv + (size_t)E * 2; at.v = qkv + (size_t)2 * E * 2; at.o = ao;
    at.q_stride_b = at.k_stride_b = at.v_stride_b = (int64_t)a->L * 3 * E;
    at.q_stride_h = at.k_stride_h = at.v_stride_h = Dh;
    at.q_stride_t = at.k_stride_t = at.v_stride_t = 3 * E;
    at.B = a->N; at.H = a->heads; at.Lq = a->L; at.Lk = a->L; at.Dh = Dh; at.scale = scale; at.dtype = AKI_DT_BF16;
    rc = aki_attn_fwd(&at, attn_ws, attn_ws_bytes, stream);
    if (rc) return rc;
    // (3) h1 = h + out_proj(attention), leaving LayerNorm2's statistics
    const int s1 = (s_in + 1) % 3, s2 = (s_in + 2) % 3;
    aki_linear_args oa = {};
    oa.x = ao; oa.w = ly.w_out; oa.bias = ly.b_out; oa.residual = h; oa.y = hbuf[0]; oa.M = M; oa.N = E; oa.K = E; oa.ldx = E; oa.ldw = E; oa.ldy = E; oa.ldr = E;
    oa.act = AKI_ACT_NONE; oa.dtype = AKI_DT_BF16; oa.stats_rstd = sr[s1]; oa.stats_mean = sm[s1]; oa.stats_eps = a->ln_eps;
    oa.stats_workspace = a->stats_workspace; oa.stats_workspace_bytes = a->stats_workspace_bytes;
    oa.splitk_workspace = a->splitk_workspace; oa.splitk_workspace_bytes = a->splitk_workspace_bytes;
    rc = aki_linear_fwd(&oa, stream);
    if (rc) return rc;
    // (4) fc1 + GELU with LayerNorm2 folded, into the K-padded buffer (its pad columns stay the zeros the caller put there)
    aki_linear_args fa = {};
    fa.x = hbuf[0]; fa.w = ly.w_fc1; fa.bias = ly.b_fc1; fa.y = a->fc1_out; fa.M = M; fa.N = a->I; fa.K = E; fa.ldx = E; fa.ldw = E; fa.ldy = a->Ip;
    fa.act = a->act; fa.dtype = AKI_DT_BF16; fa.row_scale = sr[s1]; fa.row_shift = sm[s1]; fa.col_shift = ly.c_fc1;
    rc = aki_linear_fwd(&fa, stream);
    if (rc) return rc;
    // (5) h2 = h1 + fc2(...), leaving the next layer's LayerNorm1 statistics (not after the last layer)
    void* h2 = last ? a->h_out : hbuf[1];
    aki_linear_args ga = {};
    ga.x = a->fc1_out; ga.w = ly.w_fc2; ga.bias = ly.b_fc2; ga.residual = hbuf[0]; ga.y = h2; ga.M = M; ga.N = E; ga.K = a->Ip; ga.ldx = a->Ip; ga.ldw = a->Ip; ga.ldy = E; ga.ldr = E;
    ga.act = AKI_ACT_NONE; ga.dtype = AKI_DT_BF16;
    if (!last) {
      ga.stats_rstd = sr[s2]; ga.stats_mean = sm[s2]; ga.stats_eps = a->ln_eps;
      ga.stats_workspace = a->stats_workspace; ga.stats_workspace_bytes = a->stats_workspace_bytes;
    }
    ga.splitk_workspace = a->splitk_workspace; ga.splitk_workspace_bytes = a->splitk_workspace_bytes;
    rc = aki_linear_fwd(&ga, stream);
    if (rc) return rc;
    h = h2;
    s_in = s2;
  }
  return AKI_OK;
}

size_t aki_perceiver_stack_workspace_bytes(int32_t n1, int32_t n2, int32_t D, int32_t heads, int32_t dim_head, int32_t d_ff) {
  if (n1 <= 0 || n2 <= 0 || D <= 0 || heads <= 0 || dim_head <= 0 || d_ff <= 0) return 0;
  const size_t inner = (size_t)heads * dim_head;
  return aki_align_up((size_t)(n1 + n2) * D * 2, 256) + aki_align_up((size_t)n2 * inner * 2, 256) + aki_align_up((size_t)(n1 + n2) * 2 * inner * 2, 256) +
         aki_align_up((size_t)n2 * inner * 2, 256) + 2 * aki_align_up((size_t)n2 * D * 2, 256) + aki_connector_mlp_workspace_bytes(n2, D, d_ff, AKI_DT_BF16) +
         aki_align_up((size_t)heads * dim_head * 4 + 256, 256);
}

int aki_perceiver_stack_fwd(const aki_perceiver_stack_args* a, void* stream) {
  AKI_CLEAR_ERR();
  AKI_CHECK_ARG(a && a->layers && a->n_layers > 0 && a->x && a->latents && a->out && a->norm_w && a->workspace);
  AKI_CHECK_ARG(a->n1 > 0 && a->n2 > 0 && a->D > 0 && a->heads > 0 && a->dim_head > 0 && a->d_ff > 0 && a->eps > 0.f && a->scale > 0.f);
  AKI_CHECK_ARG(!a->proj_w || a->D_out > 0);
  for (int i = 0; i < a->n_layers; ++i) {
    const aki_perceiver_layer& ly = a->layers[i];
    AKI_CHECK_ARG(ly.norm_media_w && ly.norm_latents_w && ly.w_q && ly.w_kv && ly.w_out && ly.ff_ln_w && ly.ff_w1 && ly.ff_w2);
  }
  if (a->workspace_bytes < aki_perceiver_stack_workspace_bytes(a->n1, a->n2, a->D, a->heads, a->dim_head, a->d_ff) || ((uintptr_t)a->workspace & 255))
    return AKI_ERR_WORKSPACE;
  const int n1 = a->n1, n2 = a->n2, D = a->D, inner = a->heads * a->dim_head, nk = n1 + n2;
  char* w = (char*)a->workspace;
  char* cat = carve(w, (size_t)nk * D * 2);                 // [LN_media(x) ; LN_latents(latents)]: the keys / values' input, rows n1.. the queries' input
  void* q = carve(w, (size_t)n2 * inner * 2);
  char* kv = carve(w, (size_t)nk * 2 * inner * 2);
  void* ao = carve(w, (size_t)n2 * inner * 2);
  void* lat[2] = {carve(w, (size_t)n2 * D * 2), carve(w, (size_t)n2 * D * 2)};
  void* mlp_ws = carve(w, aki_connector_mlp_workspace_bytes(n2, D, a->d_ff, AKI_DT_BF16));
  void* attn_ws = w;
  const size_t mlp_ws_bytes = aki_connector_mlp_workspace_bytes(n2, D, a->d_ff, AKI_DT_BF16), attn_ws_bytes = (size_t)inner * 4 + 256;
  const void* cur = a->latents;
  int rc;
  for (int i = 0; i < a->n_layers; ++i) {
    const aki_perceiver_layer& ly = a->layers[i];
    // PerceiverAttention (src/helpers.py:76-102): both LayerNorms straight into the concatenated key / value input
    rc = aki_layernorm_fwd(a->x, ly.norm_media_w, ly.norm_media_b, cat, n1, D, D, D, a->eps, AKI_DT_BF16, stream);
    if (rc) return rc;
    rc = aki_layernorm_fwd(cur, ly.norm_latents_w, ly.norm_latents_b, cat + (size_t)n1 * D * 2, n2, D, D, D, a->eps, AKI_DT_BF16, stream);
    if (rc) return rc;
    aki_linear_args qa = {};
    qa.x = cat + (size_t)n1 * D * 2; qa.w = ly.w_q; qa.y = q; qa.M = n2; qa.N = inner; qa.K = D; qa.ldx = D; qa.ldw = D; qa.ldy = inner;
    qa.act = AKI_ACT_NONE; qa.dtype = AKI_DT_BF16;
    rc = aki_linear_fwd(&qa, stream);
    if (rc) return rc;
    aki_linear_args ka = {};
    ka.x = cat; ka.w = ly.w_kv; ka.y = kv; ka.M = nk; ka.N = 2 * inner; ka.K = D; ka.ldx = D; ka.ldw = D; ka.ldy = 2 * inner;
    ka.act = AKI_ACT_NONE; ka.dtype = AKI_DT_BF16;
    rc = aki_linear_fwd(&ka, stream);
    if (rc) return rc;
    aki_attn_args at = {};
    at.q = q; at.k = kv; at.v = kv + (size_t)inner * 2; at.o = ao;
    at.q_stride_b = (int64_t)n2 * inner; at.q_stride_h = a->dim_head; at.q_stride_t = inner;
    at.k_stride_b = at.v_stride_b = (int64_t)nk * 2 * inner; at.k_stride_h = at.v_stride_h = a->dim_head; at.k_stride_t = at.v_stride_t = 2 * inner;
    at.B = 1; at.H = a->heads; at.Lq = n2; at.Lk = nk; at.Dh = a->dim_head; at.scale = a->scale; at.dtype = AKI_DT_BF16;
    rc = aki_attn_fwd(&at, attn_ws, attn_ws_bytes, stream);
    if (rc) return rc;
    aki_linear_args oa = {};
    oa.x = ao; oa.w = ly.w_out; oa.residual = cur; oa.y = lat[0]; oa.M = n2; oa.N = D; oa.K = inner; oa.ldx = inner; oa.ldw = inner; oa.ldy = D; oa.ldr = D;
    oa.act = AKI_ACT_NONE; oa.dtype = AKI_DT_BF16;
    rc = aki_linear_fwd(&oa, stream);
    if (rc) return rc;
    // FeedForward + residual (src/helpers.py:32-39,194)
    rc = aki_connector_mlp_fwd(lat[0], ly.ff_ln_w, ly.ff_ln_b, ly.ff_w1, ly.ff_w2, lat[1], n2, D, a->d_ff, a->eps, AKI_DT_BF16, mlp_ws, mlp_ws_bytes, stream);
    if (rc) return rc;
    cur = lat[1];
  }
  if (a->proj_w) return aki_connector_proj_fwd(cur, a->norm_w, a->norm_b, a->proj_w, a->proj_b, a->out, n2, D, a->D_out, a->eps, AKI_DT_BF16, mlp_ws, mlp_ws_bytes, stream);
  return aki_layernorm_fwd(cur, a->norm_w, a->norm_b, a->out, n2, D, D, D, a->eps, AKI_DT_BF16, stream);
}

}  // extern "C"
