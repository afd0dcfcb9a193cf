// aux_kernels.hip - HBM-bound row / byte / integer kernels of the AKI forward path:
//   RMSNorm, LayerNorm, language-stream splice (+ DecoupledEmbedding gather, mask table),
//   dense mask materialisation (API compatibility / bit-exact parity), im2col for the patch embedding.
#include "aki_device.h"

namespace aki {

// ------------------------------------------------------------------------------------------------
// Norms.  One 256-thread block per row, 16-byte vector loads, row kept in registers (cols <= 8192).
// rmsnorm semantics (HF:phi3/modeling_phi3.py:266-284): y = w * cast_to_input_dtype(x * rsqrt(mean(x^2)+eps)).
// ------------------------------------------------------------------------------------------------
template <bool RMS>
__global__ __launch_bounds__(256) void norm_bf16_kernel(const bf16_t* x, const bf16_t* w, const bf16_t* bias, bf16_t* y,
                                                        int cols, int ldx, int ldy, float eps) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const bf16_t* xr = x + (size_t)row * ldx;
  bf16_t* yr = y + (size_t)row * ldy;
  constexpr int MAXC = 4;
  u32x4 buf[MAXC];
  float s1 = 0.f, s2 = 0.f;
  const int nchunk = cols / 8;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nchunk) {
      buf[i] = *(const u32x4*)(xr + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = bf16_lo(buf[i][e]), b = bf16_hi(buf[i][e]);
        s1 += a + b;
        s2 += a * a + b * b;
      }
    }
  }
  block_sum2<256>(s1, s2, red);
  const float mean = RMS ? 0.f : s1 / cols;
  float var;
  if (RMS) {
    var = s2 / cols;
  } else {
    // two-pass variance for accuracy (matches torch's LayerNorm to f32 rounding)
    float d2 = 0.f, dummy = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = threadIdx.x + i * 256;
      if (c < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = bf16_lo(buf[i][e]) - mean, b = bf16_hi(buf[i][e]) - mean;
          d2 += a * a + b * b;
        }
      }
    }
    block_sum2<256>(d2, dummy, red);
    var = d2 / cols;
  }
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nchunk) {
      const u32x4 wv = *(const u32x4*)(w + c * 8);
      u32x4 bv = {0, 0, 0, 0};
      if (!RMS && bias) bv = *(const u32x4*)(bias + c * 8);
      u32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = (bf16_lo(buf[i][e]) - mean) * rstd, b = (bf16_hi(buf[i][e]) - mean) * rstd;
        if (RMS) {
          a = round_bf16(a) * bf16_lo(wv[e]);
          b = round_bf16(b) * bf16_hi(wv[e]);
        } else {
          a = a * bf16_lo(wv[e]) + bf16_lo(bv[e]);
          b = b * bf16_hi(wv[e]) + bf16_hi(bv[e]);
        }
        ov[e] = pack_bf16x2(a, b);
      }
      *(u32x4*)(yr + c * 8) = ov;
    }
  }
}

template <bool RMS>
__global__ __launch_bounds__(256) void norm_f32_kernel(const float* x, const float* w, const float* bias, float* y, int cols,
                                                       int ldx, int ldy, float eps) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const float* xr = x + (size_t)row * ldx;
  float* yr = y + (size_t)row * ldy;
  float s1 = 0.f, s2 = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) { const float a = xr[c]; s1 += a; s2 += a * a; }
  block_sum2<256>(s1, s2, red);
  const float mean = RMS ? 0.f : s1 / cols;
  float var = s2 / cols;
  if (!RMS) {
    float d2 = 0.f, dummy = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) { const float a = xr[c] - mean; d2 += a * a; }
    block_sum2<256>(d2, dummy, red);
    var = d2 / cols;
  }
  const float rstd = rsqrtf(var + eps);
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float n = (xr[c] - mean) * rstd;
    yr[c] = RMS ? w[c] * n : n * w[c] + (bias ? bias[c] : 0.f);
  }
}

// Row statistics on their own (input of the first block when the normalisation is folded into the GEMMs): one block per row,
// rstd = 1/sqrt(mean(x^2) + eps), or with `mean`: mean and 1/sqrt(var + eps) (two-pass variance, as the LayerNorm kernel).
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* x, int cols, int ldx, float eps, float* rstd, float* mean) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const bf16_t* xr = x + (size_t)row * ldx;
  float s1 = 0.f, s2 = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
    const u32x4 v = *(const u32x4*)(xr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float a = bf16_lo(v[e]), b = bf16_hi(v[e]); s1 += a + b; s2 += a * a + b * b; }
  }
  block_sum2<256>(s1, s2, red);
  if (mean == nullptr) {
    if (threadIdx.x == 0) rstd[row] = rsqrtf(s2 / cols + eps);
    return;
  }
  const float mu = s1 / cols;
  float d2 = 0.f, dummy = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
    const u32x4 v = *(const u32x4*)(xr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float a = bf16_lo(v[e]) - mu, b = bf16_hi(v[e]) - mu; d2 += a * a + b * b; }
  }
  block_sum2<256>(d2, dummy, red);
  if (threadIdx.x == 0) { mean[row] = mu; rstd[row] = rsqrtf(d2 / cols + eps); }
}

int row_stats_launch(const void* x, int rows, int cols, int ldx, float eps, float* rstd, float* mean, hipStream_t s) {
  if (cols % 8 || ldx % 8) return AKI_ERR_UNSUPPORTED;
  AKI_CHECK_ALIGN16(x);
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(row_stats_kernel, dim3(rows), dim3(256), 0, s, (const bf16_t*)x, cols, ldx, eps, rstd, mean);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

int norm_launch(bool rms, const void* x, const void* w, const void* b, void* y, int rows, int cols, int ldx, int ldy,
                float eps, int dtype, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return AKI_ERR_INVALID_ARG;
  if (dtype == AKI_DT_BF16) {
    if (cols % 8 || cols > 8192 || ldx % 8 || ldy % 8) return AKI_ERR_UNSUPPORTED;
    AKI_CHECK_ALIGN16(x); AKI_CHECK_ALIGN16(w); AKI_CHECK_ALIGN16(y); AKI_CHECK_ALIGN16(b);
    if (rms) hipLaunchKernelGGL(norm_bf16_kernel<true>, dim3(rows), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, cols, ldx, ldy, eps);
    else hipLaunchKernelGGL(norm_bf16_kernel<false>, dim3(rows), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, cols, ldx, ldy, eps);
  } else {
    if (rms) hipLaunchKernelGGL(norm_f32_kernel<true>, dim3(rows), dim3(256), 0, stream, (const float*)x, (const float*)w, (const float*)b, (float*)y, cols, ldx, ldy, eps);
    else hipLaunchKernelGGL(norm_f32_kernel<false>, dim3(rows), dim3(256), 0, stream, (const float*)x, (const float*)w, (const float*)b, (float*)y, cols, ldx, ldy, eps);
  }
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// ------------------------------------------------------------------------------------------------
// Splice plan: one wave per sample scans lang_x (src/vlm.py:488-496).
// plan[b] (AKI_PLAN_STRIDE int32) = {n_img, q_idx (first id == assistant, else 0), L_b, 0, t_0 .. t_7}
// where t_k is the index of the k-th <image> placeholder in the original prompt.
// ------------------------------------------------------------------------------------------------
__global__ void splice_plan_kernel(const int64_t* lang_x, int T, int64_t media_id, int64_t assistant_id, int Nv, int* plan) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t* row = lang_x + (size_t)b * T;
  int* pl = plan + b * AKI_PLAN_STRIDE;
  int n_img = 0, q = 0x7fffffff;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    const int64_t id = t < T ? row[t] : (int64_t)-1;
    unsigned long long mm = __ballot(t < T && id == media_id);
    if (t < T && id == assistant_id) q = min(q, t);
    while (mm) {  // wave-uniform loop over the placeholders of this chunk, in order
      const int bit = __builtin_ctzll(mm);
      mm &= mm - 1;
      if (lane == 0 && n_img < AKI_MAX_RECTS) pl[4 + n_img] = t0 + bit;
      n_img++;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q = min(q, __shfl_xor(q, o));
  if (lane == 0) {
    pl[0] = n_img;
    pl[1] = q == 0x7fffffff ? 0 : q;
    pl[2] = T - n_img + Nv * n_img;
    pl[3] = 0;
    for (int i = n_img; i < AKI_MAX_RECTS; ++i) pl[4 + i] = -1;
  }
}

// Source of position l of the un-padded spliced sequence.  Expanded index rule (src/vlm.py:534-546):
// the k-th <image> at original index t_k starts at E_k = t_k + k*(Nv-1) and occupies Nv slots.
__device__ __forceinline__ void splice_locate(const int* pl, int Nv, int l, int& src_t, int& img, int& slot) {
  const int n_img = min(pl[0], AKI_MAX_RECTS);
  img = -1; slot = 0; src_t = l;
  for (int k = 0; k < n_img; ++k) {
    const int start = pl[4 + k] + k * (Nv - 1);
    if (l < start) return;
    if (l < start + Nv) { img = k; slot = l - start; src_t = pl[4 + k]; return; }
    src_t = l - (k + 1) * (Nv - 1);
  }
}

// ------------------------------------------------------------------------------------------------
// Splice: one block per output row (b, l_out): DecoupledEmbedding gather (src/helpers.py:445-484),
// vision-token splice (src/vlm.py:539-577), scalar pad_token_id / -100 padding (src/vlm.py:584-598).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void splice_kernel(const aki_splice_args a) {
  const int b = blockIdx.y, l_out = blockIdx.x;
  const int64_t* ids = a.lang_x + (size_t)b * a.T;
  const int* pl = a.plan + b * AKI_PLAN_STRIDE;
  const int Lb = pl[2];
  const int shift = a.padding_side == 1 ? a.L_out - Lb : 0;  // left padding shifts embeds/labels only
  const int l = l_out - shift;
  T* dst = (T*)a.inputs_embeds + ((size_t)b * a.L_out + l_out) * a.d;
  int src_t = -1, img = -1, slot = 0;
  if (l >= 0 && l < Lb) splice_locate(pl, a.Nv, l, src_t, img, slot);
  const int nvec = a.d * (int)sizeof(T) / 16;  // d*sizeof(T) is a multiple of 16 (checked on the host)
  if (src_t < 0) {  // padding row: scalar pad_token_id in every channel
    for (int c = threadIdx.x; c < a.d; c += 256) dst[c] = (T)(float)a.pad_token_id;
  } else {
    const T* src;
    if (img >= 0) {
      src = (const T*)a.vision_tokens + (((size_t)b * a.T_img + img) * a.Nv + slot) * a.d;
    } else {
      const int64_t id = ids[src_t];
      src = (id > a.max_original_id && a.embed_additional)
                ? (const T*)a.embed_additional + (size_t)(id - a.max_original_id - 1) * a.d
                : (const T*)a.embed_weight + (size_t)id * a.d;
    }
    for (int c = threadIdx.x; c < nvec; c += 256) ((u32x4*)dst)[c] = ((const u32x4*)src)[c];
  }
  if (threadIdx.x == 0) {
    if (a.labels_out) {
      int64_t lab = -100;
      if (src_t >= 0 && img < 0 && a.labels) lab = a.labels[(size_t)b * a.T + src_t];
      a.labels_out[(size_t)b * a.L_out + l_out] = lab;
    }
    if (a.mask_1d_out) {  // mask index space is never shifted (src/utils.py:99-108 pads bottom/right)
      int64_t mv = 0;
      if (l_out < Lb) {
        int st, im, sl;
        splice_locate(pl, a.Nv, l_out, st, im, sl);
        mv = im >= 0 ? 1 : (a.attention_mask ? a.attention_mask[(size_t)b * a.T + st] : 1);
      }
      a.mask_1d_out[(size_t)b * a.L_out + l_out] = mv;
    }
  }
}

// Mask table: rects + valid bits + seq_lens.  One wave per sample.
__global__ void splice_table_kernel(const aki_splice_args a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int* pl = a.plan + b * AKI_PLAN_STRIDE;
  const int n_img = min(pl[0], AKI_MAX_RECTS), q = pl[1], Lb = pl[2];
  const int nwords = (a.L_out + 63) / 64;
  if (lane == 0) {
    a.seq_lens[b] = Lb;
    // src/vlm.py:556-564: rows [E_k, E_k+Nv), cols [E_k+Nv, text_end) under python-slice clamping to Lb,
    // text_end = expanded index of <|assistant|> + 1 (= q + Nv for the reference's single image in front of it).
    int n_before = 0;
    for (int k = 0; k < n_img; ++k) n_before += pl[4 + k] < q ? 1 : 0;
    const int text_end = q + n_before * (a.Nv - 1) + 1;
    for (int k = 0; k < a.max_rects; ++k) {
      aki_mma_rect r = {0, 0, 0, 0};
      if (k < n_img) {
        const int start = pl[4 + k] + k * (a.Nv - 1);
        r.row_lo = min(start, Lb); r.row_hi = min(start + a.Nv, Lb);
        r.col_lo = min(start + a.Nv, Lb); r.col_hi = min(text_end, Lb);
        if (r.row_hi <= r.row_lo || r.col_hi <= r.col_lo) r = aki_mma_rect{0, 0, 0, 0};
      }
      a.rects[(size_t)b * a.max_rects + k] = r;
    }
  }
  for (int w = 0; w < nwords; ++w) {
    const int lm = w * 64 + lane;
    bool valid = false;
    if (lm < Lb) {
      int st, im, sl;
      splice_locate(pl, a.Nv, lm, st, im, sl);
      valid = im >= 0 ? true : (a.attention_mask ? a.attention_mask[(size_t)b * a.T + st] != 0 : true);
    }
    const unsigned long long bits = __ballot(valid);
    if (lane == 0) a.col_valid_bits[(size_t)b * nwords + w] = bits;
  }
}

// ------------------------------------------------------------------------------------------------
// Dense (B,1,L,L) int64 0/1 mask from the table - bit-exact with src/vlm.py:410-443 + src/utils.py:99-108.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_dense_kernel(const aki_mma_rect* rects, int max_rects, const uint64_t* vbits,
                                                         const int* seq_lens, int L, int64_t* out) {
  const int b = blockIdx.y, r = blockIdx.x;
  const int Lb = seq_lens ? min(seq_lens[b], L) : L;
  int rc0 = 0, rc1 = 0;
  for (int i = 0; i < max_rects; ++i) {
    const aki_mma_rect q = rects[(size_t)b * max_rects + i];
    if (r >= q.row_lo && r < q.row_hi && q.col_hi > q.col_lo) { rc0 = q.col_lo; rc1 = q.col_hi; }
  }
  const int nwords = (L + 63) / 64;
  int64_t* orow = out + ((size_t)b * L + r) * L;
  for (int c = threadIdx.x; c < L; c += 256) {
    bool vis = (r < Lb) && ((c <= r) || (c >= rc0 && c < rc1));
    const bool cv = vbits ? ((vbits[(size_t)b * nwords + (c >> 6)] >> (c & 63)) & 1ull) != 0ull : (c < L);
    orow[c] = (vis && cv) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// Dense (B,1,L,L) int64 0/1 mask -> table: the hand-off type of the reference (`attention_mask` of
// src/vlm.py:589-603 as passed to lang_model at src/aki.py:125-130) converted back to rectangles + valid bits +
// seq_lens, so a caller that still builds the dense tensor lands on the O(L) attention path.
//   pass 1 (one block per mask row): first / one-past-last set column to the RIGHT of the diagonal, "row has a set
//           column at all", and the column-wise OR of the whole sample as valid bits (a column nobody may see is
//           indistinguishable from an invalid one);
//   pass 2 (one block per sample): consecutive rows with the same right-of-diagonal interval become one rectangle
//           (rows whose interval starts right at the diagonal, lo == r+1, are the part of a rectangle the causal
//           triangle already covers and join the run above them); seq_len = last non-empty row + 1.
// The candidate table is NOT trusted: the caller materialises it again (mask_dense_kernel) and compares it with the
// input bit for bit, so anything outside the family {causal + row-interval rectangles + invalid columns} is refused.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_rows_scan_kernel(const int64_t* mask, int L, int4* rowinfo, unsigned long long* vbits) {
  __shared__ int s_lo[4], s_hi[4], s_any[4];
  const int b = blockIdx.y, r = blockIdx.x;
  const int64_t* row = mask + ((size_t)b * L + r) * L;
  const int nwords = (L + 63) / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int lo = 0x7fffffff, hi = 0, any = 0;
  for (int c0 = wave * 64; c0 < L; c0 += 256) {
    const int c = c0 + lane;
    const bool set = c < L && row[c] != 0;
    const unsigned long long word = __ballot(set);
    if (word) {
      any = 1;
      if (lane == 0) atomicOr(vbits + (size_t)b * nwords + (c0 >> 6), word);
    }
    if (set && c > r) { lo = min(lo, c); hi = max(hi, c + 1); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
  if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_any[wave] = any; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int plo[3], phi[3], pany[3];                            // the other waves' partials into registers, ONE full wait, then the fold
#pragma unroll
    for (int w = 1; w < 4; ++w) { plo[w - 1] = s_lo[w]; phi[w - 1] = s_hi[w]; pany[w - 1] = s_any[w]; }
    lds_reads_landed();
#pragma unroll
    for (int w = 0; w < 3; ++w) asm volatile("" : "+v"(plo[w]), "+v"(phi[w]), "+v"(pany[w]));
#pragma unroll
    for (int w = 0; w < 3; ++w) { lo = min(lo, plo[w]); hi = max(hi, phi[w]); any |= pany[w]; }
    if (hi == 0) lo = 0;
    rowinfo[(size_t)b * L + r] = make_int4(lo, hi, any, 0);
  }
}

__global__ __launch_bounds__(256) void mask_table_build_kernel(const int4* rowinfo, int L, int max_rects, aki_mma_rect* rects,
                                                               int* seq_lens, int* status) {
  constexpr int CH = 2048;
  __shared__ int4 s_info[CH];
  const int b = blockIdx.x;
  int n = 0, run_lo = 0, run_hi = 0, run_start = 0, seq = 0;   // only thread 0's copies are meaningful
  bool open = false;
  for (int r0 = 0; r0 < L; r0 += CH) {
    const int nr = min(CH, L - r0);
    __syncthreads();
    for (int i = threadIdx.x; i < nr; i += 256) s_info[i] = rowinfo[(size_t)b * L + r0 + i];
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 0; i < nr; ++i) {
        const int r = r0 + i;
        const int4 q = s_info[i];
        if (q.z) seq = r + 1;
        const bool has = q.y > q.x;
        const bool joins = open && has && q.y == run_hi && (q.x == run_lo || (q.x == r + 1 && run_lo <= r));
        if (open && !joins) {
          if (n < max_rects) rects[(size_t)b * max_rects + n] = aki_mma_rect{run_start, r, run_lo, run_hi};
          ++n;
          open = false;
        }
        if (has && !open) { open = true; run_start = r; run_lo = q.x; run_hi = q.y; }
      }
    }
  }
  if (threadIdx.x == 0) {
    if (open) {
      if (n < max_rects) rects[(size_t)b * max_rects + n] = aki_mma_rect{run_start, L, run_lo, run_hi};
      ++n;
    }
    for (int i = n; i < max_rects; ++i) rects[(size_t)b * max_rects + i] = aki_mma_rect{0, 0, 0, 0};
    seq_lens[b] = seq;
    status[b] = n > max_rects ? n : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// SFT collate (train/sft_data_utils/loader_utils.py:11-91 `_pad_trunc` / `batch_collate_pad`): B ragged token sequences,
// concatenated (`offsets[b] .. offsets[b+1]`), become [B, T_out] arrays - truncated to their first T_out tokens, or padded
// on the right / left with pad_token_id (ids), ignore_index (labels) and 0 (attention mask).  One block per sample.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sft_collate_kernel(const int64_t* ids, const int64_t* labels, const int64_t* mask,
                                                          const int* offsets, int T_out, int64_t pad_id, int64_t ignore_index,
                                                          int left, int64_t* out_ids, int64_t* out_labels, int64_t* out_mask) {
  const int b = blockIdx.x;
  const int o0 = offsets[b], len = offsets[b + 1] - o0;
  const int shift = (left && len < T_out) ? T_out - len : 0;     // left padding moves the sequence to the end of the row
  for (int t = threadIdx.x; t < T_out; t += 256) {
    const int s = t - shift;
    const bool tok = s >= 0 && s < len;                           // len >= T_out: the first T_out tokens survive
    const size_t o = (size_t)b * T_out + t;
    out_ids[o] = tok ? ids[o0 + s] : pad_id;
    if (out_labels) out_labels[o] = tok ? labels[o0 + s] : ignore_index;
    if (out_mask) out_mask[o] = tok ? mask[o0 + s] : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// im2col for the SigLIP patch embedding: A[n*G*G + gy*G + gx][c*P*P + py*P + px], zero padded to Kp.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col_kernel(const T* pix, T* out, int N, int S, int P, int G, int Kp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * G * G * Kp;
  if (idx >= total) return;
  const int k = idx % Kp;
  const size_t rowi = idx / Kp;
  const int gx = rowi % G, gy = (rowi / G) % G, n = rowi / ((size_t)G * G);
  T v = (T)0.f;
  if (k < 3 * P * P) {
    const int c = k / (P * P), py = (k / P) % P, px = k % P;
    v = pix[(((size_t)n * 3 + c) * S + gy * P + py) * S + gx * P + px];
  }
  out[idx] = v;
}

int splice_plan_launch(const int64_t* lang_x, int B, int T, int64_t media, int64_t assistant, int Nv, int* plan, hipStream_t s) {
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(splice_plan_kernel, dim3(B), dim3(64), 0, s, lang_x, T, media, assistant, Nv, plan);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

int splice_launch(const aki_splice_args* a, hipStream_t s) {
  dim3 grid(a->L_out, a->B);
  if (a->dtype == AKI_DT_BF16) hipLaunchKernelGGL(splice_kernel<__bf16>, grid, dim3(256), 0, s, *a);
  else hipLaunchKernelGGL(splice_kernel<float>, grid, dim3(256), 0, s, *a);
  AKI_LAUNCH_CHECK();
  if (a->rects && a->col_valid_bits && a->seq_lens) {
    AKI_CLEAR_ERR();
    hipLaunchKernelGGL(splice_table_kernel, dim3(a->B), dim3(64), 0, s, *a);
    AKI_LAUNCH_CHECK();
  }
  return AKI_OK;
}

int mask_dense_launch(const aki_mma_rect* rects, int max_rects, const uint64_t* vbits, const int* seq_lens, int B, int L,
                      int64_t* out, hipStream_t s) {
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(mask_dense_kernel, dim3(L, B), dim3(256), 0, s, rects, rects ? max_rects : 0, vbits, seq_lens, L, out);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

size_t mask_to_table_ws_bytes(int B, int L) { return (size_t)B * L * sizeof(int4); }

int mask_to_table_launch(const int64_t* mask, int B, int L, int max_rects, aki_mma_rect* rects, uint64_t* vbits, int* seq_lens,
                         int* status, void* ws, hipStream_t s) {
  AKI_CLEAR_ERR();
  const int nwords = (L + 63) / 64;
  if (hipMemsetAsync(vbits, 0, (size_t)B * nwords * sizeof(uint64_t), s) != hipSuccess) return AKI_ERR_LAUNCH;
  hipLaunchKernelGGL(mask_rows_scan_kernel, dim3(L, B), dim3(256), 0, s, mask, L, (int4*)ws, (unsigned long long*)vbits);
  AKI_LAUNCH_CHECK();
  hipLaunchKernelGGL(mask_table_build_kernel, dim3(B), dim3(256), 0, s, (const int4*)ws, L, max_rects, rects, seq_lens, status);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

int sft_collate_launch(const int64_t* ids, const int64_t* labels, const int64_t* mask, const int* offsets, int B, int T_out,
                       int64_t pad_id, int64_t ignore_index, int left, int64_t* out_ids, int64_t* out_labels, int64_t* out_mask,
                       hipStream_t s) {
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(sft_collate_kernel, dim3(B), dim3(256), 0, s, ids, labels, mask, offsets, T_out, pad_id, ignore_index, left,
                     out_ids, out_labels, out_mask);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

int im2col_launch(const void* pix, void* out, int N, int S, int P, int Kp, int dtype, hipStream_t s) {
  const int G = S / P;
  const size_t total = (size_t)N * G * G * Kp;
  if (dtype == AKI_DT_BF16) hipLaunchKernelGGL(im2col_kernel<__bf16>, dim3((total + 255) / 256), dim3(256), 0, s, (const __bf16*)pix, (__bf16*)out, N, S, P, G, Kp);
  else hipLaunchKernelGGL(im2col_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, (const float*)pix, (float*)out, N, S, P, G, Kp);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
