/* mma_mask.c - plain-C restatement of the INTEGER part of the AKI hot path.  TEST INFRASTRUCTURE ONLY
 * (checker for tests/ and for bench.py's cpu_baseline leg; never linked into the product library).
 *
 * Restates, step by step:
 *   aki_oracle_mma_mask   : VLMWithLanguageStream._make_modality_mutual_mask, src/vlm.py:410-443
 *   aki_oracle_stack_masks: stack_with_padding_2D_attention,                  src/utils.py:99-108
 *   aki_oracle_splice_src : the index arithmetic of _prepare_inputs_for_forward, src/vlm.py:488-577
 * Pinned against the golden vectors produced by the reference itself (tests/test_oracle_golden.py).
 */
#include <stdint.h>
#include <string.h>

static int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* python slice(start, stop).indices(n) for step 1 */
static void py_slice(int64_t start, int64_t stop, int64_t n, int64_t* lo, int64_t* hi) {
  if (start < 0) start += n;
  if (stop < 0) stop += n;
  *lo = clampi(start, 0, n);
  *hi = clampi(stop, 0, n);
}

/* out: n*n int64, row-major (the reference returns (1,n,n)) */
void aki_oracle_mma_mask(const int64_t* attention_mask_1d, int64_t n, int64_t image_start, int64_t text_start,
                         int64_t text_end, int64_t* out) {
  int64_t r, c, r0, r1, c0, c1;
  for (r = 0; r < n; ++r)                      /* :424-426  mask_cond < (mask_cond + 1).view(n,1) -> tril */
    for (c = 0; c < n; ++c) out[r * n + c] = (c < r + 1) ? 1 : 0;
  py_slice(image_start, text_start, n, &r0, &r1);
  py_slice(text_start, text_end, n, &c0, &c1);
  for (r = r0; r < r1; ++r)                    /* :429  mask[image_start:text_start, text_start:text_end] = 1 */
    for (c = c0; c < c1; ++c) out[r * n + c] = 1;
  for (c = 0; c < n; ++c)                      /* :434-438  columns whose 1-D mask is 0 are zeroed for every row */
    if (attention_mask_1d[c] == 0)
      for (r = 0; r < n; ++r) out[r * n + c] = 0;
}

/* masks[b] is n_b*n_b; out is B*Lmax*Lmax zero padded bottom/right (src/utils.py:99-108) */
void aki_oracle_stack_masks(const int64_t* const* masks, const int64_t* ns, int64_t B, int64_t Lmax, int64_t* out) {
  int64_t b, r;
  memset(out, 0, (size_t)(B * Lmax * Lmax) * sizeof(int64_t));
  for (b = 0; b < B; ++b)
    for (r = 0; r < ns[b]; ++r) memcpy(out + (b * Lmax + r) * Lmax, masks[b] + r * ns[b], (size_t)ns[b] * sizeof(int64_t));
}

/* For one sample: source of every position of the spliced stream.
 * src_kind[l] = 0 text token (src_idx = index into lang_x), 1 vision token (src_idx = img*Nv + slot).
 * Returns the spliced length L = T - n_img + Nv*n_img.  Mirrors the torch.cat sequence of src/vlm.py:534-577
 * (each earlier image shifts later placeholders by Nv-1). */
int64_t aki_oracle_splice_src(const int64_t* lang_x, int64_t T, int64_t media_token_id, int64_t Nv, int32_t* src_kind,
                              int64_t* src_idx) {
  int64_t t, l = 0, img = 0, i;
  for (t = 0; t < T; ++t) {
    if (lang_x[t] == media_token_id) {
      for (i = 0; i < Nv; ++i) { src_kind[l] = 1; src_idx[l] = img * Nv + i; ++l; }
      ++img;
    } else {
      src_kind[l] = 0; src_idx[l] = t; ++l;
    }
  }
  return l;
}
