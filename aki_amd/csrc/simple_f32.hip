// simple_f32.hip - exact-f32 parity path (AKI_DT_F32): plain f32 FMA kernels, LDS tiled.
// These exist so that the 1e-5 fp32 parity bar of BASELINE.json can be checked against the oracle
// at full dimensions; the performance path is the bf16 MFMA code in gemm_bf16.hip / mma_attn_bf16.hip.
#include "aki_device.h"

namespace aki {

// ------------------------------------------------------------------------------------------------
// y = act(x W^T + bias) [+ residual], f32.  64x64 tile, BK = 16, 256 threads, 4x4 outputs / thread.
// ------------------------------------------------------------------------------------------------
struct GemmF32Params {
  const float* x; const float* w; const float* bias; const float* residual; float* y;
  int M, N, K, ldx, ldw, ldy, ldr, res_row_mod, act;
};

template <bool SWIGLU>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Params p) {
  __shared__ float sx[16][68];
  __shared__ float sw[SWIGLU ? 2 : 1][16][68];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;  // tx -> n, ty -> m
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int n_out = SWIGLU ? p.N / 2 : p.N;
  float acc[4][4] = {}, acc2[4][4] = {};
  const int lr = tid >> 2, lc = (tid & 3) * 4;  // loader: row 0..63, k 0,4,8,12
  for (int k0 = 0; k0 < p.K; k0 += 16) {
    {
      const int gm = min(m0 + lr, p.M - 1);
      const int gn = min(n0 + lr, n_out - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = k0 + lc + i;
        const bool ok = kk < p.K;
        sx[lc + i][lr] = ok ? p.x[(size_t)gm * p.ldx + kk] : 0.f;
        sw[0][lc + i][lr] = ok ? p.w[(size_t)gn * p.ldw + kk] : 0.f;
        if (SWIGLU) sw[SWIGLU ? 1 : 0][lc + i][lr] = ok ? p.w[(size_t)(n_out + gn) * p.ldw + kk] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4], b2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = sx[kk][ty * 4 + i]; b[i] = sw[0][kk][tx * 4 + i]; b2[i] = SWIGLU ? sw[SWIGLU ? 1 : 0][kk][tx * 4 + i] : 0.f; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_fmaf(a[i], b[j], acc[i][j]);
          if (SWIGLU) acc2[i][j] = __builtin_fmaf(a[i], b2[j], acc2[i][j]);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= n_out) continue;
      float v = acc[i][j];
      if (SWIGLU) {
        v = acc2[i][j] * (v / (1.0f + expf(-v)));
      } else {
        if (p.bias) v += p.bias[n];
        if (p.act == AKI_ACT_GELU_ERF) v = gelu_erf(v);
        else if (p.act == AKI_ACT_GELU_TANH) v = gelu_tanh(v);
      }
      if (p.residual) v += p.residual[(size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + n];
      p.y[(size_t)m * p.ldy + n] = v;
    }
  }
}

int linear_f32(const aki_linear_args* a, hipStream_t stream) {
  GemmF32Params p = {(const float*)a->x, (const float*)a->w, (const float*)a->bias, (const float*)a->residual, (float*)a->y,
                     a->M, a->N, a->K, a->ldx, a->ldw, a->ldy, a->ldr, a->res_row_mod, a->act};
  const int n_out = a->act == AKI_ACT_SWIGLU ? a->N / 2 : a->N;
  dim3 grid((n_out + 63) / 64, (a->M + 63) / 64);
  if (a->act == AKI_ACT_SWIGLU) {
    if (a->bias || (a->N & 1)) return AKI_ERR_UNSUPPORTED;
    AKI_CLEAR_ERR();
    hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, stream, p);
  } else {
    AKI_CLEAR_ERR();
    hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, stream, p);
  }
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// ------------------------------------------------------------------------------------------------
// RoPE + head split: qkv [M, 3*H*Dh] f32 -> q,k (rotated) and v in [B,H,L,Dh].
// HF:phi3/modeling_phi3.py:170-197,228-241.
// ------------------------------------------------------------------------------------------------
__global__ void rope_split_f32_kernel(const float* qkv, const float* cos, const float* sin, const int* position_ids,
                                      float* q, float* k, float* v, int M, int H, int L, int Dh, int cap) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)M * 3 * H * Dh;
  if (idx >= total) return;
  const int n = idx % (3 * H * Dh);
  const int m = idx / (3 * H * Dh);
  const int which = n / (H * Dh), hd = n % (H * Dh), head = hd / Dh, d = hd % Dh;
  const int b = m / L, t = m % L;
  const float x = qkv[idx];
  float out = x;
  if (which < 2) {
    const int pos = position_ids ? position_ids[m] : t;
    const int half = Dh / 2;
    const float partner = d < half ? -qkv[idx + half] : qkv[idx - half];
    out = x * cos[(size_t)pos * Dh + d] + partner * sin[(size_t)pos * Dh + d];
  }
  float* dst = which == 0 ? q : (which == 1 ? k : v);
  dst[((size_t)(b * H + head) * (which == 0 ? L : cap) + t) * Dh + d] = out;
}

// ------------------------------------------------------------------------------------------------
// Span-driven attention core, f32, any Dh that is a multiple of 4 and <= 128.
// Block = 64 query rows x 4 threads per row (each owns Dh/4 channels); KV tiles of 32 keys in LDS.
// ------------------------------------------------------------------------------------------------
struct AttnF32Params {
  const float* q; const float* k; const float* v; float* o; float* lse;
  const aki_mma_rect* rects; const uint64_t* vbits; const int* seq_lens; const float* vmean;
  int max_rects, B, H, L, Dh, nwords;
  float scale; int dead_uniform; int causal; int kvcap;
};

template <int DPT>
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnF32Params p) {
  constexpr int Dh = DPT * 4;
  __shared__ float sk[32][Dh + 4];
  __shared__ float sv[32][Dh + 4];
  const int tid = threadIdx.x;
  const int sub = tid & 3, rloc = tid >> 2;
  const int nqt = (p.L + 63) / 64;
  const int bh = blockIdx.x / nqt, qt = nqt - 1 - blockIdx.x % nqt;
  const int b = bh / p.H;
  const int L = p.L;
  const int row = qt * 64 + rloc;
  const int Lb = p.seq_lens ? min(p.seq_lens[b], L) : L;
  int rc0 = 0, rc1 = 0, hi_col = p.causal ? min(qt * 64 + 64, L) : L;
  for (int i = 0; i < p.max_rects; ++i) {
    const aki_mma_rect r = p.rects[(size_t)b * p.max_rects + i];
    if (r.row_hi > r.row_lo && r.col_hi > r.col_lo) {
      if (r.row_lo < qt * 64 + 64 && r.row_hi > qt * 64) hi_col = max(hi_col, min(r.col_hi, L));
      if (row >= r.row_lo && row < r.row_hi) { rc0 = r.col_lo; rc1 = r.col_hi; }
    }
  }
  const float* qb = p.q + (size_t)bh * L * Dh;
  const float* kb = p.k + (size_t)bh * p.kvcap * Dh;
  const float* vb = p.v + (size_t)bh * p.kvcap * Dh;
  float qr[DPT], acc[DPT];
#pragma unroll
  for (int i = 0; i < DPT; ++i) { qr[i] = qb[(size_t)min(row, L - 1) * Dh + sub * DPT + i]; acc[i] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const bool row_alive = row < Lb;
  for (int c0 = 0; c0 < hi_col; c0 += 32) {
    for (int e = tid; e < 32 * Dh; e += 256) {
      const int kr = e / Dh, kd = e % Dh;
      const size_t off = (size_t)min(c0 + kr, L - 1) * Dh + kd;
      sk[kr][kd] = kb[off];
      sv[kr][kd] = vb[off];
    }
    __syncthreads();
    const int cend = min(32, hi_col - c0);
    for (int cc = 0; cc < cend; ++cc) {
      const int c = c0 + cc;
      float part = 0.f;
      float kv[DPT];                                       // LDS reads into registers, ONE full wait, then the arithmetic (aki_device.h: lds_fold_ready)
#pragma unroll
      for (int i = 0; i < DPT; ++i) kv[i] = sk[cc][sub * DPT + i];
      lds_fold_ready(kv);
#pragma unroll
      for (int i = 0; i < DPT; ++i) part = __builtin_fmaf(qr[i], kv[i], part);
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      bool vis = (!p.causal) | (c <= row) | ((c >= rc0) & (c < rc1));
      const bool cv = p.vbits ? ((p.vbits[(size_t)b * p.nwords + (c >> 6)] >> (c & 63)) & 1ull) != 0ull : true;
      vis = vis & cv & row_alive;
      if (vis) {
        const float s = part * p.scale;
        const float m_new = fmaxf(m_run, s);
        const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first visible key
        const float pr = expf(s - m_new);
        l_run = l_run * alpha + pr;
        float vv[DPT];
#pragma unroll
        for (int i = 0; i < DPT; ++i) vv[i] = sv[cc][sub * DPT + i];
        lds_fold_ready(vv);
#pragma unroll
        for (int i = 0; i < DPT; ++i) acc[i] = __builtin_fmaf(pr, vv[i], acc[i] * alpha);
        m_run = m_new;
      }
    }
    __syncthreads();
  }
  if (row < L) {
    const bool dead = !(l_run > 0.f);
    const int head = bh % p.H;
    float* orow = p.o + ((size_t)(b * L + row) * p.H + head) * Dh + sub * DPT;
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      float v = dead ? 0.f : acc[i] / l_run;
      if (dead && p.dead_uniform) v = p.vmean[(size_t)bh * Dh + sub * DPT + i];
      orow[i] = v;
    }
    if (p.lse && sub == 0) p.lse[(size_t)bh * L + row] = dead ? -INFINITY : m_run + logf(l_run);
  }
}

__global__ void vmean_f32_kernel(const float* v, float* out, int L, int Dh, int cap) {
  const int bh = blockIdx.x, d = threadIdx.x;
  if (d >= Dh) return;
  const float* base = v + (size_t)bh * cap * Dh;
  float s = 0.f;
  for (int t = 0; t < L; ++t) s += base[(size_t)t * Dh + d];
  out[(size_t)bh * Dh + d] = s / (float)L;
}

int attn_core_f32(const aki_mma_attn_core_args* a, void* ws, size_t ws_bytes, hipStream_t stream, int causal) {
  if (a->Dh % 4 != 0 || a->Dh > 128) return AKI_ERR_UNSUPPORTED;
  if (a->max_rects < 0 || a->max_rects > AKI_MAX_RECTS) return AKI_ERR_INVALID_ARG;
  const size_t need = (size_t)a->B * a->H * a->Dh * sizeof(float);
  if (!ws || ws_bytes < need) return AKI_ERR_WORKSPACE;
  AttnF32Params p = {(const float*)a->q, (const float*)a->k, (const float*)a->v, (float*)a->o, a->lse,
                     a->rects, a->col_valid_bits, a->seq_lens, (const float*)ws,
                     a->rects ? a->max_rects : 0, a->B, a->H, a->L, a->Dh, (a->L + 63) / 64,
                     a->scale, a->dead_rows == AKI_DEAD_ROWS_UNIFORM, causal, a->kv_capacity > 0 ? a->kv_capacity : a->L};
  if (p.dead_uniform) {
    AKI_CLEAR_ERR();
    hipLaunchKernelGGL(vmean_f32_kernel, dim3(a->B * a->H), dim3(128), 0, stream, p.v, (float*)ws, a->L, a->Dh, p.kvcap);
    AKI_LAUNCH_CHECK();
  }
  const int nqt = (a->L + 63) / 64;
  dim3 grid(a->B * a->H * nqt);
  switch (a->Dh) {
    case 96: hipLaunchKernelGGL(attn_f32_kernel<24>, grid, dim3(256), 0, stream, p); break;
    case 64: hipLaunchKernelGGL(attn_f32_kernel<16>, grid, dim3(256), 0, stream, p); break;
    case 72: hipLaunchKernelGGL(attn_f32_kernel<18>, grid, dim3(256), 0, stream, p); break;
    case 128: hipLaunchKernelGGL(attn_f32_kernel<32>, grid, dim3(256), 0, stream, p); break;
    case 32: hipLaunchKernelGGL(attn_f32_kernel<8>, grid, dim3(256), 0, stream, p); break;
    default: return AKI_ERR_UNSUPPORTED;
  }
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

int qkv_rope_f32(const aki_mma_attn_args* a, void* q, void* k, void* v, float* tmp, hipStream_t stream) {
  aki_linear_args g = {};
  g.x = a->x; g.w = a->w_qkv; g.y = tmp; g.M = a->B * a->L; g.N = 3 * a->H * a->Dh; g.K = a->d_model;
  g.ldx = a->ldx; g.ldw = a->ldw; g.ldy = g.N; g.dtype = AKI_DT_F32;
  int rc = linear_f32(&g, stream);
  if (rc) return rc;
  const size_t total = (size_t)g.M * g.N;
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(rope_split_f32_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, tmp, a->cos, a->sin,
                     a->position_ids, (float*)q, (float*)k, (float*)v, g.M, a->H, a->L, a->Dh, a->kv_capacity > 0 ? a->kv_capacity : a->L);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
