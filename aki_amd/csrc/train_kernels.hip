// train_kernels.hip - the HBM-bound kernels of the training step.
//
// The reference's step is autograd over its eager forward under bf16 autocast followed by clip_grad_norm_(1.0) and
// AdamW (train/train_utils.py:242-266, train/train.py:330-337).  The GEMM-shaped part of the backward reuses the MFMA
// GEMM of the forward (gemm_bf16.hip) on transposed operands; everything here is elementwise / row-reduction work
// bounded by HBM traffic, written as 16-byte-per-lane streaming kernels:
//   transpose_bf16        y[C][ldy] = x[R][C]^T, zero-padding the new K dimension (wgrad / dgrad operands)
//   norm_bwd              RMSNorm / LayerNorm backward: dx per row, per-workgroup partial dw (db) + finishing pass
//   swiglu_fwd / _bwd     a = up * silu(gate)    (HF:phi3/modeling_phi3.py:49-64)
//   gelu_fwd / _bwd       erf GELU of the Perceiver FeedForward (src/helpers.py:32-39)
//   colsum                bias gradients
//   rope_bwd_merge        dq, dk, dv [B,H,L,Dh] -> d(qkv) [B,L,3*H*Dh] through the transpose of the rotation
//   ce_fwd_bwd            shifted cross-entropy over V' logits: per-row loss + d(logits) in one kernel (a13)
//   grad_sqnorm / adamw   global L2 norm of the bf16 gradients, then clip + AdamW on fp32 master weights that also
//                         emits the bf16 weights the next forward reads
#include "aki_device.h"

namespace aki {

// ------------------------------------------------------------------------------------------------------------
// transpose
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* x, bf16_t* y, int R, int C, int ldx, int ldy, int Rpad) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 16 * i, c = c0 + tx * 4;
    u32x2 v = {0u, 0u};
    if (r < R) {
      if (c + 3 < C && (ldx & 3) == 0) v = *(const u32x2*)(x + (size_t)r * ldx + c);
      else {
        bf16_t e[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k) if (c + k < C) e[k] = x[(size_t)r * ldx + c + k];
        v[0] = e[0] | ((unsigned)e[1] << 16); v[1] = e[2] | ((unsigned)e[3] << 16);
      }
    }
    *(unsigned*)&tile[ty + 16 * i][tx * 4] = v[0];
    *(unsigned*)&tile[ty + 16 * i][tx * 4 + 2] = v[1];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 16 * i, r = r0 + tx * 4;      // output row c, output columns r..r+3
    if (c >= C || r >= Rpad) continue;
    const unsigned lo = tile[tx * 4][ty + 16 * i] | ((unsigned)tile[tx * 4 + 1][ty + 16 * i] << 16);
    const unsigned hi = tile[tx * 4 + 2][ty + 16 * i] | ((unsigned)tile[tx * 4 + 3][ty + 16 * i] << 16);
    if (r + 3 < Rpad && (ldy & 3) == 0) *(u32x2*)(y + (size_t)c * ldy + r) = u32x2{lo, hi};
    else {
      const bf16_t e[4] = {(bf16_t)lo, (bf16_t)(lo >> 16), (bf16_t)hi, (bf16_t)(hi >> 16)};
      for (int k = 0; k < 4; ++k) if (r + k < Rpad) y[(size_t)c * ldy + r + k] = e[k];
    }
  }
}

int transpose_bf16_launch(const void* x, void* y, int R, int C, int ldx, int ldy, int Rpad, hipStream_t s) {
  const dim3 grid((C + 63) / 64, (Rpad + 63) / 64), block(256);
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, block, 0, s, (const bf16_t*)x, (bf16_t*)y, R, C, ldx, ldy, Rpad);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// ------------------------------------------------------------------------------------------------------------
// norm backward.  Forward (aux_kernels.hip): RMS: y = w * bf16(x * rstd);  LN: y = (x - mean) * rstd * w + b.
//   dxhat = dy * w;  dx = rstd * (dxhat - [mean(dxhat)] - xhat * mean(dxhat * xhat));  dw = sum_rows dy * xhat
// One workgroup walks rows blockIdx.x, +gridDim.x, ...; thread t owns column chunks t, t+256, ... (8 columns each)
// and keeps its dw/db partial sums in registers; the partials [gridDim.x][cols] are folded by norm_bwd_finish.
// ------------------------------------------------------------------------------------------------------------
template <bool RMS>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const bf16_t* x, const bf16_t* w, const bf16_t* dy, const bf16_t* dres, bf16_t* dx,
                                                       float* dw_part, float* db_part, int rows, int cols, int ldx, int lddy,
                                                       int lddx, int lddr, float eps) {
  constexpr int MAXC = 2;          // cols <= 4096
  __shared__ float red[16];
  const int nchunk = cols / 8, tid = threadIdx.x;
  float dwp[MAXC][8], dbp[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { dwp[i][e] = 0.f; dbp[i][e] = 0.f; }
  u32x4 wv[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    wv[i] = c < nchunk ? *(const u32x4*)(w + c * 8) : u32x4{0u, 0u, 0u, 0u};
  }
  // The row walked NEXT is requested before this row's three block reductions: a workgroup's ten rows were ten serial DRAM round
  // trips with two resident workgroups per CU to hide them (34.5 us per launch against 16 at the HBM rate).
  u32x4 nx[MAXC], ng[MAXC], nr[MAXC];
  auto request = [&](int row) {     // the residual-branch gradient is asked for with the row, not behind the reductions
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * 256;
      nx[i] = u32x4{0u, 0u, 0u, 0u}; ng[i] = nx[i]; nr[i] = nx[i];
      if (c < nchunk && row < rows) {
        nx[i] = *(const u32x4*)(x + (size_t)row * ldx + c * 8);
        ng[i] = *(const u32x4*)(dy + (size_t)row * lddy + c * 8);
        if (dres) nr[i] = *(const u32x4*)(dres + (size_t)row * lddr + c * 8);
      }
    }
  };
  request(blockIdx.x);
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    u32x4 xv[MAXC], gv[MAXC], rv[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) { xv[i] = nx[i]; gv[i] = ng[i]; rv[i] = nr[i]; }
    request(row + gridDim.x);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if (tid + i * 256 < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = bf16_lo(xv[i][e]), b = bf16_hi(xv[i][e]);
          s1 += a + b;
          s2 += a * a + b * b;
        }
      }
    }
    block_sum2<256>(s1, s2, red);
    const float mean = RMS ? 0.f : s1 / cols;
    float var = s2 / cols - mean * mean;
    if (!RMS) {   // two-pass variance like the forward
      float d2 = 0.f, dummy = 0.f;
#pragma unroll
      for (int i = 0; i < MAXC; ++i)
        if (tid + i * 256 < nchunk)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = bf16_lo(xv[i][e]) - mean, b = bf16_hi(xv[i][e]) - mean;
            d2 += a * a + b * b;
          }
      block_sum2<256>(d2, dummy, red);
      var = d2 / cols;
    }
    const float rstd = rsqrtf(var + eps);
    // c1 = mean(dxhat * xhat), c2 = mean(dxhat) (LN only)
    float c1 = 0.f, c2 = 0.f;
    float xh[MAXC][8], dxh[MAXC][8];
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = (bf16_lo(xv[i][e]) - mean) * rstd, x1 = (bf16_hi(xv[i][e]) - mean) * rstd;
        const float g0 = bf16_lo(gv[i][e]), g1 = bf16_hi(gv[i][e]);
        const float d0 = g0 * bf16_lo(wv[i][e]), d1 = g1 * bf16_hi(wv[i][e]);
        xh[i][2 * e] = x0; xh[i][2 * e + 1] = x1;
        dxh[i][2 * e] = d0; dxh[i][2 * e + 1] = d1;
        c1 += d0 * x0 + d1 * x1;
        c2 += d0 + d1;
        dwp[i][2 * e] += g0 * x0; dwp[i][2 * e + 1] += g1 * x1;
        dbp[i][2 * e] += g0; dbp[i][2 * e + 1] += g1;
      }
    block_sum2<256>(c1, c2, red);
    c1 /= cols;
    c2 = RMS ? 0.f : c2 / cols;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * 256;
      if (c < nchunk) {
        u32x4 o;
        const u32x4 rr = rv[i];     // gradient of the residual branch, summed in here
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_bf16x2(rstd * (dxh[i][2 * e] - c2 - xh[i][2 * e] * c1) + bf16_lo(rr[e]),
                             rstd * (dxh[i][2 * e + 1] - c2 - xh[i][2 * e + 1] * c1) + bf16_hi(rr[e]));
        *(u32x4*)(dx + (size_t)row * lddx + c * 8) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    if (c < nchunk) {
      float* dst = dw_part + (size_t)blockIdx.x * cols + c * 8;
      *(f32x4*)dst = f32x4{dwp[i][0], dwp[i][1], dwp[i][2], dwp[i][3]};
      *(f32x4*)(dst + 4) = f32x4{dwp[i][4], dwp[i][5], dwp[i][6], dwp[i][7]};
      if (!RMS && db_part) {
        float* d2 = db_part + (size_t)blockIdx.x * cols + c * 8;
        *(f32x4*)d2 = f32x4{dbp[i][0], dbp[i][1], dbp[i][2], dbp[i][3]};
        *(f32x4*)(d2 + 4) = f32x4{dbp[i][4], dbp[i][5], dbp[i][6], dbp[i][7]};
      }
    }
  }
}

// out[c] (+)= sum_g part[g][c]   (bf16 output; `accumulate` adds to the existing value)
// workgroup = 16 columns x 16 partial-row lanes (192 workgroups for 3072 columns - 64 columns per workgroup left 208 CUs idle and each thread
// walking 128 rows: 36 us per launch, 105 launches per training step); the lanes' sums are folded through LDS in a fixed order.
constexpr int FOLD_COLS = 16;
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* part, bf16_t* out, int G, int cols, int accumulate) {
  __shared__ float red[16][FOLD_COLS + 1];
  const int cl = threadIdx.x & (FOLD_COLS - 1), gl = threadIdx.x / FOLD_COLS;
  const int c = blockIdx.x * FOLD_COLS + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int g = gl;
    for (; g + 16 < G; g += 32) { s0 += part[(size_t)g * cols + c]; s1 += part[(size_t)(g + 16) * cols + c]; }
    if (g < G) s0 += part[(size_t)g * cols + c];
  }
  red[gl][cl] = s0 + s1;
  __syncthreads();
  if (gl == 0 && c < cols) {
    float t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = red[i][cl];
    lds_fold_ready(t);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += t[i];
    if (accumulate) s += bf16_bits_to_f32(out[c]);
    ((__bf16*)out)[c] = (__bf16)s;
  }
}

constexpr int NORM_BWD_GROUPS = 512;

size_t norm_bwd_ws_bytes(int cols) { return (size_t)2 * NORM_BWD_GROUPS * cols * 4; }

int norm_bwd_launch(bool rms, const void* x, const void* w, const void* dy, const void* dres, void* dx, void* dw, void* db, int rows,
                    int cols, int ldx, int lddy, int lddx, int lddr, float eps, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  if (cols % 8 || cols > 4096) return AKI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < norm_bwd_ws_bytes(cols)) return AKI_ERR_WORKSPACE;
  const int G = rows < NORM_BWD_GROUPS ? rows : NORM_BWD_GROUPS;
  float* dwp = (float*)ws;
  float* dbp = dwp + (size_t)NORM_BWD_GROUPS * cols;
  AKI_CLEAR_ERR();
  if (rms) hipLaunchKernelGGL(norm_bwd_kernel<true>, dim3(G), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)dy, (const bf16_t*)dres, (bf16_t*)dx, dwp, dbp, rows, cols, ldx, lddy, lddx, lddr, eps);
  else hipLaunchKernelGGL(norm_bwd_kernel<false>, dim3(G), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)dy, (const bf16_t*)dres, (bf16_t*)dx, dwp, dbp, rows, cols, ldx, lddy, lddx, lddr, eps);
  hipLaunchKernelGGL(fold_partials_kernel, dim3((cols + FOLD_COLS - 1) / FOLD_COLS), dim3(256), 0, s, dwp, (bf16_t*)dw, G, cols, accumulate);
  if (!rms && db) hipLaunchKernelGGL(fold_partials_kernel, dim3((cols + FOLD_COLS - 1) / FOLD_COLS), dim3(256), 0, s, dbp, (bf16_t*)db, G, cols, accumulate);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// ------------------------------------------------------------------------------------------------------------
// column sums (bias gradient): out[c] (+)= sum_r x[r][c]
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_part_kernel(const bf16_t* x, float* part, int rows, int cols, int ldx) {
  // block = 64 columns (8 chunks) x 32 row lanes
  const int cchunk = blockIdx.x * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cchunk * 8 < cols)
    for (int r = blockIdx.y * 32 + rl; r < rows; r += gridDim.y * 32) {
      const u32x4 v = *(const u32x4*)(x + (size_t)r * ldx + cchunk * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += bf16_lo(v[e]); acc[2 * e + 1] += bf16_hi(v[e]); }
    }
  __shared__ float red[32][65];
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][(threadIdx.x & 7) * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float t[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i] = red[i][threadIdx.x];
    lds_fold_ready(t);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += t[i];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < cols) part[(size_t)blockIdx.y * cols + c] = s;
  }
}

size_t colsum_ws_bytes(int cols) { return (size_t)64 * cols * 4; }

int colsum_launch(const void* x, void* out, int rows, int cols, int ldx, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  if (cols % 8 || ldx % 8) return AKI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < colsum_ws_bytes(cols)) return AKI_ERR_WORKSPACE;
  int G = (rows + 31) / 32;
  if (G > 64) G = 64;
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(colsum_part_kernel, dim3((cols + 63) / 64, G), dim3(256), 0, s, (const bf16_t*)x, (float*)ws, rows, cols, ldx);
  hipLaunchKernelGGL(fold_partials_kernel, dim3((cols + FOLD_COLS - 1) / FOLD_COLS), dim3(256), 0, s, (const float*)ws, (bf16_t*)out, G, cols, accumulate);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// ------------------------------------------------------------------------------------------------------------
// SwiGLU / GELU elementwise
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* gu, bf16_t* a, int rows, int F, int ldg, int lda) {
  const int nch = F / 8;
  const size_t total = (size_t)rows * nch;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / nch), c = (int)(i - (size_t)r * nch);
    const u32x4 g = *(const u32x4*)(gu + (size_t)r * ldg + c * 8), u = *(const u32x4*)(gu + (size_t)r * ldg + F + c * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16_lo(u[e]) * silu(bf16_lo(g[e])), bf16_hi(u[e]) * silu(bf16_hi(g[e])));
    *(u32x4*)(a + (size_t)r * lda + c * 8) = o;
  }
}

// d(gate_up) [rows][2F] from d(a) [rows][F] and the saved gate_up
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* gu, const bf16_t* da, bf16_t* dgu, int rows, int F, int ldg,
                                                         int ldda, int lddg) {
  const int nch = F / 8;
  const size_t total = (size_t)rows * nch;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / nch), c = (int)(i - (size_t)r * nch);
    const u32x4 g = *(const u32x4*)(gu + (size_t)r * ldg + c * 8), u = *(const u32x4*)(gu + (size_t)r * ldg + F + c * 8);
    const u32x4 d = *(const u32x4*)(da + (size_t)r * ldda + c * 8);
    u32x4 og, ou;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dg[2], du[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float gg = hh ? bf16_hi(g[e]) : bf16_lo(g[e]), uu = hh ? bf16_hi(u[e]) : bf16_lo(u[e]), dd = hh ? bf16_hi(d[e]) : bf16_lo(d[e]);
        const float sg = 1.0f / (1.0f + __expf(-gg));
        du[hh] = dd * gg * sg;
        dg[hh] = dd * uu * sg * (1.0f + gg * (1.0f - sg));
      }
      og[e] = pack_bf16x2(dg[0], dg[1]);
      ou[e] = pack_bf16x2(du[0], du[1]);
    }
    *(u32x4*)(dgu + (size_t)r * lddg + c * 8) = og;
    *(u32x4*)(dgu + (size_t)r * lddg + F + c * 8) = ou;
  }
}

// MODE 0: y = gelu_erf(x);  MODE 1: dx = dy * gelu_erf'(x)
template <int MODE>
__global__ __launch_bounds__(256) void gelu_kernel(const bf16_t* x, const bf16_t* dy, bf16_t* out, size_t nchunks) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
    const u32x4 v = *(const u32x4*)(x + i * 8);
    u32x4 d = {0u, 0u, 0u, 0u}, o;
    if (MODE == 1) d = *(const u32x4*)(dy + i * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float xx = hh ? bf16_hi(v[e]) : bf16_lo(v[e]);
        if (MODE == 0) r[hh] = gelu_erf(xx);
        else {
          const float cdf = 0.5f * (1.0f + erff(xx * 0.70710678118654752440f));
          const float pdf = 0.39894228040143267794f * __expf(-0.5f * xx * xx);
          r[hh] = (hh ? bf16_hi(d[e]) : bf16_lo(d[e])) * (cdf + xx * pdf);
        }
      }
      o[e] = pack_bf16x2(r[0], r[1]);
    }
    *(u32x4*)(out + i * 8) = o;
  }
}

static inline int ew_grid(size_t work_items) {
  size_t g = (work_items + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

int swiglu_launch(int bwd, const void* gu, const void* da, void* out, int rows, int F, int ldg, int ldda, int ldo, hipStream_t s) {
  if (F % 8 || ldg % 8 || ldo % 8 || (bwd && ldda % 8)) return AKI_ERR_UNSUPPORTED;
  AKI_CLEAR_ERR();
  const int grid = ew_grid((size_t)rows * (F / 8));
  if (bwd) hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)gu, (const bf16_t*)da, (bf16_t*)out, rows, F, ldg, ldda, ldo);
  else hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)gu, (bf16_t*)out, rows, F, ldg, ldo);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

int gelu_launch(int bwd, const void* x, const void* dy, void* out, size_t n, hipStream_t s) {
  if (n % 8) return AKI_ERR_UNSUPPORTED;
  AKI_CLEAR_ERR();
  const int grid = ew_grid(n / 8);
  if (bwd) hipLaunchKernelGGL(gelu_kernel<1>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, n / 8);
  else hipLaunchKernelGGL(gelu_kernel<0>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (bf16_t*)out, n / 8);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// ------------------------------------------------------------------------------------------------------------
// RoPE backward + merge: dq, dk, dv [B][H][L][Dh] -> d(qkv) [B*L][3*H*Dh].
// forward: y = x*cos + rot(x)*sin with rot(x)[d] = -x[d+half] (d < half), x[d-half] (d >= half); its transpose is
// dx = dy*cos - rot(dy*sin).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_bwd_merge_kernel(const bf16_t* dq, const bf16_t* dk, const bf16_t* dv, const float* cos,
                                                             const float* sin, const int* pos, bf16_t* dqkv, int B, int H, int L, int Dh) {
  const int half = Dh / 2, hc = half / 8;              // chunks of 8 in one half
  const size_t total = (size_t)B * H * L * hc;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % hc);
    const size_t row = i / hc;                          // (b*H + h)*L + t
    const int t = (int)(row % L);
    const size_t bh = row / L;
    const int h = (int)(bh % H), b = (int)(bh / H);
    const int p = pos ? pos[(size_t)b * L + t] : t;
    const float* cs = cos + (size_t)p * Dh + c * 8;
    const float* sn = sin + (size_t)p * Dh + c * 8;
    bf16_t* out = dqkv + ((size_t)b * L + t) * 3 * H * Dh + (size_t)h * Dh + c * 8;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const bf16_t* src = (which ? dk : dq) + row * Dh + c * 8;
      const u32x4 lo = *(const u32x4*)src, hi = *(const u32x4*)(src + half);
      u32x4 olo, ohi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float r0[2], r1[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int d = 2 * e + hh;
          const float y0 = hh ? bf16_hi(lo[e]) : bf16_lo(lo[e]), y1 = hh ? bf16_hi(hi[e]) : bf16_lo(hi[e]);
          // dx[d] = y[d] cos[d] + y[d+half] sin[d+half];  dx[d+half] = y[d+half] cos[d+half] - y[d] sin[d]
          r0[hh] = y0 * cs[d] + y1 * sn[half + d];
          r1[hh] = y1 * cs[half + d] - y0 * sn[d];
        }
        olo[e] = pack_bf16x2(r0[0], r0[1]);
        ohi[e] = pack_bf16x2(r1[0], r1[1]);
      }
      *(u32x4*)(out + (size_t)which * H * Dh) = olo;
      *(u32x4*)(out + (size_t)which * H * Dh + half) = ohi;
    }
    const bf16_t* sv = dv + row * Dh + c * 8;
    *(u32x4*)(out + (size_t)2 * H * Dh) = *(const u32x4*)sv;
    *(u32x4*)(out + (size_t)2 * H * Dh + half) = *(const u32x4*)(sv + half);
  }
}

int rope_bwd_merge_launch(const void* dq, const void* dk, const void* dv, const float* cos, const float* sin, const int* pos, void* dqkv,
                          int B, int H, int L, int Dh, hipStream_t s) {
  if (Dh % 16) return AKI_ERR_UNSUPPORTED;
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(rope_bwd_merge_kernel, dim3(ew_grid((size_t)B * H * L * (Dh / 16))), dim3(256), 0, s, (const bf16_t*)dq,
                     (const bf16_t*)dk, (const bf16_t*)dv, cos, sin, pos, (bf16_t*)dqkv, B, H, L, Dh);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Shifted cross-entropy, forward + backward in one pass over the logits (HF ForCausalLMLoss as used by
// train/losses.py:83-116): row (b, t) is scored against labels[b][t+1]; rows with label -100 and t = L-1 are ignored.
//   loss_rows[b*L+t] = lse - logit[target]   (0 for ignored rows)
//   dlogits[row][c]  = (softmax(row)[c] - [c == target]) * gscale / n_valid      (0 for ignored rows)
// n_valid comes from a device counter written by ce_count_kernel, so nothing syncs with the host.
// ------------------------------------------------------------------------------------------------------------
__global__ void ce_count_kernel(const int64_t* labels, int B, int L, int V, int* n_valid) {
  int cnt = 0;
  for (int i = threadIdx.x; i < B * L; i += 256) {
    const int t = i % L;
    if (t + 1 < L && labels[i + 1] >= 0 && labels[i + 1] < V) ++cnt;
  }
  __shared__ int red[256];
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_valid = red[0];
}

// ROWS = true: `labels` holds one already-shifted target per row of an arbitrary chunk of rows (aki_ce_rows_fwd_bwd)
template <bool ROWS>
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const bf16_t* logits, const int64_t* labels, const int* n_valid, float* loss_rows,
                                                         bf16_t* dlogits, int B, int L, int V, int ldl, int lddl, float gscale) {
  __shared__ float red[16];
  const int row = blockIdx.x, t = ROWS ? 0 : row % L, tid = threadIdx.x;
  int64_t tgt = ROWS ? labels[row] : ((t + 1 < L) ? labels[row + 1] : -100);
  if (tgt < 0 || tgt >= V) tgt = -100;   // out-of-range labels are ignored rows (ce_count_kernel counts the same way), never an OOB read
  const bf16_t* lr = logits + (size_t)row * ldl;
  bf16_t* dr = dlogits ? dlogits + (size_t)row * lddl : nullptr;
  // The target logit is read BEFORE any barrier: dlogits may alias logits, and once the last __syncthreads is behind them the
  // other waves overwrite lr[] with the gradient while thread 0 would still be reading lr[tgt] for the loss.
  float tgt_logit = 0.f;
  if (tid == 0 && tgt != -100) tgt_logit = bf16_bits_to_f32(lr[tgt]);
  if (tgt == -100) {
    if (tid == 0) loss_rows[row] = 0.f;
    if (dr) for (int c = tid * 2; c < V; c += 512) { if (c + 1 < V) *(unsigned*)(dr + c) = 0u; else dr[c] = 0; }
    return;
  }
  // online max / sum
  float m = -INFINITY, s = 0.f;
  for (int c = tid * 2; c < V; c += 512) {
    const float a = bf16_bits_to_f32(lr[c]), b = (c + 1 < V) ? bf16_bits_to_f32(lr[c + 1]) : -INFINITY;
    const float mn = fmaxf(m, fmaxf(a, b));
    s = s * __expf(m - mn) + __expf(a - mn) + __expf(b - mn);
    m = mn;
  }
  // block reduce (m, s)
  float gm = m;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o));
  if ((tid & 63) == 0) red[tid >> 6] = gm;
  __syncthreads();
  gm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sc = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
  __syncthreads();
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sc;
  __syncthreads();
  const float st = red[4] + red[5] + red[6] + red[7];
  const float lse = gm + __logf(st);
  if (tid == 0) loss_rows[row] = lse - tgt_logit;
  if (dr) {
    const float k = gscale / (float)max(*n_valid, 1);
    for (int c = tid * 2; c < V; c += 512) {
      const float p0 = __expf(bf16_bits_to_f32(lr[c]) - lse) - (c == tgt ? 1.f : 0.f);
      if (c + 1 < V) {
        const float p1 = __expf(bf16_bits_to_f32(lr[c + 1]) - lse) - (c + 1 == tgt ? 1.f : 0.f);
        *(unsigned*)(dr + c) = pack_bf16x2(p0 * k, p1 * k);
      } else ((__bf16*)dr)[c] = (__bf16)(p0 * k);
    }
  }
}

int ce_launch(const void* logits, const int64_t* labels, int* n_valid, float* loss_rows, void* dlogits, int B, int L, int V, int ldl,
              int lddl, float gscale, hipStream_t s) {
  if ((ldl & 1) || (dlogits && (lddl & 1))) return AKI_ERR_ALIGNMENT;
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, s, labels, B, L, V, n_valid);
  hipLaunchKernelGGL(ce_fwd_bwd_kernel<false>, dim3(B * L), dim3(256), 0, s, (const bf16_t*)logits, labels, n_valid, loss_rows, (bf16_t*)dlogits,
                     B, L, V, ldl, lddl, gscale);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

int ce_rows_launch(const void* logits, const int64_t* targets, const int* n_valid, float* loss_rows, void* dlogits, int rows, int V,
                   int ldl, int lddl, float gscale, hipStream_t s) {
  if ((ldl & 1) || (dlogits && (lddl & 1))) return AKI_ERR_ALIGNMENT;
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(ce_fwd_bwd_kernel<true>, dim3(rows), dim3(256), 0, s, (const bf16_t*)logits, targets, n_valid, loss_rows,
                     (bf16_t*)dlogits, 1, rows, V, ldl, lddl, gscale);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Gradient norm + AdamW (train/train_utils.py:256-258: clip_grad_norm_(1.0), optimizer.step()).
//   grad_sqnorm: sum of squares of a bf16 gradient buffer -> partial[gridDim.x]; finish folds them into *out (+=)
//   adamw:       g' = g * gscale * clip,  clip = min(1, max_norm / (sqrt(*sqnorm) * gscale + 1e-6))   (torch semantics)
//                m, v, p updated in fp32 (decoupled weight decay, bias correction); w16 = bf16(p)
// ------------------------------------------------------------------------------------------------------------
template <bool G32>
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const void* gp, size_t nchunks, float* part) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
    if constexpr (G32) {
      const f32x4 a = *(const f32x4*)((const float*)gp + i * 8), b = *(const f32x4*)((const float*)gp + i * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += a[e] * a[e] + b[e] * b[e];
    } else {
      const u32x4 v = *(const u32x4*)((const bf16_t*)gp + i * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float a = bf16_lo(v[e]), b = bf16_hi(v[e]); s += a * a + b * b; }
    }
  }
  __shared__ float red[16];
  float dummy = 0.f;
  block_sum2<256>(s, dummy, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void fold_scalar_kernel(const float* part, int n, float* out, int accumulate) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  __shared__ float red[16];
  float dummy = 0.f;
  block_sum2<256>(s, dummy, red);
  if (threadIdx.x == 0) *out = accumulate ? *out + s : s;
}

constexpr int SQNORM_GROUPS = 1024;
size_t grad_sqnorm_ws_bytes() { return SQNORM_GROUPS * 4; }

int grad_sqnorm_launch(const void* g, size_t n, float* out, int accumulate, void* ws, size_t ws_bytes, bool g32, hipStream_t s) {
  if (n % 8) return AKI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < grad_sqnorm_ws_bytes()) return AKI_ERR_WORKSPACE;
  int G = (int)((n / 8 + 255) / 256);
  if (G > SQNORM_GROUPS) G = SQNORM_GROUPS;
  if (G < 1) G = 1;
  AKI_CLEAR_ERR();
  if (g32) hipLaunchKernelGGL(grad_sqnorm_kernel<true>, dim3(G), dim3(256), 0, s, g, n / 8, (float*)ws);
  else hipLaunchKernelGGL(grad_sqnorm_kernel<false>, dim3(G), dim3(256), 0, s, g, n / 8, (float*)ws);
  hipLaunchKernelGGL(fold_scalar_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, G, out, accumulate);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

struct AdamwParams {
  float* p; float* m; float* v; const void* g; bf16_t* w16; size_t nchunks;
  const float* sqnorm; float max_norm, gscale, lr, beta1, beta2, eps, wd, bc1, bc2;
};

template <bool G32>
__global__ __launch_bounds__(256) void adamw_kernel(const AdamwParams a) {
  float clip = 1.f;
  if (a.sqnorm && a.max_norm > 0.f) clip = fminf(1.f, a.max_norm / (sqrtf(*a.sqnorm) * a.gscale + 1e-6f));
  const float gs = a.gscale * clip;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.nchunks; i += (size_t)gridDim.x * 256) {
    float gin[8];
    if constexpr (G32) {
      *(f32x4*)&gin[0] = *(const f32x4*)((const float*)a.g + i * 8);
      *(f32x4*)&gin[4] = *(const f32x4*)((const float*)a.g + i * 8 + 4);
    } else {
      const u32x4 gv = *(const u32x4*)((const bf16_t*)a.g + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) gin[j] = (j & 1) ? bf16_hi(gv[j >> 1]) : bf16_lo(gv[j >> 1]);
    }
    float p[8], m[8], v[8];
    *(f32x4*)&p[0] = *(const f32x4*)(a.p + i * 8); *(f32x4*)&p[4] = *(const f32x4*)(a.p + i * 8 + 4);
    *(f32x4*)&m[0] = *(const f32x4*)(a.m + i * 8); *(f32x4*)&m[4] = *(const f32x4*)(a.m + i * 8 + 4);
    *(f32x4*)&v[0] = *(const f32x4*)(a.v + i * 8); *(f32x4*)&v[4] = *(const f32x4*)(a.v + i * 8 + 4);
    u32x4 wo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = gin[j] * gs;
      p[j] *= 1.f - a.lr * a.wd;
      m[j] = a.beta1 * m[j] + (1.f - a.beta1) * g;
      v[j] = a.beta2 * v[j] + (1.f - a.beta2) * g * g;
      p[j] -= a.lr * (m[j] / a.bc1) / (sqrtf(v[j] / a.bc2) + a.eps);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) wo[e] = pack_bf16x2(p[2 * e], p[2 * e + 1]);
    *(f32x4*)(a.p + i * 8) = *(f32x4*)&p[0]; *(f32x4*)(a.p + i * 8 + 4) = *(f32x4*)&p[4];
    *(f32x4*)(a.m + i * 8) = *(f32x4*)&m[0]; *(f32x4*)(a.m + i * 8 + 4) = *(f32x4*)&m[4];
    *(f32x4*)(a.v + i * 8) = *(f32x4*)&v[0]; *(f32x4*)(a.v + i * 8 + 4) = *(f32x4*)&v[4];
    *(u32x4*)(a.w16 + i * 8) = wo;
  }
}

// The same update for ONE 2-D weight [N, K] that also leaves W^T [K, ldT] (ldT = N padded to x64, padding columns zero) - the operand the
// backward's input-gradient GEMMs read (train_ops._weight_t).  The bf16 weights are being written anyway: turning each 64 x 64 tile through LDS
// costs one more 2-byte store per element and saves the separate aki_transpose pass over every weight after every optimizer step (read + write).
struct AdamwTParams {
  float* p; float* m; float* v; const void* g; bf16_t* w16; bf16_t* wT; int N, K, ldT;
  const float* sqnorm; float max_norm, gscale, lr, beta1, beta2, eps, wd, bc1, bc2;
};

template <bool G32>
__global__ __launch_bounds__(256) void adamw_t_kernel(const AdamwTParams a) {
  __shared__ bf16_t tile[64][66];
  float clip = 1.f;
  if (a.sqnorm && a.max_norm > 0.f) clip = fminf(1.f, a.max_norm / (sqrtf(*a.sqnorm) * a.gscale + 1e-6f));
  const float gs = a.gscale * clip;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 16 * i, c = c0 + tx * 4;
    u32x2 wo = {0u, 0u};
    if (r < a.N && c < a.K) {                     // K is a multiple of 4: a 4-column run is inside the row or outside it
      const size_t idx = (size_t)r * a.K + c;
      float gin[4];
      if constexpr (G32) {
        *(f32x4*)&gin[0] = *(const f32x4*)((const float*)a.g + idx);
      } else {
        const u32x2 gv = *(const u32x2*)((const bf16_t*)a.g + idx);
        gin[0] = bf16_lo(gv[0]); gin[1] = bf16_hi(gv[0]); gin[2] = bf16_lo(gv[1]); gin[3] = bf16_hi(gv[1]);
      }
      float p[4], m[4], v[4];
      *(f32x4*)&p[0] = *(const f32x4*)(a.p + idx);
      *(f32x4*)&m[0] = *(const f32x4*)(a.m + idx);
      *(f32x4*)&v[0] = *(const f32x4*)(a.v + idx);
#pragma unroll
      for (int j = 0; j < 4; ++j) {               // the arithmetic of adamw_kernel, operation for operation
        const float g = gin[j] * gs;
        p[j] *= 1.f - a.lr * a.wd;
        m[j] = a.beta1 * m[j] + (1.f - a.beta1) * g;
        v[j] = a.beta2 * v[j] + (1.f - a.beta2) * g * g;
        p[j] -= a.lr * (m[j] / a.bc1) / (sqrtf(v[j] / a.bc2) + a.eps);
      }
      *(f32x4*)(a.p + idx) = *(f32x4*)&p[0];
      *(f32x4*)(a.m + idx) = *(f32x4*)&m[0];
      *(f32x4*)(a.v + idx) = *(f32x4*)&v[0];
      wo = u32x2{pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3])};
      *(u32x2*)(a.w16 + idx) = wo;
    }
    *(unsigned*)&tile[ty + 16 * i][tx * 4] = wo[0];            // rows >= N / columns >= K: zeros (the padding of W^T)
    *(unsigned*)&tile[ty + 16 * i][tx * 4 + 2] = wo[1];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 16 * i, r = r0 + tx * 4;            // output row c of W^T, output columns r..r+3
    if (c >= a.K || r >= a.ldT) continue;
    const unsigned lo = tile[tx * 4][ty + 16 * i] | ((unsigned)tile[tx * 4 + 1][ty + 16 * i] << 16);
    const unsigned hi = tile[tx * 4 + 2][ty + 16 * i] | ((unsigned)tile[tx * 4 + 3][ty + 16 * i] << 16);
    *(u32x2*)(a.wT + (size_t)c * a.ldT + r) = u32x2{lo, hi};
  }
}

int adamw_t_launch(float* p, float* m, float* v, const void* g, void* w16, void* wT, int N, int K, int ldT, const float* sqnorm, float max_norm,
                   float gscale, float lr, float beta1, float beta2, float eps, float wd, int step, bool g32, hipStream_t s) {
  if (N <= 0 || K <= 0 || K % 4 || ldT % 64 || ldT < N || step < 1) return AKI_ERR_INVALID_ARG;
  if ((((size_t)p | (size_t)m | (size_t)v) & 15) || ((size_t)w16 & 7) || ((size_t)wT & 7) || ((size_t)g & (g32 ? 15 : 7))) return AKI_ERR_ALIGNMENT;
  AdamwTParams a = {p, m, v, g, (bf16_t*)w16, (bf16_t*)wT, N, K, ldT, sqnorm, max_norm, gscale, lr, beta1, beta2, eps, wd,
                    1.f - powf(beta1, (float)step), 1.f - powf(beta2, (float)step)};
  const dim3 grid((K + 63) / 64, ldT / 64);
  AKI_CLEAR_ERR();
  if (g32) hipLaunchKernelGGL(adamw_t_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(adamw_t_kernel<false>, grid, dim3(256), 0, s, a);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

int adamw_launch(float* p, float* m, float* v, const void* g, void* w16, size_t n, const float* sqnorm, float max_norm, float gscale,
                 float lr, float beta1, float beta2, float eps, float wd, int step, bool g32, hipStream_t s) {
  if (n % 8 || step < 1) return AKI_ERR_INVALID_ARG;
  AdamwParams a = {p, m, v, g, (bf16_t*)w16, n / 8, sqnorm, max_norm, gscale, lr, beta1, beta2, eps, wd,
                   1.f - powf(beta1, (float)step), 1.f - powf(beta2, (float)step)};
  AKI_CLEAR_ERR();
  if (g32) hipLaunchKernelGGL(adamw_kernel<true>, dim3(ew_grid(n / 8)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(adamw_kernel<false>, dim3(ew_grid(n / 8)), dim3(256), 0, s, a);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
