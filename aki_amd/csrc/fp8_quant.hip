// fp8_quant.hip - per-row dynamic quantisation to OCP fp8 e4m3 (BASELINE configs[4]: fp8 weights / activations).
//
//   q[r][c] = e4m3( y[r][c] / s[r] ),   s[r] = max_c |y[r][c]| / 448,   y = x  or  y = rmsnorm(x) (Phi3RMSNorm fused in)
// One scale per row (per token for activations, per output feature for nn.Linear weights); the fp8 GEMM multiplies the
// f32 accumulator by s_x[token] * s_w[feature] (gemm_bf16.hip, FP8 variant).  HBM-bound: 2 B read + 1 B written per
// element, the row stays in registers between the reductions.  gfx950 speaks OCP e4m3fn (max 448), not MI300's fnuz.
#include "aki_device.h"

namespace aki {

template <bool RMS>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* x, const bf16_t* rms_w, float eps, uint8_t* q, float* scale,
                                                             int cols, int ldx, int ldq) {
  constexpr int MAXC = 4;                 // cols <= 8192
  __shared__ float red[16];
  const int row = blockIdx.x, tid = threadIdx.x, nchunk = cols / 8;
  const bf16_t* xr = x + (size_t)row * ldx;
  float v[MAXC][8];
  float ss = 0.f, dummy = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    u32x4 b = {0u, 0u, 0u, 0u};
    if (c < nchunk) b = *(const u32x4*)(xr + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][2 * e] = bf16_lo(b[e]); v[i][2 * e + 1] = bf16_hi(b[e]);
      ss += v[i][2 * e] * v[i][2 * e] + v[i][2 * e + 1] * v[i][2 * e + 1];
    }
  }
  if (RMS) {
    block_sum2<256>(ss, dummy, red);
    const float rstd = rsqrtf(ss / cols + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * 256;
      if (c < nchunk) {
        const u32x4 w = *(const u32x4*)(rms_w + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // HF Phi3RMSNorm: weight * (x * rstd).to(bf16), result in bf16
          v[i][2 * e] = round_bf16(round_bf16(v[i][2 * e] * rstd) * bf16_lo(w[e]));
          v[i][2 * e + 1] = round_bf16(round_bf16(v[i][2 * e + 1] * rstd) * bf16_hi(w[e]));
        }
      }
    }
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
  amax = wave_max(amax);
  if ((tid & 63) == 0) red[8 + (tid >> 6)] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
  const float s = fmaxf(amax, 1e-12f) * (1.0f / 448.0f);
  const float inv = 1.0f / s;
  if (tid == 0) scale[row] = s;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    if (c < nchunk) {
      u32x2 o;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = fminf(fmaxf(v[i][4 * h2 + e] * inv, -448.f), 448.f);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], w, true);
        o[h2] = (unsigned)w;
      }
      *(u32x2*)(q + (size_t)row * ldq + c * 8) = o;
    }
  }
}

int quant_rows_fp8_launch(const void* x, const void* rms_w, float eps, void* q, float* scale, int rows, int cols, int ldx, int ldq,
                          hipStream_t s) {
  if (cols % 8 || cols > 8192 || (ldx % 8) || (ldq % 8)) return AKI_ERR_UNSUPPORTED;
  AKI_CLEAR_ERR();
  if (rms_w) hipLaunchKernelGGL(quant_rows_fp8_kernel<true>, dim3(rows), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)rms_w, eps, (uint8_t*)q, scale, cols, ldx, ldq);
  else hipLaunchKernelGGL(quant_rows_fp8_kernel<false>, dim3(rows), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, eps, (uint8_t*)q, scale, cols, ldx, ldq);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
