// decode.hip - the token-by-token decode path after the MMA prefill.
//
// Replaces, for the generate() loop of the reference (src/aki.py:136-209 + src/aki_generation.py:36-86, HF
// GenerationMixin greedy decoding): per new token the 32 decoder layers run with M = batch rows (1..8).  That
// regime is HBM-bound (every weight is read once per token), so the kernels here stream weights at full width
// instead of using MFMA tiles:
//   gemv_bf16_kernel    y = act(x W^T + b) [+ residual] for M <= 8 rows: x lives in LDS, each wave owns 2 output
//                       features and sweeps W rows with coalesced 16-byte loads + v_dot2c_f32_bf16
//   rope_append_kernel  split the fused qkv row, rotate q/k at the token's position, append k/v to the KV cache
//   decode_attn_kernel  one query per (batch, head) against the cache: keys are spread over the 256 lanes, each
//                       lane keeps an online-softmax partial (m, l, acc[Dh]) that is merged through LDS
// After the prefill the reference switches to an all-ones 2-D mask (src/aki_generation.py:58-62), i.e. plain causal
// attention over everything cached; per-sample cache lengths and the prefill's valid-column bits are honoured here,
// which lifts the reference's batch-1 restriction.
#include "aki_device.h"

namespace aki {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// NB: indexing the u32x4 and bit-casting each dword (bit_cast<bf16x2>(a[i])) is folded by hipcc 7.2 into four uses of
// dword 0; viewing the whole 16 bytes as bf16x8 and slicing pairs with shufflevector selects the right operands.
__device__ __forceinline__ float dot8_bf16(const u32x4 a, const u32x4 b, float acc) {
  const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, a), b8 = __builtin_bit_cast(bf16x8_t, b);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a8, a8, 0, 1), __builtin_shufflevector(b8, b8, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a8, a8, 2, 3), __builtin_shufflevector(b8, b8, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a8, a8, 4, 5), __builtin_shufflevector(b8, b8, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a8, a8, 6, 7), __builtin_shufflevector(b8, b8, 6, 7), acc, false);
  return acc;
}

struct GemvParams {
  const bf16_t* x; const bf16_t* w; const bf16_t* bias; const bf16_t* residual; bf16_t* y;
  int M, N, K, ldx, ldw, ldy, ldr, res_row_mod, act;
};

// FPW output features per wave; SWIGLU: feature f pairs weight rows f (gate) and N/2 + f (up).
template <int M, bool SWIGLU>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const GemvParams p) {
  constexpr int FPW = 2;
  constexpr int NR = SWIGLU ? 2 * FPW : FPW;   // weight rows per wave
  extern __shared__ __attribute__((aligned(16))) char sx[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = p.K / 8;
  for (int i = tid; i < M * nchunk; i += 256) {
    const int m = i / nchunk, c = i - m * nchunk;
    *(u32x4*)(sx + (size_t)i * 16) = *(const u32x4*)(p.x + (size_t)m * p.ldx + c * 8);
  }
  __syncthreads();
  const int n_out = SWIGLU ? p.N / 2 : p.N;
  const int f0 = (blockIdx.x * 4 + wave) * FPW;
  if (f0 >= n_out) return;
  const bf16_t* wr[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int f = min(f0 + (r % FPW), n_out - 1);
    wr[r] = p.w + (size_t)((SWIGLU && r >= FPW) ? n_out + f : f) * p.ldw;
  }
  float acc[NR][M];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
  for (int c = lane; c < nchunk; c += 64) {
    u32x4 w[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) w[r] = *(const u32x4*)(wr[r] + c * 8);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const u32x4 xv = *(const u32x4*)(sx + ((size_t)m * nchunk + c) * 16);
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r][m] = dot8_bf16(w[r], xv, acc[r][m]);
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float v = acc[r][m];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      acc[r][m] = v;
    }
  if (lane == 0) {
#pragma unroll
    for (int f = 0; f < FPW; ++f) {
      const int n = f0 + f;
      if (n >= n_out) continue;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float v;
        if (SWIGLU) {
          v = acc[FPW + f][m] * silu_fast(acc[f][m]);
        } else {
          v = acc[f][m];
          if (p.bias) v += bf16_bits_to_f32(p.bias[n]);
          if (p.act == AKI_ACT_GELU_ERF) v = gelu_erf_fast(v);
          else if (p.act == AKI_ACT_GELU_TANH) v = gelu_tanh_fast(v);
        }
        if (p.residual) v += bf16_bits_to_f32(p.residual[(size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + n]);
        ((__bf16*)p.y)[(size_t)m * p.ldy + n] = (__bf16)v;
      }
    }
  }
}

template <int M>
static int launch_gemv(const GemvParams& p, hipStream_t stream) {
  const size_t smem = (size_t)M * p.K * 2;
  const int n_out = p.act == AKI_ACT_SWIGLU ? p.N / 2 : p.N;
  const dim3 grid((n_out + 7) / 8), block(256);
  AKI_CLEAR_ERR();
  if (p.act == AKI_ACT_SWIGLU) {
    static bool set = false;
    if (!set) { if (hipFuncSetAttribute((const void*)gemv_bf16_kernel<M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192 * 2) != hipSuccess) return AKI_ERR_LAUNCH; set = true; }
    hipLaunchKernelGGL((gemv_bf16_kernel<M, true>), grid, block, smem, stream, p);
  } else {
    static bool set = false;
    if (!set) { if (hipFuncSetAttribute((const void*)gemv_bf16_kernel<M, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192 * 2) != hipSuccess) return AKI_ERR_LAUNCH; set = true; }
    hipLaunchKernelGGL((gemv_bf16_kernel<M, false>), grid, block, smem, stream, p);
  }
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// M <= 8 rows and M*K*2 <= 128 KiB of LDS; returns AKI_ERR_UNSUPPORTED otherwise (the caller then uses the MFMA GEMM)
int gemv_bf16(const aki_linear_args* a, hipStream_t stream) {
  if (a->M > 8 || a->K % 8 || (size_t)a->M * a->K * 2 > 8 * 8192 * 2 || (a->ldx % 8) || (a->ldw % 8)) return AKI_ERR_UNSUPPORTED;
  if (a->act == AKI_ACT_SWIGLU && (a->bias || (a->N & 1))) return AKI_ERR_UNSUPPORTED;
  GemvParams p = {(const bf16_t*)a->x, (const bf16_t*)a->w, (const bf16_t*)a->bias, (const bf16_t*)a->residual, (bf16_t*)a->y,
                  a->M, a->N, a->K, a->ldx, a->ldw, a->ldy, a->ldr, a->res_row_mod, a->act};
  switch (a->M) {
    case 1: return launch_gemv<1>(p, stream);
    case 2: return launch_gemv<2>(p, stream);
    case 3: return launch_gemv<3>(p, stream);
    case 4: return launch_gemv<4>(p, stream);
    case 5: return launch_gemv<5>(p, stream);
    case 6: return launch_gemv<6>(p, stream);
    case 7: return launch_gemv<7>(p, stream);
    default: return launch_gemv<8>(p, stream);
  }
}

// ------------------------------------------------------------------------------------------------------------
// RoPE + cache append for the new token of every sequence.
//   qkv [B, 3*H*Dh] (one row per sequence), cos/sin f32 [pos_rows, Dh], pos[b] = position of the new token,
//   q_out [B,H,Dh]; k/v cache [B,H,cap,Dh], written at index cache_len[b].
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rope_append_kernel(const T* qkv, const float* cos, const float* sin, const int* pos, const int* cache_len,
                                   T* q_out, T* k_cache, T* v_cache, int H, int Dh, int cap) {
  const int b = blockIdx.x, i = blockIdx.y * 256 + threadIdx.x;   // i over H*Dh
  if (i >= H * Dh) return;
  const int head = i / Dh, d = i - head * Dh, half = Dh / 2;
  const T* row = qkv + (size_t)b * 3 * H * Dh;
  const int ps = pos[b];
  const float c = cos[(size_t)ps * Dh + d], s = sin[(size_t)ps * Dh + d];
  const float q = (float)row[i], k = (float)row[H * Dh + i], v = (float)row[2 * H * Dh + i];
  const float qp = d < half ? -(float)row[i + half] : (float)row[i - half];
  const float kp = d < half ? -(float)row[H * Dh + i + half] : (float)row[H * Dh + i - half];
  q_out[(size_t)b * H * Dh + i] = (T)(q * c + qp * s);
  const size_t at = ((size_t)(b * H + head) * cap + cache_len[b]) * Dh + d;
  k_cache[at] = (T)(k * c + kp * s);
  v_cache[at] = (T)v;
}

// ------------------------------------------------------------------------------------------------------------
// Single-query attention over the cache.  One 256-thread block per (batch, head); lane t handles keys t, t+256, ...
// n_keys[b] = number of cached keys to attend to (including the token just appended);
// valid bits (optional, [B][nwords]) mask padded columns of the prefill.
// ------------------------------------------------------------------------------------------------------------
template <typename T, int DH>
__global__ __launch_bounds__(256) void decode_attn_kernel(const T* q, const T* kc, const T* vc, T* o, const int* n_keys,
                                                          const uint64_t* vbits, int nwords, int H, int cap, float scale) {
  __shared__ float s_m[256], s_l[256];
  __shared__ float s_acc[4][DH];
  const int bh = blockIdx.x, b = bh / H, tid = threadIdx.x;
  const int n = n_keys[b];
  float qv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) qv[d] = (float)q[(size_t)bh * DH + d] * scale;
  float m = -INFINITY, l = 0.f, acc[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) acc[d] = 0.f;
  const T* kb = kc + (size_t)bh * cap * DH;
  const T* vb = vc + (size_t)bh * cap * DH;
  for (int t = tid; t < n; t += 256) {
    if (vbits && (t >> 6) < nwords && !((vbits[(size_t)b * nwords + (t >> 6)] >> (t & 63)) & 1ull)) continue;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) s = __builtin_fmaf(qv[d], (float)kb[(size_t)t * DH + d], s);
    const float mn = fmaxf(m, s);
    const float a = __expf(m - mn), pr = __expf(s - mn);
    l = l * a + pr;
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = __builtin_fmaf(pr, (float)vb[(size_t)t * DH + d], acc[d] * a);
    m = mn;
  }
  // merge the 256 partials: global max, rescale, sum
  s_m[tid] = m;
  __syncthreads();
  float gm = -INFINITY;
  for (int i = 0; i < 256; ++i) gm = fmaxf(gm, s_m[i]);
  const float f = (m == -INFINITY) ? 0.f : __expf(m - gm);
  l *= f;
#pragma unroll
  for (int d = 0; d < DH; ++d) acc[d] *= f;
  // wave-level reduction, then 4 waves through LDS
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) {
    l += __shfl_xor(l, ofs);
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] += __shfl_xor(acc[d], ofs);
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) {
    s_l[wave] = l;
#pragma unroll
    for (int d = 0; d < DH; ++d) s_acc[wave][d] = acc[d];
  }
  __syncthreads();
  if (tid < DH) {
    const float lt = s_l[0] + s_l[1] + s_l[2] + s_l[3];
    const float a = s_acc[0][tid] + s_acc[1][tid] + s_acc[2][tid] + s_acc[3][tid];
    o[(size_t)bh * DH + tid] = (T)(lt > 0.f ? a / lt : 0.f);
  }
}

int rope_append_launch(const void* qkv, const float* cos, const float* sin, const int* pos, const int* cache_len, void* q_out,
                       void* k_cache, void* v_cache, int B, int H, int Dh, int cap, int dtype, hipStream_t s) {
  const dim3 grid(B, (H * Dh + 255) / 256), block(256);
  AKI_CLEAR_ERR();
  if (dtype == AKI_DT_BF16)
    hipLaunchKernelGGL(rope_append_kernel<__bf16>, grid, block, 0, s, (const __bf16*)qkv, cos, sin, pos, cache_len, (__bf16*)q_out,
                       (__bf16*)k_cache, (__bf16*)v_cache, H, Dh, cap);
  else
    hipLaunchKernelGGL(rope_append_kernel<float>, grid, block, 0, s, (const float*)qkv, cos, sin, pos, cache_len, (float*)q_out,
                       (float*)k_cache, (float*)v_cache, H, Dh, cap);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

int decode_attn_launch(const void* q, const void* kc, const void* vc, void* o, const int* n_keys, const uint64_t* vbits, int nwords,
                       int B, int H, int Dh, int cap, float scale, int dtype, hipStream_t s) {
  if (Dh != 96) return AKI_ERR_UNSUPPORTED;
  const dim3 grid(B * H), block(256);
  AKI_CLEAR_ERR();
  if (dtype == AKI_DT_BF16)
    hipLaunchKernelGGL((decode_attn_kernel<__bf16, 96>), grid, block, 0, s, (const __bf16*)q, (const __bf16*)kc, (const __bf16*)vc,
                       (__bf16*)o, n_keys, vbits, nwords, H, cap, scale);
  else
    hipLaunchKernelGGL((decode_attn_kernel<float, 96>), grid, block, 0, s, (const float*)q, (const float*)kc, (const float*)vc,
                       (float*)o, n_keys, vbits, nwords, H, cap, scale);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
