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
#include <type_traits>

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
  const bf16_t* norm_w; float norm_eps;       // optional fused RMSNorm of the x rows (decode: the layer's pre-norm)
  int M, N, K, ldx, ldw, ldy, ldr, res_row_mod, act;
  const float* w_scale;                       // W8: w points at e4m3 bytes (ldw in bytes), one f32 scale per weight row
};

// U chunks (of 8 k) per lane for NR weight rows: every load is issued before the first dot product
template <int M, int NR, int U>
__device__ __forceinline__ void gemv_sweep(const bf16_t* const (&wr)[NR], const char* sx, int nchunk, int c, float (&acc)[NR][M]) {
  u32x4 w[U][NR];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < NR; ++r) w[u][r] = __builtin_nontemporal_load((const u32x4*)(wr[r] + (size_t)(c + 64 * u) * 8));
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const u32x4 xv = *(const u32x4*)(sx + ((size_t)m * nchunk + c + 64 * u) * 16);
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r][m] = dot8_bf16(w[u][r], xv, acc[r][m]);
    }
}

// The two halves of a sweep, for the FIRST sweep of a single-row launch: its loads go out before x is staged (weights do not depend on
// the activation - the staging's round trips and the fused RMSNorm then run under the first weight round trip instead of in front of it).
template <int NR, int U>
__device__ __forceinline__ void gemv_issue(const bf16_t* const (&wr)[NR], int c, u32x4 (&w)[U][NR]) {
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < NR; ++r) w[u][r] = __builtin_nontemporal_load((const u32x4*)(wr[r] + (size_t)(c + 64 * u) * 8));
}
template <int NR, int U>
__device__ __forceinline__ void gemv_consume(const u32x4 (&w)[U][NR], const char* sx, int c, float (&acc)[NR][1]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const u32x4 xv = *(const u32x4*)(sx + ((size_t)c + 64 * u) * 16);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r][0] = dot8_bf16(w[u][r], xv, acc[r][0]);
  }
}

// weight-only fp8 (e4m3 weights, bf16 activations): a 16-byte weight chunk holds 16 k-values and meets two 16-byte x chunks;
// v_cvt_scalef32_pk_bf16_fp8 (scale 1) turns two weights into a bf16 pair in one instruction (exact: e4m3 fits bf16) for the same dot2:
// 16 VALU operations per 16 weights (until round 5: v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32 + dot2 = 24, and the e4m3 decode was VALU-bound).
__device__ __forceinline__ float dot16_w8(const u32x4 w, const u32x4 x0, const u32x4 x1, float acc) {
  const bf16x8_t xa = __builtin_bit_cast(bf16x8_t, x0), xb = __builtin_bit_cast(bf16x8_t, x1);
#define AKI_W8_PAIR(word, hi, xv, i0)                                                                               \
  {                                                                                                                 \
    const bf16x2_t wb = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((unsigned)(word), 1.0f, hi);   /* two e4m3 -> a bf16 pair in ONE instruction, exact */ \
    acc = __builtin_amdgcn_fdot2_f32_bf16(wb, __builtin_shufflevector(xv, xv, i0, i0 + 1), acc, false);             \
  }
  AKI_W8_PAIR(w[0], false, xa, 0) AKI_W8_PAIR(w[0], true, xa, 2) AKI_W8_PAIR(w[1], false, xa, 4) AKI_W8_PAIR(w[1], true, xa, 6)
  AKI_W8_PAIR(w[2], false, xb, 0) AKI_W8_PAIR(w[2], true, xb, 2) AKI_W8_PAIR(w[3], false, xb, 4) AKI_W8_PAIR(w[3], true, xb, 6)
#undef AKI_W8_PAIR
  return acc;
}

template <int NR, int U>
__device__ __forceinline__ void gemv_issue_w8(const uint8_t* const (&wr)[NR], int c, u32x4 (&w)[U][NR]) {
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < NR; ++r) w[u][r] = __builtin_nontemporal_load((const u32x4*)(wr[r] + (size_t)(c + 64 * u) * 16));
}
template <int NR, int U>
__device__ __forceinline__ void gemv_consume_w8(const u32x4 (&w)[U][NR], const char* sx, int c, float (&acc)[NR][1]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const u32x4 x0 = *(const u32x4*)(sx + ((size_t)2 * (c + 64 * u)) * 16);
    const u32x4 x1 = *(const u32x4*)(sx + ((size_t)2 * (c + 64 * u) + 1) * 16);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r][0] = dot16_w8(w[u][r], x0, x1, acc[r][0]);
  }
}

template <int M, int NR, int U>
__device__ __forceinline__ void gemv_sweep_w8(const uint8_t* const (&wr)[NR], const char* sx, int nchunk_x, int c, float (&acc)[NR][M]) {
  u32x4 w[U][NR];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < NR; ++r) w[u][r] = __builtin_nontemporal_load((const u32x4*)(wr[r] + (size_t)(c + 64 * u) * 16));
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const u32x4 x0 = *(const u32x4*)(sx + ((size_t)m * nchunk_x + 2 * (c + 64 * u)) * 16);
      const u32x4 x1 = *(const u32x4*)(sx + ((size_t)m * nchunk_x + 2 * (c + 64 * u) + 1) * 16);
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r][m] = dot16_w8(w[u][r], x0, x1, acc[r][m]);
    }
}

// FPW output features per wave; SWIGLU: feature f pairs weight rows f (gate) and N/2 + f (up).
// The K sweep is unrolled KU chunks deep with all weight loads issued before the dot products: a wave keeps
// NR*KU 16-byte loads in flight per lane, which is what decides the streaming rate at 4-6 waves per CU.
template <int M, bool SWIGLU, int FPW = 2, bool W8 = false>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const GemvParams p) {
  constexpr int NR = SWIGLU ? 2 * FPW : FPW;   // weight rows per wave
  constexpr int KU = NR <= 2 ? 8 : 4;          // up to 16 sixteen-byte loads in flight per lane
  extern __shared__ __attribute__((aligned(16))) char sx[];
  __shared__ float s_red[M][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = p.K / 8;
  // ---- single row: the first sweep of this workgroup's first feature group is requested BEFORE x is staged ------------------------
  constexpr int PU = W8 ? 2 : 4;                 // chunks per lane and row of that sweep (the ladder below continues behind it)
  u32x4 wpre[PU][NR];
  bool pre = false;
  if constexpr (M == 1 && !W8) {     // e4m3 weights: measured 2.3 % SLOWER with the early sweep (1.488 vs 1.454 ms per token, one box) - not used there
    const int n_out0 = SWIGLU ? p.N / 2 : p.N;
    const int f00 = (blockIdx.x * 4 + wave) * FPW;
    pre = f00 < n_out0 && (W8 ? p.K / 16 : nchunk) >= 64 * PU;
    if (pre) {
      if constexpr (W8) {
        const uint8_t* wr0[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int f = min(f00 + (r % FPW), n_out0 - 1);
          wr0[r] = (const uint8_t*)p.w + (size_t)((SWIGLU && r >= FPW) ? n_out0 + f : f) * p.ldw;
        }
        gemv_issue_w8<NR, PU>(wr0, lane, wpre);
      } else {
        const bf16_t* wr0[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int f = min(f00 + (r % FPW), n_out0 - 1);
          wr0[r] = p.w + (size_t)((SWIGLU && r >= FPW) ? n_out0 + f : f) * p.ldw;
        }
        gemv_issue<NR, PU>(wr0, lane, wpre);
      }
    }
  }
  if (p.norm_w == nullptr) {
    for (int i = tid; i < M * nchunk; i += 256) {
      const int m = i / nchunk, c = i - m * nchunk;
      *(u32x4*)(sx + (size_t)i * 16) = *(const u32x4*)(p.x + (size_t)m * p.ldx + c * 8);
    }
  } else {
    // y = bf16(x * rsqrt(mean(x^2) + eps) * w): same rounding points as norm_bf16_kernel<true> (aux_kernels.hip)
    float ss[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      ss[m] = 0.f;
      for (int c = tid; c < nchunk; c += 256) {
        const u32x4 v = *(const u32x4*)(p.x + (size_t)m * p.ldx + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bf16_lo(v[e]), hi = bf16_hi(v[e]);
          ss[m] = __builtin_fmaf(lo, lo, ss[m]);
          ss[m] = __builtin_fmaf(hi, hi, ss[m]);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss[m] += __shfl_xor(ss[m], o);
      if (lane == 0) s_red[m][wave] = ss[m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const float r = rsqrtf((s_red[m][0] + s_red[m][1] + s_red[m][2] + s_red[m][3]) / (float)p.K + p.norm_eps);
      for (int c = tid; c < nchunk; c += 256) {
        const u32x4 v = *(const u32x4*)(p.x + (size_t)m * p.ldx + c * 8);
        const u32x4 g = *(const u32x4*)(p.norm_w + c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // HF Phi3RMSNorm: weight * (x * rstd).to(bf16)
          o[e] = pack_bf16x2(round_bf16(bf16_lo(v[e]) * r) * bf16_lo(g[e]), round_bf16(bf16_hi(v[e]) * r) * bf16_hi(g[e]));
        }
        *(u32x4*)(sx + ((size_t)m * nchunk + c) * 16) = o;
      }
    }
  }
  __syncthreads();
  const int n_out = SWIGLU ? p.N / 2 : p.N;
  // a workgroup owns feature groups blockIdx.x, blockIdx.x + gridDim.x, ...: the x rows staged above are reused
  for (int grp = blockIdx.x; grp * (4 * FPW) < n_out; grp += gridDim.x) {
    const int f0 = (grp * 4 + wave) * FPW;
    if (f0 >= n_out) break;
    float acc[NR][M];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
    float wsc[NR];
    if constexpr (W8) {
      const uint8_t* wr[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int f = min(f0 + (r % FPW), n_out - 1);
        const int row = (SWIGLU && r >= FPW) ? n_out + f : f;
        wr[r] = (const uint8_t*)p.w + (size_t)row * p.ldw;
        wsc[r] = p.w_scale[row];
      }
      const int nchunk_w = p.K / 16;                         // 16-byte weight chunks = 16 k-values each
      int c = lane;
      if constexpr (M == 1) {
        if (pre) { gemv_consume_w8<NR, PU>(wpre, sx, c, acc); c += 64 * PU; }
      }
      for (; c + 64 * 3 < nchunk_w; c += 64 * 4) gemv_sweep_w8<M, NR, 4>(wr, sx, nchunk, c, acc);
      for (; c + 64 < nchunk_w; c += 128) gemv_sweep_w8<M, NR, 2>(wr, sx, nchunk, c, acc);
      for (; c < nchunk_w; c += 64) gemv_sweep_w8<M, NR, 1>(wr, sx, nchunk, c, acc);
    } else {
      const bf16_t* wr[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int f = min(f0 + (r % FPW), n_out - 1);
        wr[r] = p.w + (size_t)((SWIGLU && r >= FPW) ? n_out + f : f) * p.ldw;
        wsc[r] = 1.f;
      }
      int c = lane;
      if constexpr (M == 1) {
        if (pre) { gemv_consume<NR, PU>(wpre, sx, c, acc); c += 64 * PU; }
      }
      for (; c + 64 * (KU - 1) < nchunk; c += 64 * KU) gemv_sweep<M, NR, KU>(wr, sx, nchunk, c, acc);
      if constexpr (KU > 4) {
        for (; c + 64 * 3 < nchunk; c += 64 * 4) gemv_sweep<M, NR, 4>(wr, sx, nchunk, c, acc);
      }
      for (; c + 64 < nchunk; c += 128) gemv_sweep<M, NR, 2>(wr, sx, nchunk, c, acc);
      for (; c < nchunk; c += 64) gemv_sweep<M, NR, 1>(wr, sx, nchunk, c, acc);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float v = acc[r][m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        acc[r][m] = W8 ? v * wsc[r] : v;
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
    if constexpr (M == 1) break;      // single row: the launcher gives every workgroup exactly one feature group (and the pre-issued sweep is its)
  }
}

template <int M, bool SWIGLU, int FPW, bool W8 = false>
static int launch_gemv_cfg(const GemvParams& p, int n_out, hipStream_t stream) {
  const size_t smem = (size_t)M * p.K * 2;
  // one group = 4 waves x FPW features.  Staging x costs M*K*2 bytes per workgroup against 4*FPW*K*2 bytes of weights per
  // group, so for M > 1 a workgroup takes several groups (at most ~512 workgroups stay in flight).
  const int groups = (n_out + 4 * FPW - 1) / (4 * FPW);
  const int per = M == 1 ? 1 : (groups + 511) / 512;
  const dim3 grid((groups + per - 1) / per), block(256);
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute((const void*)gemv_bf16_kernel<M, SWIGLU, FPW, W8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192 * 2) != hipSuccess)
      return AKI_ERR_LAUNCH;
    set = true;
  }
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL((gemv_bf16_kernel<M, SWIGLU, FPW, W8>), grid, block, smem, stream, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

template <int M>
static int launch_gemv(const GemvParams& p, hipStream_t stream) {
  const int n_out = p.act == AKI_ACT_SWIGLU ? p.N / 2 : p.N;
  if (p.w_scale) {                       // weight-only fp8: single-sequence decode in the fp8 configuration
    if constexpr (M == 1) {
      if (p.act == AKI_ACT_SWIGLU) return launch_gemv_cfg<1, true, 2, true>(p, n_out, stream);
      return launch_gemv_cfg<1, false, 2, true>(p, n_out, stream);
    } else {
      return AKI_ERR_UNSUPPORTED;
    }
  }
  if (p.act == AKI_ACT_SWIGLU) return launch_gemv_cfg<M, true, 2>(p, n_out, stream);
  // wide outputs (qkv, lm_head) have waves to spare: 4 features per wave doubles the loads each wave keeps in flight
  if (M <= 2 && n_out >= 8192) return launch_gemv_cfg<M, false, 4>(p, n_out, stream);
  return launch_gemv_cfg<M, false, 2>(p, n_out, stream);
}

// ------------------------------------------------------------------------------------------------------------
// Skinny MFMA GEMM for 2 <= M <= 16 rows (batched decode): y[M][N] = act(x W^T + b) [+ residual].
// The dot-product GEMV above spends M FMAs and M LDS reads per weight element and stops being HBM-bound beyond M ~ 2;
// here the tokens ride on the 16 columns of v_mfma_f32_16x16x32_bf16 instead.  One workgroup = one 16-feature tile,
// its KS waves split K; every lane streams 32 contiguous bytes of ITS weight row per step (lane = (row l15, k-group kg):
// the four k-groups of a row read one full 128-byte line) straight from HBM into registers - no LDS staging, 16 loads in
// flight per lane - and the matching 32 bytes of x come from L2 (x is M*K*2 bytes, read by every tile).  The two MFMAs of
// a step take the first / second 16 bytes of both operands (any k order is fine as long as A and B agree).  Partial tiles
// of the KS waves meet in LDS; wave 0 runs the epilogue.  SWIGLU: the wave carries the gate tile and the up tile.
// ------------------------------------------------------------------------------------------------------------
// NORM (M <= 8, the decode step's pre-norms): the x rows are RMS-normalised on their way into LDS - wave w takes rows w, w + KS, ... whole, with
// the rounding points of norm_bf16_kernel<true> (HF Phi3RMSNorm: weight * (x * rstd).to(bf16)) - and the B fragments come from there instead of
// L2; the weight loads of the first steps are requested BEFORE that prologue, so it runs under their round trip.  It replaces a 4.9 us norm launch
// in front of the qkv and gate_up GEMMs of every layer of a batched decode step (65 of 231 launches, 9 % of the step at batch 8).
template <int KS, bool SWIGLU, int FT, bool NORM = false>
__global__ __launch_bounds__(KS * 64) void skinny_gemm_bf16_kernel(const GemvParams p) {
  constexpr int NS = FT * (SWIGLU ? 2 : 1);          // weight streams per wave, all fed by one x fragment
  constexpr int UN = NS >= 4 ? 2 : (NS == 2 ? 4 : 8);   // steps of 64 k whose loads are issued together (<= 20 loads in flight)
  __shared__ float red_st[NORM ? 1 : KS][NS][256];
  extern __shared__ __attribute__((aligned(16))) char s_xn[];      // NORM: the normalised rows, (K * 2 + 16) bytes apart (the pad spreads the 16 rows over the banks)
  // NORM: the partial tiles meet in the rows' LDS once every wave is done with them - 49 KB per workgroup instead of 57: three workgroups per CU,
  // and the 576 of the qkv GEMM are resident at once (at two per CU the last 64 were a second round: + 3 us)
  float (*red)[NS][256] = NORM ? (float (*)[NS][256])s_xn : red_st;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
  const int n_out = SWIGLU ? p.N / 2 : p.N;
  const int f0 = blockIdx.x * 16 * FT;
  const int Kw = p.K / KS, kbeg = wave * Kw;
  const bf16_t* wp[NS];
#pragma unroll
  for (int t = 0; t < FT; ++t) {
    const int frow = min(f0 + 16 * t + l15, n_out - 1);
    wp[t] = p.w + (size_t)frow * p.ldw + kbeg + 16 * kg;
    if (SWIGLU) wp[FT + t] = p.w + (size_t)(n_out + frow) * p.ldw + kbeg + 16 * kg;
  }
  const int xrow = min(l15, p.M - 1);
  const bf16_t* xr = p.x + (size_t)xrow * p.ldx + kbeg + 16 * kg;
  const int xs_pitch = p.K * 2 + 16;
  const char* xs = s_xn + (size_t)xrow * xs_pitch + (size_t)(kbeg + 16 * kg) * 2;
  f32x4 acc[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = Kw / 64;
  int it = 0;
  auto load_w = [&](int step, u32x4 (&w2)[NS][2]) {
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      w2[t][0] = __builtin_nontemporal_load((const u32x4*)(wp[t] + (size_t)step * 64));
      w2[t][1] = __builtin_nontemporal_load((const u32x4*)(wp[t] + (size_t)step * 64 + 8));
    }
  };
  auto load_x = [&](int step, u32x4 (&x2)[2]) {
    if constexpr (NORM) {
      x2[0] = *(const u32x4*)(xs + (size_t)step * 128);
      x2[1] = *(const u32x4*)(xs + (size_t)step * 128 + 16);
    } else {
      x2[0] = *(const u32x4*)(xr + (size_t)step * 64);
      x2[1] = *(const u32x4*)(xr + (size_t)step * 64 + 8);
    }
  };
  auto mma_step = [&](const u32x4 (&w2)[NS][2], const u32x4 (&x2)[2]) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const bf16x8 xb = __builtin_bit_cast(bf16x8, x2[hh]);
#pragma unroll
      for (int t = 0; t < NS; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2[t][hh]), xb, acc[t], 0, 0, 0);
    }
  };
  if constexpr (NORM) {
    constexpr int UP = NS >= 2 ? 3 : 4;               // steps whose weights are requested before the prologue (more cost registers: the 576 workgroups of qkv want three per CU = 84 VGPRs)
    u32x4 wpre[UP][NS][2];
    const bool pre = nsteps >= UP;
    if (pre) {
#pragma unroll
      for (int u = 0; u < UP; ++u) load_w(u, wpre[u]);
    }
    const int nchunk = p.K / 8;
    for (int m = wave; m < p.M; m += KS) {             // a wave normalises whole rows: no cross-wave reduction
      const bf16_t* xm = p.x + (size_t)m * p.ldx;
      char* dst = s_xn + (size_t)m * xs_pitch;
      float ss = 0.f;
      for (int c = lane; c < nchunk; c += 64) {
        const u32x4 v = *(const u32x4*)(xm + (size_t)c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bf16_lo(v[e]), hi = bf16_hi(v[e]);
          ss = __builtin_fmaf(lo, lo, ss);
          ss = __builtin_fmaf(hi, hi, ss);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
      const float r = rsqrtf(ss / (float)p.K + p.norm_eps);
      for (int c = lane; c < nchunk; c += 64) {
        const u32x4 v = *(const u32x4*)(xm + (size_t)c * 8);            // second read: an L1 / L2 hit
        const u32x4 g = *(const u32x4*)(p.norm_w + (size_t)c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_bf16x2(round_bf16(bf16_lo(v[e]) * r) * bf16_lo(g[e]), round_bf16(bf16_hi(v[e]) * r) * bf16_hi(g[e]));
        *(u32x4*)(dst + (size_t)c * 16) = o;
      }
    }
    __syncthreads();
    if (pre) {
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        u32x4 x2[2];
        load_x(u, x2);
        mma_step(wpre[u], x2);
      }
      it = UP;
    }
  }
  auto run = [&](auto un_c) {
    constexpr int U = decltype(un_c)::value;
    for (; it + U <= nsteps; it += U) {
      u32x4 wa[U][NS][2], xa[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        load_w(it + u, wa[u]);
        load_x(it + u, xa[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) mma_step(wa[u], xa[u]);
    }
  };
  run(std::integral_constant<int, UN>{});          // ladder: a wave's K slice can be shorter than the deepest unroll
  if constexpr (UN > 4) run(std::integral_constant<int, 4>{});
  if constexpr (UN > 2) run(std::integral_constant<int, 2>{});
  run(std::integral_constant<int, 1>{});
  // accumulator: lane (token = l15, features 4kg..4kg+3 of each tile); fold the KS partial tiles
  if (KS > 1) {
    if constexpr (NORM) __syncthreads();             // every wave has read its last fragment of the rows
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][t][lane * 4 + r] = acc[t][r];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < KS; ++w)
#pragma unroll
      for (int t = 0; t < NS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += red[w][t][lane * 4 + r];
  }
  const int tok = l15;
  if (tok >= p.M) return;
#pragma unroll
  for (int t = 0; t < FT; ++t) {
    const int f = f0 + 16 * t + 4 * kg;
    if (f >= n_out) continue;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (SWIGLU) {
        v[r] = acc[FT + t][r] * silu_fast(acc[t][r]);
      } else {
        v[r] = acc[t][r];
        if (p.bias) v[r] += bf16_bits_to_f32(p.bias[f + r]);
        if (p.act == AKI_ACT_GELU_ERF) v[r] = gelu_erf_fast(v[r]);
        else if (p.act == AKI_ACT_GELU_TANH) v[r] = gelu_tanh_fast(v[r]);
      }
      if (p.residual) v[r] += bf16_bits_to_f32(p.residual[(size_t)(p.res_row_mod > 0 ? tok % p.res_row_mod : tok) * p.ldr + f + r]);
    }
    *(u32x2*)(p.y + (size_t)tok * p.ldy + f) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
}

template <int KS, int FT, bool NORM = false>
static int launch_skinny(const GemvParams& p, hipStream_t stream) {
  const int n_out = p.act == AKI_ACT_SWIGLU ? p.N / 2 : p.N;
  const dim3 grid((n_out + 16 * FT - 1) / (16 * FT)), block(KS * 64);
  constexpr size_t RED = (size_t)KS * 2 * 1024;      // the partial tiles (two streams with SwiGLU) alias the rows
  const size_t rows = (size_t)p.M * ((size_t)p.K * 2 + 16);
  const size_t smem = NORM ? (rows > RED ? rows : RED) : 0;
  if constexpr (NORM) {
    static bool set_s = false, set_p = false;
    bool& set = p.act == AKI_ACT_SWIGLU ? set_s : set_p;
    if (!set) {
      const void* fn = p.act == AKI_ACT_SWIGLU ? (const void*)skinny_gemm_bf16_kernel<KS, true, FT, true> : (const void*)skinny_gemm_bf16_kernel<KS, false, FT, true>;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * (8192 * 2 + 16)) != hipSuccess) return AKI_ERR_LAUNCH;
      set = true;
    }
  }
  AKI_CLEAR_ERR();
  if (p.act == AKI_ACT_SWIGLU) hipLaunchKernelGGL((skinny_gemm_bf16_kernel<KS, true, FT, NORM>), grid, block, smem, stream, p);
  else hipLaunchKernelGGL((skinny_gemm_bf16_kernel<KS, false, FT, NORM>), grid, block, smem, stream, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// 2 <= M <= 16; n_out, ldy, ldr multiples of 4; K a multiple of 64 (per wave).  AKI_ERR_UNSUPPORTED otherwise.
// rms_w != NULL (M <= 8, K <= 8192): the x rows are RMS-normalised (weight rms_w [K], eps) inside the launch.
int skinny_gemm_bf16(const aki_linear_args* a, const void* rms_w, float eps, hipStream_t stream) {
  const int n_out = a->act == AKI_ACT_SWIGLU ? a->N / 2 : a->N;
  if (a->M < 2 || a->M > 16 || a->K % 64 || (a->ldx % 8) || (a->ldw % 8) || (n_out % 4) || (a->ldy % 4) || (a->residual && (a->ldr % 4)))
    return AKI_ERR_UNSUPPORTED;
  if (a->act == AKI_ACT_SWIGLU && (a->bias || (a->N & 1))) return AKI_ERR_UNSUPPORTED;
  if (rms_w && (a->M > 8 || a->K > 8192)) return AKI_ERR_UNSUPPORTED;
  if (((uintptr_t)a->x & 15) || ((uintptr_t)a->w & 15) || ((uintptr_t)a->y & 7) || ((uintptr_t)a->bias & 7) || ((uintptr_t)rms_w & 15)) return AKI_ERR_ALIGNMENT;
  GemvParams p = {(const bf16_t*)a->x, (const bf16_t*)a->w, (const bf16_t*)a->bias, (const bf16_t*)a->residual, (bf16_t*)a->y,
                  (const bf16_t*)rms_w, eps, a->M, a->N, a->K, a->ldx, a->ldw, a->ldy, a->ldr, a->res_row_mod, a->act, nullptr};
  // One 16-feature tile per wave and a K split that keeps >= ~4 waves per CU.  (Two tiles per wave sharing the x fragment
  // were measured: fewer x loads, but the lost wave parallelism cost more - 3.85 vs 3.30 ms per step at batch 8.)
  const int tiles = (n_out + 15) / 16;
  if (rms_w) {
    // every workgroup normalises the M rows for itself: fine for the 512-1024 eight- or four-wave workgroups of qkv / gate_up, not for the 2004
    // two-wave ones of the lm_head (measured + 29 us at M = 8): wide outputs keep the norm launch (AKI_ERR_UNSUPPORTED: the caller's choice)
    if (tiles < 768 && a->K % 512 == 0) return launch_skinny<8, 1, true>(p, stream);
    if (tiles < 1536 && a->K % 256 == 0) return launch_skinny<4, 1, true>(p, stream);
    return AKI_ERR_UNSUPPORTED;
  }
  if (tiles < 768 && a->K % 512 == 0) return launch_skinny<8, 1>(p, stream);
  if (tiles < 1536 && a->K % 256 == 0) return launch_skinny<4, 1>(p, stream);
  if (a->K % 128 == 0) return launch_skinny<2, 1>(p, stream);
  return launch_skinny<1, 1>(p, stream);
}

// ------------------------------------------------------------------------------------------------------------
// The skinny GEMM on e4m3 weights (W8A16: one f32 scale per weight row, bf16 rows) for 2 <= M <= 16 - batched decode in the fp8
// configuration streams half the bytes.  Same tile and K split; a step is 128 k: lane (row l15, k-group kg) loads 32 bytes = 32 k of ITS
// weight row (the four k-groups read one 128-byte line), widens them pairwise to bf16 (v_cvt_scalef32_pk_bf16_fp8, exact) and feeds four
// MFMAs against the 64 bytes of x that carry the same k.  The row scale multiplies the finished sum, as in the one-row GEMV (gemv_bf16_kernel<W8>).
// NORM: as in skinny_gemm_bf16_kernel.
// ------------------------------------------------------------------------------------------------------------
template <int KS, bool SWIGLU, bool NORM>
__global__ __launch_bounds__(KS * 64) void skinny_gemm_w8_kernel(const GemvParams p) {
  constexpr int NS = SWIGLU ? 2 : 1;
  constexpr int UN = SWIGLU ? 2 : 4;                 // steps of 128 k whose loads are issued together
  __shared__ float red_st[NORM ? 1 : KS][NS][256];
  extern __shared__ __attribute__((aligned(16))) char s_xn[];
  float (*red)[NS][256] = NORM ? (float (*)[NS][256])s_xn : red_st;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
  const int n_out = SWIGLU ? p.N / 2 : p.N;
  const int f0 = blockIdx.x * 16;
  const int Kw = p.K / KS, kbeg = wave * Kw;
  const int frow = min(f0 + l15, n_out - 1);
  const uint8_t* wp[NS];
  wp[0] = (const uint8_t*)p.w + (size_t)frow * p.ldw + kbeg + 32 * kg;
  if (SWIGLU) wp[NS - 1] = (const uint8_t*)p.w + (size_t)(n_out + frow) * p.ldw + kbeg + 32 * kg;
  const int xrow = min(l15, p.M - 1);
  const bf16_t* xr = p.x + (size_t)xrow * p.ldx + kbeg + 32 * kg;
  const int xs_pitch = p.K * 2 + 16;
  const char* xs = s_xn + (size_t)xrow * xs_pitch + (size_t)(kbeg + 32 * kg) * 2;
  f32x4 acc[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = Kw / 128;
  if constexpr (NORM) {
    const int nchunk = p.K / 8;
    for (int m = wave; m < p.M; m += KS) {
      const bf16_t* xm = p.x + (size_t)m * p.ldx;
      char* dst = s_xn + (size_t)m * xs_pitch;
      float ss = 0.f;
      for (int c = lane; c < nchunk; c += 64) {
        const u32x4 v = *(const u32x4*)(xm + (size_t)c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bf16_lo(v[e]), hi = bf16_hi(v[e]);
          ss = __builtin_fmaf(lo, lo, ss);
          ss = __builtin_fmaf(hi, hi, ss);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
      const float r = rsqrtf(ss / (float)p.K + p.norm_eps);
      for (int c = lane; c < nchunk; c += 64) {
        const u32x4 v = *(const u32x4*)(xm + (size_t)c * 8);
        const u32x4 g = *(const u32x4*)(p.norm_w + (size_t)c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_bf16x2(round_bf16(bf16_lo(v[e]) * r) * bf16_lo(g[e]), round_bf16(bf16_hi(v[e]) * r) * bf16_hi(g[e]));
        *(u32x4*)(dst + (size_t)c * 16) = o;
      }
    }
    __syncthreads();
  }
  int it = 0;
  auto run = [&](auto un_c) {
    constexpr int U = decltype(un_c)::value;
    for (; it + U <= nsteps; it += U) {
      u32x4 wa[U][NS][2], xa[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
          wa[u][t][0] = __builtin_nontemporal_load((const u32x4*)(wp[t] + (size_t)(it + u) * 128));
          wa[u][t][1] = __builtin_nontemporal_load((const u32x4*)(wp[t] + (size_t)(it + u) * 128 + 16));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          xa[u][j] = NORM ? *(const u32x4*)(xs + (size_t)(it + u) * 256 + 16 * j) : *(const u32x4*)(xr + (size_t)(it + u) * 128 + 8 * j);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {                      // k-values 8j .. 8j+7 of the lane's 32: weight bytes 8j .. 8j+7 = dwords 2j, 2j+1 of the 32 bytes
          const bf16x8 xb = __builtin_bit_cast(bf16x8, xa[u][j]);
#pragma unroll
          for (int t = 0; t < NS; ++t) {
            const unsigned d0 = wa[u][t][j >> 1][(j & 1) * 2], d1 = wa[u][t][j >> 1][(j & 1) * 2 + 1];
            const bf16x2_t p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, false), p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, true);
            const bf16x2_t p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, false), p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, true);
            const u32x4 wq = u32x4{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), __builtin_bit_cast(unsigned, p2), __builtin_bit_cast(unsigned, p3)};
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq), xb, acc[t], 0, 0, 0);
          }
        }
    }
  };
  run(std::integral_constant<int, UN>{});
  if constexpr (UN > 2) run(std::integral_constant<int, 2>{});
  run(std::integral_constant<int, 1>{});
  if (KS > 1) {
    if constexpr (NORM) __syncthreads();
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][t][lane * 4 + r] = acc[t][r];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < KS; ++w)
#pragma unroll
      for (int t = 0; t < NS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += red[w][t][lane * 4 + r];
  }
  const int tok = l15, f = f0 + 4 * kg;
  if (tok >= p.M || f >= n_out) return;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (SWIGLU) {
      v[r] = (acc[NS - 1][r] * p.w_scale[n_out + f + r]) * silu_fast(acc[0][r] * p.w_scale[f + r]);
    } else {
      v[r] = acc[0][r] * p.w_scale[f + r];
      if (p.bias) v[r] += bf16_bits_to_f32(p.bias[f + r]);
      if (p.act == AKI_ACT_GELU_ERF) v[r] = gelu_erf_fast(v[r]);
      else if (p.act == AKI_ACT_GELU_TANH) v[r] = gelu_tanh_fast(v[r]);
    }
    if (p.residual) v[r] += bf16_bits_to_f32(p.residual[(size_t)(p.res_row_mod > 0 ? tok % p.res_row_mod : tok) * p.ldr + f + r]);
  }
  *(u32x2*)(p.y + (size_t)tok * p.ldy + f) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

template <int KS, bool NORM>
static int launch_skinny_w8(const GemvParams& p, hipStream_t stream) {
  const int n_out = p.act == AKI_ACT_SWIGLU ? p.N / 2 : p.N;
  const dim3 grid((n_out + 15) / 16), block(KS * 64);
  constexpr size_t RED = (size_t)KS * 2 * 1024;
  const size_t rows = (size_t)p.M * ((size_t)p.K * 2 + 16);
  const size_t smem = NORM ? (rows > RED ? rows : RED) : 0;
  if constexpr (NORM) {
    static bool set_s = false, set_p = false;
    bool& set = p.act == AKI_ACT_SWIGLU ? set_s : set_p;
    if (!set) {
      const void* fn = p.act == AKI_ACT_SWIGLU ? (const void*)skinny_gemm_w8_kernel<KS, true, true> : (const void*)skinny_gemm_w8_kernel<KS, false, true>;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * (8192 * 2 + 16)) != hipSuccess) return AKI_ERR_LAUNCH;
      set = true;
    }
  }
  AKI_CLEAR_ERR();
  if (p.act == AKI_ACT_SWIGLU) hipLaunchKernelGGL((skinny_gemm_w8_kernel<KS, true, NORM>), grid, block, smem, stream, p);
  else hipLaunchKernelGGL((skinny_gemm_w8_kernel<KS, false, NORM>), grid, block, smem, stream, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// 2 <= M <= 16 rows on e4m3 weights; K a multiple of 128 per wave, rows of w 16-byte aligned.  AKI_ERR_UNSUPPORTED otherwise.
int skinny_gemm_w8(const aki_linear_args* a, const void* rms_w, float eps, hipStream_t stream) {
  const int n_out = a->act == AKI_ACT_SWIGLU ? a->N / 2 : a->N;
  if (a->M < 2 || a->M > 16 || !a->w_scale || (a->ldx % 8) || (a->ldw % 16) || (n_out % 4) || (a->ldy % 4) || (a->residual && (a->ldr % 4)))
    return AKI_ERR_UNSUPPORTED;
  if (a->act == AKI_ACT_SWIGLU && (a->bias || (a->N & 1))) return AKI_ERR_UNSUPPORTED;
  if (rms_w && (a->M > 8 || a->K > 8192)) return AKI_ERR_UNSUPPORTED;
  if (((uintptr_t)a->x & 15) || ((uintptr_t)a->w & 15) || ((uintptr_t)a->y & 7) || ((uintptr_t)a->bias & 7) || ((uintptr_t)rms_w & 15)) return AKI_ERR_ALIGNMENT;
  GemvParams p = {(const bf16_t*)a->x, (const bf16_t*)a->w, (const bf16_t*)a->bias, (const bf16_t*)a->residual, (bf16_t*)a->y,
                  (const bf16_t*)rms_w, eps, a->M, a->N, a->K, a->ldx, a->ldw, a->ldy, a->ldr, a->res_row_mod, a->act, a->w_scale};
  const int tiles = (n_out + 15) / 16;
  // waves per tile: at least four steps of 128 k per wave, and enough waves per CU for the narrow outputs
  const bool k8 = a->K % 1024 == 0 && a->K / 8 >= 512, k4 = a->K % 512 == 0;
  if (rms_w) {
    if (tiles >= 1536) return AKI_ERR_UNSUPPORTED;      // lm_head-wide outputs keep the norm launch (see skinny_gemm_bf16)
    if (k8 && tiles < 768) return launch_skinny_w8<8, true>(p, stream);
    if (k4) return launch_skinny_w8<4, true>(p, stream);
    return AKI_ERR_UNSUPPORTED;
  }
  if (k8 && tiles < 768) return launch_skinny_w8<8, false>(p, stream);
  if (k4 && tiles < 1536) return launch_skinny_w8<4, false>(p, stream);
  if (a->K % 256 == 0) return launch_skinny_w8<2, false>(p, stream);
  return AKI_ERR_UNSUPPORTED;
}

// M <= 8 rows and M*K*2 <= 128 KiB of LDS; returns AKI_ERR_UNSUPPORTED otherwise (the caller then uses the MFMA GEMM).
// rms_w != NULL: the x rows are RMS-normalised (weight rms_w [K], eps) on the way into LDS.
int gemv_bf16(const aki_linear_args* a, const void* rms_w, float eps, hipStream_t stream) {
  const bool w8 = a->dtype == AKI_DT_W8A16;
  if (a->M > 8 || a->K % (w8 ? 16 : 8) || (size_t)a->M * a->K * 2 > 8 * 8192 * 2 || (a->ldx % 8) || (a->ldw % (w8 ? 16 : 8))) return AKI_ERR_UNSUPPORTED;
  if (a->act == AKI_ACT_SWIGLU && (a->bias || (a->N & 1))) return AKI_ERR_UNSUPPORTED;
  if (w8 && (!a->w_scale || a->M != 1)) return AKI_ERR_UNSUPPORTED;
  GemvParams p = {(const bf16_t*)a->x, (const bf16_t*)a->w, (const bf16_t*)a->bias, (const bf16_t*)a->residual, (bf16_t*)a->y,
                  (const bf16_t*)rms_w, eps, a->M, a->N, a->K, a->ldx, a->ldw, a->ldy, a->ldr, a->res_row_mod, a->act,
                  w8 ? a->w_scale : nullptr};
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
    float t[8] = {s_l[0], s_l[1], s_l[2], s_l[3], s_acc[0][tid], s_acc[1][tid], s_acc[2][tid], s_acc[3][tid]};
    lds_fold_ready(t);
    const float lt = t[0] + t[1] + t[2] + t[3];
    const float a = t[4] + t[5] + t[6] + t[7];
    o[(size_t)bh * DH + tid] = (T)(lt > 0.f ? a / lt : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Split-KV single-query attention (bf16, Dh = 96), optionally fused with RoPE + cache append of the new token.
//
// With one query per (batch, head) there are only B*H independent rows, far too few to pull the cache at HBM rate
// from 256 CUs, so the keys of each row are split over S single-wave workgroups of T 64-key tiles each:
//   score phase  lane = key: the lane loads its whole 192-byte K row (12 x 16 B, all in flight) and dots it with q
//   PV phase     lane = (row group g = lane>>4, 16-byte column chunk i = lane&15 < 12): 16 loads cover the tile's
//                64 V rows; the probability of row 4*t+g comes from its owner lane through a wave shuffle
// Every workgroup leaves (m, l, acc[96]) in the workspace; the one that arrives last at the row's counter merges the
// S partials, writes the bf16 output and re-arms the counter (so the workspace needs zeroing only once).
// FUSED: q comes un-rotated inside the fused qkv row; every workgroup rotates q itself (96 values), the workgroup
// whose key range contains the new position also rotates k, appends k/v to the cache and uses them from LDS.
// ------------------------------------------------------------------------------------------------------------
struct DecodeAttnParams {
  const bf16_t* q;            // FUSED: qkv rows [B][3*H*96] (un-rotated); else rotated q [B][H*96]
  const float* cos; const float* sin;   // FUSED: [capacity][96]
  const int* len;             // FUSED: cache_len[b] (= position and append index; n_keys = len + 1); else n_keys[b]
  bf16_t* kc; bf16_t* vc;     // [B][H][cap][96]
  bf16_t* o;                  // [B][H*96]
  const uint64_t* vbits; int nwords;
  unsigned* cnt; float* part; // workspace: arrival counters [B*H], partials [B*H][S][DEC_PSTRIDE]
  int H, cap, S, T; float scale;
};
constexpr int DEC_PSTRIDE = 104;   // m, l, 6 pad, acc[96]

template <bool FUSED>
__global__ __launch_bounds__(64) void decode_attn_split_kernel(const DecodeAttnParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t s_q[96], s_k[96], s_v[96];
  const int lane = threadIdx.x, split = blockIdx.x, bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int ln = p.len[b];
  const int n = FUSED ? ln + 1 : ln;
  const int k_begin = split * p.T * 64;
  const int k_end = min(n, k_begin + p.T * 64);
  bf16_t* kb = p.kc + (size_t)bh * p.cap * 96;
  bf16_t* vb = p.vc + (size_t)bh * p.cap * 96;
  float* part = p.part + ((size_t)bh * p.S + split) * DEC_PSTRIDE;
  const int g = lane >> 4, i16 = lane & 15;
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (k_begin < k_end) {
    const bool owner = FUSED && ln >= k_begin;          // ln < k_end holds by construction (k_end <= ln + 1)
    u32x4 kr[12], vr[16];
    auto issue_tile = [&](int base) {                    // all 28 loads of a tile go out back to back
      const bf16_t* krow = kb + (size_t)min(base + lane, k_end - 1) * 96;       // clamped rows carry probability 0
#pragma unroll
      for (int i = 0; i < 12; ++i) kr[i] = *(const u32x4*)(krow + i * 8);
#pragma unroll
      for (int t2 = 0; t2 < 16; ++t2) {
        const int r = min(base + 4 * t2 + g, k_end - 1);
        vr[t2] = *(const u32x4*)(vb + (size_t)r * 96 + min(i16, 11) * 8);
      }
    };
    issue_tile(k_begin);                                 // in flight while q is rotated
    if (FUSED) {
      if (lane < 48) {
        const bf16_t* row = p.q + (size_t)b * 3 * p.H * 96 + h * 96;
        const float c0 = p.cos[(size_t)ln * 96 + lane], c1 = p.cos[(size_t)ln * 96 + lane + 48];
        const float s0 = p.sin[(size_t)ln * 96 + lane], s1 = p.sin[(size_t)ln * 96 + lane + 48];
        const float q0 = bf16_bits_to_f32(row[lane]), q1 = bf16_bits_to_f32(row[lane + 48]);
        ((__bf16*)s_q)[lane] = (__bf16)(q0 * c0 - q1 * s0);          // rotate-half: d < 48 pairs with -x[d+48]
        ((__bf16*)s_q)[lane + 48] = (__bf16)(q1 * c1 + q0 * s1);
        if (owner) {
          const bf16_t* kr = row + p.H * 96;
          const bf16_t* vr = row + 2 * p.H * 96;
          const float k0 = bf16_bits_to_f32(kr[lane]), k1 = bf16_bits_to_f32(kr[lane + 48]);
          const __bf16 kn0 = (__bf16)(k0 * c0 - k1 * s0), kn1 = (__bf16)(k1 * c1 + k0 * s1);
          ((__bf16*)s_k)[lane] = kn0;
          ((__bf16*)s_k)[lane + 48] = kn1;
          ((__bf16*)kb)[(size_t)ln * 96 + lane] = kn0;
          ((__bf16*)kb)[(size_t)ln * 96 + lane + 48] = kn1;
          s_v[lane] = vr[lane];
          s_v[lane + 48] = vr[lane + 48];
          vb[(size_t)ln * 96 + lane] = vr[lane];
          vb[(size_t)ln * 96 + lane + 48] = vr[lane + 48];
        }
      }
    } else if (lane < 12) {
      *(u32x4*)(s_q + lane * 8) = *(const u32x4*)(p.q + (size_t)bh * 96 + lane * 8);
    }
    __syncthreads();
    // q is read from LDS where it is used (a broadcast): holding it (48 VGPRs) next to the K and V tiles put the kernel at 272 VGPRs = ONE wave
    // per SIMD, 1024 single-wave items in flight for the 1536 of a batch of eight; without it two fit
    for (int t = 0; t < p.T; ++t) {
      const int base = k_begin + t * 64;
      if (base >= k_end) break;
      const int j = base + lane;
      if (t > 0) issue_tile(base);
      if (owner && base <= ln && ln < base + 64) {
        // The new token's row lives in LDS: the tile loads were issued before it was stored, so every lane whose (clamped)
        // row index is ln - the row itself and all rows past k_end - 1 = ln, which carry probability 0 - holds stale
        // cache contents (0 * NaN would poison the sum) and takes the row from LDS instead.
        if (min(j, k_end - 1) == ln) {
#pragma unroll
          for (int i = 0; i < 12; ++i) kr[i] = *(const u32x4*)(s_k + i * 8);
        }
#pragma unroll
        for (int t2 = 0; t2 < 16; ++t2)
          if (min(base + 4 * t2 + g, k_end - 1) == ln) vr[t2] = *(const u32x4*)(s_v + min(i16, 11) * 8);
      }
      bool ok = j < k_end;
      if (p.vbits && (base >> 6) < p.nwords) ok = ok && ((p.vbits[(size_t)b * p.nwords + (base >> 6)] >> lane) & 1ull);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 12; ++i) s = dot8_bf16(kr[i], *(const u32x4*)(s_q + i * 8), s);
      s = ok ? s * p.scale : -INFINITY;
      const float mn = fmaxf(m, wave_max(s));
      if (mn == -INFINITY) continue;                                   // wave-uniform: nothing visible yet
      const float a = __expf(m - mn);
      const float pr = ok ? __expf(s - mn) : 0.f;
      l = l * a + wave_sum(pr);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= a;
#pragma unroll
      for (int t2 = 0; t2 < 16; ++t2) {
        const float w = __shfl(pr, 4 * t2 + g);
        u32x4 vt = vr[t2];
        asm volatile("" : "+v"(vt));                     // widened here, row by row - not all 128 values ahead of the loop (that made it 272 VGPRs: one wave per SIMD)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] = __builtin_fmaf(w, bf16_lo(vt[e]), acc[2 * e]);
          acc[2 * e + 1] = __builtin_fmaf(w, bf16_hi(vt[e]), acc[2 * e + 1]);
        }
      }
      m = mn;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      acc[e] += __shfl_xor(acc[e], 16);
      acc[e] += __shfl_xor(acc[e], 32);
    }
  }
  // Partials travel between workgroups (possibly on different XCDs, i.e. different L2s) as agent-scope relaxed atomic
  // stores / loads: those carry sc1 and are written through / read past the non-coherent levels.  A __threadfence()
  // here would instead make every workgroup write back and invalidate its whole L2 (buffer_wbl2 + buffer_inv).
#define AKI_ST_AGENT(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define AKI_LD_AGENT(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
  if (lane == 0) { AKI_ST_AGENT(part, m); AKI_ST_AGENT(part + 1, l); }
  if (lane < 12) {
#pragma unroll
    for (int e = 0; e < 8; ++e) AKI_ST_AGENT(part + 8 + lane * 8 + e, acc[e]);
  }
  // ---- last workgroup of this (batch, head) merges the partials
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores above have reached the coherence point
  unsigned prev = 0;
  if (lane == 0) prev = __hip_atomic_fetch_add(p.cnt + bh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  prev = __shfl(prev, 0);
  if (prev != (unsigned)(p.S - 1)) return;
  asm volatile("" ::: "memory");
  // lane = (split slot sl = lane / 12 in 0..4, column chunk ch = lane % 12): five splits are merged per pass with all of
  // a pass's loads in flight together (each is a round trip to memory); the five slots then meet through LDS
  __shared__ float s_mg[5][12][10];
  const float* pp = p.part + (size_t)bh * p.S * DEC_PSTRIDE;
  const int sl = lane / 12, ch = lane - sl * 12;
  float gm = -INFINITY, lt = 0.f, o8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o8[e] = 0.f;
  if (sl < 5) {
    for (int s2 = sl; s2 < p.S; s2 += 5) {
      const float* ps = pp + (size_t)s2 * DEC_PSTRIDE;
      const float ms = AKI_LD_AGENT(ps), ls = AKI_LD_AGENT(ps + 1);
      float a[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = AKI_LD_AGENT(ps + 8 + ch * 8 + e);
      const float mn = fmaxf(gm, ms);
      const float f0 = gm == -INFINITY ? 0.f : __expf(gm - mn), f1 = ms == -INFINITY ? 0.f : __expf(ms - mn);
      lt = lt * f0 + ls * f1;
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = o8[e] * f0 + a[e] * f1;
      gm = mn;
    }
    s_mg[sl][ch][0] = gm;
    s_mg[sl][ch][1] = lt;
#pragma unroll
    for (int e = 0; e < 8; ++e) s_mg[sl][ch][2 + e] = o8[e];
  }
  __syncthreads();
  if (lane < 12) {
    float sv[5][10];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int e = 0; e < 10; ++e) sv[q][e] = s_mg[q][lane][e];
    lds_reads_landed();
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int e = 0; e < 10; ++e) asm volatile("" : "+v"(sv[q][e]));
    float M5 = -INFINITY;
#pragma unroll
    for (int q = 0; q < 5; ++q) M5 = fmaxf(M5, sv[q][0]);
    lt = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const float mq = sv[q][0];
      const float f = mq == -INFINITY ? 0.f : __expf(mq - M5);
      lt += sv[q][1] * f;
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] += sv[q][2 + e] * f;
    }
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    u32x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = pack_bf16x2(o8[2 * e] * inv, o8[2 * e + 1] * inv);
    *(u32x4*)(p.o + (size_t)bh * 96 + lane * 8) = ov;
  }
  if (lane == 0) AKI_ST_AGENT(p.cnt + bh, 0u);
#undef AKI_ST_AGENT
#undef AKI_LD_AGENT
}

static inline size_t dec_cnt_bytes(int B, int H) { return (((size_t)B * H * 4) + 255) / 256 * 256; }

size_t decode_attn_ws_bytes(int B, int H, int Dh, int cap) {
  const size_t tiles = ((size_t)cap + 63) / 64;
  const size_t split = dec_cnt_bytes(B, H) + (size_t)B * H * tiles * DEC_PSTRIDE * 4;
  const size_t f32_q = (size_t)B * H * Dh * 4;      // f32 path: rotated q scratch
  return split > f32_q ? split : f32_q;
}

// max_keys: host-side upper bound of n_keys over the batch (sizes the grid; keys beyond it would be ignored).
int decode_attn_split_launch(const void* q_or_qkv, const float* cos, const float* sin, const int* len, void* kc, void* vc, void* o,
                             const uint64_t* vbits, int nwords, int B, int H, int cap, int max_keys, float scale, bool fused,
                             void* ws, size_t ws_bytes, hipStream_t s) {
  if (max_keys <= 0 || max_keys > cap) max_keys = cap;
  // Tiles per item from the cache CAPACITY, items per head from max_keys: a launch sized for the keys cached so far (eager steps) and one
  // sized for the whole cache (a captured step) then cut the keys at the same places and differ only by trailing empty items, whose
  // partials (m = -inf, l = 0) fold exactly - eager and replayed steps give the same bits at any cache size.
  const int tiles = (max_keys + 63) / 64, tiles_cap = (cap + 63) / 64;
  int T = (int)(((size_t)B * H * tiles_cap + AKI_DEC_ITEMS - 1) / AKI_DEC_ITEMS);
  if (T < 1) T = 1;
  const int S = (tiles + T - 1) / T;
  if (ws == nullptr || ws_bytes < dec_cnt_bytes(B, H) + (size_t)B * H * S * DEC_PSTRIDE * 4) return AKI_ERR_WORKSPACE;
  DecodeAttnParams p = {(const bf16_t*)q_or_qkv, cos, sin, len, (bf16_t*)kc, (bf16_t*)vc, (bf16_t*)o, vbits, nwords,
                        (unsigned*)ws, (float*)((char*)ws + dec_cnt_bytes(B, H)), H, cap, S, T, scale};
  const dim3 grid(S, B * H), block(64);
  AKI_CLEAR_ERR();
  if (fused) hipLaunchKernelGGL(decode_attn_split_kernel<true>, grid, block, 0, s, p);
  else hipLaunchKernelGGL(decode_attn_split_kernel<false>, grid, block, 0, s, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
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


// ---- greedy token pick: what the reference's generate loop does between two decode steps (HF GenerationMixin greedy branch:
// argmax over the vocabulary, finished rows take pad_token_id, append, eos check; src/aki.py:136-209 hands its kwargs to it) as
// ONE launch that can sit inside the replayed step - the loop's five small launches and its per-token host sync were 4 % of a token.
// Ties go to the lower index and a NaN outranks every number (torch.argmax's ordering).
struct PickParams {
  const bf16_t* logits; int B, V, ld;
  const int64_t* eos; int n_eos; int64_t pad;
  unsigned char* done; int64_t* ids; int64_t* tokens; int tokens_ld;
  int* cache_len; const int* start_len; int advance; int* done_at;
  // optional: the NEXT decode step's input row, gathered here (DecoupledEmbedding, src/helpers.py:350-492: ids above max_original_id index the
  // additional table) so that a greedy token needs no embedding launch
  const bf16_t* emb_main; const bf16_t* emb_extra; int64_t max_original_id; int d; bf16_t* emb_out;
};

__device__ __forceinline__ bool pick_better(float a, int ia, float b, int ib) {
  const bool na = a != a, nb = b != b;
  if (na != nb) return na;
  if (na) return ia < ib;
  return a > b || (a == b && ia < ib);
}

// One workgroup of 16 waves per row: the scan of 32 064 logits is a latency chain (load, eight compares, next load); 1024 threads walk it in 4 trips of two
// loads each instead of 16 trips of one (19.4 -> measured in profiles/r05_decode_token_trace.txt).
constexpr int PICK_THREADS = 1024;
__global__ __launch_bounds__(PICK_THREADS) void greedy_pick_kernel(const PickParams p) {
  __shared__ float s_v[PICK_THREADS / 64];
  __shared__ int s_i[PICK_THREADS / 64];
  __shared__ int64_t s_next;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // What the bookkeeping needs does not depend on the argmax: thread 0 requests it BEFORE the scan, so that behind the scan only stores are
  // left (it used to be a chain of five dependent round trips behind the reduction, about a third of the launch).
  bool was_done = false;
  int len0 = 0, start0 = 0;
  int64_t eos4[4] = {-1, -1, -1, -1};                        // token ids are >= 0: -1 matches nothing
  if (tid == 0) {
    was_done = p.done != nullptr && p.done[b] != 0;
    if (p.cache_len != nullptr) {
      len0 = p.cache_len[b];
      if (p.start_len) start0 = p.start_len[b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < p.n_eos) eos4[i] = p.eos[i];
  }
  const bf16_t* row = p.logits + (size_t)b * p.ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const bool vec = ((p.ld & 7) == 0) && ((((uintptr_t)p.logits) & 15) == 0);
  const int nvec = vec ? p.V / 8 : 0;
  for (int c0 = tid; c0 < nvec; c0 += 2 * PICK_THREADS) {
    const int c1 = c0 + PICK_THREADS;                        // both loads go out before the first compare
    const u32x4 v0 = *(const u32x4*)(row + (size_t)c0 * 8);
    const u32x4 v1 = c1 < nvec ? *(const u32x4*)(row + (size_t)c1 * 8) : u32x4{0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u};   // -inf: never picked
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32x4 v = h ? v1 : v0;
      const int c = h ? c1 : c0;
      if (h && c1 >= nvec) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf16_lo(v[e]), hi = bf16_hi(v[e]);
        if (pick_better(lo, c * 8 + 2 * e, best, bi)) { best = lo; bi = c * 8 + 2 * e; }
        if (pick_better(hi, c * 8 + 2 * e + 1, best, bi)) { best = hi; bi = c * 8 + 2 * e + 1; }
      }
    }
  }
  for (int i = nvec * 8 + tid; i < p.V; i += PICK_THREADS) {
    const float x = bf16_bits_to_f32(row[i]);
    if (pick_better(x, i, best, bi)) { best = x; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (pick_better(ov, oi, best, bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { s_v[wave] = best; s_i[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 1; w < PICK_THREADS / 64; ++w)
      if (pick_better(s_v[w], s_i[w], best, bi)) { best = s_v[w]; bi = s_i[w]; }
    const int64_t nxt = was_done ? p.pad : (int64_t)bi;
    int t = 0;
    if (p.cache_len != nullptr) {
      t = len0 + p.advance - start0;
      if (p.advance) p.cache_len[b] = len0 + 1;
    }
    if (p.tokens != nullptr && t >= 0 && t < p.tokens_ld) p.tokens[(size_t)b * p.tokens_ld + t] = nxt;
    p.ids[b] = nxt;
    s_next = nxt;
    if (p.done != nullptr && !was_done) {
      bool hit = eos4[0] == nxt || eos4[1] == nxt || eos4[2] == nxt || eos4[3] == nxt;
      for (int i = 4; i < p.n_eos; ++i) hit = hit || p.eos[i] == nxt;
      if (hit) {
        p.done[b] = 1;
        if (p.done_at) p.done_at[b] = t;
      }
    }
  }
  if (p.emb_out != nullptr) {
    __syncthreads();
    const int64_t nxt = s_next;
    const bool extra = p.emb_extra != nullptr && nxt > p.max_original_id;
    const bf16_t* src = extra ? p.emb_extra + (size_t)(nxt - p.max_original_id - 1) * p.d : p.emb_main + (size_t)nxt * p.d;
    bf16_t* dst = p.emb_out + (size_t)b * p.d;
    for (int c = tid; c < p.d / 8; c += PICK_THREADS) *(u32x4*)(dst + (size_t)c * 8) = *(const u32x4*)(src + (size_t)c * 8);
  }
}

int greedy_pick_launch(const void* logits, int B, int V, int ld, const int64_t* eos, int n_eos, int64_t pad, unsigned char* done, int64_t* ids,
                       int64_t* tokens, int tokens_ld, int* cache_len, const int* start_len, int advance, int* done_at, const void* emb_main,
                       const void* emb_extra, int64_t max_original_id, int d, void* emb_out, hipStream_t s) {
  PickParams p = {(const bf16_t*)logits, B, V, ld, eos, n_eos, pad, done, ids, tokens, tokens_ld, cache_len, start_len, advance, done_at,
                  (const bf16_t*)emb_main, (const bf16_t*)emb_extra, max_original_id, d, (bf16_t*)emb_out};
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(greedy_pick_kernel, dim3(B), dim3(PICK_THREADS), 0, s, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
