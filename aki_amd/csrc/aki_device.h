// aki_device.h - shared device helpers for the gfx950 (CDNA4 / MI355X) kernels.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aki_mi355x.h"

namespace aki {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

typedef uint16_t bf16_t;  // storage type on the host side of the ABI

// Decode attention (decode.hip, decode_chain.hip - the two must split a cache identically: their outputs are compared bit for bit): a
// (head, split) item takes T 64-key tiles with T = ceil(B * H * tiles / AKI_DEC_ITEMS); every tile after an item's first is a serial
// round trip behind the item's dependency, so items stay one tile long up to this many of them.
#ifndef AKI_DEC_ITEMS
#define AKI_DEC_ITEMS 2048
#endif
#define AKI_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define AKI_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// two f32 -> packed bf16x2 (RNE, NaN preserving: lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  bf16x2 p;
  p[0] = (__bf16)lo;
  p[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float round_bf16(float x) { return (float)((__bf16)x); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_tanh(float x) {
  const float c = 0.79788456080286535588f;
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// Fast forms for the bf16 MFMA epilogues (v_exp_f32 + v_rcp_f32, a handful of FMAs).  Absolute errors are below
// 1e-6 relative to |x| - three orders of magnitude under the bf16 output resolution; the f32 parity path keeps the
// library erff/tanhf.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float silu_fast(float x) { return x * fast_sigmoid(x); }
// 0.5 (1 + tanh u) == sigmoid(2u)
__device__ __forceinline__ float gelu_tanh_fast(float x) {
  const float u = 0.79788456080286535588f * (x + 0.044715f * x * x * x);
  return x * fast_sigmoid(2.0f * u);
}
// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7)
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfz = 1.0f - poly * __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
  return 0.5f * x * (1.0f + copysignf(erfz, x));
}

// Exchange across the two 32-lane halves with v_permlane32_swap: after swap(a = x, b = x),
// a = [x_lo, x_lo] and b = [x_hi, x_hi] (row = 32 lanes).  Inline asm on purpose: with the builtin
// (__builtin_amdgcn_permlane32_swap) hipcc / ROCm 7.2 -O3 folds the SECOND result into the first
// (extractvalue 1 -> extractvalue 0 in the optimised IR), which silently turns the cross-half
// reduction into a lane-local one.  "s_nop 1" = the 2 wait states a VALU write of either operand needs
// before v_permlane*_swap reads it (cdna_hip_programming.md T21 / section 5.7 item 2).
__device__ __forceinline__ void halves_pair(float x, float& lo, float& hi) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;
  hi = b;
}
// max / sum over lanes l and l^32, result in every lane.
// Call between a chain of 16-pass MFMAs (32x32x16) and INLINE-ASM VALU code that reads their results.  hipcc's hazard
// recognizer pads an XDL-write -> VALU-read pair only when it can see both instructions; an asm consumer placed right after
// the last MFMA reads the register's previous content (cdna_hip_programming.md 5.7 item 2).  In the attention kernels that
// consumer is the v_max3_f32 chain: a stale score only perturbs the running max, which softmax is invariant to, so every
// tolerance test passed while ~0.5 % of the outputs differed by an ulp from launch to launch.  The nops are tied to both
// accumulators by data dependency, so they can be scheduled neither before the MFMAs nor after the first reader.
__device__ __forceinline__ void mfma_results_settle(f32x16& a, f32x16& b) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ float halves_max(float x) {
  float lo, hi;
  halves_pair(x, lo, hi);
  return fmaxf(lo, hi);
}
__device__ __forceinline__ float halves_sum(float x) {
  float lo, hi;
  halves_pair(x, lo, hi);
  return lo + hi;
}

// Bijective XCD-aware block remap (cdna_hip_programming.md T1): blocks that share an XCD (bid % 8)
// get a contiguous chunk of the logical tile sequence, so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Full LDS wait between the READS of a fold through LDS and their first use.  Round 3 met one site (the row-statistics fold of
// gemm_bf16.hip) where the second half of a compiler-paired `ds_read2st64_b64`, consumed behind hipcc's counted `lgkmcnt(k > 0)`,
// was used before it had arrived in 0.5 % of launches; the cause was never established, so every fold of cross-wave partials
// through LDS now loads its values into registers, executes this, and only then adds them up (same order: bit-identical results).
// tools/lgkm_audit.py lists what is left of the pattern (profiles/r04_lgkm_audit.txt).
__device__ __forceinline__ void lds_reads_landed() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// ... and the values themselves made opaque behind that wait: a consumer is register arithmetic, which the "memory" clobber alone
// does not keep from being scheduled above the wait (the decode merge's running max was).  No instruction is emitted for the ties.
template <int N>
__device__ __forceinline__ void lds_fold_ready(float (&v)[N]) {
  lds_reads_landed();
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// ---- wave / workgroup reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// sums of a and b over a workgroup of NT threads; red: >= 2*NT/64 floats of LDS
template <int NT>
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[w] = a; red[w + NT / 64] = b; }
  __syncthreads();
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) { sa += red[i]; sb += red[i + NT / 64]; }
  a = sa; b = sb;
  __syncthreads();
}

}  // namespace aki

// ---- host side helpers (api translation units) -------------------------------------------------
#define AKI_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return AKI_ERR_INVALID_ARG; \
  } while (0)
#define AKI_CHECK_ALIGN16(p) \
  do {                       \
    if (((uintptr_t)(p)) & 15) return AKI_ERR_ALIGNMENT; \
  } while (0)
// hipGetLastError() is sticky per thread: clear whatever an earlier (unrelated) HIP call left behind
// before launching, then read the launch's own status.
#define AKI_CLEAR_ERR() ((void)hipGetLastError())
#define AKI_LAUNCH_CHECK() \
  do {                     \
    if (hipGetLastError() != hipSuccess) return AKI_ERR_LAUNCH; \
  } while (0)

static inline size_t aki_elt_size(int dtype) { return dtype == AKI_DT_BF16 ? 2 : 4; }
static inline size_t aki_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
