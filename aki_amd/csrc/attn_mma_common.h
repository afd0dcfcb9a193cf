// attn_mma_common.h - pieces shared by the two span-driven attention cores (mma_attn_bf16.hip: 32-row blocks, two waves per
// SIMD; mma_attn64_bf16.hip: 64 rows per wave, one wave per SIMD): launch parameters, the K/V tile image and the mask helpers.
#pragma once
#include <type_traits>

#include "aki_device.h"

namespace aki {

struct AttnParams {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;
  float* lse;
  const aki_mma_rect* rects;
  const uint64_t* vbits;
  const int* seq_lens;
  int max_rects;
  int B, H, L;
  int nqt, nwords;
  int splits;    // workgroups per (batch, head) pair
  int group_bh;  // pairs per dispatch group
  int kvcap;  // rows per (batch, head) of k / v (>= L when they are a KV cache)
  float scale_log2;  // scale * log2(e)
  int dead_uniform;
};

constexpr int KROW = 192;   // K rows unpadded: bank conflicts are removed by chunk ^= (row>>2)&3 (low 2 bits of the 16-B chunk)
constexpr int VROW = 192;
constexpr int KTILE = 64 * KROW;
constexpr int VTILE = 64 * VROW;
constexpr int NSTAGE = 3;   // LDS ring: tile j computing, j+1 landed or landing, j+2 being issued

constexpr int MAX_VB_WORDS = 256;  // L <= 16384

// v_max3_f32 without the canonicalising v_max hipcc inserts in front of fmaxf on MFMA results
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// The largest of a lane's 32 scores (two 16-register tiles): four chains of max3 in TWO asm statements.  (One statement per link - the
// v_max3 helper above, chained - made hipcc pad an s_nop between every two links: asm statements that depend on each other get one.)
__device__ __forceinline__ float tile_max32(const f32x16& s0, const f32x16& s1) {
  float m, m1, m2, m3;
  asm("v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %2, %10, %11, %12\n\tv_max3_f32 %3, %13, %14, %15\n\t"
      "v_max3_f32 %0, %0, %16, %17\n\tv_max3_f32 %1, %1, %18, %19\n\tv_max3_f32 %2, %2, %20, %21\n\tv_max3_f32 %3, %3, %22, %23"
      : "=&v"(m), "=&v"(m1), "=&v"(m2), "=&v"(m3)
      : "v"(s0[0]), "v"(s0[1]), "v"(s1[0]), "v"(s0[4]), "v"(s0[5]), "v"(s1[4]), "v"(s0[8]), "v"(s0[9]), "v"(s1[8]), "v"(s0[12]), "v"(s0[13]), "v"(s1[12]),
        "v"(s1[1]), "v"(s0[2]), "v"(s1[5]), "v"(s0[6]), "v"(s1[9]), "v"(s0[10]), "v"(s1[13]), "v"(s0[14]));
  asm("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %2, %2, %8, %9\n\tv_max3_f32 %3, %3, %10, %11\n\t"
      "v_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %2, %2, %14, %15\n\tv_max3_f32 %0, %0, %1, %2\n\tv_max_f32 %0, %0, %3"
      : "+v"(m), "+v"(m1), "+v"(m2), "+v"(m3)
      : "v"(s0[3]), "v"(s1[2]), "v"(s0[7]), "v"(s1[6]), "v"(s0[11]), "v"(s1[10]), "v"(s0[15]), "v"(s1[14]),
        "v"(s1[3]), "v"(s1[7]), "v"(s1[11]), "v"(s1[15]));
  return m;
}

// A lane's 32 score columns of a 64-key tile, in register order i = 16*kb + r, are
//   col(i) = c0 + 4h + (i&3) + 8*((i&15)>>2) + 32*(i>>4)      (strictly increasing in i)
// so "col <= y" is a PREFIX of the register order.  count_le(x) = number of i with col(i) - (c0+4h) <= x.
__device__ __forceinline__ int count_le(int x) {
  const int n = 4 * (x >> 3) + min((x & 7) + 1, 4);
  return x < 0 ? 0 : min(n, 32);
}

// the n lowest bits set, n in [0, 32]
__device__ __forceinline__ unsigned low_bits(int n) { return n >= 32 ? ~0u : ((1u << n) - 1u); }

// bit BIT of hid -> -inf (hidden) or 0 (visible), two VALU ops and no VCC round trip (hipcc turns the C form into
// v_and / v_cmp / v_cndmask with hazard nops)
template <int BIT>
__device__ __forceinline__ float mask_bias(int hid, int ninf) {
  float b;            // ONE statement: between two dependent asm statements hipcc pads an s_nop (32 per bias tile)
  asm("v_bfe_i32 %0, %1, %2, 1\n\tv_and_b32 %0, %3, %0" : "=&v"(b) : "v"(hid), "n"(BIT), "s"(ninf));
  return b;
}

// two columns in one ordered statement (a bias written in MFMA gaps: no pin, no pad between dependent statements)
template <int BIT0, int BIT1>
__device__ __forceinline__ void mask_bias2(int hid, int ninf, float& b0, float& b1) {
  asm volatile("v_bfe_i32 %0, %2, %3, 1\n\tv_bfe_i32 %1, %2, %4, 1\n\tv_and_b32 %0, %5, %0\n\tv_and_b32 %1, %5, %1"
               : "=&v"(b0), "=&v"(b1) : "v"(hid), "n"(BIT0), "n"(BIT1), "s"(ninf));
}

// compile-time loop (the tr-read offsets below must be immediates of an inline-asm statement)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ds_read_b64_tr_b16 through inline asm: the builtin form is treated by hipcc as "may alias any LDS-DMA in flight" and
// gets an s_waitcnt vmcnt(0) in front, which would drain the K/V ring.  The asm form is invisible to that analysis
// (and to the compiler's lgkmcnt bookkeeping: the data is only touched after wait_tr_reads below; cdna guide 5.7).
template <int OFF>
__device__ __forceinline__ u32x2 ds_read_tr(unsigned lds_addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
  return r;
}

}  // namespace aki
