// gemm_bf16.hip - y = act(x W^T + bias) [+ residual] on CDNA4 MFMA, bf16 in / f32 accumulate.
//
// Replaces every nn.Linear on the AKI forward path (include/aki_mi355x.h: aki_linear_fwd) and, with
// the QKV_ROPE epilogue, stage 1 of the fused MMA op (HF:phi3/modeling_phi3.py:228-241).
//
// Orientation.  Both operands are K-contiguous ([M,K] activations, [N,K] nn.Linear weights), so both
// MFMA fragments are one 16-byte LDS read.  The product is computed "swapped": A-operand = W rows
// (features), B-operand = x rows (tokens), i.e. each 16x16 accumulator block is C^T[feature][token] with
//   token   = lane & 15                 (on the lanes)
//   feature = 4*(lane>>4) + reg         (4 consecutive features per lane)
// so per-token epilogues are lane-local: RoPE (the feature order is permuted so that d and d+48 share lane and register),
// residual add, SwiGLU (gate/up blocks of the same features are staged into the same wave), and one
// v_permlane16_swap per dword turns two blocks' 8-byte runs into 16-byte row-major stores.
//
// MFMA shape: v_mfma_f32_16x16x32_bf16.  Measured on this MI355X (tools/mfma_peak.hip, 2 waves/SIMD, one
// barrier per K-step, random operands): 2.16 PFLOP/s vs 1.77 for 32x32x16 - the chip holds a higher clock on
// the 16x16 shape (cdna guide rule 28); in this kernel the switch was worth +8..19 %, the 16-byte stores +3..6 %.
//
// Tiles (waves = WN x WM, wave tile = NF*16 features x NT*16 tokens, BK = 64):
//   BIG   2x4 waves, 8x4 blocks -> 256 x 256     generic large GEMMs (also QKV + RoPE, see EPI_QKV_ROPE8)
//   SMALL 2x2 waves, 4x4 blocks -> 128 x 128     shapes whose 256^2 tiling cannot fill 256 CUs (2 WGs/CU)
// LDS: 2 stages x (BN + BM) rows x 128 B filled by global_load_lds (16 B/lane; 1 KiB per wave-instruction =
// 8 rows).  The LDS image is lane-linear; the bank-conflict swizzle (chunk ^= (row>>1)&7) is applied on the
// SOURCE address and again on the ds_read_b128 address (guide rule 21); conflict-free for both the 32x32 and
// the 16x16 fragment read groups.  One vmcnt(0)+barrier per K-step; tile k+1 streams in under tile k's MFMAs.
// (tools/gemm_lab.hip: a deeper BK=32 x 4-stage counted-vmcnt ring, interleaved LDS-DMA issue, fragment
// double-buffering, setprio and wave staggering were all measured and bought nothing on this structure.)
//
// K-loop variants (template parameter PIPE; which one a launch gets is decided in launch_big / run_planned):
//   0  two stages, one vmcnt(0) + barrier per K-step (small tiles, fp8); NST 3-4: a ring for one-row weight-streaming launches
//   1  256 x 256: the barrier in the MIDDLE of the step, fragment reads and DMA issue spread between the MFMAs (lab: the form before the three-deep rings)
//   2  = 1 + the residual tile prefetched under the last two K-steps into the two stage buffers (lab, as above)
//   3  lab: four waves, one per SIMD, hand-placed stream (asm MFMAs on AGPR accumulators), per-phase stamps
//   4  = 1 with the WEIGHT tiles on a three-deep ring (160 KiB of LDS): a weight tile is asked for two K-steps ahead - operands come out of HBM
//   5  = 4 + the residual tile's four 32 KiB blocks prefetched into ring slots as they fall free (o_proj)
//   6  = 5 with the TOKEN tiles three deep instead (down_proj: the activation is the larger cold operand);  7 = 4 likewise (not dispatched)
//   8  128 x 96 tile, unpipelined loop, token tiles three deep (SigLIP fc2 reads a 40 MB activation out of HBM)
#include <type_traits>

#include "aki_device.h"

namespace aki {

enum { EPI_PLAIN = 0, EPI_SWIGLU = 1, EPI_QKV_ROPE8 = 3 };

// compile-time loops (the hand-placed K-loop below needs immediates for the fragment index and the ds_read offset)
template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
// chunk swizzle of a 64-byte-row LDS image read with ds_read_b128 by MFMA 16x16x32 fragment lanes (row = lane & 15, chunk = lane >> 4):
// the instruction's four lane groups are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS), so the
// four lanes of a group that share row & 3 are rows q = 0, 3 of one chunk and q = 1, 2 of the next (q = row >> 2); 0, 3, 2, 1 keeps them apart
__device__ __forceinline__ int swz4(int q) { return (q & 3) ^ ((q & 1) << 1); }
template <class F> __device__ __forceinline__ void static_for16(F&& f) { static_for_impl<0, 16>(f); }
template <class F> __device__ __forceinline__ void static_for64(F&& f) { static_for_impl<0, 64>(f); }

// EPI_QKV_ROPE8: QKV + RoPE on the generic 256x256 / 128x128 tiles.  RoPE only needs d and d + 48 of a head in the same lane
// and register slot, not a whole head per wave, so the 3*H*96 output features are re-ordered into 32-feature UNITS - for q
// and k: (head slot hs, j) = d in [16j, 16j+16) plus its rotate-half partner block [48+16j, ...); for v: 32 consecutive d -
// and a wave with NF blocks owns NF/2 consecutive units, first halves in blocks [0, NF/2), partners in [NF/2, NF).  The
// permutation lives in the weight-row gather of the LDS staging (every LDS row already has its own source pointer) and in
// the epilogue's destination address: no weight copy, identical arithmetic, and the kernel gets the big tile's efficiency
// (the earlier 192-wide one-head-per-wave tile measured 1080 TF/s against 1190-1300 for 256x256).
__device__ __forceinline__ int qkv_unit_row(int u, int s_, int H) {   // first weight row of block (unit u, half s_)
  const int nqk = 6 * H;
  if (u < nqk) { const int hs = u / 3, j = u - 3 * hs; return hs * 96 + 16 * j + 48 * s_; }
  const int u2 = u - nqk, vh = u2 / 3, j2 = u2 - 3 * vh;
  return 2 * H * 96 + vh * 96 + 32 * j2 + 16 * s_;
}

struct GemmParams {
  const bf16_t* x;
  const bf16_t* w;
  const bf16_t* bias;
  const bf16_t* residual;
  bf16_t* y;
  int M, N, K;  // N = weight rows
  int ldx, ldw, ldy, ldr;
  int res_row_mod;
  int m_offset;  // global index of row 0 (tail launches): residual row = (m + m_offset) % res_row_mod, token index for QKV
  int act;
  int tiles_m, tiles_n;
  bf16_t* preact;  // EPI_SWIGLU, training forward: the pre-activations g | u as bf16 [M, N] (aki_linear_args.preact_out); NULL = not kept
  int ld_preact;
  int wide;  // 16-byte stores allowed (n_out % 8 == 0, ldy % 8 == 0, y 16-B aligned)
  int res_wide;  // 16-byte residual loads allowed (n_out % 8 == 0, ldr % 8 == 0, residual 16-B aligned)
  // QKV + RoPE
  bf16_t* q_out;
  bf16_t* k_out;
  bf16_t* v_out;
  const float* cos;
  const float* sin;
  const int* position_ids;
  int H, L;
  int kvcap;  // rows per (batch, head) of k_out / v_out (KV cache capacity); q_out always uses L
  // fp8 (e4m3) operands: x and w point at bytes, ldx / ldw count bytes, one f32 scale per token row / weight row
  const float* sx;
  const float* sw;
  // two-segment weight (EPI_PLAIN): logical rows >= w2_row0 come from w2 (aki_linear_args)
  const bf16_t* w2;
  int w2_row0, w2_rows;
  // Normalisation folded into the GEMMs (aki_linear_args).  Consumer: y = row_scale[m] * (acc - row_shift[m] * col_c[n]) ...
  const float* row_scale;
  const float* row_shift;
  const float* col_c;
  // Producer (EPI_PLAIN): per-row statistics of the tile's bf16 OUTPUT, reduced over the row by the last tile of the row panel
  float* st_rstd;
  float* st_mean;
  float* st_part;      // [slots][2][M] partial sums (sum of squares, sum), slot = tile column * WN + wave column
  unsigned* st_cnt;    // [tiles_m] arrival counters, zero between launches
  // Split-K (template SK; small-M launches, see plan_small_m): workgroup t = tile * ksplit + slice walks K-steps [slice * nk / ksplit, (slice + 1) * nk / ksplit)
  int ksplit;
  float* sk_part;      // [tiles][ksplit][BN * BM] f32 partial accumulators in fragment order (16 B per lane: coalesced)
  size_t sk_bytes;     // bytes behind sk_part (host side: does the planned split fit?)
  int sk_slice_major;  // workgroup order: every tile of slice 0, then slice 1, ... (see the kernel)
  unsigned* sk_cnt;    // [tiles] arrival tickets, zero between launches
#ifdef AKI_LAB_HOOKS
  int probe_block;          // lab: which workgroup stamps (default 0)
  long long* clock_probe;   // lab: 32 int64: {shader cycles, 100 MHz ticks} of workgroup 0, [2..17] phase sums of the PIPE 3 loop, [18] prologue, [19] epilogue cycles
#endif
  float st_eps;
};

// Two feature blocks (P = block n, Q = block n+1), each 4 consecutive features per lane as 2 packed dwords.
// After the swaps lane row r = lane>>4 owns 8 consecutive features of block n + (r&1), starting at 8*(r>>1).
__device__ __forceinline__ u32x4 pair_to_wide(unsigned p0, unsigned p1, unsigned q0, unsigned q1) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(p0), "+v"(q0), "+v"(p1), "+v"(q1));
  u32x4 o = {p0, p1, q0, q1};
  return o;
}

// ACT is a template parameter on purpose: with a runtime switch the 32-fold unrolled epilogue inlines 32 copies of
// erff/tanhf; even when never executed they bloat the code enough to cost 13 % on the large GEMMs (I-cache misses
// once per tile; measured in tools/gemm_lab.hip against the identical kernel without them).
typedef int v8i_t __attribute__((ext_vector_type(8)));

// FP8: e4m3 operands through v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the bf16 MFMA rate).  A 128-byte
// LDS row then holds BK = 128 k-values instead of 64, so staging, swizzle and the epilogues are byte-for-byte the same; the
// per-row dequantisation scales (one per token, one per weight row) multiply the f32 accumulators before the epilogue.
// KG (K groups, small-M launches with at most one tile per CU): the workgroup is KG x (WN x WM) waves; group g has its own stage buffers and walks
// K-steps g, g + KG, ... of the tile, so a CU holds KG waves per SIMD whose DMA waits, fragment reads and MFMAs overlap (one 4-wave workgroup alone
// serialises them: 1200-1450 cycles per K-step for 384-512 of MFMA work).  After the loop the groups' accumulators meet in LDS and are added in
// group order by group 0, which runs the epilogue alone: an on-chip, fixed-order fold - no partial tiles in memory, no tickets, bit-reproducible.
template <int NF, int NT, int WN, int WM, int EPI, int ACT, bool FP8 = false, int NST = 2, int PIPE = 0, int SK = 0, int KG = 1>
__global__ __launch_bounds__(WN* WM * 64 * KG, ((NF * NT > 32 || KG > 1) ? 1 : 2)) void gemm_bf16_kernel(const GemmParams p) {
  static_assert(!SK || (!FP8 && PIPE <= 1), "split-K: bf16, the plain K loops and the mid-step-barrier pipeline");
  static_assert(KG == 1 || (!FP8 && PIPE == 0 && NST == 2 && !SK), "K groups: bf16, the two-stage loop");
  constexpr int BK = FP8 ? 128 : 64, ES = FP8 ? 1 : 2, NWAVES = WN * WM;
  constexpr int WROWS = NF * 16;      // features per wave
  constexpr int BN = WN * WROWS;      // features per block tile
  constexpr int WTOK = NT * 16;       // tokens per wave
  constexpr int BM = WM * WTOK;       // tokens per block tile
  constexpr int ROWS = BN + BM;
  constexpr int STAGE_BYTES = ROWS * 128;
  constexpr int NLD = ROWS / 8 / NWAVES;  // global_load_lds per thread per stage
  static_assert(ROWS % (8 * NWAVES) == 0 && NF % 2 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem_all[];

  const int lane = threadIdx.x & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kgrp = KG > 1 ? wave_all / NWAVES : 0;                 // this wave's K group
  const int wave = KG > 1 ? wave_all % NWAVES : wave_all;          // ... and its place in the group's WN x WM arrangement
  const int tid = KG > 1 ? (int)threadIdx.x - kgrp * (NWAVES * 64) : (int)threadIdx.x;   // thread index inside the group (group 0 runs the epilogue)
  char* const smem = smem_all + (KG > 1 ? kgrp * (NST * STAGE_BYTES) : 0);
  const int wn = wave % WN, wm = wave / WN;
  const int l15 = lane & 15, kg = lane >> 4;
#ifdef AKI_LAB_HOOKS
  // clock probe (lab build): shader cycles and the 100 MHz wall clock over workgroup 0's lifetime -> the engine clock the
  // kernel actually ran at (tools/gemm_clock.py)
  long long probe_c0 = 0, probe_w0 = 0;
  if ((int)blockIdx.x == p.probe_block && tid == 0 && p.clock_probe) { probe_c0 = clock64(); probe_w0 = wall_clock64(); }
  // probe_block == -2: EVERY workgroup stamps the 100 MHz wall clock (one time base for the whole chip) at its start, K-loop begin, K-loop end and
  // exit into clock_probe[4 * blockIdx.x + 0..3] - the launch's timeline (dispatch ramp, stragglers, fold tails): tools/small_m_timeline.py
#define AKI_WG_STAMP(k) do { if (p.probe_block == -2 && tid == 0 && p.clock_probe) p.clock_probe[4 * (size_t)blockIdx.x + (k)] = wall_clock64(); } while (0)
  AKI_WG_STAMP(0);
#else
#define AKI_WG_STAMP(k) do { } while (0)
#endif

  // ---- tile id: XCD-contiguous chunks, grouped so concurrently running tiles of an XCD share operand panels ----
  int t = xcd_remap(blockIdx.x, gridDim.x);
  int ksp = 0, kt0 = 0;                           // split-K: this workgroup's slice of K and its first K-step
  int nk = p.K / (FP8 ? 128 : 64);
  if constexpr (SK) {
    // tile-major (product): the slices of a tile are neighbours in the remapped order - same XCD, same L2.  slice-major (lab): all tiles of slice 0,
    // then slice 1, ... - an XCD then reads one or two K slices of the activation instead of all of them (for `down` at M = 655 the activation crosses
    // the fabric 8 times: 86 MB beside 50 MB of weights).  Measured (tools/attic/slice_major_ab.py): equal at M = 655 (36.6 / 58.5 vs 37.0 / 59.2 us),
    // 19-28 % SLOWER at M = 207 (28.5 / 48.4 vs 24.0 / 37.9): the partial tiles of a fold then come from other XCDs.
    if (p.sk_slice_major) {
      const int tiles = gridDim.x / p.ksplit;
      ksp = t / tiles;
      t -= ksp * tiles;
    } else {
      ksp = t % p.ksplit;
      t /= p.ksplit;
    }
    kt0 = ksp * nk / p.ksplit;
    nk = (ksp + 1) * nk / p.ksplit - kt0;
  }
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n;
  const int group = t / per_group;
  const int first_m = group * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (t % per_group) % gsz;
  const int tn = (t % per_group) / gsz;
  const int m0 = tm * BM;
  const int n0 = (EPI == EPI_SWIGLU) ? tn * (BN / 2) : tn * BN;  // first output feature of the tile
  const int n_out = (EPI == EPI_SWIGLU) ? p.N / 2 : p.N;

  // ---- per-thread staging sources ----------------------------------------------------------------------------
  auto weight_row_n = [&](int n0, int row) {    // tile row (0 .. BN-1, wave-local feature blocks) of the tile at feature n0 -> row of the weight matrix
    if (EPI == EPI_SWIGLU) {  // wave-local blocks [0,NF/2) = gate rows, [NF/2,NF) = up rows of the same features
      const int w_ = row / WROWS, within = row % WROWS, nb = within >> 4, i = within & 15;
      const int f = min(n0 + w_ * (WROWS / 2) + (nb % (NF / 2)) * 16 + i, n_out - 1);
      return (nb < NF / 2) ? f : n_out + f;
    } else if (EPI == EPI_QKV_ROPE8) {
      const int w_ = row / WROWS, within = row % WROWS, nb = within >> 4, i = within & 15;
      const int u = n0 / 32 + w_ * (NF / 2) + (nb % (NF / 2));
      return min(qkv_unit_row(min(u, 9 * p.H - 1), nb / (NF / 2), p.H) + i, p.N - 1);
    } else {
      return min(n0 + row, p.N - 1);
    }
  };
  auto weight_row = [&](int row) { return weight_row_n(n0, row); };
  const char* src[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    // PIPE == 3 keeps the two k32 halves of a tile in separate 32 KiB regions (64-byte rows; a 1-KiB piece = 16 rows of one half,
    // 16-byte chunk c of row r at position c ^ swz4(r >> 2)): a half is re-filled as soon as ITS fragments have been read
    const int rowgroup = j * NWAVES + wave;
    const int row = PIPE == 3 ? rowgroup * 16 + (lane >> 2) : rowgroup * 8 + (lane >> 3);
    const int chunk = PIPE == 3 ? (lane & 3) ^ swz4(row >> 2) : (lane & 7) ^ ((row >> 1) & 7);
    if (PIPE == 3 && j >= NLD / 2) { src[j] = nullptr; continue; }
    if (row < BN) {
      const int wr = weight_row(row);
      const char* wp = (const char*)p.w + (size_t)wr * p.ldw * ES;
      if (EPI == EPI_PLAIN && !FP8) {
        if (p.w2 != nullptr && wr >= p.w2_row0) wp = (const char*)p.w2 + (size_t)min(wr - p.w2_row0, p.w2_rows - 1) * p.ldw * ES;
      }
      src[j] = wp + chunk * 16;
    } else {
      const int xrow = min(m0 + row - BN, p.M - 1);
      src[j] = (const char*)p.x + (size_t)xrow * p.ldx * ES + chunk * 16;
    }
    if constexpr (SK) src[j] += (size_t)kt0 * 128;
    if constexpr (KG > 1) src[j] += (size_t)kgrp * 128;               // group g starts at K-step g ...
  }
  constexpr int KSTRIDE = KG * 128;                                   // ... and advances KG steps at a time (bytes per K-step of the group)
  if constexpr (KG > 1) nk /= KG;                                     // host: the K-step count is a multiple of KG

  auto stage = [&](int s, int kt) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      char* dst = smem + s * STAGE_BYTES + (j * NWAVES + wave) * 1024;
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[j] + (size_t)kt * (KG > 1 ? KSTRIDE : 128)), AKI_LDS_PTR(dst), 16, 0, 0);
    }
  };

  f32x4 acc[NF][NT];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[n][m][r] = 0.f;

  const int swz = (l15 >> 1) & 7;  // == (row>>1)&7: every block row base is a multiple of 16
  const int wbase = (wn * WROWS + l15) * 128;
  const int xbase = BN * 128 + (wm * WTOK + l15) * 128;

  // ---- epilogue operands (bias, folded-norm scale / shift / column sums).  Their fetch is a DRAM-latency round trip that
  // nothing in the epilogue can hide; small tiles (several short launches per layer, spare registers) issue it here, under
  // the first K-step's DMA wait, the 256^2 tiles (at the register limit) right after the K loop, ahead of the residual
  // staging, whose own round trip then covers it.
  constexpr int NOUT = (EPI == EPI_SWIGLU) ? NF / 2 : NF;     // output feature blocks per wave
  const int fwave = n0 + wn * (EPI == EPI_SWIGLU ? WROWS / 2 : WROWS);
  // folded LayerNorm (consumer), kernel-uniform.  Not on the PIPELINED 256^2 tile: its epilogue has no registers left for the
  // column sums (they spilled); the host sends row_shift launches that want the big tile to its unpipelined twin.
  const bool shifted = (PIPE <= 1 || PIPE == 4 || PIPE == 7 || PIPE == 8) && (EPI == EPI_PLAIN) && p.row_shift != nullptr;
  u32x2 biasp[NOUT];
  f32x4 colc4[NOUT];
  float rsv[NT], muv[NT];
  auto fetch_epilogue_operands = [&]() {
    if (EPI == EPI_QKV_ROPE8) return;
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
      const int f = fwave + n * 16 + 4 * kg;
      biasp[n] = u32x2{0u, 0u};
      colc4[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (EPI == EPI_PLAIN && p.bias && f < n_out) biasp[n] = *(const u32x2*)(p.bias + f);
      if (EPI == EPI_PLAIN && shifted && f < n_out) colc4[n] = *(const f32x4*)(p.col_c + f);
    }
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      const int mr = min(m0 + wm * WTOK + m * 16 + l15, p.M - 1);
      rsv[m] = p.row_scale ? p.row_scale[mr] : 1.0f;
      muv[m] = (EPI == EPI_PLAIN && shifted) ? p.row_shift[mr] : 0.0f;
    }
  };
  constexpr bool EARLY_OPERANDS = (!PIPE || PIPE == 8) && !FP8 && NF * NT <= 16;
  constexpr int RCH = BN / 8;                                  // 16-byte chunks per residual tile row (epilogue staging, below)
  constexpr int CMASK = RCH >= 16 ? 15 : RCH - 1;
  // residual prefetch of the two-stage loop (small tiles): possible when the residual tile fits ONE stage buffer in whole 1-KiB pieces
  constexpr int RES_PIECES = (BM * RCH * 16) / (NWAVES * 1024);
  constexpr bool RES_PF_OK = EPI == EPI_PLAIN && ACT == 0 && !FP8 && (!PIPE || PIPE == 8) && NST == 2 && RCH <= 64 && (64 % RCH == 0) && BM * RCH * 16 <= STAGE_BYTES &&
                             (PIPE != 8 || BM * RCH * 16 <= 2 * BM * 128) &&
                             (BM * RCH * 16) % (NWAVES * 1024) == 0;
  const bool res_pf = RES_PF_OK && KG == 1 && p.residual != nullptr && p.res_wide && p.res_row_mod <= 0 && n_out >= 8;   // workgroup-uniform
  int res_off = 0;                 // where the residual tile image sits in smem
  // QKV + RoPE on the pipelined 256 x 256 tile: cos / sin rows prefetched under the last half K-step (issue_cos_sin below); workgroup-uniform
  const bool cs_pf = EPI == EPI_QKV_ROPE8 && (PIPE == 1 || PIPE == 4 || PIPE == 7) && BM == 256 && n0 / 32 < 6 * p.H && p.position_ids == nullptr && p.L >= BM;
  if constexpr (EARLY_OPERANDS) fetch_epilogue_operands();
  auto compute = [&](const char* sb) {
    if constexpr (FP8) {   // one k128 step per BK: the lane's 32 bytes are chunks 2kg and 2kg+1 of its row
      const int c0 = ((2 * kg) ^ swz) << 4, c1 = ((2 * kg + 1) ^ swz) << 4;
      v8i_t a[NF], b[NT];
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const u32x4 lo = *(const u32x4*)(sb + wbase + n * 2048 + c0), hi = *(const u32x4*)(sb + wbase + n * 2048 + c1);
        a[n] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        const u32x4 lo = *(const u32x4*)(sb + xbase + m * 2048 + c0), hi = *(const u32x4*)(sb + xbase + m * 2048 + c1);
        b[m] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < NT; ++m)
          acc[n][m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[n], b[m], acc[n][m], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {  // two k32 steps per BK
        const int coff = ((4 * ks + kg) ^ swz) << 4;
        bf16x8 a[NF], b[NT];
#pragma unroll
        for (int n = 0; n < NF; ++n) a[n] = *(const bf16x8*)(sb + wbase + n * 2048 + coff);
#pragma unroll
        for (int m = 0; m < NT; ++m) b[m] = *(const bf16x8*)(sb + xbase + m * 2048 + coff);
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int m = 0; m < NT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], b[m], acc[n][m], 0, 0, 0);
      }
    }
  };
#ifdef AKI_LAB_HOOKS
  long long probe_l0 = 0, probe_l1 = 0;                  // K-loop begin / end of workgroup 0 -> clock_probe[18] = prologue, [19] = epilogue cycles
  if ((int)blockIdx.x == p.probe_block && tid == 0 && p.clock_probe) probe_l0 = clock64();
#endif
  AKI_WG_STAMP(1);
  if constexpr (PIPE == 3) {
    // One wave per SIMD with a hand-placed stream (lab).  Four waves, wave tile 128 features x 128 tokens: 256 accumulator registers
    // in the AGPR half of the file, the fragments of both k32 halves (128 VGPRs) in the other - a third fewer LDS fragment bytes
    // per FLOP than the 8-wave tile, and 95 instead of 134 other instructions per 128 MFMAs and SIMD.  This is the structure of the
    // vendor library's kernel for these shapes (MT256x256x64, 256 threads; profiles/r03_hipblaslt_kernel_name.csv).  Left to hipcc's
    // register allocation the same tile moved fragments through AGPRs (132 v_accvgpr_* per K-step) and ran 16-20 % slower; here every
    // MFMA is an asm statement with the accumulator constrained to an AGPR and the fragment reads are asm too, in program order:
    // one ds_read_b128 behind every 4th MFMA of the first half-step (the k32-half-1 fragments of tile kt), one read or one LDS-DMA
    // piece behind every 2nd MFMA of the second (half-0 fragments of tile kt+1, the 16 pieces of tile kt+2); waits are explicit.
    static_assert(PIPE != 3 || (NF == 8 && NT == 8 && WN == 2 && WM == 2 && NST == 2 && !FP8 && NLD == 16), "256 x 256 tile on four waves");
    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
    // stage buffer = [k32 half][512 rows: 256 features, 256 tokens][64 B]; stage 1 = address ^ STAGE_BYTES (64 KiB buffers, 64 KiB aligned)
    constexpr int HALF_BYTES = STAGE_BYTES / 2;
    const unsigned cpos = (unsigned)((kg ^ swz4(l15 >> 2)) << 4);
    unsigned adrA0 = lds0 + (wn * WROWS + l15) * 64 + cpos, adrB0 = lds0 + (BN + wm * WTOK + l15) * 64 + cpos;
    unsigned adrA1 = adrA0 + HALF_BYTES, adrB1 = adrB0 + HALF_BYTES;
    bf16x8 a0[8], b0[8], a1[8], b1[8];
#define AKI_DSR(dst, adr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(adr), "n"(OFF))
#define AKI_MFMA(n_, m_, A_, B_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[n_][m_]) : "v"(A_[n_]), "v"(B_[m_]))
    auto read_frag = [&](auto idx, bf16x8 (&fa)[8], bf16x8 (&fb)[8], unsigned adra, unsigned adrb) {   // fragment 0..7 = features, 8..15 = tokens
      constexpr int I = decltype(idx)::value;
      if constexpr (I < 8) AKI_DSR(fa[I], adra, I * 1024); else AKI_DSR(fb[I - 8], adrb, (I - 8) * 1024);
    };
    constexpr int NH = NLD / 2;                                  // DMA pieces per wave and half tile
    auto dma_piece = [&](auto jj, int s, int h, int kt_) {
      constexpr int J = decltype(jj)::value;
      char* dst = smem + s * STAGE_BYTES + h * HALF_BYTES + (J * NWAVES + wave) * 1024;
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[J] + (size_t)kt_ * 128 + h * 64), AKI_LDS_PTR(dst), 16, 0, 0);
    };
    // DMA order: (tile 0, half 0), (0, 1), (1, 0), (1, 1), then one half per half-step: (kt+2, h) during half-step (kt, h), into the region
    // whose fragments every wave has finished reading at the barrier in front of that half-step.  A half then has a step and a half to land:
    // it is awaited - counted, the two younger halves stay in flight - at the barrier in front of the half-step that READS it.
    static_for_impl<0, NH>([&](auto j) { dma_piece(j, 0, 0, 0); });
    static_for_impl<0, NH>([&](auto j) { dma_piece(j, 0, 1, 0); });
    if (nk > 1) {
      static_for_impl<0, NH>([&](auto j) { dma_piece(j, 1, 0, 1); });
      static_for_impl<0, NH>([&](auto j) { dma_piece(j, 1, 1, 1); });
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NH) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    static_for16([&](auto i) { read_frag(i, a0, b0, adrA0, adrB0); });
#ifdef AKI_LAB_HOOKS
    // phase stamps (lab): shader cycles per wave of workgroup 0 spent in {top wait + barrier, first half-step, mid wait + barrier,
    // second half-step}, summed over the K loop -> clock_probe[2 + 4 * wave ..].  s_memtime returns through lgkmcnt: a stamp is only
    // read behind one of the loop's own lgkmcnt(0) waits (SETTLE pins that for the compiler), so the stamps add no wait of their own.
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, ph_a = 0, ph_b = 0, ph_c = 0, ph_d = 0;
#define AKI_STAMP(t) asm volatile("s_memtime %0" : "=s"(t))
#define AKI_SETTLE() asm volatile("" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+s"(t4))
#else
#define AKI_STAMP(t)
#define AKI_SETTLE()
#endif
    // one K-step; NEXT1 / NEXT2 (compile time): tiles kt+1 / kt+2 exist.  The 16 fragment reads of a half-step go out behind its FIRST
    // 16 MFMAs (they are then 48 MFMAs = ~770 cycles old at the wait that needs them), the 8 DMA pieces behind every 4th MFMA after those
    auto step3 = [&](int kt, auto n1, auto n2) {
      constexpr bool NEXT1 = decltype(n1)::value, NEXT2 = decltype(n2)::value;
      // half-0 fragments of tile kt are in a0 / b0; my pieces of (kt, half 1) have landed
      AKI_STAMP(t0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NEXT1 ? 2 * NH : 0) : "memory");
      __builtin_amdgcn_s_barrier();                              // ... everybody's have, and nobody reads (kt, half 0) from LDS any more
#ifdef AKI_LAB_HOOKS
      AKI_SETTLE();
      ph_c += t3 - t2; ph_d += t4 - t3;                          // the previous step's second half
#endif
      AKI_STAMP(t1);
      static_for64([&](auto ii) {                                // first half-step: MFMAs on a0 / b0, half-1 fragments of tile kt -> a1 / b1
        constexpr int I = decltype(ii)::value, N_ = I >> 3, M_ = I & 7;
        AKI_MFMA(N_, M_, a0, b0);
        if constexpr (I < 16) read_frag(std::integral_constant<int, I>{}, a1, b1, adrA1, adrB1);
        if constexpr (NEXT2 && I >= 17 && I < 65 && (I - 17) % 6 == 0) dma_piece(std::integral_constant<int, ((I - 17) / 6)>{}, kt & 1, 0, kt + 2);
      });
      AKI_STAMP(t2);
      if constexpr (NEXT1) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NEXT2 ? 2 * NH : NH) : "memory");   // (kt+1, half 0) has landed; my reads of (kt, half 1) are done
        __builtin_amdgcn_s_barrier();
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#ifdef AKI_LAB_HOOKS
      AKI_SETTLE();
      ph_a += t1 - t0; ph_b += t2 - t1;
#endif
      AKI_STAMP(t3);
      adrA0 ^= STAGE_BYTES; adrA1 ^= STAGE_BYTES; adrB0 ^= STAGE_BYTES; adrB1 ^= STAGE_BYTES;       // tile kt+1's buffer
      static_for64([&](auto ii) {                                // second half-step: MFMAs on a1 / b1; tile kt+1's half-0 fragments
        constexpr int I = decltype(ii)::value, N_ = I >> 3, M_ = I & 7;
        AKI_MFMA(N_, M_, a1, b1);
        if constexpr (NEXT1 && I < 16) read_frag(std::integral_constant<int, I>{}, a0, b0, adrA0, adrB0);
        if constexpr (NEXT2 && I >= 17 && I < 65 && (I - 17) % 6 == 0) dma_piece(std::integral_constant<int, ((I - 17) / 6)>{}, kt & 1, 1, kt + 2);
      });
      AKI_STAMP(t4);
    };
    {
      int kt = 0;
      for (; kt + 2 < nk; ++kt) step3(kt, std::true_type{}, std::true_type{});
      if (kt + 1 < nk) { step3(kt, std::true_type{}, std::false_type{}); ++kt; }
      step3(kt, std::false_type{}, std::false_type{});
    }
#ifdef AKI_LAB_HOOKS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    AKI_SETTLE();
    ph_c += t3 - t2; ph_d += t4 - t3;
    if (blockIdx.x == 0 && lane == 0 && p.clock_probe) {
      p.clock_probe[2 + 4 * wave + 0] = (long long)ph_a; p.clock_probe[2 + 4 * wave + 1] = (long long)ph_b;
      p.clock_probe[2 + 4 * wave + 2] = (long long)ph_c; p.clock_probe[2 + 4 * wave + 3] = (long long)ph_d;
    }
#endif
#undef AKI_STAMP
#undef AKI_SETTLE
    // hipcc does not see the asm MFMAs as matrix instructions: it pads no XDL-write -> VALU-read hazard in front of the epilogue's
    // accumulator reads - the last MFMA (8 passes) has to have retired
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#undef AKI_DSR
#undef AKI_MFMA
  } else if constexpr (PIPE && PIPE != 8) {
    // Mid-step barrier pipeline (bf16, 2 stages).  In the loop below every LDS fragment read of a K-step sits between
    // the barrier and the MFMAs that need it, and all eight waves do theirs at the same time: the LDS array (24
    // ds_read_b128 per wave and step, ~770 array cycles per CU) and then the matrix cores (2048 cycles per SIMD) take
    // turns.  Here the barrier moves to the MIDDLE of the step: the k32-half-1 fragments of tile kt are read under the
    // half-0 MFMAs, and behind the barrier - which publishes tile kt+1 - its half-0 fragments are read under the half-1
    // MFMAs of tile kt, into the registers the half-0 MFMAs have just released.  The DMA of tile kt+2 goes out right
    // after the barrier (nobody reads that buffer any more) and has a full step to land, as before.
    static_assert(NST == 2 && !FP8, "bf16, two stages");
    auto load_frags = [&](const char* sb, int ks, bf16x8 (&a)[NF], bf16x8 (&b)[NT]) {
      const int coff = ((4 * ks + kg) ^ swz) << 4;
#pragma unroll
      for (int n = 0; n < NF; ++n) a[n] = *(const bf16x8*)(sb + wbase + n * 2048 + coff);
#pragma unroll
      for (int m = 0; m < NT; ++m) b[m] = *(const bf16x8*)(sb + xbase + m * 2048 + coff);
    };
    auto mma = [&](const bf16x8 (&a)[NF], const bf16x8 (&b)[NT]) {
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], b[m], acc[n][m], 0, 0, 0);
    };
    auto interleave_reads = [&]() {     // scheduling directives: NF + NT groups of (MFMAs, one ds_read)
#pragma unroll
      for (int i = 0; i < NF + NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, (NF * NT) / (2 * (NF + NT)) > 0 ? (NF * NT) / (2 * (NF + NT)) : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    };
    auto interleave_dma = [&]() {       // then NLD groups of (one MFMA, one global_load_lds)
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    };
    // QKV + RoPE: the cos / sin rows of the tile's tokens (the epilogue's LDS image: token row r at r * 400, 12 chunks of cos[0:48], 12 of
    // sin[0:48], one pad chunk) come in by LDS-DMA under the LAST half K-step - behind its barrier nobody reads the stage buffers any more -
    // instead of after the loop, where the round trip plus 12 ds_write_b128 per thread cost every tile ~6 k cycles (tools/
    // gemm_epilogue_probe.py).  100 pieces of 1 KiB, lane i of piece q = chunk (64 q + i) % 25 of row (64 q + i) / 25.  Positions implied
    // (token index mod L, L >= the tile's rows so that a tile wraps at most once); explicit position_ids keep the staging after the loop.
    auto issue_cos_sin = [&]() {
      if constexpr (EPI == EPI_QKV_ROPE8 && BM == 256) {
        const int t0 = (m0 + p.m_offset) % p.L, last = min(BM, p.M - m0) - 1;
#pragma unroll
        for (int j = 0; j < (100 + NWAVES - 1) / NWAVES; ++j) {
          const int q = j * NWAVES + wave;
          if (q < 100) {
            const int idx = q * 64 + lane, row = idx / 25, c = idx - row * 25;
            int pos = t0 + min(row, last);
            pos -= pos >= p.L ? p.L : 0;
            const float* src_ = (c < 12 ? p.cos + 4 * c : p.sin + 4 * (c == 24 ? 0 : c - 12)) + (size_t)pos * 96;
            __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src_), AKI_LDS_PTR(smem + q * 1024), 16, 0, 0);
          }
        }
      }
    };
    bf16x8 a0[NF], b0[NT], a1[NF], b1[NT];
    if constexpr (PIPE >= 4 && PIPE <= 7) {
      // PIPE == 4: the same loop with the WEIGHT tiles on a three-deep ring (3 x 32 KiB) beside the two token-panel buffers (2 x 32 KiB; 160 KiB, all of
      // the LDS): a weight tile is asked for TWO steps ahead instead of one.  Weights come out of HBM in every launch of the forward, and with the
      // eight tiles that share a weight panel in lock-step every one of them waits for that first touch: workgroup 0's K-step is 2436 cycles with the
      // weight on-die and 3032 with it in HBM (tools/gemm_cold_stamps.py), and the lab loop that asks 1.5 steps ahead loses a third of that.
      // Issue order per step: token tile kt+2, then weight tile kt+3 - so the counted wait at the barrier leaves exactly the youngest weight tile in flight.
      // PIPE == 5 = PIPE 4 + the residual tile prefetched under the last 1.5 K-steps (what PIPE 2 does on the two-stage layout): the tile's four
      // blocks of 64 token rows (32 KiB each - the rows of one wave row wm) go wherever a 32 KiB slot has been read for the last time: block 0 into the
      // slot of weight tile nk-3 after the barrier of step nk-3, blocks 1 and 2 into the slots of weight tile nk-2 and token tile nk-2 after the barrier
      // of step nk-2, block 3 into the slot of weight tile nk-1 after the last barrier.  The epilogue finds a wave's block through res_off (set per wave below).
      // PIPE == 7 / 6 = PIPE 4 / 5 with the roles swapped: the TOKEN tiles on the three-deep ring, the weight tiles on two buffers - for launches whose
      // activation is the larger cold operand (down_proj: 86 MB of activations against 50 MB of weights).
      constexpr bool DX = PIPE == 6 || PIPE == 7, RESPF = PIPE == 5 || PIPE == 6;
      static_assert(PIPE < 4 || PIPE > 7 || (BN == BM && (BN / 8) % NWAVES == 0 && NLD % 2 == 0), "weight / token pieces split evenly");
      static_assert(!RESPF || (WTOK == 64 && RCH == 32 && BM == 256 && NWAVES == 8), "one 32 KiB residual block per wave row");
      constexpr int WB = BN * 128, XB = BM * 128, NLH = NLD / 2;
      static_assert(WB == XB, "equal slots");
      char* const sS = smem + 3 * WB;                   // the two-deep ring behind the three-deep one
      constexpr int JD = DX ? NLH : 0, JS = DX ? 0 : NLH;   // src[] indices of the deep / shallow operand's pieces (weights are src[0 .. NLH))
      auto stage_w = [&](int slot, int kt_) {            // the DEEP operand's tile kt_ (weights unless DX)
#pragma unroll
        for (int j = 0; j < NLH; ++j)
          __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[JD + j] + (size_t)kt_ * 128), AKI_LDS_PTR(smem + slot * WB + (j * NWAVES + wave) * 1024), 16, 0, 0);
      };
      auto stage_x = [&](int slot, int kt_) {            // the SHALLOW operand's tile kt_
#pragma unroll
        for (int j = 0; j < NLH; ++j)
          __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[JS + j] + (size_t)kt_ * 128), AKI_LDS_PTR(sS + slot * XB + (j * NWAVES + wave) * 1024), 16, 0, 0);
      };
      const int xb0 = (wm * WTOK + l15) * 128;
      auto load_frags2 = [&](const char* sdeep, const char* sshal, int ks, bf16x8 (&a)[NF], bf16x8 (&b)[NT]) {
        const char* const sw = DX ? sshal : sdeep;
        const char* const sx = DX ? sdeep : sshal;
        const int coff = ((4 * ks + kg) ^ swz) << 4;
#pragma unroll
        for (int n = 0; n < NF; ++n) a[n] = *(const bf16x8*)(sw + wbase + n * 2048 + coff);
#pragma unroll
        for (int m = 0; m < NT; ++m) b[m] = *(const bf16x8*)(sx + xb0 + m * 2048 + coff);
      };
      auto interleave_pieces = [&](int n_) {
        for (int i = 0; i < n_; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      };
      int blkoff[4] = {0, 0, 0, 0};                     // PIPE 5: LDS byte offset of residual block b
      auto issue_res_block = [&](int b_, int dst_off) {
        if constexpr (RESPF) {
          blkoff[b_] = dst_off;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = j * NWAVES + wave;                       // 1 KiB piece of the block = two token rows
            const int tk = b_ * 64 + 2 * q + (lane >> 5);
            const int ch = (lane & (RCH - 1)) ^ (tk & CMASK);
            const int f = min(n0 + 8 * ch, n_out - 8);
            const bf16_t* src_ = p.residual + (size_t)min(m0 + tk, p.M - 1) * p.ldr + f;
            __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src_), AKI_LDS_PTR(smem + dst_off + q * 1024), 16, 0, 0);
          }
        }
      };
      stage_w(0, 0); stage_x(0, 0);
      if (nk > 1) { stage_w(1, 1); stage_x(1, 1); }
      if (nk > 2) stage_w(2, 2);
      if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NLH) : "memory");      // tile 0 landed; tile 1 and weight tile 2 may be in flight
      else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#ifdef AKI_LAB_HOOKS
      if ((int)blockIdx.x == p.probe_block && tid == 0 && p.clock_probe) p.clock_probe[21] = clock64() - probe_l0;   // first tile landed
#endif
      load_frags2(smem, sS, 0, a0, b0);
      int ws = 0;                                      // ring slot of weight tile kt
      auto step4 = [&](int kt, auto next1, auto next2, auto next3) {
        constexpr bool NEXT1 = decltype(next1)::value, NEXT2 = decltype(next2)::value, NEXT3 = decltype(next3)::value;
        const int ws1 = ws == 2 ? 0 : ws + 1;
        load_frags2(smem + ws * WB, sS + (kt & 1) * XB, 1, a1, b1);
        mma(a0, b0);
        interleave_reads();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLH) : "memory");   // tile kt+1 landed (weight tile kt+2 stays in flight); my reads of tile kt are done
        else if constexpr (NEXT1 || !RESPF) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // PIPE 5, last step: no tile is awaited, the residual blocks of step nk-2 stay in flight
        __builtin_amdgcn_s_barrier();
        if constexpr (NEXT1) load_frags2(smem + ws1 * WB, sS + ((kt + 1) & 1) * XB, 0, a0, b0);
        if constexpr (NEXT2) stage_x(kt & 1, kt + 2);
        if constexpr (NEXT3) stage_w(ws, kt + 3);
        if constexpr (RESPF && NEXT2 && !NEXT3) issue_res_block(0, ws * WB);   // step nk-3: no weight tile nk will ask for this slot
        if constexpr (RESPF && NEXT1 && !NEXT2) {             // step nk-2: two more slots have been read for the last time
          if (nk == 2) issue_res_block(0, 2 * WB);                //   (no step nk-3: the third weight slot was never used)
          issue_res_block(1, ws * WB);
          issue_res_block(2, 3 * WB + (kt & 1) * XB);
        }
        if constexpr (RESPF && !NEXT1) {                     // last step
          if (nk == 1) { issue_res_block(0, 1 * WB); issue_res_block(1, 2 * WB); issue_res_block(2, 3 * WB + XB); }
          issue_res_block(3, ws * WB);
        }
        if constexpr (EPI == EPI_QKV_ROPE8 && !NEXT1) { if (cs_pf) issue_cos_sin(); }
        mma(a1, b1);
        if constexpr (NEXT1) interleave_reads();
        if constexpr (NEXT2 || NEXT3) interleave_pieces((NEXT2 ? NLH : 0) + (NEXT3 ? NLH : 0));
        if constexpr (RESPF && !NEXT3) interleave_pieces(NEXT2 ? 4 : (NEXT1 ? 8 : 4));
        __builtin_amdgcn_sched_barrier(0);
        ws = ws1;
      };
      int kt = 0;
      for (; kt + 3 < nk; ++kt) step4(kt, std::true_type{}, std::true_type{}, std::true_type{});
      if (kt + 2 < nk) { step4(kt, std::true_type{}, std::true_type{}, std::false_type{}); ++kt; }
      if (kt + 1 < nk) { step4(kt, std::true_type{}, std::false_type{}, std::false_type{}); ++kt; }
      step4(kt, std::false_type{}, std::false_type{}, std::false_type{});
      if constexpr (RESPF) res_off = (wm == 0 ? blkoff[0] : wm == 1 ? blkoff[1] : wm == 2 ? blkoff[2] : blkoff[3]) - wm * (64 * RCH * 16);
    } else {
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");   // tile 0 landed (tile 1 may be in flight)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_frags(smem, 0, a0, b0);
    // one K-step; NEXT1 / NEXT2: tiles kt+1 / kt+2 exist (compile-time, so that the steady-state step is one basic
    // block and the scheduling directives can spread the reads and the DMA issue between the MFMAs instead of leaving
    // them in front: right after the barrier both waves of a SIMD would otherwise spend ~250 issue cycles on them with
    // the matrix core idle)
    // PIPE == 2 (plain GEMM + residual, 16-byte path, host-checked): the residual tile - BM rows x BN*2 bytes = exactly the two
    // stage buffers - is brought in by LDS-DMA under the LAST TWO K-steps instead of after the loop, where its round trip
    // (every workgroup of a one-round launch at the same moment: 32 MB chip-wide) was fully exposed: +8 us on o_proj / down.
    // Half h (token rows [BM/2*h, +BM/2)) goes into stage buffer h as soon as nobody reads that buffer any more: buffer
    // (nk-2)&1 after the mid-step barrier of step nk-2 (no tile nk to fetch), the other one after the barrier of step nk-1.
    // Image = the epilogue's: chunk c of token row t at chunk position c ^ (t & CMASK), swizzle on the SOURCE address.
    auto issue_res_half = [&](int hb) {
      static_assert(PIPE != 2 || (BM * RCH * 16 == 2 * STAGE_BYTES && NLD * NWAVES * 1024 == STAGE_BYTES), "residual tile = the two stage buffers");
#pragma unroll
      for (int j = 0; j < NLD; ++j) {
        const int q = j * NWAVES + wave;                         // 1 KiB piece of the half = two token rows
        const int tk = hb * (BM / 2) + 2 * q + (lane >> 5);
        const int ch = (lane & (RCH - 1)) ^ (tk & CMASK);
        const int f = min(n0 + 8 * ch, n_out - 8);
        const bf16_t* src_ = p.residual + (size_t)min(m0 + tk, p.M - 1) * p.ldr + f;
        __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src_), AKI_LDS_PTR(smem + hb * STAGE_BYTES + q * 1024), 16, 0, 0);
      }
    };
    auto step = [&](int kt, auto next1, auto next2) {
      constexpr bool NEXT1 = decltype(next1)::value, NEXT2 = decltype(next2)::value;
      const char* sb = smem + (kt & 1) * STAGE_BYTES;
      load_frags(sb, 1, a1, b1);
      mma(a0, b0);
      interleave_reads();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NEXT1 || PIPE != 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my pieces of tile kt+1 landed; my reads of tile kt are done
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // last step: no tile is awaited, and the residual DMA of step nk-2 stays in flight
      __builtin_amdgcn_s_barrier();
      if constexpr (NEXT1) load_frags(smem + ((kt + 1) & 1) * STAGE_BYTES, 0, a0, b0);
      if constexpr (NEXT2) stage(kt & 1, kt + 2);
      if constexpr (PIPE == 2 && !NEXT2) {
        issue_res_half(kt & 1);
        if constexpr (!NEXT1) { if (nk == 1) issue_res_half((kt & 1) ^ 1); }
      }
      if constexpr (EPI == EPI_QKV_ROPE8 && PIPE == 1 && !NEXT1) { if (cs_pf) issue_cos_sin(); }
      mma(a1, b1);
      if constexpr (NEXT1) interleave_reads();
      if constexpr (NEXT2 || PIPE == 2) interleave_dma();
      __builtin_amdgcn_sched_barrier(0);
    };
    int kt = 0;
    for (; kt + 2 < nk; ++kt) step(kt, std::true_type{}, std::true_type{});
    if (kt + 1 < nk) { step(kt, std::true_type{}, std::false_type{}); ++kt; }
    step(kt, std::false_type{}, std::false_type{});
    }
  } else if constexpr (PIPE == 8) {
    // Small tiles (two workgroups per CU), token tiles on a THREE-deep ring beside two weight buffers: a token tile is asked for two steps ahead.
    // For launches whose activation comes out of HBM - SigLIP fc2 reads the 40 MB fc1 has just written: 53.8 us with it on-die, 65.1 from HBM
    // (tools/siglip_cold_probe.py).  LDS: 2 weight slots, 3 token slots and one spare token-sized slot (for the residual tile of the last step).
    static_assert(PIPE != 8 || (!FP8 && NST == 2 && (BN / 8) % NWAVES == 0 && (BM / 8) % NWAVES == 0), "whole pieces per operand and wave");
    constexpr int WB = BN * 128, XB = BM * 128, NLW = BN / 8 / NWAVES, NLX = BM / 8 / NWAVES;
    char* const sXr = smem + 2 * WB;
    auto stage_w8 = [&](int slot, int kt_) {
#pragma unroll
      for (int j = 0; j < NLW; ++j)
        __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[j] + (size_t)kt_ * 128), AKI_LDS_PTR(smem + slot * WB + (j * NWAVES + wave) * 1024), 16, 0, 0);
    };
    auto stage_x8 = [&](int slot, int kt_) {
#pragma unroll
      for (int j = 0; j < NLX; ++j)
        __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[NLW + j] + (size_t)kt_ * 128), AKI_LDS_PTR(sXr + slot * XB + (j * NWAVES + wave) * 1024), 16, 0, 0);
    };
    const int xb8 = (wm * WTOK + l15) * 128;
    auto compute8 = [&](const char* sw, const char* sx) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {  // two k32 steps per BK
        const int coff = ((4 * ks + kg) ^ swz) << 4;
        bf16x8 a[NF], b[NT];
#pragma unroll
        for (int n = 0; n < NF; ++n) a[n] = *(const bf16x8*)(sw + wbase + n * 2048 + coff);
#pragma unroll
        for (int m = 0; m < NT; ++m) b[m] = *(const bf16x8*)(sx + xb8 + m * 2048 + coff);
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int m = 0; m < NT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], b[m], acc[n][m], 0, 0, 0);
      }
    };
    stage_w8(0, 0); stage_x8(0, 0);
    if (nk > 1) stage_x8(1, 1);
    int xs = 0;                                         // ring slot of token tile kt
    for (int kt = 0; kt < nk; ++kt) {
      // weight tile kt and token tile kt have landed: the only younger transfer is token tile kt+1 (issue order per step: weights kt+1, tokens kt+2)
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLX) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // ... for every wave; nobody reads the slots of tile kt-1 any more
      const int xs1 = xs == 2 ? 0 : xs + 1, xs2 = xs1 == 2 ? 0 : xs1 + 1;
      if (kt + 1 < nk) stage_w8((kt + 1) & 1, kt + 1);
      if (kt + 2 < nk) stage_x8(xs2, kt + 2);
      if (kt + 1 == nk && RES_PF_OK && res_pf) {
        // last K-step: two adjacent token-sized slots are free (slots 0-1 when this tile sits in slot 2, else slot 2 and the spare) - the residual tile
        // goes there by LDS-DMA under this step's MFMAs (image as in the epilogue's staging: chunk c of token row t at c ^ (t & CMASK))
        res_off = 2 * WB + (xs == 2 ? 0 : 2 * XB);
#pragma unroll
        for (int j = 0; j < RES_PIECES; ++j) {
          const int q = j * NWAVES + wave;
          const int tk = q * (64 / RCH) + lane / RCH;
          const int ch = (lane & (RCH - 1)) ^ (tk & CMASK);
          const int f = min(n0 + 8 * ch, n_out - 8);
          const bf16_t* src_ = p.residual + (size_t)min(m0 + tk, p.M - 1) * p.ldr + f;
          __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src_), AKI_LDS_PTR(smem + res_off + q * 1024), 16, 0, 0);
        }
      }
      compute8(smem + (kt & 1) * WB, sXr + xs * XB);
      xs = xs1;
    }
  } else if constexpr (NST == 2) {
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();  // (vmcnt(0) + barrier): tile kt landed, the other buffer is no longer being read
      if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
      else if (RES_PF_OK && res_pf) {
        // last K-step: the other stage buffer is free - the residual tile goes there by LDS-DMA, under this step's MFMAs,
        // instead of being fetched after the loop (same image as the epilogue's staging: chunk c of token row t at c ^ (t & CMASK))
        res_off = ((kt + 1) & 1) * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < RES_PIECES; ++j) {
          const int q = j * NWAVES + wave;                       // 1 KiB piece = 1024 / (RCH * 16) token rows
          const int tk = q * (64 / RCH) + lane / RCH;
          const int ch = (lane & (RCH - 1)) ^ (tk & CMASK);
          const int f = min(n0 + 8 * ch, n_out - 8);
          const bf16_t* src_ = p.residual + (size_t)min(m0 + tk, p.M - 1) * p.ldr + f;
          __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src_), AKI_LDS_PTR(smem + res_off + q * 1024), 16, 0, 0);
        }
      }
      compute(smem + (kt & 1) * STAGE_BYTES);
    }
  } else {
    // Deep ring for weight-streaming launches (launch_small: one row of tiles, at most one workgroup per CU): nothing
    // co-resident fills the wait and a small tile's K-step (<= 512 MFMA cycles per wave) is far shorter than the
    // global -> LDS latency, which the 2-stage loop above pays every step.  NST - 1 tiles are in flight; the
    // counted vmcnt retires exactly tile kt (each stage() is NLD DMA instructions per thread), the raw barrier makes it a
    // workgroup-wide fact, and tile kt + NST - 1 then goes into the buffer tile kt - 1 was read from.
    static_assert(NST >= 3 && NST <= 6 && (NST - 2) * NLD < 64, "the counted waits below are written for three to six stages");
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_)
      if (s_ < nk) stage(s_, s_);
    for (int kt = 0; kt < nk; ++kt) {
      const int rem = nk - 1 - kt;                    // tiles behind tile kt; min(rem, NST - 2) of them stay in flight
      if (NST >= 6 && rem >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NLD) : "memory");
      else if (NST >= 5 && rem >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NLD) : "memory");
      else if (NST >= 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
      else if (rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + NST - 1 < nk) stage((kt + NST - 1) % NST, kt + NST - 1);
      compute(smem + (kt % NST) * STAGE_BYTES);
    }
  }

#ifdef AKI_LAB_HOOKS
  if ((int)blockIdx.x == p.probe_block && tid == 0 && p.clock_probe) probe_l1 = clock64();
#endif
  AKI_WG_STAMP(2);
  if constexpr (KG > 1) {
    // fold of the K groups through LDS, in group order (g0 + g1 + ...): the stage buffers are free once every wave has left the loop
    static_assert(KG == 1 || BN * BM * 4 <= NST * STAGE_BYTES, "a group's accumulators fit its own stage buffers");
    __syncthreads();
    f32x4* const mine = (f32x4*)smem + (wave * NF * NT) * 64 + lane;             // fragment order: one 16-byte slot per lane and block
    if (kgrp != 0) {
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < NT; ++m) mine[(n * NT + m) * 64] = acc[n][m];
    }
    __syncthreads();
    if (kgrp != 0) { AKI_WG_STAMP(3); return; }                                   // (a finished wave leaves the workgroup's barriers)
#pragma unroll
    for (int g = 1; g < KG; ++g) {
      const f32x4* const theirs = (const f32x4*)(smem_all + g * (NST * STAGE_BYTES)) + (wave * NF * NT) * 64 + lane;
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[n][m] += theirs[(n * NT + m) * 64];
    }
  }
  if constexpr (SK) {
    // Split-K fold, deterministic and without a second launch.  Every slice writes its f32 accumulators in fragment order (one 16-byte
    // store per lane and block: 1 KiB per wave-instruction, write-through), drains, and draws a ticket (cdna_hip_programming.md
    // Guideline 16, ticket form - the protocol of the row statistics below).  The LAST slice to arrive - whichever it is - reads all
    // `ksplit` partials back in slice order, its own included, and adds them up in that order: the sum does not depend on the arrival
    // order, so a launch is bit-reproducible.  It then runs the ordinary epilogue (residual, statistics, ...); the others are done.
    if (p.ksplit > 1) {
      __shared__ unsigned sk_last;
      constexpr int TILE_BYTES = BN * BM * 4;
      float* const tile_base = p.sk_part + (size_t)t * p.ksplit * (BN * BM);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tile_base, (short)0, p.ksplit * TILE_BYTES, 0x00020000);
      const int lane_off = (wave * NF * NT * 64 + lane) * 16;
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < NT; ++m)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[n][m]), rs, ksp * TILE_BYTES + lane_off + (n * NT + m) * 1024, 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.sk_cnt + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sk_last = (old == (unsigned)p.ksplit - 1u) ? 1u : 0u;
        if (old == (unsigned)p.ksplit - 1u) __hip_atomic_store(p.sk_cnt + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again when the launch ends
      }
      __syncthreads();
#ifdef AKI_LAB_HOOKS
      if ((int)blockIdx.x == p.probe_block && tid == 0 && p.clock_probe) {
        p.clock_probe[0] = clock64() - probe_c0; p.clock_probe[1] = wall_clock64() - probe_w0;
        p.clock_probe[18] = probe_l0 - probe_c0; p.clock_probe[19] = clock64() - probe_l1; p.clock_probe[22] = sk_last;
      }
#endif
      if (sk_last == 0u) { AKI_WG_STAMP(3); return; }
      // The fold: sum over the slices IN SLICE ORDER, ((p0 + p1) + p2) ... - with this workgroup's own partial taken from its registers (the
      // same bits it stored) and, for the splits the planner hands out (2 and 3), every other slice's loads in flight at once: one memory
      // round trip instead of one per pair of slices (the fold was ~6 us of a 30 us launch, tools/small_m_timeline.py).
      auto load_slice = [&](int i, f32x4 (&dst)[NF][NT]) {
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int m = 0; m < NT; ++m)
            dst[n][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, i * TILE_BYTES + lane_off + (n * NT + m) * 1024, 0, 16));
      };
      if (p.ksplit == 2) {
        f32x4 other[NF][NT];
        load_slice(1 - ksp, other);
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int m = 0; m < NT; ++m) acc[n][m] += other[n][m];               // p0 + p1 either way round: the addition commutes
      } else if (p.ksplit == 3) {
        f32x4 pa[NF][NT], pb[NF][NT];                                           // the two other slices, in slice order
        load_slice(ksp == 0 ? 1 : 0, pa);
        load_slice(ksp == 2 ? 1 : 2, pb);
        if (ksp == 2) {                                                         // (p0 + p1) + own
#pragma unroll
          for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int m = 0; m < NT; ++m) acc[n][m] = (pa[n][m] + pb[n][m]) + acc[n][m];
        } else {                                                                // (own + p1) + p2  or  (p0 + own) + p2
#pragma unroll
          for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int m = 0; m < NT; ++m) acc[n][m] = (acc[n][m] + pa[n][m]) + pb[n][m];
        }
      } else {
        for (int i = 0; i < p.ksplit; ++i) {
#pragma unroll
          for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int m = 0; m < NT; ++m) {
              const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, i * TILE_BYTES + lane_off + (n * NT + m) * 1024, 0, 16));
              if (i == 0) acc[n][m] = v;
              else acc[n][m] += v;
            }
        }
      }
    }
  }
  if constexpr (FP8) {   // dequantise: acc[feature][token] *= sw[weight row] * sx[token]
    float sxm[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) sxm[m] = p.sx[min(m0 + wm * WTOK + m * 16 + l15, p.M - 1)];
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      int wrow;
      if (EPI == EPI_SWIGLU) {
        const int f = n0 + wn * (WROWS / 2) + (n % (NF / 2)) * 16 + 4 * kg;
        wrow = (n < NF / 2) ? min(f, n_out - 4) : n_out + min(f, n_out - 4);
      } else if (EPI == EPI_QKV_ROPE8) {
        const int u = min(n0 / 32 + wn * (NF / 2) + (n % (NF / 2)), 9 * p.H - 1);
        wrow = min(qkv_unit_row(u, n / (NF / 2), p.H) + 4 * kg, p.N - 4);
      } else {
        wrow = min(n0 + wn * WROWS + n * 16 + 4 * kg, p.N - 4);
      }
      const f32x4 s4 = *(const f32x4*)(p.sw + wrow);
#pragma unroll
      for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][m][r] *= s4[r] * sxm[m];
    }
  }

  // ---- epilogue: lane (token = l15 of block m, row group kg) owns features 4kg..4kg+3 of every feature block ----
  if (EPI == EPI_QKV_ROPE8) {
    constexpr int UW = NF / 2;
    const int u0 = n0 / 32 + wn * UW, nqk = 6 * p.H;
    // cos/sin of the tile's BM token rows go through LDS (idle after the K loop): fetched from the tables every lane
    // would issue 2 x 16 B per (token block, unit) - 32 loads per lane, 256 KB per workgroup through a 64 B/clk L1, ~6 us
    // of the launch at the benchmark shape.  Staged, a token row's first halves (cos[d] = cos[d + 48]) are loaded once by
    // two threads, 12 x 16 B each, and read back with ds_read_b128 (row pitch 400 B: the 16 token rows of a read land in
    // 16 different bank groups).  v-only tiles (u0 >= nqk for the whole workgroup) skip all of it.
    constexpr int CSROW = 400;
    static_assert(EPI != EPI_QKV_ROPE8 || BM * CSROW <= NST * STAGE_BYTES, "cos/sin staging fits the K-loop buffers");
    const bool tile_has_rope = n0 / 32 < nqk;                 // workgroup-uniform
    if (cs_pf) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my pieces of the prefetched cos / sin image have landed ...
      __syncthreads();                                        // ... and so have everybody else's
    } else if (tile_has_rope) {
      __syncthreads();                                        // every wave is done reading the K-loop stages
      for (int tk = tid >> 1; tk < BM; tk += (NWAVES * 64) >> 1) {
        const int mr = min(m0 + tk, p.M - 1) + p.m_offset;
        const int pos = p.position_ids ? p.position_ids[mr] : mr % p.L;
        const int half = tid & 1;
        const float* cp = p.cos + (size_t)pos * 96 + half * 24;
        const float* sp = p.sin + (size_t)pos * 96 + half * 24;
        char* dst = smem + tk * CSROW + half * 96;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          *(f32x4*)(dst + i * 16) = *(const f32x4*)(cp + 4 * i);
          *(f32x4*)(dst + 192 + i * 16) = *(const f32x4*)(sp + 4 * i);
        }
      }
      __syncthreads();
    }
#ifdef AKI_LAB_HOOKS
    long long probe_s = 0;
    if ((int)blockIdx.x == p.probe_block && tid == 0 && p.clock_probe) probe_s = clock64();
#endif
    // Units outside, token blocks inside: what a unit is (q / k with RoPE or v, its head, its plane of the output) is wave-uniform and
    // decided once, and the NT token blocks under it are one basic block - their cos / sin reads, the rotation and the stores overlap.
    // (Token blocks outside, every (block, unit) pair was a basic block of its own behind two uniform branches: read, wait, rotate, store,
    // 16 times over - 13.8 k of the epilogue's cycles; tools/gemm_epilogue_probe.py.)
    // Per token row: where it lives in a q plane (L rows per batch and head) and in a k / v plane (kvcap rows), as element offsets - the
    // stores below then add a wave-uniform plane base and nothing else.
    size_t offq[NT], dkv[NT];
    int csoff[NT];
    float rsq[NT];
    bool okq[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      const int mrow = m0 + wm * WTOK + m * 16 + l15;
      okq[m] = mrow < p.M;
      const int mr = min(mrow, p.M - 1) + p.m_offset;   // global token index
      const int b = mr / p.L, tt = mr - b * p.L;
      offq[m] = ((size_t)b * ((size_t)p.H * p.L) + tt) * 96;
      dkv[m] = (size_t)b * ((size_t)p.H * (p.kvcap - p.L)) * 96;   // k / v plane offset = offq + dkv
      csoff[m] = (wm * WTOK + m * 16 + l15) * CSROW + 16 * kg;
      rsq[m] = p.row_scale ? p.row_scale[min(mrow, p.M - 1)] : 1.0f;   // folded RMSNorm: 1 / rms of the token's hidden state
    }
    const __attribute__((address_space(3))) char* const lcs = (const __attribute__((address_space(3))) char*)AKI_LDS_PTR(smem);
    // FULLM: every token row of the tile exists - no exec-mask branch around the stores, a unit's token blocks are one basic block
    auto rope_rows = [&](auto fullm_) {
      constexpr bool FULLM = decltype(fullm_)::value;
#pragma unroll
      for (int q = 0; q < UW; ++q) {
        const int u = u0 + q;                             // wave-uniform
        if (u >= 9 * p.H) continue;
        // the lane's two 8-byte runs (block d1 and its partner block d2) become one 16-byte run of either block:
        // lane row kg ends up with features 8*(kg>>1) .. +7 of block (kg & 1 ? d2 : d1) - half the store instructions
        if (u < nqk) {
          const int hs = u / 3, j = u - 3 * hs, which = hs / p.H, head = hs - which * p.H;
          bf16_t* const plane = which == 0 ? p.q_out + (size_t)head * p.L * 96 : p.k_out + (size_t)head * p.kvcap * 96;
          const int d1 = 16 * j + 4 * kg, d2 = d1 + 48;
          const int dw = ((kg & 1) ? d2 : d1) - 4 * kg + 8 * (kg >> 1);
#pragma unroll
          for (int m = 0; m < NT; ++m) {
            const f32x4 c4 = *(const __attribute__((address_space(3))) f32x4*)(lcs + csoff[m] + 64 * j);
            const f32x4 s4 = *(const __attribute__((address_space(3))) f32x4*)(lcs + csoff[m] + 192 + 64 * j);
            float v1[4], v2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float x1 = acc[q][m][r] * rsq[m], x2 = acc[q + UW][m][r] * rsq[m];
              v1[r] = x1 * c4[r] - x2 * s4[r];
              v2[r] = x2 * c4[r] + x1 * s4[r];
            }
            const u32x4 o = pair_to_wide(pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3]), pack_bf16x2(v2[0], v2[1]), pack_bf16x2(v2[2], v2[3]));
            bf16_t* dst = plane + offq[m] + (which == 0 ? (size_t)0 : dkv[m]) + dw;
            if (FULLM || okq[m]) *(u32x4*)dst = o;
          }
        } else {
          const int u2 = u - nqk, vh = u2 / 3, j2 = u2 - 3 * vh;
          bf16_t* const plane = p.v_out + (size_t)vh * p.kvcap * 96;
          const int d1 = 32 * j2 + 4 * kg, d2 = d1 + 16;
          const int dw = ((kg & 1) ? d2 : d1) - 4 * kg + 8 * (kg >> 1);
#pragma unroll
          for (int m = 0; m < NT; ++m) {
            float v1[4], v2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v1[r] = acc[q][m][r] * rsq[m]; v2[r] = acc[q + UW][m][r] * rsq[m]; }
            const u32x4 o = pair_to_wide(pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3]), pack_bf16x2(v2[0], v2[1]), pack_bf16x2(v2[2], v2[3]));
            bf16_t* dst = plane + offq[m] + dkv[m] + dw;
            if (FULLM || okq[m]) *(u32x4*)dst = o;
          }
        }
      }
    };
    if (m0 + BM <= p.M) rope_rows(std::true_type{});
    else rope_rows(std::false_type{});
#ifdef AKI_LAB_HOOKS
    if ((int)blockIdx.x == p.probe_block && tid == 0 && p.clock_probe) {
      p.clock_probe[0] = clock64() - probe_c0;
      p.clock_probe[1] = wall_clock64() - probe_w0;
      p.clock_probe[18] = probe_l0 - probe_c0;
      p.clock_probe[19] = clock64() - probe_l1;
      p.clock_probe[20] = probe_s - probe_l1;            // cos / sin staging
    }
#endif
    AKI_WG_STAMP(3);
    return;
  }

  if constexpr (!EARLY_OPERANDS) fetch_epilogue_operands();
  // The residual tile goes through LDS (idle after the K loop), like cos/sin in the QKV epilogue: fetched per lane it
  // is NT*NOUT 8-byte loads (32 on the big tile, 128 KB per workgroup through the L1); staged it is 16-byte loads of whole
  // rows, two to sixteen per thread.  Chunk c of token row t sits at chunk c ^ (t & CMASK): the 16 token rows of a
  // ds_read_b64 land in 16 different bank groups without padding (a padded 256 x 256 tile would not fit).
  const bool res_lds = (EPI == EPI_PLAIN) && p.residual != nullptr && p.res_wide;   // workgroup-uniform
  if constexpr (PIPE == 2 || PIPE == 5 || PIPE == 6) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my pieces of the prefetched residual tile have landed ...
    __syncthreads();                                           // ... and so have everybody else's
  } else if (RES_PF_OK && res_pf) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else if (EPI == EPI_PLAIN && res_lds) {
    static_assert(EPI != EPI_PLAIN || BM * BN * 2 <= NST * STAGE_BYTES, "residual tile fits the K-loop buffers");
    __syncthreads();                                           // every wave is done reading the K-loop stages
    for (int c = tid; c < BM * RCH; c += NWAVES * 64) {
      const int tk = c / RCH, ch = c - tk * RCH;
      const int f = n0 + 8 * ch;
      if (f < n_out) {
        const int mr = min(m0 + tk, p.M - 1);
        const bf16_t* rrow = p.residual + (size_t)(p.res_row_mod > 0 ? (mr + p.m_offset) % p.res_row_mod : mr) * p.ldr;
        *(u32x4*)(smem + (tk * RCH + (ch ^ (tk & CMASK))) * 16) = *(const u32x4*)(rrow + f);
      }
    }
    __syncthreads();
  }
  const bool stats = (EPI == EPI_PLAIN) && p.st_part != nullptr;         // kernel-uniform
  // The rows of the tile.  Each of the switches (tile inside the output, bias, residual, statistics, row scale, row shift) is
  // 0 = no, 1 = yes or 2 = decided at run time.  With all of them decided at run time - the general path - the per-lane predicates put an
  // exec-mask branch around every bias add, residual read and store (~20 basic blocks per token block), the residual pointer is LDS-or-
  // global and so a FLAT load with a vmcnt(0) + lgkmcnt(0) wait behind each of the 32 reads (draining the stores in front), and the
  // statistics arithmetic is speculated into launches that do not ask for it: 9-14 k cycles per tile, uncontended (tools/
  // gemm_epilogue_probe.py).  The combinations the model launches on interior tiles are compiled as straight-line variants instead.
  auto write_rows = [&](auto full_, auto hb_, auto hr_, auto hs_, auto hsc_, auto hsh_) {
    constexpr int FULLK = decltype(full_)::value, HB = decltype(hb_)::value, HR = decltype(hr_)::value, HS = decltype(hs_)::value,
                  HSC = decltype(hsc_)::value, HSH = decltype(hsh_)::value;
    constexpr bool FULL = FULLK == 1;
    const bool has_bias = EPI == EPI_PLAIN && (HB == 2 ? p.bias != nullptr : HB == 1);
    const bool has_res = HR == 2 ? p.residual != nullptr : HR == 1;
    const bool in_lds = HR == 2 ? res_lds : true;               // HR == 1: the dispatch below only takes staged residual tiles
    const bool has_stats = EPI == EPI_PLAIN && (HS == 2 ? stats : HS == 1);
    const bool has_shift = EPI == EPI_PLAIN && (HSH == 2 ? shifted : HSH == 1);
    const bool wide = FULL || (NOUT % 2 == 0 && p.wide);
    const __attribute__((address_space(3))) char* const lres = (const __attribute__((address_space(3))) char*)AKI_LDS_PTR(smem) + res_off;
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      const int tk = wm * WTOK + m * 16 + l15;
      const int mrow = m0 + tk;
      const bool ok = FULL || mrow < p.M;
      const int mr = FULL ? mrow : min(mrow, p.M - 1);
      bf16_t* yrow = p.y + (size_t)mr * p.ldy;
      const bf16_t* rrow = nullptr;
      if (HR == 2 && has_res) rrow = p.residual + (size_t)(p.res_row_mod > 0 ? (mr + p.m_offset) % p.res_row_mod : mr) * p.ldr;
      const float rs = rsv[m], mu = muv[m];
      float v[NOUT][4];
      unsigned gpk[NOUT][2], upk[NOUT][2];       // EPI_SWIGLU with preact_out: the packed pre-activations of this token row
      u32x2 rr[NOUT];
      if (HR == 1) {                                             // all of the token block's residual reads first: one wait for the lot
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
          const int ch = (wn * WROWS + n * 16) / 8 + (kg >> 1);
          rr[n] = *(const __attribute__((address_space(3))) u32x2*)(lres + (tk * RCH + (ch ^ (tk & CMASK))) * 16 + (kg & 1) * 8);
        }
      }
#pragma unroll
      for (int n = 0; n < NOUT; ++n) {
        const int f = fwave + n * 16 + 4 * kg;       // this lane's 4 features of block n
        const bool fin = FULL || f < n_out;
        float pre_g[4], pre_u[4];                    // EPI_SWIGLU: what the activation is taken of
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (EPI == EPI_SWIGLU) {
            const float g = HSC == 0 ? acc[n][m][r] : acc[n][m][r] * rs, u = HSC == 0 ? acc[n + NF / 2][m][r] : acc[n + NF / 2][m][r] * rs;
            v[n][r] = u * silu_fast(g);
            pre_g[r] = g; pre_u[r] = u;
          } else if (has_shift) {
            v[n][r] = (acc[n][m][r] - mu * colc4[n][r]) * rs;
          } else {
            v[n][r] = HSC == 0 ? acc[n][m][r] : acc[n][m][r] * rs;
          }
        }
        if (EPI == EPI_SWIGLU) {
          // Training forward (preact_out): the pre-activations leave as bf16 for the backward pass, and the activation is taken of
          // THOSE values with the training kernels' own silu - the step computes what aki_linear + aki_swiglu_fwd computed, bit for
          // bit, without the second pass over [M, 2 F].
          if (p.preact) {
            const unsigned g01 = pack_bf16x2(pre_g[0], pre_g[1]), g23 = pack_bf16x2(pre_g[2], pre_g[3]);
            const unsigned u01 = pack_bf16x2(pre_u[0], pre_u[1]), u23 = pack_bf16x2(pre_u[2], pre_u[3]);
            v[n][0] = bf16_lo(u01) * silu(bf16_lo(g01)); v[n][1] = bf16_hi(u01) * silu(bf16_hi(g01));
            v[n][2] = bf16_lo(u23) * silu(bf16_lo(g23)); v[n][3] = bf16_hi(u23) * silu(bf16_hi(g23));
            gpk[n][0] = g01; gpk[n][1] = g23; upk[n][0] = u01; upk[n][1] = u23;
          }
        }
        if (EPI == EPI_PLAIN) {
          if (has_bias && fin) {
            v[n][0] += bf16_lo(biasp[n][0]); v[n][1] += bf16_hi(biasp[n][0]); v[n][2] += bf16_lo(biasp[n][1]); v[n][3] += bf16_hi(biasp[n][1]);
          }
          if (ACT == AKI_ACT_GELU_ERF) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[n][r] = gelu_erf_fast(v[n][r]);
          } else if (ACT == AKI_ACT_GELU_TANH) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[n][r] = gelu_tanh_fast(v[n][r]);
          }
        }
        if (HR == 2 && has_res && fin) {
          if (in_lds) {
            const int ch = (wn * WROWS + n * 16) / 8 + (kg >> 1);
            rr[n] = *(const __attribute__((address_space(3))) u32x2*)(lres + (tk * RCH + (ch ^ (tk & CMASK))) * 16 + (kg & 1) * 8);
          } else {
            rr[n] = *(const u32x2*)(rrow + f);
          }
        }
        if (has_res && fin) {
          v[n][0] += bf16_lo(rr[n][0]); v[n][1] += bf16_hi(rr[n][0]); v[n][2] += bf16_lo(rr[n][1]); v[n][3] += bf16_hi(rr[n][1]);
        }
      }
      if (EPI == EPI_SWIGLU) {
        if (p.preact) {         // the pre-activations leave like y does: 16-byte stores that pair two feature blocks where the layout allows
          bf16_t* const prow = p.preact + (size_t)mr * p.ld_preact;
          const bool pwide = NOUT % 2 == 0 && n_out % 8 == 0 && p.ld_preact % 8 == 0 && (((uintptr_t)p.preact) & 15) == 0;
          if (pwide) {
#pragma unroll
            for (int n = 0; n + 1 < NOUT; n += 2) {
              const u32x4 og = pair_to_wide(gpk[n][0], gpk[n][1], gpk[n + 1][0], gpk[n + 1][1]);
              const u32x4 ou = pair_to_wide(upk[n][0], upk[n][1], upk[n + 1][0], upk[n + 1][1]);
              const int f = fwave + (n + (kg & 1)) * 16 + 8 * (kg >> 1);
              if (ok && (FULL || f < n_out)) { *(u32x4*)(prow + f) = og; *(u32x4*)(prow + n_out + f) = ou; }
            }
          } else {
#pragma unroll
            for (int n = 0; n < NOUT; ++n) {
              const int f = fwave + n * 16 + 4 * kg;
              if (ok && f < n_out) { *(u32x2*)(prow + f) = u32x2{gpk[n][0], gpk[n][1]}; *(u32x2*)(prow + n_out + f) = u32x2{upk[n][0], upk[n][1]}; }
            }
          }
        }
      }
      unsigned pk[NOUT][2];                                      // the values AS STORED
#pragma unroll
      for (int n = 0; n < NOUT; ++n) { pk[n][0] = pack_bf16x2(v[n][0], v[n][1]); pk[n][1] = pack_bf16x2(v[n][2], v[n][3]); }
      if (has_stats) {
        // producer: this wave's share of the token's sum / sum of squares over the tile's features, of the values AS STORED
        // (bf16); the four lanes that hold a token (kg = 0..3) fold theirs, lane kg == 0 writes through (sc1) for the reducer
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
          const bool fin = FULL || fwave + n * 16 + 4 * kg < n_out;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned w_ = pk[n][r >> 1];
            const float vb = fin ? ((r & 1) ? bf16_hi(w_) : bf16_lo(w_)) : 0.f;
            s1 += vb;
            s2 += vb * vb;
          }
        }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (kg == 0 && ok) {   // one 8-byte write-through store: {sum of squares, sum}
          unsigned long long* pp = (unsigned long long*)p.st_part + (size_t)(tn * WN + wn) * p.M + mrow;
          const unsigned long long both = (unsigned long long)__float_as_uint(s2) | ((unsigned long long)__float_as_uint(s1) << 32);
          __hip_atomic_store(pp, both, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // (A 16-byte store out of the fragment layout covers 16 rows x 64 bytes; alone on the chip such stores leave a CU at 74 cycles apiece
      // against 20 for 4 rows x 256 bytes - tools/store_tail_probe.hip - but turning the wave's rows through an LDS patch to get that shape
      // bought nothing at the model's shapes: the tail of a full round is bound by the write-back of 32 MB, and in a multi-round launch it
      // drains under the next workgroup's K loop; profiles/r03_store_tail_probe.txt.)
      if (wide) {   // 16-byte stores pair two feature blocks
#pragma unroll
        for (int n = 0; n + 1 < NOUT; n += 2) {
          const u32x4 o = pair_to_wide(pk[n][0], pk[n][1], pk[n + 1][0], pk[n + 1][1]);
          const int f = fwave + (n + (kg & 1)) * 16 + 8 * (kg >> 1);
          if (ok && (FULL || f < n_out)) *(u32x4*)(yrow + f) = o;
        }
      } else {
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
          const int f = fwave + n * 16 + 4 * kg;
          if (ok && f < n_out) {
            u32x2 o = {pk[n][0], pk[n][1]};
            *(u32x2*)(yrow + f) = o;
          }
        }
      }
    }
  };
  {
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using c2 = std::integral_constant<int, 2>;
    const bool inside = m0 + BM <= p.M && n0 + (EPI == EPI_SWIGLU ? BN / 2 : BN) <= n_out;
    const bool fast = inside && NOUT % 2 == 0 && p.wide && (p.residual == nullptr || (EPI == EPI_PLAIN && res_lds));   // workgroup-uniform
    const int key = (EPI == EPI_PLAIN && p.bias ? 1 : 0) | (p.residual ? 2 : 0) | (stats ? 4 : 0) | (p.row_scale ? 8 : 0) | (shifted ? 16 : 0);
    bool done = false;
    if (fast) {
      done = true;
      if (key == 0) write_rows(c1{}, c0{}, c0{}, c0{}, c0{}, c0{});                    // plain
      else if (key == 8) write_rows(c1{}, c0{}, c0{}, c0{}, c1{}, c0{});               // folded RMSNorm (gate_up, lm_head)
      else if constexpr (EPI == EPI_PLAIN) {
        if (key == 1) write_rows(c1{}, c1{}, c0{}, c0{}, c0{}, c0{});                  // bias
        else if (key == 2) write_rows(c1{}, c0{}, c1{}, c0{}, c0{}, c0{});             // residual
        else if (key == 3) write_rows(c1{}, c1{}, c1{}, c0{}, c0{}, c0{});             // bias + residual
        else if (key == 6) write_rows(c1{}, c0{}, c1{}, c1{}, c0{}, c0{});             // residual + statistics (o_proj, down)
        else if (key == 7) write_rows(c1{}, c1{}, c1{}, c1{}, c0{}, c0{});             // + bias (SigLIP out / fc2)
        else if constexpr (PIPE <= 1 || PIPE == 4 || PIPE == 7 || PIPE == 8) {
          if (key == 25) write_rows(c1{}, c1{}, c0{}, c0{}, c1{}, c1{});               // folded LayerNorm + bias (SigLIP qkv / fc1)
          else done = false;
        } else done = false;
      } else done = false;
    }
    if (!done) write_rows(c0{}, c2{}, c2{}, c2{}, c2{}, c2{});
  }
  if (EPI == EPI_PLAIN && stats) {
    // Row statistics: the LAST tile of a row panel to get here folds the panel's partial sums in a fixed order (the result
    // does not depend on which tile that is) and writes 1/rms (and the mean) of the panel's rows.  Hand-off as in
    // cdna_hip_programming.md Guideline 16 (R1 / ticket form): write-through partials, every storing wave drains them,
    // barrier, one lane draws a ticket; the reducer reads the partials with agent-scope loads.  It puts the counter back
    // to zero - the counters are zero again when the launch ends.
    // (Drawing the ticket AHEAD of the tile's output stores, so that its round trip runs under their issue-bound tail, was measured and is
    // slower: the write-through partials then drain alone, 21.8 k against 16.9 k epilogue cycles; profiles/r03_gemm_epilogue_probe.txt.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* const flag = (unsigned*)smem;
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(p.st_cnt + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = (old == (unsigned)p.tiles_n - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (*flag != 0u) {
      // All (slot, row) pairs of the panel are fetched in parallel (agent-scope loads: the writers sit behind other L2s) and
      // parked in the K-loop buffers, which nobody reads any more; then one thread per row adds its column top to bottom.
      const int slots = p.tiles_n * WN;
      constexpr int NTHR = NWAVES * 64;
      constexpr int CH = (NST * STAGE_BYTES - 16) / (BM * 8);          // slots per pass through LDS
      static_assert(CH >= 8, "statistics staging");
      unsigned long long* const park = (unsigned long long*)(smem + 16);
      const unsigned long long* const part = (const unsigned long long*)p.st_part;
      float s1 = 0.f, s2 = 0.f;
      for (int sl0 = 0; sl0 < slots; sl0 += CH) {
        const int total = min(CH, slots - sl0) * BM;
        for (int it0 = tid; it0 < total; it0 += 8 * NTHR) {
          unsigned long long got[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int it = it0 + u * NTHR;
            const int sl = it / BM, row = it - sl * BM;
            got[u] = 0ull;
            if (it < total && m0 + row < p.M)
              got[u] = __hip_atomic_load(part + (size_t)(sl0 + sl) * p.M + m0 + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int it = it0 + u * NTHR;
            if (it < total) park[it] = got[u];
          }
        }
        __syncthreads();
        if (tid < BM) {
          for (int sl = 0; sl < total / BM; ++sl) {
            unsigned long long both = park[sl * BM + tid];
            // Explicit wait: with hipcc's own lgkmcnt for this read (merged into ds_read2st64_b64 on the 96- and 64-row tiles) the SUM half of one slot was
            // consumed before it had arrived - lanes 48..63, the last quarter the LDS returns - in 0.5 % of LayerNorm-statistics launches at long K
            // (tools/stats_stress_modes.py; outputs and the partials in memory were right every time).  The fold runs once per row panel: the wait is free.
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(both));
            s2 += __uint_as_float((unsigned)both);
            s1 += __uint_as_float((unsigned)(both >> 32));
          }
        }
        __syncthreads();
      }
      if (tid < BM && m0 + tid < p.M) {
        const int mrow = m0 + tid;
        const float inv_n = 1.0f / (float)n_out;
        const float mean = s1 * inv_n;
        if (p.st_mean) {        // LayerNorm statistics
          p.st_mean[mrow] = mean;
          p.st_rstd[mrow] = rsqrtf(fmaxf(s2 * inv_n - mean * mean, 0.f) + p.st_eps);
        } else {                // RMSNorm
          p.st_rstd[mrow] = rsqrtf(s2 * inv_n + p.st_eps);
        }
      }
      if (tid == 0) __hip_atomic_store(p.st_cnt + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef AKI_LAB_HOOKS
  if ((int)blockIdx.x == p.probe_block && tid == 0 && p.clock_probe) {
    p.clock_probe[0] = clock64() - probe_c0;
    p.clock_probe[1] = wall_clock64() - probe_w0;
    p.clock_probe[18] = probe_l0 - probe_c0;
    p.clock_probe[19] = clock64() - probe_l1;
  }
#endif
  AKI_WG_STAMP(3);
}
#undef AKI_WG_STAMP

#ifdef AKI_LAB_HOOKS
int g_sk_slice_major = 0;                            // set by aki_lab_set_slice_major (lab A/B of the split-K workgroup order)
int g_sm_variant = -1, g_sm_ksplit = 1;              // set by aki_lab_set_small_m: force a small-M tile variant (launch_variant) and its K split
int g_force_tile = 0, g_deep_ring = 1, g_pipe = 1, g_deepx = 0;   // set by aki_lab_set_gemm_tile (lab build only); g_pipe: 0 off, 1 on, 2 on without the residual prefetch
long long* g_clock_probe = nullptr;                  // set by aki_lab_set_clock_probe
int g_probe_block = 0;                               // set by aki_lab_set_probe_block
#else
static constexpr int g_force_tile = 0, g_deep_ring = 1, g_pipe = 1, g_deepx = 0, g_sm_variant = -1, g_sm_ksplit = 1, g_sk_slice_major = 0;
#endif

template <int NF, int NT, int WN, int WM, int EPI, int ACT = 0, bool FP8 = false, int NST = 2, int PIPE = 0, int SK = 0, int KG = 1>
static int launch_gemm(GemmParams& p, hipStream_t stream) {
  constexpr int BN = WN * NF * 16, BM = WM * NT * 16;
  constexpr int SMEM = (PIPE >= 4 && PIPE <= 7) ? (BN > BM ? 3 * BN + 2 * BM : 3 * BM + 2 * BN) * 128   // PIPE 4-7: three + two tiles
                       : (PIPE == 8 ? (2 * BN + 4 * BM) * 128 : KG * NST * (BN + BM) * 128);                 // PIPE 8: two weight, three token slots + a spare
  static_assert(SMEM <= 160 * 1024, "LDS");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_bf16_kernel<NF, NT, WN, WM, EPI, ACT, FP8, NST, PIPE, SK, KG>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return AKI_ERR_LAUNCH;
    attr_set = true;
  }
  const int n_out = (EPI == EPI_SWIGLU) ? p.N / 2 : p.N;
  const int bn_out = (EPI == EPI_SWIGLU) ? BN / 2 : BN;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (n_out + bn_out - 1) / bn_out;
  AKI_CLEAR_ERR();
#ifdef AKI_LAB_HOOKS
  p.clock_probe = g_clock_probe;
  p.probe_block = g_probe_block;
#endif
  const int slices = SK ? (p.ksplit > 1 ? p.ksplit : (p.ksplit = 1)) : 1;
  p.sk_slice_major = g_sk_slice_major;
  if (KG > 1 && (p.K / 64) % KG) return AKI_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((gemm_bf16_kernel<NF, NT, WN, WM, EPI, ACT, FP8, NST, PIPE, SK, KG>), dim3(p.tiles_m * p.tiles_n * slices), dim3(WN * WM * 64 * KG), SMEM, stream, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

// g_force_tile: 0 = heuristic, 1 = 256^2, 2 = 128^2, 3 = 128 features x 96 tokens, 5 = 64 x 64 on a four-stage ring (plain bf16 only), 4 = lab loop

// Cost model in units of one 256x256 tile's run time.  256^2 tiles: one workgroup per CU.  128-token tiles do a
// quarter of the work; up to 256 of them run one per CU at ~75 % of the big tile's efficiency, beyond that two share a CU
// and overlap each other's prologue / epilogue (~90 %: tools/siglip_gemm_bench.py - SigLIP fc1, 1224 small tiles in three
// half-rounds, 903 TF/s vs 696 for 306 big tiles in two rounds at 60 % occupancy).
// A last round that fills at most half the CUs runs faster than a full one - the part is power-bound on these tiles, and the
// busy half gets the clock and the fabric of the idle half: 384 tiles of a K = 5248 weight-gradient GEMM take 1.65 full rounds,
// not 2 (209 us against 127 us per round and 239 us for the 128^2 tiling the whole-round model preferred).  Only claimed for
// long K loops (>= 64 steps): with SigLIP's 18 steps the fixed costs of a round dominate and the discount is not there.
static double cost_big(long tiles, int ksteps = 0) {
  const long full = tiles / 256, rem = tiles % 256;
  if (rem == 0) return (double)full;
  return (double)full + ((ksteps >= 64 && full >= 1 && rem <= 128) ? 0.65 : 1.0);
}
// `work` = small-tile work relative to the big tile (0.25 for 128x128 vs 256x256, 0.5 for the 192x128 QKV tile).
static double cost_small(long tiles, double work) {
  return tiles <= 256 ? (tiles ? work / 0.75 : 0.0) : (double)((tiles + 511) / 512) * (2.0 * work / 0.9);
}

// plan: 0 = all big, 1 = all small, 2 = big on the first floor(M/256)*256 rows + small tiles on the M tail
// (removes the wave-quantisation loss of a last, mostly idle round: 1344 tiles on 256 CUs = 5.25 rounds).
// (A 128-feature x 256-token tile was measured too: never better than the 128^2 tiles on any AKI shape.)
// plan 3 (plain bf16 GEMMs only, `mid_ok`): 128-feature x 96-token tiles when they fit ONE round of the 2-per-CU slots and
// the 128^2 tiling would leave that round partly empty - SigLIP out-proj / fc2 (N = 1152: 324 -> 432 tiles, 21.7 -> 17.9 us
// and 61.6 -> 50.5 us), Perceiver kv (360 -> 480 tiles, 20.4 -> 17.2 us).  Same K order per output element: bit-identical.
static int plan_tiles(int M, int n_out, int bn_big, int bn_small, double small_work = 0.25, bool mid_ok = false, int ksteps = 0) {
  if (g_force_tile) return g_force_tile == 2 ? 1 : (g_force_tile == 3 ? (mid_ok ? 3 : 1) : (g_force_tile == 5 ? (mid_ok ? 4 : 1) : 0));
  const long nb = (n_out + bn_big - 1) / bn_big, ns = (n_out + bn_small - 1) / bn_small;
  const double all_big = cost_big((long)((M + 255) / 256) * nb, ksteps);
  const double all_small = cost_small((long)((M + 127) / 128) * ns, small_work);
  const int m_main = M / 256 * 256, tail = M - m_main;
  double split = 1e30;
  if (m_main > 0 && tail > 0) split = cost_big((long)(m_main / 256) * nb) + cost_small((long)((tail + 127) / 128) * ns, small_work);
  if (mid_ok) {
    const long nm = (long)((M + 95) / 96) * ns;
    // plan 4: a few rows against a long K (Perceiver ff2: 1152 x 1152 x 4608 = 108 tiles of 128 x 96 on 256 CUs, 72 K-steps each) - 64 x 64 tiles on a
    // four-stage ring put 324 workgroups on the chip, two per CU: 43.2 -> 28.4 us (tools/perceiver_gemm_ab.py); same K order per output element: bit-identical.
    if (M > 128 && nm <= 128 && ksteps >= 32 && (long)((M + 63) / 64) * ((n_out + 63) / 64) <= 512) return 4;   // (M <= 128: the weight-streaming tiles of launch_small)
    if (nm <= 512) {
      const double all_mid = cost_small(nm, 0.75 * small_work);
      if (all_mid < all_small && all_mid < all_big && all_mid < split) return 3;
    }
  }
  if (all_small < all_big && all_small <= split) return 1;
  if (split < all_big) return 2;
  return 0;
}

// g_deep_ring (lab bit 8 clears it): 4-stage ring for sparse small-tile launches

// 128 x 128 tiles (two workgroups per CU), except for a single row of tiles - see below
template <int EPI, int ACT, bool FP8>
static int launch_small(GemmParams& p, hipStream_t stream) {
  if constexpr (!FP8) {
    const int n_out = (EPI == EPI_SWIGLU) ? p.N / 2 : p.N, bn_out = (EPI == EPI_SWIGLU) ? 64 : 128;
    const long tiles = (long)((p.M + 127) / 128) * ((n_out + bn_out - 1) / bn_out);
    // One row of tiles (an M tail of <= 128 tokens against a wide weight matrix) is a weight-streaming problem: it is
    // bound by how many CUs pull the stream and how many loads each keeps in flight.  64-feature tiles double the
    // workgroup count and the ring goes four stages deep: gate_up tail (120 x 16384 x 3072) 35.0 -> 23.5 us, lm_head
    // tail 46.8 -> 37.3 us.  With several rows of tiles (B=1 prefill: 6 x 24) the deep ring measured 4 % slower.
    if (g_deep_ring && p.M <= 128 && p.K / 64 >= 4) {
      if (EPI != EPI_QKV_ROPE8 && tiles <= 128) return launch_gemm<2, 4, 2, 2, EPI, ACT, FP8, 4>(p, stream);   // (six stages: 24.7 vs 22.8 us on the gate_up tail)
      if (tiles <= 256) return launch_gemm<4, 4, 2, 2, EPI, ACT, FP8, 4>(p, stream);
    }
  }
  return launch_gemm<4, 4, 2, 2, EPI, ACT, FP8>(p, stream);
}

// g_pipe (lab bit 9 clears it): mid-step barrier pipeline on the 256 x 256 tile

// Which operand of a residual GEMM gets the three-deep ring: the tokens when the activation panel is the larger cold operand (lab bit 12 forces it, bit 13 forbids it)
static bool deep_tokens(const GemmParams& p) { return g_deepx == 1 || (g_deepx == 0 && (long)p.M > 3L * p.N / 2 && p.K >= 4096); }

template <int EPI, int ACT, bool FP8>
static int launch_big(GemmParams& p, hipStream_t stream) {
  if constexpr (!FP8) {
#ifdef AKI_LAB_HOOKS
    // lab: the same 256 x 256 tile on FOUR waves, one per SIMD (wave tile 128 features x 128 tokens, 256 accumulator registers)
    if (g_force_tile == 4 && p.row_shift == nullptr) return launch_gemm<8, 8, 2, 2, EPI, ACT, FP8, 2, 3>(p, stream);
#endif
    if constexpr (EPI == EPI_PLAIN && ACT == 0) {   // residual tile prefetched under the last K-steps (o_proj, down_proj)
      if ((g_pipe == 1 || g_pipe == 3) && p.row_shift == nullptr && p.residual != nullptr && p.res_wide && p.res_row_mod <= 0 && p.N % 8 == 0 && p.N >= 8)
        return g_pipe == 3 ? launch_gemm<8, 4, 2, 4, EPI, ACT, FP8, 2, 2>(p, stream)
               : (deep_tokens(p) ? launch_gemm<8, 4, 2, 4, EPI, ACT, FP8, 2, 6>(p, stream) : launch_gemm<8, 4, 2, 4, EPI, ACT, FP8, 2, 5>(p, stream));
    }
    if (g_pipe == 3) return launch_gemm<8, 4, 2, 4, EPI, ACT, FP8, 2, 1>(p, stream);   // lab: two-deep weight ring
    if (g_pipe) return launch_gemm<8, 4, 2, 4, EPI, ACT, FP8, 2, 4>(p, stream);
  }
  return launch_gemm<8, 4, 2, 4, EPI, ACT, FP8>(p, stream);
}

// ---- small M (one-sample prefill: M = 655 / 207; one image through the SigLIP tower: M = 576) --------------------------------------
// Tile variants on the plain K loops (two stages, or a ring of NST stages with NST - 1 tiles in flight), each optionally with the K range split
// over `ksplit` workgroups (EPI_PLAIN without activation).  What these launches lack is not FLOPs but requests in flight: a 128 x 128 tile on
// the two-stage loop has ONE 32 KiB tile outstanding per workgroup and pays the memory latency every K-step (o_proj at M = 655: 144
// workgroups x 48 steps x ~2000 cycles = 46 us for 12 GFLOP).
//   id  features x tokens  stages  LDS      per CU
//   0   128 x 128          2       64 KiB   2          5   128 x 64   3   72 KiB   2
//   1   128 x 128          3       96 KiB   1          6   64 x 128   3   72 KiB   2
//   2   128 x 128          4       128 KiB  1          7   64 x 64    4   64 KiB   2
//   3   128 x 96           2       56 KiB   2          8   128 x 64   2   48 KiB   3
//   4   128 x 96           3       84 KiB   1          9   64 x 128   4   96 KiB   1
constexpr int kSmallMVariants = 21;
struct SmallMTile { int bn, bm, lds_kib; };
static const SmallMTile kSmallMTiles[kSmallMVariants] = {{128, 128, 64}, {128, 128, 96}, {128, 128, 128}, {128, 96, 56}, {128, 96, 84},
                                                        {128, 64, 72}, {64, 128, 72}, {64, 64, 64}, {128, 64, 48}, {64, 128, 96},
                                                        // 10-15: the mid-step-barrier pipeline (PIPE 1: fragments double-buffered in registers, reads and DMA issue spread between the MFMAs)
                                                        {128, 128, 64}, {128, 96, 56}, {256, 128, 96}, {128, 128, 64}, {256, 64, 80}, {128, 256, 96},
                                                        // 16-20: K groups inside the workgroup (KG x 4 waves, fold through LDS): 128 x 96 x 2, 128 x 128 x 2, 128 x 64 x 2 / x 3, 64 x 64 x 4
                                                        {128, 96, 112}, {128, 128, 128}, {128, 64, 96}, {128, 64, 144}, {64, 64, 128}};
size_t linear_splitk_cnt_bytes() { return size_t(64) << 10; }     // 16 384 tile tickets, at the front of the split-K workspace
size_t linear_splitk_ws_bytes(int variant, int ksplit, int M, int n_out) {
  if (ksplit <= 1 || variant < 0 || variant >= kSmallMVariants) return 0;
  const SmallMTile& tl = kSmallMTiles[variant];
  const size_t tiles = (size_t)((M + tl.bm - 1) / tl.bm) * ((n_out + tl.bn - 1) / tl.bn);
  return linear_splitk_cnt_bytes() + tiles * ksplit * tl.bn * tl.bm * sizeof(float);
}

// The planner's choice for a bf16 GEMM with few rows (17 .. 1024: a one-sample prefill, one image through the tower): tile variant (-1: none, the
// ordinary plans) and K split.  From tools/small_m_sweep.py on cold operands (profiles/r05_small_m_sweep.txt, us per launch, planner before -> now):
//   plain, N <= 4096, K >= 2304 (o_proj / down / SigLIP fc2): 128 x 96 (M > 256) or 128 x 64 tiles, K split 3 ways when the tiles alone
//     leave the 512 two-per-CU slots under-filled - M 655: o_proj 46.6 -> 37.2, down 97.2 -> 59.4; M 207: 27.1 -> 24.5, 47.1 -> 38.5; fc2 26.9 -> 22.7.
//     Deeper rings, 256-wide tiles and the register-pipelined loop on these tiles all lost: one 4-wave workgroup per CU serialises DMA wait,
//     fragment reads and MFMAs (1200-1450 cycles per K-step for 384-512 of MFMA work, warm or cold), two per CU overlap them, and the
//     launch's fixed costs (prologue 1.5 us, fold / statistics tails 3-15 us) are then as long as the K loop (tools/small_m_timeline.py).
//   QKV + RoPE: 128 x 96 tiles (M > 256: 57.8 -> 48.5) or the same on a three-stage ring (M 207: 46.5 -> 31.0).
//   gate_up + SwiGLU, M <= 256: 128 x 128 on a three-stage ring (47.4 -> 36.8); above that the 256 x 256 tile stays (76 us at M 655).
static void plan_small_m(int epi, int M, int n_out, int K, bool can_split, int& variant, int& ksplit) {
  variant = -1; ksplit = 1;
  if (M < 17 || M > 1024 || g_force_tile) return;
  const int nk = K / 64;
  if (epi == EPI_QKV_ROPE8) { variant = M > 256 ? 3 : 4; return; }
  if (epi == EPI_SWIGLU) { if (M <= 256 && nk >= 8) variant = 1; return; }
  if (n_out > 4096 || nk < 36 || !can_split) return;
  const int v = M > 256 ? 3 : 5;
  const SmallMTile& tl = kSmallMTiles[v];
  const long tiles = (long)((M + tl.bm - 1) / tl.bm) * ((n_out + tl.bn - 1) / tl.bn);
  const int ks = (int)(512 / tiles) < 3 ? (int)(512 / tiles) : 3;
  if (ks < 2) return;
  variant = v; ksplit = ks;
}
size_t linear_splitk_plan_ws_bytes(int M, int N, int K) {
  int v, ks;
  plan_small_m(EPI_PLAIN, M, N, K, true, v, ks);
  return linear_splitk_ws_bytes(v, ks, M, N);
}

template <int EPI, int ACT>
static int launch_variant(GemmParams& p, int variant, int ksplit, hipStream_t stream) {
  constexpr int SKV = (EPI == EPI_PLAIN && ACT == 0) ? 1 : 0;
  p.ksplit = SKV ? ksplit : 1;
  // Every variant exists for the plain GEMM (the one that can split K); the epilogue families that cannot (QKV + RoPE, SwiGLU, GELU) get the
  // variants whose numbers are on record for them (EXPERIMENTS.md, round 5) - each instantiation is 2-3 s of compile time in the lab object.
  constexpr bool ALL = EPI == EPI_PLAIN && ACT == 0, MOST = EPI != EPI_PLAIN;
  switch (variant) {
    case 0: return launch_gemm<4, 4, 2, 2, EPI, ACT, false, 2, 0, SKV>(p, stream);
    case 3: return launch_gemm<4, 3, 2, 2, EPI, ACT, false, 2, 0, SKV>(p, stream);
    case 4: return launch_gemm<4, 3, 2, 2, EPI, ACT, false, 3, 0, SKV>(p, stream);
    case 6: return launch_gemm<2, 4, 2, 2, EPI, ACT, false, 3, 0, SKV>(p, stream);
    case 16: p.ksplit = 1; return launch_gemm<4, 3, 2, 2, EPI, ACT, false, 2, 0, 0, 2>(p, stream);
  }
  if constexpr (ALL || MOST) {
    switch (variant) {
      case 1: return launch_gemm<4, 4, 2, 2, EPI, ACT, false, 3, 0, SKV>(p, stream);
      case 2: return launch_gemm<4, 4, 2, 2, EPI, ACT, false, 4, 0, SKV>(p, stream);
      case 5: return launch_gemm<4, 2, 2, 2, EPI, ACT, false, 3, 0, SKV>(p, stream);
      case 7: return launch_gemm<2, 2, 2, 2, EPI, ACT, false, 4, 0, SKV>(p, stream);
      case 8: return launch_gemm<4, 2, 2, 2, EPI, ACT, false, 2, 0, SKV>(p, stream);
      case 9: return launch_gemm<2, 4, 2, 2, EPI, ACT, false, 4, 0, SKV>(p, stream);
      case 10: return launch_gemm<4, 4, 2, 2, EPI, ACT, false, 2, 1, SKV>(p, stream);
      case 11: return launch_gemm<4, 3, 2, 2, EPI, ACT, false, 2, 1, SKV>(p, stream);
      case 17: p.ksplit = 1; return launch_gemm<4, 4, 2, 2, EPI, ACT, false, 2, 0, 0, 2>(p, stream);
    }
  }
  if constexpr (ALL) {
    switch (variant) {
      case 12: return launch_gemm<8, 4, 2, 2, EPI, ACT, false, 2, 1, SKV>(p, stream);
      case 13: return launch_gemm<8, 4, 1, 2, EPI, ACT, false, 2, 1, SKV>(p, stream);     // two waves, wave tile 128 features x 64 tokens
      case 14: return launch_gemm<8, 2, 2, 2, EPI, ACT, false, 2, 1, SKV>(p, stream);
      case 15: return launch_gemm<4, 4, 2, 4, EPI, ACT, false, 2, 1, SKV>(p, stream);
      case 18: p.ksplit = 1; return launch_gemm<4, 2, 2, 2, EPI, ACT, false, 2, 0, 0, 2>(p, stream);
      case 19: p.ksplit = 1; return launch_gemm<4, 2, 2, 2, EPI, ACT, false, 2, 0, 0, 3>(p, stream);
      case 20: p.ksplit = 1; return launch_gemm<2, 2, 2, 2, EPI, ACT, false, 2, 0, 0, 4>(p, stream);
    }
  }
  return AKI_ERR_UNSUPPORTED;
}

// the variants plan_small_m hands out (the product library instantiates these and no others)
template <int EPI, int ACT>
static int launch_small_m(GemmParams& p, int variant, int ksplit, hipStream_t stream) {
  p.ksplit = ksplit;
  if constexpr (EPI == EPI_QKV_ROPE8) {
    if (variant == 3) return launch_gemm<4, 3, 2, 2, EPI, ACT, false, 2, 0, 0>(p, stream);
    if (variant == 4) return launch_gemm<4, 3, 2, 2, EPI, ACT, false, 3, 0, 0>(p, stream);
  } else if constexpr (EPI == EPI_SWIGLU) {
    if (variant == 1) return launch_gemm<4, 4, 2, 2, EPI, ACT, false, 3, 0, 0>(p, stream);
  } else if constexpr (ACT == 0) {
    if (variant == 3) return launch_gemm<4, 3, 2, 2, EPI, ACT, false, 2, 0, 1>(p, stream);
    if (variant == 5) return launch_gemm<4, 2, 2, 2, EPI, ACT, false, 3, 0, 1>(p, stream);
  }
  return AKI_ERR_UNSUPPORTED;
}

template <int EPI, int ACT, bool FP8 = false>
static int run_planned(GemmParams& p, int plan, hipStream_t stream) {
#ifdef AKI_LAB_HOOKS
  if constexpr (!FP8) {
    if (g_sm_variant >= 0) {                       // lab: forced small-M variant
      int ks = g_sm_ksplit;
      if (ks > 1 && (p.sk_part == nullptr || ks > p.K / 64)) ks = 1;
      return launch_variant<EPI, ACT>(p, g_sm_variant, ks, stream);
    }
  }
#endif
  if constexpr (!FP8 && (EPI != EPI_PLAIN || ACT == 0)) {
    if (g_sm_variant != -2 && (p.w2 == nullptr)) {        // (lab: -2 = the plans as they were before the small-M planner)
      int v, ks;
      const int n_out = (EPI == EPI_SWIGLU) ? p.N / 2 : p.N;
      plan_small_m(EPI, p.M, n_out, p.K, p.sk_part != nullptr, v, ks);
      if (v >= 0 && ks > 1 && p.sk_bytes < linear_splitk_ws_bytes(v, ks, p.M, n_out) - linear_splitk_cnt_bytes()) v = -1;
      if (v >= 0 && !(EPI == EPI_PLAIN && p.row_shift != nullptr)) return launch_small_m<EPI, ACT>(p, v, ks, stream);
    }
  }
  if (plan == 1) return launch_small<EPI, ACT, FP8>(p, stream);
  if (plan == 0) return launch_big<EPI, ACT, FP8>(p, stream);
  if constexpr (EPI == EPI_PLAIN && !FP8) {
    if (plan == 3) return g_pipe == 3 ? launch_gemm<4, 3, 2, 2, EPI, ACT, FP8>(p, stream) : launch_gemm<4, 3, 2, 2, EPI, ACT, FP8, 2, 8>(p, stream);   // 128 features x 96 tokens (token tiles three deep)
    if (plan == 4) return launch_gemm<2, 2, 2, 2, EPI, ACT, FP8, 4>(p, stream);   // 64 features x 64 tokens, four-stage ring
  }
  const int m_main = p.M / 256 * 256;
  GemmParams a = p, b = p;
  a.M = m_main;
  int rc = launch_big<EPI, ACT, FP8>(a, stream);
  if (rc) return rc;
  b.M = p.M - m_main;
  b.m_offset = p.m_offset + m_main;
  b.x = (const bf16_t*)((const char*)p.x + (size_t)m_main * p.ldx * (FP8 ? 1 : 2));
  if (FP8) b.sx = p.sx + m_main;
  b.y = p.y + (size_t)m_main * p.ldy;
  if (p.preact) b.preact = p.preact + (size_t)m_main * p.ld_preact;
  if (p.residual && p.res_row_mod <= 0) b.residual = p.residual + (size_t)m_main * p.ldr;
  if (p.row_scale) b.row_scale = p.row_scale + m_main;
  if (p.row_shift) b.row_shift = p.row_shift + m_main;
  if (p.st_rstd) b.st_rstd = p.st_rstd + m_main;
  if (p.st_mean) b.st_mean = p.st_mean + m_main;
  return launch_small<EPI, ACT, FP8>(b, stream);
}

// Arrival counters: one per row panel (>= 64 rows), at the front of the statistics workspace; the partial sums follow.  The
// counter area has ONE size whatever M is (1 MiB = 262 144 panels: M <= 16 777 216 rows; more is refused), so the layout of a
// workspace never depends on the launches it served before: zero-filled once really means once.  (Until ABI 12 the area grew
// with M and a shrink-then-grow sequence on one buffer - M = 70 000, 3 000, 70 000 - left partial sums of the middle launch
// where the third one's counters were: ADVICE r3.)
constexpr size_t kStatsCntBytes = size_t(1) << 20;
constexpr int kStatsMaxRows = int(kStatsCntBytes / sizeof(unsigned)) * 64;
size_t linear_stats_cnt_bytes(int M) {
  (void)M;
  return kStatsCntBytes;
}
size_t linear_stats_ws_bytes(int M, int n_out) {
  return linear_stats_cnt_bytes(M) + (size_t)(2 * ((n_out + 63) / 64) + 2) * 2 * (size_t)M * sizeof(float);   // <= 2 waves x N/64 tile columns
}

int linear_bf16(const aki_linear_args* a, hipStream_t stream) {
  if (a->K % 64 != 0) return AKI_ERR_UNSUPPORTED;
  const int n_out = a->act == AKI_ACT_SWIGLU ? a->N / 2 : a->N;
  if (n_out % 4 != 0 || (a->act == AKI_ACT_SWIGLU && (a->N % 8 != 0))) return AKI_ERR_UNSUPPORTED;
  if ((a->ldx % 8) || (a->ldw % 8) || (a->ldy % 4) || (a->residual && (a->ldr % 4))) return AKI_ERR_ALIGNMENT;
  AKI_CHECK_ALIGN16(a->x);
  AKI_CHECK_ALIGN16(a->w);
  if (((uintptr_t)a->y & 7) || ((uintptr_t)a->residual & 7) || ((uintptr_t)a->bias & 7)) return AKI_ERR_ALIGNMENT;
  GemmParams p = {};
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w; p.bias = (const bf16_t*)a->bias;
  p.residual = (const bf16_t*)a->residual; p.y = (bf16_t*)a->y;
  p.M = a->M; p.N = a->N; p.K = a->K; p.ldx = a->ldx; p.ldw = a->ldw; p.ldy = a->ldy; p.ldr = a->ldr;
  p.res_row_mod = a->res_row_mod; p.act = a->act;
  p.wide = (n_out % 8 == 0) && (a->ldy % 8 == 0) && (((uintptr_t)a->y & 15) == 0);
  p.res_wide = a->residual && (n_out % 8 == 0) && (a->ldr % 8 == 0) && (((uintptr_t)a->residual & 15) == 0);
  if (a->w2) {
    AKI_CHECK_ALIGN16(a->w2);
    p.w2 = (const bf16_t*)a->w2; p.w2_row0 = a->w2_row0; p.w2_rows = a->w2_rows;
  }
  p.row_scale = a->row_scale; p.row_shift = a->row_shift; p.col_c = a->col_shift;
  if (a->row_shift && (!a->col_shift || !a->row_scale || (((uintptr_t)a->col_shift) & 15) || a->act == AKI_ACT_SWIGLU)) return AKI_ERR_INVALID_ARG;
  if (a->stats_rstd) {
    if (a->act == AKI_ACT_SWIGLU || a->M > kStatsMaxRows) return AKI_ERR_UNSUPPORTED;
    if (!a->stats_workspace || a->stats_workspace_bytes < linear_stats_ws_bytes(a->M, n_out) || (((uintptr_t)a->stats_workspace) & 15)) return AKI_ERR_WORKSPACE;
    p.st_rstd = a->stats_rstd; p.st_mean = a->stats_mean; p.st_eps = a->stats_eps;
    p.st_cnt = (unsigned*)a->stats_workspace;
    p.st_part = (float*)((char*)a->stats_workspace + linear_stats_cnt_bytes(a->M));
  }
  if (a->preact_out) {
    if (a->act != AKI_ACT_SWIGLU || a->row_scale) return AKI_ERR_INVALID_ARG;
    if ((a->ld_preact % 4) || a->ld_preact < a->N || ((uintptr_t)a->preact_out & 7)) return AKI_ERR_ALIGNMENT;
    p.preact = (bf16_t*)a->preact_out; p.ld_preact = (int)a->ld_preact;
  }
  if (a->act == AKI_ACT_SWIGLU) {
    if (a->bias) return AKI_ERR_UNSUPPORTED;
    return run_planned<EPI_SWIGLU, 0>(p, plan_tiles(a->M, n_out, 128, 64), stream);
  }
  if (a->splitk_workspace && a->splitk_workspace_bytes > linear_splitk_cnt_bytes() && !(((uintptr_t)a->splitk_workspace) & 255)) {
    p.sk_cnt = (unsigned*)a->splitk_workspace;
    p.sk_part = (float*)((char*)a->splitk_workspace + linear_splitk_cnt_bytes());
    p.sk_bytes = a->splitk_workspace_bytes - linear_splitk_cnt_bytes();
#ifdef AKI_LAB_HOOKS
    if (g_sm_variant >= 0 && a->splitk_workspace_bytes < linear_splitk_ws_bytes(g_sm_variant, g_sm_ksplit, a->M, n_out)) p.sk_part = nullptr;
#endif
  }
  const int plan = plan_tiles(a->M, n_out, 256, 128, 0.25, true, a->K / 64);
  switch (a->act) {
    case AKI_ACT_GELU_ERF: return run_planned<EPI_PLAIN, AKI_ACT_GELU_ERF>(p, plan, stream);
    case AKI_ACT_GELU_TANH: return run_planned<EPI_PLAIN, AKI_ACT_GELU_TANH>(p, plan, stream);
    default: return run_planned<EPI_PLAIN, 0>(p, plan, stream);
  }
}

int qkv_rope_bf16(const aki_mma_attn_args* a, void* q, void* k, void* v, hipStream_t stream) {
  if (a->Dh != 96) return AKI_ERR_UNSUPPORTED;
  if (a->d_model % 64 != 0) return AKI_ERR_UNSUPPORTED;
  if ((a->ldx % 8) || (a->ldw % 8)) return AKI_ERR_ALIGNMENT;
  AKI_CHECK_ALIGN16(a->x); AKI_CHECK_ALIGN16(a->w_qkv); AKI_CHECK_ALIGN16(a->cos); AKI_CHECK_ALIGN16(a->sin);
  AKI_CHECK_ALIGN16(q); AKI_CHECK_ALIGN16(k); AKI_CHECK_ALIGN16(v);
  GemmParams p = {};
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w_qkv;
  p.M = a->B * a->L; p.N = 3 * a->H * a->Dh; p.K = a->d_model; p.ldx = a->ldx; p.ldw = a->ldw;
  p.q_out = (bf16_t*)q; p.k_out = (bf16_t*)k; p.v_out = (bf16_t*)v;
  p.cos = a->cos; p.sin = a->sin; p.position_ids = a->position_ids; p.H = a->H; p.L = a->L;
  p.kvcap = a->kv_capacity > 0 ? a->kv_capacity : a->L;
  p.row_scale = a->row_scale;
  return run_planned<EPI_QKV_ROPE8, 0>(p, plan_tiles(p.M, p.N, 256, 128), stream);
}

// ---- fp8 (e4m3) operands, bf16 output: BASELINE configs[4] ------------------------------------------------------
int linear_fp8(const aki_linear_args* a, hipStream_t stream) {
  if (a->K % 128 != 0 || !a->x_scale || !a->w_scale) return AKI_ERR_UNSUPPORTED;
  const int n_out = a->act == AKI_ACT_SWIGLU ? a->N / 2 : a->N;
  if (n_out % 4 != 0 || (a->act == AKI_ACT_SWIGLU && (a->N % 8 != 0))) return AKI_ERR_UNSUPPORTED;
  if ((a->ldx % 16) || (a->ldw % 16) || (a->ldy % 4) || (a->residual && (a->ldr % 4))) return AKI_ERR_ALIGNMENT;
  AKI_CHECK_ALIGN16(a->x);
  AKI_CHECK_ALIGN16(a->w);
  AKI_CHECK_ALIGN16(a->w_scale);
  if (((uintptr_t)a->y & 7) || ((uintptr_t)a->residual & 7) || ((uintptr_t)a->bias & 7)) return AKI_ERR_ALIGNMENT;
  GemmParams p = {};
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w; p.bias = (const bf16_t*)a->bias;
  p.residual = (const bf16_t*)a->residual; p.y = (bf16_t*)a->y;
  p.M = a->M; p.N = a->N; p.K = a->K; p.ldx = a->ldx; p.ldw = a->ldw; p.ldy = a->ldy; p.ldr = a->ldr;
  p.res_row_mod = a->res_row_mod; p.act = a->act; p.sx = a->x_scale; p.sw = a->w_scale;
  p.wide = (n_out % 8 == 0) && (a->ldy % 8 == 0) && (((uintptr_t)a->y & 15) == 0);
  p.res_wide = a->residual && (n_out % 8 == 0) && (a->ldr % 8 == 0) && (((uintptr_t)a->residual & 15) == 0);
  if (a->act == AKI_ACT_SWIGLU) {
    if (a->bias) return AKI_ERR_UNSUPPORTED;
    return run_planned<EPI_SWIGLU, 0, true>(p, plan_tiles(a->M, n_out, 128, 64), stream);
  }
  if (a->act != AKI_ACT_NONE) return AKI_ERR_UNSUPPORTED;      // the fp8 path serves the language model's projections
  return run_planned<EPI_PLAIN, 0, true>(p, plan_tiles(a->M, n_out, 256, 128), stream);
}

int qkv_rope_fp8(const aki_mma_attn_args* a, void* q, void* k, void* v, hipStream_t stream) {
  if (a->Dh != 96 || a->d_model % 128 != 0 || !a->x_scale || !a->w_scale) return AKI_ERR_UNSUPPORTED;
  if ((a->ldx % 16) || (a->ldw % 16)) return AKI_ERR_ALIGNMENT;
  AKI_CHECK_ALIGN16(a->x); AKI_CHECK_ALIGN16(a->w_qkv); AKI_CHECK_ALIGN16(a->cos); AKI_CHECK_ALIGN16(a->sin);
  AKI_CHECK_ALIGN16(q); AKI_CHECK_ALIGN16(k); AKI_CHECK_ALIGN16(v); AKI_CHECK_ALIGN16(a->w_scale);
  GemmParams p = {};
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w_qkv;
  p.M = a->B * a->L; p.N = 3 * a->H * a->Dh; p.K = a->d_model; p.ldx = a->ldx; p.ldw = a->ldw;
  p.q_out = (bf16_t*)q; p.k_out = (bf16_t*)k; p.v_out = (bf16_t*)v;
  p.cos = a->cos; p.sin = a->sin; p.position_ids = a->position_ids; p.H = a->H; p.L = a->L;
  p.kvcap = a->kv_capacity > 0 ? a->kv_capacity : a->L;
  p.sx = a->x_scale; p.sw = a->w_scale;
  return run_planned<EPI_QKV_ROPE8, 0, true>(p, plan_tiles(p.M, p.N, 256, 128), stream);
}

}  // namespace aki
