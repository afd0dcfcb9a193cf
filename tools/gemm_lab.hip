// gemm_lab.hip - standalone experiment harness for the bf16 MFMA GEMM main loop (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -I include -I aki_amd/csrc tools/gemm_lab.hip -o gpurun_out/gemm_lab && ./gemm_lab
// Variants of the K-loop schedule and ablations of its components, interleaved rounds in one process on random
// data (cdna guide rules 24/25).  Output: TF/s per variant per shape + max error vs a host reference on sampled rows.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "aki_device.h"
#include "aki_mi355x.h"

using namespace aki;

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } \
  } while (0)

struct P {
  const bf16_t* x; const bf16_t* w; bf16_t* y;
  int M, N, K, tiles_m, tiles_n;
};

// VAR: 0 baseline (as shipped), 1 fragment double-buffer across k-steps, 2 = 1 + setprio around MFMA clusters,
//      3 = 2 + stagger waves 4-7 by half a K-step (extra barrier), 4 = baseline + setprio
// ABL: 0 none, 1 no global->LDS staging in the loop (stale LDS), 2 no ds_reads in the loop (fragments loaded once),
//      3 no epilogue stores
template <int TN, int VAR, int ABL>
__global__ __launch_bounds__(512, 2) void lab_kernel(const P p) {
  constexpr int TM = 2, BK = 64;
  constexpr int WROWS = TN * 32, BN = 2 * WROWS, BM = 256, ROWS = BN + BM, STAGE_BYTES = ROWS * 128, NLD = ROWS / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1, l31 = lane & 31, h = lane >> 5;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n, group = t / per_group, first_m = group * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (t % per_group) % gsz, tn = (t % per_group) / gsz;
  int m0 = tm * BM, n0 = tn * BN;
  if (VAR >= 6 && (p.M % BM) != 0) {
    const int full_m = p.tiles_m - 1;                       // M tiles that are completely inside M
    const int n_full = full_m * p.tiles_n;
    if (t < n_full) {
      const int g2 = t / per_group, fm = g2 * GM, gs = min(full_m - fm, GM);
      m0 = (fm + (t % per_group) % gs) * BM; n0 = ((t % per_group) / gs) * BN;
    } else {
      m0 = full_m * BM; n0 = (t - n_full) * BN;
    }
  }
  const char* src[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int rowgroup = j * 8 + wave, row = rowgroup * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (rowgroup * 8 < BN) src[j] = (const char*)(p.w + (size_t)min(n0 + row, p.N - 1) * p.K + chunk * 8);
    else src[j] = (const char*)(p.x + (size_t)min(m0 + row - BN, p.M - 1) * p.K + chunk * 8);
  }
  auto stage_piece = [&](int s, int kt, int j) {
    __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[j] + (size_t)kt * (BK * 2)), AKI_LDS_PTR(smem + s * STAGE_BYTES + (j * 8 + wave) * 1024), 16, 0, 0);
  };
  auto stage = [&](int s, int kt) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) stage_piece(s, kt, j);
  };
  f32x16 acc[TN][TM];
#pragma unroll
  for (int n = 0; n < TN; ++n)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
  const int swz = (lane >> 1) & 7;
  const int wbase = (wn * WROWS + l31) * 128, xbase = BN * 128 + (wm * TM * 32 + l31) * 128;
  const int nk = p.K / BK;
  const bool wave_has_rows = (VAR < 6) || (m0 + wm * TM * 32 < p.M);   // VAR>=6: waves without valid tokens skip the MFMAs
  auto ldf = [&](const char* sb, int ks, bf16x8 (&a)[TN], bf16x8 (&b)[TM]) {
    const int coff = ((2 * ks + h) ^ swz) << 4;
#pragma unroll
    for (int n = 0; n < TN; ++n) a[n] = *(const bf16x8*)(sb + wbase + n * 4096 + coff);
#pragma unroll
    for (int m = 0; m < TM; ++m) b[m] = *(const bf16x8*)(sb + xbase + m * 4096 + coff);
  };
  auto mma = [&](bf16x8 (&a)[TN], bf16x8 (&b)[TM]) {
    if (VAR == 2 || VAR == 3 || VAR == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int m = 0; m < TM; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n], b[m], acc[n][m], 0, 0, 0);
    if (VAR == 2 || VAR == 3 || VAR == 4) __builtin_amdgcn_s_setprio(0);
  };
  stage(0, 0);
  bf16x8 a0[TN], b0[TM], a1[TN], b1[TM];
  if (ABL == 2 || ABL == 5) {
    __syncthreads();
    ldf(smem, 0, a0, b0);
    ldf(smem, 1, a1, b1);
  }
  if (VAR == 3 && wave >= 4) __builtin_amdgcn_s_sleep(8);
  for (int kt = 0; kt < nk; ++kt) {
    if (ABL == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    else __syncthreads();
    if (VAR < 5) { if (ABL != 1 || kt == 0) { if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1); } }
    const char* sb = smem + (kt & 1) * STAGE_BYTES;
    if (VAR >= 5) {
      const bool more = kt + 1 < nk;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (wave_has_rows) {
          ldf(sb, ks, a0, b0);
          mma(a0, b0);
        }
        if (more) {   // LDS-DMA pieces of tile kt+1 issued behind this k-step's MFMAs (NLD = 8: 2 per k-step)
#pragma unroll
          for (int j = 2 * ks; j < 2 * ks + 2; ++j) stage_piece((kt + 1) & 1, kt + 1, j);
        }
      }
    } else if (ABL == 2 || ABL == 5) {
      mma(a0, b0); mma(a1, b1); mma(a0, b0); mma(a1, b1);
    } else if (VAR == 0 || VAR == 4) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        ldf(sb, ks, a0, b0);
        mma(a0, b0);
      }
    } else {
      ldf(sb, 0, a0, b0);
      ldf(sb, 1, a1, b1);
      mma(a0, b0);
      ldf(sb, 2, a0, b0);
      mma(a1, b1);
      ldf(sb, 3, a1, b1);
      mma(a0, b0);
      mma(a1, b1);
    }
  }
  if (ABL == 3) {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int m = 0; m < TM; ++m) s += acc[n][m][0] + acc[n][m][15];
    if (s == 123.456f) p.y[0] = 1;
    return;
  }
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    const int mrow = m0 + wm * TM * 32 + m * 32 + l31;
    if (mrow >= p.M) continue;
    bf16_t* yrow = p.y + (size_t)mrow * p.N;
    if (VAR >= 7) {
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {   // pair the register quads (2*q2, 2*q2+1): after the half swap each lane owns 16 contiguous bytes
          unsigned a0_ = pack_bf16x2(acc[n][m][8 * q2 + 0], acc[n][m][8 * q2 + 1]), a1_ = pack_bf16x2(acc[n][m][8 * q2 + 2], acc[n][m][8 * q2 + 3]);
          unsigned b0_ = pack_bf16x2(acc[n][m][8 * q2 + 4], acc[n][m][8 * q2 + 5]), b1_ = pack_bf16x2(acc[n][m][8 * q2 + 6], acc[n][m][8 * q2 + 7]);
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a0_), "+v"(b0_), "+v"(a1_), "+v"(b1_));
          // lanes 0-31: [own quad 2q2 | upper half's quad 2q2] = features 16*q2 + 0..7 ; lanes 32-63: features 16*q2 + 8..15
          const int f = n0 + wn * WROWS + n * 32 + q2 * 16 + 8 * h;
          if (f >= p.N) continue;
          u32x4 o = {a0_, a1_, b0_, b1_};
          *(u32x4*)(yrow + f) = o;
        }
    } else {
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int f = n0 + wn * WROWS + n * 32 + q4 * 8 + 4 * h;
        if (f >= p.N) continue;
        u32x2 o = {pack_bf16x2(acc[n][m][4 * q4], acc[n][m][4 * q4 + 1]), pack_bf16x2(acc[n][m][4 * q4 + 2], acc[n][m][4 * q4 + 3])};
        *(u32x2*)(yrow + f) = o;
      }
    }
  }
}


// ---- ring variant: BK = 32, NS-stage LDS ring, counted vmcnt (never 0 in the steady state), raw s_barrier ------------
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NS, int WIDE>
__global__ __launch_bounds__(512, 2) void ring_kernel(const P p) {
  constexpr int TN = 4, TM = 2, BK = 32;
  constexpr int WROWS = TN * 32, BN = 2 * WROWS, BM = 256, ROWS = BN + BM, STAGE_BYTES = ROWS * 64, NLD = ROWS / 128;  // 4 pieces / thread / stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1, l31 = lane & 31, h = lane >> 5;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n, group = t / per_group, first_m = group * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (t % per_group) % gsz, tn = (t % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  // one wave-instruction = 1 KiB = 16 rows x 64 B; lane -> (row = lane>>2, chunk pos = lane&3); swizzle pos ^ ((row>>2)&3)
  const char* src[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int rowgroup = j * 8 + wave, row = rowgroup * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    if (rowgroup * 16 < BN) src[j] = (const char*)(p.w + (size_t)min(n0 + row, p.N - 1) * p.K + chunk * 8);
    else src[j] = (const char*)(p.x + (size_t)min(m0 + row - BN, p.M - 1) * p.K + chunk * 8);
  }
  auto stage = [&](int s, int kt) {
#pragma unroll
    for (int j = 0; j < NLD; ++j)
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[j] + (size_t)kt * (BK * 2)), AKI_LDS_PTR(smem + s * STAGE_BYTES + (j * 8 + wave) * 1024), 16, 0, 0);
  };
  f32x16 acc[TN][TM];
#pragma unroll
  for (int n = 0; n < TN; ++n)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
  const int swz = (l31 >> 2) & 3;  // (row>>2)&3, block row bases are multiples of 32
  const int wbase = (wn * WROWS + l31) * 64, xbase = BN * 64 + (wm * TM * 32 + l31) * 64;
  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) stage(s, min(s, nk - 1));
  int cs = 0;  // stage of tile kt
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt's pieces are older than the (NS-2) newest tiles: wait for them, then make it a workgroup-wide fact
    wait_vmcnt<(NS - 2) * NLD>();
    __builtin_amdgcn_s_barrier();
    {  // refill the stage read in the previous iteration (clamped k: the tail re-loads valid memory, keeps the count constant)
      int ls = cs + NS - 1; if (ls >= NS) ls -= NS;
      stage(ls, min(kt + NS - 1, nk - 1));
    }
    const char* sb = smem + cs * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((2 * ks + h) ^ swz) << 4;
      bf16x8 a[TN], b[TM];
#pragma unroll
      for (int n = 0; n < TN; ++n) a[n] = *(const bf16x8*)(sb + wbase + n * 2048 + coff);
#pragma unroll
      for (int m = 0; m < TM; ++m) b[m] = *(const bf16x8*)(sb + xbase + m * 2048 + coff);
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int m = 0; m < TM; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n], b[m], acc[n][m], 0, 0, 0);
    }
    if (++cs == NS) cs = 0;
  }
  wait_vmcnt<0>();
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    const int mrow = m0 + wm * TM * 32 + m * 32 + l31;
    if (mrow >= p.M) continue;
    bf16_t* yrow = p.y + (size_t)mrow * p.N;
    if (WIDE) {
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          unsigned a0_ = pack_bf16x2(acc[n][m][8 * q2 + 0], acc[n][m][8 * q2 + 1]), a1_ = pack_bf16x2(acc[n][m][8 * q2 + 2], acc[n][m][8 * q2 + 3]);
          unsigned b0_ = pack_bf16x2(acc[n][m][8 * q2 + 4], acc[n][m][8 * q2 + 5]), b1_ = pack_bf16x2(acc[n][m][8 * q2 + 6], acc[n][m][8 * q2 + 7]);
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a0_), "+v"(b0_), "+v"(a1_), "+v"(b1_));
          const int f = n0 + wn * WROWS + n * 32 + q2 * 16 + 8 * h;
          if (f >= p.N) continue;
          u32x4 o = {a0_, a1_, b0_, b1_};
          *(u32x4*)(yrow + f) = o;
        }
    } else {
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int f = n0 + wn * WROWS + n * 32 + q4 * 8 + 4 * h;
          if (f >= p.N) continue;
          u32x2 o = {pack_bf16x2(acc[n][m][4 * q4], acc[n][m][4 * q4 + 1]), pack_bf16x2(acc[n][m][4 * q4 + 2], acc[n][m][4 * q4 + 3])};
          *(u32x2*)(yrow + f) = o;
        }
    }
  }
}

template <int NS, int WIDE>
static float run_ring(P p, int iters, hipStream_t s) {
  constexpr int SMEM = NS * 512 * 64;
  static bool set = false;
  if (!set) { CHECK(hipFuncSetAttribute((const void*)ring_kernel<NS, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM)); set = true; }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((ring_kernel<NS, WIDE>), dim3(p.tiles_m * p.tiles_n), dim3(512), SMEM, s, p);
  CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

// ---- 16x16x32 variant: same tile / LDS image / staging as the baseline, MFMA shape 16x16x32, 16-byte stores --------
typedef __attribute__((ext_vector_type(4))) float f32x4_;
template <int WIDE>
__global__ __launch_bounds__(512, 2) void k16_kernel(const P p) {
  constexpr int BK = 64, NF = 8, NT = 4;             // wave tile: 8 feature blocks x 4 token blocks of 16x16
  constexpr int WROWS = NF * 16, BN = 2 * WROWS, BM = 256, ROWS = BN + BM, STAGE_BYTES = ROWS * 128, NLD = ROWS / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1, l15 = lane & 15, kg = lane >> 4;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n, group = t / per_group, first_m = group * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (t % per_group) % gsz, tn = (t % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  const char* src[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int rowgroup = j * 8 + wave, row = rowgroup * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (rowgroup * 8 < BN) src[j] = (const char*)(p.w + (size_t)min(n0 + row, p.N - 1) * p.K + chunk * 8);
    else src[j] = (const char*)(p.x + (size_t)min(m0 + row - BN, p.M - 1) * p.K + chunk * 8);
  }
  auto stage = [&](int s, int kt) {
#pragma unroll
    for (int j = 0; j < NLD; ++j)
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[j] + (size_t)kt * (BK * 2)), AKI_LDS_PTR(smem + s * STAGE_BYTES + (j * 8 + wave) * 1024), 16, 0, 0);
  };
  f32x4_ acc[NF][NT];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[n][m][r] = 0.f;
  const int swz = (l15 >> 1) & 7;   // (row>>1)&7 for row = 16*blk + l15
  const int wbase = (wn * WROWS + l15) * 128, xbase = BN * 128 + (wm * NT * 16 + l15) * 128;
  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sb = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {       // two k32 steps per BK = 64
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
  // epilogue: lane (token = l15, group kg) owns features 4*kg .. 4*kg+3 of each 16-feature block
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    const int mrow = m0 + wm * NT * 16 + m * 16 + l15;
    const bool ok = mrow < p.M;
    bf16_t* yrow = p.y + (size_t)min(mrow, p.M - 1) * p.N;
    if (WIDE) {
#pragma unroll
      for (int n = 0; n < NF; n += 2) {   // pair feature blocks (n, n+1): v_permlane16_swap gives every lane 8 contiguous features
        unsigned p0 = pack_bf16x2(acc[n][m][0], acc[n][m][1]), p1 = pack_bf16x2(acc[n][m][2], acc[n][m][3]);
        unsigned q0 = pack_bf16x2(acc[n + 1][m][0], acc[n + 1][m][1]), q1 = pack_bf16x2(acc[n + 1][m][2], acc[n + 1][m][3]);
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(p0), "+v"(q0), "+v"(p1), "+v"(q1));
        // rows of 16 lanes r = kg: p = [P.r0, Q.r0, P.r2, Q.r2], q = [P.r1, Q.r1, P.r3, Q.r3]
        const int blk = n + (kg & 1), f = n0 + wn * WROWS + blk * 16 + 8 * (kg >> 1);
        if (ok && f < p.N) { u32x4 o = {p0, p1, q0, q1}; *(u32x4*)(yrow + f) = o; }
      }
    } else {
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int f = n0 + wn * WROWS + n * 16 + 4 * kg;
        if (ok && f < p.N) { u32x2 o = {pack_bf16x2(acc[n][m][0], acc[n][m][1]), pack_bf16x2(acc[n][m][2], acc[n][m][3])}; *(u32x2*)(yrow + f) = o; }
      }
    }
  }
}

// ---- register-staged variant: global -> VGPR two K-steps ahead, VGPR -> LDS one step ahead (two staging sets), so the
// global latency has two K-steps of MFMA work to hide behind with only two LDS buffers.  DEPTH 1 = one staging set
// (distance 1, isolates the cost of ds_write staging against global_load_lds), DEPTH 2 = two sets.
template <int DEPTH>
__global__ __launch_bounds__(512, 1) void k16r_kernel(const P p) {
  constexpr int BK = 64, NF = 8, NT = 4;
  constexpr int WROWS = NF * 16, BN = 2 * WROWS, BM = 256, ROWS = BN + BM, STAGE_BYTES = ROWS * 128, NLD = ROWS / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1, l15 = lane & 15, kg = lane >> 4;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n, group = t / per_group, first_m = group * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (t % per_group) % gsz, tn = (t % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  const char* src[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int rowgroup = j * 8 + wave, row = rowgroup * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (rowgroup * 8 < BN) src[j] = (const char*)(p.w + (size_t)min(n0 + row, p.N - 1) * p.K + chunk * 8);
    else src[j] = (const char*)(p.x + (size_t)min(m0 + row - BN, p.M - 1) * p.K + chunk * 8);
  }
  u32x4 pre[DEPTH][NLD];
  auto gload = [&](int set, int kt) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) pre[set][j] = *(const u32x4*)(src[j] + (size_t)kt * (BK * 2));
  };
  auto lstore = [&](int set, int s) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) *(u32x4*)(smem + s * STAGE_BYTES + (j * 8 + wave) * 1024 + lane * 16) = pre[set][j];
  };
  f32x4_ acc[NF][NT];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[n][m][r] = 0.f;
  const int swz = (l15 >> 1) & 7;
  const int wbase = (wn * WROWS + l15) * 128, xbase = BN * 128 + (wm * NT * 16 + l15) * 128;
  const int nk = p.K / BK;
  // prologue: tile 0 -> LDS stage 0; tiles 1 (and 2) in flight in the staging sets
  gload(0, 0);
  lstore(0, 0);
  if (nk > 1) gload(DEPTH == 2 ? 1 : 0, 1);
  if (DEPTH == 2 && nk > 2) gload(0, 2);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt+1 leaves its staging set for LDS stage (kt+1)&1 (free since the barrier at the end of iteration kt-1),
    // and the set is refilled with tile kt+1+DEPTH
    if (kt + 1 < nk) {
      const int set = DEPTH == 2 ? ((kt + 1) & 1) : 0;
      if (DEPTH == 2) {
        if (set == 1) { lstore(1, (kt + 1) & 1); if (kt + 3 < nk) gload(1, kt + 3); }
        else { lstore(0, (kt + 1) & 1); if (kt + 3 < nk) gload(0, kt + 3); }
      } else {
        lstore(0, (kt + 1) & 1);
        if (kt + 2 < nk) gload(0, kt + 2);
      }
    }
    const char* sb = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
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
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    const int mrow = m0 + wm * NT * 16 + m * 16 + l15;
    const bool ok = mrow < p.M;
    bf16_t* yrow = p.y + (size_t)min(mrow, p.M - 1) * p.N;
#pragma unroll
    for (int n = 0; n < NF; n += 2) {
      unsigned p0 = pack_bf16x2(acc[n][m][0], acc[n][m][1]), p1 = pack_bf16x2(acc[n][m][2], acc[n][m][3]);
      unsigned q0 = pack_bf16x2(acc[n + 1][m][0], acc[n + 1][m][1]), q1 = pack_bf16x2(acc[n + 1][m][2], acc[n + 1][m][3]);
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(p0), "+v"(q0), "+v"(p1), "+v"(q1));
      const int blk = n + (kg & 1), f = n0 + wn * WROWS + blk * 16 + 8 * (kg >> 1);
      if (ok && f < p.N) { u32x4 o = {p0, p1, q0, q1}; *(u32x4*)(yrow + f) = o; }
    }
  }
}

template <int DEPTH>
static float run_k16r(P p, int iters, hipStream_t s) {
  constexpr int SMEM = 2 * 512 * 128;
  static bool set = false;
  if (!set) { CHECK(hipFuncSetAttribute((const void*)k16r_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM)); set = true; }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k16r_kernel<DEPTH>), dim3(p.tiles_m * p.tiles_n), dim3(512), SMEM, s, p);
  CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

template <int WIDE>
static float run_k16(P p, int iters, hipStream_t s) {
  constexpr int SMEM = 2 * 512 * 128;
  static bool set = false;
  if (!set) { CHECK(hipFuncSetAttribute((const void*)k16_kernel<WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM)); set = true; }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k16_kernel<WIDE>), dim3(p.tiles_m * p.tiles_n), dim3(512), SMEM, s, p);
  CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

static inline float bf2f(bf16_t b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static inline bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

static float run_product(P p, int iters, hipStream_t s, int act) {
  aki_linear_args a = {};
  a.x = p.x; a.w = p.w; a.y = p.y; a.M = p.M; a.N = p.N; a.K = p.K; a.ldx = p.K; a.ldw = p.K; a.ldy = act == 3 ? p.N / 2 : p.N;
  a.act = act; a.dtype = AKI_DT_BF16;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) { int rc = aki_linear_fwd(&a, s); if (rc) { printf("aki_linear_fwd rc=%d\n", rc); exit(1); } }
  CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

template <int VAR, int ABL>
static float run(P p, int iters, hipStream_t s) {
  constexpr int SMEM = 2 * 512 * 128;
  static bool set = false;
  if (!set) { CHECK(hipFuncSetAttribute((const void*)lab_kernel<4, VAR, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM)); set = true; }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((lab_kernel<4, VAR, ABL>), dim3(p.tiles_m * p.tiles_n), dim3(512), SMEM, s, p);
  CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main() {
  struct Shape { int M, N, K; const char* name; } shapes[] = {{5240, 16384, 3072, "gate_up"}, {5240, 3072, 8192, "down"}, {5240, 9216, 3072, "qkv"}, {4096, 4096, 4096, "4k^3"}, {5120, 16384, 3072, "M5120"}};
  hipStream_t s; CHECK(hipStreamCreate(&s));
  for (auto sh : shapes) {
    size_t nx = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, ny = (size_t)sh.M * sh.N;
    std::vector<bf16_t> hx(nx), hw(nw);
    srand(1);
    for (auto& v : hx) v = f2bf((rand() / (float)RAND_MAX) * 2 - 1);
    for (auto& v : hw) v = f2bf(((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f);
    bf16_t *dx, *dw, *dy;
    CHECK(hipMalloc(&dx, nx * 2)); CHECK(hipMalloc(&dw, nw * 2)); CHECK(hipMalloc(&dy, ny * 2));
    CHECK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    P p = {dx, dw, dy, sh.M, sh.N, sh.K, 0, 0};
    const double fl = 2.0 * sh.M * sh.N * sh.K;
    float best[16] = {0};
    for (int i = 0; i < 16; ++i) best[i] = 1e9f;
    for (int round = 0; round < 4; ++round) {
      float t;
      t = run<0, 0>(p, 10, s); best[0] = fminf(best[0], t);
      t = run_k16<0>(p, 10, s); best[1] = fminf(best[1], t);
      t = run_k16<1>(p, 10, s); best[2] = fminf(best[2], t);
      t = run_product(p, 10, s, 0); best[3] = fminf(best[3], t);
      t = run_product(p, 10, s, 3); best[4] = fminf(best[4], t);
      t = run_k16r<1>(p, 10, s); best[5] = fminf(best[5], t);
      t = run_k16r<2>(p, 10, s); best[6] = fminf(best[6], t);
      t = run<0, 1>(p, 10, s); best[7] = fminf(best[7], t);
    }
    // correctness of the schedule variants on sampled rows
    double maxerr[7] = {0};
    for (int v = 0; v < 7; ++v) {
      CHECK(hipMemset(dy, 0, ny * 2));
      if (v == 0) run<0, 0>(p, 1, s); if (v == 1) run_k16<0>(p, 1, s); if (v == 2) run_k16<1>(p, 1, s); if (v == 3) run_product(p, 1, s, 0); if (v == 4) continue;
      if (v == 5) run_k16r<1>(p, 1, s); if (v == 6) run_k16r<2>(p, 1, s);
      CHECK(hipStreamSynchronize(s));
      int rows[3] = {0, sh.M / 2 + 1, sh.M - 1};
      for (int r : rows) {
        std::vector<bf16_t> hy(sh.N);
        CHECK(hipMemcpy(hy.data(), dy + (size_t)r * sh.N, sh.N * 2, hipMemcpyDeviceToHost));
        for (int n = 0; n < sh.N; n += 37) {
          double ref = 0;
          for (int k = 0; k < sh.K; ++k) ref += (double)bf2f(hx[(size_t)r * sh.K + k]) * bf2f(hw[(size_t)n * sh.K + k]);
          maxerr[v] = fmax(maxerr[v], fabs(ref - bf2f(hy[n])) / (1.0 + fabs(ref)));
        }
      }
    }
    const char* names[8] = {"baseline", "16x16x32", "16x16x32 wide stores", "PRODUCT aki_linear_fwd plain", "PRODUCT aki_linear_fwd swiglu", "reg-staged depth 1", "reg-staged depth 2", "ABL no-glds"};
    printf("== %s M=%d N=%d K=%d\n", sh.name, sh.M, sh.N, sh.K);
    for (int i = 0; i < 8; ++i) printf("   %-20s %8.4f ms  %7.1f TF/s%s\n", names[i], best[i], fl / best[i] / 1e9, (i < 7 && i != 4) ? (maxerr[i] < 2e-2 ? "  ok" : "  WRONG") : "");
    CHECK(hipFree(dx)); CHECK(hipFree(dw)); CHECK(hipFree(dy));
  }
  return 0;
}
