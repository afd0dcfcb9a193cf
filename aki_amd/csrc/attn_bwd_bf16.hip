// attn_bwd_bf16.hip - backward of the attention core (bf16, head_dim 96 or 64, CDNA4 MFMA), with or without the
// modality-mutual mask.  What the reference gets from autograd over HF:phi3/modeling_phi3.py:145-167 (eager attention
// under the dense (B,1,L,L) mask of src/vlm.py:410-443) and over src/helpers.py:76-102 (Perceiver attention), computed
// flash-style from the forward's log-sum-exp without ever forming an Lq x Lk tensor:
//     P  = exp(S*scale - lse)            (recomputed; masked pairs are 0)
//     dV = P^T dO
//     dP = dO V^T,   dS = P o (dP - delta),   delta[q] = sum_d dO[q,d] O[q,d]
//     dQ = scale * dS K,   dK = scale * dS^T Q
// Two kernels, no atomics, deterministic:
//   attn_bwd_dkv   workgroup = 128 keys of one (batch, head) (wave = 32 keys, K and V fragments live in registers);
//                  32-row Q / dO tiles stream through LDS.  S = Q K^T puts the KEY on the lane and 16 query rows in the
//                  accumulator registers, which is directly the A operand (P^T, dS^T) of the two accumulating products;
//                  their B operands (dO, Q with the query as contraction index) come from transposed LDS reads.
//   attn_bwd_dq    the mirror image: workgroup = 128 queries (Q, dO fragments, lse and delta lane-local), 32-key K / V
//                  tiles stream through LDS, S^T = K Q^T puts the QUERY on the lane; dQ^T += K^T dS^T.
// Rows at or beyond seq_len (batch-stacking padding) are skipped: every gradient that reaches them is exactly zero
// (their labels are -100 and no valid row attends to a padded column).
#include <type_traits>

#include "aki_device.h"

#ifndef AKI_BWD_RANK_MAJOR
#define AKI_BWD_RANK_MAJOR 1
#endif

namespace aki {

struct AttnBwdParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;   // [B*H][Lq|Lk][DH]
  const bf16_t* dout;                                   // [B][Lq][H][DH]
  const float* lse; const float* delta;                 // [B*H][Lq]
  bf16_t* dq; bf16_t* dk; bf16_t* dv;                   // [B*H][Lq|Lk][DH]
  const aki_mma_rect* rects; const uint64_t* vbits; const int* seq_lens;
  int max_rects, nwords;
  int B, H, Lq, Lk;
  float scale, scale_log2;
};

template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (N > 0) {
    sfor<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// delta[bh][q] = sum_d dO[b][q][h][d] * O[b][q][h][d]
template <int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* o, const bf16_t* dout, float* delta, int B, int H, int Lq) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;     // (b*Lq + q)*H + h
  if (i >= (size_t)B * Lq * H) return;
  const int h = (int)(i % H);
  const size_t bq = i / H;
  const int q = (int)(bq % Lq), b = (int)(bq / Lq);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < DH / 8; ++c) {
    const u32x4 a = *(const u32x4*)(o + i * DH + c * 8), d = *(const u32x4*)(dout + i * DH + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += bf16_lo(a[e]) * bf16_lo(d[e]) + bf16_hi(a[e]) * bf16_hi(d[e]);
  }
  delta[((size_t)b * H + h) * Lq + q] = s;
}

// transposed 16-row fragment of a [rows][DH] LDS tile: lane (c = lane&31, h = lane>>5) receives, for column dt*32 + c,
// rows r0 + 4h + {0,1,2,3} (first read) and r0 + 8 + 4h + {0,1,2,3} (second read) - the contraction-index order in which
// the 32x32 accumulator of the score product hands over its 8 values per 16-wide step.
typedef short v4i16_t __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) v4i16_t* lds_v4i16_ptr;

// LDS tiles are stored with the 16-byte chunk index XOR-ed by (row >> 2) & 3 (as the K tile of the forward kernel): the
// ds_read_b128 row reads of the score products become bank-conflict free with 192-byte rows, and the transposed reads keep
// their pattern because the four rows one 16-lane group touches share the same XOR value.  off_lo / off_hi are the lane's
// byte offsets for the first / second read (rows r0 + 4h + j and r0 + 8 + 4h + j have XOR values h and (h + 2) & 3).
template <int ROW>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int off_lo, int off_hi, int r0, int dt) {
  const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(tile + off_lo + r0 * ROW + dt * 64));
  const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(tile + off_hi + (r0 + 8) * ROW + dt * 64));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  const u32x4 v = {l2[0], l2[1], h2[0], h2[1]};
  return __builtin_bit_cast(bf16x8, v);
}

// lane byte offset of a transposed read inside a 16-row block whose rows have XOR value x
__device__ __forceinline__ int tr_lane_off(int lane, int row_bytes, int x) {
  const int h = lane >> 5, j = (lane & 15) >> 2;
  const int cl = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);       // low two bits of the 16-byte chunk index
  return (4 * h + j) * row_bytes + ((cl ^ x) << 4) + (lane & 1) * 8;
}

// ------------------------------------------------------------------------------------------------------------
template <int DH, bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnBwdParams p) {
  constexpr int KS = DH / 16, DT = DH / 32, ROW = DH * 2, TILE = 32 * ROW;
  constexpr int CPT = DH / 32;                       // 16-B chunks per thread per tile pair: 2*32*DH/8 / 256
  constexpr int STAGE = 2 * TILE + 256;              // Q tile, dO tile, lse[32], delta[32]
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + AKI_MAX_RECTS * 16];
  const aki_mma_rect* sR = (const aki_mma_rect*)(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
#if AKI_BWD_RANK_MAJOR
  // rank-major dispatch: all first (heaviest) key blocks of every (batch, head) pair, then all second ones, ... - the
  // hardware's in-order dispatch packs the long workgroups first and fills the tail with short ones
  const int nbh_ = p.B * p.H;
  const int bh = blockIdx.x % nbh_, kb0 = (blockIdx.x / nbh_) * 128;
#else
  const int nkb = (p.Lk + 127) / 128;
  const int bh = blockIdx.x / nkb, kb0 = (blockIdx.x - bh * nkb) * 128;
#endif
  const int b = bh / p.H, head = bh - b * p.H;
  const int Lb = (MASKED && p.seq_lens) ? min(p.seq_lens[b], p.Lq) : p.Lq;
  const int key = kb0 + wave * 32 + l31;

  if (MASKED) {
    if (tid < AKI_MAX_RECTS) {
      u32x4 r = {0u, 0u, 0u, 0u};
      if (tid < p.max_rects) r = ((const u32x4*)p.rects)[(size_t)b * p.max_rects + tid];
      ((u32x4*)sR)[tid] = r;
    }
    __syncthreads();
  }
  bool key_ok = key < p.Lk;
  if (MASKED && p.vbits && key_ok) key_ok = ((p.vbits[(size_t)b * p.nwords + (key >> 6)] >> (key & 63)) & 1ull) != 0ull;
  // rectangles as wave-uniform scalars (SGPRs): every test against them below is branch-free
  int r_rlo[AKI_MAX_RECTS], r_rhi[AKI_MAX_RECTS], r_clo[AKI_MAX_RECTS], r_chi[AKI_MAX_RECTS];
#pragma unroll
  for (int i = 0; i < AKI_MAX_RECTS; ++i) {
    const aki_mma_rect r = sR[i];
    const bool ok = MASKED && i < p.max_rects && r.row_hi > r.row_lo && r.col_hi > r.col_lo;
    r_rlo[i] = __builtin_amdgcn_readfirstlane(ok ? r.row_lo : 0);
    r_rhi[i] = __builtin_amdgcn_readfirstlane(ok ? r.row_hi : 0);
    r_clo[i] = __builtin_amdgcn_readfirstlane(ok ? r.col_lo : 0);
    r_chi[i] = __builtin_amdgcn_readfirstlane(ok ? r.col_hi : 0);
  }
  // which query tiles can see any of the workgroup's keys
  auto need = [&](int t) -> bool {
    if (t * 32 >= Lb) return false;
    if (!MASKED) return true;
    bool nd = t * 32 + 31 >= kb0;
#pragma unroll
    for (int i = 0; i < AKI_MAX_RECTS; ++i)
      nd = nd || (r_rlo[i] < t * 32 + 32 && r_rhi[i] > t * 32 && r_clo[i] < kb0 + 128 && r_chi[i] > kb0);
    return nd;
  };
  const int nqt = (Lb + 31) / 32;
  auto next_needed = [&](int t) { while (t < nqt && !need(t)) ++t; return t; };

  // K / V fragments of this wave's 32 keys (B operands of S = Q K^T and dP = dO V^T)
  bf16x8 kf[KS], vf[KS];
  {
    const size_t off = ((size_t)bh * p.Lk + min(key, p.Lk - 1)) * DH + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *(const bf16x8*)(p.k + off + 16 * ks);
      vf[ks] = *(const bf16x8*)(p.v + off + 16 * ks);
    }
  }
  f32x16 dv[DT], dk[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv[dt][r] = 0.f; dk[dt][r] = 0.f; }

  const bf16_t* qbase = p.q + (size_t)bh * p.Lq * DH;
  const bf16_t* dobase = p.dout + (size_t)b * p.Lq * p.H * DH + (size_t)head * DH;
  const size_t do_stride = (size_t)p.H * DH;
  const float* lsebase = p.lse + (size_t)bh * p.Lq;
  const float* deltabase = p.delta + (size_t)bh * p.Lq;

  u32x4 pre[CPT];
  float pre_s = 0.f;
  auto fetch = [&](int t) {                            // tile t -> registers
    const int q0 = t * 32;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int g = i * 256 + tid;                     // chunk index over [Q tile | dO tile]
      const int which = g / (32 * DH / 8), gg = g - which * (32 * DH / 8);
      const int r = gg / (DH / 8), c = gg - r * (DH / 8);
      const int row = min(q0 + r, p.Lq - 1);
      pre[i] = which ? *(const u32x4*)(dobase + (size_t)row * do_stride + c * 8) : *(const u32x4*)(qbase + (size_t)row * DH + c * 8);
    }
    if (tid < 64) {
      const int row = min(q0 + (tid & 31), p.Lq - 1);
      pre_s = tid < 32 ? lsebase[row] : deltabase[row];
    }
  };
  auto stash = [&](int stage) {                        // registers -> LDS stage
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int g = i * 256 + tid;
      const int which = g / (32 * DH / 8), gg = g - which * (32 * DH / 8);
      const int r = gg / (DH / 8), c = gg - r * (DH / 8);
      *(u32x4*)(base + which * TILE + r * ROW + ((c ^ ((r >> 2) & 3)) << 4)) = pre[i];
    }
    if (tid < 64) ((float*)(base + 2 * TILE))[tid] = pre_s;
  };

  const int tr_lo = tr_lane_off(lane, ROW, h), tr_hi = tr_lane_off(lane, ROW, (h + 2) & 3);
  const int rsw = (l31 >> 2) & 3;                    // XOR value of this lane's row in the row reads
  const float lse_k = 1.44269504088896340736f;

  int t = next_needed(0);
  if (t < nqt) { fetch(t); stash(0); }
  __syncthreads();
  int stage = 0;
  while (t < nqt) {
    const int tn = next_needed(t + 1);
    if (tn < nqt) fetch(tn);
    const char* Qs = smem + stage * STAGE;
    const char* Ds = Qs + TILE;
    const float* Ls = (const float*)(Qs + 2 * TILE);
    const int q0 = t * 32;
    const bool causal_full = q0 >= kb0 + wave * 32 + 31;     // every row of the tile is at or past every key of the wave
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 qa = *(const bf16x8*)(Qs + l31 * ROW + (((2 * ks + h) ^ rsw) << 4));
      const bf16x8 da = *(const bf16x8*)(Ds + l31 * ROW + (((2 * ks + h) ^ rsw) << 4));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
    }
    // accumulator register r <-> query row q0 + (r&3) + 8*(r>>2) + 4h; this lane's key is fixed.
    // visibility of the 16 (row, key) pairs as a bit mask: causal part, then one branch-free pass per rectangle that
    // touches this tile (wave-uniform test), finally rows beyond seq_len and invalid keys
    unsigned vis = 0;
    const int qb = q0 + 4 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qrow = qb + (r & 3) + 8 * (r >> 2);
      const bool c = MASKED ? (key <= qrow) : true;
      vis |= (c && qrow < Lb) ? (1u << r) : 0u;
    }
    if (MASKED && !causal_full) {
#pragma unroll
      for (int i = 0; i < AKI_MAX_RECTS; ++i) {
        if (r_rlo[i] < q0 + 32 && r_rhi[i] > q0) {                   // uniform
          const bool kin = key >= r_clo[i] && key < r_chi[i];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int qrow = qb + (r & 3) + 8 * (r >> 2);
            vis |= (kin && qrow >= r_rlo[i] && qrow < r_rhi[i] && qrow < Lb) ? (1u << r) : 0u;
          }
        }
      }
    }
    if (!key_ok) vis = 0;
    float ds[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 l4 = *(const f32x4*)(Ls + 8 * g4 + 4 * h), d4 = *(const f32x4*)(Ls + 32 + 8 * g4 + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        const float pv = ((vis >> r) & 1u) ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.scale_log2, -l4[e] * lse_k)) : 0.f;
        s[r] = pv;
        ds[r] = pv * (dp[r] - d4[e]);
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x8 pa, dsa;
#pragma unroll
      for (int e = 0; e < 8; ++e) { pa[e] = (__bf16)s[8 * m + e]; dsa[e] = (__bf16)ds[8 * m + e]; }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, tr_frag<ROW>(Ds, tr_lo, tr_hi, 16 * m, dt), dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, tr_frag<ROW>(Qs, tr_lo, tr_hi, 16 * m, dt), dk[dt], 0, 0, 0);
      }
    }
    if (tn < nqt) stash(stage ^ 1);
    __syncthreads();
    stage ^= 1;
    t = tn;
  }
  // D layout: lane (column d = 32dt + l31, h) holds keys (r&3) + 8(r>>2) + 4h of this wave's 32
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = kb0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (kk < p.Lk) {
        const size_t o = ((size_t)bh * p.Lk + kk) * DH + 32 * dt + l31;
        ((__bf16*)p.dk)[o] = (__bf16)(dk[dt][r] * p.scale);
        ((__bf16*)p.dv)[o] = (__bf16)dv[dt][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
template <int DH, bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnBwdParams p) {
  constexpr int KS = DH / 16, DT = DH / 32, ROW = DH * 2, TILE = 32 * ROW;
  constexpr int CPT = DH / 32;
  constexpr int STAGE = 2 * TILE;                    // K tile, V tile
  constexpr int MAXW = 256;                          // valid-column words kept in LDS (Lk <= 16384): a per-tile global
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + AKI_MAX_RECTS * 16 + MAXW * 8];   // load would drain the prefetch
  const aki_mma_rect* sR = (const aki_mma_rect*)(smem + 2 * STAGE);
  unsigned long long* sVB = (unsigned long long*)(smem + 2 * STAGE + AKI_MAX_RECTS * 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int nqb = (p.Lq + 127) / 128;
#if AKI_BWD_RANK_MAJOR
  const int nbh_ = p.B * p.H;
  const int bh = blockIdx.x % nbh_, q0 = (nqb - 1 - blockIdx.x / nbh_) * 128;              // heavy (late) query blocks first, across all pairs
#else
  const int bh = blockIdx.x / nqb, q0 = (nqb - 1 - (blockIdx.x - bh * nqb)) * 128;     // heavy (late) query blocks first
#endif
  const int b = bh / p.H, head = bh - b * p.H;
  const int Lb = (MASKED && p.seq_lens) ? min(p.seq_lens[b], p.Lq) : p.Lq;
  const int row = q0 + wave * 32 + l31;
  const bool row_ok = row < Lb;

  if (MASKED) {
    if (tid < AKI_MAX_RECTS) {
      u32x4 r = {0u, 0u, 0u, 0u};
      if (tid < p.max_rects) r = ((const u32x4*)p.rects)[(size_t)b * p.max_rects + tid];
      ((u32x4*)sR)[tid] = r;
    }
    for (int w = tid; w < p.nwords && w < MAXW; w += 256) sVB[w] = p.vbits ? p.vbits[(size_t)b * p.nwords + w] : ~0ull;
    __syncthreads();
  }
  int rc0 = 0, rc1 = 0;                               // unlocked column interval of this lane's row (one rectangle per row)
  int r_rlo[AKI_MAX_RECTS], r_rhi[AKI_MAX_RECTS], r_clo[AKI_MAX_RECTS], r_chi[AKI_MAX_RECTS];
#pragma unroll
  for (int i = 0; i < AKI_MAX_RECTS; ++i) {
    const aki_mma_rect r = sR[i];
    const bool ok = MASKED && i < p.max_rects && r.row_hi > r.row_lo && r.col_hi > r.col_lo;
    r_rlo[i] = __builtin_amdgcn_readfirstlane(ok ? r.row_lo : 0);
    r_rhi[i] = __builtin_amdgcn_readfirstlane(ok ? r.row_hi : 0);
    r_clo[i] = __builtin_amdgcn_readfirstlane(ok ? r.col_lo : 0);
    r_chi[i] = __builtin_amdgcn_readfirstlane(ok ? r.col_hi : 0);
    if (row >= r_rlo[i] && row < r_rhi[i]) { rc0 = r_clo[i]; rc1 = r_chi[i]; }
  }
  auto need = [&](int t) -> bool {
    if (t * 32 >= p.Lk || q0 >= Lb) return false;
    if (!MASKED) return true;
    bool nd = t * 32 <= q0 + 127;
#pragma unroll
    for (int i = 0; i < AKI_MAX_RECTS; ++i)
      nd = nd || (r_rlo[i] < q0 + 128 && r_rhi[i] > q0 && r_clo[i] < t * 32 + 32 && r_chi[i] > t * 32);
    return nd;
  };
  const int nkt = (p.Lk + 31) / 32;
  auto next_needed = [&](int t) { while (t < nkt && !need(t)) ++t; return t; };

  bf16x8 qf[KS], dof[KS];
  {
    const int rr = min(row, p.Lq - 1);
    const bf16_t* qrow = p.q + ((size_t)bh * p.Lq + rr) * DH + 8 * h;
    const bf16_t* drow = p.dout + ((size_t)b * p.Lq + rr) * p.H * DH + (size_t)head * DH + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
      dof[ks] = *(const bf16x8*)(drow + 16 * ks);
    }
  }
  const float lse_q = p.lse[(size_t)bh * p.Lq + min(row, p.Lq - 1)] * 1.44269504088896340736f;
  const float delta_q = p.delta[(size_t)bh * p.Lq + min(row, p.Lq - 1)];
  f32x16 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  const bf16_t* kbase = p.k + (size_t)bh * p.Lk * DH;
  const bf16_t* vbase = p.v + (size_t)bh * p.Lk * DH;
  u32x4 pre[CPT];
  auto fetch = [&](int t) {
    const int k0 = t * 32;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int g = i * 256 + tid;
      const int which = g / (32 * DH / 8), gg = g - which * (32 * DH / 8);
      const int r = gg / (DH / 8), c = gg - r * (DH / 8);
      const size_t off = (size_t)min(k0 + r, p.Lk - 1) * DH + c * 8;
      pre[i] = which ? *(const u32x4*)(vbase + off) : *(const u32x4*)(kbase + off);
    }
  };
  auto stash = [&](int stage) {
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int g = i * 256 + tid;
      const int which = g / (32 * DH / 8), gg = g - which * (32 * DH / 8);
      const int r = gg / (DH / 8), c = gg - r * (DH / 8);
      *(u32x4*)(base + which * TILE + r * ROW + ((c ^ ((r >> 2) & 3)) << 4)) = pre[i];
    }
  };
  const int tr_lo = tr_lane_off(lane, ROW, h), tr_hi = tr_lane_off(lane, ROW, (h + 2) & 3);
  const int rsw = (l31 >> 2) & 3;

  int t = next_needed(0);
  if (t < nkt) { fetch(t); stash(0); }
  __syncthreads();
  int stage = 0;
  while (t < nkt) {
    const int tn = next_needed(t + 1);
    if (tn < nkt) fetch(tn);
    const char* Ks = smem + stage * STAGE;
    const char* Vs = Ks + TILE;
    const int k0 = t * 32;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 ka = *(const bf16x8*)(Ks + l31 * ROW + (((2 * ks + h) ^ rsw) << 4));
      const bf16x8 va = *(const bf16x8*)(Vs + l31 * ROW + (((2 * ks + h) ^ rsw) << 4));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[ks], dp, 0, 0, 0);
    }
    // accumulator register r <-> key k0 + (r&3) + 8*(r>>2) + 4h; this lane's query row is fixed
    unsigned vword = 0xffffffffu;
    if (MASKED) vword = (unsigned)(sVB[k0 >> 6] >> (k0 & 63));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = (r & 3) + 8 * (r >> 2) + 4 * h, kk = k0 + kl;
      bool vis = row_ok && kk < p.Lk && ((vword >> kl) & 1u);
      if (MASKED) vis = vis && (kk <= row || (kk >= rc0 && kk < rc1));
      const float pv = vis ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.scale_log2, -lse_q)) : 0.f;
      s[r] = pv * (dp[r] - delta_q);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x8 dsb;
#pragma unroll
      for (int e = 0; e < 8; ++e) dsb[e] = (__bf16)s[8 * m + e];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<ROW>(Ks, tr_lo, tr_hi, 16 * m, dt), dsb, dq[dt], 0, 0, 0);
    }
    if (tn < nkt) stash(stage ^ 1);
    __syncthreads();
    stage ^= 1;
    t = tn;
  }
  // dQ^T layout: lane (query = l31, h) holds d = 32dt + (r&3) + 8(r>>2) + 4h
  if (row < p.Lq) {
    bf16_t* out = p.dq + ((size_t)bh * p.Lq + row) * DH + 4 * h;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const u32x2 pk = {pack_bf16x2(dq[dt][4 * q4] * p.scale, dq[dt][4 * q4 + 1] * p.scale),
                          pack_bf16x2(dq[dt][4 * q4 + 2] * p.scale, dq[dt][4 * q4 + 3] * p.scale)};
        *(u32x2*)(out + dt * 32 + q4 * 8) = pk;
      }
  }
}

size_t attn_bwd_ws_bytes(int B, int H, int Lq) { return (size_t)B * H * Lq * 4; }

template <int DH, bool MASKED>
static void launch_bwd(const AttnBwdParams& p, hipStream_t s) {
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, MASKED>), dim3(p.B * p.H * ((p.Lk + 127) / 128)), dim3(256), 0, s, p);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, MASKED>), dim3(p.B * p.H * ((p.Lq + 127) / 128)), dim3(256), 0, s, p);
}

// q,k,v [B*H][L][Dh]; o, dout [B][Lq][H][Dh]; lse [B*H][Lq] from the forward; outputs dq/dk/dv like q/k/v.
// masked: rects / col_valid_bits / seq_lens as in aki_mma_attn_core_fwd (requires Lq == Lk); else plain attention.
int attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq, void* dk,
                  void* dv, const aki_mma_rect* rects, int max_rects, const uint64_t* vbits, const int* seq_lens, int masked, int B,
                  int H, int Lq, int Lk, int Dh, float scale, void* ws, size_t ws_bytes, hipStream_t s) {
  if (Dh != 96 && Dh != 64) return AKI_ERR_UNSUPPORTED;
  if (masked && Lq != Lk) return AKI_ERR_INVALID_ARG;
  if (masked && (Lk + 63) / 64 > 256) return AKI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < attn_bwd_ws_bytes(B, H, Lq)) return AKI_ERR_WORKSPACE;
  AttnBwdParams p = {(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout, lse, (const float*)ws,
                     (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, rects, vbits, seq_lens, rects ? max_rects : 0, (Lk + 63) / 64,
                     B, H, Lq, Lk, scale, scale * 1.44269504088896340736f};
  AKI_CLEAR_ERR();
  const int nrow = (int)(((size_t)B * Lq * H + 255) / 256);
  if (Dh == 96) hipLaunchKernelGGL(attn_delta_kernel<96>, dim3(nrow), dim3(256), 0, s, (const bf16_t*)o, (const bf16_t*)dout, (float*)ws, B, H, Lq);
  else hipLaunchKernelGGL(attn_delta_kernel<64>, dim3(nrow), dim3(256), 0, s, (const bf16_t*)o, (const bf16_t*)dout, (float*)ws, B, H, Lq);
  if (Dh == 96) { if (masked) launch_bwd<96, true>(p, s); else launch_bwd<96, false>(p, s); }
  else { if (masked) launch_bwd<64, true>(p, s); else launch_bwd<64, false>(p, s); }
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
