// attn_nc_bf16.hip - plain (non-causal) softmax(Q K^T * scale) V for the vision side of AKI, bf16, CDNA4 MFMA.
//
// Replaces: SigLIP `eager_attention_forward` (HF:siglip/modeling_siglip.py:226-247; 16 heads x 72) and the softmax
// attention inside `PerceiverAttention.forward` (src/helpers.py:93-100; 8 heads x 64, 144 queries over 873 keys).
// Same machinery as mma_attn_bf16.hip (swapped S^T = K Q^T so the softmax row is lane-local, P from accumulators as
// the PV B-operand, V^T via ds_read_b64_tr_b16) with the mask logic removed: every key < Lk is visible.
// Q/K/V are read IN PLACE from the projection outputs through element strides (token stride / head stride), so the
// fused QKV / KV GEMM outputs need no transposition or copy.
//
// head_dim DH in {32, 64, 72, 96}: K rows in LDS are padded to a pitch whose dword count is 4 x odd (conflict-free
// ds_read_b128: 144 / 176 / 208 B); V rows use a 192-B pitch (conflict-free transposed reads).  For DH = 72 the
// fifth k-step covers d = 64..79: the K pad chunk is zero-filled once, so whatever finite bits the matching Q
// fragment holds contribute nothing; V columns >= DH are zero-filled and their outputs are not stored.
#include "aki_device.h"

namespace aki {

struct AttnNcParams {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;
  long q_sb, q_sh, q_st;   // element strides: batch, head, token
  long k_sb, k_sh, k_st;
  long v_sb, v_sh, v_st;
  int B, H, Lq, Lk;
  int nqt;
  float scale_log2;
  float* lse;   // [B*H][Lq] natural-log log-sum-exp of the scaled scores, or NULL (needed by the backward)
};

__device__ __forceinline__ float max3_nc(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int NW, int DH>
__global__ __launch_bounds__(NW * 64, 2) void attn_nc_bf16_kernel(const AttnNcParams p) {
  constexpr int BQ = NW * 32, NT = NW * 64;
  constexpr int KS = (DH + 15) / 16;            // QK^T k-steps of 16
  constexpr int DT = (DH + 31) / 32;            // PV output tiles of 32 channels
  constexpr int CPR = DH / 8;                   // 16-byte chunks per row
  constexpr int KROW = DH == 96 ? 208 : (DH == 72 ? 176 : (DH == 64 ? 144 : 80));  // dword pitch = 4 x odd
  constexpr int VROW = 192;
  constexpr int KTILE = 64 * KROW, VTILE = 64 * VROW;
  constexpr int NCHUNK = 64 * CPR;
  constexpr int NCH = (NCHUNK + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) char smem[2 * KTILE + 2 * VTILE];
  char* const sK = smem;
  char* const sV = smem + 2 * KTILE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = t / p.nqt, qt = t - bh * p.nqt;
  const int b = bh / p.H, head = bh - b * p.H;
  const int Lq = p.Lq, Lk = p.Lk;
  const int row = qt * BQ + wave * 32 + l31;

  // zero the pad chunks of both K buffers (chunk CPR of every row) and the V columns >= DH, once
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (KS * 16 > DH)
      for (int r = tid; r < 2 * 64; r += NT) *(u32x4*)(sK + (r >> 6) * KTILE + (r & 63) * KROW + CPR * 16) = z;
    if (DT * 32 > DH) {
      constexpr int PADC = 12 - CPR;  // pad chunks per V row
      for (int e = tid; e < 2 * 64 * PADC; e += NT) {
        const int buf = e / (64 * PADC), rr = (e / PADC) & 63, pc = e % PADC;
        *(u32x4*)(sV + buf * VTILE + rr * VROW + (CPR + pc) * 16) = z;
      }
    }
  }

  const bf16_t* qrow = p.q + (size_t)b * p.q_sb + (size_t)head * p.q_sh + (size_t)min(row, Lq - 1) * p.q_st;
  const char* kb = (const char*)(p.k + (size_t)b * p.k_sb + (size_t)head * p.k_sh);
  const char* vb = (const char*)(p.v + (size_t)b * p.v_sb + (size_t)head * p.v_sh);

  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d = min(16 * ks + 8 * h, DH - 8);  // DH = 72, ks = 4, h = 1: stay inside the row (K pad chunk is zero)
    qf[ks] = *(const bf16x8*)(qrow + d);
  }

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_part = 0.f;

  // K/V tiles go global -> LDS directly (global_load_lds, 16 B per lane, lane-linear LDS image = rows at their padded pitch): nothing of
  // the next tile is held in registers while this one is computed (the register-staged form kept 24 VGPRs for it; with the V fragments
  // fetched in two halves the DH = 72 kernel fits 170 VGPRs = three waves per SIMD, and SigLIP's 640 workgroups are ONE round of 768
  // slots instead of 1.25 rounds of 512).  Lanes that map to a row's pad chunks are switched off: the pads were zeroed once.
  constexpr int KCP = KROW / 16, VCP = VROW / 16;            // chunks per row pitch
  constexpr int KN = (64 * KCP + NT - 1) / NT, VN = (64 * VCP + NT - 1) / NT;
  auto issue_tile = [&](int j, int buf) {
    const int c0 = j * 64;
#pragma unroll
    for (int i = 0; i < KN; ++i) {
      const int c = i * NT + tid;
      const int kr = c / KCP, kc = c - kr * KCP;
      if (kc < CPR && c < 64 * KCP) {
        const size_t trow = (size_t)min(c0 + kr, Lk - 1);
        __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(kb + trow * p.k_st * 2 + kc * 16), AKI_LDS_PTR(sK + buf * KTILE + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const int c = i * NT + tid;
      const int kr = c / VCP, kc = c - kr * VCP;
      if (kc < CPR && c < 64 * VCP) {
        const size_t trow = (size_t)min(c0 + kr, Lk - 1);
        __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(vb + trow * p.v_st * 2 + kc * 16), AKI_LDS_PTR(sV + buf * VTILE + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
    }
  };

  const int koff = l31 * KROW + h * 16;
  const int voff = (4 * h + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int jend = (Lk + 63) >> 6;

  __syncthreads();                     // the pad zeroing above is in LDS before any DMA piece lands beside it
  issue_tile(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing pending at loop entry (see mma_attn_bf16.hip)
  __syncthreads();

  for (int j = 0; j < jend; ++j) {
    if (j + 1 < jend) issue_tile(j + 1, (j + 1) & 1);      // the other buffer: every wave passed the barrier that ended tile j-1
    const int c0 = j * 64;
    const char* Kb = sK + (j & 1) * KTILE;
    const char* Vb = sV + (j & 1) * VTILE;
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    // All K fragments of the tile are fetched before the first MFMA (one LDS wait instead of one in front of every
    // MFMA pair: with 2 waves per SIMD that ~128-cycle LDS latency, 24 times per tile, was the dominant stall).
    bf16x8 ka[KS], kc[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      ka[ks] = *(const bf16x8*)(Kb + koff + ks * 32);
      kc[ks] = *(const bf16x8*)(Kb + koff + 32 * KROW + ks * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[ks], qf[ks], s1, 0, 0, 0);
    }
    // The V^T fragments do not depend on the softmax: the transposed reads of the first 32 keys are issued now and land under the VALU
    // work; those of keys 32-63 follow behind the first half's P V MFMAs (all four at once cost 24 more VGPRs - the third wave per SIMD).
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto read_v = [&](int ks4, s16x8 (&dst)[DT]) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const char* va = Vb + voff + ks4 * 16 * VROW + dt * 64;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + 8 * VROW));
        const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        dst[dt] = vv;
      }
    };
    s16x8 vfa[2][DT];
    read_v(0, vfa[0]);
    read_v(1, vfa[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (c0 + 64 > Lk) {   // last tile: keys >= Lk are masked (prefix in register order, see mma_attn_bf16.hip count_le)
      const int x = Lk - 1 - c0 - 4 * h;
      const int nv = x < 0 ? 0 : min(4 * (x >> 3) + min((x & 7) + 1, 4), 32);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = (r < nv) ? s0[r] : -INFINITY;
        s1[r] = (r + 16 < nv) ? s1[r] : -INFINITY;
      }
    }
    mfma_results_settle(s0, s1);   // the max3 chain below is inline asm
    float mx = max3_nc(s0[0], s0[1], s1[0]);
    mx = max3_nc(mx, s1[1], s0[2]);
#pragma unroll
    for (int r = 3; r < 16; ++r) mx = max3_nc(mx, s0[r], s1[r - 1]);
    mx = fmaxf(mx, s1[15]);
    mx = halves_max(mx) * p.scale_log2;
    const float m_new = fmaxf(m_run, mx);
    const bool moved = m_new != m_run;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], p.scale_log2, -m_new));
      s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], p.scale_log2, -m_new));
      ps += s0[r] + s1[r];
    }
    l_part = l_part * alpha + ps;
    if (__any(moved)) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
#pragma unroll
    for (int ks4 = 0; ks4 < 2; ++ks4) {
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s0[8 * ks4 + e];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vfa[ks4][dt]), pf, o[dt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      s16x8 vfb[2][DT];
      read_v(2, vfb[0]);
      read_v(3, vfb[1]);
#pragma unroll
      for (int ks4 = 0; ks4 < 2; ++ks4) {
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s1[8 * ks4 + e];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vfb[ks4][dt]), pf, o[dt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of tile j+1 have landed ...
    __syncthreads();                                         // ... and so have everybody's; tile j's buffer is free
  }

  const float l_tot = halves_sum(l_part);
  if (row < Lq) {
    const float inv = 1.0f / l_tot;
    bf16_t* orow = p.o + ((size_t)(b * Lq + row) * p.H + head) * DH + 4 * h;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + q4 * 8 + 4 * h;
        if (d < DH) {
          u32x2 pk = {pack_bf16x2(o[dt][4 * q4] * inv, o[dt][4 * q4 + 1] * inv), pack_bf16x2(o[dt][4 * q4 + 2] * inv, o[dt][4 * q4 + 3] * inv)};
          *(u32x2*)(orow + dt * 32 + q4 * 8) = pk;
        }
      }
    if (p.lse && h == 0) p.lse[(size_t)bh * Lq + row] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.69314718055994530942f;
  }
}

int attn_nc_bf16(const aki_attn_args* a, hipStream_t stream) {
  AttnNcParams p = {(const bf16_t*)a->q, (const bf16_t*)a->k, (const bf16_t*)a->v, (bf16_t*)a->o,
                    a->q_stride_b, a->q_stride_h, a->q_stride_t, a->k_stride_b, a->k_stride_h, a->k_stride_t,
                    a->v_stride_b, a->v_stride_h, a->v_stride_t, a->B, a->H, a->Lq, a->Lk, 0,
                    a->scale * 1.44269504088896340736f, a->lse};
  constexpr int NW = 4;
  p.nqt = (a->Lq + NW * 32 - 1) / (NW * 32);
  // 16-byte loads: every stride a multiple of 8 elements, bases 16-B aligned
  const long ss[] = {a->q_stride_b, a->q_stride_h, a->q_stride_t, a->k_stride_b, a->k_stride_h, a->k_stride_t,
                     a->v_stride_b, a->v_stride_h, a->v_stride_t};
  for (long s : ss)
    if (s % 8) return AKI_ERR_ALIGNMENT;
  AKI_CHECK_ALIGN16(a->q); AKI_CHECK_ALIGN16(a->k); AKI_CHECK_ALIGN16(a->v);
  if ((uintptr_t)a->o & 7) return AKI_ERR_ALIGNMENT;
  const dim3 grid(a->B * a->H * p.nqt), block(NW * 64);
  AKI_CLEAR_ERR();
  switch (a->Dh) {
    case 96: hipLaunchKernelGGL((attn_nc_bf16_kernel<NW, 96>), grid, block, 0, stream, p); break;
    case 72: hipLaunchKernelGGL((attn_nc_bf16_kernel<NW, 72>), grid, block, 0, stream, p); break;
    case 64: hipLaunchKernelGGL((attn_nc_bf16_kernel<NW, 64>), grid, block, 0, stream, p); break;
    case 32: hipLaunchKernelGGL((attn_nc_bf16_kernel<NW, 32>), grid, block, 0, stream, p); break;
    default: return AKI_ERR_UNSUPPORTED;
  }
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
