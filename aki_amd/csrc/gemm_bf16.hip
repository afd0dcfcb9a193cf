// gemm_bf16.hip - y = act(x W^T + bias) [+ residual] on CDNA4 MFMA, bf16 in / f32 accumulate.
//
// Replaces every nn.Linear on the AKI forward path (include/aki_mi355x.h: aki_linear_fwd) and, with
// the QKV_ROPE epilogue, stage 1 of the fused MMA op (HF:phi3/modeling_phi3.py:228-241).
//
// Orientation.  Both operands are K-contiguous ([M,K] activations, [N,K] nn.Linear weights), so both
// MFMA fragments are one 16-byte LDS read.  The product is computed "swapped": A-operand = W rows
// (features), B-operand = x rows (tokens), i.e. the accumulator tile is C^T[feature][token] with
//   token   = lane & 31                      (on the lanes)
//   feature = (reg&3) + 8*(reg>>2) + 4*(lane>>5)   (in the 16 accumulator registers)
// Every lane therefore owns runs of 4 consecutive features of ONE token: row-major bf16 stores are
// 8 bytes wide, per-token epilogues (RoPE with cos/sin[pos(token)], residual add, SwiGLU pairs) are
// lane-local, and rotate-half partners d <-> d+48 (48 = 6*8) sit in the same lane.
//
// Tile: 8 waves = 2 (features) x 4 (tokens); wave tile = TN*32 features x 64 tokens; BK = 64.
//   TN = 4: 256 x 256 block tile (generic);  TN = 3: 192 x 256 (two 96-wide heads, QKV+RoPE).
// LDS: 2 stages x (BN + 256) rows x 128 B.  Staging is global_load_lds (16 B/lane, 1 KiB/wave-
// instruction = 8 rows); the LDS image is lane-linear and the bank-conflict swizzle
// (chunk ^= (row>>1)&7) is applied on the SOURCE address and again on the ds_read_b128 address
// (cdna_hip_programming.md rule 21).  One barrier per K-step; tile k+1 streams in under tile k's MFMAs.
#include "aki_device.h"

namespace aki {

enum { EPI_PLAIN = 0, EPI_SWIGLU = 1, EPI_QKV_ROPE = 2 };

struct GemmParams {
  const bf16_t* x;
  const bf16_t* w;
  const bf16_t* bias;
  const bf16_t* residual;
  bf16_t* y;
  int M, N, K;  // N = weight rows
  int ldx, ldw, ldy, ldr;
  int res_row_mod;
  int act;
  int tiles_m, tiles_n;
  // QKV + RoPE
  bf16_t* q_out;
  bf16_t* k_out;
  bf16_t* v_out;
  const float* cos;
  const float* sin;
  const int* position_ids;
  int H, L;
};

template <int TN, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel(const GemmParams p) {
  constexpr int TM = 2, BK = 64;
  constexpr int WROWS = TN * 32;     // features per wave
  constexpr int BN = 2 * WROWS;      // features per block tile
  constexpr int BM = 4 * TM * 32;    // tokens per block tile (256)
  constexpr int ROWS = BN + BM;
  constexpr int STAGE_BYTES = ROWS * 128;
  constexpr int NLD = ROWS / 64;     // global_load_lds per thread per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- tile id: XCD-contiguous chunks, grouped so 32 concurrent tiles of an XCD share operands ----
  const int t = xcd_remap(blockIdx.x, gridDim.x);
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

  // ---- per-thread staging sources --------------------------------------------------------------
  const char* src[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int rowgroup = j * 8 + wave;
    const int row = rowgroup * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (rowgroup * 8 < BN) {
      int wrow;
      if (EPI == EPI_SWIGLU) {
        const int w_ = row / WROWS, within = row % WROWS, nb = within >> 5, i = within & 31;
        const int f = n0 + w_ * (WROWS / 2) + (nb % (TN / 2 > 0 ? TN / 2 : 1)) * 32 + i;
        wrow = (nb < TN / 2) ? min(f, n_out - 1) : n_out + min(f, n_out - 1);
      } else {
        wrow = min(n0 + row, p.N - 1);
      }
      src[j] = (const char*)(p.w + (size_t)wrow * p.ldw + chunk * 8);
    } else {
      const int xrow = min(m0 + row - BN, p.M - 1);
      src[j] = (const char*)(p.x + (size_t)xrow * p.ldx + chunk * 8);
    }
  }

  auto stage = [&](int s, int kt) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      char* dst = smem + s * STAGE_BYTES + (j * 8 + wave) * 1024;
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(src[j] + (size_t)kt * (BK * 2)), AKI_LDS_PTR(dst), 16, 0, 0);
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int n = 0; n < TN; ++n)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

  // per-lane fragment addressing: row byte offset + swizzled 16-B chunk for each k-step
  const int swz = (lane >> 1) & 7;  // == (row>>1)&7 because every block row base is a multiple of 32
  const int wbase = (wn * WROWS + l31) * 128;
  const int xbase = BN * 128 + (wm * TM * 32 + l31) * 128;

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // (vmcnt(0) + barrier): tile kt landed, the other buffer is no longer being read
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sb = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((2 * ks + h) ^ swz) << 4;
      bf16x8 a[TN], b[TM];
#pragma unroll
      for (int n = 0; n < TN; ++n) a[n] = *(const bf16x8*)(sb + wbase + n * 4096 + coff);
#pragma unroll
      for (int m = 0; m < TM; ++m) b[m] = *(const bf16x8*)(sb + xbase + m * 4096 + coff);
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int m = 0; m < TM; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n], b[m], acc[n][m], 0, 0, 0);
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  if (EPI == EPI_QKV_ROPE) {
    // wave = one 96-wide head slot of q|k|v
    const int hs = (n0 + wn * WROWS) / 96;
    if (hs >= 3 * p.H) return;
    const int which = hs / p.H, head = hs % p.H;
    bf16_t* outp = which == 0 ? p.q_out : (which == 1 ? p.k_out : p.v_out);
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const int mrow = m0 + wm * TM * 32 + m * 32 + l31;
      if (mrow >= p.M) continue;
      const int b = mrow / p.L, tt = mrow - b * p.L;
      bf16_t* dst = outp + ((size_t)(b * p.H + head) * p.L + tt) * 96 + 4 * h;
      if (which < 2) {
        const int pos = p.position_ids ? p.position_ids[mrow] : tt;
        const float* cp = p.cos + (size_t)pos * 96 + 4 * h;
        const float* sp = p.sin + (size_t)pos * 96 + 4 * h;
#pragma unroll
        for (int idx = 0; idx < 6; ++idx) {  // feature run d = idx*8 + 4h + e and its partner d + 48
          const f32x4 c4 = *(const f32x4*)(cp + idx * 8);
          const f32x4 s4 = *(const f32x4*)(sp + idx * 8);
          float lo[4], hi[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x1 = acc[idx / 4][m][4 * (idx % 4) + e];
            const float x2 = acc[(idx + 6) / 4][m][4 * ((idx + 6) % 4) + e];
            lo[e] = x1 * c4[e] - x2 * s4[e];
            hi[e] = x2 * c4[e] + x1 * s4[e];
          }
          u32x2 vlo = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3])};
          u32x2 vhi = {pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
          *(u32x2*)(dst + idx * 8) = vlo;
          *(u32x2*)(dst + idx * 8 + 48) = vhi;
        }
      } else {
#pragma unroll
        for (int idx = 0; idx < 12; ++idx) {
          const float* a4 = nullptr;
          (void)a4;
          u32x2 v = {pack_bf16x2(acc[idx / 4][m][4 * (idx % 4) + 0], acc[idx / 4][m][4 * (idx % 4) + 1]),
                     pack_bf16x2(acc[idx / 4][m][4 * (idx % 4) + 2], acc[idx / 4][m][4 * (idx % 4) + 3])};
          *(u32x2*)(dst + idx * 8) = v;
        }
      }
    }
    return;
  }

  constexpr int NOUT_BLOCKS = (EPI == EPI_SWIGLU) ? TN / 2 : TN;
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    const int mrow = m0 + wm * TM * 32 + m * 32 + l31;
    if (mrow >= p.M) continue;
    bf16_t* yrow = p.y + (size_t)mrow * p.ldy;
    const bf16_t* rrow = nullptr;
    if (p.residual) rrow = p.residual + (size_t)(p.res_row_mod > 0 ? mrow % p.res_row_mod : mrow) * p.ldr;
#pragma unroll
    for (int n = 0; n < NOUT_BLOCKS; ++n) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int f = n0 + wn * (EPI == EPI_SWIGLU ? WROWS / 2 : WROWS) + n * 32 + q4 * 8 + 4 * h;
        if (f >= n_out) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (EPI == EPI_SWIGLU) {
            const float g = acc[n][m][4 * q4 + e], u = acc[n + TN / 2][m][4 * q4 + e];
            v[e] = u * silu(g);
          } else {
            v[e] = acc[n][m][4 * q4 + e];
          }
        }
        if (EPI == EPI_PLAIN) {
          if (p.bias) {
            const u32x2 bb = *(const u32x2*)(p.bias + f);
            v[0] += bf16_lo(bb[0]); v[1] += bf16_hi(bb[0]); v[2] += bf16_lo(bb[1]); v[3] += bf16_hi(bb[1]);
          }
          if (p.act == AKI_ACT_GELU_ERF) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          } else if (p.act == AKI_ACT_GELU_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
          }
        }
        if (rrow) {
          const u32x2 rr = *(const u32x2*)(rrow + f);
          v[0] += bf16_lo(rr[0]); v[1] += bf16_hi(rr[0]); v[2] += bf16_lo(rr[1]); v[3] += bf16_hi(rr[1]);
        }
        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *(u32x2*)(yrow + f) = o;
      }
    }
  }
}

template <int TN, int EPI>
static int launch_gemm(GemmParams& p, hipStream_t stream) {
  constexpr int BN = 2 * TN * 32, BM = 256;
  constexpr int SMEM = 2 * (BN + BM) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_bf16_kernel<TN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return AKI_ERR_LAUNCH;
    attr_set = true;
  }
  const int n_out = (EPI == EPI_SWIGLU) ? p.N / 2 : p.N;
  const int bn_out = (EPI == EPI_SWIGLU) ? BN / 2 : BN;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (n_out + bn_out - 1) / bn_out;
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL((gemm_bf16_kernel<TN, EPI>), dim3(p.tiles_m * p.tiles_n), dim3(512), SMEM, stream, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
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
  if (a->act == AKI_ACT_SWIGLU) {
    if (a->bias) return AKI_ERR_UNSUPPORTED;
    return launch_gemm<4, EPI_SWIGLU>(p, stream);
  }
  return launch_gemm<4, EPI_PLAIN>(p, stream);
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
  return launch_gemm<3, EPI_QKV_ROPE>(p, stream);
}

}  // namespace aki
