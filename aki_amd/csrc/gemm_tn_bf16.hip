// Weight-gradient GEMM on operands as they lie in memory:  C[i][j] = sum_c A[c][i] * B[c][j]   (bf16 in, f32 accumulate, bf16 out).
//
// dW = dY^T X  (LinearFn.backward, aki_amd/train_ops.py; reference: torch autograd of nn.Linear under train/train_utils.py:242-252): the contraction
// index c is the TOKEN index, which is the row index of both operands - for aki_linear_fwd both would have to be transposed first (two HBM passes
// per weight gradient, 482 transpose launches = 9.6 ms of the 159 ms training step).  gfx950 reads an MFMA operand out of a row-major
// [contraction][16 columns] LDS block with ds_read_b64_tr_b16, so this kernel stages both tiles exactly as they are in memory:
//
//   tile          256 (i) x 256 (j), 64 contraction rows per step; 8 waves as 2 (i) x 4 (j), wave tile 128 x 64 = 8 x 4 accumulators of
//                 v_mfma_f32_16x16x32_bf16 (the j fragment is the first operand: a lane then holds 4 consecutive j of one i - 8-byte stores)
//   LDS           A tiles (64 rows x 512 B) on a three-deep ring, B tiles on a two-deep one = 160 KB, filled by buffer_load ... lds (16 B per lane, 1 KB per wave
//                 instruction = two tile rows).  The 32-byte piece q of row r sits at piece q ^ g(r), g(r) = (r & 3) | ((r >> 3) & 1) << 2 (applied on
//                 the SOURCE side - the DMA image is lane-linear): the 8 rows a 32-lane half of a transposed read touches (r, r+1, r+2, r+3 of two
//                 8-row groups) then cover all 64 banks once.
//   K loop        counted wait for the step's tiles, one barrier, issue B one step and A two steps ahead, then 16 fragment steps (see the loop).
//   tails         rows c >= Kc of the last step lie beyond the buffer descriptor (zeros); columns beyond I / J are clamped on the load, dropped on the store.
//
// HBM: reads (I + J) * Kc * 2 B per tile row / column panel, writes I * J * 2 B.  MFMA-bound like aki_linear_fwd: 2 * I * J * Kc FLOP.
#include "aki_device.h"
#include "attn_mma_common.h"

namespace aki {

struct GemmTnParams {
  const bf16_t* a;   // [Kc][lda], columns 0..I-1
  const bf16_t* b;   // [Kc][ldb], columns 0..J-1
  bf16_t* c;         // [I][ldc]
  int Kc, I, J;
  long lda, ldb, ldc;
  int tiles_i, tiles_j;
  int wide;          // c and ldc allow 16-byte stores
};

constexpr int TN_BI = 256, TN_BJ = 256, TN_BK = 64;
constexpr int TN_OP_BYTES = TN_BK * 512;            // one operand tile: 64 rows x 256 bf16

__global__ __launch_bounds__(512, 1) void gemm_tn_bf16_kernel(const GemmTnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 1, wj = wave >> 1;          // wave tile: i in [128 wi, +128), j in [64 wj, +64)

  // tile id: XCD-contiguous chunks, groups of 8 i-tiles share their j panel
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_j;
  const int group = t / per_group, first_i = group * GM;
  const int gsz = min(p.tiles_i - first_i, GM);
  const int ti = first_i + (t % per_group) % gsz, tj = (t % per_group) / gsz;
  const int i0 = ti * TN_BI, j0 = tj * TN_BJ;

  // ---- staging sources: wave instruction n (0..7) of a stage fills 1 KB = tile rows 2 blk, 2 blk + 1 of operand (n >> 2), blk = (n & 3) * 8 + wave
  unsigned voff[8];                                              // byte offset of the lane's chunk from the operand's base (step 0)
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int blk = (n & 3) * 8 + wave;
    const int row = 2 * blk + (lane >> 5);
    const int ch = lane & 31;                                    // 16-byte chunk of the LDS row
    const int g = (row & 3) | (((row >> 3) & 1) << 2);
    const int col = (((ch >> 1) ^ g) * 2 + (ch & 1)) * 8;        // first column (of the tile) this chunk holds
    if (n < 4) voff[n] = (unsigned)(((size_t)row * p.lda + min(i0 + col, p.I - 8)) * 2);
    else voff[n] = (unsigned)(((size_t)row * p.ldb + min(j0 + col, p.J - 8)) * 2);
  }
  const size_t step_a = (size_t)TN_BK * p.lda * 2, step_b = (size_t)TN_BK * p.ldb * 2;
  // The A tiles sit on a three-deep ring (asked for two K-steps ahead: both operands are activations that come out of HBM, and a panel of A is shared by
  // fewer concurrently running tiles than a panel of B), the B tiles on a two-deep one: 3 x 32 KB + 2 x 32 KB = the CU's 160 KB.
  // Buffer loads: the address of a piece is (descriptor base) + the lane's fixed 32-bit offset + a wave-uniform SGPR offset - no vector arithmetic per piece
  // (≈55 VALU instructions of 64-bit address selects per K-step beside the MFMAs cost 8 %), and rows >= Kc of the last step lie beyond the descriptor's
  // size and come back as zeros.
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, (short)0, (int)(((size_t)(p.Kc - 1) * p.lda + p.I) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, (short)0, (int)(((size_t)(p.Kc - 1) * p.ldb + p.J) * 2), 0x00020000);
  auto piece_a = [&](int n, int slot, int kt) {       // n = 0..3: one 1 KB piece (two tile rows) of this wave
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, AKI_LDS_PTR(smem + slot * TN_OP_BYTES + (n * 8 + wave) * 1024), 16, (int)voff[n], (int)((unsigned)kt * (unsigned)step_a), 0, 0);
  };
  auto piece_b = [&](int n, int slot, int kt) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, AKI_LDS_PTR(smem + (3 + slot) * TN_OP_BYTES + (n * 8 + wave) * 1024), 16, (int)voff[4 + n], (int)((unsigned)kt * (unsigned)step_b), 0, 0);
  };
  auto stage_a = [&](int slot, int kt) {
#pragma unroll
    for (int n = 0; n < 4; ++n) piece_a(n, slot, kt);
  };
  auto stage_b = [&](int slot, int kt) {
#pragma unroll
    for (int n = 0; n < 4; ++n) piece_b(n, slot, kt);
  };

  // ---- transposed-read addresses: fragment cb (16 columns = the 32-byte piece cb of a row) for contraction rows 8 kg + {0..3} (+4: second read)
  const int kg = lane >> 4, li = lane & 15;
  const int gl = (li >> 2) | ((kg & 1) << 2);
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
  const unsigned rowoff = (unsigned)((kg * 8 + (li >> 2)) * 512 + (lane & 3) * 8);
  unsigned fa[8], fb[4];                                         // byte address in stage 0, k32 = 0, first read
#pragma unroll
  for (int n = 0; n < 8; ++n) fa[n] = lds0 + rowoff + (unsigned)(((wi * 8 + n) ^ gl) << 5);
  unsigned fa2[8];                                               // the same in A slot 2 (64 KB further: beyond a DS instruction's 16-bit offset)
#pragma unroll
  for (int n = 0; n < 8; ++n) fa2[n] = fa[n] + 2 * TN_OP_BYTES;
#pragma unroll
  for (int m = 0; m < 4; ++m) fb[m] = lds0 + 3 * TN_OP_BYTES + rowoff + (unsigned)(((wj * 4 + m) ^ gl) << 5);

  f32x4 acc[8][4];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.Kc + TN_BK - 1) / TN_BK;
  stage_b(0, 0);
  stage_a(0, 0);
  if (nk > 1) stage_a(1, 1);
  int aslot = 0;                                                 // kt % 3
  // fullc: steps kt+1 and kt+2 exist (no wave-uniform branches around the DMA pieces).  phc: kt % 6 as a compile-time constant (the hot loop is unrolled six
  // times: every ring slot is then an immediate - DS offsets, M0 values - and the loop carries no slot arithmetic), or -1: slots computed at run time (tail).
  auto kstep = [&](auto fullc, auto phc, int kt) {
    constexpr bool FULL = decltype(fullc)::value;
    constexpr int PH = decltype(phc)::value;
    // tiles A(kt), B(kt) have to be there; A(kt+1), the newest four pieces of this wave, may stay in flight (loads complete in issue order)
    if (FULL || kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // everyone's pieces have landed, and everyone is done reading step kt-1's slots
    // The step's eight DMA pieces are issued one at a time BEHIND the MFMAs of fragment steps 0-7 (B's four first: the wait above must not cover the A tile
    // asked for two steps ahead): issued in one block behind the barrier, the 64 pieces of the eight waves queue up in the CU's address path and every wave
    // sits in its VMEM issue while the matrix cores idle (gate_up 1080 -> 1158 TF/s).
    const bool more_b = FULL || kt + 1 < nk, more_a = FULL || kt + 2 < nk;
    constexpr int AS_C = PH >= 0 ? PH % 3 : 0, BS_C = PH >= 0 ? (PH & 1) : 0;
    const int a_next = PH >= 0 ? (PH + 2) % 3 : (aslot == 0 ? 2 : aslot - 1), b_next = PH >= 0 ? ((PH + 1) & 1) : ((kt + 1) & 1);
    const unsigned sa = PH >= 0 ? 0u : (unsigned)(aslot * TN_OP_BYTES), sb = PH >= 0 ? 0u : (unsigned)((kt & 1) * TN_OP_BYTES);
    if constexpr (PH < 0) aslot = aslot == 2 ? 0 : aslot + 1;
    constexpr int AOFF = AS_C == 1 ? TN_OP_BYTES : 0, BOFF = BS_C * TN_OP_BYTES;   // immediates (slot 2 of A goes through fa2)
    // One K-step = 8 pairs of fragment steps (step s: k-half h = s >> 3, i fragment n = s & 7; it consumes A fragment s - ring of four - and the four B
    // fragments of its half).  The transposed reads of a pair are issued one pair ahead of its eight MFMAs and the second half's B fragments during pair 0,
    // so the LDS time of all eight waves (768 of a K-step's 2048 matrix cycles) sits under MFMAs instead of in a lock-step phase behind the barrier.  DS reads
    // return in order: `lgkmcnt(N)` with N = the reads issued behind a pair is the exact wait for it.  One DMA piece follows each pair's MFMAs.
    u32x2 bl[2][4], bh[2][4], al[4], ah[4];
    auto issue_b = [&](auto hc) {
      constexpr int H = decltype(hc)::value, KO = H * 32 * 512;
#pragma unroll
      for (int m = 0; m < 4; ++m) { bl[H][m] = ds_read_tr<KO + BOFF>(fb[m] + sb); bh[H][m] = ds_read_tr<KO + BOFF + 4 * 512>(fb[m] + sb); }
    };
    auto issue_a = [&](auto sc) {
      constexpr int S = decltype(sc)::value, KO = (S >> 3) * 32 * 512, N = S & 7;
      const unsigned base = (AS_C == 2 ? fa2[N] : fa[N]) + sa;
      al[S & 3] = ds_read_tr<KO + AOFF>(base);
      ah[S & 3] = ds_read_tr<KO + AOFF + 4 * 512>(base);
    };
    issue_b(std::integral_constant<int, 0>{});
    issue_a(std::integral_constant<int, 0>{});
    issue_a(std::integral_constant<int, 1>{});
    static_for<8>([&](auto pc) {
      constexpr int P = decltype(pc)::value, S0 = 2 * P, S1 = 2 * P + 1, H = S0 >> 3, R0 = S0 & 3, R1 = S1 & 3;
      if constexpr (P == 0) issue_b(std::integral_constant<int, 1>{});
      if constexpr (P < 7) {
        issue_a(std::integral_constant<int, (P < 7 ? S0 + 2 : 0)>{});
        issue_a(std::integral_constant<int, (P < 7 ? S1 + 2 : 0)>{});
      }
      constexpr int BEHIND = (P == 0 ? 8 : 0) + (P < 7 ? 4 : 0);
      if constexpr (P == 0 || P == 4)
        asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(al[R0]), "+v"(ah[R0]), "+v"(al[R1]), "+v"(ah[R1]), "+v"(bl[H][0]), "+v"(bh[H][0]), "+v"(bl[H][1]), "+v"(bh[H][1]),
                     "+v"(bl[H][2]), "+v"(bh[H][2]), "+v"(bl[H][3]), "+v"(bh[H][3]) : "n"(BEHIND));
      else
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(al[R0]), "+v"(ah[R0]), "+v"(al[R1]), "+v"(ah[R1]) : "n"(BEHIND));
      static_for<2>([&](auto ec) {
        constexpr int S = S0 + decltype(ec)::value, N = S & 7, R = S & 3;
        const u32x4 av = {al[R][0], al[R][1], ah[R][0], ah[R][1]};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const u32x4 bv = {bl[H][m][0], bl[H][m][1], bh[H][m][0], bh[H][m][1]};
          acc[N][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv), __builtin_bit_cast(bf16x8, av), acc[N][m], 0, 0, 0);
        }
      });
      if constexpr (P < 4) { if (more_b) piece_b(P, b_next, kt + 1); }
      else { if (more_a) piece_a(P - 4, a_next, kt + 2); }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  int kt = 0;
  for (; kt + 7 < nk; kt += 6)
    static_for<6>([&](auto ph) { kstep(std::true_type{}, ph, kt + decltype(ph)::value); });
  for (; kt + 2 < nk; ++kt) kstep(std::true_type{}, std::integral_constant<int, -1>{}, kt);      // kt is a multiple of 6 here: aslot = 0 = kt % 3
  for (; kt < nk; ++kt) kstep(std::false_type{}, std::integral_constant<int, -1>{}, kt);

  // ---- epilogue: lane holds C[i = i0 + 128 wi + 16 n + (lane & 15)][j = j0 + 64 wj + 16 m + 4 kg + {0..3}]: 8 bytes per fragment.  Two neighbouring j fragments
  // trade halves across the 16-lane rows (v_permlane16_swap: the odd rows of the first operand against the even rows of the second), after which a lane holds
  // 8 consecutive j of fragment m + (kg & 1) - one 16-byte store instead of two 8-byte ones, 64 contiguous bytes per output row and store instruction.
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int i = i0 + wi * 128 + n * 16 + li;
    if (p.wide) {
#pragma unroll
      for (int m = 0; m < 4; m += 2) {
        unsigned a0 = pack_bf16x2(acc[n][m][0], acc[n][m][1]), a1 = pack_bf16x2(acc[n][m][2], acc[n][m][3]);
        unsigned b0 = pack_bf16x2(acc[n][m + 1][0], acc[n][m + 1][1]), b1 = pack_bf16x2(acc[n][m + 1][2], acc[n][m + 1][3]);
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
        const int j = j0 + wj * 64 + (m + (kg & 1)) * 16 + (kg >> 1) * 8;
        if (i < p.I && j < p.J) *(u32x4*)(p.c + (size_t)i * p.ldc + j) = u32x4{a0, a1, b0, b1};
      }
    } else if (i < p.I) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int j = j0 + wj * 64 + m * 16 + kg * 4;
        if (j < p.J) {
          const u32x2 o = {pack_bf16x2(acc[n][m][0], acc[n][m][1]), pack_bf16x2(acc[n][m][2], acc[n][m][3])};
          *(u32x2*)(p.c + (size_t)i * p.ldc + j) = o;
        }
      }
    }
  }
}

int gemm_tn_launch(const void* a, const void* b, void* c, int Kc, int I, int J, long lda, long ldb, long ldc, hipStream_t stream) {
  if (Kc <= 0 || I < 8 || J < 8 || I % 8 || J % 8 || lda % 8 || ldb % 8 || ldc % 4 || lda < I || ldb < J || ldc < J) return AKI_ERR_UNSUPPORTED;
  if (((size_t)a | (size_t)b) % 16 || (size_t)c % 8) return AKI_ERR_UNSUPPORTED;
  if ((((size_t)(Kc - 1) * lda + I) * 2) >> 32 || (((size_t)(Kc - 1) * ldb + J) * 2) >> 32) return AKI_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  GemmTnParams p;
  p.a = (const bf16_t*)a; p.b = (const bf16_t*)b; p.c = (bf16_t*)c;
  p.Kc = Kc; p.I = I; p.J = J; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.tiles_i = (I + TN_BI - 1) / TN_BI;
  p.tiles_j = (J + TN_BJ - 1) / TN_BJ;
  p.wide = (ldc % 8 == 0 && (size_t)c % 16 == 0) ? 1 : 0;
  constexpr int SMEM = 5 * TN_OP_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return AKI_ERR_LAUNCH;
    attr_set = true;
  }
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(p.tiles_i * p.tiles_j), dim3(512), SMEM, stream, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
