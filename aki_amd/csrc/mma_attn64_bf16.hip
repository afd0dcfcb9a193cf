// mma_attn64_bf16.hip - the long-sequence form of the span-driven modality-mutual attention core (bf16, head_dim 96).
//
// Same arithmetic and the same mask rule as mma_attn_bf16.hip (HF:phi3/modeling_phi3.py:145-167 under the reference's dense mask,
// src/vlm.py:410-443); a different machine mapping, chosen for sequences where the core is MFMA-bound (BASELINE.json configs[3],
// L = 4096: ~1000 FLOP per byte):
//
//   * ONE wave per SIMD (256-thread workgroup, one per CU, the whole 512-entry register file per lane) and TWO 32-row query
//     blocks per wave, A and B.  Every K fragment (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16) read from LDS feeds
//     two MFMAs instead of one - half the LDS bytes per FLOP of the 32-row kernel.
//   * The two blocks run half a tile apart so that the softmax VALU of one sits in the MFMA gaps of the other:
//         slot E(j):  MFMA  P_B V (tile j-1), K Q_B^T (tile j)      VALU  softmax of S_A(j)
//         slot O(j):  MFMA  P_A V (tile j),   K Q_A^T (tile j+1)    VALU  softmax of S_B(j)
//     The softmax is cut into 24 chunks, one per MFMA gap (sm_chunk); every MFMA is an inline-asm statement and
//     __builtin_amdgcn_sched_barrier(0) pins each chunk to its gap (MI355X guide, cycle constants: an MFMA 32x32x16 holds the
//     vector issue port for 8 of its 32 cycles; a gap hides ~5 plain VALU issues).
//   * O (2 x 48), Q (2 x 24) and the K fragments (48) live in ACCUMULATOR registers that only this file's asm statements name
//     (a[64:255], map below): the VALU never touches them, so the 256 architectural VGPRs are left to the scores, P, the V^T
//     fragments and the bookkeeping.  hipcc is kept out of that range by never being given a reason to use it (<= 256 VGPRs, no
//     "a" constraints); tools/attn64_audit.py checks the .s for compiler v_accvgpr_* and scratch after every edit.
//   * Fragments are reloaded in place: right after the LAST MFMA that reads a fragment of tile j its LDS read for tile j+1 is
//     issued into the same registers (an MFMA reads A/B at issue; the LDS round trip is > 64 cycles), so the reads ride in the
//     MFMA gaps too and one counted lgkmcnt per slot is all the waiting there is.
//   * Product build (THR != 0): the softmax is BLIND.  A rank's first tile goes through the exact path (row maximum, p, in one
//     piece behind the slot's MFMAs) and leaves every row a reference maximum; from there on p = exp2(s c - m_ref) is taken
//     without looking at it - no row maximum, no sums, no decision: 32 fma + 32 exp2 + 16 cvt per block and tile in the slot's 28
//     MFMA gaps.  bf16 has f32's exponent range, so P stays exact to rounding however far a later score exceeds m_ref; the row
//     sums come off the matrix pipe (mfma_ones: two 16x16x32 MFMAs per 16 keys against a 0/1 operand, 8 accumulator registers) and
//     are looked at ONCE per rank: a sum of 2^64 or more (or inf / NaN) sends the workgroup through that rank again with every tile
//     on the exact path.  Three loop bodies: fast tiles in runs that leave both blocks' per-lane hide unchanged (nothing per tile but
//     the softmax), bias tiles (diagonal, rectangle edges, padding) one at a time, and the exact serial iteration.
//     THR = 0 (lab variant, tests) keeps the 32-row kernel's running maximum and reproduces it bit for bit.
//
// K/V tiles (64 keys) arrive by global_load_lds in the 32-row kernel's LDS image (K: source-side XOR swizzle; V: plain rows read
// transposed); the ring holds three K and three V tiles, V one tile behind K: "unit" u = {K(u+1), V(u)} is what slot E(u) reads.
// One barrier per tile: iteration j waits for its own pieces of unit j (counted vmcnt, unit j+1 stays in flight), meets the
// other waves, and issues unit j+2.
//
// Work decomposition: the 32-row blocks of a (batch, head) pair are ranked by the columns they walk (as in the 32-row kernel,
// two blocks per lane: up to 128 blocks, L <= 4096; longer sequences keep position order); rank g of a pair = ranked blocks
// 8g .. 8g+7, two per wave; persistent workgroups snake over the ranks of their pair.
#include "attn_mma_common.h"

namespace aki {

// accumulator-register map (literal names inside the asm strings)
constexpr int A64_O = 64;     // O^T of block X, feature tile dt: a[64 + 48 X + 16 dt : +15]
constexpr int A64_Q = 160;    // Q fragment ks of block X:        a[160 + 24 X + 4 ks : +3]
constexpr int A64_KA = 208;   // K fragment ks, keys 0-31:        a[208 + 4 ks : +3]
constexpr int A64_KC = 232;   //                keys 32-63:       a[232 + 4 ks : +3]
constexpr int A64_L = 56;     // row sums of block X (product build): a[56 + 4 X], register 0 of a 16x16 accumulator tile (1-3 stay zero)
// a[0:55] are left to hipcc: when it runs out of VGPRs outside the tile loop it parks values in the LOWEST free accumulator
// registers (clobber lists do not keep it from doing so); tools/attn64_audit.py fails the build if it ever names a56 or above (a64 in the exact build, which has no row-sum tiles).
// (The row-sum tiles as "a"-constrained operands were tried: hipcc then copies them between register sets at loop joins with
// v_accvgpr_mov right behind the asm MFMA that writes them - it takes asm outputs as ready - and the sums are lost.)

template <int R> __device__ __forceinline__ void acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(R)); }
template <int R> __device__ __forceinline__ void acc_set(unsigned x) { asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(x), "n"(R)); }
template <int R> __device__ __forceinline__ float acc_get() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "n"(R));
  return x;
}
// a[R .. R+3] *= alpha (four at a time so that the reads, multiplies and writes do not wait on each other)
template <int R> __device__ __forceinline__ void acc_scale4(float alpha) {
  float t0, t1, t2, t3;
  asm volatile(
      "v_accvgpr_read_b32 %0, a%c5\n\tv_accvgpr_read_b32 %1, a%c6\n\tv_accvgpr_read_b32 %2, a%c7\n\tv_accvgpr_read_b32 %3, a%c8\n\t"
      "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t"
      "v_accvgpr_write_b32 a%c5, %0\n\tv_accvgpr_write_b32 a%c6, %1\n\tv_accvgpr_write_b32 a%c7, %2\n\tv_accvgpr_write_b32 a%c8, %3"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(alpha), "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
}
// S^T (VGPRs) = K fragment (a) * Q fragment (a) + S^T
template <int KA, int QA> __device__ __forceinline__ void mfma_qk(f32x16& s) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));
}
template <int KA, int QA> __device__ __forceinline__ void mfma_qk_c(f32x16& s, const f32x16& cin) {   // C from another register tuple
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], a[%c4:%c5], %1" : "=v"(s) : "v"(cin), "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));
}
template <int KA, int QA> __device__ __forceinline__ void mfma_qk_zero(f32x16& s) {   // FULL tiles: the constant 0 as C
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));
}
// The same two with the result's 20 wait states INSIDE the statement (hipcc takes an asm statement's outputs as ready when it ends): for
// MFMAs that sit under an if / else of their own - the rank prologue's first links - where hipcc copies the tile right behind them.
template <int KA, int QA> __device__ __forceinline__ void mfma_qk_settled(f32x16& s) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0\n\ts_nop 15\n\ts_nop 3" : "+v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));
}
template <int KA, int QA> __device__ __forceinline__ void mfma_qk_zero_settled(f32x16& s) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0\n\ts_nop 15\n\ts_nop 3" : "=v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));
}
// O^T (a) += V^T fragment (VGPRs) * P fragment (VGPRs)
template <int OA> __device__ __forceinline__ void mfma_pv(const u32x4 vv, const u32x4 pf) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vv), "v"(pf), "n"(OA), "n"(OA + 15));
}
// Row sums on the matrix pipe (product build): l[row] += sum over the 16 keys of a P fragment.  The fragment is the B operand of the
// 32x32x16 P V MFMAs - lane (n = lane & 31, g = lane >> 5) holds 8 keys of row n - and is handed AS IT IS to a 16x16x32 MFMA, which reads
// the same registers as B'[k' = 8 (lane >> 4) + i][col' = lane & 15]: k' groups 0 and 2 are rows col' (first / second 8 keys), groups 1 and
// 3 rows col' + 16.  The A operand picks the groups: output rows 0 and 8 sum groups {0, 2}, rows 4 and 12 groups {1, 3}; with the 16x16
// result layout (col = lane & 15, row = 4 (lane >> 4) + reg) register 0 of EVERY lane then holds the sum of row lane & 31 over all 16
// keys.  64 v_add_f32 per tile leave the vector port (a slot is VALU-issue bound) for 8 short MFMAs on a pipe with room.
template <int LA> __device__ __forceinline__ void mfma_ones(const u32x4 sel, const u32x4 pf) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(sel), "v"(pf), "n"(LA), "n"(LA + 3));
}
// 16 bytes of a Q row straight into accumulator registers (a load hipcc neither counts nor waits for: the rank prologue's vmcnt(0) does)
template <int QA, int OFF> __device__ __forceinline__ void q_frag_load(const void* row) {
  asm volatile("global_load_dwordx4 a[%c1:%c2], %0, off offset:%c3" ::"v"(row), "n"(QA), "n"(QA + 3), "n"(OFF) : "memory");
}
template <int KA, int OFF> __device__ __forceinline__ void k_frag_read(unsigned addr) {
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "n"(KA), "n"(KA + 3), "n"(OFF));
}

struct A64Blk {                 // one 32-row block of a wave
  int wq0;                      // first row; L = the rank has no such block (the wave's half idles)
  int jend;                     // tiles it walks
  int touch_lo, touch_hi, full_lo, full_hi;   // wave-uniform rectangle summary
  int row, rc0, rc1;            // per lane: its row and that row's unlock columns
  bool exists, alive, has_dead, row_alive, has_uniform, row_uniform;
  float m_ref, l;               // reference maximum (log2 domain) and the row sum against it (exact build; product build: mfma_ones)
  int l_dbg;                    // lab stamps: redo count
  float hide;                   // per lane, for the score tile in flight: 0, or -inf = this row sees nothing of the tile
  float thr_raw;                // (m_ref + THR) / (scale log2 e): a raw score above it raises the reference maximum
  unsigned long long fast, fullm;   // tile masks of the current window of 64 tiles (bit t <-> tile win + t): fast = FULL, HIDDEN or ROWWISE; fullm = FULL
};
// End of a softmax chunk: the values it produced are USED by an empty asm volatile statement.  asm volatile statements keep their
// order (every MFMA is one), so the chunk's instructions cannot be sunk below the next MFMA - which is what the IR passes did with
// sixteen chunks' exp2 work when only sched_barrier (a machine-scheduler fence) stood between the chunks.  Inputs only: an asm
// OUTPUT read by the next VALU instruction costs an s_nop (hipcc's asm boundary pad), one per gap.
template <class A> __device__ __forceinline__ void pin(const A& a) { asm volatile("" ::"v"(a)); }
template <class A, class B> __device__ __forceinline__ void pin(const A& a, const B& b) { asm volatile("" ::"v"(a), "v"(b)); }
template <class A, class B, class C> __device__ __forceinline__ void pin(const A& a, const B& b, const C& c) { asm volatile("" ::"v"(a), "v"(b), "v"(c)); }
template <class A, class B, class C, class D> __device__ __forceinline__ void pin(const A& a, const B& b, const C& c, const D& d) { asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d)); }
template <class A, class B, class C, class D, class E> __device__ __forceinline__ void pin(const A& a, const B& b, const C& c, const D& d, const E& e) {
  asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d), "v"(e));
}
template <class A, class B, class C, class D, class E, class F> __device__ __forceinline__ void pin(const A& a, const B& b, const C& c, const D& d, const E& e, const F& f) {
  asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f));
}
template <class A, class B, class C, class D, class E, class F, class G> __device__ __forceinline__ void pin(const A& a, const B& b, const C& c, const D& d, const E& e, const F& f, const G& g) {
  asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g));
}

struct A64Tmp {
  float m0, m1, m2, m3;        // four independent row-maximum chains (a wave alone on its SIMD stalls on every dependent VALU pair)
  float thr, nm, ps, T;        // raise threshold (raw score units), -m_ref + hide, running row sum, the pair sum waiting to enter it
  float a0, a1;                // exp2 arguments of the next score pair
  float ah0[4], ah1[4];        // product softmax: exp2 arguments of pairs 12-15, made early (the raise decision needs them at gap 20)
  float e0[16], e1[16];        // p = exp2(...) of the score registers, each alive for three chunks
  unsigned long long need;     // lanes whose tile maximum calls for a raise of the reference maximum
};

// One chunk of a block's softmax: what is issued in MFMA gap G of the other block's slot.  Inside a chunk no instruction depends
// on another one of the same chunk (except the short tails of chunks 4-6 and 23): every value is consumed one chunk after it is made.
//   0       threshold and exp2 offset of this tile (from the reference maximum as it stands)
//   1-4     row maximum: four chains of four links
//   5       chains merged, the two half rows merged (v_permlane32_swap), compared with the threshold
//   6       (rare) the reference maximum is raised: l, O and the threshold follow; first exp2 arguments
//   7-22    score pair r = G - 7: exp2 of pair r, arguments of pair r + 1, pair sum r - 1, row sum + pair sum r - 2, bf16 packing of
//           pairs two chunks old
//   23      the rest of the sum and of the packing
template <int THR, int OA, int G, int ABL>
__device__ __forceinline__ void sm_chunk(const f32x16& s0, const f32x16& s1, u32x4 (&pf)[4], A64Blk& X, A64Tmp& t, const float c, const float rc) {
  if constexpr ((ABL & 16) && G >= 1 && G <= 5) {
    if constexpr (G == 5) t.need = 0ull;
  } else if constexpr ((ABL & 1) && G >= 7) {
    if constexpr (G == 7) {
#pragma unroll
      for (int i = 0; i < 4; ++i) pf[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    }
  } else if constexpr (G == 0) {
    t.thr = X.hide < 0.f ? INFINITY : X.thr_raw;
    t.nm = -X.m_ref + X.hide;       // a hidden row: exp2(-inf) = 0 on every column
    pin(t.thr, t.nm);
  } else if constexpr (G == 1) {
    asm("v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %2, %10, %11, %12\n\tv_max3_f32 %3, %13, %14, %15"
        : "=&v"(t.m0), "=&v"(t.m1), "=&v"(t.m2), "=&v"(t.m3)
        : "v"(s0[0]), "v"(s0[1]), "v"(s1[0]), "v"(s0[4]), "v"(s0[5]), "v"(s1[4]), "v"(s0[8]), "v"(s0[9]), "v"(s1[8]), "v"(s0[12]), "v"(s0[13]), "v"(s1[12]));
    pin(t.m0, t.m1, t.m2, t.m3);
  } else if constexpr (G == 2) {
    asm("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %2, %2, %8, %9\n\tv_max3_f32 %3, %3, %10, %11"
        : "+v"(t.m0), "+v"(t.m1), "+v"(t.m2), "+v"(t.m3)
        : "v"(s1[1]), "v"(s0[2]), "v"(s1[5]), "v"(s0[6]), "v"(s1[9]), "v"(s0[10]), "v"(s1[13]), "v"(s0[14]));
    pin(t.m0, t.m1, t.m2, t.m3);
  } else if constexpr (G == 3) {
    asm("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %2, %2, %8, %9\n\tv_max3_f32 %3, %3, %10, %11"
        : "+v"(t.m0), "+v"(t.m1), "+v"(t.m2), "+v"(t.m3)
        : "v"(s0[3]), "v"(s1[2]), "v"(s0[7]), "v"(s1[6]), "v"(s0[11]), "v"(s1[10]), "v"(s0[15]), "v"(s1[14]));
    pin(t.m0, t.m1, t.m2, t.m3);
  } else if constexpr (G == 4) {
    asm("v_max_f32 %0, %0, %4\n\tv_max_f32 %1, %1, %5\n\tv_max_f32 %2, %2, %6\n\tv_max_f32 %3, %3, %7\n\tv_max3_f32 %0, %0, %1, %2"
        : "+v"(t.m0), "+v"(t.m1), "+v"(t.m2), "+v"(t.m3)
        : "v"(s1[3]), "v"(s1[7]), "v"(s1[11]), "v"(s1[15]));
    pin(t.m0, t.m3);
  } else if constexpr (G == 5) {
    float a = t.m0, b_;
    asm("v_max_f32 %0, %0, %2\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "=&v"(b_) : "v"(t.m3));
    t.m0 = a;                        // the row's maximum over the tile, raw score units, in both half rows
    if constexpr (THR == 0) t.need = __ballot(a * c + X.hide > X.m_ref);      // the 32-row kernel's rule, bit for bit
    else t.need = __ballot(a > t.thr);
    pin(t.m0);
  } else if constexpr (G == 6) {
    if (__builtin_expect(t.need != 0ull, 0)) {
      const float mx = t.m0 * c + X.hide;
      const float m_new = fmaxf(X.m_ref, mx);
      const float alpha = __builtin_amdgcn_exp2f(X.m_ref - m_new);
      X.m_ref = m_new;
      X.thr_raw = (m_new + (float)THR) * rc;
      X.l *= alpha;
      static_for<12>([&](auto I) { acc_scale4<OA + 4 * decltype(I)::value>(alpha); });
      t.nm = -m_new + X.hide;
    }
    t.ps = 0.f;
    t.a0 = __builtin_fmaf(s0[0], c, t.nm);
    t.a1 = __builtin_fmaf(s1[0], c, t.nm);
    pin(t.a0, t.a1, t.nm, t.ps);
  } else if constexpr (G >= 7 && G <= 22) {
    constexpr int r = G - 7;
    t.e0[r] = __builtin_amdgcn_exp2f(t.a0);
    t.e1[r] = __builtin_amdgcn_exp2f(t.a1);
    if constexpr (r < 15) {
      t.a0 = __builtin_fmaf(s0[r + 1], c, t.nm);
      t.a1 = __builtin_fmaf(s1[r + 1], c, t.nm);
    }
    if constexpr (r >= 2) t.ps += t.T;                                  // pair sum r - 2
    if constexpr (r >= 1) t.T = t.e0[r - 1] + t.e1[r - 1];
    if constexpr (r >= 2 && !(r & 1)) {           // s0 pair (r-2, r-1)
      constexpr int e = r - 2;
      pf[e >> 3][(e & 7) >> 1] = pack_bf16x2(t.e0[e], t.e0[e + 1]);
    }
    if constexpr (r >= 3 && (r & 1)) {            // s1 pair (r-3, r-2)
      constexpr int e = r - 3;
      pf[2 + (e >> 3)][(e & 7) >> 1] = pack_bf16x2(t.e1[e], t.e1[e + 1]);
    }
    if constexpr (r >= 2 && !(r & 1)) pin(t.e0[r], t.e1[r], t.a0, t.a1, t.ps, t.T, pf[(r - 2) >> 3]);
    else if constexpr (r >= 3) pin(t.e0[r], t.e1[r], t.a0, t.a1, t.ps, t.T, pf[2 + ((r - 3) >> 3)]);
    else pin(t.e0[r], t.e1[r], t.a0, t.a1, t.ps, t.T);
  } else if constexpr (G == 23) {
    t.ps += t.T;                                   // pair 14
    t.ps += t.e0[15] + t.e1[15];
    X.l += t.ps;
    pf[1][3] = pack_bf16x2(t.e0[14], t.e0[15]);
    pf[3][3] = pack_bf16x2(t.e1[14], t.e1[15]);
    pin(X.l, pf[1], pf[3]);
  }
}

// ---- the blind softmax of the product build: no sums, no decision -------------------------------------------------------------
// Once every row of a block has a reference maximum (its first tiles went through the checked path above), p = exp2(s c - m_ref) is
// taken WITHOUT looking at it: bf16 has f32's exponent range, so P, the row sum (on the matrix pipe: mfma_ones) and O stay finite
// and exact to rounding as long as no later score exceeds the reference by ~64 log2 units; whether that held is read off the row
// sum ONCE per rank (l < 2^64, the kernel body's "verification") and a rank that fails it is walked again through the checked
// path.  Per tile and block: 32 fma + 32 exp2 + 16 cvt over the slot's 28 MFMA gaps (16 beside P V + row sums, 12 beside K Q^T):
// element k (score register k / 2 of half k % 2) has its exp2 in gap 1 + 26 k / 32, its argument one gap earlier, its bf16 word one
// gap after its partner's exp2.
struct A64Blind { float nm; float a[32]; float e[32]; };
// Which gap element k's exp2 goes to.  The gaps are not equal: slot E's gaps carry fragment reloads (two V^T reads behind every P V
// MFMA, one K read behind every K Q^T MFMA), slot O's first six carry the tile's LDS-DMA pieces, and the gap behind a row-sum MFMA is
// 16 pipe cycles, not 32.  SLOT 0 = E, 1 = O; weights in vector-issue cycles a gap has to spare (stamps: tools/attn64_halves.py).
template <int SLOT> __host__ __device__ constexpr int bl_weight(int g) {       // g = 0..27: 16 gaps beside P V + row sums, 12 beside K Q^T
  if (g >= 16) return SLOT == 0 ? 15 : 20;
  const bool ones = (g & 3) == 3;
  if (SLOT == 0) return ones ? 7 : 9;
  if (g < 6) return ones ? 2 : 3;            // a DMA piece rides here
  return ones ? 6 : 14;
}
template <int SLOT> __host__ __device__ constexpr int bl_ge(int k) {
  int total = 0;
  for (int g = 1; g <= 26; ++g) total += bl_weight<SLOT>(g);
  // element k sits where the running weight passes (k + 1/2) / 32 of the total; gaps 1..26 (gap 0 makes the first arguments, 27 the last words)
  int run = 0;
  for (int g = 1; g <= 26; ++g) {
    run += bl_weight<SLOT>(g);
    if (64 * run >= (2 * k + 1) * total) return g;
  }
  return 26;
}
template <int G, int SLOT, int ABL>
__device__ __forceinline__ void sm_blind_chunk(const f32x16& s0, const f32x16& s1, u32x4 (&pf)[4], const A64Blk& X, A64Blind& t, const float c) {
  if constexpr (ABL & 1) {           // lab: no softmax VALU at all (P = 1)
    if constexpr (G == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) pf[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    }
    return;
  }
  if constexpr (G == 0) t.nm = -X.m_ref + X.hide;
  static_for<32>([&](auto K) {
    constexpr int k = decltype(K)::value, r = k >> 1;
    if constexpr (bl_ge<SLOT>(k) == G) { t.e[k] = __builtin_amdgcn_exp2f(t.a[k]); pin(t.e[k]); }
  });
  static_for<32>([&](auto K) {
    constexpr int k = decltype(K)::value, r = k >> 1;
    if constexpr (bl_ge<SLOT>(k) - 1 == G) { t.a[k] = __builtin_fmaf((k & 1) ? s1[r] : s0[r], c, t.nm); pin(t.a[k]); }
  });
  static_for<16>([&](auto W) {          // word w: half w & 1, score registers 2 (w >> 1) and 2 (w >> 1) + 1 = elements k0 and k0 + 2
    constexpr int w = decltype(W)::value, hf = w & 1, r = 2 * (w >> 1), k0 = 2 * r + hf;
    if constexpr (bl_ge<SLOT>(k0 + 2) + 1 == G) {
      const unsigned wd = pack_bf16x2(t.e[k0], t.e[k0 + 2]);
      pin(wd);
      pf[2 * hf + (r >> 3)][(r & 7) >> 1] = wd;
    }
  });
}
// ---- a rank's FIRST tile, in the gaps (product build) ----------------------------------------------------------------------------
// Nothing has been accumulated yet (O and the row sums are zero, no row has a reference maximum), so the exact softmax of this tile is
// the row maximum followed by the blind arithmetic against it - no rescale.  The maximum takes the gaps beside P V and the row sums
// (four chains of max3, merged and swapped across the half rows by gap 9; gap 10 sets m_ref), the 32 elements the twelve gaps beside K Q^T
// and the last four of the first half: two fma + two exp2 per gap, words packed one gap behind.  The serial form of this tile (sm_redo
// behind empty gaps) cost a workgroup ~2000 cycles per rank.
struct A64First { float m0, m1, m2, m3, nm; float a[32], e[32]; };
template <int G>
__device__ __forceinline__ void sm_first_chunk(const f32x16& s0, const f32x16& s1, u32x4 (&pf)[4], A64Blk& X, A64First& t, const float c) {
  if constexpr (G == 1) {
    asm("v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %2, %10, %11, %12\n\tv_max3_f32 %3, %13, %14, %15"
        : "=&v"(t.m0), "=&v"(t.m1), "=&v"(t.m2), "=&v"(t.m3)
        : "v"(s0[0]), "v"(s0[1]), "v"(s1[0]), "v"(s0[4]), "v"(s0[5]), "v"(s1[4]), "v"(s0[8]), "v"(s0[9]), "v"(s1[8]), "v"(s0[12]), "v"(s0[13]), "v"(s1[12]));
    pin(t.m0, t.m1, t.m2, t.m3);
  } else if constexpr (G == 3) {
    asm("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %2, %2, %8, %9\n\tv_max3_f32 %3, %3, %10, %11"
        : "+v"(t.m0), "+v"(t.m1), "+v"(t.m2), "+v"(t.m3)
        : "v"(s1[1]), "v"(s0[2]), "v"(s1[5]), "v"(s0[6]), "v"(s1[9]), "v"(s0[10]), "v"(s1[13]), "v"(s0[14]));
    pin(t.m0, t.m1, t.m2, t.m3);
  } else if constexpr (G == 5) {
    asm("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %2, %2, %8, %9\n\tv_max3_f32 %3, %3, %10, %11"
        : "+v"(t.m0), "+v"(t.m1), "+v"(t.m2), "+v"(t.m3)
        : "v"(s0[3]), "v"(s1[2]), "v"(s0[7]), "v"(s1[6]), "v"(s0[11]), "v"(s1[10]), "v"(s0[15]), "v"(s1[14]));
    pin(t.m0, t.m1, t.m2, t.m3);
  } else if constexpr (G == 7) {
    asm("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %2, %2, %6, %7\n\tv_max3_f32 %0, %0, %1, %2\n\tv_max_f32 %0, %0, %3"
        : "+v"(t.m0), "+v"(t.m1), "+v"(t.m2), "+v"(t.m3)
        : "v"(s1[3]), "v"(s1[7]), "v"(s1[11]), "v"(s1[15]));
    pin(t.m0);
  } else if constexpr (G == 9) {
    t.m0 = halves_max(t.m0);
    pin(t.m0);
  } else if constexpr (G == 10) {
    const float mx = t.m0 * c + X.hide;
    X.m_ref = fmaxf(X.m_ref, mx);
    t.nm = -X.m_ref + X.hide;
    pin(t.nm, X.m_ref);
  }
  // element k (score register k / 2 of half k % 2): arguments in gap 11 + k / 2 (two per gap), exp2 one gap later, words one more
  static_for<32>([&](auto K) {
    constexpr int k = decltype(K)::value, r = k >> 1;
    if constexpr (12 + k / 2 == G) { t.e[k] = __builtin_amdgcn_exp2f(t.a[k]); pin(t.e[k]); }
  });
  static_for<32>([&](auto K) {
    constexpr int k = decltype(K)::value, r = k >> 1;
    if constexpr (11 + k / 2 == G) { t.a[k] = __builtin_fmaf((k & 1) ? s1[r] : s0[r], c, t.nm); pin(t.a[k]); }
  });
  static_for<16>([&](auto W) {
    constexpr int w = decltype(W)::value, hf = w & 1, r = 2 * (w >> 1), k0 = 2 * r + hf;
    if constexpr (13 + (k0 + 2) / 2 == G || (G == 27 && 13 + (k0 + 2) / 2 > 27)) {
      const unsigned wd = pack_bf16x2(t.e[k0], t.e[k0 + 2]);
      pin(wd);
      pf[2 * hf + (r >> 3)][(r & 7) >> 1] = wd;
    }
  });
}
// the exact path for a tile whose optimistic pass overflowed the bound (rare; cold)
template <int THR, int OA>
__device__ __forceinline__ void sm_redo(const f32x16& s0, const f32x16& s1, u32x4 (&pf)[4], A64Blk& X, A64Tmp& t, const float c) {
  const float m = tile_max32(s0, s1);
  const float mx = halves_max(m) * c + X.hide;
  const float m_new = fmaxf(X.m_ref, mx);
  const float alpha = __builtin_amdgcn_exp2f(X.m_ref - m_new);
  // a rank's first tile (no row of the wave's block has a reference yet: O and the row sums are still zero) has nothing to rescale
  const bool started = __ballot(X.m_ref > -1e29f) != 0ull;
  X.m_ref = m_new;
  if (started) {
    {                                 // the row sum lives on the matrix pipe's side (mfma_ones); its last MFMA is a slot old
      float lt;
      asm volatile("v_accvgpr_read_b32 %0, a%c2\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a%c2, %0" : "=&v"(lt) : "v"(alpha), "n"(A64_L + 4 * ((OA - A64_O) / 48)));
    }
    static_for<12>([&](auto I) { acc_scale4<OA + 4 * decltype(I)::value>(alpha); });
  }
  const float nm = -m_new + X.hide;
  // two score registers of each half at a time, packed at once: four temporaries alive (the slot's whole state is - a redo that
  // kept all 32 p made hipcc park ~60 registers in accumulator registers, up into this file's own)
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const float a = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, nm)), b = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r + 1], c, nm));
    const float d = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, nm)), e = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r + 1], c, nm));
    const unsigned w0 = pack_bf16x2(a, b), w1 = pack_bf16x2(d, e);
    pin(w0, w1);
    pf[r >> 3][(r & 7) >> 1] = w0;
    pf[2 + (r >> 3)][(r & 7) >> 1] = w1;
  }
}
// THR: 0 = the exact build (running maximum per tile, the 32-row kernel's arithmetic bit for bit; lab and tests); anything else = the
// product build (blind softmax, verified per rank - see the head of the file).
// ABL (lab library only; the product build is instantiated with them, read in CYCLES through the stamps - tools/attn64_ablate.py):
// timing ablations with wrong results: 1 no softmax VALU (P = 1), 2 no LDS-DMA and no vmcnt wait, 4 no fragment reloads and no
// lgkmcnt waits, 8 no tile barrier; right results: 64 every tile through the exact serial path, 512 cycle stamps (prologue / loops /
// epilogue / barrier wait), 1024 stamps around the four halves of the blind iteration.
template <int THR, int ABL>
__global__ __launch_bounds__(256, 1) void mma_attn64_bf16_kernel(const AttnParams p) {
  constexpr int NT = 256;
  constexpr int OROW = 208;              // output staging row: 192 B + 16 (the 8-B writes of 32 rows land 2-way instead of 8-way conflicted)
  constexpr int LDS_RING = NSTAGE * KTILE + NSTAGE * VTILE, LDS_VB = MAX_VB_WORDS * 8, LDS_OUT = 8 * 32 * OROW;
  // ring | valid words | output staging of its own (the next rank's first tiles stream into the ring WHILE this rank's outputs are staged)
  __shared__ __attribute__((aligned(16))) char smem[LDS_RING + LDS_VB + LDS_OUT + 16];   // + the two verification words (below)
  char* const sK = smem;
  char* const sV = smem + NSTAGE * KTILE;
  // the sample's valid-column words: read only on the bias path, through asm (a load hipcc can see would be given a vmcnt(0) that
  // drains the K/V ring); in registers they were eight VGPRs of cold state that pushed hot state into spills
  const unsigned sVB_a = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)(smem + NSTAGE * KTILE + NSTAGE * VTILE));
  // the accumulator registers named in the asm strings below: declared once, ALL of them: the kernel descriptor allocates them and hipcc's VGPR-to-AGPR spilling only takes accumulator registers no instruction of the function names
  if constexpr (THR != 0) asm volatile("" ::: "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
  asm volatile("" ::: "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  // product build: the A operand of the row-sum MFMAs (mfma_ones) and the workgroup's two verification words (ranks alternate)
  // A[row = lane & 15][k' group = lane >> 4] = 1 for (rows 0, 8; groups 0, 2) and (rows 4, 12; groups 1, 3): lanes 0 8 32 40 / 20 28 52 60
  const unsigned ones_w = (((lane & 7) == 0 && ((lane >> 4) & 1) == 0) || ((lane & 7) == 4 && ((lane >> 4) & 1) == 1)) ? 0x3f803f80u : 0u;
  const u32x4 sel = {ones_w, ones_w, ones_w, ones_w};
  const unsigned sFlag_a = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)(smem + LDS_RING + LDS_VB + LDS_OUT));
  if constexpr (THR != 0) {
    if (tid == 0) { *(unsigned*)(smem + LDS_RING + LDS_VB + LDS_OUT) = 0u; *(unsigned*)(smem + LDS_RING + LDS_VB + LDS_OUT + 4) = 0u; }
  }
  bool force_checked = false;          // this rank failed its verification: walk it again through the checked path
  int vpar = 0;

  // persistent workgroups, pair groups, splits: as in mma_attn_bf16.hip ("Persistent workgroups")
  const int grp = blockIdx.x / (p.group_bh * p.splits);
  const int bh0 = grp * p.group_bh;
  const int gbh = min(p.group_bh, p.B * p.H - bh0);
  const int within = blockIdx.x - bh0 * p.splits;
  const int sidx = within / gbh;
  const int bh = bh0 + within - sidx * gbh;
  const int b = bh / p.H, head = bh - b * p.H;
  const int L = p.L;
  const bf16_t* qb = p.q + ((size_t)bh * L) * 96;
  const char* kb = (const char*)(p.k + ((size_t)bh * p.kvcap) * 96);
  const char* vb_ = (const char*)(p.v + ((size_t)bh * p.kvcap) * 96);
  const int Lb = p.seq_lens ? min(p.seq_lens[b], L) : L;

  // ---- LDS-DMA pieces: piece I of a tile = 16-B chunks I*256 + tid of its 12-KiB image (row = chunk / 12) ----------------
  // Buffer addressing: descriptor = this pair's L rows of K (V), per-lane constant byte offset inside a tile's 12 KiB, tile and
  // piece in the scalar offset - no VALU per piece.  Rows at and beyond L (the last tile of a sequence that is no multiple of 64,
  // the prefetches past the end, unwritten rows of a KV cache) are out of the descriptor's range and arrive as ZEROS: their
  // columns are hidden by the valid words, and P = 0 meets V = 0.
  int d_ko[3];               // K: chunk swizzled on the source side (the LDS image is lane-linear)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int ch = i * NT + tid;
    const int kr = ch / 12, pos = ch - kr * 12;
    d_ko[i] = kr * 192 + (pos ^ ((kr >> 2) & 3)) * 16;
  }
  const int d_vo = tid * 16;  // V: the image is the tile
  const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, L * 192, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc((void*)vb_, 0, L * 192, 0x00020000);
  auto dma_k = [&](auto I, int t, int st) {
    constexpr int i = decltype(I)::value;
    if constexpr (ABL & 2) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, AKI_LDS_PTR(sK + st * KTILE + (i * NT + wave * 64) * 16), 16, d_ko[i], t * 12288, 0, 0);
  };
  auto dma_v = [&](auto I, int t, int st) {
    constexpr int i = decltype(I)::value;
    if constexpr (ABL & 2) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, AKI_LDS_PTR(sV + st * VTILE + (i * NT + wave * 64) * 16), 16, d_vo, t * 12288 + i * 4096, 0, 0);
  };

  // ---- once per workgroup: valid-column words (lane i holds words i, i+64, i+128, i+192), block extents and their ranking ----
  int first_bad = 0x7fffffff;          // valid words below this index are all ones (wave-uniform)
  {
    unsigned long long bad = 0ull;
    int base = 0;
#pragma unroll
    for (int q4 = 3; q4 >= 0; --q4) {
      const int w = lane + 64 * q4;
      unsigned long long vbw = ~0ull;
      if (w < p.nwords) {
        if (p.vbits) vbw = p.vbits[(size_t)b * p.nwords + w];
        else vbw = (w * 64 + 64 <= L) ? ~0ull : ((1ull << (L - w * 64)) - 1ull);
        if (wave == 0) *(unsigned long long*)(smem + NSTAGE * KTILE + NSTAGE * VTILE + 8 * w) = vbw;
      }
      const unsigned long long m = __ballot(vbw != ~0ull);
      if (m != 0ull) { bad = m; base = 64 * q4; }
    }
    if (bad != 0ull) first_bad = base + __builtin_ctzll(bad);
    __syncthreads();                     // the words are in LDS for every wave (once per workgroup)
  }
  auto valid_word = [&](int w) -> unsigned long long {          // wave-uniform w; only called for w >= first_bad
    const unsigned addr = sVB_a + 8u * (unsigned)min(w, p.nwords - 1);
    u32x2 r;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    const unsigned lo = __builtin_amdgcn_readfirstlane(r[0]), hi = __builtin_amdgcn_readfirstlane(r[1]);
    return ((unsigned long long)hi << 32) | lo;
  };

  // The sample's rectangles (at most AKI_MAX_RECTS = 8): lane i holds rectangle i, read once per workgroup; rect_at is four
  // v_readlane.  (As scalar loads behind their own lgkmcnt(0) they were ~200 cycles each, 4 * max_rects of them per rank.)
  u32x4 rect_reg = {0u, 0u, 0u, 0u};
  if (lane < p.max_rects) rect_reg = *(const u32x4*)(p.rects + (size_t)b * p.max_rects + lane);
  // rectangles in use: up to the last non-empty one (callers pad the table to a fixed width; every scan below is per rank or per run)
  const unsigned long long rv_ = __ballot(rect_reg[1] > rect_reg[0] && rect_reg[3] > rect_reg[2]);
  const int nrect = rv_ == 0ull ? 0 : 64 - (int)__builtin_clzll(rv_);
  auto rect_at = [&](int i) -> aki_mma_rect {       // wave-uniform i
    aki_mma_rect o;
    o.row_lo = (int)__builtin_amdgcn_readlane(rect_reg[0], i); o.row_hi = (int)__builtin_amdgcn_readlane(rect_reg[1], i);
    o.col_lo = (int)__builtin_amdgcn_readlane(rect_reg[2], i); o.col_hi = (int)__builtin_amdgcn_readlane(rect_reg[3], i);
    return o;
  };
  // columns a 32-row block starting at r0 walks (mma_attn_bf16.hip, block_extent)
  auto block_extent = [&](int r0) -> int {
    if (r0 >= L) return -1;
    int ext = min(r0 + 32, L);
    for (int i = 0; i < nrect; ++i) {
      const aki_mma_rect r = rect_at(i);
      if (r.row_hi > r.row_lo && r.col_hi > r.col_lo && r.row_lo < r0 + 32 && r.row_hi > r0) ext = max(ext, min(r.col_hi, L));
    }
    if (p.dead_uniform && min(r0 + 32, L) > Lb) ext = L;
    return ext;
  };
  const int nblk = (L + 31) >> 5;
  const bool sched = nblk <= 128;                 // kernel-uniform: ranked order (lane i <-> blocks i and i + 64)
  int er0 = 0xff, er1 = 0xff;        // lane i: (extent << 8) | rank of blocks i and i + 64 (rank 0xff: no such block) - cold state, kept small
  if (sched) {
    const int ext0 = block_extent(32 * lane);
    const int ext1 = block_extent(32 * (lane + 64));
    const int key0 = ext0 < 0 ? -1 : ext0 * 256 + lane;          // unique; later block first on ties
    const int key1 = ext1 < 0 ? -1 : ext1 * 256 + lane + 64;
    int r0 = 0, r1 = 0;
    for (int i = 0; i < 64; ++i) {
      const int k0 = __builtin_amdgcn_readlane(key0, i), k1 = __builtin_amdgcn_readlane(key1, i);
      r0 += (k0 > key0 ? 1 : 0) + (k1 > key0 ? 1 : 0);
      r1 += (k0 > key1 ? 1 : 0) + (k1 > key1 ? 1 : 0);
    }
    er0 = ext0 >= 0 ? (ext0 << 8) | r0 : 0xff;
    er1 = ext1 >= 0 ? (ext1 << 8) | r1 : 0xff;
  }
  // block with rank R of the ranked order: first row and extent (wave-uniform); no such block -> (L, 0)
  auto ranked_block = [&](int R, int& wq0, int& ext) {
    wq0 = L; ext = 0;
    const unsigned long long m0 = __ballot((er0 & 0xff) == R);
    const unsigned long long m1 = __ballot((er1 & 0xff) == R);
    if (m0 != 0ull) {
      const int l = __builtin_ctzll(m0);
      wq0 = 32 * l; ext = __builtin_amdgcn_readlane(er0, l) >> 8;
    } else if (m1 != 0ull) {
      const int l = __builtin_ctzll(m1);
      wq0 = 32 * (l + 64); ext = __builtin_amdgcn_readlane(er1, l) >> 8;
    }
  };

  // per-lane LDS read offsets (mma_attn_bf16.hip): K rows with the XOR swizzle on the low two chunk bits - fragment ks reads
  // chunk (2 ks + h) ^ kswz = ((2 ks) & ~3) + (((2 (ks & 1)) + h) ^ kswz): one lane address per parity of ks, the rest immediate
  const int kswz = (l31 >> 2) & 3;
  const int k_even = l31 * KROW + ((h ^ kswz) << 4);
  const int k_odd = l31 * KROW + (((2 + h) ^ kswz) << 4);
  const int voff = (4 * h + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const unsigned sK_a = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)sK);
  const unsigned sV_a = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)sV);
  const float c = p.scale_log2;
  const float rc = 1.0f / c;

  // lab stamps (ABL & 512): shader-clock cycles of prologue / tile loops / epilogue, tiles walked, and the 100 MHz real-time
  // counter over the whole workgroup -> lse[8 * blockIdx .. +7] (timing build: its lse output is not an lse)
  unsigned long long st_pro = 0, st_loop = 0, st_epi = 0, st_tiles = 0, st_t = 0, st_rt0 = 0, st_redo = 0, st_wait = 0, st_dma = 0;
  unsigned long long st_h[4] = {0, 0, 0, 0}, st_n = 0, st_slow = 0, st_nslow = 0, st_full = 0;   // ... and the bias / exact iterations whole (barrier wait included)      // ABL & 1024: the four halves of the BLIND iterations (wave 0..3 each its own), and how many
  if constexpr (ABL & 512) { st_rt0 = __builtin_amdgcn_s_memrealtime(); }

  // ---- the next rank's blocks, Q rows and first K/V tiles, asked for one rank ahead ------------------------------------------
  // Stamps of the first working build: 9-10 % of a workgroup's time went into rank prologues that waited for Q and K(0) with nothing
  // to overlap.  Now the rank search, the Q loads and the LDS-DMA of K(0), {K(1), V(0)}, {K(2), V(1)} of rank r + 1 are issued right
  // behind the tile loops of rank r (the ring is idle from there on - the two units the loops ask for past the rank's end are waited for first - and the
  // outputs are staged in LDS of their own) and land under its last P V MFMAs and its epilogue.
  struct { int wq0A, wq0B, jendA, jendB, jend; } nx = {0, 0, 0, 0, 0};
  auto rank_of = [&](int kk) -> int { return kk * p.splits + ((kk & 1) ? p.splits - 1 - sidx : sidx); };   // the snake over the pair's ranks
  auto prefetch_rank = [&](int g) {
    int wq0A = L, wq0B = L, extA = 0, extB = 0, hi_col = 0;
    if (sched) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        int wa, ea, wb, eb;
        ranked_block(8 * g + 2 * w, wa, ea);
        ranked_block(8 * g + 2 * w + 1, wb, eb);
        hi_col = max(hi_col, max(ea, eb));
        if (w == wave) { wq0A = wa; extA = ea; wq0B = wb; extB = eb; }
      }
    } else {                                     // position order, late rows first; a wave's blocks are neighbours
      const int q0 = (p.nqt - 1 - g) * 256;
      const int e = max(block_extent(q0 + 32 * (lane & 7)), 0);
#pragma unroll
      for (int w = 0; w < 8; ++w) hi_col = max(hi_col, __builtin_amdgcn_readlane(e, w));
      wq0A = min(q0 + 64 * wave, L); extA = __builtin_amdgcn_readlane(e, 2 * wave);
      wq0B = min(q0 + 64 * wave + 32, L); extB = __builtin_amdgcn_readlane(e, 2 * wave + 1);
    }
    nx.wq0A = __builtin_amdgcn_readfirstlane(wq0A); nx.wq0B = __builtin_amdgcn_readfirstlane(wq0B);
    nx.jendA = __builtin_amdgcn_readfirstlane((extA + 63) >> 6); nx.jendB = __builtin_amdgcn_readfirstlane((extB + 63) >> 6);
    nx.jend = (__builtin_amdgcn_readfirstlane(hi_col) + 63) >> 6;
    // K/V stream first (the oldest requests), then Q of both blocks
    static_for<3>([&](auto I) { dma_k(I, 0, 0); });
    static_for<3>([&](auto I) { dma_k(I, 1, 1); });
    static_for<3>([&](auto I) { dma_v(I, 0, 0); });
    static_for<3>([&](auto I) { dma_k(I, 2, 2); });
    static_for<3>([&](auto I) { dma_v(I, 1, 1); });
    // the Q fragments go straight into their accumulator registers (every score MFMA of the previous rank has been issued; its last
    // P V MFMAs and its epilogue do not name them): lane (q = l31, h) holds Q[q][16 ks + 8 h .. +7] of each block
    const bf16_t* ra = qb + (size_t)min(nx.wq0A + l31, L - 1) * 96 + 8 * h;
    const bf16_t* rb = qb + (size_t)min(nx.wq0B + l31, L - 1) * 96 + 8 * h;
    static_for<6>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      q_frag_load<A64_Q + 4 * ks, 32 * ks>(ra);
      q_frag_load<A64_Q + 24 + 4 * ks, 32 * ks>(rb);
    });
  };
  if (rank_of(0) < p.nqt) prefetch_rank(rank_of(0));

  for (int kk = 0;; ++kk) {
  const int g = rank_of(kk);   // this workgroup's next rank
  if (g >= p.nqt) break;
  if constexpr (ABL & 512) st_t = __builtin_amdgcn_s_memtime();

  // ---- which blocks: the rank's eight, this wave's two (found one rank ahead) ----------------------------------------------
  A64Blk A, B;
  A.wq0 = nx.wq0A; B.wq0 = nx.wq0B; A.jend = nx.jendA; B.jend = nx.jendB;
  const int jend = nx.jend;                       // the rank's K/V stream (workgroup-uniform)
  const int jend_w = max(A.jend, B.jend);         // this wave's part of it

  // ---- per block: rectangle summary, per-lane unlock range, row classes --------------------------------
  auto setup = [&](A64Blk& X) {
    X.exists = X.wq0 < L;
    X.row = X.wq0 + l31;
    X.touch_lo = 0x7fffffff; X.touch_hi = 0; X.full_lo = 0; X.full_hi = 0;
    X.rc0 = 0; X.rc1 = 0;
    for (int i = 0; i < nrect; ++i) {
      const aki_mma_rect r = rect_at(i);
      if (r.row_hi > r.row_lo && r.col_hi > r.col_lo) {
        if (r.row_lo < X.wq0 + 32 && r.row_hi > X.wq0) {
          X.touch_lo = min(X.touch_lo, r.col_lo);
          X.touch_hi = max(X.touch_hi, r.col_hi);
          if (r.row_lo <= X.wq0 && r.row_hi >= X.wq0 + 32) { X.full_lo = r.col_lo; X.full_hi = r.col_hi; }
        }
        if (X.row >= r.row_lo && X.row < r.row_hi) { X.rc0 = r.col_lo; X.rc1 = r.col_hi; }
      }
    }
    X.alive = X.wq0 < Lb;
    X.hide = 0.f;
    X.has_dead = min(X.wq0 + 32, L) > Lb;              // some rows of the block are beyond seq_len
    X.row_alive = X.row < Lb;
    X.has_uniform = p.dead_uniform && X.has_dead && X.exists;
    X.row_uniform = p.dead_uniform && !X.row_alive && X.exists;
    X.m_ref = -1e30f;
    X.l_dbg = 0;
    X.thr_raw = (-1e30f + (float)THR) * rc;
    X.l = 0.f;
  };
  setup(A);
  setup(B);
  static_for<96>([&](auto R) { acc_zero<A64_O + decltype(R)::value>(); });
  if constexpr (THR != 0) static_for<8>([&](auto R) { acc_zero<A64_L + decltype(R)::value>(); });

  // ---- tile classes (the 32-row kernel's, decided for 64 tiles at a time) -------------------------------------------------
  //   FULL     every column visible to every row of the block: the first MFMA of each chain takes the constant 0 as C;
  //   ROWWISE  right of the diagonal, every column valid, every rectangle that touches the block's rows covers the tile's columns
  //            or misses them: a row sees the whole tile or nothing of it.  Nothing is put into the accumulators: the softmax
  //            adds `hide` (0 or -inf per lane) to the exp2 offset, so a hidden row's p is 0 and its row sum unchanged;
  //            HIDDEN tiles (beyond the block's extent - the other block of the wave still needs the slot) are the case
  //            "every lane hidden";
  //   PARTIAL  everything else (the diagonal, tiles cut by a rectangle edge, padding, blocks with dead rows): the per-lane
  //            visibility word is expanded into a 0 / -inf bias, the initial value of the score accumulators.
  // Lane t decides tile win + t for its block; two ballots make the masks the tile loop tests one bit of per slot.
  auto block_masks = [&](A64Blk& X, int win) {
    const int tl = win + lane, c0 = tl * 64;
    const bool ok = X.exists && !X.has_dead && tl < first_bad;
    bool rfull = false, clean = true;
    for (int i = 0; i < nrect; ++i) {
      const aki_mma_rect r = rect_at(i);
      if (r.row_hi > r.row_lo && r.col_hi > r.col_lo && r.row_lo < X.wq0 + 32 && r.row_hi > X.wq0) {
        const bool inside = c0 >= r.col_lo && c0 + 64 <= r.col_hi;
        const bool outside = c0 + 64 <= r.col_lo || c0 >= r.col_hi;
        if (r.row_lo <= X.wq0 && r.row_hi >= X.wq0 + 32) rfull = rfull || inside;
        clean = clean && (inside || outside);
      }
    }
    const bool F = ok && (c0 + 63 <= X.wq0 || rfull);
    const bool R = ok && c0 > X.wq0 + 31 && clean;
    X.fullm = __ballot(F);
    X.fast = __ballot(F || R);
    if constexpr (ABL & 32) { X.fullm = ~0ull; X.fast = ~0ull; }
  };
  block_masks(A, 0);
  block_masks(B, 0);
  // a fast tile's per-lane hide (FULL: nobody; ROWWISE / HIDDEN: the rows whose rectangle does not cover the tile)
  auto hide_val = [&](const A64Blk& X, int jt) -> float {
    const int c0 = jt * 64;
    // wave-uniform, all scalar: -1 when the tile is FULL (nobody hidden), 0 when each lane's rectangle decides
    const unsigned fb = __builtin_amdgcn_readfirstlane((unsigned)((X.fullm >> (jt & 63)) & 1ull));     // (the masks may sit in spilled lanes: say it is uniform)
    const unsigned long long fm = fb ? ~0ull : 0ull;
    // Per lane, five VALU instructions: hide = fm ? 0 : (rc0 <= c0 && c0 + 64 <= rc1 ? 0 : -inf).  (The C form compiles to two
    // vector compares combined on the scalar unit plus a select; no measurable difference - kept as asm for its fixed length.)
    float hd;
    const float ninf = -INFINITY;
    asm("v_cmp_ge_i32_e64 vcc, %2, %3\n\t"            // c0 >= rc0
        "v_cndmask_b32_e64 %0, %1, 0, vcc\n\t"         // ? 0 : -inf
        "v_cmp_le_i32_e64 vcc, %4, %5\n\t"            // c0 + 64 <= rc1
        "v_cndmask_b32_e64 %0, %1, %0, vcc\n\t"        // ? keep : -inf
        "v_cndmask_b32_e64 %0, %0, 0, %6"              // FULL tile: 0
        : "=&v"(hd) : "v"(ninf), "s"(c0), "v"(X.rc0), "s"(c0 + 64), "v"(X.rc1), "s"(fm) : "vcc");
    return hd;
  };
  auto fast_hide = [&](A64Blk& X, int jt) { X.hide = hide_val(X, jt); };
  // for how many tiles from jt on (this window) a fast tile's hide stays what it is at jt: FULL tiles up to the first one that is
  // not; otherwise up to the next column where a rectangle that touches the block's rows begins or ends (tiles are "clean": none
  // straddles such a column)
  auto const_len = [&](const A64Blk& X, int jt) -> int {
    const unsigned long long f = X.fullm >> (jt & 63);
    if (f & 1ull) return f == ~0ull ? 64 : (int)__builtin_ctzll(~f);
    const int c0 = jt * 64;
    int nb = 0x7fffffff;
    for (int i = 0; i < nrect; ++i) {
      const aki_mma_rect r = rect_at(i);
      if (r.row_hi > r.row_lo && r.col_hi > r.col_lo && r.row_lo < X.wq0 + 32 && r.row_hi > X.wq0) {
        if (r.col_lo > c0) nb = min(nb, r.col_lo);
        if (r.col_hi > c0) nb = min(nb, r.col_hi);
      }
    }
    return nb == 0x7fffffff ? 64 : (nb - c0) >> 6;
  };
  // the per-lane part of a tile's bias: which of the lane's 32 score columns are hidden (bit r <-> score register r) and the hide the
  // tile travels with (product build, short form: a fast tile - no column hidden by the bias, the fast tile's hide)
  auto bias_hid = [&](const A64Blk& X, int j, int& hid, float& hide_new) {
    hid = 0;
    hide_new = 0.f;
    const bool short_form = THR != 0 && (j == 0 || (j & 63) != 0) && ((X.fast >> (j & 63)) & 1ull) != 0ull;       // (tile j + 1 of the next window: masks not made yet)
    if (short_form) {
      hide_new = hide_val(X, j);
    } else {
      const int c0 = j * 64;
      const unsigned long long vb = j < first_bad ? ~0ull : valid_word(j);
      const int base = c0 + 4 * h;
      unsigned valid;                                                       // valid columns, register order
      if ((vb & (vb + 1ull)) == 0ull) {                                     // wave-uniform: the bits form a prefix
        valid = low_bits(count_le(c0 + (int)__builtin_popcountll(vb) - 1 - base));
      } else {                                                              // holes in the 1-D mask (rare)
        const unsigned long long vbh = vb >> (4 * h);
        valid = 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) valid |= ((unsigned)(vbh >> (8 * k)) & 0xFu) << (4 * k);
      }
      const unsigned alive = (low_bits(count_le(X.row - base)) | (low_bits(count_le(X.rc1 - 1 - base)) & ~low_bits(count_le(X.rc0 - 1 - base)))) & valid;
      const unsigned uniform = low_bits(count_le(L - 1 - base));            // every column < L
      unsigned vis = X.row_uniform ? uniform : (X.row_alive ? alive : 0u);
      if (!X.exists) vis = 0u;
      hid = (int)~vis;
    }
  };
  auto tile_bias = [&](A64Blk& X, int j, f32x16& s0, f32x16& s1) {
    // Product build: a bias iteration biases both tiles it makes - block B's tile j and block A's tile j + 1 - because ONE of them
    // needs it (a block's diagonal tile, once per rank) or because the softmax mode asks for this iteration (a rank's first tile).
    // The other tile is fast by the masks more often than not: its bias is all zeros and its per-lane hide the fast tile's.  Both
    // cases write the score tiles with the SAME 64 instructions (an if / else over the tiles made hipcc copy them at the join: 64
    // v_mov per bias iteration) - every wave's slow iteration is a wait at the tile barrier for the other three.
    int hid;
    float hide_new;
    bias_hid(X, j, hid, hide_new);
    const int ninf = 0xFF800000;
    static_for<16>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      s0[r] = mask_bias<r>(hid, ninf);
      s1[r] = mask_bias<r + 16>(hid, ninf);
    });
    asm volatile("s_nop 1" : "+v"(s0), "+v"(s1));   // VALU write -> MFMA C operand inside an asm statement: two wait states (guide 5.7 item 2)
    X.hide = hide_new;
  };

  // ---- the two halves of a slot -------------------------------------------------------------------
  f32x16 sA0, sA1, sB0, sB1;          // score tiles S^T of the two blocks (keys 0-31 / 32-63 of the tile)
  u32x4 pA[4], pB[4];                 // P as packed bf16: the B operand of the four 16-key steps
  u32x2 vlo[4][3], vhi[4][3];         // V^T fragments of the current V tile (both blocks use them)
  A64Tmp tA, tB;
  A64Blind uA, uB;
  A64First fx;                        // (a rank's first tile: one block at a time)
  f32x16 zt;                          // lab (ABL & 16384): a tuple of zeros as the C operand of the first score MFMAs
  if constexpr (ABL & 16384) {
#pragma unroll
    for (int r = 0; r < 16; ++r) zt[r] = 0.f;
    asm volatile("" : "+v"(zt));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pA[i] = u32x4{0u, 0u, 0u, 0u}; pB[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int d = 0; d < 3; ++d) { vlo[i][d] = u32x2{0u, 0u}; vhi[i][d] = u32x2{0u, 0u}; }    // slot E(0) multiplies them by P = 0
  }

  // K fragments of a tile: 12 x ds_read_b128 into a[208:255]
  auto k_frag = [&](auto KS, auto HALF, unsigned ke, unsigned ko) {
    constexpr int ks = decltype(KS)::value, half = decltype(HALF)::value;
    constexpr int off = (ks >> 1) * 64 + half * 32 * KROW;
    if constexpr (ABL & 4) return;
    if constexpr (ks & 1) k_frag_read<(half ? A64_KC : A64_KA) + 4 * ks, off>(ko);
    else k_frag_read<(half ? A64_KC : A64_KA) + 4 * ks, off>(ke);
  };
  auto v_frag = [&](auto KS4, auto DT, unsigned va) {
    constexpr int ks4 = decltype(KS4)::value, dt = decltype(DT)::value;
    constexpr int off = ks4 * 16 * VROW + dt * 64;
    if constexpr (ABL & 4) { vlo[ks4][dt] = u32x2{va, va}; vhi[ks4][dt] = u32x2{va, va}; return; }
    vlo[ks4][dt] = ds_read_tr<off>(va);
    vhi[ks4][dt] = ds_read_tr<off + 8 * VROW>(va);
  };
  auto wait_v_frags = [&]() {        // the 24 V^T reads are older than the 12 K reads issued behind them
    if constexpr (ABL & 4) return;
    asm volatile("s_waitcnt lgkmcnt(12)"
                 : "+v"(vlo[0][0]), "+v"(vhi[0][0]), "+v"(vlo[0][1]), "+v"(vhi[0][1]), "+v"(vlo[0][2]), "+v"(vhi[0][2]),
                   "+v"(vlo[1][0]), "+v"(vhi[1][0]), "+v"(vlo[1][1]), "+v"(vhi[1][1]), "+v"(vlo[1][2]), "+v"(vhi[1][2]),
                   "+v"(vlo[2][0]), "+v"(vhi[2][0]), "+v"(vlo[2][1]), "+v"(vhi[2][1]), "+v"(vlo[2][2]), "+v"(vhi[2][2]),
                   "+v"(vlo[3][0]), "+v"(vhi[3][0]), "+v"(vlo[3][1]), "+v"(vhi[3][1]), "+v"(vlo[3][2]), "+v"(vhi[3][2]));
  };
#define A64_PIN() __builtin_amdgcn_sched_barrier(0)

  // first half of a slot: [P V of block Y] beside softmax chunks 0-11 of block X.  PV: the 12 MFMAs are issued; RELOAD: behind
  // each MFMA the V^T fragment it was the last to read is fetched again from the V tile at va; DMA: K (E slots) or V (O slots)
  // pieces of the unit being prefetched go into gaps 0-2.
  // BIASY (product build's bias iterations): the bias tuple of the tile the SECOND half makes (y0, y1; hidden columns hidY) is written
  // here, two columns per gap - its registers are free while P V runs, and a bias written between the halves is ~100 vector issues the
  // other three waves wait for at the next barrier
  auto half1 = [&](auto SER, auto RELOAD, auto YB, auto BIASY, f32x16& y0, f32x16& y1, const int hidY, f32x16& x0, f32x16& x1, u32x4 (&px)[4], A64Blk& X, A64Tmp& tx, A64Blind& ux, u32x4 (&py)[4], unsigned va, auto&& dma) {
    constexpr int kind = decltype(SER)::value;          // softmax of block X in this slot: 0 blind chunks, 1 serial exact (behind the slot), 2 first-tile chunks
    constexpr bool serial = kind == 1, reload = decltype(RELOAD)::value;
    constexpr int yb = decltype(YB)::value;
    constexpr int OAY = A64_O + 48 * yb, OAX = A64_O + 48 * (1 - yb);
    constexpr int NG = THR != 0 ? 16 : 12;          // product build: every fourth gap follows a row-sum MFMA of the 16 keys just multiplied
    static_for<NG>([&](auto I) {
      constexpr int g = decltype(I)::value;
      constexpr int ks4 = THR != 0 ? g / 4 : g / 3, dt = THR != 0 ? g % 4 : g % 3;
      constexpr int i = ks4 * 3 + (dt < 3 ? dt : 2);          // checked softmax: chunk i sits behind P V MFMA i
      if constexpr (dt < 3) {
        const u32x4 vv = {vlo[ks4][dt][0], vlo[ks4][dt][1], vhi[ks4][dt][0], vhi[ks4][dt][1]};
        mfma_pv<OAY + 16 * dt>(vv, py[ks4]);
        if constexpr (reload) v_frag(std::integral_constant<int, ks4>{}, std::integral_constant<int, dt>{}, va);
      } else {
        mfma_ones<A64_L + 4 * yb>(sel, py[ks4]);
      }
      A64_PIN();
      // Slot O opens one MFMA behind the score MFMAs that wrote S_B (slot E ended with them), and chunk 0 of the product's softmax
      // reads S: an MFMA result is not readable by the VALU for ~20 wait states after the instruction issued, and hipcc pads no
      // hazard between an asm MFMA and anything (guide 5.7 item 2).  Counted: 6 (chunk 23) + 4 (tail) + 3 (waits) + 1 (this slot's
      // first MFMA) + 4 here; slot E has the tile barrier in front.  The exact variant reads S one gap later.
      if constexpr (THR != 0 && g == 0 && yb == 0) asm volatile("s_nop 3" : "+v"(x0), "+v"(x1));
      if constexpr (THR == 0) sm_chunk<THR, OAX, i, ABL>(x0, x1, px, X, tx, c, rc);
      else if constexpr (kind == 2) sm_first_chunk<g>(x0, x1, px, X, fx, c);
      else if constexpr (!serial) sm_blind_chunk<g, 1 - yb, ABL>(x0, x1, px, X, ux, c);      // (P V of block B runs in slot E: SLOT 0)
      if constexpr (decltype(BIASY)::value && g < 16) { float b0_, b1_; mask_bias2<g, g + 16>(hidY, (int)0xFF800000, b0_, b1_); y0[g] = b0_; y1[g] = b1_; }
      dma(I);                                  // the caller's lambda decides which gaps carry a piece
      A64_PIN();
    });
  };
  // second half: [K Q_Y^T of tile jy] beside chunks 12-23 of block X; RELOAD: behind its last reader each K fragment is fetched
  // again from the K tile at (ke, ko)
  auto half2 = [&](auto SER, auto RELOAD, auto YB, auto FULLT, f32x16& y0, f32x16& y1, f32x16& x0, f32x16& x1, u32x4 (&px)[4], A64Blk& X, A64Tmp& tx, A64Blind& ux, unsigned ke, unsigned ko) {
    constexpr int kind = decltype(SER)::value;
    constexpr bool serial = kind == 1, reload = decltype(RELOAD)::value;
    constexpr int QAY = A64_Q + 24 * decltype(YB)::value, OAX = A64_O + 48 * (1 - decltype(YB)::value);
    static_for<12>([&](auto I) {
      constexpr int i = decltype(I)::value, ks = i >> 1, half = i & 1;
      if constexpr (ks == 0) {
        if constexpr (decltype(FULLT)::value && (ABL & 16384)) { if constexpr (half == 0) mfma_qk_c<A64_KA, QAY>(y0, zt); else mfma_qk_c<A64_KC, QAY>(y1, zt); }     // lab: C from a VGPR tuple of zeros
        else if constexpr (decltype(FULLT)::value) { if constexpr (half == 0) mfma_qk_zero<A64_KA, QAY>(y0); else mfma_qk_zero<A64_KC, QAY>(y1); }
        else { if constexpr (half == 0) mfma_qk<A64_KA, QAY>(y0); else mfma_qk<A64_KC, QAY>(y1); }
      } else {
        if constexpr (half == 0) mfma_qk<A64_KA + 4 * ks, QAY + 4 * ks>(y0); else mfma_qk<A64_KC + 4 * ks, QAY + 4 * ks>(y1);
      }
      if constexpr (reload) k_frag(std::integral_constant<int, ks>{}, std::integral_constant<int, half>{}, ke, ko);
      A64_PIN();
      if constexpr (THR == 0) sm_chunk<THR, OAX, 12 + i, ABL>(x0, x1, px, X, tx, c, rc);
      else if constexpr (kind == 2) sm_first_chunk<16 + i>(x0, x1, px, X, fx, c);
      else if constexpr (!serial) sm_blind_chunk<16 + i, 1 - decltype(YB)::value, ABL>(x0, x1, px, X, ux, c);
      if constexpr (ABL & 8192) asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1" : "+v"(tx.m3), "+v"(tx.thr));
      A64_PIN();
    });
    // (bias iterations once ended in 20 wait states tied to both tiles: hipcc copied score tiles at their joins and takes an asm MFMA's
    // result as ready.  With one definition of every tile per iteration there are no such copies; tools/attn64_hazards.py holds the line.)
    if constexpr (serial) {
      // the exact iteration of the product build - while a row of the wave has no reference maximum yet (the rank's first tile, as a
      // rule) and on the second walk of a rank that failed its verification: the tile's softmax in one piece behind the slot's MFMAs
      // (blind chunks AND this tail in one iteration would keep S and P alive together: ~60 registers over the file)
      sm_redo<THR, OAX>(x0, x1, px, X, tx, c);
      if constexpr (ABL & 512) X.l_dbg += 1;
      A64_PIN();
    }
  };
  auto no_dma = [](auto) {};
  using T_ = std::true_type;
  using F_ = std::false_type;
  using M0 = std::integral_constant<int, 0>;
  using M1 = std::integral_constant<int, 1>;
  using M2 = std::integral_constant<int, 2>;
  using M4 = std::integral_constant<int, 4>;
  using M7 = std::integral_constant<int, 7>;
  using BA = std::integral_constant<int, 0>;
  using BB = std::integral_constant<int, 1>;

  // ---- K(0) has landed (this wave's pieces: all but the 12 youngest), for everybody: fragments, then K Q_A^T of tile 0 ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // Q, K(0) and units 0, 1 (asked for a whole epilogue ago, first rank excepted)
  if (A.row_uniform) static_for<24>([&](auto R) { acc_zero<A64_Q + decltype(R)::value>(); });         // uniform-softmax rows: score 0 on every
  if (B.row_uniform) static_for<24>([&](auto R) { acc_zero<A64_Q + 24 + decltype(R)::value>(); });    // column = an all-zero query (per lane)
  __builtin_amdgcn_s_barrier();
  static_for<12>([&](auto I) { k_frag(std::integral_constant<int, decltype(I)::value / 2>{}, std::integral_constant<int, decltype(I)::value % 2>{}, sK_a + k_even, sK_a + k_odd); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  A64_PIN();
  {
    const bool fa = (A.fast & 1ull) != 0ull;       // wave-uniform
    if (fa) fast_hide(A, 0); else tile_bias(A, 0, sA0, sA1);
    A64_PIN();
    static_for<12>([&](auto I) {
      constexpr int i = decltype(I)::value, ks = i >> 1, half = i & 1;
      if constexpr (ks == 0) {
        // the two forms of a chain's first MFMA meet behind this if / else: hipcc copies the tile on both sides of the join and takes an
        // asm MFMA's result as ready (tools/attn64_hazards.py found a v_mov one wait state behind the MFMA in one build).  Once per rank.
        if (fa) { if constexpr (half == 0) mfma_qk_zero_settled<A64_KA, A64_Q>(sA0); else mfma_qk_zero_settled<A64_KC, A64_Q>(sA1); }
        else { if constexpr (half == 0) mfma_qk_settled<A64_KA, A64_Q>(sA0); else mfma_qk_settled<A64_KC, A64_Q>(sA1); }
      } else {
        if constexpr (half == 0) mfma_qk<A64_KA + 4 * ks, A64_Q + 4 * ks>(sA0); else mfma_qk<A64_KC + 4 * ks, A64_Q + 4 * ks>(sA1);
      }
    });
    // hipcc takes an asm statement's outputs as ready when it ends: here it once spilled sA0[0] three instructions behind the last
    // MFMA (v_accvgpr_write of a register the MFMA had not written yet) and restored the stale score in the tile loop.  Once per
    // rank, 20 wait states tied to both tiles; inside the tile loop tools/attn64_hazards.py checks every build for such reads.
    mfma_results_settle(sA0, sA1);
    A64_PIN();
  }

  if constexpr (ABL & 512) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_pro += t_ - st_t; st_t = t_; st_tiles += jend; }
  int st0 = 0, st1 = 1, st2 = 2;     // ring stages of tiles j, j+1, j+2 (K and V rings alike)
  // Taken branches are not free for a wave alone on its SIMD (the instruction buffer refills): the tile loop has none but its own
  // back edge on the common path.  Slot E(0) runs its P_B V MFMAs on P = 0 and V^T = 0 (zeroed above), slot O of the last tile
  // computes a score tile nobody reads, and the tiles a wave only has to keep the stream going for are a second loop.
  // one tile = slots E(j) and O(j).  FT: the score tiles produced in this iteration - block B's tile j and block A's tile j + 1 -
  // are both fast by the masks (FULL / ROWWISE / HIDDEN): no bias, the first MFMA of each chain takes the constant 0; otherwise both
  // go through the bias path, which is right for every class
  // MODE 0: bias path; 1: fast (per-lane hide, decided per tile); product build only: 2 - fast, inside a run that leaves both
  // blocks' hide as it is (nothing per tile), 4 - bias path with the serial exact softmax
  auto iter = [&](auto MODE, int j) {
    constexpr int mode = decltype(MODE)::value;
    constexpr bool ft = mode == 1 || mode == 2;
    using FT = std::integral_constant<bool, ft>;
    using BL = std::integral_constant<int, mode == 4 ? 1 : mode == 7 ? 2 : 0>;
    // unit j = {K(j+1), V(j)}: this wave's pieces are all but the 6 youngest (unit j+1); then a workgroup-wide fact
    unsigned long long tw0 = 0, tit0 = 0;
    if constexpr (ABL & 512) tw0 = __builtin_amdgcn_s_memtime();
    if constexpr (ABL & 1024) tit0 = __builtin_amdgcn_s_memtime();
    if constexpr (!(ABL & 2)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (ABL & 512) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_dma += t_ - tw0; tw0 = t_; }
    if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    if constexpr (ABL & 512) st_wait += __builtin_amdgcn_s_memtime() - tw0;
    A64_PIN();
    // unit j+2 = {K(j+3) -> the stage K(j) has left, V(j+2) -> the stage V(j-1) has left}
    const int tk = j + 3, tv = j + 2;
    // the last two iterations ask for units past the rank's end (branch-free loop body; rows past L cost no traffic): they are waited
    // for behind the loops.  Product build: all six pieces ride in slot O - slot E's gaps already carry the 36 fragment reloads
    // (stamps: a slot E gap is issue-bound, a slot O gap MFMA-bound with room).
    auto dma_e = [&](auto I) { if constexpr (THR == 0 && decltype(I)::value < 3) dma_k(I, tk, st0); };
    auto dma_o = [&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (THR == 0) { if constexpr (i < 3) dma_v(I, tv, st2); }
      else if constexpr (i < 3) dma_k(I, tk, st0);
      else if constexpr (i < 6) dma_v(std::integral_constant<int, i - 3>{}, tv, st2);
    };
    const unsigned va = sV_a + st0 * VTILE + voff;                 // V(j)
    const unsigned ke = sK_a + st1 * KTILE + k_even, ko = sK_a + st1 * KTILE + k_odd;   // K(j+1)
    // ---- slot E(j): P_B V (j-1), K Q_B^T (j)  beside  softmax of S_A(j) ----
    unsigned long long th_ = 0;
    if constexpr ((ABL & 1024) && ft) th_ = __builtin_amdgcn_s_memtime();
    constexpr bool gap_bias = THR != 0 && (mode == 0 || mode == 4 || mode == 7);
    using GB = std::integral_constant<bool, gap_bias>;
    int hidY = 0;
    float hideY = 0.f;
    if constexpr (gap_bias) bias_hid(B, j, hidY, hideY);
    half1(BL{}, T_{}, BB{}, GB{}, sB0, sB1, hidY, sA0, sA1, pA, A, tA, uA, pB, va, dma_e);
    if constexpr ((ABL & 1024) && ft) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_h[0] += t_ - th_; th_ = t_; }
    if constexpr (gap_bias) { B.hide = hideY; asm volatile("s_nop 1" : "+v"(sB0), "+v"(sB1)); }
    else if constexpr (mode == 1) fast_hide(B, j); else if constexpr (mode != 2) tile_bias(B, j, sB0, sB1);
    A64_PIN();
    half2(BL{}, T_{}, BB{}, FT{}, sB0, sB1, sA0, sA1, pA, A, tA, uA, ke, ko);
    if constexpr ((ABL & 1024) && ft) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_h[1] += t_ - th_; }
    // ---- slot O(j): P_A V (j), K Q_A^T (j+1)  beside  softmax of S_B(j) ----
    wait_v_frags();
    asm volatile("s_nop 1" : "+v"(pA[0]), "+v"(pA[1]), "+v"(pA[2]), "+v"(pA[3]));   // packed by the VALU just above -> MFMA operand
    A64_PIN();
    if constexpr ((ABL & 1024) && ft) th_ = __builtin_amdgcn_s_memtime();
    if constexpr (gap_bias) bias_hid(A, j + 1, hidY, hideY);
    half1(BL{}, F_{}, BA{}, GB{}, sA0, sA1, hidY, sB0, sB1, pB, B, tB, uB, pA, va, dma_o);
    if constexpr ((ABL & 1024) && ft) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_h[2] += t_ - th_; th_ = t_; }
    if constexpr (!(ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // K(j+1) fragments
    A64_PIN();
    if constexpr (gap_bias) { A.hide = hideY; asm volatile("s_nop 1" : "+v"(sA0), "+v"(sA1)); }
    else if constexpr (mode == 1) fast_hide(A, j + 1); else if constexpr (mode != 2) tile_bias(A, j + 1, sA0, sA1);
    A64_PIN();
    half2(BL{}, F_{}, BA{}, FT{}, sA0, sA1, sB0, sB1, pB, B, tB, uB, ke, ko);
    if constexpr ((ABL & 1024) && ft) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_h[3] += t_ - th_; st_n += 1; }
    asm volatile("s_nop 1" : "+v"(pB[0]), "+v"(pB[1]), "+v"(pB[2]), "+v"(pB[3]));
    A64_PIN();
    const int t_ = st0; st0 = st1; st1 = st2; st2 = t_;
    if constexpr ((ABL & 1024) && !ft) { st_slow += __builtin_amdgcn_s_memtime() - tit0; st_nslow += 1; }
    if constexpr ((ABL & 1024) && ft) st_full += __builtin_amdgcn_s_memtime() - tit0;
  };
  auto both_fast = [&](int j) -> bool {
    const int t = j & 63;
    return t != 63 && ((B.fast >> t) & (A.fast >> (t + 1)) & 1ull) != 0ull;      // tile j + 1 of the next window: through the bias path
  };
  // runs of fast tiles are an inner loop of their own: an if / else per tile would join sixteen-register score tiles defined
  // by asm statements on two paths, and hipcc copies them through spill slots at every join
  // Product build: the fast iteration is also the BLIND one (sm_blind_chunk), entered only once every row of both blocks has a
  // reference maximum - i.e. behind the rank's first tile(s), which go through the bias path and its checked softmax - and never on
  // the second walk of a rank that failed its verification.
  bool settled = (THR == 0 || (ABL & 1)) && !(ABL & 64);
  int j = 0;
  while (j < jend_w) {
    // runs are counted once, up front: with one wave per SIMD every scalar instruction of the loop control is an issue slot of the
    // tile, and a taken branch an instruction-buffer refill
    const int t = j & 63;
    // Product build: a run = tiles that are fast for both blocks AND leave each block's per-lane hide as it is (FULL runs: nobody
    // hidden; the ROWWISE tiles of an image's edge rows between two rectangle boundaries; HIDDEN tiles): the loop body then has no
    // per-tile mask work at all.  Tiles where the hide changes go through the single iterations below.
    int nrun = 0;
    if constexpr (THR != 0) {
      if (settled) {
        const unsigned long long mR = (B.fast & (A.fast >> 1)) >> t;
        nrun = min(min(mR == ~0ull ? 64 : (int)__builtin_ctzll(~mR), 63 - t), jend_w - j);
        if (nrun > 0) {
          nrun = min(nrun, min(const_len(B, j), const_len(A, j + 1)));
          const float hA1 = hide_val(A, j + 1);
          if (__ballot(hA1 != A.hide) != 0ull) nrun = 0;          // block A's tile j, already in flight, was made under another hide
          if (nrun > 0) B.hide = hide_val(B, j);
        }
      }
    }
    if (__builtin_expect(nrun > 0, 1)) {
      // two tiles per turn: the loop control and its taken branch are issue slots of the tile (28 -> 19.5 scalar issues per tile, -1.3 % kernel
      // time; four per turn: no further gain; three per turn with the ring stages as compile-time constants: 12 scalar issues, but the
      // single iterations that align a run to it cost more than it saves - EXPERIMENTS.md)
      if (nrun & 1) { iter(M2{}, j); ++j; --nrun; }
      while (nrun > 0) { iter(M2{}, j); iter(M2{}, j + 1); j += 2; nrun -= 2; }
    } else if (THR == 0 && both_fast(j)) {        // the exact build's fast (per-lane hide) iteration, in runs
      do { iter(M1{}, j); ++j; } while (j < jend_w && both_fast(j));
    } else if (THR == 0 || settled) {      // (product build: a fast tile where the hide changes comes here too - tile_bias has a short form for it)
      iter(M0{}, j); ++j;
    } else {
      // no reference maximum yet: a rank's first tile has its exact softmax in the gaps (nothing accumulated, nothing to rescale);
      // later tiles of rows that have seen nothing so far, and every tile of a rank walked again, take the serial exact iteration
      if constexpr (THR != 0) { if (j == 0 && !force_checked && !(ABL & 64)) iter(M7{}, j); else iter(M4{}, j); }
      ++j;
      if constexpr (THR != 0) {
        // (rows beyond seq_len under the zero convention see nothing, ever: they must not hold the wave on the exact path)
        if (!settled && !force_checked && !(ABL & 64))
          settled = __ballot((A.m_ref < -1e29f && (A.row_alive || A.row_uniform)) || (B.m_ref < -1e29f && (B.row_alive || B.row_uniform))) == 0ull;
      }
    }
    if ((j & 63) == 0) { block_masks(A, j); block_masks(B, j); }     // next window of 64 tiles (L > 4096)
  }
  for (; j < jend; ++j) {             // this wave's blocks are done: keep the stream and the barriers going
    if constexpr (!(ABL & 2)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    const int tk = j + 3, tv = j + 2;
    static_for<3>([&](auto I) { dma_k(I, tk, st0); });
    static_for<3>([&](auto I) { dma_v(I, tv, st2); });
    const int t_ = st0; st0 = st1; st1 = st2; st2 = t_;
  }
  // P_B V of the wave's last tile: its V^T fragments are still in registers
  if (jend_w > 0) {
    static_for<12>([&](auto I) {
      constexpr int i = decltype(I)::value, ks4 = i / 3, dt = i % 3;
      const u32x4 vv = {vlo[ks4][dt][0], vlo[ks4][dt][1], vhi[ks4][dt][0], vhi[ks4][dt][1]};
      mfma_pv<A64_O + 48 + 16 * dt>(vv, pB[ks4]);
      if constexpr (THR != 0 && dt == 2) mfma_ones<A64_L + 4>(sel, pB[ks4]);
    });
  }
  // The two units asked for past the rank's end have landed (they were issued one and two tiles ago) and every wave has read the last
  // tiles out of the ring: the next rank's first tiles may stream in, its Q rows be asked for.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bool walk_again = false;
  if constexpr (THR != 0) {
    // Verification of the blind softmax: every row sum of the rank must be below 2^64 (a score 64 log2 units above its row's
    // reference maximum - unheard of behind a first tile that holds the row's own first keys - makes it larger; inf and NaN fail
    // the comparison too).  One LDS word per rank parity collects the four waves' verdicts across the barrier that ends the rank
    // anyway; a failed rank is walked again with the blind iteration switched off (the checked path raises the reference as it goes).
    asm volatile("s_nop 15" ::: "memory");           // the last row-sum MFMAs
    const float lA = acc_get<A64_L>(), lB = acc_get<A64_L + 4>();
    const bool bad = !force_checked && __ballot(!(lA < 0x1p64f) || !(lB < 0x1p64f)) != 0ull;
    const unsigned fa_ = sFlag_a + 4u * (unsigned)vpar, one = 1u, zero = 0u;
    if (bad) asm volatile("ds_write_b32 %0, %1" ::"v"(fa_), "v"(one) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned fl;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fl) : "v"(fa_) : "memory");
    walk_again = __builtin_amdgcn_readfirstlane(fl) != 0u;
    // the other parity's word is cleared for the rank after this one (its writers are a whole tile loop of barriers away)
    const unsigned fo_ = sFlag_a + 4u * (unsigned)(vpar ^ 1);
    if (wave == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(fo_), "v"(zero) : "memory");
    vpar ^= 1;
  } else {
    __builtin_amdgcn_s_barrier();
  }
  if (walk_again) {
    force_checked = true;
    prefetch_rank(g);
    --kk;
    continue;
  }
  force_checked = false;
  if (rank_of(kk + 1) < p.nqt) prefetch_rank(rank_of(kk + 1));
  if constexpr (ABL & 512) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_loop += t_ - st_t; st_t = t_; st_redo += A.l_dbg + B.l_dbg; }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMA results
  A64_PIN();

  // ---- epilogue per block: O = O^T / l through the wave's own piece of the output staging area, whole 192-B rows out -------------
  int lane_o = lane;                     // opaque per rank: the epilogue's per-lane addresses are not hoisted over the tile loop (they would be spilled)
  asm volatile("" : "+v"(lane_o));
  const int l31_o = lane_o & 31, h_o = lane_o >> 5;
  const unsigned sO_a = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)(smem + LDS_RING + LDS_VB));
  auto store_block = [&](auto XB, A64Blk& X) {
    constexpr int xb = decltype(XB)::value;
    float l_tot;
    if constexpr (THR != 0) l_tot = acc_get<A64_L + 4 * xb>();       // the matrix pipe's row sum is whole (both half rows)
    else l_tot = halves_sum(X.l);
    const bool dead = !(l_tot > 0.f);
    // wave-private staging through asm LDS accesses: an LDS access hipcc can see gets an s_waitcnt vmcnt(0) in front (it may alias an LDS-DMA
    // in flight as far as the compiler knows) - here that would be a wait for the next rank's prefetch
    const unsigned sO = sO_a + (unsigned)((wave * 2 + xb) * (32 * OROW));
    const float inv = dead ? 0.f : 1.0f / l_tot;
    static_for<12>([&](auto I) {
      constexpr int i = decltype(I)::value, dt = i / 4, q4 = i % 4;
      float v[4];
      v[0] = acc_get<A64_O + 48 * xb + 16 * dt + 4 * q4 + 0>() * inv;
      v[1] = acc_get<A64_O + 48 * xb + 16 * dt + 4 * q4 + 1>() * inv;
      v[2] = acc_get<A64_O + 48 * xb + 16 * dt + 4 * q4 + 2>() * inv;
      v[3] = acc_get<A64_O + 48 * xb + 16 * dt + 4 * q4 + 3>() * inv;
      if (dead && p.dead_uniform && X.row < L) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const char* vp = vb_ + (dt * 32 + q4 * 8 + 4 * h_o) * 2;
        for (int t2 = 0; t2 < L; ++t2) {
          const u32x2 w2 = *(const u32x2*)(vp + (size_t)t2 * 192);
          a0 += bf16_lo(w2[0]); a1 += bf16_hi(w2[0]); a2 += bf16_lo(w2[1]); a3 += bf16_hi(w2[1]);
        }
        const float il = 1.0f / (float)L;
        v[0] = a0 * il; v[1] = a1 * il; v[2] = a2 * il; v[3] = a3 * il;
      }
      const u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      const unsigned wa = sO + (unsigned)(l31_o * OROW + (dt * 32 + q4 * 8 + 4 * h_o) * 2);
      asm volatile("ds_write_b64 %0, %1" ::"v"(wa), "v"(pk) : "memory");
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same-wave round trip through LDS: the writes are in before the reads go out
    bf16_t* const obase = p.o + ((size_t)b * L * p.H + head) * 96;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int ch = it * 64 + lane_o;                 // 16-B chunk of the block's tile: row ch/12, chunk ch%12
      const int r = ch / 12, cc = ch - r * 12;
      u32x4 w4;
      const unsigned ra_ = sO + (unsigned)(r * OROW + cc * 16);
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w4) : "v"(ra_) : "memory");
      if (X.wq0 + r < L) *(u32x4*)((char*)(obase + (size_t)(X.wq0 + r) * p.H * 96) + cc * 16) = w4;
    }
    if (!(ABL & 512) && p.lse && h_o == 0 && X.row < L) p.lse[(size_t)bh * L + X.row] = dead ? -INFINITY : (X.m_ref + __builtin_amdgcn_logf(l_tot)) * 0.69314718055994530942f;
  };
  store_block(BA{}, A);
  store_block(BB{}, B);
  if constexpr (ABL & 512) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_epi += t_ - st_t; }
  }                  // next rank of this workgroup
  if constexpr (ABL & 512) {
    if (p.lse && tid == 0) {
      float* d = p.lse + 8 * blockIdx.x;
      d[0] = (float)st_pro; d[1] = (float)st_loop; d[2] = (float)st_epi; d[3] = (float)st_tiles;
      d[4] = (float)(__builtin_amdgcn_s_memrealtime() - st_rt0);
      d[5] = (float)st_redo;
      d[6] = (float)st_dma;
      d[7] = (float)st_wait;
      if constexpr (ABL & 1024) { d[0] = (float)st_h[0]; d[1] = (float)st_h[1]; d[2] = (float)st_h[2]; d[5] = (float)st_h[3]; d[3] = (float)st_n; d[6] = (float)st_slow; d[7] = (float)st_nslow; d[4] = (float)st_full; }
    }
  }
#undef A64_PIN
}

#ifdef AKI_LAB_HOOKS
extern int g_attn_variant;
#endif

// host side: called by attn_core_bf16 (mma_attn_bf16.hip) with the common parameters filled in
int attn_core64_bf16_launch(AttnParams p, int cus, hipStream_t stream, int exact_max) {
  const int nbh = p.B * p.H;
  p.nqt = (p.L + 255) / 256;                       // ranks of eight 32-row blocks
  int splits = (cus + nbh - 1) / nbh;              // one workgroup per CU
  p.splits = splits < 1 ? 1 : (splits > p.nqt ? p.nqt : splits);
  int grp = ((cus + p.splits - 1) / p.splits + 7) & ~7;     // one round of resident slots per group
  p.group_bh = grp > nbh ? nbh : grp;
#ifdef AKI_LAB_HOOKS
#define A64_ABL_CASE(m) if (g_attn_variant == 100 + (m)) { hipLaunchKernelGGL((mma_attn64_bf16_kernel<8, (m)>), dim3(nbh * p.splits), dim3(256), 0, stream, p); return AKI_OK; }
  A64_ABL_CASE(64) A64_ABL_CASE(512) A64_ABL_CASE(513) A64_ABL_CASE(514) A64_ABL_CASE(515) A64_ABL_CASE(520) A64_ABL_CASE(521) A64_ABL_CASE(523) A64_ABL_CASE(1536) A64_ABL_CASE(1537)
#undef A64_ABL_CASE
#endif
  if (exact_max) hipLaunchKernelGGL((mma_attn64_bf16_kernel<0, 0>), dim3(nbh * p.splits), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((mma_attn64_bf16_kernel<8, 0>), dim3(nbh * p.splits), dim3(256), 0, stream, p);
  return AKI_OK;
}

}  // namespace aki
