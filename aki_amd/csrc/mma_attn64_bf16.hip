// mma_attn64_bf16.hip - span-driven modality-mutual attention core, second structure: 64 query rows per wave, ONE wave per
// SIMD (a 512-register kernel), one workgroup of four waves per CU.  Same contract and arithmetic as mma_attn_bf16.hip
// (HF:phi3/modeling_phi3.py:145-167 under the reference's mask, src/vlm.py:410-443; no L x L tensor), different machine
// mapping:
//
//  * A wave owns TWO 32-row blocks (A, B) of one (batch, head) and walks the K/V tile stream once for both: the K
//    fragments (12 ds_read_b128) and the transposed V fragments (24 ds_read_b64_tr_b16) of a tile are read once and used
//    by 2 x 26 MFMAs - half the LDS traffic and half the barriers per unit of work of the 32-row kernel - and the two
//    blocks are independent dependency chains inside one instruction stream: the softmax VALU of one block sits between
//    the MFMAs of the other (mma_attn_bf16.hip relies on a second wave per SIMD for that and measured ~50 % issue idle:
//    all waves of a workgroup meet at a barrier every tile).
//  * The subtraction of the running maximum and most of the mask ride on the matrix core.  head_dim 96 is six k16 steps;
//    a seventh, "augmented" k-step multiplies four extra columns of K by four extra columns of Q:
//        K_aug[key] = ( 1, 1, colbias[key], limbias[key] )      Q_aug[row] = ( -m_ref[row], rowbias[row], colsel[row], 1 )
//    so the score accumulator comes out as  s - m_ref + rowbias + colsel*colbias + limbias  with bias values 0 / -1e30:
//      - m_ref is the row's reference offset of the online softmax (a bf16-exact value within 2^8 of the running maximum,
//        re-based only when the maximum runs away from it), so p = exp2(S) needs NO per-element VALU subtract or scale
//        (Q is pre-multiplied by scale*log2(e) when it is loaded): 32 v_exp + 32 v_add + 16 v_cvt_pk per block and tile;
//      - every tile whose mask is SEPARABLE - entirely below the diagonal (valid columns only), or right of it under one
//        rectangle (row in rectangle rows AND column in rectangle columns AND valid) - needs no per-element mask either;
//        the per-lane visibility word of the 32-row kernel is only built for the one tile per block that the diagonal cuts
//        (and for blocks that straddle two rectangles).
//    Two extra MFMAs per block and tile (+8 % matrix work) replace ~190 VALU issue cycles of the ~750 per block and tile
//    that made the 32-row kernel VALU-issue bound.
//  * The K/V ring is NS stages deep (the workgroup has the CU's LDS to itself): the barrier at the top of tile j publishes
//    tile j+1, whose K fragments are read under tile j's last MFMAs.
//
// Work decomposition: the 32-row blocks of a pair are ranked by the number of columns they walk (as in the 32-row kernel);
// a rank is 8 consecutive blocks of that order (2 per wave, neighbours in the order - similar extents share a wave), and
// workgroup `sidx` of the pair's `splits` workgroups walks the ranks in a snake.  At the benchmark shape (B*H = 256 pairs on
// 256 CUs) every workgroup owns one pair.
#include "attn_mma_common.h"

#ifndef AKI_ATTN64_STAGES
#define AKI_ATTN64_STAGES 4
#endif

namespace aki {

namespace {

constexpr float NEGBIG = -1e30f;       // finite on purpose: 0 * NEGBIG must be 0 inside the augmented k-step
constexpr float REBASE_THR = 8.0f;     // |running max - m_ref| beyond which the reference is re-based (p <= 2^8)

struct Blk {            // one 32-row block of the wave (all members are register arrays after unrolling)
  bf16x8 qf[6];         // Q fragments, pre-scaled by scale*log2(e)
  f32x16 o[3];          // O^T accumulators
  float m_ref, m_true, l_part;
  int wq0;              // first row (wave-uniform); >= L: the slot is empty
  int row;              // this lane's row
  int rc0, rc1;         // this lane's rectangle columns (empty when the row is in no rectangle)
  int touch_lo, touch_hi, full_lo, full_hi;   // wave-level rectangle summary (as in the 32-row kernel)
  int nrect;            // rectangles touching the block's rows
  int sr_lo, sr_hi;     // columns of the one rectangle touching the block (nrect == 1)
  bool rowin;           // lane's row inside that rectangle
  bool exists, wave_alive, wave_has_dead, row_alive, has_uniform, row_uniform;
};

}  // namespace

template <int NS>
__global__ __launch_bounds__(256, 1) void mma_attn64_bf16_kernel(const AttnParams p) {
  constexpr int NW = 4, NT = 256, NCH = 3;          // 64 x 12 sixteen-byte chunks per K (and V) tile / 256 threads
  constexpr int BQ = 256;                            // rows per rank
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sK = smem;
  char* const sV = smem + NS * KTILE;
  unsigned long long* const sVB = (unsigned long long*)(smem + NS * (KTILE + VTILE));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  // workgroup -> (pair, split): as in the 32-row kernel (groups of pairs, split-major inside a group)
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

  int tid_o = tid;   // made opaque once per rank (keeps the per-chunk address arithmetic inside the rank loop)
  auto issue_tile = [&](int j, int stage) {
    const int c0 = j * 64;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = i * NT + tid_o;
      const int kr = ch / 12, pos = ch - kr * 12;
      const size_t rowoff = (size_t)min(c0 + kr, L - 1) * 192;
      const int srcchunk = pos ^ ((kr >> 2) & 3);
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(kb + rowoff + srcchunk * 16), AKI_LDS_PTR(sK + stage * KTILE + (i * NT + wave * 64) * 16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(vb_ + rowoff + pos * 16), AKI_LDS_PTR(sV + stage * VTILE + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
  };

  // ---- once per workgroup: valid words, rectangles, block extents and their ranking (see mma_attn_bf16.hip) -----------
  constexpr int SCHED_MAX = 64;
  const int nblk = (L + 31) >> 5;
  const bool sched = nblk <= SCHED_MAX;
  const int Lb = p.seq_lens ? min(p.seq_lens[b], L) : L;
  for (int w = tid; w < p.nwords; w += NT) {
    unsigned long long vbw;
    if (p.vbits) vbw = p.vbits[(size_t)b * p.nwords + w];
    else vbw = (w * 64 + 64 <= L) ? ~0ull : ((1ull << (L - w * 64)) - 1ull);
    sVB[w] = vbw;
  }
  const aki_mma_rect* const rects_b = p.rects + (size_t)b * p.max_rects;
  auto rect_at = [&](int i) -> aki_mma_rect {
    const unsigned long long pa = (unsigned long long)(rects_b + i);
    const unsigned long long pu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pa);
    u32x4 r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(pu) : "memory");
    aki_mma_rect o;
    o.row_lo = (int)r[0]; o.row_hi = (int)r[1]; o.col_lo = (int)r[2]; o.col_hi = (int)r[3];
    return o;
  };
  auto block_extent = [&](int r0) -> int {
    if (r0 >= L) return -1;
    int ext = min(r0 + 32, L);
    for (int i = 0; i < p.max_rects; ++i) {
      const aki_mma_rect r = rect_at(i);
      if (r.row_hi > r.row_lo && r.col_hi > r.col_lo && r.row_lo < r0 + 32 && r.row_hi > r0) ext = max(ext, min(r.col_hi, L));
    }
    if (p.dead_uniform && min(r0 + 32, L) > Lb) ext = L;
    return ext;
  };
  int ext_s = -1, rank_s = 0x7fff;
  if (sched) {
    ext_s = block_extent(32 * lane);
    const int key = ext_s < 0 ? -1 : ext_s * 64 + lane;
    int rank = 0;
    for (int i = 0; i < nblk; i += 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) rank += __builtin_amdgcn_readlane(key, i + k) > key ? 1 : 0;
    }
    rank_s = lane < nblk ? rank : 0x7fff;
  }

  // per-lane LDS offsets
  const int kswz = (l31 >> 2) & 3;
  const int krow = l31 * KROW;
  const int voff = (4 * h + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const unsigned bf_one = 0x3F80u;                                     // bf16 1.0
  const unsigned bf_neg = __builtin_bit_cast(unsigned, NEGBIG) >> 16;  // bf16 of -1e30 (truncated: any value ~ -1e30 will do)
  const float scale = p.scale_log2;

  Blk blk[2];

  for (int kk = 0;; ++kk) {
  const int g = kk * p.splits + ((kk & 1) ? p.splits - 1 - sidx : sidx);
  if (g >= p.nqt) break;
  int lane_o = lane;
  asm volatile("" : "+v"(tid_o), "+v"(lane_o));

  // ---- rank prologue: tiles 0 .. NS-2 in flight, block assignment, Q ------------------------------------------------
  int hi_col = 0, wqs[2] = {L, L};
  if (sched) {
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
      const unsigned long long m = __ballot(rank_s == 8 * g + s_);
      if (m != 0ull) {
        const int l = __builtin_ctzll(m);
        hi_col = max(hi_col, __builtin_amdgcn_readlane(ext_s, l));
        if ((s_ >> 1) == wave) wqs[s_ & 1] = 32 * l;
      }
    }
  } else {
    const int q0 = (p.nqt - 1 - g) * BQ;
    const int ext = block_extent(q0 + 32 * (lane & 7));
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) hi_col = max(hi_col, __builtin_amdgcn_readlane(ext, s_));
    wqs[0] = q0 + wave * 64;
    wqs[1] = q0 + wave * 64 + 32;
  }
  hi_col = __builtin_amdgcn_readfirstlane(hi_col);
  const int jend = (hi_col + 63) >> 6;
#pragma unroll
  for (int s_ = 0; s_ < NS - 1; ++s_)
    if (s_ < jend) issue_tile(s_, s_);

#pragma unroll
  for (int x = 0; x < 2; ++x) {
    Blk& B_ = blk[x];
    const int wq0 = __builtin_amdgcn_readfirstlane(min(wqs[x], L));
    B_.wq0 = wq0;
    B_.exists = wq0 < L;
    B_.row = wq0 + l31;
    const bf16_t* qrow = qb + (size_t)min(B_.row, L - 1) * 96 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const bf16x8 raw = *(const bf16x8*)(qrow + 16 * ks);
      bf16x8 sc;
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[e] = (__bf16)((float)raw[e] * scale);
      B_.qf[ks] = sc;
    }
    int touch_lo = 0x7fffffff, touch_hi = 0, full_lo = 0, full_hi = 0, rc0 = 0, rc1 = 0, nrect = 0, sr_lo = 0, sr_hi = 0;
    bool rowin = false;
    for (int i = 0; i < p.max_rects; ++i) {
      const aki_mma_rect r = rect_at(i);
      if (r.row_hi > r.row_lo && r.col_hi > r.col_lo) {
        if (r.row_lo < wq0 + 32 && r.row_hi > wq0) {
          touch_lo = min(touch_lo, r.col_lo);
          touch_hi = max(touch_hi, r.col_hi);
          if (r.row_lo <= wq0 && r.row_hi >= wq0 + 32) { full_lo = r.col_lo; full_hi = r.col_hi; }
          ++nrect;
          sr_lo = r.col_lo; sr_hi = r.col_hi;
        }
        if (B_.row >= r.row_lo && B_.row < r.row_hi) { rc0 = r.col_lo; rc1 = r.col_hi; rowin = true; }
      }
    }
    B_.touch_lo = touch_lo; B_.touch_hi = touch_hi; B_.full_lo = full_lo; B_.full_hi = full_hi;
    B_.rc0 = rc0; B_.rc1 = rc1; B_.nrect = nrect; B_.sr_lo = sr_lo; B_.sr_hi = sr_hi; B_.rowin = rowin;
    B_.wave_alive = wq0 < Lb;
    B_.wave_has_dead = min(wq0 + 32, L) > Lb;
    B_.row_alive = B_.row < Lb;
    B_.has_uniform = p.dead_uniform && B_.wave_has_dead && B_.exists;
    B_.row_uniform = p.dead_uniform && !B_.row_alive;
    if (B_.row_uniform) {
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) B_.qf[ks] = bf16x8{};
    }
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) B_.o[dt][r] = 0.f;
    B_.m_ref = 0.f; B_.m_true = -3e38f; B_.l_part = 0.f;
  }

  // everything issued above is retired with a wait hipcc can see (see mma_attn_bf16.hip): Q is in registers, and the ring
  // restarts from "tiles 0 .. NS-2 landed"
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __builtin_amdgcn_s_barrier();

  auto valid_word = [&](int w) -> unsigned long long {
    const unsigned long long vbv = sVB[min(w, p.nwords - 1)];
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)vbv), hi = __builtin_amdgcn_readfirstlane((unsigned)(vbv >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };

  bf16x8 ka[6], kc[6];
  auto read_k = [&](int stage) {
    const char* Kb = sK + stage * KTILE;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const int coff = ((2 * ks + h) ^ kswz) << 4;
      ka[ks] = *(const bf16x8*)(Kb + krow + coff);
      kc[ks] = *(const bf16x8*)(Kb + krow + 32 * KROW + coff);
    }
  };
  if (jend > 0) read_k(0);

  int stage = 0;
  for (int j = 0; j < jend; ++j) {
    // Ring: tiles up to j+NS-2 have been issued (prologue: 0 .. NS-2, iteration i: tile i+NS-1).  Tile j is visible; retire
    // tile j+1 - its K fragments are read during this iteration - and make that a workgroup-wide fact, then refill the stage
    // tile j-1 was read from.
    if (j > 0) {
      const int ahead = min(jend - 1, j + NS - 2) - (j + 1);     // tiles that may stay in flight: j+2 .. j+NS-2
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * NCH) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NCH) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (j + NS - 1 < jend) issue_tile(j + NS - 1, (stage + NS - 1) % NS);   // into the stage tile j-1 was read from
    const int c0 = j * 64;
    const unsigned long long vb = valid_word(j);

    // ---- per block: does it need the tile, and is its mask separable here? (all wave-uniform) ----
    bool need[2], sep[2];
    unsigned long long colmask[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const Blk& B_ = blk[x];
      const bool causal_full = (c0 + 63 <= B_.wq0);
      const bool causal_none = (c0 > B_.wq0 + 31);
      const bool rect_touch = (c0 < B_.touch_hi && c0 + 64 > B_.touch_lo);
      need[x] = B_.exists && (B_.has_uniform || (B_.wave_alive && vb != 0ull && !(causal_none && !rect_touch)));
      sep[x] = causal_full || (causal_none && B_.nrect <= 1);
      unsigned long long cm = vb;
      if (causal_none) {      // right of the diagonal: only the rectangle's columns
        const int lo = max(B_.sr_lo - c0, 0), hi = min(B_.sr_hi - c0, 64);
        const unsigned long long span = (hi <= lo) ? 0ull : (((hi >= 64) ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull));
        cm = (B_.nrect == 1) ? (vb & span) : 0ull;
      }
      colmask[x] = cm;
    }
    const bool any_need = need[0] || need[1];
    const bool general = (need[0] && !sep[0]) || (need[1] && !sep[1]);

    if (any_need) {
      // V^T fragments of tile j (shared by both blocks): issued first, they land under the score MFMAs
      u32x2 vlo[4][3], vhi[4][3];
      {
        const unsigned vaddr = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)sV) + stage * VTILE + voff;
        static_for<4>([&](auto ks4) {
          static_for<3>([&](auto dt) {
            constexpr int off = ks4 * 16 * VROW + dt * 64;
            vlo[ks4][dt] = ds_read_tr<off>(vaddr);
            vhi[ks4][dt] = ds_read_tr<off + 8 * VROW>(vaddr);
          });
        });
      }

      f32x16 s0[2], s1[2];
      // ---- scores: S^T = K Q^T (+ the augmented k-step) for both blocks ----
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        Blk& B_ = blk[x];
        const bool use_word = general && need[x] && !sep[x];      // wave-uniform: per-element mask through the accumulator
        // augmented operands
        float rowbias, colsel;
        if (!need[x]) { rowbias = NEGBIG; colsel = 0.f; }
        else if (use_word) { rowbias = 0.f; colsel = 0.f; }
        else {
          const bool causal_none = (c0 > B_.wq0 + 31);
          const bool vis_row = B_.row_uniform ? true : (B_.row_alive && (!causal_none || B_.rowin));
          rowbias = vis_row ? 0.f : NEGBIG;
          colsel = B_.row_uniform ? 0.f : 1.f;
        }
        u32x4 qa = {0u, 0u, 0u, 0u};
        u32x4 k0 = {0u, 0u, 0u, 0u}, k1 = {0u, 0u, 0u, 0u};
        if (h == 0) {
          qa[0] = pack_bf16x2(-B_.m_ref, rowbias);
          qa[1] = pack_bf16x2(colsel, 1.0f);
          const unsigned long long cm = colmask[x];
          const unsigned cb0 = ((unsigned)(cm >> l31) & 1u) ? 0u : bf_neg;
          const unsigned cb1 = ((unsigned)(cm >> (32 + l31)) & 1u) ? 0u : bf_neg;
          const unsigned lb0 = (c0 + l31 < L) ? 0u : bf_neg;
          const unsigned lb1 = (c0 + 32 + l31 < L) ? 0u : bf_neg;
          k0[0] = bf_one | (bf_one << 16); k0[1] = cb0 | (lb0 << 16);
          k1[0] = bf_one | (bf_one << 16); k1[1] = cb1 | (lb1 << 16);
        }
        const bf16x8 qaug = __builtin_bit_cast(bf16x8, qa);
        const bf16x8 kaug0 = __builtin_bit_cast(bf16x8, k0), kaug1 = __builtin_bit_cast(bf16x8, k1);

        if (use_word) {
          // per-lane visibility word in register order (mma_attn_bf16.hip: count_le / low_bits)
          const int base = c0 + 4 * h;
          unsigned valid;
          if ((vb & (vb + 1ull)) == 0ull) {
            valid = low_bits(count_le(c0 + (int)__builtin_popcountll(vb) - 1 - base));
          } else {
            const unsigned long long vbh = vb >> (4 * h);
            valid = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) valid |= ((unsigned)(vbh >> (8 * k)) & 0xFu) << (4 * k);
          }
          const unsigned alive = (low_bits(count_le(B_.row - base)) | (low_bits(count_le(B_.rc1 - 1 - base)) & ~low_bits(count_le(B_.rc0 - 1 - base)))) & valid;
          const unsigned uniform = low_bits(count_le(L - 1 - base));
          const unsigned vis = B_.row_uniform ? uniform : (B_.row_alive ? alive : 0u);
          const int hid = (int)~vis;
          const int ninf = 0xFF800000;
          static_for<16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            s0[x][r] = mask_bias<r>(hid, ninf);
            s1[x][r] = mask_bias<r + 16>(hid, ninf);
          });
          asm volatile("s_nop 1" : "+v"(s0[x]), "+v"(s1[x]));
          s0[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0], B_.qf[0], s0[x], 0, 0, 0);
          s1[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[0], B_.qf[0], s1[x], 0, 0, 0);
        } else {
          s0[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0], B_.qf[0], f32x16{}, 0, 0, 0);
          s1[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[0], B_.qf[0], f32x16{}, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 1; ks < 6; ++ks) {
          s0[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], B_.qf[ks], s0[x], 0, 0, 0);
          s1[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[ks], B_.qf[ks], s1[x], 0, 0, 0);
        }
        s0[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kaug0, qaug, s0[x], 0, 0, 0);
        s1[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kaug1, qaug, s1[x], 0, 0, 0);
      }

      // the K fragments of tile j are consumed: fetch tile j+1's (published by this iteration's barrier) under the rest
      if (j + 1 < jend) read_k((stage + 1) % NS);

      // ---- online softmax per block: p = exp2(S), S already carries -m_ref and the separable mask ----
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        Blk& B_ = blk[x];
        mfma_results_settle(s0[x], s1[x]);
        float mx = max3(s0[x][0], s0[x][1], s1[x][0]);
        mx = max3(mx, s1[x][1], s0[x][2]);
#pragma unroll
        for (int r = 3; r < 16; r += 1) mx = max3(mx, s0[x][r], s1[x][r - 1]);
        mx = fmaxf(mx, s1[x][15]);
        mx = halves_max(mx);
        const float m_true = fmaxf(B_.m_true, mx + B_.m_ref);
        B_.m_true = m_true;
        const bool bad = (m_true > -1e29f) && (fabsf(m_true - B_.m_ref) > REBASE_THR);
        if (__any(bad)) {      // re-base the reference (first significant tile, then rarely): everything at the old scale moves
          const float m_new = bad ? round_bf16(m_true) : B_.m_ref;
          const float delta = m_new - B_.m_ref;
          const float alpha = __builtin_amdgcn_exp2f(fminf(fmaxf(-delta, -126.f), 126.f));
          B_.m_ref = m_new;
          B_.l_part *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { s0[x][r] -= delta; s1[x][r] -= delta; }
#pragma unroll
          for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) B_.o[dt][r] *= alpha;
        }
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[x][r] = __builtin_amdgcn_exp2f(s0[x][r]);
          s1[x][r] = __builtin_amdgcn_exp2f(s1[x][r]);
          ps += s0[x][r] + s1[x][r];
        }
        B_.l_part += ps;
      }

      // every transposed read has to be back before its registers are touched: one wait naming all destinations
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vlo[0][0]), "+v"(vhi[0][0]), "+v"(vlo[0][1]), "+v"(vhi[0][1]), "+v"(vlo[0][2]), "+v"(vhi[0][2]),
                     "+v"(vlo[1][0]), "+v"(vhi[1][0]), "+v"(vlo[1][1]), "+v"(vhi[1][1]), "+v"(vlo[1][2]), "+v"(vhi[1][2]),
                     "+v"(vlo[2][0]), "+v"(vhi[2][0]), "+v"(vlo[2][1]), "+v"(vhi[2][1]), "+v"(vlo[2][2]), "+v"(vhi[2][2]),
                     "+v"(vlo[3][0]), "+v"(vhi[3][0]), "+v"(vlo[3][1]), "+v"(vhi[3][1]), "+v"(vlo[3][2]), "+v"(vhi[3][2]));
      // ---- O^T += V^T P ----
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        Blk& B_ = blk[x];
#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) {
          bf16x8 pf;
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[e] = (__bf16)((ks4 < 2) ? s0[x][8 * (ks4 & 1) + e] : s1[x][8 * (ks4 & 1) + e]);
#pragma unroll
          for (int dt = 0; dt < 3; ++dt) {
            const u32x4 vv = {vlo[ks4][dt][0], vlo[ks4][dt][1], vhi[ks4][dt][0], vhi[ks4][dt][1]};
            B_.o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, B_.o[dt], 0, 0, 0);
          }
        }
      }
    } else {
      if (j + 1 < jend) read_k((stage + 1) % NS);
    }
    if (++stage == NS) stage = 0;
  }

  // ---- epilogue: O = O^T / l through LDS, whole 192-B rows out (see mma_attn_bf16.hip) ----
  __syncthreads();                       // every wave is done reading the ring
  constexpr int OROW = 208;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    Blk& B_ = blk[x];
    const float l_tot = halves_sum(B_.l_part);
    const bool dead = !(l_tot > 0.f);
    char* const sO = sK + (wave * 2 + x) * (32 * OROW);
    const float inv = dead ? 0.f : 1.0f / l_tot;
    if (B_.exists) {
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = B_.o[dt][4 * q4 + e] * inv;
          if (dead && p.dead_uniform && B_.row < L) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            const char* vp = vb_ + (dt * 32 + q4 * 8 + 4 * h) * 2;
            for (int t2 = 0; t2 < L; ++t2) {
              const u32x2 w2 = *(const u32x2*)(vp + (size_t)t2 * 192);
              a0 += bf16_lo(w2[0]); a1 += bf16_hi(w2[0]); a2 += bf16_lo(w2[1]); a3 += bf16_hi(w2[1]);
            }
            const float il = 1.0f / (float)L;
            v[0] = a0 * il; v[1] = a1 * il; v[2] = a2 * il; v[3] = a3 * il;
          }
          const u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *(u32x2*)(sO + l31 * OROW + (dt * 32 + q4 * 8 + 4 * h) * 2) = pk;
        }
      bf16_t* const obase = p.o + ((size_t)b * L * p.H + head) * 96;
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int ch = it * 64 + lane_o;
        const int r = ch / 12, c = ch - r * 12;
        const u32x4 w4 = *(const u32x4*)(sO + r * OROW + c * 16);
        if (B_.wq0 + r < L) *(u32x4*)((char*)(obase + (size_t)(B_.wq0 + r) * p.H * 96) + c * 16) = w4;
      }
      if (p.lse && h == 0 && B_.row < L)
        p.lse[(size_t)bh * L + B_.row] = dead ? -INFINITY : (B_.m_ref + __builtin_amdgcn_logf(l_tot)) * 0.69314718055994530942f;
    }
  }
  __syncthreads();   // the staged output tiles live in the K ring
  }                  // next rank of this workgroup
}

int attn_core64_bf16(const aki_mma_attn_core_args* a, hipStream_t stream) {
  constexpr int NS = AKI_ATTN64_STAGES;
  constexpr int SMEM = NS * (KTILE + VTILE) + MAX_VB_WORDS * 8;
  static_assert(NS >= 4 && NS <= 5, "the counted waits are written for four or five stages");
  static_assert(4 * 2 * 32 * 208 <= NS * (KTILE + VTILE), "output staging fits the K/V ring");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)mma_attn64_bf16_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return AKI_ERR_LAUNCH;
    attr_set = true;
  }
  AttnParams p = {};
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.v = (const bf16_t*)a->v; p.o = (bf16_t*)a->o; p.lse = a->lse;
  p.rects = a->rects; p.vbits = a->col_valid_bits; p.seq_lens = a->seq_lens;
  p.max_rects = a->rects ? a->max_rects : 0;
  p.B = a->B; p.H = a->H; p.L = a->L;
  p.nqt = (a->L + 255) / 256;                       // ranks of 8 blocks
  p.nwords = (a->L + 63) / 64;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  const int nbh = a->B * a->H;
  int splits = (cus + nbh - 1) / nbh;               // one workgroup per CU
  p.splits = splits < 1 ? 1 : (splits > p.nqt ? p.nqt : splits);
  int grp = ((cus + p.splits - 1) / p.splits + 7) & ~7;
  p.group_bh = grp > nbh ? nbh : grp;
  p.kvcap = a->kv_capacity > 0 ? a->kv_capacity : a->L;
  if (p.kvcap < a->L) return AKI_ERR_INVALID_ARG;
  p.scale_log2 = a->scale * 1.44269504088896340736f;
  p.dead_uniform = a->dead_rows == AKI_DEAD_ROWS_UNIFORM;
  AKI_CLEAR_ERR();
  hipLaunchKernelGGL((mma_attn64_bf16_kernel<NS>), dim3(nbh * p.splits), dim3(256), SMEM, stream, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
