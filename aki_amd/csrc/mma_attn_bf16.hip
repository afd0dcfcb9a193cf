// mma_attn_bf16.hip - span-driven modality-mutual attention core, bf16 / head_dim 96, CDNA4 MFMA.
//
// Replaces the eager path HF:phi3/modeling_phi3.py:145-167 under the reference's dense
// (B,1,L,L) mask (src/vlm.py:410-443) without ever forming an L x L tensor:
//   visible(r,c) = valid(c) && r < seq_len && ( c <= r || r in rect rows && c in rect cols )
// The mask enters as the INITIAL VALUE of the score accumulators (0 = visible, -inf = hidden): per wave and 64-key tile the
// rectangle table gives FULL (constant-0 C operand), ROWWISE (one bias value per lane), PARTIAL (per-lane visibility word
// expanded to a bias) or skipped, and the softmax code is the same for all of them.
//
// Work decomposition.  A wave owns one 32-row block of one (batch, head); four blocks of similar column extent form a
// "rank" that shares a K/V tile stream, and a persistent workgroup of NW = 4 waves walks a snake over the ranks of its
// pair (kernel body: "Persistent workgroups").  KV tiles of 64 keys go global -> LDS by global_load_lds into a 3-stage
// ring (K rows carry an XOR swizzle on the source side, V rows are read transposed with ds_read_b64_tr_b16).
// Per wave and KV tile:
//   S^T = K Q^T      12 x v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = Q kept in 24 VGPRs)
//                    -> lane (q = lane&31) holds 32 of the 64 scores of ITS OWN row: the row max /
//                    row sum are in-register reductions plus one v_permlane32_swap across halves.
//   online softmax in f32 (log2 domain: p = exp2(s*c - m)), masked scores = -inf
//   O^T += V^T P     12 MFMAs; P goes from the S^T accumulators to the B operand with no lane
//                    movement (cdna_hip_programming.md section 3, "accumulator tile as the next MFMA's
//                    operand": element j of lane-half h <-> key 16s + 8(j>>2) + 4h + (j&3)); the
//                    matching V^T A-operand comes from two ds_read_b64_tr_b16 per fragment.
// O^T keeps the query on the lane, so the rescale factor and the final 1/l are lane-local.
#include "attn_mma_common.h"

// Lab knobs (compile-time; tools/attn_*.py build variants with -D):
//   AKI_ATTN_L2_ROWS   sequences get L / AKI_ATTN_L2_ROWS workgroups per pair at least (fewer pairs in flight per L2)
//   AKI_ATTN_SCHED_MAX block ranking by extent up to this many 32-row blocks (<= 64: one lane per block), 0 = position order
//   AKI_ATTN_SPLITS    force the number of workgroups per pair
#ifndef AKI_ATTN_L2_ROWS
#define AKI_ATTN_L2_ROWS 256
#endif
#ifndef AKI_ATTN_SCHED_MAX
#define AKI_ATTN_SCHED_MAX 64
#endif

namespace aki {

// SWP (lab library only) = software-pipelined tile loop (see "Software-pipelined loop" in the kernel body): P V of tile j-1 is
// issued in front of K Q^T of tile j, so a wave's matrix work comes in one burst of 24 MFMAs per tile that covers its own LDS
// fragment reads, and its softmax VALU sits alone between two bursts.  Same results bit for bit; measured 6-9 % SLOWER than the
// plain loop on one box (B8 L655: 68.3 vs 64.3 us; B4 L4096: 670 vs 617 us; tools/attn_ab.py) - kept as a recorded experiment.
template <int NW, int MODE>      // MODE 0: plain tile loop (product); 1: software-pipelined; 2: software-pipelined + 8-wave ping-pong (NW = 8); 3: plain, DMA issued behind the score MFMAs; 4 (lab): plain, per-stage arrival counters in LDS instead of the tile barrier
__global__ __launch_bounds__(NW * 64, 2) void mma_attn_bf16_kernel(const AttnParams p) {
  constexpr bool SWP = MODE == 1 || MODE == 2;
  static_assert(MODE != 2 || NW == 8, "the ping-pong loop is written for two groups of four waves");
  constexpr int BQ = NW * 32;
  constexpr int NT = NW * 64;
  constexpr int NCH = (64 * 12 + NT - 1) / NT;  // 16-B chunks per thread per tile (K and V each)
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * KTILE + NSTAGE * VTILE + MAX_VB_WORDS * 8 + (MODE == 4 ? 64 : 0)];
  char* const sK = smem;
  char* const sV = smem + NSTAGE * KTILE;
  unsigned long long* const sVB = (unsigned long long*)(smem + NSTAGE * KTILE + NSTAGE * VTILE);
  // MODE 4: sCnt[s] = waves whose pieces of the tile in stage s have landed, sCnt[NSTAGE + s] = waves that have finished reading it
  // (both per rank, monotonic inside a rank, zeroed between the rank-end barriers)
  unsigned* const sCnt = (unsigned*)(smem + NSTAGE * KTILE + NSTAGE * VTILE + MAX_VB_WORDS * 8);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  // Persistent workgroups.  A (batch, head) pair has nqt workgroup-sized pieces of work ("ranks", 0 = heaviest, four
  // 32-row blocks each); it is served by p.splits workgroups, and workgroup `sidx` of the pair walks the ranks
  // k*splits + (k even ? sidx : splits-1-sidx), k = 0, 1, ... - a snake over the descending order, so the splits of a
  // pair carry about the same work and the grid (B*H*splits, chosen by the host to be about the number of resident
  // workgroup slots) is one balanced round with no dispatch gaps, where rank-sized workgroups packed at ~75 %.  All
  // ranks of a workgroup read the same K/V (L2 / Infinity-Cache hits after the first), and the per-sample work - valid
  // words to LDS, block extents and their ranking - is done once.  blockIdx = sidx * (B*H) + pair: the splits of a pair
  // share blockIdx % 8, i.e. the XCD and its L2, whenever B*H is a multiple of 8.
  // Pairs are issued in groups of p.group_bh (a multiple of 8), split-major inside a group, so that all splits of a
  // pair are resident together and few enough pairs are in flight per XCD for their K/V to stay in its L2.
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

  // K/V tiles go global -> LDS directly (global_load_lds, 16 B/lane): the LDS image is lane-linear, i.e. exactly the
  // contiguous 12 KiB tile of the head-major layout; the K swizzle is applied on the per-lane SOURCE address.
  int tid_o = tid;   // re-made opaque once per rank: keeps hipcc from hoisting the per-chunk address arithmetic out of
                     // the rank loop (with it hoisted the tile loop spilled: 256 VGPRs + 31 in scratch)
  auto issue_k = [&](int j, int stage) {
    const int c0 = j * 64;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (NW == 8 && i == 1 && wave >= 4) break;  // 768 chunks on 512 threads: the half round of K belongs to waves 0-3 ...
      const int ch = i * NT + tid_o;              // 16-B chunk index inside the tile image
      const int kr = ch / 12, pos = ch - kr * 12;
      const size_t rowoff = (size_t)min(c0 + kr, L - 1) * 192;
      const int srcchunk = pos ^ ((kr >> 2) & 3);
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(kb + rowoff + srcchunk * 16), AKI_LDS_PTR(sK + stage * KTILE + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto issue_v = [&](int j, int stage) {
    const int c0 = j * 64;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (NW == 8 && i == 1 && wave < 4) break;   // ... the half round of V to waves 4-7: three pieces per wave and tile either way
      const int sh = (NW == 8 && i == 1) ? 256 : 0;
      const int ch = i * NT + tid_o - sh;
      const int kr = ch / 12, pos = ch - kr * 12;
      const size_t rowoff = (size_t)min(c0 + kr, L - 1) * 192;
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(vb_ + rowoff + pos * 16), AKI_LDS_PTR(sV + stage * VTILE + (i * NT + wave * 64 - sh) * 16), 16, 0, 0);
    }
  };
  auto issue_tile = [&](int j, int stage) {
    const int c0 = j * 64;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = i * NT + tid_o;              // 16-B chunk index inside the tile image
      const int kr = ch / 12, pos = ch - kr * 12;
      const size_t rowoff = (size_t)min(c0 + kr, L - 1) * 192;
      const int srcchunk = pos ^ ((kr >> 2) & 3);
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(kb + rowoff + srcchunk * 16), AKI_LDS_PTR(sK + stage * KTILE + (i * NT + wave * 64) * 16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(AKI_GLOBAL_PTR(vb_ + rowoff + pos * 16), AKI_LDS_PTR(sV + stage * VTILE + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
  };

  // ---- once per workgroup: the sample's valid words, rectangles, block extents and their ranking -------------
  constexpr int SCHED_MAX = AKI_ATTN_SCHED_MAX;   // <= 64: one lane per 32-row block
  const int nblk = (L + 31) >> 5;
  const bool sched = nblk <= SCHED_MAX;        // kernel-uniform
  const int Lb = p.seq_lens ? min(p.seq_lens[b], L) : L;
  // Valid-column words of this sample go to LDS once: a per-tile global load would share the vmcnt queue with the
  // K/V prefetch and its wait would drain the prefetch before the tile's compute (measured: 3x slower loop).  The
  // first barrier of the tile loop orders these writes before their first read.
  for (int w = tid; w < p.nwords; w += NT) {
    unsigned long long vbw;
    if (p.vbits) vbw = p.vbits[(size_t)b * p.nwords + w];
    else vbw = (w * 64 + 64 <= L) ? ~0ull : ((1ull << (L - w * 64)) - 1ull);
    sVB[w] = vbw;
  }
  if constexpr (MODE == 4) {
    if (tid < 2 * NSTAGE) sCnt[tid] = 0u;
    __syncthreads();                                // also orders the valid words: this mode has no barrier in its tile loop
  }
  const aki_mma_rect* const rects_b = p.rects + (size_t)b * p.max_rects;
  auto rect_at = [&](int i) -> aki_mma_rect {       // wave-uniform: one s_load_dwordx4 through the scalar cache
    const unsigned long long pa = (unsigned long long)(rects_b + i);
    const unsigned long long pu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pa);
    u32x4 r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(pu) : "memory");
    aki_mma_rect o;
    o.row_lo = (int)r[0]; o.row_hi = (int)r[1]; o.col_lo = (int)r[2]; o.col_hi = (int)r[3];
    return o;
  };
  // Columns a 32-row block starting at r0 has to walk: its causal extent, widened by every rectangle that touches
  // it.  Rows >= seq_len are the all-zero mask rows of batch stacking: under the reference's finfo.min hand-off they
  // get a UNIFORM softmax over all L columns; they run through the normal MFMA path as "every column visible,
  // score 0", so a block that owns such rows walks all KV tiles.
  auto block_extent = [&](int r0) -> int {
    if (r0 >= L) return -1;                       // no such block
    int ext = min(r0 + 32, L);
    for (int i = 0; i < p.max_rects; ++i) {
      const aki_mma_rect r = rect_at(i);
      if (r.row_hi > r.row_lo && r.col_hi > r.col_lo && r.row_lo < r0 + 32 && r.row_hi > r0) ext = max(ext, min(r.col_hi, L));
    }
    if (p.dead_uniform && min(r0 + 32, L) > Lb) ext = L;
    return ext;
  };
  // Which 32-row block a wave owns: the four waves of a workgroup share one K/V tile stream, so a rank walks to the
  // LARGEST extent among its blocks and a wave whose block ends earlier idles at the barriers.  In position order that
  // wastes a lot when the sequence is short (image rows inside a rectangle walk ~L columns, their text neighbours a few
  // tiles): up to SCHED_MAX blocks are therefore ranked by extent (descending, later block first on ties) and rank g
  // takes blocks 4g .. 4g+3 of that order - blocks of similar length share a stream.  Every wave does the ranking for
  // itself (lane i <-> block i, v_readlane): no LDS, no barrier.  Longer sequences keep position order (neighbouring
  // blocks differ by at most two tiles there), late blocks first.
  int ext_s = -1, rank_s = 0x7fff;               // sched: lane i holds extent and rank of block i
  if (sched) {
    ext_s = block_extent(32 * lane);
    const int key = ext_s < 0 ? -1 : ext_s * 64 + lane;        // unique; lanes past nblk never outrank a block
    int rank = 0;
    for (int i = 0; i < nblk; i += 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) rank += __builtin_amdgcn_readlane(key, i + k) > key ? 1 : 0;
    }
    rank_s = lane < nblk ? rank : 0x7fff;
  }
  bf16x8 qf[6];
  auto load_q = [&](int row) {   // Q fragments (B operand of S^T = K Q^T): lane (q=l31, h) holds Q[q][16ks + 8h .. +7]
    const bf16_t* qrow = qb + (size_t)min(row, L - 1) * 96 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
  };
  // per-lane LDS offsets
  const int kswz = (l31 >> 2) & 3;   // (row>>2)&3 for row = 32*blk + l31
  const int krow = l31 * KROW;
  const int voff = (4 * h + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  for (int kk = 0;; ++kk) {
  const int g = kk * p.splits + ((kk & 1) ? p.splits - 1 - sidx : sidx);   // this workgroup's next rank
  if (g >= p.nqt) break;
  int lane_o = lane;
  asm volatile("" : "+v"(tid_o), "+v"(lane_o));
  // ---- per rank: the first two K/V tiles and the Q fragments are put in flight together ----------------------
  if constexpr (SWP) {
    issue_k(0, 0);
    issue_v(0, 0);
    if (L > 64) issue_k(1, 1);
  } else {
    issue_tile(0, 0);
    if (L > 64) issue_tile(1, 1);
  }
  int wq0, hi_col;
  if (sched) {
    hi_col = 0;
    wq0 = L;                                      // rank past nblk (last rank of the sample): the wave idles
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const unsigned long long m = __ballot(rank_s == NW * g + w);
      if (m != 0ull) {
        const int l = __builtin_ctzll(m);
        hi_col = max(hi_col, __builtin_amdgcn_readlane(ext_s, l));
        if (w == wave) wq0 = 32 * l;
      }
    }
  } else {
    const int q0 = (p.nqt - 1 - g) * BQ;
    wq0 = q0 + wave * 32;
    const int ext = block_extent(q0 + 32 * (lane & (NW - 1)));
    hi_col = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) hi_col = max(hi_col, __builtin_amdgcn_readlane(ext, w));
  }
  wq0 = __builtin_amdgcn_readfirstlane(wq0);
  hi_col = __builtin_amdgcn_readfirstlane(hi_col);
  load_q(wq0 + l31);
  const int row = wq0 + l31;

  // ---- per-wave rectangle summary, per-lane unlock range -----------------------------------------
  int touch_lo = 0x7fffffff, touch_hi = 0, full_lo = 0, full_hi = 0;
  int rc0 = 0, rc1 = 0;
  for (int i = 0; i < p.max_rects; ++i) {
    const aki_mma_rect r = rect_at(i);
    if (r.row_hi > r.row_lo && r.col_hi > r.col_lo) {
      if (r.row_lo < wq0 + 32 && r.row_hi > wq0) {
        touch_lo = min(touch_lo, r.col_lo);
        touch_hi = max(touch_hi, r.col_hi);
        if (r.row_lo <= wq0 && r.row_hi >= wq0 + 32) { full_lo = r.col_lo; full_hi = r.col_hi; }
      }
      if (row >= r.row_lo && row < r.row_hi) { rc0 = r.col_lo; rc1 = r.col_hi; }
    }
  }
  const int jend = (hi_col + 63) >> 6;
  const bool wave_alive = wq0 < Lb;            // wave-uniform
  const bool wave_has_dead = min(wq0 + 32, L) > Lb;   // some rows of the wave are beyond seq_len (rows >= L do not exist)
  const bool row_alive = row < Lb;
  const bool has_uniform = p.dead_uniform && wave_has_dead;   // wave-uniform
  const bool row_uniform = p.dead_uniform && !row_alive;

  if (row_uniform) {   // uniform-softmax rows: score 0 on every column = an all-zero query
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) qf[ks] = bf16x8{};
  }

  f32x16 o[3];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_part = 0.f;

  // Retire everything issued above with a wait hipcc can see (builtin) before the loop: otherwise the compiler guards
  // the Q registers with its own vmcnt(0) in front of the first MFMA of EVERY iteration, which drains the LDS-DMA
  // prefetch (its model does not see the inline-asm counted waits below).
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

  auto valid_word = [&](int w) -> unsigned long long {
    const unsigned long long vbv = sVB[min(w, p.nwords - 1)];
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)vbv), hi = __builtin_amdgcn_readfirstlane((unsigned)(vbv >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  unsigned long long vb_next = 0ull;

  if constexpr (SWP) {
  // ---- Software-pipelined loop (lab).  A tile is processed in two phases:
  //   M(j): P V of tile j-1 (V^T fragment reads, 12 MFMAs) followed by K Q^T of tile j (K fragment reads issued behind the P V
  //         MFMAs, mask bias, 12 MFMAs): one burst of 24 MFMAs that covers its own LDS reads;
  //   V(j): the softmax of tile j - VALU only - leaving P as packed bf16 (16 registers) for the next M phase.
  // V lags K by one tile in the ring: at iteration j the DMA of K tile j+2 replaces K tile j-1 and V tile j+1 replaces
  // V tile j-2 (V tile j-1 is still being read).
  // MODE 1 (4 waves): barrier, M(j), V(j) per iteration.  MODE 2 (8 waves = two groups of four on the same four SIMDs): two
  // barriers per iteration and the groups half an iteration apart - group A runs M(j) then V(j), group B runs V(j-1) then M(j) -
  // so that on every SIMD one wave is in its matrix phase while its partner is in its VALU phase (MI355X guide, "Two waves per
  // SIMD"); waves 4-7, the younger half, get static priority 1.
  bf16x8 pp[4];
  f32x16 s0, s1;
  const bool grp_b = (MODE == 2) && wave >= NW / 2;           // wave-uniform
  if (MODE == 2 && grp_b) __builtin_amdgcn_s_setprio(1);
  auto tile_active = [&](int j) -> bool {       // wave-uniform: does this wave's block see anything of tile j?
    if (j >= jend) return false;
    const int c0 = j * 64;
    const unsigned long long vb = valid_word(j);
    const bool causal_none = (c0 > wq0 + 31);
    const bool rect_touch = (c0 < touch_hi && c0 + 64 > touch_lo);
    return has_uniform || !(!wave_alive || vb == 0ull || (causal_none && !rect_touch));
  };
  auto pv = [&](int j) {           // O^T += V^T P of tile j-1 (P in pp): the V^T fragment reads, then 12 MFMAs
    {
      u32x2 vlo[4][3], vhi[4][3];
      const unsigned vaddr = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)sV) + ((j + NSTAGE - 1) % NSTAGE) * VTILE + voff;
      static_for<4>([&](auto ks4) {
        static_for<3>([&](auto dt) {
          constexpr int off = ks4 * 16 * VROW + dt * 64;
          vlo[ks4][dt] = ds_read_tr<off>(vaddr);
          vhi[ks4][dt] = ds_read_tr<off + 8 * VROW>(vaddr);
        });
      });
      // every transposed read has to be back before its registers are touched
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vlo[0][0]), "+v"(vhi[0][0]), "+v"(vlo[0][1]), "+v"(vhi[0][1]), "+v"(vlo[0][2]), "+v"(vhi[0][2]),
                     "+v"(vlo[1][0]), "+v"(vhi[1][0]), "+v"(vlo[1][1]), "+v"(vhi[1][1]), "+v"(vlo[1][2]), "+v"(vhi[1][2]),
                     "+v"(vlo[2][0]), "+v"(vhi[2][0]), "+v"(vlo[2][1]), "+v"(vhi[2][1]), "+v"(vlo[2][2]), "+v"(vhi[2][2]),
                     "+v"(vlo[3][0]), "+v"(vhi[3][0]), "+v"(vlo[3][1]), "+v"(vhi[3][1]), "+v"(vlo[3][2]), "+v"(vhi[3][2]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks4 = 0; ks4 < 4; ++ks4) {
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
          const u32x4 vv = {vlo[ks4][dt][0], vlo[ks4][dt][1], vhi[ks4][dt][0], vhi[ks4][dt][1]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pp[ks4], o[dt], 0, 0, 0);
        }
      }
    }
  };
  auto qk = [&](int j) {           // S^T = bias + K Q^T of tile j -> s0, s1.  Issued behind the P V MFMAs: the K fragment reads land while those execute
    const int c0 = j * 64;
    const unsigned long long vb = valid_word(j);
    const bool causal_full = (c0 + 63 <= wq0);
    const bool causal_none = (c0 > wq0 + 31);
    const bool rect_full = (c0 >= full_lo && c0 + 64 <= full_hi);
    {
      bf16x8 ka[6], kc[6];
      const char* Kb = sK + (j % NSTAGE) * KTILE;
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        const int coff = ((2 * ks + h) ^ kswz) << 4;
        ka[ks] = *(const bf16x8*)(Kb + krow + coff);
        kc[ks] = *(const bf16x8*)(Kb + krow + 32 * KROW + coff);
      }
      // the mask bias = the initial value of the score accumulators (see the plain loop)
      const bool full = (vb == ~0ull) && (causal_full || rect_full) && !wave_has_dead;
      const bool lane_covers = rc0 <= c0 && c0 + 64 <= rc1;
      const bool lane_cut = !lane_covers && rc0 < c0 + 64 && rc1 > c0;
      const bool rowwise = !full && causal_none && vb == ~0ull && !wave_has_dead && !__any(lane_cut);
      if (full) {
        // no bias: the first MFMA of each chain takes the constant 0 as its C operand (below)
      } else if (rowwise) {
        const float lane_bias = lane_covers ? 0.f : -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = lane_bias; s1[r] = lane_bias; }
      } else {
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
        const unsigned alive = (low_bits(count_le(row - base)) | (low_bits(count_le(rc1 - 1 - base)) & ~low_bits(count_le(rc0 - 1 - base)))) & valid;
        const unsigned uniform = low_bits(count_le(L - 1 - base));            // every column < L
        const unsigned vis = row_uniform ? uniform : (row_alive ? alive : 0u);
        const int hid = (int)~vis;
        const int ninf = 0xFF800000;
        static_for<16>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          s0[r] = mask_bias<r>(hid, ninf);
          s1[r] = mask_bias<r + 16>(hid, ninf);
        });
        asm volatile("s_nop 1" : "+v"(s0), "+v"(s1));   // VALU write inside inline asm -> MFMA C operand (see the plain loop)
      }
      if (full) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0], qf[0], f32x16{}, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[0], qf[0], f32x16{}, 0, 0, 0);
      } else {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0], qf[0], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[0], qf[0], s1, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 1; ks < 6; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[ks], qf[ks], s1, 0, 0, 0);
      }
    }
  };
  auto softmax_tile = [&]() {      // softmax of the score tile in s0, s1 -> P (packed bf16) in pp, running max / sum, O brought to the new maximum
    __builtin_amdgcn_sched_barrier(0);
    mfma_results_settle(s0, s1);   // the max3 chain below is inline asm
    float mx = tile_max32(s0, s1);
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
    if (__any(moved)) {   // O holds every tile before this one (their P V has been issued): bring it to the new maximum
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
#pragma unroll
    for (int ks4 = 0; ks4 < 4; ++ks4)
#pragma unroll
      for (int e = 0; e < 8; ++e) pp[ks4][e] = (__bf16)((ks4 < 2) ? s0[8 * (ks4 & 1) + e] : s1[8 * (ks4 & 1) + e]);
  };
  auto top_of_iteration = [&](int j) {
    // needed now: K tile j and V tile j-1, i.e. everything issued before iteration j-1; iteration j-1 issued K(j+1) and then V(j).
    // Pieces per wave and tile: NCH of each with 4 waves; with 8 waves K 2 + V 1 for waves 0-3, K 1 + V 2 for waves 4-7.
    if constexpr (MODE == 2) {
      if (j + 1 < jend) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (j < jend) { if (wave < 4) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (j + 1 < jend) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NCH) : "memory");
      else if (j < jend) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (j + 2 < jend) issue_k(j + 2, (j + 2) % NSTAGE);
    if (j + 1 < jend) issue_v(j + 1, (j + 1) % NSTAGE);
  };
  // The branches below put the second barrier of an iteration INSIDE both arms so that the score tile (group A) / the packed P
  // (group B) is provably dead on the arm that does not use it - with the condition tested twice hipcc kept both live across
  // the whole loop and spilled 94 registers.
  if (!grp_b) {                    // MODE 1, and group A of MODE 2: [P V (j-1), K Q^T (j)] | softmax (j)
    bool have_p = false;
    for (int j = 0; j <= jend; ++j) {
      top_of_iteration(j);
      if (have_p) pv(j);
      if (tile_active(j)) {
        qk(j);
        if constexpr (MODE == 2) __builtin_amdgcn_s_barrier();
        softmax_tile();
        have_p = true;
      } else {
        if constexpr (MODE == 2) __builtin_amdgcn_s_barrier();
        have_p = false;
      }
    }
  } else {                         // group B of MODE 2, half an iteration behind: softmax (j-1) | [P V (j-1), K Q^T (j)]
    bool pend = false;
    for (int j = 0; j <= jend; ++j) {
      top_of_iteration(j);
      if (pend) {
        softmax_tile();
        __builtin_amdgcn_s_barrier();
        pv(j);
      } else {
        __builtin_amdgcn_s_barrier();
      }
      pend = tile_active(j);
      if (pend) qk(j);
    }
  }
  if (MODE == 2 && grp_b) __builtin_amdgcn_s_setprio(0);
  } else {
  int stage = 0;
  for (int j = 0; j < jend; ++j) {
    // tile j's pieces are older than tile j+1's 2*NCH: wait for them, then make it a workgroup-wide fact
    if (j + 1 < jend) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NCH) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (MODE == 4) {
      // this wave's pieces of tile j are in LDS: say so, then wait for the other waves' pieces of THIS tile only - nobody waits
      // for a wave that is still computing tile j-1 (hypothesis of VERDICT r3 item 4: the tile barrier makes every wave pay for the
      // slowest block of its rank)
      if (lane == 0) __hip_atomic_fetch_add(sCnt + stage, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned want = (unsigned)NW * (unsigned)(j / NSTAGE + 1);
      while (__hip_atomic_load(sCnt + stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) { }
      asm volatile("" ::: "memory");
    } else {
      __builtin_amdgcn_s_barrier();
    }
    // DMA of tile j+2 into the stage read in iteration j-1.  MODE 3 (lab): issued BEHIND the score MFMAs instead of here - a piece
    // costs the issuing wave 60-185 cycles (MI355X guide, cycle constants), six of them in front of the K fragment reads are a
    // good part of the ~730 cycles between the barrier and the first MFMA (profiles/r01i_attn_phase_cycles.txt)
    constexpr bool DMA_LATE = (MODE == 3 || MODE == 4);
    const int dma_stage = stage >= 1 ? stage - 1 : NSTAGE - 1;
    if (!DMA_LATE && j + 2 < jend) issue_tile(j + 2, dma_stage);
    const int c0 = j * 64;
    // valid-column word of the tile (wave-uniform, SGPRs): word 0 is read here, behind the barrier that publishes the
    // prologue's LDS writes; every later word is fetched at the end of the previous tile, under its PV MFMAs
    const unsigned long long vb = j == 0 ? valid_word(0) : vb_next;
    const bool causal_full = (c0 + 63 <= wq0);
    const bool causal_none = (c0 > wq0 + 31);
    const bool rect_full = (c0 >= full_lo && c0 + 64 <= full_hi);
    const bool rect_touch = (c0 < touch_hi && c0 + 64 > touch_lo);
    const bool skip = !has_uniform && (!wave_alive || vb == 0ull || (causal_none && !rect_touch));
    if (!skip) {
      const char* Kb = sK + stage * KTILE;
      // All K fragments of the tile are fetched before the first MFMA (one LDS wait instead of one in front of every
      // MFMA pair: with 2 waves per SIMD that ~128-cycle LDS latency, 24 times per tile, was the dominant stall).
      bf16x8 ka[6], kc[6];
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        const int coff = ((2 * ks + h) ^ kswz) << 4;
        ka[ks] = *(const bf16x8*)(Kb + krow + coff);
        kc[ks] = *(const bf16x8*)(Kb + krow + 32 * KROW + coff);
      }
      // The mask enters as the INITIAL VALUE of the score accumulators (0 = visible, -inf = hidden): S = bias + K Q^T
      // costs the MFMAs nothing, the bias is computed while the K reads above are in flight, and the softmax below
      // is the same code for FULL and PARTIAL tiles.  (Rows that get the uniform softmax carry an all-zero Q.)
      f32x16 s0, s1;
      const bool full = (vb == ~0ull) && (causal_full || rect_full) && !wave_has_dead;
      // ROWWISE tiles: right of the diagonal, every column valid, and each row's rectangle either covers the whole tile
      // or misses it (image rows sharing a 32-row block with text rows): the bias is one value per lane.
      const bool lane_covers = rc0 <= c0 && c0 + 64 <= rc1;
      const bool lane_cut = !lane_covers && rc0 < c0 + 64 && rc1 > c0;
      const bool rowwise = !full && causal_none && vb == ~0ull && !wave_has_dead && !__any(lane_cut);
      if (full) {
        // no bias: the first MFMA of each chain takes the constant 0 as its C operand (no 32 v_mov per tile)
        __builtin_amdgcn_sched_barrier(0);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0], qf[0], f32x16{}, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[0], qf[0], f32x16{}, 0, 0, 0);
      } else if (rowwise) {
        const float lane_bias = lane_covers ? 0.f : -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = lane_bias; s1[r] = lane_bias; }
      } else {
        // Per-lane visibility word in register order (bit i <-> register i of s0:s1, see count_le): the causal part is a
        // prefix, the rectangle an interval; valid bits that form a prefix (right padding, the usual case) are one
        // more prefix, anything else is gathered bit group by bit group.
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
        const unsigned alive = (low_bits(count_le(row - base)) | (low_bits(count_le(rc1 - 1 - base)) & ~low_bits(count_le(rc0 - 1 - base)))) & valid;
        const unsigned uniform = low_bits(count_le(L - 1 - base));            // every column < L
        const unsigned vis = row_uniform ? uniform : (row_alive ? alive : 0u);
        const int hid = (int)~vis;
        const int ninf = 0xFF800000;
        static_for<16>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          s0[r] = mask_bias<r>(hid, ninf);
          s1[r] = mask_bias<r + 16>(hid, ninf);
        });
        // VALU write inside inline asm -> MFMA C operand: hipcc pads no hazard whose producer it cannot see (guide 5.7
        // item 2).  Without these two wait states the MFMA occasionally read the register's previous content - one score
        // of a PARTIAL tile off, run-to-run differences of an ulp in ~0.5 % of the outputs.
        asm volatile("s_nop 1" : "+v"(s0), "+v"(s1));   // data-dependent on every bias register: cannot be moved ahead of them
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!full) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0], qf[0], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[0], qf[0], s1, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 1; ks < 6; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[ks], qf[ks], s1, 0, 0, 0);
      }
      if (DMA_LATE && j + 2 < jend) {
        if constexpr (MODE == 4) {   // the stage held tile j-1: every wave must have finished reading it
          if (j >= 1) {
            const unsigned wantf = (unsigned)NW * (unsigned)((j - 1) / NSTAGE + 1);
            while (__hip_atomic_load(sCnt + NSTAGE + dma_stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < wantf) { }
          }
        }
        issue_tile(j + 2, dma_stage);
      }
      // The V^T fragments do not depend on the softmax: issue their transposed reads now, they land under the VALU work.
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
      __builtin_amdgcn_sched_barrier(0);
      mfma_results_settle(s0, s1);   // the max3 chain below is inline asm
      float mx = tile_max32(s0, s1);
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
      if (__any(moved)) {   // after the first tiles the running max rarely moves: skip the 48-register rescale
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }

      // every transposed read has to be back before its registers are touched: one wait naming all destinations
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vlo[0][0]), "+v"(vhi[0][0]), "+v"(vlo[0][1]), "+v"(vhi[0][1]), "+v"(vlo[0][2]), "+v"(vhi[0][2]),
                     "+v"(vlo[1][0]), "+v"(vhi[1][0]), "+v"(vlo[1][1]), "+v"(vhi[1][1]), "+v"(vlo[1][2]), "+v"(vhi[1][2]),
                     "+v"(vlo[2][0]), "+v"(vhi[2][0]), "+v"(vlo[2][1]), "+v"(vhi[2][1]), "+v"(vlo[2][2]), "+v"(vhi[2][2]),
                     "+v"(vlo[3][0]), "+v"(vhi[3][0]), "+v"(vlo[3][1]), "+v"(vhi[3][1]), "+v"(vlo[3][2]), "+v"(vhi[3][2]));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MODE == 4) {     // every LDS read of this tile has returned (the wait above): its stage may be overwritten
        if (lane == 0) __hip_atomic_fetch_add(sCnt + NSTAGE + stage, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
#pragma unroll
      for (int ks4 = 0; ks4 < 4; ++ks4) {
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (__bf16)((ks4 < 2) ? s0[8 * (ks4 & 1) + e] : s1[8 * (ks4 & 1) + e]);
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
          const u32x4 vv = {vlo[ks4][dt][0], vlo[ks4][dt][1], vhi[ks4][dt][0], vhi[ks4][dt][1]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[dt], 0, 0, 0);
        }
      }
    } else {
      if constexpr (MODE == 4) {     // a skipped tile is "read" at once
        if (lane == 0) __hip_atomic_fetch_add(sCnt + NSTAGE + stage, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (DMA_LATE && j + 2 < jend) {
        if constexpr (MODE == 4) {
          if (j >= 1) {
            const unsigned wantf = (unsigned)NW * (unsigned)((j - 1) / NSTAGE + 1);
            while (__hip_atomic_load(sCnt + NSTAGE + dma_stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < wantf) { }
          }
        }
        issue_tile(j + 2, dma_stage);
      }
    }
    vb_next = valid_word(j + 1);
    if (++stage == NSTAGE) stage = 0;
  }

  }
  // ---- epilogue: O = O^T / l.  A row that is inside seq_len but saw no visible column at all (e.g. left padding)
  // is rare: under AKI_DEAD_ROWS_UNIFORM its lanes average V themselves; otherwise it is written as zeros.
  // The lane-local layout (one query row per lane, 4 features per register quad) would store 8-byte pieces at a
  // 6 KiB row stride - 32 cache lines per store instruction.  Each wave therefore passes its 32 x 96 tile through
  // its own 6.5 KiB of the (now idle) K ring and writes whole 192-B rows, 16 B per lane. ----
  const float l_tot = halves_sum(l_part);
  const bool dead = !(l_tot > 0.f);
  __syncthreads();                       // every wave is done reading the ring
  if constexpr (MODE == 4) {
    if (tid < 2 * NSTAGE) sCnt[tid] = 0u;   // the next rank counts from zero (ordered by the barrier at the end of this epilogue)
  }
  constexpr int OROW = 208;              // 192 B + 16: the 8-B writes of 32 rows land 2-way instead of 8-way conflicted
  char* const sO = sK + wave * (32 * OROW);
  {
    const float inv = dead ? 0.f : 1.0f / l_tot;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[dt][4 * q4 + e] * inv;
        if (dead && p.dead_uniform && row < L) {
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
  }
  // same-wave LDS round trip: the compiler's lgkmcnt wait between the writes and the reads is all the ordering needed
  bf16_t* const obase = p.o + ((size_t)b * L * p.H + head) * 96;
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int ch = it * 64 + lane_o;               // 16-B chunk of the wave's tile: row ch/12, chunk ch%12
    const int r = ch / 12, c = ch - r * 12;
    const u32x4 w4 = *(const u32x4*)(sO + r * OROW + c * 16);
    if (wq0 + r < L) *(u32x4*)((char*)(obase + (size_t)(wq0 + r) * p.H * 96) + c * 16) = w4;
  }
  if (p.lse && h == 0 && row < L) p.lse[(size_t)bh * L + row] = dead ? -INFINITY : (m_run + __builtin_amdgcn_logf(l_tot)) * 0.69314718055994530942f;
  __syncthreads();   // the staged output tiles live in the K ring: every wave has read its tile back before the next rank's DMA
  }                  // next rank of this workgroup
}

// the long-sequence core (mma_attn64_bf16.hip): 64 rows per wave, one wave per SIMD
int attn_core64_bf16_launch(AttnParams p, int cus, hipStream_t stream, int exact_max);
// Sequences of at least this many rows go to it (measured crossover, tools/attn64_crossover.py -> profiles/r06_attn64_crossover.txt: the two
// kernels tie at 1536 rows, the long one is 5 % ahead at 1792 for batch 1, 4 and 8 alike)
#ifndef AKI_ATTN64_MIN_L
#define AKI_ATTN64_MIN_L 1792
#endif

#ifdef AKI_LAB_HOOKS
// Lab library only: structures of this core switched in for an A/B in one process (tools/attn_ab.py).  (Variant 2 was the 64-rows-per-wave core,
// one wave per SIMD - as compiled by hipcc 1.7x SLOWER than this kernel; its source left the tree in round 5, EXPERIMENTS.md keeps the numbers.)
int g_attn_variant = 0;   // 0 = product rule (this kernel below AKI_ATTN64_MIN_L rows, the 64-row kernel from there on), 1 / 2 = this kernel at every length, 9 = the 64-row kernel at every length, 10 = the 64-row kernel with the exact running maximum (THR 0: bit-identical to this kernel), 3 = this kernel with the software-pipelined tile loop (lab: 6-9 % slower), 4 = 8-wave ping-pong on the software-pipelined loop, 5 = DMA behind the score MFMAs, 6 / 7 = one / three workgroups per pair, 8 = LDS arrival counters instead of the tile barrier
#endif

int attn_core_bf16(const aki_mma_attn_core_args* a, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (a->Dh != 96) return AKI_ERR_UNSUPPORTED;
  if (a->max_rects < 0 || a->max_rects > AKI_MAX_RECTS) return AKI_ERR_INVALID_ARG;
  if ((a->L + 63) / 64 > MAX_VB_WORDS) return AKI_ERR_UNSUPPORTED;
  AKI_CHECK_ALIGN16(a->q); AKI_CHECK_ALIGN16(a->k); AKI_CHECK_ALIGN16(a->v); AKI_CHECK_ALIGN16(a->o);
  (void)ws; (void)ws_bytes;  // the bf16 path needs no scratch (kept in the signature for the f32 path)
  constexpr int NW = 4;
  AttnParams p = {};
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.v = (const bf16_t*)a->v; p.o = (bf16_t*)a->o; p.lse = a->lse;
  p.rects = a->rects; p.vbits = a->col_valid_bits; p.seq_lens = a->seq_lens;
  p.max_rects = a->rects ? a->max_rects : 0;
  p.B = a->B; p.H = a->H; p.L = a->L;
  p.nqt = (a->L + NW * 32 - 1) / (NW * 32);
  p.nwords = (a->L + 63) / 64;
  {
    // workgroups per pair: enough to fill the resident slots (2 per CU) once, at most one per rank
    static int cus = 0;                      // queried once: hipGetDeviceProperties takes milliseconds
    if (cus == 0) {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      cus = n;
    }
    const int nbh = a->B * a->H;
    int splits = (2 * cus + nbh - 1) / nbh;
#ifdef AKI_ATTN_SPLITS
    splits = AKI_ATTN_SPLITS;
#endif
#ifdef AKI_LAB_HOOKS
    if (g_attn_variant == 6) splits = 1;     // ONE workgroup per (batch, head) pair: K/V cross the fabric once, half the resident slots stay empty at B*H = 256
    if (g_attn_variant == 7) splits = 3;
#endif
    const int s_l2 = a->L / AKI_ATTN_L2_ROWS;   // long sequences: more splits per pair, fewer pairs in flight per L2
    if (splits < s_l2) splits = s_l2;
    p.splits = splits < 1 ? 1 : (splits > p.nqt ? p.nqt : splits);
    int grp = ((2 * cus + p.splits - 1) / p.splits + 7) & ~7;     // one round of resident slots per group
    p.group_bh = grp > nbh ? nbh : grp;
  }
  p.kvcap = a->kv_capacity > 0 ? a->kv_capacity : a->L;
  if (p.kvcap < a->L) return AKI_ERR_INVALID_ARG;
  p.scale_log2 = a->scale * 1.44269504088896340736f;
  p.dead_uniform = a->dead_rows == AKI_DEAD_ROWS_UNIFORM;
  AKI_CLEAR_ERR();
  {
    bool long_core = a->L >= AKI_ATTN64_MIN_L;
    int exact_max = 0;
#ifdef AKI_LAB_HOOKS
    if (g_attn_variant >= 1 && g_attn_variant <= 8) long_core = false;
    if (g_attn_variant == 9 || g_attn_variant == 10 || g_attn_variant > 100) { long_core = true; exact_max = g_attn_variant == 10; }   // 100 + m: timing ablation m of the 64-row kernel
#endif
    if (long_core) {
      static int cus64 = 0;
      if (cus64 == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus64 = n;
      }
      const int rc = attn_core64_bf16_launch(p, cus64, stream, exact_max);
      if (rc != AKI_OK) return rc;
      AKI_LAUNCH_CHECK();
      return AKI_OK;
    }
  }
#ifdef AKI_LAB_HOOKS
  if (g_attn_variant == 3) hipLaunchKernelGGL((mma_attn_bf16_kernel<NW, 1>), dim3(a->B * a->H * p.splits), dim3(NW * 64), 0, stream, p);
  else if (g_attn_variant == 5) hipLaunchKernelGGL((mma_attn_bf16_kernel<NW, 3>), dim3(a->B * a->H * p.splits), dim3(NW * 64), 0, stream, p);
  else if (g_attn_variant == 8) hipLaunchKernelGGL((mma_attn_bf16_kernel<NW, 4>), dim3(a->B * a->H * p.splits), dim3(NW * 64), 0, stream, p);
  else if (g_attn_variant == 4) {
    // 8-wave ping-pong (lab): one 512-thread workgroup per CU, ranks of eight 32-row blocks
    constexpr int NW8 = 8;
    static int cus8 = 0;
    if (cus8 == 0) {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      cus8 = n;
    }
    const int nbh = a->B * a->H;
    p.nqt = (a->L + NW8 * 32 - 1) / (NW8 * 32);
    int splits = (cus8 + nbh - 1) / nbh;
    const int s_l2 = a->L / (2 * AKI_ATTN_L2_ROWS);
    if (splits < s_l2) splits = s_l2;
    p.splits = splits < 1 ? 1 : (splits > p.nqt ? p.nqt : splits);
    int grp = ((cus8 + p.splits - 1) / p.splits + 7) & ~7;
    p.group_bh = grp > nbh ? nbh : grp;
    hipLaunchKernelGGL((mma_attn_bf16_kernel<NW8, 2>), dim3(a->B * a->H * p.splits), dim3(NW8 * 64), 0, stream, p);
  } else
#endif
  hipLaunchKernelGGL((mma_attn_bf16_kernel<NW, 0>), dim3(a->B * a->H * p.splits), dim3(NW * 64), 0, stream, p);
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki
