// decode_chain.hip - the whole decoder stack of ONE decode step (batch 1) as ONE launch: a dataflow chain of
// weight-streaming workgroups (what `lang_model.generate` runs per token after the MMA prefill,
// src/aki.py:136-209 + src/aki_generation.py:36-86, HF:phi3/modeling_phi3.py:287-328 per layer).
//
// Why.  At batch 1 a decode step is 7.4 GB of weights read once; decode.hip runs it as 5 launches per layer, and every launch
// of that chain is a latency chain of its own - launch, stage x, first weight round trip, reduce, store, drain - during which
// HBM idles: the three K = 3072 GEMVs are ONE batch of loads per wave (10.7 us for 19-57 MB), 1.96 ms per token = 0.47 of the
// HBM peak.  Weights do not depend on activations.  Here a workgroup requests its weights FIRST and only then
// waits for its input vector, so the stream runs ahead of the dependency chain by everything the register files of the
// resident workgroups hold (~60 MB chip-wide), across phase AND layer seams.  1.64 ms per greedy token (0.59 of the peak).
//
// How.  grid = n_layers x (qkv | attention | o_proj | gate_up | down) workgroups in DISPATCH ORDER = dependency order.  A
// workgroup: (1) finds its (layer, phase, slice) from blockIdx, (2) requests the weights of its first CH_PF* batches (non-temporal
// loads into register slots), (3) one lane polls a READY flag of its producer phase (relaxed agent-scope loads + s_sleep; attention
// items poll the flag of THEIR HEAD's 18 qkv workgroups), (4) stages its input vector from the hand-off buffer into LDS (RMSNorm
// fused where the layer has one), (5) dot products out of registers, batch by batch, the next batch requested as a slot frees;
// wave reduce, epilogue (SwiGLU / residual), (6) publishes: outputs stored write-through (sc1), every storing wave drains
// vmcnt, barrier, one arrival on a sharded counter whose completer raises the phase's replicated flags.
// (cdna_hip_programming.md Guideline 16, R1 + counter form; every handed-off byte is stored sc1 and loaded sc1, so no acquire
// fence is needed.)
//
// Progress.  A workgroup only ever waits for workgroups with SMALLER block indices.  The dispatcher hands out workgroups of
// a grid in index order (per XCD queue), so every workgroup a resident one waits for is resident or finished: the lowest
// unfinished index is always resident with all its producers finished.  Every spin is bounded all the same: a poller that
// gives up writes its code to the error word and releases every counter of the launch (so the rest drains in microseconds);
// that step's output is garbage.  The host reads the word wherever it synchronises and at the end of a generation, switches the
// chain off for that KV cache and decodes the unverified tokens again on the five-launch path (Phi3ForCausalLM.decode_verified,
// AKI.generate; ops.DecodeChain.check raises): a broken assumption costs a warning and time, never a hung GPU or a wrong token.
// ONE CHAIN IN FLIGHT PER DEVICE: the argument needs the lowest unfinished index to be resident; two chain launches running
// side by side (two streams, two processes) can fill the CUs with each other's waiters, and then only the bounded spin ends
// them.  ops.DecodeChain.step serialises chain launches of one process across streams; processes sharing a GPU are the caller's
// to keep apart (the error word still catches it).
//
// Arithmetic = decode.hip's, row for row: one wave owns whole weight rows, lane l takes 16-byte chunks l, l+64, ... of a row
// in ascending order, the same xor-shuffle reduction, the same epilogue formulas, the same split-KV attention (same tiles,
// same merge order) - a chained step reproduces the five-launch step bit for bit (tests/test_decode_gpu.py).
#include <type_traits>

#include "aki_device.h"

namespace aki {

typedef __attribute__((ext_vector_type(8))) __bf16 chain_bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 chain_bf16x2_t;

__device__ __forceinline__ float cdot8(const u32x4 a, const u32x4 b, float acc) {
  const chain_bf16x8_t a8 = __builtin_bit_cast(chain_bf16x8_t, a), b8 = __builtin_bit_cast(chain_bf16x8_t, b);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a8, a8, 0, 1), __builtin_shufflevector(b8, b8, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a8, a8, 2, 3), __builtin_shufflevector(b8, b8, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a8, a8, 4, 5), __builtin_shufflevector(b8, b8, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a8, a8, 6, 7), __builtin_shufflevector(b8, b8, 6, 7), acc, false);
  return acc;
}
// weight-only fp8: same pairing as decode.hip's dot16_w8 (a 16-byte weight chunk = 16 k-values against two x chunks)
__device__ __forceinline__ float cdot16_w8(const u32x4 w, const u32x4 x0, const u32x4 x1, float acc) {
  const chain_bf16x8_t xa = __builtin_bit_cast(chain_bf16x8_t, x0), xb = __builtin_bit_cast(chain_bf16x8_t, x1);
#define AKI_CW8_PAIR(word, hi, xv, i0)                                                                              \
  {                                                                                                                 \
    const chain_bf16x2_t wb = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((unsigned)(word), 1.0f, hi);   /* two e4m3 -> a bf16 pair in ONE instruction, exact */ \
    acc = __builtin_amdgcn_fdot2_f32_bf16(wb, __builtin_shufflevector(xv, xv, i0, i0 + 1), acc, false);             \
  }
  AKI_CW8_PAIR(w[0], false, xa, 0) AKI_CW8_PAIR(w[0], true, xa, 2) AKI_CW8_PAIR(w[1], false, xa, 4) AKI_CW8_PAIR(w[1], true, xa, 6)
  AKI_CW8_PAIR(w[2], false, xb, 0) AKI_CW8_PAIR(w[2], true, xb, 2) AKI_CW8_PAIR(w[3], false, xb, 4) AKI_CW8_PAIR(w[3], true, xb, 6)
#undef AKI_CW8_PAIR
  return acc;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for_chain_impl(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_chain_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void static_for_chain(F&& f) { static_for_chain_impl<0, N>(f); }

#define AKI_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// pointers that arrive through the layer table are generic to the compiler: say "global" or every load is a flat_load
typedef const __attribute__((address_space(1))) u32x4* gptr_u32x4;
typedef const __attribute__((address_space(1))) unsigned* gptr_u32;
typedef const __attribute__((address_space(1))) float* gptr_f32;
#define AKI_G128(p) ((gptr_u32x4)(p))
constexpr int CH_PSTRIDE = 104;          // decode.hip's DEC_PSTRIDE: m, l, 6 pad, acc[96]
// Synchronisation block of one (layer, phase), every word in a 128-byte line of its own:
//   [0..15] arrival shards (producer workgroup i adds to shard i % 16), [16] the top counter (a completed shard adds 1),
//   [17..17+CH_FLAGS) READY flags, all written by the arriver that completes the top counter; consumer workgroup j polls flag j % CH_FLAGS.
// Why not one counter: a phase has 400-2000 producers and ~1000 resident consumers.  One word takes ~90 atomics/us
// (MI355X_MICROARCH.md, dequeue) - 2048 arrivals would cost 23 us - and a thousand lanes polling one word (agent-scope loads
// are served memory-side, not by the per-XCD L2) saturate the channel that owns it, which every weight stream crosses: the
// first version of this file, with one counter per phase, ran 3x SLOWER than five launches per layer.
constexpr int CH_SHARDS = 16;
constexpr int CH_FLAGS = 32;
constexpr int CH_SYNC_WORDS = (CH_SHARDS + 1 + CH_FLAGS) * 32;   // 128-byte lines, in 4-byte words
constexpr int CH_PHASES = 5;             // [0] qkv [1] attention (per-head mergers) [2] o_proj [3] gate_up [4] down
// feature batches per workgroup (one staging of x each) of the qkv / o_proj / gate_up / down phases
// Two each: 576 / 192 / 1024 / 384 workgroups per layer.  More batches cut the x traffic (the bare stream runs 1.41 ms per token at two,
// 1.20 ms at {8,4,16,4}) but every batch after the first is loaded AFTER the dependency wait, on the critical path of its phase:
// with the waits in, {1,1,1,1} 1.90, {2,2,2,2} 1.80, {4,4,4,4} 2.40, {8,4,16,4} 3.04 ms per token (tools/decode_chain_regimes.py).
constexpr int CH_NBQ = 2, CH_NBO = 2, CH_NBG = 4, CH_NBD = 2;
// ... of which this many are requested BEFORE the wait (register slots): with both batches of qkv / o_proj on chip when their input arrives, those phases
// take x staging + 1.2 us instead of + 4 us (tools/decode_chain_edges.py); gate_up is bandwidth-bound whatever is prefetched; 1.78 -> 1.62 ms per token.
constexpr int CH_PFQ = 2, CH_PFO = 2, CH_PFG = 2, CH_PFD = 2;
// e4m3 weights (half the bytes, 24 VALU operations per 16 weights to widen them): ONE batch per workgroup, requested before the wait - that chain is
// all dependency latency and the second batch's dot products (2 us) sat on it: {1,1,1,1} 1.37 ms per token, {2,2,2,2}/{2,2,2,2} 1.48, five launches 1.46
// (tools/decode_chain_w8.py).
constexpr int CH_TOUCH = 0;
constexpr int CH_QKV_BY_HEAD = 1;
constexpr int CH_XREP = 8;               // room for copies of every hand-off vector: consumer j reads copy j % xrep
constexpr int CH_XREP_USED = 1;          // copies in use: every copy is one more write-through store per producing lane, and at the product's
                                         // 100-600 consumers per phase one copy reads fastest (1 / 2 / 4 copies: 1.66 / 1.68 / 1.71 ms per token)
constexpr unsigned CH_SPIN_LIMIT = 200000u;
// the batched chain's ring depths (steps of 64 k per wave held in registers; see chain_gemm): qkv, o_proj, gate_up (two streams), down (+ its x ring)
constexpr int CH_B_RQ = 6, CH_B_RO = 6, CH_B_RG = 4, CH_B_RD = 6;         // <= 128 VGPRs: two 512-thread workgroups per CU

struct ChainParams {
  const aki_decode_chain_layer* layers;
  int n_layers;
  const bf16_t* h_in; bf16_t* h_out;
  const float* cos; const float* sin; const int* cache_len; const uint64_t* vbits; int nwords;
  int d, H, F, cap, S, T; float scale, eps;
  int B;                  // sequences (rows) of the step: 1 for decode_chain_kernel, 2..8 for decode_chain_b_kernel (hand-off vectors are [B][n] then)
  unsigned* sync;         // [n_layers][CH_PHASES][CH_SYNC_WORDS], then [n_layers][H] attention tickets: all zero when the call starts (two sets)
  unsigned* attn_cnt;     // a ticket per (layer, head): nothing is re-armed inside the launch
  unsigned* head_sync;    // [n_layers][H][2 lines]: arrivals of the 96 / rows-per-workgroup x 3 qkv workgroups that produce head h's q, k, v and
                          // the head's READY flag (qkv_by_head): an attention item waits for ITS head's 18 producers, not for the phase's 576
  int qkv_by_head;        // wgs per (section, head) of the qkv phase when its workgroups are laid out head-major, else 0 (one flag for the phase)
  unsigned* err;          // sticky error word (outside the counter sets)
  unsigned* epoch;        // calls completed on this workspace: its parity says which of the TWO counter sets this call uses (see the kernel)
  int cnt_words;          // 4-byte words of one counter set; sync / head_sync / attn_cnt point into set 0
  bf16_t* qkv; bf16_t* attn_o; bf16_t* h1; bf16_t* act; bf16_t* hbuf;   // hand-off vectors (copy 0); hbuf = 2 x d
  int rep_stride;         // elements between the copies of a hand-off vector
  float* part;            // [H][S][CH_PSTRIDE]
  int n_qkv, n_attn, n_o, n_gu, n_down, wg_layer;
  int sleep_n, xrep, nflags, nowait;   // product: 8, CH_XREP_USED, CH_FLAGS, 0; the lab library can change them (aki_lab_set_chain)
  int nbq, nbo, nbg, nbd;              // batches of 4 x FPW features per workgroup of the qkv / o_proj / gate_up / down phases
  int touch;                           // 1: a waiting workgroup pulls the batches it holds no registers for towards L2 / the Infinity Cache
#ifdef AKI_LAB_HOOKS
  unsigned long long* stamps;          // lab: [phase 0..4][workgroup < 2048][8] wall-clock stamps (100 MHz) of layer `stamp_layer`
  int stamp_layer;
  unsigned fault_code;                 // lab: the wait with this code (layer << 8 | phase) gives up at once (fault injection for the host's recovery path)
#endif
};

#ifdef AKI_LAB_HOOKS
#define AKI_CHAIN_STAMP(p, layer, phase, wg, k)                                                                   \
  do {                                                                                                            \
    if ((p).stamps && (layer) == (p).stamp_layer && threadIdx.x == 0 && (wg) < 2048)                              \
      (p).stamps[((size_t)(phase) * 2048 + (wg)) * 8 + (k)] = wall_clock64();                                     \
  } while (0)
#else
#define AKI_CHAIN_STAMP(p, layer, phase, wg, k) do { } while (0)
#endif

// ---- hand-off primitives ------------------------------------------------------------------------------------------------
// Vector loads of handed-off bytes: buffer_load_dwordx4 ... sc1 (aux 16), 16-byte aligned offsets.
__device__ __forceinline__ u32x4 ld_sc1_b128(const __amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, bytes, 0x00020000);
}

// One lane polls this workgroup's READY flag of the producer phase; bounded.  On give-up: error word, then every flag of the
// launch is raised (so the rest drains in microseconds).
__device__ __forceinline__ void chain_wait(const ChainParams& p, unsigned* sync, int wg, unsigned code) {
  if (threadIdx.x == 0 && sync != nullptr && !p.nowait) {
    unsigned* flag = sync + (CH_SHARDS + 1 + (wg % p.nflags)) * 32;
    unsigned spins = 0;
    auto give_up = [&]() {
      __hip_atomic_store(p.err, code, AKI_RLX_AGENT);
      for (int i = 0; i < p.n_layers * CH_PHASES; ++i)
        for (int f = 0; f < CH_FLAGS; ++f) __hip_atomic_store(p.sync + (size_t)i * CH_SYNC_WORDS + (CH_SHARDS + 1 + f) * 32, 2u, AKI_RLX_AGENT);
    };
#ifdef AKI_LAB_HOOKS
    if (p.fault_code != 0u && code == p.fault_code) give_up();
    else
#endif
    while (__hip_atomic_load(flag, AKI_RLX_AGENT) == 0u) {
      for (int i = 0; i < p.sleep_n; ++i) __builtin_amdgcn_s_sleep(1);
      if (++spins > CH_SPIN_LIMIT) { give_up(); break; }
    }
  }
  __syncthreads();
}

// Arrival of producer `idx` of `n` (wave 0, after the outputs are at the coherence point): its shard; the arriver that completes
// a shard adds to the top counter; the one that completes the top counter raises every READY flag (32 lanes, one store each).
__device__ __forceinline__ unsigned chain_arrive(unsigned* sync, int idx, int n, int lane) {
  unsigned done = 0;
  if (lane == 0) {
    if (n <= 64) {     // few producers (the 32 head mergers): straight to the top counter - one memory round trip less on the edge
      done = (__hip_atomic_fetch_add(sync + CH_SHARDS * 32, 1u, AKI_RLX_AGENT) + 1u == (unsigned)n) ? 1u : 0u;
    } else {
      const int shard = idx % CH_SHARDS;
      const unsigned target = (unsigned)(n / CH_SHARDS + ((n % CH_SHARDS) > shard ? 1 : 0));
      const unsigned prev = __hip_atomic_fetch_add(sync + shard * 32, 1u, AKI_RLX_AGENT);
      if (prev + 1u == target) done = (__hip_atomic_fetch_add(sync + CH_SHARDS * 32, 1u, AKI_RLX_AGENT) + 1u == (unsigned)CH_SHARDS) ? 1u : 0u;
    }
  }
  done = __shfl(done, 0);
  if (done && lane < CH_FLAGS) __hip_atomic_store(sync + (CH_SHARDS + 1 + lane) * 32, 1u, AKI_RLX_AGENT);
  return done;             // 1 in the wave that completed the phase
}

// Every storing wave has drained (vmcnt(0)) before the barrier; wave 0 signals.
__device__ __forceinline__ void chain_publish(unsigned* sync, int idx, int n) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) chain_arrive(sync, idx, n, threadIdx.x);
}

// ---- one GEMV phase ----------------------------------------------------------------------------------------------------
// NR weight rows per wave and batch, KC 16-byte chunks per lane and row (bf16: K = 512 KC; W8: K = 1024 KC).  SWIGLU: the wave's
// rows are gate rows f.. and up rows n_out + f.. (NR/2 features).  NORM: x is RMS-normalised (weight norm_w) on its way into LDS.
// A workgroup takes `nb` batches of 4 x FPW consecutive features against ONE staging of x: every workgroup reads its whole input
// vector through the fabric (sc1), and at one batch per workgroup those reads were 15 % of all bytes moved - the weight stream
// ran at exactly 6.3 TB/s / 1.15 with the dependency waits switched off (tools/decode_chain_ab.py).  Batch 0 is loaded before the
// wait; batch b+1 as soon as the dot products have released the registers of batch b, under its reduction and epilogue.
template <int NR, int KC, bool SWIGLU, bool NORM, bool W8, int NB, int PF>
__device__ __forceinline__ unsigned chain_gemv(const ChainParams& p, int wg, int n_wg, const void* w, const float* w_scale, int K, int n_out,
                                           const bf16_t* x, int x_rep, const bf16_t* norm_w, const bf16_t* residual, int res_rep, bf16_t* y,
                                           int y_reps, unsigned* wait_sync, unsigned* done_sync, unsigned code, char* sx, float* s_red,
                                           int head_per = 0, unsigned* head_sync = nullptr) {
  constexpr int FPW = SWIGLU ? NR / 2 : NR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int nb = NB;
  // qkv phase, head-major (head_per = workgroups per 96 rows): workgroup -> (head, section q / k / v, slice) so that the producers of one
  // head's q, k and v are 3 x head_per consecutive workgroups with an arrival counter of their own
  int fbase = wg * (4 * FPW * nb) + wave * FPW;
  int head_of_wg = 0;
  if (head_per > 0) {
    head_of_wg = wg / (3 * head_per);
    const int part = wg - head_of_wg * 3 * head_per, sec = part / head_per, sub = part - sec * head_per;
    fbase = sec * (n_out / 3) + head_of_wg * 96 + sub * (4 * FPW * nb) + wave * FPW;
  }
  const size_t row_bytes = W8 ? (size_t)K : (size_t)K * 2;
  // PF batches are requested before the wait (register slots b % PF); with PF = NB nothing is left to load once the input is there
  static_assert(PF >= 1 && PF <= NB, "prefetch depth");
  u32x4 wv[PF][NR][KC];
  float wsc[PF][NR];
  auto issue = [&](int f0, auto slot_c) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int f = min(f0 + (r % FPW), n_out - 1);
      const int row = (SWIGLU && r >= FPW) ? n_out + f : f;
      const char* wr = (const char*)w + (size_t)row * row_bytes;
      wsc[SL][r] = W8 ? ((gptr_f32)w_scale)[row] : 1.f;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) wv[SL][r][kc] = __builtin_nontemporal_load(AKI_G128(wr + (size_t)(lane + 64 * kc) * 16));
    }
  };
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 0);
  // (2) the weight loads of this wave's first PF batches, before anything that depends on another workgroup
  static_for_chain<PF>([&](auto b_c) { issue(fbase + decltype(b_c)::value * 4 * FPW, b_c); });
  // (2b) the batches beyond the register slots: one dword per 128-byte line, default cache policy, result unused - the lines travel
  // HBM -> Infinity Cache -> this XCD's L2 while the workgroup waits, and the real (nt) loads after the wait find them on the die
  unsigned touched = 0;
  if constexpr (PF < NB) {
    if (p.touch) {
#pragma unroll
      for (int b = PF; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int f = min(fbase + b * 4 * FPW + (r % FPW), n_out - 1);
          const int row = (SWIGLU && r >= FPW) ? n_out + f : f;
          const char* wr = (const char*)w + (size_t)row * row_bytes;
#pragma unroll
          for (int o = 0; o < KC * 1024; o += 64 * 128)
            if (o + lane * 128 < KC * 1024) touched |= *(volatile const __attribute__((address_space(1))) unsigned*)(wr + o + lane * 128);
        }
    }
  }
  // (3) the producer phase has published
  chain_wait(p, wait_sync, wg, code);
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 1);
  // (4) x -> LDS.  Handed-off bytes: sc1 loads only, from this workgroup's copy of the vector.
  const int nchunk = K / 8;
  x += (size_t)(x_rep ? (wg % p.xrep) * p.rep_stride : 0);
  if (residual != nullptr && res_rep) residual += (size_t)(wg % p.xrep) * p.rep_stride;
  const __amdgpu_buffer_rsrc_t rx = chain_rsrc(x, K * 2);
  if constexpr (NORM) {
    // y = bf16(x * rsqrt(mean(x^2) + eps) * w): decode.hip's gemv_bf16_kernel staging, the row kept in registers between the passes
    u32x4 xv[2];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i;
      xv[i] = u32x4{0u, 0u, 0u, 0u};
      if (c < nchunk) xv[i] = ld_sc1_b128(rx, c * 16);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf16_lo(xv[i][e]), hi = bf16_hi(xv[i][e]);
        ss = __builtin_fmaf(lo, lo, ss);
        ss = __builtin_fmaf(hi, hi, ss);
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) s_red[wave] = ss;
    __syncthreads();
    const float rs = rsqrtf((s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)K + p.eps);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i;
      if (c < nchunk) {
        const u32x4 g = *AKI_G128(norm_w + c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_bf16x2(round_bf16(bf16_lo(xv[i][e]) * rs) * bf16_lo(g[e]), round_bf16(bf16_hi(xv[i][e]) * rs) * bf16_hi(g[e]));
        *(u32x4*)(sx + (size_t)c * 16) = o;
      }
    }
  } else {
    for (int c = tid; c < nchunk; c += 256) *(u32x4*)(sx + (size_t)c * 16) = ld_sc1_b128(rx, c * 16);
  }
  __syncthreads();
  asm volatile("" :: "v"(touched));    // the touch loads are older than the x loads above: nothing waits here
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 2);
  static_for_chain<NB>([&](auto b_c) {
    constexpr int b = decltype(b_c)::value;
    constexpr int SL = b % PF;
    const int f0 = fbase + b * 4 * FPW;
    unsigned long long res_bits = 0;
    if (residual != nullptr && lane == 0 && f0 < n_out) {            // in flight under the dot products
      if constexpr (FPW == 4) res_bits = __hip_atomic_load((const unsigned long long*)(residual + f0), AKI_RLX_AGENT);
      else if constexpr (FPW == 2) res_bits = __hip_atomic_load((const unsigned*)(residual + f0), AKI_RLX_AGENT);
      else res_bits = __hip_atomic_load((const unsigned short*)(residual + f0), AKI_RLX_AGENT);
    }
    // (5) dot products out of the registers, chunk order ascending per lane and row
    float acc[NR];
    float sc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { acc[r] = 0.f; sc[r] = wsc[SL][r]; }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int c = lane + 64 * kc;
      if constexpr (W8) {
        const u32x4 x0 = *(const u32x4*)(sx + (size_t)(2 * c) * 16), x1 = *(const u32x4*)(sx + (size_t)(2 * c + 1) * 16);
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = cdot16_w8(wv[SL][r][kc], x0, x1, acc[r]);
      } else {
        const u32x4 xc = *(const u32x4*)(sx + (size_t)c * 16);
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = cdot8(wv[SL][r][kc], xc, acc[r]);
      }
    }
    // the next batch's loads, as soon as the registers are free (the barrier keeps the scheduler from renaming them upwards)
    asm volatile("" : "+v"(acc[0]));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (b + PF < NB) issue(f0 + PF * 4 * FPW, std::integral_constant<int, SL>{});
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      float v = acc[r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      acc[r] = W8 ? v * sc[r] : v;
    }
    // (6) epilogue
    if (lane == 0 && f0 < n_out) {
      float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int f = 0; f < FPW; ++f) {
        float v = SWIGLU ? acc[FPW + f] * silu_fast(acc[f]) : acc[f];
        if (residual != nullptr) v += bf16_bits_to_f32((unsigned short)(res_bits >> (16 * f)));
        out[f] = v;
      }
      // n_out is a multiple of FPW on every matrix of the stack: a wave's features are all in range or none is
      for (int rep = 0; rep < min(y_reps, p.xrep); ++rep) {
        bf16_t* yr = y + (size_t)rep * p.rep_stride + f0;
        if constexpr (FPW == 4) {
          const unsigned long long o = (unsigned long long)pack_bf16x2(out[0], out[1]) | ((unsigned long long)pack_bf16x2(out[2], out[3]) << 32);
          __hip_atomic_store((unsigned long long*)yr, o, AKI_RLX_AGENT);
        } else if constexpr (FPW == 2) {
          __hip_atomic_store((unsigned*)yr, pack_bf16x2(out[0], out[1]), AKI_RLX_AGENT);
        } else {
          __hip_atomic_store((unsigned short*)yr, (unsigned short)(pack_bf16x2(out[0], 0.f) & 0xffffu), AKI_RLX_AGENT);
        }
      }
    }
  });
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chain_publish, with a stamp between the drain and the arrival
  __syncthreads();
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 4);
  unsigned completed = 0;
  if (head_per > 0) {             // one counter per head: 3 x head_per arrivals, the last one raises the head's flag
    if (threadIdx.x == 0) {
      unsigned* hs = head_sync + (size_t)head_of_wg * 64;
      if (__hip_atomic_fetch_add(hs, 1u, AKI_RLX_AGENT) + 1u == (unsigned)(3 * head_per)) __hip_atomic_store(hs + 32, 1u, AKI_RLX_AGENT);
    }
  } else if (threadIdx.x < 64) {
    completed = chain_arrive(done_sync, wg, n_wg, threadIdx.x);
  }
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 5);
  return completed;        // wave 0: 1 when this workgroup's arrival completed the phase
}

// ---- one GEMM phase of the BATCHED chain (2..8 sequences; 512-thread workgroups) ---------------------------------------------------
// = decode.hip's skinny_gemm_bf16_kernel<8, SWIGLU, 1, NORM> inside the dataflow launch: a workgroup is one 16-feature tile of
// v_mfma_f32_16x16x32_bf16 (the rows of the step ride on the tile's columns), its EIGHT waves split K and the partial tiles are added in wave
// order by wave 0 - the same K order, the same fold order, the same epilogue: a chained step reproduces the five-launch batched step bit for
// bit.  Lane (row l15 of the tile, k-group kg) streams 32 contiguous bytes of ITS weight row per step of 64 k (the four k-groups of a row read
// one 128-byte line).  A wave's KSL steps run through a register ring of R steps: the first R are requested BEFORE the dependency wait, step
// i + R as soon as the MFMAs of step i have released its slot (R = KSL: everything is on chip when the flag rises).
// XMODE 0: the input rows are RMS-normalised into LDS (wave m takes row m: the skinny kernel's prologue - HF's rounding points, the same
//          summation order) and the B fragments are read from there;  1: the rows are copied into LDS as they are (K = 3072: 49 KB for eight);
//          2: K = 8192 does not fit - the B fragments come through their own register ring of sc1 loads, R steps ahead.
// First version (four waves, two K slices each, rings of 10 / 12 / 4 / 8 steps, 167 VGPRs): 104 us per layer at batch 8 = the five launches
// (tools/decode_chain_edges.py --batch 8: down 20 us of ring refills, gate_up 7-15, x staging 3.4-4.8 us per phase).
template <int KSL, bool SWIGLU, int XMODE, int R>
__device__ __forceinline__ unsigned chain_gemm(const ChainParams& p, int wg, int n_wg, const void* w, int n_out, const bf16_t* x,
                                               const bf16_t* norm_w, const bf16_t* residual, bf16_t* y, unsigned* wait_sync, unsigned* done_sync,
                                               unsigned code, char* smem, int head_per = 0, unsigned* head_sync = nullptr) {
  constexpr int K = KSL * 512, NS = SWIGLU ? 2 : 1;                    // K = 8 waves x KSL steps of 64
  static_assert(R >= 1 && R <= KSL, "ring depth");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
  const int M = p.B;
  int f0 = wg * 16, head_of_wg = 0;
  if (head_per > 0) {            // qkv, head-major: the 3 x head_per tiles of one head's q, k and v are consecutive workgroups with an arrival counter of their own
    head_of_wg = wg / (3 * head_per);
    const int part = wg - head_of_wg * 3 * head_per, sec = part / head_per, sub = part - sec * head_per;
    f0 = sec * (n_out / 3) + head_of_wg * 96 + sub * 16;
  }
  const int frow = min(f0 + l15, n_out - 1);
  const int kw = wave * (KSL * 64) + 16 * kg;                          // this lane's first element of the wave's K slice
  const bf16_t* w0 = (const bf16_t*)w + (size_t)frow * K + kw;
  const bf16_t* w1 = (const bf16_t*)w + (size_t)(n_out + frow) * K + kw;      // SwiGLU: the up row of the gate row
  u32x4 wr[R][NS][2];
  auto issue_w = [&](int i, auto slot_c) {
    constexpr int SL = decltype(slot_c)::value;
    wr[SL][0][0] = __builtin_nontemporal_load(AKI_G128(w0 + i * 64));
    wr[SL][0][1] = __builtin_nontemporal_load(AKI_G128(w0 + i * 64 + 8));
    if constexpr (SWIGLU) {
      wr[SL][1][0] = __builtin_nontemporal_load(AKI_G128(w1 + i * 64));
      wr[SL][1][1] = __builtin_nontemporal_load(AKI_G128(w1 + i * 64 + 8));
    }
  };
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 0);
  // (2) the weights of the first R steps before anything that depends on another workgroup
  static_for_chain<R>([&](auto i_c) { issue_w(decltype(i_c)::value, i_c); });
  // (3) the producer phase has published
  chain_wait(p, wait_sync, wg, code);
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 1);
  // (4) the input rows
  constexpr int PITCH = K * 2 + 16;                       // LDS row pitch (XMODE 0 / 1): the pad spreads the 16 rows of a fragment read over the banks
  const int xrow = min(l15, M - 1);
  if constexpr (XMODE == 0) {
    if (wave < M) {                                       // wave m normalises row m: the skinny kernel's prologue, chunk for chunk
      const int m = wave;
      const __amdgpu_buffer_rsrc_t rx = chain_rsrc(x + (size_t)m * K, K * 2);
      u32x4 v[KSL], gch[KSL];
#pragma unroll
      for (int i = 0; i < KSL; ++i) v[i] = ld_sc1_b128(rx, (lane + 64 * i) * 16);
#pragma unroll
      for (int i = 0; i < KSL; ++i) gch[i] = *AKI_G128(norm_w + (size_t)(lane + 64 * i) * 8);      // the gain: an L2 hit, in flight with the row
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < KSL; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bf16_lo(v[i][e]), hi = bf16_hi(v[i][e]);
          ss = __builtin_fmaf(lo, lo, ss);
          ss = __builtin_fmaf(hi, hi, ss);
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
      const float r = rsqrtf(ss / (float)K + p.eps);
      char* dst = smem + (size_t)m * PITCH;
#pragma unroll
      for (int i = 0; i < KSL; ++i) {
        __builtin_amdgcn_sched_barrier(0);                // chunk by chunk: the widened floats of one chunk at a time (this prologue sets the kernel's VGPR count)
        u32x4 gi = gch[i], vi = v[i], o;
        asm volatile("" : "+v"(gi), "+v"(vi));            // ... and the gain and the row are widened HERE, not kept as 96 floats from the moment they arrived
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_bf16x2(round_bf16(bf16_lo(vi[e]) * r) * bf16_lo(gi[e]), round_bf16(bf16_hi(vi[e]) * r) * bf16_hi(gi[e]));
        *(u32x4*)(dst + (size_t)(lane + 64 * i) * 16) = o;
      }
    }
    __syncthreads();
  } else if constexpr (XMODE == 1) {
    const __amdgpu_buffer_rsrc_t rx = chain_rsrc(x, M * K * 2);
    for (int c = tid; c < M * (K / 8); c += 512) {
      const int m = c / (K / 8), cc = c - m * (K / 8);
      *(u32x4*)(smem + (size_t)m * PITCH + (size_t)cc * 16) = ld_sc1_b128(rx, c * 16);
    }
    __syncthreads();
  }
  const char* xs = smem + (size_t)xrow * PITCH + (size_t)kw * 2;                        // XMODE 0 / 1
  const __amdgpu_buffer_rsrc_t rxg = chain_rsrc(x + (size_t)xrow * K, K * 2);           // XMODE 2
  u32x4 xr[XMODE == 2 ? R : 1][2];
  auto issue_x = [&](int i, auto slot_c) {
    constexpr int SL = decltype(slot_c)::value;
    xr[SL][0] = ld_sc1_b128(rxg, (kw + i * 64) * 2);
    xr[SL][1] = ld_sc1_b128(rxg, (kw + i * 64) * 2 + 16);
  };
  if constexpr (XMODE == 2) static_for_chain<R>([&](auto i_c) { issue_x(decltype(i_c)::value, i_c); });
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 2);
  // (5) the steps, ring slot i % R
  f32x4 acc[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  static_for_chain<KSL>([&](auto i_c) {
    constexpr int i = decltype(i_c)::value, SL = i % R;
    u32x4 x2[2];
    if constexpr (XMODE == 2) {
      x2[0] = xr[SL][0];
      x2[1] = xr[SL][1];
    } else {
      x2[0] = *(const u32x4*)(xs + (size_t)i * 128);
      x2[1] = *(const u32x4*)(xs + (size_t)i * 128 + 16);
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const chain_bf16x8_t xb = __builtin_bit_cast(chain_bf16x8_t, x2[hh]);
#pragma unroll
      for (int t = 0; t < NS; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(chain_bf16x8_t, wr[SL][t][hh]), xb, acc[t], 0, 0, 0);
    }
    if constexpr (i + R < KSL) {
      // the slot is free once the MFMAs above have read it; the barriers keep the scheduler from renaming the ring into KSL live steps
      asm volatile("" : "+v"(acc[0]));
      __builtin_amdgcn_sched_barrier(0);
      issue_w(i + R, std::integral_constant<int, SL>{});
      if constexpr (XMODE == 2) issue_x(i + R, std::integral_constant<int, SL>{});
      __builtin_amdgcn_sched_barrier(0);
    }
  });
  // (6) the eight partial tiles meet in LDS (over the rows: every wave is done with them), wave 0 adds them in wave order and runs the epilogue
  float (*red)[NS][256] = (float (*)[NS][256])smem;
  if constexpr (XMODE != 2) __syncthreads();
#pragma unroll
  for (int t = 0; t < NS; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][t][lane * 4 + r] = acc[t][r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int sw = 1; sw < 8; ++sw)
#pragma unroll
      for (int t = 0; t < NS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += red[sw][t][lane * 4 + r];
    const int tok = l15, f = f0 + 4 * kg;
    if (tok < M && f < n_out) {
      unsigned long long res_bits = 0;
      if (residual != nullptr) res_bits = __hip_atomic_load((const unsigned long long*)(residual + (size_t)tok * n_out + f), AKI_RLX_AGENT);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = SWIGLU ? acc[NS - 1][r] * silu_fast(acc[0][r]) : acc[0][r];
        if (residual != nullptr) v[r] += bf16_bits_to_f32((unsigned short)(res_bits >> (16 * r)));
      }
      const unsigned long long o = (unsigned long long)pack_bf16x2(v[0], v[1]) | ((unsigned long long)pack_bf16x2(v[2], v[3]) << 32);
      __hip_atomic_store((unsigned long long*)(y + (size_t)tok * n_out + f), o, AKI_RLX_AGENT);
    }
  }
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 4);
  unsigned completed = 0;
  if (head_per > 0) {
    if (threadIdx.x == 0) {
      unsigned* hs = head_sync + (size_t)head_of_wg * 64;
      if (__hip_atomic_fetch_add(hs, 1u, AKI_RLX_AGENT) + 1u == (unsigned)(3 * head_per)) __hip_atomic_store(hs + 32, 1u, AKI_RLX_AGENT);
    }
  } else if (threadIdx.x < 64) {
    completed = chain_arrive(done_sync, wg, n_wg, threadIdx.x);
  }
  AKI_CHAIN_STAMP(p, (int)(code >> 8), (int)(code & 255) - 1, wg, 5);
  return completed;
}

// ---- the attention phase: decode.hip's decode_attn_split_kernel<true>, one (head, split) item per WAVE ---------------------
// qkv (un-rotated, handed off by phase 0) -> rotated q; the item whose key range holds the new position also rotates k, appends
// k / v to the cache and uses them from LDS.  Partials (m, l, acc[96]) meet in the workspace; the item that arrives last at its
// head's ticket merges, stores the head's 96 outputs write-through and adds 1 to the layer's attention counter.
// BT (the batched chain): an item is (sequence b, head, split); the K/V caches are [B][H][cap][96], qkv / attn_o are [B][n] rows, lengths,
// valid bits, tickets and partials are per sequence.  BT = false compiles to the one-sequence code unchanged (b = 0).
template <bool BT = false>
__device__ __forceinline__ void chain_attn(const ChainParams& p, const aki_decode_chain_layer& ly, int layer, int wg, unsigned* wait_sync,
                                           unsigned* done_sync, unsigned code, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // per-wave LDS: s_q, s_k, s_v (96 bf16 each, 16-byte aligned) and the merge scratch s_mg[5][12][10] floats
  bf16_t* s_q = (bf16_t*)(smem + wave * 3072);
  bf16_t* s_k = s_q + 96;
  bf16_t* s_v = s_k + 96;
  float* s_mg = (float*)(smem + wave * 3072 + 576);
  const int item = wg * (BT ? 8 : 4) + wave;             // the batched chain runs 512-thread workgroups: eight items each
  const int nB = BT ? p.B : 1;
  const bool live = item < nB * p.H * p.S;
  const int bh = live ? item / p.S : 0, split = live ? item - bh * p.S : 0;        // bh = b * H + h
  const int b = BT ? bh / p.H : 0, h = BT ? bh - b * p.H : bh;
  const int ln = p.cache_len[b];
  const int n = ln + 1;
  const int k_begin = split * p.T * 64;
  const int k_end = live ? min(n, k_begin + p.T * 64) : 0;
  bf16_t* kb = (bf16_t*)ly.k_cache + (size_t)bh * p.cap * 96;
  bf16_t* vb = (bf16_t*)ly.v_cache + (size_t)bh * p.cap * 96;
  float* part = p.part + ((size_t)bh * p.S + split) * CH_PSTRIDE;
  const int g = lane >> 4, i16 = lane & 15;
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const bool work = k_begin < k_end;
  const bool owner = work && ln >= k_begin;              // ln < k_end by construction
  // Register budget: the whole launch runs at this phase's VGPR count, and the depth of the weight prefetch is what the
  // resident workgroups' registers hold - so only the K tile (48 VGPRs) waits in registers across the dependency.  The V
  // tile is TOUCHED ahead of it (one dword per 128-byte line: the tile's 64 rows are 12 KiB contiguous, the lines then sit in
  // this XCD's L2) and loaded for real once the scores have released the K registers; q is re-read from LDS (a broadcast).
  u32x4 kr[12], vr[16];
  auto issue_k = [&](int base) {
    const bf16_t* krow = kb + (size_t)min(base + lane, k_end - 1) * 96;
#pragma unroll
    for (int i = 0; i < 12; ++i) kr[i] = *AKI_G128(krow + i * 8);
  };
  auto issue_v = [&](int base) {
#pragma unroll
    for (int t2 = 0; t2 < 16; ++t2) {
      const int r = min(base + 4 * t2 + g, k_end - 1);
      vr[t2] = *AKI_G128(vb + (size_t)r * 96 + min(i16, 11) * 8);
    }
  };
  unsigned touch0 = 0, touch1 = 0;
  AKI_CHAIN_STAMP(p, layer, 1, wg, 0);
  if (work) {                                            // the cache does not depend on this token's qkv: in flight under the wait
    issue_k(k_begin);
    const int rows = min(64, k_end - k_begin);
    const char* vt = (const char*)(vb + (size_t)k_begin * 96);
    if (lane * 128 < rows * 192) touch0 = *(gptr_u32)(vt + lane * 128);
    if ((lane + 64) * 128 < rows * 192) touch1 = *(gptr_u32)(vt + (lane + 64) * 128);
  }
  if (p.qkv_by_head && !p.nowait) {
    // per wave: the flag of THIS item's head (lane 0 polls; a wave is one item and nothing below is a workgroup barrier)
    if (live && lane == 0) {
      const unsigned* flag = p.head_sync + ((size_t)layer * p.H + h) * 64 + 32;
      unsigned spins = 0;
      while (__hip_atomic_load(flag, AKI_RLX_AGENT) == 0u) {
        for (int i = 0; i < p.sleep_n; ++i) __builtin_amdgcn_s_sleep(1);
        if (++spins > CH_SPIN_LIMIT) { __hip_atomic_store(p.err, code, AKI_RLX_AGENT); break; }
        if ((spins & 63u) == 0u && __hip_atomic_load(p.err, AKI_RLX_AGENT) != 0u) break;      // somebody gave up: drain
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  } else {
    chain_wait(p, wait_sync, wg, code);
  }
  AKI_CHAIN_STAMP(p, layer, 1, wg, 1);
  if (!live) return;                                     // no workgroup barrier below this line
  if (work) {
    if (lane < 48) {
      const bf16_t* row = BT ? p.qkv + (size_t)b * 3 * p.H * 96 + h * 96 : p.qkv + (size_t)(item % p.xrep) * p.rep_stride + h * 96;
      const float c0 = p.cos[(size_t)ln * 96 + lane], c1 = p.cos[(size_t)ln * 96 + lane + 48];
      const float s0 = p.sin[(size_t)ln * 96 + lane], s1 = p.sin[(size_t)ln * 96 + lane + 48];
      const float q0 = bf16_bits_to_f32(__hip_atomic_load(row + lane, AKI_RLX_AGENT));
      const float q1 = bf16_bits_to_f32(__hip_atomic_load(row + lane + 48, AKI_RLX_AGENT));
      ((__bf16*)s_q)[lane] = (__bf16)(q0 * c0 - q1 * s0);
      ((__bf16*)s_q)[lane + 48] = (__bf16)(q1 * c1 + q0 * s1);
      if (owner) {
        const bf16_t* krw = row + p.H * 96;
        const bf16_t* vrw = row + 2 * p.H * 96;
        const float k0 = bf16_bits_to_f32(__hip_atomic_load(krw + lane, AKI_RLX_AGENT));
        const float k1 = bf16_bits_to_f32(__hip_atomic_load(krw + lane + 48, AKI_RLX_AGENT));
        const unsigned short v0 = __hip_atomic_load(vrw + lane, AKI_RLX_AGENT), v1 = __hip_atomic_load(vrw + lane + 48, AKI_RLX_AGENT);
        const __bf16 kn0 = (__bf16)(k0 * c0 - k1 * s0), kn1 = (__bf16)(k1 * c1 + k0 * s1);
        ((__bf16*)s_k)[lane] = kn0;
        ((__bf16*)s_k)[lane + 48] = kn1;
        ((__bf16*)kb)[(size_t)ln * 96 + lane] = kn0;
        ((__bf16*)kb)[(size_t)ln * 96 + lane + 48] = kn1;
        s_v[lane] = v0;
        s_v[lane + 48] = v1;
        vb[(size_t)ln * 96 + lane] = v0;
        vb[(size_t)ln * 96 + lane + 48] = v1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its LDS operations complete in order
    __builtin_amdgcn_wave_barrier();
    AKI_CHAIN_STAMP(p, layer, 1, wg, 2);
    // the touches are consumed here (never true): the loads stay, their wait falls where the tile's lines are needed anyway
    if (touch0 == 0x7fc0dead && touch1 == 0x7fc0beef && p.scale < 0.f) l = 1.f;
    for (int t = 0; t < p.T; ++t) {
      const int base = k_begin + t * 64;
      if (base >= k_end) break;
      const int j = base + lane;
      if (t > 0) issue_k(base);
      const bool fresh = owner && base <= ln && ln < base + 64;
      // The new token's row lives in LDS: the tile loads were issued before it was stored, so every lane whose (clamped) row
      // index is ln - the row itself and all rows past k_end - 1 = ln, which carry probability 0 - takes the row from LDS
      // instead of stale cache contents (0 * NaN would poison the sum).
      if (fresh && min(j, k_end - 1) == ln) {
#pragma unroll
        for (int i = 0; i < 12; ++i) kr[i] = *(const u32x4*)(s_k + i * 8);
      }
      bool ok = j < k_end;
      if (p.vbits && (base >> 6) < p.nwords) ok = ok && ((p.vbits[(BT ? (size_t)b * p.nwords : 0) + (base >> 6)] >> lane) & 1ull);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 12; ++i) s = cdot8(kr[i], *(const u32x4*)(s_q + i * 8), s);
      asm volatile("" : "+v"(s));
      __builtin_amdgcn_sched_barrier(0);
      issue_v(base);                                     // the K registers are free now; the lines were touched before the wait
      s = ok ? s * p.scale : -INFINITY;
      const float mn = fmaxf(m, wave_max(s));
      if (mn == -INFINITY) continue;
      const float a = __expf(m - mn);
      const float pr = ok ? __expf(s - mn) : 0.f;
      l = l * a + wave_sum(pr);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= a;
      if (fresh) {
#pragma unroll
        for (int t2 = 0; t2 < 16; ++t2)
          if (min(base + 4 * t2 + g, k_end - 1) == ln) vr[t2] = *(const u32x4*)(s_v + min(i16, 11) * 8);
      }
#pragma unroll
      for (int t2 = 0; t2 < 16; ++t2) {
        const float w = __shfl(pr, 4 * t2 + g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] = __builtin_fmaf(w, bf16_lo(vr[t2][e]), acc[2 * e]);
          acc[2 * e + 1] = __builtin_fmaf(w, bf16_hi(vr[t2][e]), acc[2 * e + 1]);
        }
      }
      m = mn;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      acc[e] += __shfl_xor(acc[e], 16);
      acc[e] += __shfl_xor(acc[e], 32);
    }
  }
  AKI_CHAIN_STAMP(p, layer, 1, wg, 3);
  if (lane == 0) { __hip_atomic_store(part, m, AKI_RLX_AGENT); __hip_atomic_store(part + 1, l, AKI_RLX_AGENT); }
  if (lane < 12) {
#pragma unroll
    for (int e = 0; e < 8; ++e) __hip_atomic_store(part + 8 + lane * 8 + e, acc[e], AKI_RLX_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned prev = 0;
  if (lane == 0) prev = __hip_atomic_fetch_add(p.attn_cnt + (size_t)layer * nB * p.H + bh, 1u, AKI_RLX_AGENT);
  prev = __shfl(prev, 0);
  AKI_CHAIN_STAMP(p, layer, 1, wg, 4);
  if (prev != (unsigned)(p.S - 1)) return;
  asm volatile("" ::: "memory");
  // the last arriver of head h merges the S partials (decode.hip's order: five split slots per pass, the slots then meet in LDS)
  const float* pp = p.part + (size_t)bh * p.S * CH_PSTRIDE;
  const int sl = lane / 12, ch = lane - sl * 12;
  float gm = -INFINITY, lt = 0.f, o8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o8[e] = 0.f;
  if (sl < 5) {
    // every partial of this slot is requested before the first is used (each load is a round trip to memory; S <= 15 is three of
    // them in a row otherwise, on the critical path of the layer); the combination runs in decode.hip's order
    for (int s0 = sl; s0 < p.S; s0 += 15) {
      float ms[3], ls[3], a[3][8];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int s2 = s0 + 5 * u;
        const float* ps = pp + (size_t)min(s2, p.S - 1) * CH_PSTRIDE;
        ms[u] = __hip_atomic_load(ps, AKI_RLX_AGENT);
        ls[u] = __hip_atomic_load(ps + 1, AKI_RLX_AGENT);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[u][e] = __hip_atomic_load(ps + 8 + ch * 8 + e, AKI_RLX_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (s0 + 5 * u >= p.S) break;
        const float mn = fmaxf(gm, ms[u]);
        const float fa = gm == -INFINITY ? 0.f : __expf(gm - mn), fb = ms[u] == -INFINITY ? 0.f : __expf(ms[u] - mn);
        lt = lt * fa + ls[u] * fb;
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = o8[e] * fa + a[u][e] * fb;
        gm = mn;
      }
    }
    float* sm = s_mg + (sl * 12 + ch) * 10;
    sm[0] = gm;
    sm[1] = lt;
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[2 + e] = o8[e];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (lane < 12) {
    float sv[5][10];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int e = 0; e < 10; ++e) sv[q][e] = s_mg[(q * 12 + lane) * 10 + e];
    lds_reads_landed();
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int e = 0; e < 10; ++e) asm volatile("" : "+v"(sv[q][e]));
    float M5 = -INFINITY;
#pragma unroll
    for (int q = 0; q < 5; ++q) M5 = fmaxf(M5, sv[q][0]);
    lt = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const float mq = sv[q][0];
      const float f = mq == -INFINITY ? 0.f : __expf(mq - M5);
      lt += sv[q][1] * f;
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] += sv[q][2 + e] * f;
    }
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    u32x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = pack_bf16x2(o8[2 * e] * inv, o8[2 * e + 1] * inv);
    if constexpr (BT) {
      const __amdgpu_buffer_rsrc_t ro = chain_rsrc(p.attn_o + (size_t)b * p.H * 96, p.H * 96 * 2);
      __builtin_amdgcn_raw_buffer_store_b128(ov, ro, (h * 96 + lane * 8) * 2, 0, 16);     // sc1: write-through
    } else {
      for (int rep = 0; rep < p.xrep; ++rep) {
        const __amdgpu_buffer_rsrc_t ro = chain_rsrc(p.attn_o + (size_t)rep * p.rep_stride, p.H * 96 * 2);
        __builtin_amdgcn_raw_buffer_store_b128(ov, ro, (h * 96 + lane * 8) * 2, 0, 16);     // sc1: write-through
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  chain_arrive(done_sync, bh, nB * p.H, lane);           // one arrival per (sequence, head)
#ifdef AKI_LAB_HOOKS
  if (p.stamps && layer == p.stamp_layer && lane == 0) p.stamps[((size_t)1 * 2048 + 1024 + bh) * 8 + 5] = wall_clock64();
#endif
}

// KCD = d / 512, KCF = F / 512 (bf16) - the register arrays are static; W8 halves both.
template <int KCD, int KCF, bool W8, int NBQ, int NBO, int NBG, int NBD, int PFQ = 1, int PFO = 1, int PFG = 1, int PFDN = 1, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void decode_chain_kernel(const ChainParams p0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_red = (float*)(smem + 16384);
  char* sx = smem;
  const int bid = blockIdx.x;
  // ---- the counters of this call: one of two sets, by the parity of the calls completed on this workspace --------------------
  // Every polled word must be zero when a call starts.  Until round 5 a small kernel in front of the chain zeroed them (4.7-5.3 us of a
  // 1637 us token, profiles/r05_decode_token_trace.txt).  Now call k uses set k & 1 and zeroes the OTHER set for call k + 1 (a store or two per
  // workgroup, any time during the call: nobody polls that set now - the previous call on this workspace has ended, it is stream-ordered).  k lives
  // in the workspace: the workgroup that completes the LAST phase writes k + 1.  Every workgroup of the launch has read k by then: the last
  // phase completes only after every phase before it has, and a workgroup arrives at its phase's counter after all it does with k.  (After a
  // give-up - sticky error word - the sets are in no defined state: the caller zero-fills the workspace before using it again.)  The word and
  // the sets are read and written at agent scope or across a kernel boundary only.
  ChainParams p = p0;
  const unsigned ep = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p0.epoch, AKI_RLX_AGENT));
  {
    const size_t cur = (ep & 1u) ? (size_t)p0.cnt_words : 0;
    p.sync = p0.sync + cur; p.head_sync = p0.head_sync + cur; p.attn_cnt = p0.attn_cnt + cur;
    u32x4* other = (u32x4*)(p0.sync + ((size_t)p0.cnt_words - cur));
    const int n16 = p0.cnt_words / 4, per = (n16 + (int)gridDim.x - 1) / (int)gridDim.x;
    const int z_end = min(n16, (bid + 1) * per);
    for (int i = bid * per + (int)threadIdx.x; i < z_end; i += 256) other[i] = u32x4{0u, 0u, 0u, 0u};
  }
  const int layer = bid / p.wg_layer;
  int r = bid - layer * p.wg_layer;
  const aki_decode_chain_layer& ly = p.layers[layer];
  unsigned* sy = p.sync + (size_t)layer * CH_PHASES * CH_SYNC_WORDS;
  unsigned* prev_down = layer > 0 ? sy - CH_SYNC_WORDS : nullptr;          // phase 4 of the layer before
  const bf16_t* h0 = layer == 0 ? p.h_in : p.hbuf + ((layer - 1) & 1) * p.d;
  const int h0_rep = layer == 0 ? 0 : 1;                                   // the embedding has one copy
  const bool last = layer == p.n_layers - 1;
  bf16_t* h2 = last ? p.h_out : p.hbuf + (layer & 1) * p.d;
  const unsigned code = ((unsigned)layer << 8);
  constexpr int NRD = W8 ? 4 : 2;          // rows per wave on the K = d matrices
  constexpr int KD = W8 ? KCD / 2 : KCD;
  constexpr int NRF = W8 ? 2 : 1;          // rows per wave on the K = F matrix
  constexpr int KF = W8 ? KCF / 2 : KCF;
  if (r < p.n_qkv) {
    chain_gemv<NRD, KD, false, true, W8, NBQ, PFQ>(p, r, p.n_qkv, ly.w_qkv, ly.s_qkv, p.d, 3 * p.H * 96, h0, h0_rep, (const bf16_t*)ly.norm1, nullptr, 0, p.qkv,
                                          CH_XREP, prev_down, sy + 0 * CH_SYNC_WORDS, code | 1u, sx, s_red, p.qkv_by_head,
                                          p.head_sync + (size_t)layer * p.H * 64);
    return;
  }
  r -= p.n_qkv;
  if (r < p.n_attn) {
    chain_attn<false>(p, ly, layer, r, sy + 0 * CH_SYNC_WORDS, sy + 1 * CH_SYNC_WORDS, code | 2u, smem);
    return;
  }
  r -= p.n_attn;
  if (r < p.n_o) {
    chain_gemv<NRD, KD, false, false, W8, NBO, PFO>(p, r, p.n_o, ly.w_o, ly.s_o, p.H * 96, p.d, p.attn_o, 1, nullptr, h0, h0_rep, p.h1, CH_XREP,
                                           sy + 1 * CH_SYNC_WORDS, sy + 2 * CH_SYNC_WORDS, code | 3u, sx, s_red);
    return;
  }
  r -= p.n_o;
  if (r < p.n_gu) {
    chain_gemv<NRD, KD, true, true, W8, NBG, PFG>(p, r, p.n_gu, ly.w_gate_up, ly.s_gate_up, p.d, p.F, p.h1, 1, (const bf16_t*)ly.norm2, nullptr, 0, p.act,
                                         CH_XREP, sy + 2 * CH_SYNC_WORDS, sy + 3 * CH_SYNC_WORDS, code | 4u, sx, s_red);
    return;
  }
  r -= p.n_gu;
  const unsigned fin = chain_gemv<NRF, KF, false, false, W8, NBD, PFDN>(p, r, p.n_down, ly.w_down, ly.s_down, p.F, p.d, p.act, 1, nullptr, p.h1, 1, h2,
                                                                         last ? 1 : CH_XREP, sy + 3 * CH_SYNC_WORDS, sy + 4 * CH_SYNC_WORDS, code | 5u, sx, s_red);
  if (last && fin && threadIdx.x == 0) __hip_atomic_store(p.epoch, ep + 1u, AKI_RLX_AGENT);      // the call is complete: the next one takes the other set
}

// ---- the batched chain: 2..8 sequences per step ------------------------------------------------------------------------------
// Same dataflow launch, same counters, same attention items (now per sequence); the four GEMV phases become 16-feature MFMA tiles
// (chain_gemm).  Workgroups per layer: qkv 576 (head-major: 18 per head) | attention ceil(B * H * S / 4) | o_proj 192 | gate_up 512 | down 192.
// RQ / RO / RG / RD: the ring depths (steps of 64 k whose operands a wave holds; the first ring-full is requested before the wait).
template <int RQ, int RO, int RG, int RD>
__global__ __launch_bounds__(512) void decode_chain_b_kernel(const ChainParams p0) {      // 144 VGPRs = ONE workgroup per CU; held to 128 (two per CU) the bare stream is faster, the step with its waits slower: EXPERIMENTS.md
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  ChainParams p = p0;                                      // the counter set of this call: see decode_chain_kernel
  const unsigned ep = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p0.epoch, AKI_RLX_AGENT));
  {
    const size_t cur = (ep & 1u) ? (size_t)p0.cnt_words : 0;
    p.sync = p0.sync + cur; p.head_sync = p0.head_sync + cur; p.attn_cnt = p0.attn_cnt + cur;
    u32x4* other = (u32x4*)(p0.sync + ((size_t)p0.cnt_words - cur));
    const int n16 = p0.cnt_words / 4, per = (n16 + (int)gridDim.x - 1) / (int)gridDim.x;
    const int z_end = min(n16, (bid + 1) * per);
    for (int i = bid * per + (int)threadIdx.x; i < z_end; i += 512) other[i] = u32x4{0u, 0u, 0u, 0u};
  }
  const int layer = bid / p.wg_layer;
  int r = bid - layer * p.wg_layer;
  const aki_decode_chain_layer& ly = p.layers[layer];
  unsigned* sy = p.sync + (size_t)layer * CH_PHASES * CH_SYNC_WORDS;
  unsigned* prev_down = layer > 0 ? sy - CH_SYNC_WORDS : nullptr;
  const size_t rows_d = (size_t)p.B * p.d;
  const bf16_t* h0 = layer == 0 ? p.h_in : p.hbuf + ((layer - 1) & 1) * rows_d;
  const bool last = layer == p.n_layers - 1;
  bf16_t* h2 = last ? p.h_out : p.hbuf + (layer & 1) * rows_d;
  const unsigned code = ((unsigned)layer << 8);
  if (r < p.n_qkv) {
    chain_gemm<6, false, 0, RQ>(p, r, p.n_qkv, ly.w_qkv, 3 * p.H * 96, h0, (const bf16_t*)ly.norm1, nullptr, p.qkv, prev_down, sy + 0 * CH_SYNC_WORDS,
                                code | 1u, smem, p.qkv_by_head, p.head_sync + (size_t)layer * p.H * 64);
    return;
  }
  r -= p.n_qkv;
  if (r < p.n_attn) {
    chain_attn<true>(p, ly, layer, r, sy + 0 * CH_SYNC_WORDS, sy + 1 * CH_SYNC_WORDS, code | 2u, smem);
    return;
  }
  r -= p.n_attn;
  if (r < p.n_o) {
    chain_gemm<6, false, 1, RO>(p, r, p.n_o, ly.w_o, p.d, p.attn_o, nullptr, h0, p.h1, sy + 1 * CH_SYNC_WORDS, sy + 2 * CH_SYNC_WORDS, code | 3u, smem);
    return;
  }
  r -= p.n_o;
  if (r < p.n_gu) {
    chain_gemm<6, true, 0, RG>(p, r, p.n_gu, ly.w_gate_up, p.F, p.h1, (const bf16_t*)ly.norm2, nullptr, p.act, sy + 2 * CH_SYNC_WORDS, sy + 3 * CH_SYNC_WORDS,
                               code | 4u, smem);
    return;
  }
  r -= p.n_gu;
  const unsigned fin = chain_gemm<16, false, 2, RD>(p, r, p.n_down, ly.w_down, p.d, p.act, nullptr, p.h1, h2, sy + 3 * CH_SYNC_WORDS, sy + 4 * CH_SYNC_WORDS,
                                                    code | 5u, smem);
  if (last && fin && threadIdx.x == 0) __hip_atomic_store(p.epoch, ep + 1u, AKI_RLX_AGENT);
}

// ---- host side ------------------------------------------------------------------------------------------------------
static inline size_t chain_cnt_bytes(int n_layers, int H, int B = 1) { return aki_align_up((size_t)n_layers * ((size_t)CH_PHASES * CH_SYNC_WORDS + (size_t)B * H + (size_t)H * 64) * 4, 256); }
static inline size_t chain_vec_elems(int d, int H, int F) { return aki_align_up((size_t)(3 * H * 96 + H * 96 + d + F + 2 * d) * 2 + 256, 256) / 2; }   // one copy (one row)

#ifdef AKI_LAB_HOOKS
static int g_chain_sleep = 8, g_chain_xrep = CH_XREP_USED, g_chain_nflags = CH_FLAGS, g_chain_nowait = 0, g_chain_nb = 0, g_chain_lds_pad = 0, g_chain_stamp_layer = -1, g_chain_touch = -1;
static unsigned long long* g_chain_stamps = nullptr;
static unsigned g_chain_fault_code = 0;
static int g_chain_fault_skip = 0;
#endif

static void chain_split(int H, int cap, int max_keys, int& S, int& T, int B = 1) {
  if (max_keys <= 0 || max_keys > cap) max_keys = cap;
  const int tiles = (max_keys + 63) / 64, tiles_cap = (cap + 63) / 64;     // decode.hip's rule: T from the capacity, S from max_keys
  T = (int)(((size_t)B * H * tiles_cap + AKI_DEC_ITEMS - 1) / AKI_DEC_ITEMS);
  if (T < 1) T = 1;
  S = (tiles + T - 1) / T;
}

// workspace: counter set 0 | counter set 1 | 256 bytes: the sticky error word (+0) and the count of completed calls (+64) | hand-off vectors | partials
size_t decode_chain_err_offset(int n_layers, int H) { return 2 * chain_cnt_bytes(n_layers, H); }

// the batched chain: counters (tickets per sequence and head) x 2 | error word, call count | the hand-off rows [B][n] | partials per sequence
size_t decode_chain_b_err_offset(int n_layers, int H, int B) { return 2 * chain_cnt_bytes(n_layers, H, B); }
size_t decode_chain_b_ws_bytes(int n_layers, int d, int H, int F, int cap, int B) {
  const size_t tiles = ((size_t)cap + 63) / 64;
  return 2 * chain_cnt_bytes(n_layers, H, B) + 256 + (size_t)B * chain_vec_elems(d, H, F) * 2 + (size_t)B * H * tiles * CH_PSTRIDE * 4;
}

size_t decode_chain_ws_bytes(int n_layers, int d, int H, int F, int cap) {
  const size_t tiles = ((size_t)cap + 63) / 64;
  return 2 * chain_cnt_bytes(n_layers, H) + 256 /* error word, call count */ + chain_vec_elems(d, H, F) * 2 * CH_XREP + (size_t)H * tiles * CH_PSTRIDE * 4;
}

// The batched chain is built, bit-identical to the five launches per layer and NOT faster (3.3-3.5 vs 3.2 ms per step at batch 8: 1664 fat
// workgroups per layer against 512 resident ones, EXPERIMENTS.md round 5): it is compiled into the LAB library only.  The product library
// answers batch > 1 with AKI_ERR_UNSUPPORTED and the host path keeps its five launches per layer.
#ifdef AKI_LAB_HOOKS
static int decode_chain_b_launch(const aki_decode_chain_args* a, hipStream_t stream) {
  const int d = a->d, H = a->H, F = a->F, B = a->batch;
  if (a->Dh != 96 || d != 3072 || H != 32 || F != 8192 || a->dtype != AKI_DT_BF16 || B < 2 || B > 8) return AKI_ERR_UNSUPPORTED;
  if (a->workspace_bytes < decode_chain_b_ws_bytes(a->n_layers, d, H, F, a->capacity, B) || ((uintptr_t)a->workspace & 255)) return AKI_ERR_WORKSPACE;
  ChainParams p;
  p.layers = a->layers; p.n_layers = a->n_layers; p.B = B;
  p.h_in = (const bf16_t*)a->h_in; p.h_out = (bf16_t*)a->h_out;
  p.cos = a->cos; p.sin = a->sin; p.cache_len = a->cache_len; p.vbits = a->col_valid_bits; p.nwords = a->nwords;
  p.d = d; p.H = H; p.F = F; p.cap = a->capacity; p.scale = a->scale; p.eps = a->rms_eps;
  chain_split(H, a->capacity, a->max_keys, p.S, p.T, B);
  char* ws = (char*)a->workspace;
  const size_t cb = chain_cnt_bytes(a->n_layers, H, B);
  p.sync = (unsigned*)ws;
  p.head_sync = p.sync + (size_t)a->n_layers * CH_PHASES * CH_SYNC_WORDS;
  p.attn_cnt = p.head_sync + (size_t)a->n_layers * H * 64;
  p.err = (unsigned*)(ws + 2 * cb);
  p.epoch = p.err + 16;
  p.cnt_words = (int)(cb / 4);
  bf16_t* v = (bf16_t*)(ws + 2 * cb + 256);
  p.rep_stride = 0;
  p.qkv = v; v += (size_t)B * 3 * H * 96;
  p.attn_o = v; v += (size_t)B * H * 96;
  p.h1 = v; v += (size_t)B * d;
  p.act = v; v += (size_t)B * F;
  p.hbuf = v;
  p.part = (float*)(ws + 2 * cb + 256 + (size_t)B * chain_vec_elems(d, H, F) * 2);
  p.nbq = p.nbo = p.nbg = p.nbd = 1;
  p.n_qkv = 3 * H * 96 / 16;
  p.n_attn = (B * H * p.S + 7) / 8;
  p.n_o = d / 16;
  p.n_gu = F / 16;
  p.n_down = d / 16;
  p.wg_layer = p.n_qkv + p.n_attn + p.n_o + p.n_gu + p.n_down;
  p.qkv_by_head = 6;                      // 96 rows of a head = six 16-feature tiles: 18 workgroups produce one head's q, k and v
  p.sleep_n = 8; p.xrep = 1; p.nflags = CH_FLAGS; p.nowait = 0; p.touch = 0;
#ifdef AKI_LAB_HOOKS
  p.sleep_n = g_chain_sleep; p.nowait = g_chain_nowait;
  p.stamps = g_chain_stamps; p.stamp_layer = g_chain_stamp_layer;
  p.fault_code = 0;
  if (g_chain_fault_code != 0u && g_chain_fault_skip-- == 0) { p.fault_code = g_chain_fault_code; g_chain_fault_code = 0; }
#endif
  // LDS: the normalised / copied rows of a K = 3072 phase (49 KB at eight rows); the partial tiles (16 KB with SwiGLU) and the attention phase's
  // per-wave scratch (12 KB) lie over them
  int SMEM = B * (3072 * 2 + 16);
  if (SMEM < 8 * 3072) SMEM = 8 * 3072;       // eight attention items x 3 KB; the partial tiles take 16 KB
  const dim3 grid((unsigned)a->n_layers * (unsigned)p.wg_layer), block(512);
  AKI_CLEAR_ERR();
#define AKI_CHAIN_B_LAUNCH(RQ, RO, RG, RD)                                                                                                  \
  do {                                                                                                                                      \
    static bool attr = false;                                                                                                               \
    if (!attr) {                                                                                                                            \
      if (hipFuncSetAttribute((const void*)decode_chain_b_kernel<RQ, RO, RG, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * (3072 * 2 + 16)) != hipSuccess) \
        return AKI_ERR_LAUNCH;                                                                                                              \
      attr = true;                                                                                                                          \
    }                                                                                                                                       \
    hipLaunchKernelGGL((decode_chain_b_kernel<RQ, RO, RG, RD>), grid, block, SMEM, stream, p);                                               \
  } while (0)
#ifdef AKI_LAB_HOOKS
  if (g_chain_nb == 1) AKI_CHAIN_B_LAUNCH(6, 6, 2, 4);
  else if (g_chain_nb == 2) AKI_CHAIN_B_LAUNCH(6, 6, 6, 8);
  else if (g_chain_nb == 3) AKI_CHAIN_B_LAUNCH(6, 6, 4, 8);
  else if (g_chain_nb == 4) AKI_CHAIN_B_LAUNCH(3, 3, 2, 4);
  else
#endif
  AKI_CHAIN_B_LAUNCH(CH_B_RQ, CH_B_RO, CH_B_RG, CH_B_RD);
#undef AKI_CHAIN_B_LAUNCH
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

#endif   // AKI_LAB_HOOKS: the batched chain

int decode_chain_launch(const aki_decode_chain_args* a, hipStream_t stream) {
#ifdef AKI_LAB_HOOKS
  if (a->batch > 1) return decode_chain_b_launch(a, stream);
#else
  if (a->batch > 1) return AKI_ERR_UNSUPPORTED;      // lab library only (see above)
#endif
  const int d = a->d, H = a->H, F = a->F;
  if (a->Dh != 96 || d != H * 96 || d % 512 || F % 512) return AKI_ERR_UNSUPPORTED;
  const bool w8 = a->dtype == AKI_DT_W8A16;
  if (!w8 && a->dtype != AKI_DT_BF16) return AKI_ERR_UNSUPPORTED;
  if (!(d == 3072 && F == 8192)) return AKI_ERR_UNSUPPORTED;     // the register arrays are sized at compile time: Phi-3.5-mini
  if (a->workspace_bytes < decode_chain_ws_bytes(a->n_layers, d, H, F, a->capacity) || ((uintptr_t)a->workspace & 255)) return AKI_ERR_WORKSPACE;
  ChainParams p;
  p.layers = a->layers; p.n_layers = a->n_layers; p.B = 1;
  p.h_in = (const bf16_t*)a->h_in; p.h_out = (bf16_t*)a->h_out;
  p.cos = a->cos; p.sin = a->sin; p.cache_len = a->cache_len; p.vbits = a->col_valid_bits; p.nwords = a->nwords;
  p.d = d; p.H = H; p.F = F; p.cap = a->capacity; p.scale = a->scale; p.eps = a->rms_eps;
  chain_split(H, a->capacity, a->max_keys, p.S, p.T);
  char* ws = (char*)a->workspace;
  const size_t cb = chain_cnt_bytes(a->n_layers, H);
  p.sync = (unsigned*)ws;
  p.head_sync = p.sync + (size_t)a->n_layers * CH_PHASES * CH_SYNC_WORDS;          // 128-byte lines first, the tickets behind them
  p.attn_cnt = p.head_sync + (size_t)a->n_layers * H * 64;
  p.err = (unsigned*)(ws + 2 * cb);
  p.epoch = p.err + 16;
  p.cnt_words = (int)(cb / 4);
  bf16_t* v = (bf16_t*)(ws + 2 * cb + 256);
  p.rep_stride = (int)chain_vec_elems(d, H, F);
  p.qkv = v; v += 3 * H * 96;
  p.attn_o = v; v += H * 96;
  p.h1 = v; v += d;
  p.act = v; v += F;
  p.hbuf = v;
  p.part = (float*)(ws + 2 * cb + 256 + chain_vec_elems(d, H, F) * 2 * CH_XREP);
  const int rd = w8 ? 4 : 2, rf = w8 ? 2 : 1;     // rows per wave and batch (see the kernel)
  p.nbq = w8 ? 1 : CH_NBQ; p.nbo = w8 ? 1 : CH_NBO; p.nbg = w8 ? 1 : CH_NBG; p.nbd = w8 ? 1 : CH_NBD;
#ifdef AKI_LAB_HOOKS
  const int presets[][4] = {{w8 ? 1 : CH_NBQ, w8 ? 1 : CH_NBO, w8 ? 1 : CH_NBG, w8 ? 1 : CH_NBD}, {8, 8, 8, 8}, {4, 2, 8, 2}, {8, 2, 8, 4}, {4, 4, 4, 4}, {16, 4, 16, 4}, {8, 2, 16, 2}, {1, 1, 1, 1}, {8, 4, 16, 4}, {2, 1, 4, 1}, {4, 2, 4, 2}, {2, 2, 4, 2}, {2, 2, 2, 2}, {2, 2, 4, 2}, {2, 2, 4, 2}, {2, 2, 8, 2}, {2, 2, 4, 4}, {2, 2, 4, 2}, {2, 2, 8, 2}, {2, 2, 4, 2}, {2, 2, 4, 4}, {4, 2, 8, 4}, {2, 2, 4, 2}, {4, 4, 8, 4}, {2, 2, 4, 2}, {2, 2, 4, 2}, {2, 2, 4, 2}};
  const int* ps = presets[(w8 && g_chain_nb != 9 && g_chain_nb != 20 && g_chain_nb != 21 && (g_chain_nb < 12 || g_chain_nb > 14)) ? 0 : g_chain_nb];
  p.nbq = ps[0]; p.nbo = ps[1]; p.nbg = ps[2]; p.nbd = ps[3];
  if (w8 && g_chain_nb == 20) { p.nbq = 1; p.nbo = 1; p.nbg = 2; p.nbd = 1; }
  if (w8 && g_chain_nb == 21) { p.nbq = 1; p.nbo = 1; p.nbg = 1; p.nbd = 2; }
  if (w8 && g_chain_nb == 7) { p.nbq = 2; p.nbo = 2; p.nbg = 2; p.nbd = 2; }
#endif
  auto wgs = [&](int n_out, int fpw, int nb) { return (n_out + 4 * fpw * nb - 1) / (4 * fpw * nb); };
  p.n_qkv = wgs(3 * H * 96, rd, p.nbq);
  p.n_attn = (H * p.S + 3) / 4;
  p.n_o = wgs(d, rd, p.nbo);
  p.n_gu = wgs(F, rd / 2, p.nbg);
  p.n_down = wgs(d, rf, p.nbd);
  p.wg_layer = p.n_qkv + p.n_attn + p.n_o + p.n_gu + p.n_down;
  {
    const int rows_wg = 4 * rd * p.nbq;                 // rows of w_qkv per workgroup
    p.qkv_by_head = (CH_QKV_BY_HEAD && 96 % rows_wg == 0) ? 96 / rows_wg : 0;
  }
  p.sleep_n = 8; p.xrep = CH_XREP_USED; p.nflags = CH_FLAGS; p.nowait = 0; p.touch = CH_TOUCH;
#ifdef AKI_LAB_HOOKS
  p.sleep_n = g_chain_sleep; p.xrep = g_chain_xrep; p.nflags = g_chain_nflags; p.nowait = g_chain_nowait;
  if (g_chain_touch >= 0) { p.touch = g_chain_touch & 1; if (g_chain_touch & 2) p.qkv_by_head = 0; }      // lab: touch | (one flag for the whole qkv phase) << 1
  p.stamps = g_chain_stamps; p.stamp_layer = g_chain_stamp_layer;
  p.fault_code = 0;
  if (g_chain_fault_code != 0u && g_chain_fault_skip-- == 0) { p.fault_code = g_chain_fault_code; g_chain_fault_code = 0; }     // one shot
#endif
  AKI_CLEAR_ERR();
  // Every polled word is zero when the kernel starts: the caller zero-fills the workspace once, and each call leaves the counter set of the NEXT
  // call zeroed (see the top of the kernel).  History: hipMemsetAsync in front of every call was used first - eager calls were fine, but as a
  // captured memset node (ROCm 7.2, 1 MB) it left a constant non-zero word pattern in the block in some processes: every READY flag then read
  // "set", no workgroup ever waited, and the replayed step ran at the speed of the bare weight stream (1.28 ms) with wrong logits, while the
  // 2-layer graph test of the time happened to pass.  tests/test_decode_gpu.py compares a FULL-DEPTH graph replay with the five-launch path,
  // and every timing tool checks logits.  A zeroing kernel of this library in front of every call came next (4.7-5.3 us per token).
  int SMEM = 16384 + 64;                 // x (<= 8192 bf16) + the norm's partial sums; the attention phase carves 4 x 3 KiB of it
#ifdef AKI_LAB_HOOKS
  SMEM += g_chain_lds_pad;               // lab: unused LDS that limits the workgroups resident per CU (160 KiB / SMEM)
#endif
  const dim3 grid((unsigned)a->n_layers * (unsigned)p.wg_layer), block(256);
#define AKI_CHAIN_LAUNCH(W8V, A, B, C, D)                                                                                                   \
  do {                                                                                                                                      \
    if (SMEM > 48 * 1024) (void)hipFuncSetAttribute((const void*)decode_chain_kernel<6, 16, W8V, A, B, C, D>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); \
    hipLaunchKernelGGL((decode_chain_kernel<6, 16, W8V, A, B, C, D>), grid, block, SMEM, stream, p);                                         \
  } while (0)
#define AKI_CHAIN_LAUNCH3(W8V, A, B, C, D, PA, PB, PC, PD, OC)                                                                               \
  do {                                                                                                                                      \
    if (SMEM > 48 * 1024) (void)hipFuncSetAttribute((const void*)decode_chain_kernel<6, 16, W8V, A, B, C, D, PA, PB, PC, PD, OC>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); \
    hipLaunchKernelGGL((decode_chain_kernel<6, 16, W8V, A, B, C, D, PA, PB, PC, PD, OC>), grid, block, SMEM, stream, p);                     \
  } while (0)
#define AKI_CHAIN_LAUNCH2(W8V, A, B, C, D, PA, PB, PC, PD)                                                                                   \
  do {                                                                                                                                      \
    if (SMEM > 48 * 1024) (void)hipFuncSetAttribute((const void*)decode_chain_kernel<6, 16, W8V, A, B, C, D, PA, PB, PC, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); \
    hipLaunchKernelGGL((decode_chain_kernel<6, 16, W8V, A, B, C, D, PA, PB, PC, PD>), grid, block, SMEM, stream, p);                         \
  } while (0)
#ifdef AKI_LAB_HOOKS
  if (w8 && g_chain_nb == 12) AKI_CHAIN_LAUNCH2(true, 2, 2, 2, 2, 2, 2, 2, 2);
  else if (w8 && g_chain_nb == 13) AKI_CHAIN_LAUNCH2(true, 2, 2, 4, 2, 2, 2, 2, 2);
  else if (w8 && g_chain_nb == 14) AKI_CHAIN_LAUNCH2(true, 2, 2, 4, 2, 2, 2, 2, 1);
  else if (w8 && g_chain_nb == 7) AKI_CHAIN_LAUNCH2(true, 2, 2, 2, 2, 1, 1, 1, 1);
  else if (w8 && g_chain_nb == 9) AKI_CHAIN_LAUNCH2(true, 2, 1, 4, 1, 2, 1, 4, 1);
  else if (w8 && g_chain_nb == 20) AKI_CHAIN_LAUNCH2(true, 1, 1, 2, 1, 1, 1, 2, 1);
  else if (w8 && g_chain_nb == 21) AKI_CHAIN_LAUNCH2(true, 1, 1, 1, 2, 1, 1, 1, 2);
  else
#endif
  if (w8) AKI_CHAIN_LAUNCH2(true, 1, 1, 1, 1, 1, 1, 1, 1);
#ifdef AKI_LAB_HOOKS
  else if (g_chain_nb == 1) AKI_CHAIN_LAUNCH(false, 8, 8, 8, 8);
  else if (g_chain_nb == 2) AKI_CHAIN_LAUNCH(false, 4, 2, 8, 2);
  else if (g_chain_nb == 3) AKI_CHAIN_LAUNCH(false, 8, 2, 8, 4);
  else if (g_chain_nb == 4) AKI_CHAIN_LAUNCH(false, 4, 4, 4, 4);
  else if (g_chain_nb == 5) AKI_CHAIN_LAUNCH(false, 16, 4, 16, 4);
  else if (g_chain_nb == 6) AKI_CHAIN_LAUNCH(false, 8, 2, 16, 2);
  else if (g_chain_nb == 7) AKI_CHAIN_LAUNCH(false, 1, 1, 1, 1);
  else if (g_chain_nb == 8) AKI_CHAIN_LAUNCH(false, 8, 4, 16, 4);
  else if (g_chain_nb == 9) AKI_CHAIN_LAUNCH(false, 2, 1, 4, 1);
  else if (g_chain_nb == 10) AKI_CHAIN_LAUNCH(false, 4, 2, 4, 2);
  else if (g_chain_nb == 11) AKI_CHAIN_LAUNCH(false, 2, 2, 4, 2);
  else if (g_chain_nb == 12) AKI_CHAIN_LAUNCH2(false, 2, 2, 2, 2, 2, 2, 2, 2);      // batches requested before the wait: qkv, o, gate_up, down
  else if (g_chain_nb == 13) AKI_CHAIN_LAUNCH2(false, 2, 2, 4, 2, 2, 2, 2, 2);
  else if (g_chain_nb == 14) AKI_CHAIN_LAUNCH2(false, 2, 2, 4, 2, 2, 2, 2, 1);
  else if (g_chain_nb == 15) AKI_CHAIN_LAUNCH2(false, 2, 2, 8, 2, 2, 2, 2, 1);
  else if (g_chain_nb == 16) AKI_CHAIN_LAUNCH2(false, 2, 2, 4, 4, 2, 2, 2, 1);
  else if (g_chain_nb == 17) AKI_CHAIN_LAUNCH2(false, 2, 2, 4, 2, 2, 2, 1, 2);
  else if (g_chain_nb == 18) AKI_CHAIN_LAUNCH2(false, 2, 2, 8, 2, 2, 2, 2, 2);
  else if (g_chain_nb == 19) AKI_CHAIN_LAUNCH2(false, 2, 2, 4, 2, 2, 2, 1, 1);
  else if (g_chain_nb == 20) AKI_CHAIN_LAUNCH2(false, 2, 2, 4, 4, 2, 2, 1, 1);
  else if (g_chain_nb == 21) AKI_CHAIN_LAUNCH2(false, 4, 2, 8, 4, 2, 2, 2, 1);
  else if (g_chain_nb == 22) AKI_CHAIN_LAUNCH2(false, 2, 2, 4, 2, 1, 1, 1, 1);
  else if (g_chain_nb == 23) AKI_CHAIN_LAUNCH2(false, 4, 4, 8, 4, 1, 1, 1, 1);
  else if (g_chain_nb == 24) AKI_CHAIN_LAUNCH3(false, 2, 2, 4, 2, 2, 2, 2, 1, 4);
  else if (g_chain_nb == 25) AKI_CHAIN_LAUNCH2(false, 2, 2, 4, 2, 2, 2, 4, 2);   // every weight of every phase before the wait: ~250 VGPRs, 2 workgroups per CU
  else if (g_chain_nb == 26) AKI_CHAIN_LAUNCH2(false, 2, 2, 4, 2, 2, 2, 3, 2);   // the product's batches held to 128 VGPRs (4 workgroups per CU; spills)
#endif
  else AKI_CHAIN_LAUNCH2(false, CH_NBQ, CH_NBO, CH_NBG, CH_NBD, CH_PFQ, CH_PFO, CH_PFG, CH_PFD);
#undef AKI_CHAIN_LAUNCH
  AKI_LAUNCH_CHECK();
  return AKI_OK;
}

}  // namespace aki

#ifdef AKI_LAB_HOOKS
// Lab build only: poll period (x 64 cycles), copies of the hand-off vectors (1..8), READY flags per phase (1..32), and
// nowait = 1: no dependency waits at all (WRONG results - the time of the bare weight stream in this workgroup structure).
// device buffer of 5 x 2048 x 8 uint64 (or NULL) and the layer whose workgroups stamp their wall clock into it (tools/decode_chain_edges.py)
extern "C" void aki_lab_set_chain_stamps(void* buf, int layer) { aki::g_chain_stamps = (unsigned long long*)buf; aki::g_chain_stamp_layer = layer; }
extern "C" void aki_lab_set_chain_lds(int pad_bytes) { aki::g_chain_lds_pad = pad_bytes < 0 ? 0 : (pad_bytes > 140 * 1024 ? 140 * 1024 : pad_bytes); }
// preset of batches per workgroup (qkv, o_proj, gate_up, down): 0 product {2,2,2,2}, 1 {8,8,8,8}, 2 {4,2,8,2}, 3 {8,2,8,4}, 4 {4,4,4,4}, 5 {16,4,16,4}, 6 {8,2,16,2}, 7 {1,1,1,1}, 8 {8,4,16,4}, 9 {2,1,4,1}, 10 {4,2,4,2}, 11 {2,2,4,2}
extern "C" void aki_lab_set_chain_nb(int preset) { aki::g_chain_nb = (preset >= 0 && preset <= 26) ? preset : 0; }
extern "C" void aki_lab_set_chain(int sleep_n, int xrep, int nflags, int nowait) {
  aki::g_chain_sleep = sleep_n < 0 ? 0 : sleep_n;
  aki::g_chain_xrep = xrep < 1 ? 1 : (xrep > aki::CH_XREP ? aki::CH_XREP : xrep);
  aki::g_chain_nflags = nflags < 1 ? 1 : (nflags > aki::CH_FLAGS ? aki::CH_FLAGS : nflags);
  aki::g_chain_nowait = nowait ? 1 : 0;
}
// Fault injection: the chain launch number `skip` from now (0 = the next one) treats the wait `code` = layer << 8 | phase (phase 1 qkv, 3 o_proj,
// 4 gate_up, 5 down; layer >= 1 for phase 1) as given up at once - error word set, every flag raised, garbage output.  One shot; code 0 disarms.
extern "C" void aki_lab_set_chain_fault(int code, int skip) { aki::g_chain_fault_code = (unsigned)code; aki::g_chain_fault_skip = skip < 0 ? 0 : skip; }
// -1: the product setting; 0 / 1: touch loads of the batches beyond the register slots off / on.
extern "C" void aki_lab_set_chain_touch(int touch) { aki::g_chain_touch = touch; }
#endif
