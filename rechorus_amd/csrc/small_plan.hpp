// small_plan.hpp -- grouping of a SMALL batch's row ids (B (1 + K) + B <= 32,768 keys) by row, inside the first launch
// of the small-batch BPRMF step (small_step.hip, bprmf_fused.hip): what bucket_plan.hip does in five launches for
// millions of keys, done by 128 workgroups that share the grid with the fused forward / backward workgroups.
//
// Reference being replaced: aten::embedding_dense_backward behind loss.backward() (helpers/BaseRunner.py:205) at the
// reference's own default --batch_size 256 (helpers/BaseRunner.py:33), where a step of ~10 dependent launches is pure
// launch / kernel latency.
//
// Workgroup w owns the rows whose joint key (item id, or n_items + user id) has key % 128 == w -- a hash, so the Zipf
// head spreads.  It scans all keys (each of its 8 waves a contiguous slice, in position order), keeps its own as
// (key << 15 | position) in LDS, sorts them (rank sort up to 1,024 entries, a bitonic network beyond) -- positions are unique, so the order is total and
// "stable" by construction: every row's positions ascend -- and writes its rows as rc_plan_row records plus the
// grouped positions into its OWN segment of the output arrays: no global atomics, no counters to zero, a
// deterministic layout.  Per-workgroup row counts go to cnt[]; the update kernel prefix-sums the 128 entries itself.
// A workgroup whose keys overflow the LDS budget (thousands of duplicates of a few ids -- never at random) falls back
// to selecting its rows one by one with scans: slow, correct.
#pragma once
#include "common.hpp"

namespace rc {

constexpr int kSmallThreads = 512;
constexpr int kSmallWaves = kSmallThreads / 64;
constexpr int kSmallPlanWgs = 128;         // power of two (owner = key & 127): ~200 keys each at B = 256, K = 99 (64: 19.7 us front launch, 128: 16.5, 256: 21.4 -- more workgroups than CUs)
constexpr int kSmallMaxKeys = 32768;       // positions fit 15 bits
constexpr int kSmallWaveCap = 256;         // LDS entries per wave region
constexpr int kSmallCap = 2048;            // entries the sort buffer holds
constexpr size_t small_lds_bytes(int cap, int wave_cap) {
  return ((size_t)kSmallWaves * wave_cap + cap) * sizeof(uint64_t) + ((size_t)cap + 2) * sizeof(uint16_t) + 64 * sizeof(uint32_t);
}
constexpr size_t kSmallLdsBytes = small_lds_bytes(kSmallCap, kSmallWaveCap);
// the plan-only launch (rc_small_row_sums: no fused workgroups share the launch, one plan workgroup per CU at most) can afford
// buffers that hold EVERY key of a list of up to 8,192 ids in one workgroup: such a list can never overflow, whatever its skew
// (a sequence model's padding id collects most of a batch's history slots)
constexpr int kSmallCapBig = 8192, kSmallWaveCapBig = 1024;
constexpr size_t kSmallLdsBytesBig = small_lds_bytes(kSmallCapBig, kSmallWaveCapBig);

// a key no row can have (n_rows < 2^32 - 1): an id of -1 narrows to it and takes no part in the grouping -- the occurrences of a
// context model's numeric fields, which own no row of the virtual concatenated table (rc_gather_fields_mixed)
constexpr uint32_t kSmallSkipKey = 0xFFFFFFFFu;

struct SmallCnt { uint32_t rows_a, rows_b, occ, pad; };   // per plan workgroup

struct SmallPlanArgs {
  const int64_t* ids_a;   // [n_a] item ids, batch order
  const int64_t* ids_b;   // [n - n_a] user ids
  uint32_t n_a, n;
  uint32_t base_b;        // joint key of list b = base_b + id
  rc_plan_row* rows;      // [kSmallPlanWgs][n] segment w: list-a rows, then list-b rows
  uint32_t* occ;          // [kSmallPlanWgs][n]
  SmallCnt* cnt;          // [kSmallPlanWgs]
};

#ifdef RC_X_TIMING
#define RC_T(k) do { if (threadIdx.x == 0 && w == 0) tstamp[k] = wall_clock64(); } while (0)
#else
#define RC_T(k) do { } while (0)
#endif

// where position p's key comes from: one or two id lists (the default), or whatever a caller's functor derives it from (the
// composite (field, id) keys of a context model's gather, formed from the fields' own id tensors: fm_bce.hip)
// (a functor enumerates the keys in CHUNKS of 64 -- one per lane of a wave --, the chunks in GROUPS that share a wave-uniform context
// (loaded once per group: open()): chunks(a), group_end(a, c) (first chunk behind the group of chunk c), key(a, ctx, c, lane)
// (kSmallSkipKey: no key in this slot) and the position pos(a, ctx, c, lane) the plan records for it.  The scan order is free -- the
// sort below orders by (key, position) -- but the positions of one key must ascend with (c, lane).)
struct SmallListKey {
  struct Ctx {};
  __device__ __forceinline__ uint32_t chunks(const SmallPlanArgs& a) const { return (a.n + 63u) >> 6; }
  __device__ __forceinline__ uint32_t group_end(const SmallPlanArgs& a, uint32_t) const { return (a.n + 63u) >> 6; }
  __device__ __forceinline__ Ctx open(const SmallPlanArgs&, uint32_t) const { return Ctx(); }
  __device__ __forceinline__ uint32_t key(const SmallPlanArgs& a, const Ctx&, uint32_t c, uint32_t lane) const {
    const uint32_t p = c * 64u + lane;
    if (p >= a.n) return 0xFFFFFFFFu;
    return p < a.n_a ? (uint32_t)a.ids_a[p] : a.base_b + (uint32_t)a.ids_b[p - a.n_a];
  }
  __device__ __forceinline__ uint32_t pos(const SmallPlanArgs&, const Ctx&, uint32_t c, uint32_t lane) const { return c * 64u + lane; }
};

// exact n / d for n * d < 2^32 by one v_mul_hi_u32 (magic = ceil(2^32 / d), d >= 2; d == 1: magic 0 stands for "n itself")
__host__ __device__ __forceinline__ uint32_t small_div_magic(uint32_t d) { return d <= 1u ? 0u : (uint32_t)((((uint64_t)1 << 32) + d - 1) / d); }
__device__ __forceinline__ uint32_t small_div(uint32_t n, uint32_t magic) { return magic == 0u ? n : __umulhi(n, magic); }

// one plan workgroup (blockDim = kSmallThreads); smem: small_lds_bytes(CAP, WCAP) of dynamic LDS
template <int CAP = kSmallCap, int WCAP = kSmallWaveCap, class KEY = SmallListKey>
__device__ __forceinline__ void small_plan_block(const SmallPlanArgs& a, uint32_t w, unsigned char* smem, const KEY& small_key = KEY()) {
  constexpr int kSmallCap = CAP, kSmallWaveCap = WCAP;   // (shadow the defaults below)
  uint64_t* region = reinterpret_cast<uint64_t*>(smem);                       // [kSmallWaves][kSmallWaveCap]
  uint64_t* buf = region + (size_t)kSmallWaves * kSmallWaveCap;               // [kSmallCap]
  uint16_t* hl = reinterpret_cast<uint16_t*>(buf + kSmallCap);                // [kSmallCap + 2] head positions
  uint32_t* sc = reinterpret_cast<uint32_t*>(hl + kSmallCap + 2);             // scratch words
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  rc_plan_row* rows = a.rows + (size_t)w * a.n;
  uint32_t* occ = a.occ + (size_t)w * a.n;
  const uint32_t occ_base = w * a.n;
#ifdef RC_X_TIMING
  uint64_t tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  RC_T(0);

  // ---- 1. scan: every wave a contiguous range of chunks, owned keys into its region in scan order
  const uint32_t n_chunks = small_key.chunks(a);
  const uint32_t per_wave = (n_chunks + kSmallWaves - 1) / kSmallWaves;
  const uint32_t c_beg = wave * per_wave, c_end = (c_beg + per_wave < n_chunks) ? c_beg + per_wave : n_chunks;
  uint32_t cnt = 0;
  constexpr int kBatch = 32;  // chunks requested together: a 32,768-key batch is two trips per wave
  for (uint32_t cg = c_beg; cg < c_end;) {
    const typename KEY::Ctx ctx = small_key.open(a, cg);
    uint32_t g_end = small_key.group_end(a, cg);
    if (g_end > c_end) g_end = c_end;
    for (uint32_t c0 = cg; c0 < g_end; c0 += kBatch) {
      uint32_t key[kBatch];
#pragma unroll
      for (int q = 0; q < kBatch; ++q) key[q] = c0 + q < g_end ? small_key.key(a, ctx, c0 + q, (uint32_t)lane) : 0xFFFFFFFFu;
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
        if (c0 + q >= g_end) break;  // wave-uniform
        const bool mine = key[q] != kSmallSkipKey && (key[q] & (kSmallPlanWgs - 1)) == w;
        const uint64_t m = __ballot(mine);
        const uint32_t at = cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (mine && at < (uint32_t)kSmallWaveCap)
          region[(size_t)wave * kSmallWaveCap + at] = ((uint64_t)key[q] << 15) | small_key.pos(a, ctx, c0 + q, (uint32_t)lane);
        cnt += (uint32_t)__popcll(m);
      }
    }
    cg = g_end;
  }
  if (lane == 0) sc[wave] = cnt;
  RC_T(1);
  __syncthreads();
  RC_T(2);
  uint32_t m_tot = 0, my_off = 0;
  bool overflow = false;
  for (int q = 0; q < kSmallWaves; ++q) {
    const uint32_t c = sc[q];
    if (q < wave) my_off += c;
    m_tot += c;
    overflow |= c > (uint32_t)kSmallWaveCap;
  }
  overflow |= m_tot > (uint32_t)kSmallCap;
  __syncthreads();  // sc[] is reused below

  if (!overflow) {
    // ---- 2. compact, then sort.  Up to 1,024 entries: rank sort -- every entry counts the smaller ones (entries are
    // unique, so the count IS its place; broadcast LDS reads, one barrier) -- else a bitonic network (padding sorts last).
    if (m_tot <= 1024u) {
      uint64_t* stage = buf + kSmallCap / 2;   // [1024] beside the result area buf[0, 1024)
      for (uint32_t i = lane; i < cnt; i += 64) stage[my_off + i] = region[(size_t)wave * kSmallWaveCap + i];
      __syncthreads();
      // a thread ranks its (up to) two entries in one sweep over the staged array: 16-byte broadcast reads, four
      // entries per trip, unrolled -- the sweep is LDS-latency bound otherwise (measured: the plan part of the
      // launch took 19 us with a one-entry-per-trip loop)
      const uint64_t mine0 = tid < m_tot ? stage[tid] : ~0ull;
      const uint64_t mine1 = tid + kSmallThreads < m_tot ? stage[tid + kSmallThreads] : ~0ull;
      const bool two = m_tot > (uint32_t)kSmallThreads;   // workgroup-uniform
      uint32_t rank0 = 0, rank1 = 0;
      uint32_t j = 0;
      for (; j + 8 <= m_tot; j += 8) {   // four 16-byte reads in flight per trip
        ulonglong2 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const ulonglong2*>(stage + j + 2 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) rank0 += (uint32_t)(x[q].x < mine0) + (uint32_t)(x[q].y < mine0);
        if (two) {
#pragma unroll
          for (int q = 0; q < 4; ++q) rank1 += (uint32_t)(x[q].x < mine1) + (uint32_t)(x[q].y < mine1);
        }
      }
      for (; j < m_tot; ++j) {
        rank0 += (uint32_t)(stage[j] < mine0);
        rank1 += (uint32_t)(stage[j] < mine1);
      }
      if (tid < m_tot) buf[rank0] = mine0;
      if (tid + kSmallThreads < m_tot) buf[rank1] = mine1;
      __syncthreads();
    } else {
      uint32_t M = 2048;
      while (M < m_tot) M <<= 1;
      for (uint32_t i = lane; i < cnt; i += 64) buf[my_off + i] = region[(size_t)wave * kSmallWaveCap + i];
      for (uint32_t i = m_tot + tid; i < M; i += kSmallThreads) buf[i] = ~0ull;
      __syncthreads();
      for (uint32_t k = 2; k <= M; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
          for (uint32_t i = tid; i < M; i += kSmallThreads) {
            const uint32_t x = i ^ j;
            if (x > i) {
              const uint64_t vi = buf[i], vx = buf[x];
              const bool up = (i & k) == 0;
              if ((vi > vx) == up) { buf[i] = vx; buf[x] = vi; }
            }
          }
          __syncthreads();
        }
      }
    }
    RC_T(3);
    // ---- 3. heads: hl[r] = index of the first entry of the workgroup's r-th row; list-a rows come first (smaller keys)
    if (tid == 0) { sc[8] = 0; sc[9] = 0; }
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t i0 = 0; i0 < m_tot; i0 += kSmallThreads) {
      const uint32_t i = i0 + tid;
      const bool in = i < m_tot;
      const uint32_t key = in ? (uint32_t)(buf[i] >> 15) : 0u;
      const bool head = in && (i == 0 || (uint32_t)(buf[i - 1] >> 15) != key);
      const uint64_t hb = __ballot(head);
      if (lane == 0) sc[10 + wave] = (uint32_t)__popcll(hb);
      __syncthreads();
      uint32_t before = carry, total = 0;
      for (int q = 0; q < kSmallWaves; ++q) {
        if (q < wave) before += sc[10 + q];
        total += sc[10 + q];
      }
      if (head) {
        hl[before + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull))] = (uint16_t)i;
        if (key < a.base_b) atomicAdd(&sc[8], 1u);  // integer count, order-free
      }
      carry += total;
      __syncthreads();
    }
    const uint32_t R = carry;
    if (tid == 0) hl[R] = (uint16_t)m_tot;   // (m_tot <= 4096 fits)
    __syncthreads();
    RC_T(4);
    // ---- 4. records + grouped positions into this workgroup's segment
    for (uint32_t r = tid; r < R; r += kSmallThreads) {
      const uint32_t i = hl[r];
      const uint32_t key = (uint32_t)(buf[i] >> 15);
      rc_plan_row e;
      e.row = key < a.base_b ? key : key - a.base_b;
      e.start = occ_base + i;
      e.n = (uint32_t)hl[r + 1] - i;
      e.reserved = (uint32_t)(buf[i] & 0x7FFFu);   // the row's first position: saves the update kernel one dependent load
      rows[r] = e;
    }
    for (uint32_t i = tid; i < m_tot; i += kSmallThreads) occ[i] = (uint32_t)(buf[i] & 0x7FFFu);
    if (tid == 0) {
      SmallCnt c;
      c.rows_a = sc[8]; c.rows_b = R - sc[8]; c.occ = m_tot; c.pad = 0;
      a.cnt[w] = c;
    }
    RC_T(5);
#ifdef RC_X_TIMING
    if (tid == 0 && w == 0)
      printf("plan wg0 (10 ns ticks): scan %llu wait %llu sort %llu heads %llu emit %llu  m=%u\n", tstamp[1] - tstamp[0], tstamp[2] - tstamp[1],
             tstamp[3] - tstamp[2], tstamp[4] - tstamp[3], tstamp[5] - tstamp[4], m_tot);
#endif
    return;
  }

  // ---- overflow: select the rows one by one in ascending key order (two scans per row)
  uint32_t n_rows = 0, n_rows_a = 0, n_occ = 0;
  uint64_t last = 0;  // smallest key not yet taken, + 1 (0: nothing taken)
  for (;;) {
    uint32_t best = 0xFFFFFFFFu;
    bool any = false;
    for (uint32_t c = wave; c < n_chunks; c += kSmallWaves) {
      const uint32_t key = small_key.key(a, small_key.open(a, c), c, (uint32_t)lane);
      if (key != kSmallSkipKey && (key & (kSmallPlanWgs - 1)) == w && (uint64_t)key + 1 > last && (!any || key < best)) { best = key; any = true; }
    }
    // block min
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t ob = __shfl_xor(best, off, 64);
      const int oa = __shfl_xor((int)any, off, 64);
      if (oa && (!any || ob < best)) { best = ob; any = true; }
    }
    if (lane == 0) { sc[16 + wave] = best; sc[24 + wave] = any ? 1u : 0u; }
    __syncthreads();
    best = 0xFFFFFFFFu; any = false;
    for (int q = 0; q < kSmallWaves; ++q)
      if (sc[24 + q] && (!any || sc[16 + q] < best)) { best = sc[16 + q]; any = true; }
    __syncthreads();
    if (!any) break;
    // occurrences of `best` in scan order (ascending positions): tiles of kSmallWaves consecutive chunks, wave q the q-th
    uint32_t taken = 0;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += kSmallWaves) {
      const uint32_t c = c0 + wave;
      const uint32_t cc = c < n_chunks ? c : 0u;      // (wave-uniform; a wave past the end re-reads chunk 0 and drops it)
      const typename KEY::Ctx hctx = small_key.open(a, cc);
      const bool hit = c < n_chunks && small_key.key(a, hctx, cc, (uint32_t)lane) == best;
      const uint64_t hb = __ballot(hit);
      if (lane == 0) sc[32 + wave] = (uint32_t)__popcll(hb);
      __syncthreads();
      uint32_t before = taken, total = 0;
      for (int q = 0; q < kSmallWaves; ++q) {
        if (q < wave) before += sc[32 + q];
        total += sc[32 + q];
      }
      if (hit) {
        const uint32_t at = before + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        const uint32_t p = small_key.pos(a, hctx, cc, (uint32_t)lane);
        occ[n_occ + at] = p;
        if (at == 0) sc[40] = p;   // the row's first position
      }
      taken += total;
      __syncthreads();
    }
    if (tid == 0) {
      rc_plan_row e;
      e.row = best < a.base_b ? best : best - a.base_b;
      e.start = occ_base + n_occ;
      e.n = taken;
      e.reserved = sc[40];
      rows[n_rows] = e;
    }
    ++n_rows;
    if (best < a.base_b) ++n_rows_a;
    n_occ += taken;
    last = (uint64_t)best + 1;
  }
  if (tid == 0) {
    SmallCnt c;
    c.rows_a = n_rows_a; c.rows_b = n_rows - n_rows_a; c.occ = n_occ; c.pad = 0;
    a.cnt[w] = c;
  }
}

}  // namespace rc
