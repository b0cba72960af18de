// seg_update.hip -- atomic-free segmented scatter of embedding-row gradients, fused with
// the optimizer's row update.
//
// Reference semantics being replaced: autograd's EmbeddingBackward ->
// aten::embedding_dense_backward (zero-fill a dense [n_rows,d] grad, index_add every
// occurrence) followed by torch.optim's step over the table
// (helpers/BaseRunner.py:193,205-206; optimizer built at :110-114).
//
// Input: the stably sorted id list of the batch (sort_ids.hip).  Pipeline:
//   1. segment_heads_kernel   one thread per sorted position: flags single-occurrence rows
//                             (for the fused BPRMF kernel) and appends the positions that
//                             start a segment to a COMPACT list, so that in step 2 every
//                             lane-group of every wave has a row to work on.
//   2. seg_update_kernel      one lane-group (d/4 lanes, a float4 each) per listed head:
//                             sums the segment's gradient rows in ascending sorted position
//                             (fixed order => bit-reproducible, no float atomics), reads the
//                             table row once, applies the optimizer, writes it once.  The
//                             per-occurrence gradient row is rebuilt as coef[o]*Src[srow(o)]
//                             (BPRMF items: g[b,c] * U[uid[b]], U rows come from L2) and never
//                             exists in HBM.
//   3. hot rows (> kLongSeg occurrences, the head of a Zipf distribution) are cut into
//      kChunk-occurrence chunks (long_plan), each chunk reduced by one workgroup with a
//      fixed LDS tree (long_chunk), the chunk partials of a row combined in chunk order
//      (long_final).  Work per block is bounded whatever the skew.
// The only atomics are INTEGER appends / slot reservations (list order does not influence
// any floating-point result).
#include <mutex>

#include "common.hpp"
#include "opt_math.hpp"

namespace rc {

constexpr int kLongSeg = 32;  // occurrences one lane-group sums sequentially
constexpr int kChunk = 256;   // occurrences per workgroup for hot rows

struct RowInfo { uint32_t j0, end, pbase, nchunks; };
struct ChunkInfo { uint32_t row, k; };

enum { CNT_LONG = 0, CNT_CHUNKS = 1, CNT_PARTIAL = 2, CNT_HEADS = 3, CNT_N = 4 };

struct SegArgs {
  float* W;
  float* M;
  float* V;
  const uint32_t* keys;
  const uint32_t* perm;
  int64_t n_occ;
  const float* coef;
  const float* src;
  const int64_t* src_index;
  const float* src2;   // optional second source: occurrences o >= n_split read src2[o - n_split]
  uint32_t n_split;
  uint32_t key_base;   // table row = key - key_base   (keys / perm may be a slice of a joint sort)
  uint32_t occ_base;   // occurrence = perm[j] - occ_base
  int div;
  int d;  // generic kernels only
  float* dense_grad;
  const uint32_t* heads;
  const uint32_t* n_heads;
  uint32_t* counters;  // CNT_*
  uint32_t* long_list;
  RowInfo* rows;
  ChunkInfo* chunks;
  float* partial;
  uint32_t long_cap, chunk_cap, partial_cap;
  int skip_single;
  uint32_t* narrow_ws;   // narrow rows, dense gradient: room for the tiles' boundary sums (seg_narrow_tiles_kernel); null = one wave per segment
  int planned;   // rows route: the hot rows are already listed in rows[] / chunks[] (rc_rows_plan_build): no hand-over atomics
  OptScalars o;
  // capturable mode (hipGraph replay, rc_segmented_update_rows_dev): Adam's step count is read from device memory and the two
  // bias-correction scalars are derived from it in the kernel (as rc_dense_update_multi_dev does) instead of on the host
  const int64_t* step_dev;
  double beta1, beta2, lr;
  // pair mode (rc_segmented_update_pair): two tables of width D/2 that share keys / perm / heads are updated as
  // ONE row of width D -- the lower half of a row's lanes works on table a (W, M, V, src, dense_grad), the upper
  // half on table b.  Zero for the ordinary single-table call.
  int pair;
  float *Wb, *Mb, *Vb;
  const float* srcb;
  float* dense_grad_b;
};

// pair mode: which table a lane belongs to and its float4 slot inside that table's row
template <int D>
__device__ __forceinline__ bool pair_upper(int l, int& ll) {
  constexpr int H = D / 8;
  const bool up = l >= H;
  ll = up ? l - H : l;
  return up;
}

__device__ __forceinline__ void add4(float4& x, const float4& y) {
  x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
}

// ---- 1. heads -------------------------------------------------------------------------
// A workgroup owns a tile of kHeadIters*256 consecutive sorted positions and reserves its
// slice of the list with ONE atomicAdd (a device-scope counter retires only ~80 atomics/us,
// MI355X_MICROARCH.md "fanin"/"dequeue": one atomic per head, even wave-aggregated, cost
// 1.2 ms here).  Inside the tile the list keeps sorted-position order (ballot + popcount
// prefix), so neighbouring lane-groups of seg_update_kernel read neighbouring keys/perm.
constexpr int kHeadIters = 16;

__global__ __launch_bounds__(kBlock) void segment_heads_kernel(
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm, int64_t n,
    int only_multi, uint8_t* __restrict__ single, uint32_t* __restrict__ heads,
    uint32_t* __restrict__ n_heads) {
  constexpr int kWaves = kBlock / 64;
  __shared__ uint32_t s_cnt[kHeadIters * kWaves];
  __shared__ uint32_t s_base;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t tile0 = (int64_t)blockIdx.x * (kHeadIters * kBlock);
  unsigned long long ballots[kHeadIters];
#pragma unroll
  for (int it = 0; it < kHeadIters; ++it) {
    const int64_t j = tile0 + (int64_t)it * kBlock + threadIdx.x;
    bool listed = false;
    if (j < n) {
      const uint32_t k = keys[j];
      const bool head = j == 0 || keys[j - 1] != k;
      const bool multi = j + 1 < n && keys[j + 1] == k;
      // flags are pre-set to 1 (hipMemsetAsync); only members of multi-occurrence segments are
      // cleared, which halves the scattered byte stores when most rows occur once
      if (single && (multi || !head)) single[perm[j]] = 0;
      listed = head && (multi || !only_multi);
    }
    ballots[it] = __ballot(listed);
    if (lane == 0) s_cnt[it * kWaves + wave] = (uint32_t)__popcll(ballots[it]);
  }
  if (heads == nullptr) return;  // block-uniform
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int i = 0; i < kHeadIters * kWaves; ++i) {  // exclusive scan, (it, wave) order
      const uint32_t c = s_cnt[i];
      s_cnt[i] = total;
      total += c;
    }
    s_base = total ? atomicAdd(n_heads, total) : 0u;
  }
  __syncthreads();
  const uint32_t base = s_base;
#pragma unroll
  for (int it = 0; it < kHeadIters; ++it) {
    const unsigned long long b = ballots[it];
    if ((b >> lane) & 1ull) {
      const uint32_t below = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
      const int64_t j = tile0 + (int64_t)it * kBlock + threadIdx.x;
      heads[base + s_cnt[it * kWaves + wave] + below] = (uint32_t)j;
    }
  }
}

// the optimizer scalars of this launch (same expressions as fill_opt_scalars: double, then narrowed like torch does)
template <int MODE>
__device__ __forceinline__ OptScalars seg_scalars(const SegArgs& a) {
  OptScalars o = a.o;
  if (MODE == MODE_ADAM && a.step_dev != nullptr) {
    const double step = (double)*a.step_dev;
    const double bc1 = 1.0 - pow(a.beta1, step), bc2 = 1.0 - pow(a.beta2, step);
    o.neg_step = (float)(-(a.lr / bc1));
    o.bc2_sqrt = (float)sqrt(bc2);
  }
  return o;
}

// ---- 2. one lane-group per head ---------------------------------------------------------
template <int D, int MODE>
__device__ __forceinline__ void apply_row4(const SegArgs& a, uint32_t key, int l, float4 w,
                                           const float4& g) {
  constexpr int LPR = D / 4;
  if (a.pair) {  // uniform
    int ll;
    const bool up = pair_upper<D>(l, ll);
    const size_t idx = (size_t)(key - a.key_base) * (LPR / 2) + ll;
    if (MODE == MODE_DENSE_GRAD) {
      reinterpret_cast<float4*>(up ? a.dense_grad_b : a.dense_grad)[idx] = g;
      return;
    }
    opt_row4<MODE>(seg_scalars<MODE>(a), up ? a.Wb : a.W, up ? a.Mb : a.M, up ? a.Vb : a.V, idx, w, g);
    return;
  }
  const size_t idx = (size_t)(key - a.key_base) * LPR + l;
  if (MODE == MODE_DENSE_GRAD) {
    reinterpret_cast<float4*>(a.dense_grad)[idx] = g;
    return;
  }
  opt_row4<MODE>(seg_scalars<MODE>(a), a.W, a.M, a.V, idx, w, g);
}

template <int D, int MODE>
__device__ __forceinline__ float4 load_row4(const SegArgs& a, uint32_t key, int l) {
  constexpr int LPR = D / 4;
  if (MODE == MODE_DENSE_GRAD) return make_float4(0, 0, 0, 0);
  if (a.pair) {
    int ll;
    const bool up = pair_upper<D>(l, ll);
    return load_stream4(reinterpret_cast<const float4*>(up ? a.Wb : a.W) + (size_t)(key - a.key_base) * (LPR / 2) + ll);
  }
  // read-once row: non-temporal, so that the gathered source rows (a.src, re-read per occurrence)
  // keep their L2 lines (measured at config 2: item update 0.402 -> 0.384 ms, user update 0.085 -> 0.068)
  return load_stream4(reinterpret_cast<const float4*>(a.W) + (size_t)(key - a.key_base) * LPR + l);
}

// gradient row of occurrence o (= perm[j]), this lane's float4
template <int D>
__device__ __forceinline__ float4 occ_grad4_o(const SegArgs& a, uint32_t o, int l) {
  constexpr int LPR = D / 4;
  o -= a.occ_base;
  if (a.pair) {  // plain gradient rows of the two tables, one per occurrence
    int ll;
    const bool up = pair_upper<D>(l, ll);
    return reinterpret_cast<const float4*>(up ? a.srcb : a.src)[(size_t)o * (LPR / 2) + ll];
  }
  if (a.src2 && o >= a.n_split)  // second source: plain gradient rows, one per occurrence
    return reinterpret_cast<const float4*>(a.src2)[(size_t)(o - a.n_split) * LPR + l];
  const float c = a.coef ? a.coef[o] : 1.0f;  // (a non-temporal load here costs 10 %: measured)
  int64_t sr = (a.div == 1) ? (int64_t)o : (int64_t)(o / (uint32_t)a.div);
  if (a.src_index) sr = a.src_index[sr];
  float4 s = reinterpret_cast<const float4*>(a.src)[(size_t)sr * LPR + l];
  s.x *= c; s.y *= c; s.z *= c; s.w *= c;
  return s;
}
template <int D>
__device__ __forceinline__ float4 occ_grad4(const SegArgs& a, int64_t jj, int l) {
  return occ_grad4_o<D>(a, a.perm[jj], l);
}

// ONLY_MULTI: the list holds only heads with >= 2 occurrences (singletons were updated by
// the fused BPRMF kernel), so the first two gradient rows are always fetched together.
// Dependent-load depth: heads[g] -> {keys[j], keys[j+1], perm[j], perm[j+1]} ->
// {W row, coef, index} -> {src rows} -> store.
template <int D, int MODE, bool ONLY_MULTI>
__global__ __launch_bounds__(kBlock) void seg_update_kernel(SegArgs a) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  const int l = threadIdx.x % LPR;
  const int64_t n = a.n_occ;
  const int64_t g = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
  if (g >= (int64_t)*a.n_heads) return;  // no cross-lane ops in this kernel
  const int64_t j = a.heads[g];
  const bool has_next = j + 1 < n;
  const uint32_t key = a.keys[j];
  const uint32_t knext = has_next ? a.keys[j + 1] : ~key;
  const uint32_t o0 = a.perm[j];
  const uint32_t o1 = a.perm[has_next ? j + 1 : j];
  const bool multi = knext == key;
  // hot row (more than kLongSeg occurrences; keys are sorted, so one probe decides): hand it to the chunked path
  // BEFORE touching its rows -- tables where most segments are hot (a small catalogue under a large batch) spent
  // most of this kernel summing 32 occurrences per row only to drop them
  if (j + kLongSeg < n && a.keys[j + kLongSeg] == key) {
    if (l == 0) {
      const uint32_t slot = atomicAdd(&a.counters[CNT_LONG], 1u);
      if (slot < a.long_cap) a.long_list[slot] = (uint32_t)j;
    }
    return;
  }
  const float4 w = load_row4<D, MODE>(a, key, l);
  float4 acc = occ_grad4_o<D>(a, o0, l);
  if (ONLY_MULTI || multi) {
    const float4 s1 = occ_grad4_o<D>(a, o1, l);
    add4(acc, s1);
    int64_t jj = j + 2;
    while (jj < n && a.keys[jj] == key) {
      if (jj - j >= kLongSeg) {  // hot row: hand over to the chunked path
        if (l == 0) {
          const uint32_t slot = atomicAdd(&a.counters[CNT_LONG], 1u);
          if (slot < a.long_cap) a.long_list[slot] = (uint32_t)j;
        }
        return;
      }
      const float4 s = occ_grad4<D>(a, jj, l);
      add4(acc, s);
      ++jj;
    }
  }
  apply_row4<D, MODE>(a, key, l, w, acc);
}

// Two heads per lane-group (ONLY_MULTI lists: every head has >= 2 occurrences).  The kernel is bound by
// the latency of its dependent gathers (head -> keys/perm -> coef/index -> source row), not by bandwidth:
// interleaving two independent heads doubles the loads in flight per lane.  Same summation order per row as
// seg_update_kernel, so results are bit-identical.
template <int D, int MODE>
__device__ __forceinline__ bool seg_tail(const SegArgs& a, int64_t j, uint32_t key, int l, float4& acc) {
  const int64_t n = a.n_occ;
  int64_t jj = j + 2;
  while (jj < n && a.keys[jj] == key) {
    if (jj - j >= kLongSeg) {  // hot row: hand over to the chunked path
      if (l == 0) {
        const uint32_t slot = atomicAdd(&a.counters[CNT_LONG], 1u);
        if (slot < a.long_cap) a.long_list[slot] = (uint32_t)j;
      }
      return true;
    }
    const float4 s = occ_grad4<D>(a, jj, l);
    add4(acc, s);
    ++jj;
  }
  return false;
}

#ifndef RC_SEG_HPG
#define RC_SEG_HPG 2
#endif
constexpr int kSegHpg = RC_SEG_HPG;  // heads per lane-group

template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void seg_update_multi_x2_kernel(SegArgs a) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  constexpr int H = kSegHpg;
  const int l = threadIdx.x % LPR;
  const int64_t nh = (int64_t)*a.n_heads;
  const int64_t g = ((int64_t)blockIdx.x * GPB + threadIdx.x / LPR) * H;
  if (g >= nh) return;  // no cross-lane ops in this kernel
  int64_t j[H];
  uint32_t key[H], o0[H], o1[H];
  float4 w[H], acc[H], s1[H];
#pragma unroll
  for (int h = 0; h < H; ++h) j[h] = a.heads[g + h < nh ? g + h : g];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    key[h] = a.keys[j[h]];
    o0[h] = a.perm[j[h]];
    o1[h] = a.perm[j[h] + 1];
  }
  bool hot[H];  // more than kLongSeg occurrences: goes to the chunked path untouched (see seg_update_kernel)
#pragma unroll
  for (int h = 0; h < H; ++h) hot[h] = j[h] + kLongSeg < a.n_occ && a.keys[j[h] + kLongSeg] == key[h];
#pragma unroll
  for (int h = 0; h < H; ++h)
    if (!hot[h]) w[h] = load_row4<D, MODE>(a, key[h], l);
#pragma unroll
  for (int h = 0; h < H; ++h)
    if (!hot[h]) acc[h] = occ_grad4_o<D>(a, o0[h], l);
#pragma unroll
  for (int h = 0; h < H; ++h)
    if (!hot[h]) s1[h] = occ_grad4_o<D>(a, o1[h], l);
#pragma unroll
  for (int h = 0; h < H; ++h) {
    if (g + h >= nh) break;
    if (hot[h]) {
      if (l == 0) {
        const uint32_t slot = atomicAdd(&a.counters[CNT_LONG], 1u);
        if (slot < a.long_cap) a.long_list[slot] = (uint32_t)j[h];
      }
      continue;
    }
    add4(acc[h], s1[h]);
    if (!seg_tail<D, MODE>(a, j[h], key[h], l, acc[h])) apply_row4<D, MODE>(a, key[h], l, w[h], acc[h]);
  }
}

// ---- 3. hot rows ----------------------------------------------------------------------------
// end of the segment that starts at j0 (first index with a different key), found by the 16 lanes of a group
// together: every round the lanes probe 16 positions at once and keep the sub-interval that contains the
// boundary (a one-thread gallop + bisection needs ~2 log2(len) DEPENDENT loads of ~1 us each; this needs
// ~log16(len) rounds -- the planning kernel went from 32 to ~8 us per table and step).
__device__ __forceinline__ int group16_first_fail(bool ok, int l) {  // smallest lane index with !ok, 16 if none
  int v = ok ? 16 : l;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int64_t segment_end16(const uint32_t* __restrict__ keys, int64_t n, int64_t j0, int l) {
  const uint32_t key = keys[j0];
  int64_t lo = j0, hi = n;  // keys[lo] == key; hi == n or keys[hi] != key
  // gallop: lane l probes j0 + kLongSeg * 2^l (l < 16), repeated from the last hit while all lanes hit
  for (;;) {
    const int64_t p = lo + ((int64_t)kLongSeg << l);
    const bool ok = p < n && keys[p] == key;
    const int f = group16_first_fail(ok, l);
    const int64_t last_ok = __shfl(p, (threadIdx.x & 48) + (f > 0 ? f - 1 : 0), 64);
    const int64_t first_bad = __shfl(p, (threadIdx.x & 48) + (f < 16 ? f : 15), 64);
    if (f > 0) lo = last_ok;
    if (f < 16) {
      hi = first_bad < n ? first_bad : n;
      break;
    }
  }
  while (hi - lo > 1) {  // 17-way split of (lo, hi)
    const int64_t step = (hi - lo + 16) / 17;
    const int64_t q = lo + (int64_t)(l + 1) * step;
    const bool ok = q < hi ? keys[q] == key : false;
    const int f = group16_first_fail(ok, l);
    const int64_t last_ok = __shfl(q, (threadIdx.x & 48) + (f > 0 ? f - 1 : 0), 64);
    const int64_t first_bad = __shfl(q, (threadIdx.x & 48) + (f < 16 ? f : 15), 64);
    if (f > 0) lo = last_ok;
    if (f < 16 && first_bad < hi) hi = first_bad;
  }
  return hi;
}

// one 16-lane group per hot row
__global__ __launch_bounds__(kBlock) void long_plan_kernel(SegArgs a) {
  uint32_t n_long = a.counters[CNT_LONG];
  if (n_long > a.long_cap) n_long = a.long_cap;
  const int l = threadIdx.x & 15;
  const uint32_t groups = gridDim.x * (kBlock / 16);
  for (uint32_t i = blockIdx.x * (kBlock / 16) + threadIdx.x / 16; i < n_long; i += groups) {
    const int64_t j0 = a.long_list[i];
    const int64_t end = segment_end16(a.keys, a.n_occ, j0, l);
    const uint32_t nch = (uint32_t)((end - j0 + kChunk - 1) / kChunk);
    uint32_t pbase = 0xFFFFFFFFu, cbase = 0;
    if (l == 0) {
      if (nch >= 2) pbase = atomicAdd(&a.counters[CNT_PARTIAL], nch);
      cbase = atomicAdd(&a.counters[CNT_CHUNKS], nch);
      RowInfo r;
      r.j0 = (uint32_t)j0;
      r.end = (uint32_t)end;
      r.nchunks = nch;
      r.pbase = pbase;
      a.rows[i] = r;
    }
    cbase = __shfl(cbase, threadIdx.x & 48, 64);
    for (uint32_t k = l; k < nch; k += 16)
      if (cbase + k < a.chunk_cap) {
        ChunkInfo c;
        c.row = i;
        c.k = k;
        a.chunks[cbase + k] = c;
      }
  }
}

// LDS tree over the GPB lane-groups of a block, fixed order; result in group 0's slots
template <int LPR>
__device__ __forceinline__ void block_tree_sum(float4* part, int g) {
  constexpr int GPB = kBlock / LPR;
#pragma unroll
  for (int off = GPB / 2; off >= 1; off >>= 1) {
    __syncthreads();
    if (g < off) {
      float4 x = part[threadIdx.x];
      add4(x, part[threadIdx.x + off * LPR]);
      part[threadIdx.x] = x;
    }
  }
  __syncthreads();
}

template <int D, int MODE>
__device__ __forceinline__ void long_chunk_body(const SegArgs& a, uint32_t block, uint32_t n_blocks) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  __shared__ float4 part[kBlock];
  const int l = threadIdx.x % LPR;
  const int g = threadIdx.x / LPR;
  uint32_t n_chunks = a.counters[CNT_CHUNKS];
  if (n_chunks > a.chunk_cap) n_chunks = a.chunk_cap;
  for (uint32_t c = block; c < n_chunks; c += n_blocks) {
    const ChunkInfo ci = a.chunks[c];
    const RowInfo ri = a.rows[ci.row];
    const int64_t start = (int64_t)ri.j0 + (int64_t)ci.k * kChunk;
    const int64_t end = (start + kChunk < (int64_t)ri.end) ? start + kChunk : (int64_t)ri.end;
    // four independent occurrences per lane-group per trip (fixed pattern -> fixed order)
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t jj = start + g; jj < end; jj += 4 * GPB) {
      const float4 z = make_float4(0, 0, 0, 0);
      const float4 s0 = occ_grad4<D>(a, jj, l);
      const float4 s1 = (jj + GPB < end) ? occ_grad4<D>(a, jj + GPB, l) : z;
      const float4 s2 = (jj + 2 * GPB < end) ? occ_grad4<D>(a, jj + 2 * GPB, l) : z;
      const float4 s3 = (jj + 3 * GPB < end) ? occ_grad4<D>(a, jj + 3 * GPB, l) : z;
      add4(acc, s0); add4(acc, s1); add4(acc, s2); add4(acc, s3);
    }
    part[threadIdx.x] = acc;
    block_tree_sum<LPR>(part, g);
    if (g == 0) {
      if (ri.nchunks == 1) {
        const uint32_t key = a.keys[ri.j0];
        apply_row4<D, MODE>(a, key, l, load_row4<D, MODE>(a, key, l), part[threadIdx.x]);
      } else if (ri.pbase + ci.k < a.partial_cap) {
        reinterpret_cast<float4*>(a.partial)[(size_t)(ri.pbase + ci.k) * LPR + l] = part[threadIdx.x];
      }
    }
    __syncthreads();  // part[] is reused by the next chunk
  }
}

template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void long_chunk_kernel(SegArgs a) {
  long_chunk_body<D, MODE>(a, blockIdx.x, gridDim.x);
}

template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void long_final_kernel(SegArgs a) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  __shared__ float4 part[kBlock];
  const int l = threadIdx.x % LPR;
  const int g = threadIdx.x / LPR;
  uint32_t n_long = a.counters[CNT_LONG];
  if (n_long > a.long_cap) n_long = a.long_cap;
  for (uint32_t i = blockIdx.x; i < n_long; i += gridDim.x) {
    const RowInfo ri = a.rows[i];
    if (ri.nchunks < 2) continue;  // block-uniform
    float4 acc = make_float4(0, 0, 0, 0);
    for (uint32_t k = g; k < ri.nchunks; k += GPB)
      if (ri.pbase + k < a.partial_cap)
        add4(acc, reinterpret_cast<const float4*>(a.partial)[(size_t)(ri.pbase + k) * LPR + l]);
    part[threadIdx.x] = acc;
    block_tree_sum<LPR>(part, g);
    if (g == 0) {
      const uint32_t key = a.keys[ri.j0];
      apply_row4<D, MODE>(a, key, l, load_row4<D, MODE>(a, key, l), part[threadIdx.x]);
    }
    __syncthreads();
  }
}

// ---- 4. tables whose rows collect many occurrences each ----------------------------------------
// A small catalogue under a large batch (the reference's own datasets: 8.7 K items against 0.5 M candidate + history
// occurrences per SASRec step) makes EVERY row a hot row: the head list + chunk planning above then costs four
// latency-bound launches and one device-scope atomic per row.  With the number of table rows known, one pass over
// the sorted keys records each row's [start, end) and ONE wave per table row sums its occurrences: the wave's
// lane-groups take positions start + g, start + g + G, ... (four trips in flight each), the group sums are
// combined in a fixed xor tree.  Rows past kRowsWaveMax occurrences are handed to the chunked path.
constexpr int kRowsWaveMax = 192;

__global__ __launch_bounds__(kBlock) void segment_bounds_kernel(const uint32_t* __restrict__ keys, int64_t n, uint32_t key_base,
                                                                uint32_t n_rows, uint32_t* __restrict__ start,
                                                                uint32_t* __restrict__ end) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= n) return;
  const uint32_t k = keys[j];
  const uint32_t r = k - key_base;
  if (r >= n_rows) return;   // (keys outside the table: not this call's rows)
  if (j == 0 || keys[j - 1] != k) start[r] = (uint32_t)j;
  if (j + 1 == n || keys[j + 1] != k) end[r] = (uint32_t)(j + 1);
}

template <int D, int MODE>
__device__ __forceinline__ void seg_rows_body(const SegArgs& a, const uint32_t* __restrict__ start,
                                              const uint32_t* __restrict__ end, uint32_t n_rows, uint32_t block) {
  constexpr int LPR = D / 4;
  constexpr int G = 64 / LPR;   // lane-groups per wave
  const int lane = threadIdx.x & 63, l = lane % LPR, g = lane / LPR;
  const uint32_t r = block * (kBlock / 64) + (threadIdx.x >> 6);
  if (r >= n_rows) return;      // wave-uniform from here on
  const int64_t j0 = start[r], j1 = end[r];
  if (j1 <= j0) return;
  if (j1 - j0 > kRowsWaveMax) {
    if (lane == 0 && !a.planned) {
      const uint32_t slot = atomicAdd(&a.counters[CNT_LONG], 1u);
      if (slot < a.long_cap) a.long_list[slot] = (uint32_t)j0;
    }
    return;
  }
  const uint32_t key = r + a.key_base;
  const float4 w = load_row4<D, MODE>(a, key, l);
  // The segment's occurrence numbers first (at most kRowsWaveMax = 3 per lane, one round of independent loads), handed out by
  // shuffles; then 8 gradient rows in flight per lane-group.  (Before: perm[j] -> row per occurrence, four per group and round:
  // a row of 192 occurrences was twelve rounds of two dependent loads, and the longest rows set the kernel's time -- 36 us at
  // SASRec config 3 for 40 MB of traffic.)
  static_assert(kRowsWaveMax <= 192, "three occurrence numbers per lane");
  const int cnt = (int)(j1 - j0);
  uint32_t pm[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) pm[k] = 64 * k + lane < cnt ? a.perm[j0 + 64 * k + lane] : 0u;
  float4 acc = make_float4(0, 0, 0, 0);
  constexpr int U = 8;   // (U * G = 32 or more occurrences per round; a round never straddles a multiple of 64)
  for (int base = 0; base < cnt; base += U * G) {
    const int k = base >> 6;   // wave-uniform
    const uint32_t mine = k == 0 ? pm[0] : (k == 1 ? pm[1] : pm[2]);
    const uint32_t next = k == 0 ? pm[1] : pm[2];   // (U * G > 64 only for D < 32: the round's second half)
    float4 sv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * G + g;
      const uint32_t o = __shfl((idx >> 6) == k ? mine : next, idx & 63, 64);
      sv[u] = idx < cnt ? occ_grad4_o<D>(a, o, l) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) add4(acc, sv[u]);
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {   // both partners form the same sum: every group ends with the row's total
    acc.x += __shfl_xor(acc.x, off, 64);
    acc.y += __shfl_xor(acc.y, off, 64);
    acc.z += __shfl_xor(acc.z, off, 64);
    acc.w += __shfl_xor(acc.w, off, 64);
  }
  if (g == 0) apply_row4<D, MODE>(a, key, l, w, acc);
}

template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void seg_rows_kernel(SegArgs a, const uint32_t* __restrict__ start,
                                                          const uint32_t* __restrict__ end, uint32_t n_rows) {
  seg_rows_body<D, MODE>(a, start, end, n_rows, blockIdx.x);
}

// rows route with the hot rows planned ahead (section 5): the chunks of the hot rows and the one-wave rows in ONE launch -- the
// first n_chunk_blocks workgroups take the chunk list (they are the long ones: started first), the others four rows each
template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void seg_planned_kernel(SegArgs a, const uint32_t* __restrict__ start,
                                                             const uint32_t* __restrict__ end, uint32_t n_rows,
                                                             uint32_t n_chunk_blocks) {
  if (blockIdx.x < n_chunk_blocks) long_chunk_body<D, MODE>(a, blockIdx.x, n_chunk_blocks);   // (workgroup-uniform)
  else seg_rows_body<D, MODE>(a, start, end, n_rows, blockIdx.x - n_chunk_blocks);
}

template <int D, int MODE>
static int launch_seg_planned(const SegArgs& a, const uint32_t* start, const uint32_t* end, uint32_t n_rows, hipStream_t s) {
  const unsigned row_blocks = (n_rows + (kBlock / 64) - 1) / (kBlock / 64);
  const unsigned chunk_blocks = 1024;
  hipLaunchKernelGGL((seg_planned_kernel<D, MODE>), dim3(chunk_blocks + row_blocks), dim3(kBlock), 0, s, a, start, end, n_rows,
                     chunk_blocks);
  RC_LAUNCH_CHECK();
  hipLaunchKernelGGL((long_final_kernel<D, MODE>), dim3(256), dim3(kBlock), 0, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int MODE>
static int launch_seg_planned_mode(const SegArgs& a, const uint32_t* start, const uint32_t* end, uint32_t n_rows, hipStream_t s) {
  switch (a.d) {
    case 16: return launch_seg_planned<16, MODE>(a, start, end, n_rows, s);
    case 32: return launch_seg_planned<32, MODE>(a, start, end, n_rows, s);
    case 64: return launch_seg_planned<64, MODE>(a, start, end, n_rows, s);
    case 128: return launch_seg_planned<128, MODE>(a, start, end, n_rows, s);
    default: return launch_seg_planned<256, MODE>(a, start, end, n_rows, s);
  }
}

template <int D, int MODE>
static int launch_seg_rows(const SegArgs& a, const uint32_t* start, const uint32_t* end, uint32_t n_rows, hipStream_t s) {
  const unsigned blocks = (n_rows + (kBlock / 64) - 1) / (kBlock / 64);
  hipLaunchKernelGGL((seg_rows_kernel<D, MODE>), dim3(blocks), dim3(kBlock), 0, s, a, start, end, n_rows);
  RC_LAUNCH_CHECK();
  if (a.n_occ > kRowsWaveMax) {   // otherwise no row was handed over
    hipLaunchKernelGGL(long_plan_kernel, dim3(64), dim3(kBlock), 0, s, a);
    RC_LAUNCH_CHECK();
    hipLaunchKernelGGL((long_chunk_kernel<D, MODE>), dim3(1024), dim3(kBlock), 0, s, a);
    RC_LAUNCH_CHECK();
    hipLaunchKernelGGL((long_final_kernel<D, MODE>), dim3(256), dim3(kBlock), 0, s, a);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}

template <int MODE>
static int launch_seg_rows_mode(const SegArgs& a, const uint32_t* start, const uint32_t* end, uint32_t n_rows, hipStream_t s) {
  switch (a.d) {
    case 16: return launch_seg_rows<16, MODE>(a, start, end, n_rows, s);
    case 32: return launch_seg_rows<32, MODE>(a, start, end, n_rows, s);
    case 64: return launch_seg_rows<64, MODE>(a, start, end, n_rows, s);
    case 128: return launch_seg_rows<128, MODE>(a, start, end, n_rows, s);
    default: return launch_seg_rows<256, MODE>(a, start, end, n_rows, s);
  }
}

// ---- 5. the rows route without the radix sort: per-row [start, end) from a counting sort of the batch's ids ----------------
// The table of section 4 is small (n_rows <= kRpMaxRows: SASRec on the reference's own datasets, 8.7 K items) and the occurrences
// are the batch's two id tensors as the reference hands them over: ids_a (the candidates, [B, C]) and ids_b (the history windows,
// [B, L], slots l >= lengths[b] are padding and take no part).  Three launches replace the radix sort of candidate + history ids
// and the tensor glue that parked the padding behind the table:
//   count    one workgroup per tile of kRpTile occurrences: histogram over the table's rows in LDS -> matrix[tile][row]
//   prefix   one lane per row: exclusive prefix over the tiles in place, the row's total
//   scatter  every tile forms the rows' starts (an LDS scan of the totals) + its own offsets, then hands out the positions of its
//            occurrences IN OCCURRENCE ORDER: every wave owns a quarter of the tile, the waves' shares of a row follow from their
//            own counts (16-bit counters in LDS), inside a wave round after round, inside a round the lanes that hold the same
//            row find each other with one ballot per key bit -- perm is the stable sort's, so every row sum keeps its order and
//            the update is bit-identical to the sorted route; workgroup 0 also lists the hot rows' chunks (what long_plan_kernel
//            did with one atomic per row after the rows kernel had run).
// The first padding slot of the batch stays an occurrence of row 0 (its gradient row is zero), as on the sorted route.
constexpr int kRpTile = 4096;
constexpr int kRpRounds = kRpTile / kBlock;
constexpr int kRpMaxRows = 12288;   // LDS of the scatter: 4 B (first position) + 4 x 2 B (the waves' offsets) per row = 144 KB

struct RowsPlanArgs {
  const int64_t* ids_a;
  const int64_t* ids_b;
  const int64_t* len_b;
  int64_t n_a, n_b;
  int L_b;
  uint32_t n_rows;
  uint32_t n_tiles;
  int key_bits;
  uint32_t* matrix;     // [n_tiles][n_rows]
  uint32_t* totals;     // [n_rows]
  uint32_t* tile_pad;   // [n_tiles]: the tile's first padding occurrence, 0xFFFFFFFF if none
  uint32_t* start;
  uint32_t* end;
  uint32_t* keys;
  uint32_t* perm;
  uint32_t* counters;
  RowInfo* rows;
  ChunkInfo* chunks;
  uint32_t long_cap, chunk_cap;
  uint32_t* status;     // [0]: ids outside the table (they take no part)
};

// occurrence o of the batch: its id and, for a slot of the history windows, lengths[b] - l (<= 0: padding).  No branches: the
// address is clamped into the batch instead, so that a caller that asks for several occurrences has all their loads in flight at
// once (with a branch per occurrence the sixteen loads of a lane went one round trip after the other: 32 us per pass over 0.6 M ids)
template <bool HAS_LEN>
__device__ __forceinline__ void rp_load(const RowsPlanArgs& a, int64_t o, int64_t n_occ, int64_t& id, int64_t& room) {
  const int64_t oo = o < n_occ ? o : n_occ - 1;
  const bool in_b = oo >= a.n_a;
  const uint32_t j = in_b ? (uint32_t)(oo - a.n_a) : 0u;
  const int64_t* p = in_b ? a.ids_b + j : a.ids_a + oo;
  id = *p;
  room = 1;
  if (HAS_LEN) {
    const uint32_t b = j / (uint32_t)a.L_b;
    // (no select on the loaded value: the compiler turns a select with a load behind one arm back into a branch)
    const int64_t slot = in_b ? (int64_t)(j - b * (uint32_t)a.L_b) : -((int64_t)1 << 40);
    room = a.len_b[b] - slot;
  }
}
// -> the table row, -1: takes no part (pad: a padding slot; bad: an id outside the table)
__device__ __forceinline__ int rp_classify(const RowsPlanArgs& a, bool inside, int64_t id, int64_t room, bool& pad, bool& bad) {
  pad = inside && room <= 0;
  bad = inside && !pad && (id < 0 || id >= (int64_t)a.n_rows);
  return (inside && !pad && !bad) ? (int)id : -1;
}

// occurrence of (wave, round, lane) inside a tile: every wave owns kRpTile / 4 consecutive occurrences (the scatter hands out
// positions wave by wave without waiting: the waves' shares of a row are known from their counts)
__device__ __forceinline__ int64_t rp_occ(int64_t tile0, int wave, int r, int lane) {
  return tile0 + (int64_t)wave * (kRpTile / (kBlock / 64)) + r * 64 + lane;
}

template <bool HAS_LEN>
__global__ __launch_bounds__(kBlock) void rp_count_kernel(RowsPlanArgs a) {
  extern __shared__ uint32_t rp_hist[];
  __shared__ uint32_t s_pad, s_bad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_occ = a.n_a + a.n_b;
  const int64_t tile0 = (int64_t)blockIdx.x * kRpTile;
  int64_t id[kRpRounds], room[kRpRounds];
#pragma unroll
  for (int r = 0; r < kRpRounds; ++r) rp_load<HAS_LEN>(a, rp_occ(tile0, wave, r, lane), n_occ, id[r], room[r]);
  for (uint32_t i = threadIdx.x; i < a.n_rows; i += kBlock) rp_hist[i] = 0;
  if (threadIdx.x == 0) {
    s_pad = 0xFFFFFFFFu;
    s_bad = 0;
  }
  __syncthreads();
  uint32_t first_pad = 0xFFFFFFFFu, n_bad = 0;
#pragma unroll
  for (int r = 0; r < kRpRounds; ++r) {
    const int64_t o = rp_occ(tile0, wave, r, lane);
    bool pad, bad;
    const int k = rp_classify(a, o < n_occ, id[r], room[r], pad, bad);
    if (k >= 0) atomicAdd(&rp_hist[k], 1u);
    if (pad && (uint32_t)o < first_pad) first_pad = (uint32_t)o;
    n_bad += bad ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {   // (half of the history slots are padding: one LDS atomic per wave, not per slot)
    const uint32_t v = __shfl_xor(first_pad, off, 64);
    first_pad = v < first_pad ? v : first_pad;
    n_bad += __shfl_xor(n_bad, off, 64);
  }
  if (lane == 0) {
    atomicMin(&s_pad, first_pad);
    if (n_bad) atomicAdd(&s_bad, n_bad);
  }
  __syncthreads();
  uint32_t* out = a.matrix + (size_t)blockIdx.x * a.n_rows;
  for (uint32_t i = threadIdx.x; i < a.n_rows; i += kBlock) out[i] = rp_hist[i];
  if (threadIdx.x == 0) {
    a.tile_pad[blockIdx.x] = s_pad;
    if (s_bad) atomicAdd(a.status, s_bad);   // (an error count, read by the host on request only)
  }
}

__global__ __launch_bounds__(64) void rp_prefix_kernel(RowsPlanArgs a) {
  const uint32_t r = blockIdx.x * 64 + threadIdx.x;
  if (r >= a.n_rows) return;
  uint32_t run = 0;
  constexpr int U = 10;
  uint32_t t = 0;
  for (; t + U <= a.n_tiles; t += U) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a.matrix[(size_t)(t + u) * a.n_rows + r];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a.matrix[(size_t)(t + u) * a.n_rows + r] = run;
      run += v[u];
    }
  }
  for (; t < a.n_tiles; ++t) {
    const uint32_t v = a.matrix[(size_t)t * a.n_rows + r];
    a.matrix[(size_t)t * a.n_rows + r] = run;
    run += v;
  }
  a.totals[r] = run;
}

// exclusive scan of one uint4 per thread over the workgroup (x, y, z, w independently), the totals in tot
__device__ __forceinline__ uint4 rp_block_scan4(uint4 v, uint4* s_wave /* [kBlock / 64] */, uint4& tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t x = __shfl_up(inc.x, off, 64), y = __shfl_up(inc.y, off, 64), z = __shfl_up(inc.z, off, 64),
                   w = __shfl_up(inc.w, off, 64);
    if (lane >= off) {
      inc.x += x; inc.y += y; inc.z += z; inc.w += w;
    }
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint4 base = make_uint4(0, 0, 0, 0);
  tot = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    const uint4 t = s_wave[w];
    if (w < wave) {
      base.x += t.x; base.y += t.y; base.z += t.z; base.w += t.w;
    }
    tot.x += t.x; tot.y += t.y; tot.z += t.z; tot.w += t.w;
  }
  __syncthreads();
  return make_uint4(base.x + inc.x - v.x, base.y + inc.y - v.y, base.z + inc.z - v.z, base.w + inc.w - v.w);
}

// dst[i] (LDS) = or += src[i] (global) for i < n over the workgroup, eight loads in flight per lane (a plain loop is one round
// trip per element: the compiler does not overlap the iterations)
template <bool ADD>
__device__ __forceinline__ void rp_lds_from_global(uint32_t* dst, const uint32_t* __restrict__ src, uint32_t n) {
  constexpr int U = 8;
  for (uint32_t i0 = threadIdx.x; i0 < n; i0 += U * kBlock) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + u * kBlock;
      v[u] = src[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + u * kBlock;
      if (i < n) dst[i] = ADD ? dst[i] + v[u] : v[u];
    }
  }
}

template <bool HAS_LEN>
__global__ __launch_bounds__(kBlock) void rp_scatter_kernel(RowsPlanArgs a) {
  extern __shared__ uint32_t rp_hist[];   // totals -> starts -> this tile's first position in every row | the waves' shares
  __shared__ uint4 s_wave[kBlock / 64];
  __shared__ uint32_t s_min[kBlock / 64];
  constexpr int NW = kBlock / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool plan = blockIdx.x == 0;
  const int64_t n_occ = a.n_a + a.n_b;
  const int64_t tile0 = (int64_t)blockIdx.x * kRpTile;
  int64_t id[kRpRounds], room[kRpRounds];   // the tile's ids: requested first, needed after the rows' starts are known
#pragma unroll
  for (int r = 0; r < kRpRounds; ++r) rp_load<HAS_LEN>(a, rp_occ(tile0, wave, r, lane), n_occ, id[r], room[r]);
  // the batch's first padding slot (tiles are in occurrence order: the first tile that has one holds it)
  uint32_t fp = 0xFFFFFFFFu;
  for (uint32_t t = threadIdx.x; t < a.n_tiles; t += kBlock) {
    const uint32_t v = a.tile_pad[t];
    fp = v < fp ? v : fp;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t v = __shfl_xor(fp, off, 64);
    fp = v < fp ? v : fp;
  }
  if (lane == 0) s_min[wave] = fp;
  rp_lds_from_global<false>(rp_hist, a.totals, a.n_rows);
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) fp = s_min[w] < fp ? s_min[w] : fp;
  const uint32_t has_pad = fp != 0xFFFFFFFFu ? 1u : 0u;
  // rows [r0, r1) of this thread: sums, then the scan over the threads
  const uint32_t seg = (a.n_rows + kBlock - 1) / kBlock;
  const uint32_t r0 = threadIdx.x * seg < a.n_rows ? threadIdx.x * seg : a.n_rows;
  const uint32_t r1 = r0 + seg < a.n_rows ? r0 + seg : a.n_rows;
  uint4 mine = make_uint4(0, 0, 0, 0);   // occurrences, hot rows, chunks, partial rows
  for (uint32_t r = r0; r < r1; ++r) {
    const uint32_t cnt = rp_hist[r] + (r == 0 ? has_pad : 0u);
    mine.x += cnt;
    if (cnt > (uint32_t)kRowsWaveMax) {
      const uint32_t nch = (cnt + kChunk - 1) / kChunk;
      mine.y += 1;
      mine.z += nch;
      mine.w += nch >= 2 ? nch : 0u;
    }
  }
  uint4 tot;
  uint4 run = rp_block_scan4(mine, s_wave, tot);
  for (uint32_t r = r0; r < r1; ++r) {
    const uint32_t cnt = rp_hist[r] + (r == 0 ? has_pad : 0u);
    const uint32_t st = run.x;
    rp_hist[r] = st;
    run.x += cnt;
    if (plan) {
      a.start[r] = st;
      a.end[r] = st + cnt;
      if (cnt > (uint32_t)kRowsWaveMax) {
        const uint32_t nch = (cnt + kChunk - 1) / kChunk;
        if (run.y < a.long_cap) {
          RowInfo ri;
          ri.j0 = st;
          ri.end = st + cnt;
          ri.nchunks = nch;
          ri.pbase = nch >= 2 ? run.w : 0xFFFFFFFFu;
          a.rows[run.y] = ri;
          a.keys[st] = r;   // (the chunk kernels read a hot row's number at keys[j0]; nothing else reads keys on this route)
          for (uint32_t k = 0; k < nch; ++k)
            if (run.z + k < a.chunk_cap) {
              ChunkInfo c;
              c.row = run.y;
              c.k = k;
              a.chunks[run.z + k] = c;
            }
        }
        run.y += 1;
        run.z += nch;
        run.w += nch >= 2 ? nch : 0u;
      }
    }
  }
  if (plan && threadIdx.x == 0) {
    a.counters[CNT_LONG] = tot.y;
    a.counters[CNT_CHUNKS] = tot.z;
    a.counters[CNT_PARTIAL] = tot.w;
    if (has_pad) a.perm[a.totals[0]] = fp;   // row 0 starts at 0: its list ends with the padding slot
  }
  __syncthreads();
  // this tile's first position in every row
  rp_lds_from_global<true>(rp_hist, a.matrix + (size_t)blockIdx.x * a.n_rows, a.n_rows);
  // the waves' shares: every wave counts the rows of its own kRpTile / 4 occurrences (16-bit counters, two per word) ...
  const uint32_t half = (a.n_rows + 1) / 2;              // words per wave
  uint32_t* wcnt = rp_hist + a.n_rows;                   // [NW][half] words = [NW][2 half] uint16
  for (uint32_t i = threadIdx.x; i < NW * half; i += kBlock) wcnt[i] = 0;
  __syncthreads();
  int key[kRpRounds];
#pragma unroll
  for (int r = 0; r < kRpRounds; ++r) {
    bool pad, bad;
    key[r] = rp_classify(a, rp_occ(tile0, wave, r, lane) < n_occ, id[r], room[r], pad, bad);
    if (key[r] >= 0) atomicAdd(&wcnt[wave * half + (key[r] >> 1)], 1u << (16 * (key[r] & 1)));
  }
  __syncthreads();
  // ... which become the waves' first offsets inside the tile's share of the row
  uint16_t* wrel = reinterpret_cast<uint16_t*>(wcnt);    // [NW][2 half]
  for (uint32_t i = threadIdx.x; i < a.n_rows; i += kBlock) {
    uint32_t run16 = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t c = wrel[(size_t)w * 2 * half + i];
      wrel[(size_t)w * 2 * half + i] = (uint16_t)run16;
      run16 += c;
    }
  }
  __syncthreads();
  // positions in occurrence order: round after round inside the wave, inside a round the lanes that hold the same row find each
  // other with one ballot per key bit (ds operations of a wave are executed in order: no barrier between the rounds)
  uint16_t* wmine = wrel + (size_t)wave * 2 * half;
#pragma unroll 1
  for (int r = 0; r < kRpRounds; ++r) {
    const int k = key[r];
    const bool valid = k >= 0;
    const uint64_t vm = __ballot(valid);
    if (!vm) continue;   // wave-uniform
    uint64_t peers = vm;
    for (int b = 0; b < a.key_bits; ++b) {
      const bool bit = (k >> b) & 1;
      const uint64_t m = __ballot(valid && bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
    const uint32_t cnt = __popcll(peers);
    const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
    uint32_t base = 0;
    if (valid && rank == 0) {
      const uint32_t rel = wmine[k];
      wmine[k] = (uint16_t)(rel + cnt);
      base = rp_hist[k] + rel;
    }
    base = __shfl(base, leader, 64);
    if (valid) a.perm[base + rank] = (uint32_t)rp_occ(tile0, wave, r, lane);
  }
}

// ---- any d (<= 512): one wave per sorted position, lanes stride over the row ----------
constexpr int kGenChunks = 8;

template <int MODE>
__device__ __forceinline__ void apply_row_generic(const SegArgs& a, uint32_t key, int lane,
                                                  const float* acc) {
#pragma unroll
  for (int q = 0; q < kGenChunks; ++q) {
    const int k = lane + 64 * q;
    if (k >= a.d) continue;
    const size_t idx = (size_t)(key - a.key_base) * a.d + k;
    if (MODE == MODE_DENSE_GRAD) {
      a.dense_grad[idx] = acc[q];
      continue;
    }
    float w = a.W[idx], m = 0.f, v = 0.f;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) m = a.M[idx];
    if (MODE == MODE_ADAM) v = a.V[idx];
    opt_elem<MODE>(seg_scalars<MODE>(a), acc[q], w, m, v);
    a.W[idx] = w;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) a.M[idx] = m;
    if (MODE == MODE_ADAM) a.V[idx] = v;
  }
}

__device__ __forceinline__ void occ_grad_generic(const SegArgs& a, int64_t jj, int lane,
                                                 float* acc) {
  const uint32_t o = a.perm[jj] - a.occ_base;
  float c = a.coef ? a.coef[o] : 1.0f;
  int64_t sr = (a.div == 1) ? (int64_t)o : (int64_t)(o / (uint32_t)a.div);
  if (a.src_index) sr = a.src_index[sr];
  const float* s = a.src + (size_t)sr * a.d;
  if (a.src2 && o >= a.n_split) {
    s = a.src2 + (size_t)(o - a.n_split) * a.d;
    c = 1.0f;
  }
#pragma unroll
  for (int q = 0; q < kGenChunks; ++q) {
    const int k = lane + 64 * q;
    if (k < a.d) acc[q] += c * s[k];
  }
}

// generic path: position-indexed, segments of any length summed sequentially (fall-back)
template <int MODE>
__global__ __launch_bounds__(kBlock) void seg_update_generic_kernel(SegArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (j >= a.n_occ) return;
  const uint32_t key = a.keys[j];
  if (j > 0 && a.keys[j - 1] == key) return;
  if (a.skip_single && !(j + 1 < a.n_occ && a.keys[j + 1] == key)) return;
  float acc[kGenChunks];
#pragma unroll
  for (int q = 0; q < kGenChunks; ++q) acc[q] = 0.f;
  for (int64_t jj = j; jj < a.n_occ && a.keys[jj] == key; ++jj) occ_grad_generic(a, jj, lane, acc);
  apply_row_generic<MODE>(a, key, lane, acc);
}

// narrow rows (d <= 4, e.g. the [vocab, 1] first-order tables of FM-family models, whose few
// rows each collect thousands of occurrences): lanes stride over the SEGMENT, then one wave
// all-reduce per element.  Fixed order -> deterministic.
template <int MODE>
__global__ __launch_bounds__(kBlock) void seg_update_narrow_kernel(SegArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (j >= a.n_occ) return;
  const uint32_t key = a.keys[j];
  if (j > 0 && a.keys[j - 1] == key) return;
  if (a.skip_single && !(j + 1 < a.n_occ && a.keys[j + 1] == key)) return;
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t jj = j + lane; jj < a.n_occ && a.keys[jj] == key; jj += 64) {
    const uint32_t o = a.perm[jj] - a.occ_base;
    float c = a.coef ? a.coef[o] : 1.0f;
    int64_t sr = (a.div == 1) ? (int64_t)o : (int64_t)(o / (uint32_t)a.div);
    if (a.src_index) sr = a.src_index[sr];
    const float* s = a.src + (size_t)sr * a.d;
    if (a.src2 && o >= a.n_split) {
      s = a.src2 + (size_t)(o - a.n_split) * a.d;
      c = 1.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < a.d) part[k] += c * s[k];
  }
  float acc[kGenChunks];
#pragma unroll
  for (int q = 0; q < kGenChunks; ++q) acc[q] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = wave_allreduce_sum(part[k]);
    if (lane == k) acc[0] = t;
  }
  apply_row_generic<MODE>(a, key, lane, acc);
}

// ---- narrow rows, dense gradient, any skew: tiles of sorted positions instead of one wave per segment ---------------------------
// seg_update_narrow_kernel gives a segment to ONE wave: the seven weekdays of a CTR batch of 131,072 rows are seven segments of
// 18,700 occurrences, each walked by one wave 64 positions (three dependent loads) at a time -- 306 us for 1 M occurrences of a
// [vocab, 1] table, all of it the tail of those waves.  Here a wave owns a TILE of 512 consecutive sorted positions whatever
// the segments: eight rounds of 64 positions, all loads of the tile in flight at once, a segmented scan over the lanes (keys are
// sorted: equal keys are neighbours) with the running sum carried from round to round.  A segment that lies inside the tile is
// written to the gradient at once; the piece of a segment that crosses the tile's border goes to one of the tile's two slots --
// `head`: the segment came in from the left (it may also leave to the right), `tail`: it starts here and leaves to the right --
// and seg_narrow_chains_kernel adds a segment's pieces (its tail slot, then the head slots of the following tiles) in a fixed
// order.  No atomics, every sum in a fixed order.
constexpr int kNtRounds = 8;
constexpr int kNtTile = 64 * kNtRounds;
struct NarrowSlot { uint32_t key, flag; float v[4]; };   // flag: 0 empty, 1 the segment ends in this tile, 2 it goes on

__device__ __forceinline__ void narrow_occ_val(const SegArgs& a, uint32_t o, float (&v)[4]) {
  o -= a.occ_base;
  float c = a.coef ? a.coef[o] : 1.0f;
  int64_t sr = (a.div == 1) ? (int64_t)o : (int64_t)(o / (uint32_t)a.div);
  if (a.src_index) sr = a.src_index[sr];
  const float* src = a.src + (size_t)sr * a.d;
  if (a.src2 && o >= a.n_split) {
    src = a.src2 + (size_t)(o - a.n_split) * a.d;
    c = 1.0f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = k < a.d ? c * src[k] : 0.f;
}

__global__ __launch_bounds__(kBlock) void seg_narrow_tiles_kernel(SegArgs a, NarrowSlot* __restrict__ slots) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t j0 = w * kNtTile, n = a.n_occ;
  if (j0 >= n) return;   // wave-uniform
  // the tile's keys, the keys one position to the right, and the gradient rows: everything in flight at once
  uint32_t key[kNtRounds], knext[kNtRounds], occ[kNtRounds];
  float val[kNtRounds][4];
#pragma unroll
  for (int r = 0; r < kNtRounds; ++r) {
    const int64_t j = j0 + 64 * r + lane;
    const int64_t jc = j < n ? j : n - 1;
    key[r] = a.keys[jc];
    knext[r] = a.keys[jc + 1 < n ? jc + 1 : jc];
    occ[r] = a.perm[jc];
  }
  const uint32_t first_key = __shfl(key[0], 0, 64);
  const bool first_cont = j0 > 0 && a.keys[j0 - 1] == first_key;
#pragma unroll
  for (int r = 0; r < kNtRounds; ++r) narrow_occ_val(a, occ[r], val[r]);
  const int64_t je = (j0 + kNtTile < n ? j0 + kNtTile : n) - 1;   // the tile's last position
  uint32_t carry_key = 0xFFFFFFFFu;
  float carry[4] = {0.f, 0.f, 0.f, 0.f};
  NarrowSlot head, tail;
  head.flag = tail.flag = 0;
  head.key = tail.key = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) head.v[k] = tail.v[k] = 0.f;
#pragma unroll
  for (int r = 0; r < kNtRounds; ++r) {
    const int64_t j = j0 + 64 * r + lane;
    const bool valid = j < n;
    const uint32_t k_ = valid ? key[r] : 0xFFFFFFFFu;   // (no table has this row: positions past the end form their own segment)
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = valid ? val[r][k] : 0.f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {             // segmented inclusive scan: equal keys are neighbours
      const uint32_t pk = __shfl_up(k_, off, 64);
      float pv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pv[k] = __shfl_up(v[k], off, 64);
      if (lane >= off && pk == k_) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += pv[k];
      }
    }
    if (k_ == carry_key) {                               // the round's first segment goes on from the round before
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += carry[k];
    }
    const bool last_of_tile = valid && j == je;
    const bool goes_on = last_of_tile && je + 1 < n && knext[r] == k_;     // leaves the tile to the right
    const bool ends = valid && !goes_on && (j + 1 >= n || knext[r] != k_);
    const bool from_left = k_ == first_key && first_cont;
    if (ends && !from_left) {                            // a whole segment (or the rest of one that began in this tile)
      for (int k = 0; k < a.d; ++k) a.dense_grad[(size_t)(k_ - a.key_base) * a.d + k] = v[k];
    }
    // at most one lane of the tile ends a segment that came in from the left, at most one is the tile's last position going on
    const uint64_t mh = __ballot((ends && from_left) || (goes_on && from_left));
    const uint64_t mt = __ballot(goes_on && !from_left);
    if (mh) {
      const int src = __ffsll((long long)mh) - 1;
      head.key = __shfl(k_, src, 64);
      head.flag = __shfl(goes_on ? 2u : 1u, src, 64);
#pragma unroll
      for (int k = 0; k < 4; ++k) head.v[k] = __shfl(v[k], src, 64);
    }
    if (mt) {
      const int src = __ffsll((long long)mt) - 1;
      tail.key = __shfl(k_, src, 64);
      tail.flag = 2u;
#pragma unroll
      for (int k = 0; k < 4; ++k) tail.v[k] = __shfl(v[k], src, 64);
    }
    carry_key = __shfl(k_, 63, 64);
#pragma unroll
    for (int k = 0; k < 4; ++k) carry[k] = __shfl(v[k], 63, 64);
  }
  if (lane == 0) {
    slots[2 * w] = head;
    slots[2 * w + 1] = tail;
  }
}

// one wave per tile whose tail slot is set: the segment's pieces are that slot and the head slots of the tiles that follow, up to
// and including the first one where the segment ends
__global__ __launch_bounds__(kBlock) void seg_narrow_chains_kernel(SegArgs a, const NarrowSlot* __restrict__ slots, int64_t n_tiles) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (w >= n_tiles) return;
  const NarrowSlot t = slots[2 * w + 1];
  if (t.flag == 0) return;   // wave-uniform
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t base = w + 1; base < n_tiles; base += 64) {
    const int64_t x = base + lane;
    NarrowSlot h;
    h.flag = 0;
    h.key = 0;
    if (x < n_tiles) h = slots[2 * x];
    const bool mine = x < n_tiles && h.flag != 0 && h.key == t.key;
    const uint64_t m = __ballot(mine);
    const uint64_t done = __ballot(mine && h.flag == 1);
    // the chain is the run of set lanes from lane 0 up to the first that ends the segment
    int len = m == ~0ull ? 64 : __ffsll((long long)~m) - 1;
    if (done) {
      const int e = __ffsll((long long)done) - 1;
      if (e < len) len = e + 1;
    }
    if (lane < len) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sum[k] += h.v[k];
    }
    if (len < 64 || done) break;   // wave-uniform
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sum[k] = wave_allreduce_sum(sum[k]);
  if (lane < a.d) {
    const float total = (lane == 0 ? sum[0] + t.v[0] : lane == 1 ? sum[1] + t.v[1] : lane == 2 ? sum[2] + t.v[2] : sum[3] + t.v[3]);
    a.dense_grad[(size_t)(t.key - a.key_base) * a.d + lane] = total;
  }
}

// ---- launchers ------------------------------------------------------------------------------
static int launch_heads(const uint32_t* keys, const uint32_t* perm, int64_t n, int only_multi,
                        uint8_t* single, uint32_t* heads, uint32_t* n_heads, hipStream_t s) {
  if (heads) RC_HIP(hipMemsetAsync(n_heads, 0, sizeof(uint32_t), s));
  if (single) RC_HIP(hipMemsetAsync(single, 1, (size_t)n, s));
  const int64_t tile = (int64_t)kHeadIters * kBlock;
  const int64_t blocks = (n + tile - 1) / tile;
  hipLaunchKernelGGL(segment_heads_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, keys, perm,
                     n, only_multi, single, heads, n_heads);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int D, int MODE>
static int launch_seg(const SegArgs& a, hipStream_t s) {
  constexpr int GPB = kBlock / (D / 4);
  // upper bound on the number of listed heads: every position, or every second one
  const int64_t max_heads = a.skip_single ? (a.n_occ + 1) / 2 : a.n_occ;
  const int64_t blocks = (max_heads + GPB - 1) / GPB;
  if (blocks > kMaxGridX) return fail(RC_ERR_UNSUPPORTED, "seg_update: grid too large");
#ifndef RC_SEG_X1
  if (a.skip_single) {
    const int64_t blocks2 = (max_heads + kSegHpg * GPB - 1) / (kSegHpg * GPB);
    hipLaunchKernelGGL((seg_update_multi_x2_kernel<D, MODE>), dim3((unsigned)blocks2), dim3(kBlock), 0, s, a);
  }
#else
  if (a.skip_single)
    hipLaunchKernelGGL((seg_update_kernel<D, MODE, true>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
#endif
  else
    hipLaunchKernelGGL((seg_update_kernel<D, MODE, false>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  RC_LAUNCH_CHECK();
  if (a.n_occ > kLongSeg) {  // otherwise no segment can be long
    hipLaunchKernelGGL(long_plan_kernel, dim3(64), dim3(kBlock), 0, s, a);
    RC_LAUNCH_CHECK();
    hipLaunchKernelGGL((long_chunk_kernel<D, MODE>), dim3(1024), dim3(kBlock), 0, s, a);
    RC_LAUNCH_CHECK();
    hipLaunchKernelGGL((long_final_kernel<D, MODE>), dim3(256), dim3(kBlock), 0, s, a);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}

template <int MODE>
static int launch_seg_mode(const SegArgs& a, bool vec_ok, hipStream_t s) {
  if (vec_ok) {
    switch (a.d) {
      case 16: return launch_seg<16, MODE>(a, s);
      case 32: return launch_seg<32, MODE>(a, s);
      case 64: return launch_seg<64, MODE>(a, s);
      case 128: return launch_seg<128, MODE>(a, s);
      case 256: return launch_seg<256, MODE>(a, s);
      default: break;
    }
  }
  if (a.d > 64 * kGenChunks)
    return fail(RC_ERR_UNSUPPORTED, "rc_segmented_update: d=%d > %d", a.d, 64 * kGenChunks);
  const int64_t blocks = (a.n_occ + (kBlock / 64) - 1) / (kBlock / 64);
  if (blocks > kMaxGridX) return fail(RC_ERR_UNSUPPORTED, "seg_update: grid too large");
  if (a.d <= 4 && MODE == MODE_DENSE_GRAD && a.narrow_ws != nullptr && !a.skip_single && !a.pair && a.n_occ >= 4 * kNtTile) {
    const int64_t n_tiles = (a.n_occ + kNtTile - 1) / kNtTile;
    const int64_t wg = (n_tiles + (kBlock / 64) - 1) / (kBlock / 64);
    NarrowSlot* slots = reinterpret_cast<NarrowSlot*>(a.narrow_ws);
    hipLaunchKernelGGL(seg_narrow_tiles_kernel, dim3((unsigned)wg), dim3(kBlock), 0, s, a, slots);
    RC_LAUNCH_CHECK();
    hipLaunchKernelGGL(seg_narrow_chains_kernel, dim3((unsigned)wg), dim3(kBlock), 0, s, a, slots, n_tiles);
  } else if (a.d <= 4)
    hipLaunchKernelGGL((seg_update_narrow_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       a);
  else
    hipLaunchKernelGGL((seg_update_generic_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

static bool vector_kernel_for(int d) { return d == 16 || d == 32 || d == 64 || d == 128 || d == 256; }

struct SegWs {
  uint32_t* counters;
  uint32_t* heads;
  uint32_t* long_list;
  RowInfo* rows;
  ChunkInfo* chunks;
  float* partial;
  uint32_t long_cap, chunk_cap, partial_cap;
  size_t total;
};

static SegWs carve_seg_ws(void* base, int64_t n_occ, int d) {
  Carver cv(base);
  SegWs w;
  w.long_cap = (uint32_t)(n_occ / kLongSeg) + 1;
  w.chunk_cap = (uint32_t)(n_occ / kChunk) + w.long_cap + 1;
  w.partial_cap = 2 * (uint32_t)(n_occ / kChunk) + 2;
  w.counters = cv.take<uint32_t>(CNT_N);
  w.heads = cv.take<uint32_t>((size_t)n_occ);
  w.long_list = cv.take<uint32_t>(w.long_cap);
  w.rows = cv.take<RowInfo>(w.long_cap);
  w.chunks = cv.take<ChunkInfo>(w.chunk_cap);
  w.partial = cv.take<float>((size_t)w.partial_cap * (size_t)d);
  w.total = cv.off;
  return w;
}

}  // namespace rc

using namespace rc;

extern "C" int rc_segment_heads(const uint32_t* keys, const uint32_t* perm, int64_t n_occ,
                                int only_multi, uint8_t* single, uint32_t* heads,
                                uint32_t* n_heads, rc_stream_t stream) {
  RC_REQUIRE(n_occ >= 0 && n_occ < ((int64_t)1 << 31), "rc_segment_heads: bad n_occ");
  RC_REQUIRE((heads == nullptr) == (n_heads == nullptr), "rc_segment_heads: heads and n_heads go together");
  if (n_occ == 0) {
    if (n_heads) RC_HIP(hipMemsetAsync(n_heads, 0, sizeof(uint32_t), as_stream(stream)));
    return RC_OK;
  }
  RC_REQUIRE(keys && perm, "rc_segment_heads: null pointer");
  RC_REQUIRE(single || heads, "rc_segment_heads: nothing to compute");
  return launch_heads(keys, perm, n_occ, only_multi, single, heads, n_heads, as_stream(stream));
}

extern "C" size_t rc_segmented_workspace_bytes(int64_t n_occ, int d) {
  if (n_occ < 1) n_occ = 1;
  if (d < 1) d = 1;
  return carve_seg_ws(nullptr, n_occ, d).total;
}

extern "C" int rc_segmented_update(float* W, float* m, float* v, int d, const uint32_t* keys,
                                   const uint32_t* perm, int64_t n_occ, const float* coef,
                                   const float* src, const int64_t* src_index, int div,
                                   const rc_opt_hyper* h, float* dense_grad,
                                   const uint32_t* heads, const uint32_t* n_heads, int flags,
                                   void* ws, size_t ws_bytes, rc_stream_t stream) {
  return rc_segmented_update2(W, m, v, d, keys, perm, n_occ, coef, src, src_index, div, nullptr, n_occ, 0, 0,
                              h, dense_grad, heads, n_heads, flags, ws, ws_bytes, stream);
}

extern "C" int rc_segmented_update2(float* W, float* m, float* v, int d, const uint32_t* keys,
                                    const uint32_t* perm, int64_t n_occ, const float* coef,
                                    const float* src, const int64_t* src_index, int div,
                                    const float* src2, int64_t n_split, int64_t key_base,
                                    int64_t occ_base, const rc_opt_hyper* h, float* dense_grad,
                                    const uint32_t* heads, const uint32_t* n_heads, int flags,
                                    void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (n_occ == 0) return RC_OK;
  RC_REQUIRE(n_split >= 0 && n_split <= n_occ, "rc_segmented_update2: n_split out of range");
  RC_REQUIRE(key_base >= 0 && occ_base >= 0 && key_base < ((int64_t)1 << 32) && occ_base < ((int64_t)1 << 31),
             "rc_segmented_update2: key_base / occ_base out of range");
  RC_REQUIRE(keys && perm && src && ws, "rc_segmented_update: null pointer");
  RC_REQUIRE(d >= 1 && div >= 1 && n_occ > 0 && n_occ < ((int64_t)1 << 31),
             "rc_segmented_update: bad shape d=%d div=%d n_occ=%lld", d, div, (long long)n_occ);
  RC_REQUIRE(dense_grad != nullptr || W != nullptr, "rc_segmented_update: no output (W or dense_grad)");
  RC_REQUIRE((heads == nullptr) == (n_heads == nullptr),
             "rc_segmented_update: heads and n_heads go together");
  const SegWs w = carve_seg_ws(ws, n_occ, d);
  if (ws_bytes < w.total)
    return fail(RC_ERR_WORKSPACE, "rc_segmented_update: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t s = as_stream(stream);
  SegArgs a;
  memset(&a, 0, sizeof(a));
  a.W = W; a.M = m; a.V = v;
  a.keys = keys; a.perm = perm; a.n_occ = n_occ;
  a.coef = coef; a.src = src; a.src_index = src_index; a.div = div; a.d = d;
  a.src2 = src2; a.n_split = (uint32_t)n_split;
  a.key_base = (uint32_t)key_base; a.occ_base = (uint32_t)occ_base;
  a.dense_grad = dense_grad;
  a.skip_single = (flags & RC_SEG_SKIP_SINGLETONS) ? 1 : 0;
  a.counters = w.counters; a.long_list = w.long_list; a.rows = w.rows; a.chunks = w.chunks;
  a.partial = w.partial;
  a.long_cap = w.long_cap; a.chunk_cap = w.chunk_cap; a.partial_cap = w.partial_cap;
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  bool vec_ok = vector_kernel_for(d) && al(src) && al(src2);
  int mode = MODE_DENSE_GRAD;
  if (dense_grad) {
    vec_ok = vec_ok && al(dense_grad);
  } else {
    RC_TRY(fill_opt_scalars(h, &a.o));
    mode = mode_of(h);
    RC_REQUIRE(mode != MODE_ADAM || (m && v), "rc_segmented_update: Adam needs m and v");
    RC_REQUIRE(mode != MODE_ADAGRAD || m, "rc_segmented_update: Adagrad needs m (state_sum)");
    vec_ok = vec_ok && al(W) && al(m) && al(v);
  }
  RC_HIP(hipMemsetAsync(w.counters, 0, CNT_N * sizeof(uint32_t), s));
  // (off the vector route the head list's room -- n_occ words -- is free: 2 slots of 6 words per 512 positions fit)
  if (!vec_ok && n_occ >= 4 * kNtTile) a.narrow_ws = w.heads;
  if (vec_ok) {
    if (heads) {
      a.heads = heads;
      a.n_heads = n_heads;
    } else {  // build the compact head list here
      RC_TRY(launch_heads(keys, perm, n_occ, a.skip_single, nullptr, w.heads,
                          &w.counters[CNT_HEADS], s));
      a.heads = w.heads;
      a.n_heads = &w.counters[CNT_HEADS];
    }
  }
  switch (mode) {
    case MODE_DENSE_GRAD: return launch_seg_mode<MODE_DENSE_GRAD>(a, vec_ok, s);
    case MODE_SGD: return launch_seg_mode<MODE_SGD>(a, vec_ok, s);
    case MODE_ADAM: return launch_seg_mode<MODE_ADAM>(a, vec_ok, s);
    default: return launch_seg_mode<MODE_ADAGRAD>(a, vec_ok, s);
  }
}

extern "C" size_t rc_segmented_rows_workspace_bytes(int64_t n_rows, int64_t n_occ, int d) {
  if (n_rows < 1) n_rows = 1;
  return rc_segmented_workspace_bytes(n_occ, d) + align_up((2 * (size_t)n_rows + 64) * sizeof(uint32_t), 256) + 256;
}

// rc_segmented_update2 for a table of n_rows rows that collect many occurrences each (section 4 above); keys / perm
// from a plain rc_sort_ids.  Same gradient sources and outputs; d in {16, 32, 64, 128, 256}, 16-byte aligned buffers.
static int segmented_update_rows(float* W, float* m, float* v, int d, int64_t n_rows, const uint32_t* keys,
                                 const uint32_t* perm, int64_t n_occ, const float* coef, const float* src,
                                 const int64_t* src_index, int div, const float* src2, int64_t n_split,
                                 const rc_opt_hyper* h, const int64_t* step_dev, float* dense_grad, void* ws, size_t ws_bytes,
                                 rc_stream_t stream) {
  if (n_occ == 0) return RC_OK;
  RC_REQUIRE(n_split >= 0 && n_split <= n_occ, "rc_segmented_update_rows: n_split out of range");
  RC_REQUIRE(keys && perm && src && ws, "rc_segmented_update_rows: null pointer");
  RC_REQUIRE(div >= 1 && n_occ > 0 && n_occ < ((int64_t)1 << 31) && n_rows >= 1 && n_rows < ((int64_t)1 << 31),
             "rc_segmented_update_rows: bad shape div=%d n_occ=%lld n_rows=%lld", div, (long long)n_occ, (long long)n_rows);
  RC_REQUIRE(dense_grad != nullptr || W != nullptr, "rc_segmented_update_rows: no output (W or dense_grad)");
  if (!vector_kernel_for(d)) return fail(RC_ERR_UNSUPPORTED, "rc_segmented_update_rows: d=%d (16/32/64/128/256)", d);
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  RC_REQUIRE(al(src) && al(src2) && al(W) && al(m) && al(v) && al(dense_grad), "rc_segmented_update_rows: buffers must be 16-byte aligned");
  if (ws_bytes < rc_segmented_rows_workspace_bytes(n_rows, n_occ, d))
    return fail(RC_ERR_WORKSPACE, "rc_segmented_update_rows: workspace %zu < %zu", ws_bytes,
                rc_segmented_rows_workspace_bytes(n_rows, n_occ, d));
  const SegWs w = carve_seg_ws(ws, n_occ, d);
  // [counters (64 words) | start | end]: zeroed by ONE fill (the counters of the long-row pass sit beside the bounds)
  uint32_t* counters = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + align_up(w.total, 256));
  uint32_t* start = counters + 64;
  uint32_t* end = start + n_rows;
  hipStream_t s = as_stream(stream);
  SegArgs a;
  memset(&a, 0, sizeof(a));
  a.W = W; a.M = m; a.V = v;
  a.keys = keys; a.perm = perm; a.n_occ = n_occ;
  a.coef = coef; a.src = src; a.src_index = src_index; a.div = div; a.d = d;
  a.src2 = src2; a.n_split = (uint32_t)n_split;
  a.dense_grad = dense_grad;
  a.counters = counters; a.long_list = w.long_list; a.rows = w.rows; a.chunks = w.chunks;
  a.partial = w.partial;
  a.long_cap = w.long_cap; a.chunk_cap = w.chunk_cap; a.partial_cap = w.partial_cap;
  int mode = MODE_DENSE_GRAD;
  if (!dense_grad) {
    RC_TRY(fill_opt_scalars(h, &a.o));
    mode = mode_of(h);
    RC_REQUIRE(mode != MODE_ADAM || (m && v), "rc_segmented_update_rows: Adam needs m and v");
    RC_REQUIRE(mode != MODE_ADAGRAD || m, "rc_segmented_update_rows: Adagrad needs m (state_sum)");
    if (mode == MODE_ADAM && step_dev) {
      a.step_dev = step_dev; a.beta1 = h->beta1; a.beta2 = h->beta2; a.lr = h->lr;
    }
  }
  static_assert(CNT_N <= 64, "counters beside the bounds");
  RC_HIP(hipMemsetAsync(counters, 0, (64 + 2 * (size_t)n_rows) * sizeof(uint32_t), s));   // absent rows: start = end = 0
  hipLaunchKernelGGL(segment_bounds_kernel, dim3((unsigned)((n_occ + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, keys, n_occ, 0u,
                     (uint32_t)n_rows, start, end);
  RC_LAUNCH_CHECK();
  switch (mode) {
    case MODE_DENSE_GRAD: return launch_seg_rows_mode<MODE_DENSE_GRAD>(a, start, end, (uint32_t)n_rows, s);
    case MODE_SGD: return launch_seg_rows_mode<MODE_SGD>(a, start, end, (uint32_t)n_rows, s);
    case MODE_ADAM: return launch_seg_rows_mode<MODE_ADAM>(a, start, end, (uint32_t)n_rows, s);
    default: return launch_seg_rows_mode<MODE_ADAGRAD>(a, start, end, (uint32_t)n_rows, s);
  }
}

extern "C" int rc_segmented_update_rows(float* W, float* m, float* v, int d, int64_t n_rows, const uint32_t* keys,
                                        const uint32_t* perm, int64_t n_occ, const float* coef, const float* src,
                                        const int64_t* src_index, int div, const float* src2, int64_t n_split,
                                        const rc_opt_hyper* h, float* dense_grad, void* ws, size_t ws_bytes,
                                        rc_stream_t stream) {
  return segmented_update_rows(W, m, v, d, n_rows, keys, perm, n_occ, coef, src, src_index, div, src2, n_split, h, nullptr,
                               dense_grad, ws, ws_bytes, stream);
}

// The same with Adam's step count in device memory (step_dev[0] >= 1 when the kernels run; h->step is not used): the launch can be
// captured in a hipGraph and replayed while the host advances the counter with rc_step_increment.  Other optimizers: as above.
extern "C" int rc_segmented_update_rows_dev(float* W, float* m, float* v, int d, int64_t n_rows, const uint32_t* keys,
                                            const uint32_t* perm, int64_t n_occ, const float* coef, const float* src,
                                            const int64_t* src_index, int div, const float* src2, int64_t n_split,
                                            const rc_opt_hyper* h, const int64_t* step_dev, float* dense_grad, void* ws,
                                            size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(step_dev != nullptr, "rc_segmented_update_rows_dev: step_dev is null");
  return segmented_update_rows(W, m, v, d, n_rows, keys, perm, n_occ, coef, src, src_index, div, src2, n_split, h, step_dev,
                               dense_grad, ws, ws_bytes, stream);
}

// ---- rows route from a counting sort (section 5) -----------------------------------------------------------------------------
struct RowsPlanWs {
  SegWs seg;
  uint32_t *counters, *start, *end, *keys, *perm, *tile_pad, *totals, *matrix, *status;
  uint32_t n_tiles;
  size_t total;
};

static RowsPlanWs carve_rows_plan_ws(void* base, int64_t n_rows, int64_t n_occ, int d) {
  RowsPlanWs w;
  // [status (256 B: at the front, where a workspace that is reused for another shape finds it again) | SegWs | the plan]
  w.status = reinterpret_cast<uint32_t*>(base);
  w.seg = carve_seg_ws(base ? static_cast<char*>(base) + 256 : nullptr, n_occ, d);
  Carver cv(base);
  cv.off = 256 + align_up(w.seg.total, 256);
  w.n_tiles = (uint32_t)((n_occ + kRpTile - 1) / kRpTile);
  w.counters = cv.take<uint32_t>(64);
  w.start = cv.take<uint32_t>((size_t)n_rows);
  w.end = cv.take<uint32_t>((size_t)n_rows);
  w.keys = cv.take<uint32_t>((size_t)n_occ + 1);
  w.perm = cv.take<uint32_t>((size_t)n_occ + 1);
  w.tile_pad = cv.take<uint32_t>(w.n_tiles);
  w.totals = cv.take<uint32_t>((size_t)n_rows);
  w.matrix = cv.take<uint32_t>((size_t)w.n_tiles * (size_t)n_rows);
  w.total = cv.off;
  return w;
}

extern "C" int rc_rows_plan_supported(int64_t n_rows, int64_t n_occ, int d) {
  return (n_rows >= 1 && n_rows <= kRpMaxRows && n_occ >= 1 && n_occ < ((int64_t)1 << 31) && vector_kernel_for(d) &&
          ((n_occ + kRpTile - 1) / kRpTile) * n_rows < ((int64_t)1 << 28)) ? 1 : 0;
}

extern "C" size_t rc_rows_plan_workspace_bytes(int64_t n_rows, int64_t n_occ, int d) {
  if (!rc_rows_plan_supported(n_rows, n_occ, d)) return 0;
  return carve_rows_plan_ws(nullptr, n_rows, n_occ, d).total + 256;
}

extern "C" int rc_rows_plan_build(const int64_t* ids_a, int64_t n_a, const int64_t* ids_b, int64_t n_b, const int64_t* lengths_b,
                                  int L_b, int64_t n_rows, int d, void* ws, size_t ws_bytes, rc_stream_t stream) {
  const int64_t n_occ = n_a + n_b;
  RC_REQUIRE(n_a >= 0 && n_b >= 0 && (n_a == 0 || ids_a) && (n_b == 0 || ids_b) && ws, "rc_rows_plan_build: null pointer / negative count");
  RC_REQUIRE(!lengths_b || (L_b >= 1 && n_b % L_b == 0), "rc_rows_plan_build: lengths_b needs ids_b as [n_b / L_b, L_b] (L_b=%d)", L_b);
  if (!rc_rows_plan_supported(n_rows, n_occ, d))
    return fail(RC_ERR_UNSUPPORTED, "rc_rows_plan_build: n_rows=%lld (1..%d), n_occ=%lld, d=%d (16/32/64/128/256)", (long long)n_rows,
                kRpMaxRows, (long long)n_occ, d);
  const RowsPlanWs w = carve_rows_plan_ws(ws, n_rows, n_occ, d);
  if (ws_bytes < w.total) return fail(RC_ERR_WORKSPACE, "rc_rows_plan_build: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t s = as_stream(stream);
  RowsPlanArgs a;
  memset(&a, 0, sizeof(a));
  a.ids_a = ids_a; a.ids_b = ids_b; a.len_b = n_b ? lengths_b : nullptr; a.n_a = n_a; a.n_b = n_b; a.L_b = a.len_b ? L_b : 1;
  a.n_rows = (uint32_t)n_rows; a.n_tiles = w.n_tiles;
  a.key_bits = 1;
  while (((int64_t)1 << a.key_bits) < n_rows) ++a.key_bits;
  a.matrix = w.matrix; a.totals = w.totals; a.tile_pad = w.tile_pad; a.start = w.start; a.end = w.end; a.keys = w.keys; a.perm = w.perm;
  a.counters = w.counters; a.rows = w.seg.rows; a.chunks = w.seg.chunks; a.long_cap = w.seg.long_cap; a.chunk_cap = w.seg.chunk_cap;
  a.status = w.status;
  const size_t lds = (size_t)n_rows * sizeof(uint32_t);
  const size_t lds_scatter = lds + (kBlock / 64) * (size_t)((n_rows + 1) / 2) * sizeof(uint32_t);
  // (per device: hipFuncSetAttribute acts on the current device's copy of the function)
  static std::mutex mu;
  static bool attr_done[64] = {};
  int dev = 0;
  RC_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rp_count_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kRpMaxRows * 4));
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rp_count_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kRpMaxRows * 4));
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rp_scatter_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kRpMaxRows * 12));
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rp_scatter_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kRpMaxRows * 12));
      attr_done[dev] = true;
    }
  }
  if (a.len_b) hipLaunchKernelGGL(rp_count_kernel<true>, dim3(w.n_tiles), dim3(kBlock), lds, s, a);
  else hipLaunchKernelGGL(rp_count_kernel<false>, dim3(w.n_tiles), dim3(kBlock), lds, s, a);
  RC_LAUNCH_CHECK();
  hipLaunchKernelGGL(rp_prefix_kernel, dim3((unsigned)((n_rows + 63) / 64)), dim3(64), 0, s, a);
  RC_LAUNCH_CHECK();
  if (a.len_b) hipLaunchKernelGGL(rp_scatter_kernel<true>, dim3(w.n_tiles), dim3(kBlock), lds_scatter, s, a);
  else hipLaunchKernelGGL(rp_scatter_kernel<false>, dim3(w.n_tiles), dim3(kBlock), lds_scatter, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// device pointers into a built plan (tests, and callers that want the grouping itself): perm [n_live], start / end [n_rows] (keys:
// only keys[start[r]] of the hot rows is written), status [1] = ids outside the table since the caller zeroed the workspace
extern "C" int rc_rows_plan_views(void* ws, int64_t n_rows, int64_t n_occ, int d, const uint32_t** keys, const uint32_t** perm,
                                  const uint32_t** start, const uint32_t** end, const uint32_t** status) {
  RC_REQUIRE(ws && rc_rows_plan_supported(n_rows, n_occ, d), "rc_rows_plan_views: bad arguments");
  const RowsPlanWs w = carve_rows_plan_ws(ws, n_rows, n_occ, d);
  if (keys) *keys = w.keys;
  if (perm) *perm = w.perm;
  if (start) *start = w.start;
  if (end) *end = w.end;
  if (status) *status = w.status;
  return RC_OK;
}

// rc_segmented_update_rows on a plan of rc_rows_plan_build (same ws, same n_rows / n_occ / d; occurrence o < n_split = position o
// of ids_a, the others position o - n_split of ids_b): two launches, nothing to zero.  step_dev as rc_segmented_update_rows_dev
// (null: h->step).
extern "C" int rc_rows_plan_update(float* W, float* m, float* v, int d, int64_t n_rows, int64_t n_occ, const float* coef,
                                   const float* src, const int64_t* src_index, int div, const float* src2, int64_t n_split,
                                   const rc_opt_hyper* h, const int64_t* step_dev, float* dense_grad, void* ws, size_t ws_bytes,
                                   rc_stream_t stream) {
  RC_REQUIRE(src && ws && div >= 1 && n_split >= 0 && n_split <= n_occ, "rc_rows_plan_update: bad arguments");
  RC_REQUIRE(dense_grad != nullptr || W != nullptr, "rc_rows_plan_update: no output (W or dense_grad)");
  if (!rc_rows_plan_supported(n_rows, n_occ, d)) return fail(RC_ERR_UNSUPPORTED, "rc_rows_plan_update: shape not supported");
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  RC_REQUIRE(al(src) && al(src2) && al(W) && al(m) && al(v) && al(dense_grad), "rc_rows_plan_update: buffers must be 16-byte aligned");
  const RowsPlanWs w = carve_rows_plan_ws(ws, n_rows, n_occ, d);
  if (ws_bytes < w.total) return fail(RC_ERR_WORKSPACE, "rc_rows_plan_update: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t s = as_stream(stream);
  SegArgs a;
  memset(&a, 0, sizeof(a));
  a.W = W; a.M = m; a.V = v;
  a.keys = w.keys; a.perm = w.perm; a.n_occ = n_occ;
  a.coef = coef; a.src = src; a.src_index = src_index; a.div = div; a.d = d;
  a.src2 = src2; a.n_split = (uint32_t)n_split;
  a.dense_grad = dense_grad;
  a.counters = w.counters; a.long_list = w.seg.long_list; a.rows = w.seg.rows; a.chunks = w.seg.chunks;
  a.partial = w.seg.partial;
  a.long_cap = w.seg.long_cap; a.chunk_cap = w.seg.chunk_cap; a.partial_cap = w.seg.partial_cap;
  a.planned = 1;
  int mode = MODE_DENSE_GRAD;
  if (!dense_grad) {
    RC_TRY(fill_opt_scalars(h, &a.o));
    mode = mode_of(h);
    RC_REQUIRE(mode != MODE_ADAM || (m && v), "rc_rows_plan_update: Adam needs m and v");
    RC_REQUIRE(mode != MODE_ADAGRAD || m, "rc_rows_plan_update: Adagrad needs m (state_sum)");
    if (mode == MODE_ADAM && step_dev) {
      a.step_dev = step_dev; a.beta1 = h->beta1; a.beta2 = h->beta2; a.lr = h->lr;
    }
  }
  switch (mode) {
    case MODE_DENSE_GRAD: return launch_seg_planned_mode<MODE_DENSE_GRAD>(a, w.start, w.end, (uint32_t)n_rows, s);
    case MODE_SGD: return launch_seg_planned_mode<MODE_SGD>(a, w.start, w.end, (uint32_t)n_rows, s);
    case MODE_ADAM: return launch_seg_planned_mode<MODE_ADAM>(a, w.start, w.end, (uint32_t)n_rows, s);
    default: return launch_seg_planned_mode<MODE_ADAGRAD>(a, w.start, w.end, (uint32_t)n_rows, s);
  }
}

// Two tables that share their ids (NeuMF's mf / mlp embedding of a user or an item: models/general/NeuMF.py:37-40
// looks both up with the same index tensor) updated in ONE pass over keys / perm / heads: the row of width 2 d is
// the concatenation [table a | table b], every kernel of rc_segmented_update runs unchanged on it.  src_a / src_b:
// per-occurrence gradient rows [n_occ, d].  Either both dense_grad_* (dense gradients out) or W_* (+ m, v) with h.
extern "C" int rc_segmented_update_pair(float* W_a, float* m_a, float* v_a, float* W_b, float* m_b, float* v_b, int d,
                                        const uint32_t* keys, const uint32_t* perm, int64_t n_occ, const float* src_a,
                                        const float* src_b, const rc_opt_hyper* h, float* dense_grad_a,
                                        float* dense_grad_b, const uint32_t* heads, const uint32_t* n_heads, void* ws,
                                        size_t ws_bytes, rc_stream_t stream) {
  if (n_occ == 0) return RC_OK;
  RC_REQUIRE(keys && perm && src_a && src_b && ws, "rc_segmented_update_pair: null pointer");
  RC_REQUIRE(n_occ > 0 && n_occ < ((int64_t)1 << 31), "rc_segmented_update_pair: bad n_occ=%lld", (long long)n_occ);
  RC_REQUIRE((dense_grad_a != nullptr) == (dense_grad_b != nullptr), "rc_segmented_update_pair: give both dense gradients or none");
  RC_REQUIRE(dense_grad_a != nullptr || (W_a != nullptr && W_b != nullptr), "rc_segmented_update_pair: no output");
  RC_REQUIRE((heads == nullptr) == (n_heads == nullptr), "rc_segmented_update_pair: heads and n_heads go together");
  if (!vector_kernel_for(2 * d))
    return fail(RC_ERR_UNSUPPORTED, "rc_segmented_update_pair: d=%d (2 d must be 16/32/64/128/256)", d);
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  RC_REQUIRE(al(W_a) && al(W_b) && al(m_a) && al(m_b) && al(v_a) && al(v_b) && al(src_a) && al(src_b) && al(dense_grad_a) &&
                 al(dense_grad_b), "rc_segmented_update_pair: buffers must be 16-byte aligned");
  const SegWs w = carve_seg_ws(ws, n_occ, 2 * d);
  if (ws_bytes < w.total) return fail(RC_ERR_WORKSPACE, "rc_segmented_update_pair: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t s = as_stream(stream);
  SegArgs a;
  memset(&a, 0, sizeof(a));
  a.pair = 1;
  a.W = W_a; a.M = m_a; a.V = v_a; a.Wb = W_b; a.Mb = m_b; a.Vb = v_b;
  a.keys = keys; a.perm = perm; a.n_occ = n_occ; a.src = src_a; a.srcb = src_b; a.div = 1; a.d = 2 * d;
  a.n_split = (uint32_t)n_occ;
  a.dense_grad = dense_grad_a; a.dense_grad_b = dense_grad_b;
  a.counters = w.counters; a.long_list = w.long_list; a.rows = w.rows; a.chunks = w.chunks; a.partial = w.partial;
  a.long_cap = w.long_cap; a.chunk_cap = w.chunk_cap; a.partial_cap = w.partial_cap;
  int mode = MODE_DENSE_GRAD;
  if (!dense_grad_a) {
    RC_TRY(fill_opt_scalars(h, &a.o));
    mode = mode_of(h);
    RC_REQUIRE(mode != MODE_ADAM || (m_a && v_a && m_b && v_b), "rc_segmented_update_pair: Adam needs m and v");
    RC_REQUIRE(mode != MODE_ADAGRAD || (m_a && m_b), "rc_segmented_update_pair: Adagrad needs m (state_sum)");
  }
  RC_HIP(hipMemsetAsync(w.counters, 0, CNT_N * sizeof(uint32_t), s));
  if (heads) {
    a.heads = heads;
    a.n_heads = n_heads;
  } else {
    RC_TRY(launch_heads(keys, perm, n_occ, 0, nullptr, w.heads, &w.counters[CNT_HEADS], s));
    a.heads = w.heads;
    a.n_heads = &w.counters[CNT_HEADS];
  }
  switch (mode) {
    case MODE_DENSE_GRAD: return launch_seg_mode<MODE_DENSE_GRAD>(a, true, s);
    case MODE_SGD: return launch_seg_mode<MODE_SGD>(a, true, s);
    case MODE_ADAM: return launch_seg_mode<MODE_ADAM>(a, true, s);
    default: return launch_seg_mode<MODE_ADAGRAD>(a, true, s);
  }
}
