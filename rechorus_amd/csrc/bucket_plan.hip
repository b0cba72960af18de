// bucket_plan.hip -- groups the occurrences of a batch's row ids by row, without a device-wide sort.
//
// What it replaces in the reference: aten::embedding_dense_backward's index_add over every occurrence
// (reached through loss.backward(), helpers/BaseRunner.py:205, for the nn.Embedding tables of
// models/general/BPRMF.py:31-32).  The atomic-free segmented update (plan_update.hip) needs, per distinct
// row, the list of batch positions that touch it, in ascending position order (fixed summation order =>
// bit-reproducible gradients).  Round 1 got that from a stable LSD radix sort of all B*(1+K) ids (rocPRIM,
// three 8-bit passes + histogram + copy-back = 0.25 ms of a 1.33 ms step) followed by a head-marking pass.
// Ids are small integers (< n_rows), so a full sort is more than is needed:
//
//   1. plan_count     per tile of 8,192 positions: LDS histogram over id ranges ("buckets" of 2^shift ids)
//   2. plan_colscan   per bucket: exclusive scan of its tile counts (one wave per bucket)
//   3. plan_scatter   stable partition of (id, position) into the buckets: ranks inside a tile come from
//                     wave-level ballot matching in position order, tile offsets from step 2 -- no atomics
//                     in the placement, so a bucket holds its keys in ascending position
//   4. plan_bucket    one wave per bucket: an LDS table indexed by (id - bucket start) counts the bucket's
//                     ids, a scan turns counts into cursors, and a second pass over the bucket's keys drops
//                     each position into its row's slot range -- stable again, so every row's positions come
//                     out ascending.  Emits rc_plan_row {row, start, n} records, the grouped positions
//                     occ[], and the single-occurrence flags the fused BPRMF kernel consumes.
//
// HBM traffic per key: 8 B read twice (ids), 8 B written + read twice (bucketed keys), 4 B written (occ)
// against 3 x (8 B read + 8 B written) + histogram + copy-back + the marking pass of the sort pipeline.
// The only atomics are integer: LDS histogram increments (order-free) and one global list reservation per
// bucket (the ORDER of rows in the list influences no floating-point result).
#include "plan.hpp"

namespace rc {

int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

PlanGeom plan_geometry(int64_t n_a, int64_t n_b, int64_t range_a, int64_t range_b, int want_mode) {
  PlanGeom g;
  memset(&g, 0, sizeof(g));
  g.n = n_a + n_b;
  if (g.n <= 0 || g.n >= ((int64_t)1 << 31) || range_a < 1 || (n_b > 0 && range_b < 1)) return g;
  if (n_b == 0) range_b = 0;
  auto buckets = [&](int shift, uint32_t* na, uint32_t* nbb) {
    const int64_t w = (int64_t)1 << shift;
    *na = (uint32_t)((range_a + w - 1) / w);
    *nbb = (uint32_t)((range_b + w - 1) / w);
    return (int64_t)*na + *nbb;
  };
  uint32_t na, nbb;
  g.tiles = (uint32_t)((g.n + kPlanTile - 1) / kPlanTile);
  // Direct buckets cost table cells in proportion to the id RANGE (every bucket zeroes and scans 8,192 cells), hashed
  // buckets in proportion to the number of KEYS: hashed where the range is more than 16 cells per key (NeuMF's 0.33 M
  // item lookups over 10 M - 100 M rows; a rank's lookups in a sharded table) or too wide for one direct level at all.
  const bool direct_ok = buckets(kPlanMaxShift, &na, &nbb) <= kPlanMaxBuckets;
  // want_mode 2: the wide id-range geometry or nothing (bitmap callers)
  const int force = want_mode == 2 ? 0 : (want_mode >= 0 ? want_mode : env_int("RC_PLAN_HASHED", -1));
  const bool hashed = force >= 0 ? (force != 0 || !direct_ok) : (!direct_ok || range_a + range_b > 16 * g.n);
  if (hashed) {
    auto pow2_for = [](int64_t n_keys, int* lg) {
      int l = 0;
      while (((int64_t)kPlanHashKeysPerBucket << l) < n_keys && l < 11) ++l;   // <= 2,048 buckets per list
      *lg = l;
      return n_keys > 0 ? (uint32_t)1 << l : 0u;
    };
    g.nb_a = pow2_for(n_a, &g.log_nb_a);
    g.nb_b = pow2_for(n_b, &g.log_nb_b);
    // keys per bucket beyond what the LDS table holds with room to spare (every key distinct in the worst case)
    if ((n_a >> g.log_nb_a) > (int64_t)kPlanHashSlots / 2 || (n_b >> g.log_nb_b) > (int64_t)kPlanHashSlots / 2) return g;
    if (range_a > ((int64_t)1 << 32) - 2 || range_b > ((int64_t)1 << 32) - 2) return g;
    g.hashed = 1;
    g.nb = g.nb_a + g.nb_b;
    g.shift = 0;
    g.base_b = 0;
    g.bucket_bits = 1;
    while ((1u << g.bucket_bits) < g.nb) ++g.bucket_bits;
    g.ok = 1;
    return g;
  }
  // dense batches (>= 4 keys per id of the tables): as many buckets as one level allows, i.e. as few ids per bucket as
  // possible -- narrow buckets, finished by ballot ranks
  if (want_mode != 2 && env_int("RC_PLAN_NARROW", 1) != 0 && g.n >= 4 * (range_a + range_b)) {
    int sh = 0;
    while (buckets(sh, &na, &nbb) > kPlanMaxBuckets) ++sh;
    if (sh <= kPlanNarrowMaxShift) {
      buckets(sh, &na, &nbb);
      g.narrow = 1;
      g.shift = sh;
      g.nb_a = na; g.nb_b = nbb; g.nb = na + nbb;
      g.base_b = na << sh;
      g.bucket_bits = 1;
      while ((1u << g.bucket_bits) < g.nb) ++g.bucket_bits;
      g.ok = 1;
      return g;
    }
  }
  // enough buckets to fill the chip (one wave per bucket in step 4), but not more than the scatter's
  // write runs can afford: aim at ~8 K keys per bucket, at least 128 buckets
  int64_t want = g.n / 8192;
  if (want < 128) want = 128;
  if (want > 1024) want = 1024;
  int shift = env_int("RC_PLAN_SHIFT", kPlanMaxShift);
  if (shift > kPlanMaxShift) shift = kPlanMaxShift;
  if (shift < kPlanMinShift) shift = kPlanMinShift;
  while (shift > kPlanMinShift && buckets(shift, &na, &nbb) < want && buckets(shift - 1, &na, &nbb) <= kPlanMaxBuckets) --shift;
  while (buckets(shift, &na, &nbb) > kPlanMaxBuckets) ++shift;
  buckets(shift, &na, &nbb);
  g.shift = shift;
  g.nb_a = na;
  g.nb_b = nbb;
  g.nb = na + nbb;
  g.base_b = na << shift;
  g.bucket_bits = 1;
  while ((1u << g.bucket_bits) < g.nb) ++g.bucket_bits;
  g.ok = 1;
  return g;
}

PlanWs carve_plan_ws(void* base, int64_t n) {
  Carver cv(base);
  PlanWs w;
  if (n < 1) n = 1;
  const size_t tiles = (size_t)((n + kPlanTile - 1) / kPlanTile);
  w.hist = cv.take<uint32_t>((size_t)kPlanMaxBuckets * tiles);
  w.totals = cv.take<uint32_t>(kPlanMaxBuckets);
  w.bucket_base = cv.take<uint32_t>(kPlanMaxBuckets + 1);
  w.lid = cv.take<uint16_t>((size_t)n);
  w.pos = cv.take<uint32_t>((size_t)n);
  w.lid32 = cv.take<uint32_t>((size_t)n);
  w.counters = cv.take<uint32_t>(PC_N);
  w.total = cv.off;
  return w;
}

PlanLongWs carve_plan_long_ws(void* base, int64_t n, int d) {
  Carver cv(base);
  PlanLongWs w;
  if (n < 1) n = 1;
  w.long_cap = (uint32_t)(n / (kPlanLongSeg + 1)) + 1;
  w.chunk_cap = (uint32_t)(n / 64) + w.long_cap + 1;   // 64 = the smallest chunk any consumer cuts (plan_update.hip, side_chunk)
  w.lrows = cv.take<PlanLongRow>(w.long_cap);
  w.chunks = cv.take<PlanChunkInfo>(w.chunk_cap);
  w.partial = cv.take<float>((size_t)w.chunk_cap * (size_t)d);
  w.total = cv.off;
  return w;
}

// ---- device helpers -----------------------------------------------------------------------------------
// the 32-bit key of position p.  Direct geometry: list a = the id, list b = base_b + id (one joint range, bucket = key >>
// shift).  Hashed geometry: the id itself; the list follows from p.
// A NEGATIVE id marks a position that takes no part (padding slots of a history window): it yields the "no key" value
// and is neither counted nor placed.
template <bool HASHED>
__device__ __forceinline__ uint32_t plan_key(const PlanArgs& a, uint32_t p) {
  const int64_t id = p < a.n_a ? a.ids_a[p] : a.ids_b[p - a.n_a];
  if (id < 0) return 0xFFFFFFFFu;
  if (HASHED) return (uint32_t)id;
  return p < a.n_a ? (uint32_t)id : a.g.base_b + (uint32_t)id;
}
// bucket of a key (hashed: `side_b` says which list the position belongs to); *bad: an id outside its table (direct only)
template <bool HASHED>
__device__ __forceinline__ uint32_t plan_bucket_of(const PlanArgs& a, uint32_t key, bool side_b, bool* bad) {
  if (HASHED) {
    // (the bucket of a hashed key is always in range; the id itself may still lie outside its table -- flagged like the direct
    //  geometry's, so that PC_STATUS tells a caller that nn.Embedding would have raised)
    if (bad && (int64_t)key >= (side_b ? a.range_b : a.range_a)) *bad = true;
    const uint32_t h = key * 0x9E3779B1u;
    const int lg = side_b ? a.g.log_nb_b : a.g.log_nb_a;
    const uint32_t b = lg ? h >> (32 - lg) : 0u;
    return side_b ? a.g.nb_a + b : b;
  }
  uint32_t b = key >> a.g.shift;
  if (b >= a.g.nb) {  // memory-safe: counted in the last bucket, flagged by the scatter
    b = a.g.nb - 1;
    if (bad) *bad = true;
  }
  // an id beyond its table that still falls into the last bucket's range (or into the other list's buckets) is flagged too
  if (bad && (side_b ? (int64_t)key - (int64_t)a.g.base_b >= a.range_b || key < a.g.base_b : (int64_t)key >= a.range_a)) *bad = true;
  return b;
}

// lanes of the wave (among `valid`) whose v equals this lane's v; v < 2^bits
__device__ __forceinline__ uint64_t match_lanes(uint32_t v, int bits, uint64_t valid) {
  uint64_t m = valid;
  for (int b = 0; b < bits; ++b) {
    const bool bit = (v >> b) & 1u;
    const uint64_t bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

__device__ __forceinline__ uint64_t lanes_below(int lane) { return (1ull << lane) - 1ull; }

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  return x;
}

// Which tile a workgroup of the count / scatter kernels takes.  Workgroup b is observed to run on XCD b mod 8
// (MI355X_MICROARCH.md; placement only affects speed, never results): every XCD gets a CONTIGUOUS range of tiles, in
// dispatch order.  The tiles in flight on one XCD are then neighbours, and what neighbouring tiles write lies side
// by side: a bucket's 6-key run of tile t is followed by the run of tile t + 1 (12 / 24 bytes each), the histogram
// column entries hist[b][t], hist[b][t + 1] share a sector -- so the XCD's L2 merges them into full lines before they
// leave for HBM, and the scatter's column reads of the histogram hit it.  (Round 2 handed tiles out round-robin: every
// partial sector crossed the fabric on its own -- 189 MB counted for 91 MB of keys + ids.)
// grid = 8 * ceil(tiles / 8); returns tiles (= "none") for the surplus workgroups.
__device__ __forceinline__ uint32_t plan_tile_of_block(uint32_t b, uint32_t tiles) {
  const uint32_t per = (tiles + 7u) / 8u;
  const uint32_t t = (b % 8u) * per + b / 8u;
  return t < tiles ? t : tiles;
}
static inline uint32_t plan_tile_grid(uint32_t tiles) { return 8u * ((tiles + 7u) / 8u); }

// ---- 1. per-tile bucket histogram ---------------------------------------------------------------------
template <bool HASHED>
__global__ __launch_bounds__(kPlanThreads) void plan_count_kernel(PlanArgs a) {
  extern __shared__ uint32_t s_hist[];  // [nb]
  const uint32_t nb = a.g.nb;
  if (blockIdx.x == 0 && threadIdx.x < PC_N) a.w.counters[threadIdx.x] = 0;
  // the row counts the back kernels add to (rc_bucket_plan hands them in from outside the workspace): zeroed here, by the plan's first
  // kernel, not by two memset launches in front of it
  if (blockIdx.x == 0 && threadIdx.x == PC_N) {
    if (a.n_rows_a) *a.n_rows_a = 0;
    if (a.n_rows_b) *a.n_rows_b = 0;
  }
  const uint32_t tile = plan_tile_of_block(blockIdx.x, a.g.tiles);
  if (tile == a.g.tiles) return;
  for (uint32_t i = threadIdx.x; i < nb; i += kPlanThreads) s_hist[i] = 0;
  __syncthreads();
  const uint32_t tile0 = tile * (uint32_t)kPlanTile;
  uint32_t key[kPlanRounds];
#pragma unroll
  for (int r = 0; r < kPlanRounds; ++r) {
    const uint32_t p = tile0 + r * kPlanThreads + threadIdx.x;
    key[r] = p < a.n ? plan_key<HASHED>(a, p) : 0xFFFFFFFFu;
  }
  if (a.single_a && tile0 < a.n_a)  // flags of this tile's list-a positions start at 0 (padded buffer: whole uint4s)
    reinterpret_cast<uint4*>(a.single_a + tile0)[threadIdx.x] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < kPlanRounds; ++r) {
    if (key[r] == 0xFFFFFFFFu) continue;
    const uint32_t p = tile0 + r * kPlanThreads + threadIdx.x;
    atomicAdd(&s_hist[plan_bucket_of<HASHED>(a, key[r], p >= a.n_a, nullptr)], 1u);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb; b += kPlanThreads) a.w.hist[(size_t)b * a.g.tiles + tile] = s_hist[b];
}

// ---- 2. per bucket: exclusive scan over the tiles -------------------------------------------------------
__global__ __launch_bounds__(kBlock) void plan_colscan_kernel(PlanArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t b = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (b >= a.g.nb) return;
  uint32_t* h = a.w.hist + (size_t)b * a.g.tiles;
  uint32_t carry = 0;
  constexpr int kBatch = 16;  // 1,024 tiles per trip: all loads of a trip in flight together
  for (uint32_t t0 = 0; t0 < a.g.tiles; t0 += 64 * kBatch) {
    uint32_t v[kBatch];
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const uint32_t t = t0 + q * 64 + lane;
      v[q] = t < a.g.tiles ? h[t] : 0u;
    }
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const uint32_t t = t0 + q * 64 + lane;
      const uint32_t incl = wave_inclusive_scan(v[q], lane);
      if (t < a.g.tiles) h[t] = carry + incl - v[q];
      carry += __shfl(incl, 63, 64);
    }
  }
  if (lane == 0) a.w.totals[b] = carry;
}

// ---- 3. stable partition into the buckets -----------------------------------------------------------------
// Ranks inside the tile come from wave-level ballot matching (phase A: position order within a wave, waves in
// order), tile offsets from step 2.  The keys are first placed into an LDS image of the tile sorted by bucket,
// then written out in image order: consecutive threads write consecutive addresses inside a bucket's run, so a
// wave store touches ~tile/nb-key runs instead of 64 scattered 8-byte slots (measured: 85 -> see DESIGN.md).
// LDS: u16 wave counters [8][nb], u32 run deltas [nb], the image (u32 packed lid|local position + u16 bucket; hashed
// geometry: u32 id + u16 local position + u16 bucket).
template <bool HASHED>
__global__ __launch_bounds__(kPlanThreads) void plan_scatter_kernel(PlanArgs a) {
  extern __shared__ uint32_t smem[];
  const uint32_t nb = a.g.nb;
  const uint32_t nbp = (nb + 1) & ~1u;
  uint32_t* gdelta = smem;                                   // [nb]   global index of image slot i = gdelta[bucket(i)] + i
  uint32_t* image = gdelta + nbp;                            // [kPlanTile]
  uint32_t* wsum = image + kPlanTile;                        // [2 * kPlanWaves]
  uint16_t* image_b = reinterpret_cast<uint16_t*>(wsum + 2 * kPlanWaves);  // [kPlanTile]
  uint16_t* image_p = image_b + kPlanTile;                   // [kPlanTile] (hashed only)
  uint16_t* cnt = image_p + (HASHED ? kPlanTile : 0);        // [kPlanWaves][nbp]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint32_t tile = plan_tile_of_block(blockIdx.x, a.g.tiles);
  if (tile == a.g.tiles) return;
  for (uint32_t i = threadIdx.x; i < kPlanWaves * nbp / 2; i += kPlanThreads) reinterpret_cast<uint32_t*>(cnt)[i] = 0;

  // issue this wave's id loads (1,024 consecutive positions) before anything else
  const uint32_t tile0 = tile * (uint32_t)kPlanTile;
  const uint32_t pos0 = tile0 + wave * (kPlanTile / kPlanWaves);
  uint32_t key[kPlanRounds];
#pragma unroll
  for (int r = 0; r < kPlanRounds; ++r) {
    const uint32_t p = pos0 + r * 64 + lane;
    key[r] = p < a.n ? plan_key<HASHED>(a, p) : 0xFFFFFFFFu;
  }

  // bucket bases: exclusive scan of the bucket totals (nb <= 4096 = 8 consecutive buckets per thread)
  constexpr int kPer = kPlanMaxBuckets / kPlanThreads;
  uint32_t tv[kPer];
  uint32_t tsum = 0;
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const uint32_t b = threadIdx.x * kPer + q;
    tv[q] = b < nb ? a.w.totals[b] : 0u;
    tsum += tv[q];
  }
  uint32_t incl = wave_inclusive_scan(tsum, lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();  // also: cnt[] zeroed
  uint32_t run = incl - tsum;
  for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const uint32_t b = threadIdx.x * kPer + q;
    if (b < nb) {
      gdelta[b] = run + a.w.hist[(size_t)b * a.g.tiles + tile];  // bucket base + this bucket's keys in earlier tiles
      if (tile == 0) a.w.bucket_base[b] = run;
    }
    run += tv[q];
  }
  if (tile == 0 && threadIdx.x == kPlanThreads - 1) a.w.bucket_base[nb] = run;

  // phase A: rank of every key among the keys of its bucket in THIS WAVE's 1,024 positions, in position order
  uint32_t rank[kPlanRounds];
  uint16_t* mycnt = cnt + (size_t)wave * nbp;
#pragma unroll
  for (int r = 0; r < kPlanRounds; ++r) {
    const bool valid = key[r] != 0xFFFFFFFFu;
    bool bad = false;
    const uint32_t b = valid ? plan_bucket_of<HASHED>(a, key[r], pos0 + r * 64 + lane >= a.n_a, &bad) : 0u;
    if (bad) a.w.counters[PC_STATUS] = 1u;  // id outside its table (nn.Embedding would raise a device assert): kept in bounds, status bit raised
    const uint64_t m = match_lanes(b, a.g.bucket_bits, __ballot(valid));
    const uint32_t old = mycnt[b];
    const uint32_t below = (uint32_t)__popcll(m & lanes_below(lane));
    if (valid && below == 0) mycnt[b] = (uint16_t)(old + (uint32_t)__popcll(m));
    rank[r] = old + below;
  }
  __syncthreads();
  // phase B: per bucket, wave counts -> image slots (bucket's run start in the image + keys of earlier waves)
  uint32_t tc[kPer];
  uint32_t csum = 0;
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const uint32_t b = threadIdx.x * kPer + q;
    uint32_t c = 0;
    if (b < nb) {
#pragma unroll
      for (int w = 0; w < kPlanWaves; ++w) c += cnt[(size_t)w * nbp + b];
    }
    tc[q] = c;
    csum += c;
  }
  incl = wave_inclusive_scan(csum, lane);
  if (lane == 63) wsum[kPlanWaves + wave] = incl;
  __syncthreads();
  uint32_t lrun = incl - csum;
  for (int w = 0; w < wave; ++w) lrun += wsum[kPlanWaves + w];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const uint32_t b = threadIdx.x * kPer + q;
    if (b < nb) {
      gdelta[b] -= lrun;  // (mod 2^32; the sum below is in range again)
      uint32_t at = lrun;
#pragma unroll
      for (int w = 0; w < kPlanWaves; ++w) {
        const uint32_t c = cnt[(size_t)w * nbp + b];
        cnt[(size_t)w * nbp + b] = (uint16_t)at;
        at += c;
      }
    }
    lrun += tc[q];
  }
  __syncthreads();
  // phase C: the tile's image, sorted by bucket, position order inside a bucket
  const uint32_t lid_mask = (1u << a.g.shift) - 1u;
#pragma unroll
  for (int r = 0; r < kPlanRounds; ++r) {
    if (key[r] == 0xFFFFFFFFu) continue;
    const uint32_t lp = (uint32_t)(wave * (kPlanTile / kPlanWaves) + r * 64 + lane);   // position within the tile
    const uint32_t b = plan_bucket_of<HASHED>(a, key[r], tile0 + lp >= a.n_a, nullptr);
    const uint32_t slot = (uint32_t)mycnt[b] + rank[r];
    if (HASHED) {
      image[slot] = key[r];
      image_p[slot] = (uint16_t)lp;
    } else {
      image[slot] = ((key[r] & lid_mask) << 13) | lp;
    }
    image_b[slot] = (uint16_t)b;
  }
  __syncthreads();
  // phase D: write the image out; slot i of bucket b goes to gdelta[b] + i
  uint32_t tile_n = 0;   // keys of this tile = all positions except those past n and those with a negative id
  for (int w = 0; w < kPlanWaves; ++w) tile_n += wsum[kPlanWaves + w];
  for (uint32_t i = threadIdx.x; i < tile_n; i += kPlanThreads) {
    const uint32_t e = image[i];
    const uint32_t at = gdelta[image_b[i]] + i;
    if (HASHED) {
      a.w.lid32[at] = e;
      a.w.pos[at] = tile0 + image_p[i];
    } else {
      a.w.lid[at] = (uint16_t)(e >> 13);
      a.w.pos[at] = tile0 + (e & (kPlanTile - 1));
    }
  }
}

// ---- 4. one workgroup per bucket: rows, grouped positions, singleton flags -----------------------------------
// LDS table indexed by (id - bucket start): occurrence counts (all four waves, order-free LDS atomics), then a
// block scan turns counts into slot cursors and emits the row records, then wave 0 alone walks the bucket's keys
// in position order and gives every position the next slot of its row: the returning LDS atomic hands out the
// slots; when two lanes of a round share a row (the read-back differs from old + 1) the lanes of that row are
// re-ranked in lane order, so every row's positions come out ascending whatever order the LDS served the atomics in.
//
// Two instantiations.  Buckets of fewer than 32,768 keys (all of them, unless an id range is very hot) keep 16-bit
// cells, two per LDS word: 17.5 KB of LDS per workgroup, so every bucket of the bench shape is resident at once
// (with 32-bit cells, 33.8 KB, the 1,344 workgroups ran in two rounds: 107 us; profiled phases, one round each:
// zeroing 5, count 8, scan + row records 23, ordered pass 46, its scattered stores 26).  Larger buckets -- and the
// narrowest bucket width, where a thread owns half a word -- are left to the 32-bit instantiation, launched after it.
constexpr int kBucketThreads = 256;
constexpr int kBucketBatch = 16;  // rounds of 64 keys per load batch of wave 0 (double-buffered)
constexpr uint32_t kNarrowLimit = 32768;

template <bool WIDE>
struct BucketCells {
  uint32_t* tab;
  int per_shift;  // ids per scan thread = 1 << per_shift
  static constexpr uint32_t kSingle = WIDE ? 0x80000000u : 0x8000u;
  __device__ __forceinline__ uint32_t widx(uint32_t lid) const {
    if (WIDE) return lid + (lid >> per_shift);
    const uint32_t w = lid >> 1;
    return w + (w >> (per_shift - 1));
  }
  __device__ __forceinline__ int sh(uint32_t lid) const { return WIDE ? 0 : 16 * (int)(lid & 1u); }
  __device__ __forceinline__ uint32_t get(uint32_t lid) const {
    const uint32_t v = tab[widx(lid)];
    return WIDE ? v : (v >> sh(lid)) & 0xFFFFu;
  }
  __device__ __forceinline__ void set(uint32_t lid, uint32_t v) const {  // by the thread that owns lid's whole word
    if (WIDE) tab[widx(lid)] = v;
    else tab[widx(lid)] = (tab[widx(lid)] & ~(0xFFFFu << sh(lid))) | (v << sh(lid));
  }
  __device__ __forceinline__ void add1(uint32_t lid) const { atomicAdd(&tab[widx(lid)], 1u << sh(lid)); }
  __device__ __forceinline__ uint32_t add1_rtn(uint32_t lid) const {
    const uint32_t old = atomicAdd(&tab[widx(lid)], 1u << sh(lid));
    return WIDE ? old : (old >> sh(lid)) & 0xFFFFu;
  }
  __device__ __forceinline__ uint32_t reload(uint32_t lid) const {  // after every lane's add of the round
    const uint32_t v = __hip_atomic_load(&tab[widx(lid)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return WIDE ? v : (v >> sh(lid)) & 0xFFFFu;
  }
};

constexpr uint32_t kNoLid = 0xFFFFFFFFu;

// a listed row with more than kPlanLongSeg occurrences: long-row record + one entry per 256-occurrence chunk (one thread;
// integer atomics only -- the order of the lists influences no floating-point result: chunks are combined in chunk order)
__device__ __forceinline__ void plan_register_long(const PlanArgs& a, const rc_plan_row& e, uint32_t side) {
  const uint32_t nch = (e.n + kPlanChunk - 1) / kPlanChunk;
  const uint32_t slot = atomicAdd(&a.w.counters[PC_LONG], 1u);
  const uint32_t cbase = atomicAdd(&a.w.counters[PC_CHUNKS], nch);
  if (slot < a.lw.long_cap) {
    PlanLongRow r;
    r.row = e.row; r.start = e.start; r.n = e.n; r.cbase = cbase; r.nchunks = nch; r.side = side;
    r.pad0 = r.pad1 = 0;
    a.lw.lrows[slot] = r;
  }
  for (uint32_t k = 0; k < nch; ++k)
    if (cbase + k < a.lw.chunk_cap) {
      PlanChunkInfo c;
      c.lrow = slot;
      c.k = k;
      a.lw.chunks[cbase + k] = c;
    }
}

// occurrences per id of one bucket into the LDS cells (all threads of the workgroup, order-free LDS atomics);
// reads the 2-byte id stream only
template <bool WIDE>
__device__ __forceinline__ void bucket_count_pass(const BucketCells<WIDE>& cells, const uint16_t* __restrict__ lid, uint32_t beg,
                                                  uint32_t end, int tid) {
  constexpr int kCountBatch = 8;
  for (uint32_t j0 = beg; j0 < end; j0 += kBucketThreads * kCountBatch) {
    uint32_t k[kCountBatch];
#pragma unroll
    for (int q = 0; q < kCountBatch; ++q) {
      const uint32_t j = j0 + q * kBucketThreads + tid;
      k[q] = j < end ? (uint32_t)lid[j] : kNoLid;
    }
#pragma unroll
    for (int q = 0; q < kCountBatch; ++q)
      if (k[q] != kNoLid) cells.add1(k[q]);
  }
}

template <bool WIDE>
__global__ __launch_bounds__(kBucketThreads) void plan_bucket_kernel(PlanArgs a) {
  extern __shared__ uint32_t tab[];  // table words + 256 pad words, then 16 words of scan scratch
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t bkt = blockIdx.x;
  const int shift = a.g.shift;
  const uint32_t ids = 1u << shift;
  const int per_shift = shift - 8;          // ids per thread in the scan = 1 << per_shift
  const uint32_t per = 1u << per_shift;
  const uint32_t beg = a.w.bucket_base[bkt], end = a.w.bucket_base[bkt + 1];
  if (beg == end) return;  // block-uniform
  if (WIDE != (end - beg >= kNarrowLimit || per_shift < 1)) return;  // the other instantiation's bucket
  // thread i owns ids [i*per, (i+1)*per); one pad word per thread (odd stride) keeps the walks off each other's banks
  const uint32_t words = (WIDE ? ids : ids / 2) + kBucketThreads;
  const BucketCells<WIDE> cells{tab, per_shift};
  constexpr uint32_t kSingle = BucketCells<WIDE>::kSingle;
  uint32_t* scratch = tab + words;
  const bool side_b = bkt >= a.g.nb_a;
  const bool list_all = side_b || a.list_single_a != 0;
#ifdef RC_X_TIMING
  uint64_t tq[5];
  tq[0] = wall_clock64();
#endif
  for (uint32_t i = tid; i < words; i += kBucketThreads) tab[i] = 0;
  __syncthreads();

  // pass 1: occurrences per id
  bucket_count_pass<WIDE>(cells, a.w.lid, beg, end, tid);
  __syncthreads();
#ifdef RC_X_TIMING
  tq[1] = wall_clock64();
#endif
  if (a.bitmap_in_bucket && a.bitmap_a && bkt < a.g.nb_a) {   // block-uniform; same words as plan_bitmap_kernel
    uint32_t* out = a.bitmap_a + ((size_t)bkt << (shift - 5));
    for (uint32_t wd = tid; wd < (ids >> 5); wd += kBucketThreads) {
      uint32_t bits = 0;
#pragma unroll 8
      for (uint32_t j = 0; j < 32; ++j) bits |= (cells.get(wd * 32 + j) >= 2u ? 1u : 0u) << j;
      out[wd] = bits;
    }
  }

  // scan: counts -> cursors (listed rows) / marker (rows that are only flagged); row records
  uint32_t my_occ = 0, my_rows = 0;
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t c = cells.get(tid * per + j);
    if (c != 0 && (list_all || c >= 2)) {
      my_occ += c;
      ++my_rows;
    }
  }
  const uint32_t occ_incl = wave_inclusive_scan(my_occ, lane);
  const uint32_t rows_incl = wave_inclusive_scan(my_rows, lane);
  if (lane == 63) {
    scratch[wave] = occ_incl;
    scratch[4 + wave] = rows_incl;
  }
  __syncthreads();
  uint32_t occ_off = occ_incl - my_occ, row_at = rows_incl - my_rows, total_rows = 0;
  for (int w = 0; w < kBucketThreads / 64; ++w) {
    if (w < wave) {
      occ_off += scratch[w];
      row_at += scratch[4 + w];
    }
    total_rows += scratch[4 + w];
  }
  if (tid == 0) scratch[8] = total_rows ? atomicAdd(side_b ? a.n_rows_b : a.n_rows_a, total_rows) : 0u;
  __syncthreads();
  row_at += scratch[8];
  rc_plan_row* rows = side_b ? a.rows_b : a.rows_a;
  const uint32_t row0 = (bkt - (side_b ? a.g.nb_a : 0u)) << shift;  // table-local id of the bucket's first row
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t lid = tid * per + j;
    const uint32_t c = cells.get(lid);
    if (c == 0) continue;
    if (list_all || c >= 2) {
      rc_plan_row e;
      e.row = row0 + lid;
      e.start = beg + occ_off;
      e.n = c;
      e.reserved = 0;
      rows[row_at++] = e;
      if (a.emit_long && c > (uint32_t)kPlanLongSeg) plan_register_long(a, e, side_b ? 1u : 0u);
      cells.set(lid, occ_off);
      occ_off += c;
    } else {
      cells.set(lid, kSingle);
    }
  }
  __syncthreads();
  if (wave != 0) return;
#ifdef RC_X_TIMING
  tq[2] = wall_clock64();
#endif

  // pass 2 (wave 0): positions into their row's slots, ascending
  uint8_t* single = (side_b || a.flags_done) ? nullptr : a.single_a;
  uint32_t cur_l[kBucketBatch], nxt_l[kBucketBatch], cur_p[kBucketBatch], nxt_p[kBucketBatch];
#pragma unroll
  for (int q = 0; q < kBucketBatch; ++q) {
    const uint32_t j = beg + q * 64 + lane;
    nxt_l[q] = j < end ? (uint32_t)a.w.lid[j] : kNoLid;
    nxt_p[q] = j < end ? a.w.pos[j] : 0u;
  }
  for (uint32_t j0 = beg; j0 < end; j0 += 64 * kBucketBatch) {
#pragma unroll
    for (int q = 0; q < kBucketBatch; ++q) {
      cur_l[q] = nxt_l[q];
      cur_p[q] = nxt_p[q];
    }
    const uint32_t jn = j0 + 64 * kBucketBatch;
#pragma unroll
    for (int q = 0; q < kBucketBatch; ++q) {  // next batch's keys travel while this one is placed
      const uint32_t j = jn + q * 64 + lane;
      nxt_l[q] = j < end ? (uint32_t)a.w.lid[j] : kNoLid;
      nxt_p[q] = j < end ? a.w.pos[j] : 0u;
    }
#pragma unroll
    for (int q = 0; q < kBucketBatch; ++q) {
      if (j0 + q * 64 >= end) break;  // wave-uniform
      const bool valid = cur_l[q] != kNoLid;
      const uint32_t lid = cur_l[q];
      const uint32_t p = cur_p[q];
      uint32_t old = 0, now = 1;
      if (valid) {
        old = cells.add1_rtn(lid);
        now = cells.reload(lid);
      }
      uint32_t slot = old;
      uint64_t pending = __ballot(valid && now != old + 1u);  // rows shared by several lanes of this round
      while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const uint32_t lid0 = __shfl(lid, leader, 64);
        const uint32_t now0 = __shfl(now, leader, 64);
        const uint64_t same = __ballot(valid && lid == lid0);
        if (valid && lid == lid0) slot = now0 - (uint32_t)__popcll(same) + (uint32_t)__popcll(same & lanes_below(lane));
        pending &= ~same;
      }
      if (!valid) continue;
      if (slot & kSingle) {
        if (single) single[p] = 1;
      } else {
        a.occ[beg + slot] = p;
      }
    }
  }
#ifdef RC_X_TIMING
  if ((bkt == 5 || bkt == 700) && lane == 0)
    printf("plan_bucket<%d> bkt %u: %u keys; ticks (100 MHz): zero+count %llu scan+rows %llu ordered pass %llu\n", (int)WIDE, bkt, end - beg,
           (unsigned long long)(tq[1] - tq[0]), (unsigned long long)(tq[2] - tq[1]), (unsigned long long)(wall_clock64() - tq[2]));
#endif
}

// ---- 4h. hashed geometry: one workgroup per bucket, the bucket's ids grouped through an LDS hash table ---------------
// A bucket holds the keys whose id hashes to it -- any ids of the table, not an id range -- so the count / cursor cell
// of a row is found by open addressing (linear probing, atomicCAS claims a slot) instead of by index.  Otherwise the
// passes are those of plan_bucket_kernel: pass 1 counts (all waves, order-free), a scan over the SLOTS turns counts
// into cursors and emits the row records, wave 0 alone walks the keys in position order and hands out the slots of
// every row ascending.  8,192 slots for ~1,024 keys per bucket (plan_geometry): load <= 1/8 when all keys differ.
// The table of a bucket has S = 2^lg slots, S >= 2 x its keys (256 .. 8,192): zeroing and scanning follow the bucket's
// size, not the maximum.
__device__ __forceinline__ uint32_t hash_slot0(uint32_t id, int lg) { return (id * 0x85EBCA6Bu) >> (32 - lg); }

// slot of `id`, claiming an empty one on first sight (pass 1) -- or S when the table is full
__device__ __forceinline__ uint32_t hash_insert(uint32_t* keys, uint32_t id, int lg) {
  const uint32_t S = 1u << lg;
  uint32_t h = hash_slot0(id, lg);
  for (uint32_t probe = 0; probe < S; ++probe) {
    const uint32_t k = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k == id) return h;
    if (k == 0xFFFFFFFFu) {
      const uint32_t old = atomicCAS(&keys[h], 0xFFFFFFFFu, id);
      if (old == 0xFFFFFFFFu || old == id) return h;
    }
    h = (h + 1) & (S - 1);
  }
  return S;
}
// slot of an id that pass 1 inserted
__device__ __forceinline__ uint32_t hash_find(const uint32_t* keys, uint32_t id, int lg) {
  const uint32_t S = 1u << lg;
  uint32_t h = hash_slot0(id, lg);
  for (uint32_t probe = 0; probe < S; ++probe) {
    if (keys[h] == id) return h;
    h = (h + 1) & (S - 1);
  }
  return 0;
}

// Hot rows: when the first valid lane's id is shared by at least kHotGroup lanes of the round (a row with thousands of
// occurrences fills whole rounds), ONE lane moves the cell for the whole group; 64 same-address LDS atomics would be served
// one after the other.  -> mask of the lanes that were taken care of (0: nobody)
constexpr int kHotGroup = 8;

constexpr int kHashBatch = 8;   // rounds of 64 keys requested together by the ordered pass

__global__ __launch_bounds__(kBucketThreads) void plan_bucket_hash_kernel(PlanArgs a) {
  extern __shared__ uint32_t tab[];  // keys [S], cells [S], 16 words of scan scratch (carved for the largest table)
  constexpr uint32_t kSingle = 0x80000000u;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t bkt = blockIdx.x;
  const uint32_t beg = a.w.bucket_base[bkt], end = a.w.bucket_base[bkt + 1];
  if (beg == end) return;  // block-uniform
  int lg = 8;
  while ((1u << lg) < 2u * (end - beg) && lg < 13) ++lg;
  const uint32_t S = 1u << lg;
  uint32_t* keys = tab;
  uint32_t* cells = tab + S;
  uint32_t* scratch = tab + 2 * kPlanHashSlots;
  const bool side_b = bkt >= a.g.nb_a;
  const bool list_all = side_b || a.list_single_a != 0;
  for (uint32_t i = tid; i < S; i += kBucketThreads) {
    keys[i] = 0xFFFFFFFFu;
    cells[i] = 0;
  }
  __syncthreads();

  // pass 1: occurrences per id
  constexpr int kCountBatch = 8;
  for (uint32_t j0 = beg; j0 < end; j0 += kBucketThreads * kCountBatch) {
    uint32_t k[kCountBatch];
#pragma unroll
    for (int q = 0; q < kCountBatch; ++q) {
      const uint32_t j = j0 + q * kBucketThreads + tid;
      k[q] = j < end ? a.w.lid32[j] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int q = 0; q < kCountBatch; ++q) {
      const bool valid = k[q] != 0xFFFFFFFFu;
      const uint64_t vm = __ballot(valid);
      if (vm == 0) continue;  // wave-uniform
      const uint32_t k0 = (uint32_t)__shfl((int)k[q], __ffsll((long long)vm) - 1, 64);
      const uint64_t grp = __ballot(valid && k[q] == k0);
      const bool hot = __popcll(grp) >= kHotGroup;
      const bool mine = valid && (!hot || k[q] != k0 || lane == __ffsll((long long)grp) - 1);
      if (mine) {
        const uint32_t h = hash_insert(keys, k[q], lg);
        if (h < S) atomicAdd(&cells[h], (hot && k[q] == k0) ? (uint32_t)__popcll(grp) : 1u);
        else a.w.counters[PC_STATUS] = 2u;   // more distinct rows than slots: the plan is incomplete (never at the sizes plan_geometry admits)
      }
    }
  }
  __syncthreads();

  // scan over the slots: counts -> cursors (listed rows) / marker (rows that are only flagged); row records
  const uint32_t per = S / kBucketThreads;
  uint32_t my_occ = 0, my_rows = 0;
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t c = cells[tid * per + j];
    if (c != 0 && (list_all || c >= 2)) {
      my_occ += c;
      ++my_rows;
    }
  }
  const uint32_t occ_incl = wave_inclusive_scan(my_occ, lane);
  const uint32_t rows_incl = wave_inclusive_scan(my_rows, lane);
  if (lane == 63) {
    scratch[wave] = occ_incl;
    scratch[4 + wave] = rows_incl;
  }
  __syncthreads();
  uint32_t occ_off = occ_incl - my_occ, row_at = rows_incl - my_rows, total_rows = 0;
  for (int w = 0; w < kBucketThreads / 64; ++w) {
    if (w < wave) {
      occ_off += scratch[w];
      row_at += scratch[4 + w];
    }
    total_rows += scratch[4 + w];
  }
  if (tid == 0) scratch[8] = total_rows ? atomicAdd(side_b ? a.n_rows_b : a.n_rows_a, total_rows) : 0u;
  __syncthreads();
  row_at += scratch[8];
  rc_plan_row* rows = side_b ? a.rows_b : a.rows_a;
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t sl = tid * per + j;
    const uint32_t c = cells[sl];
    if (c == 0) continue;
    if (list_all || c >= 2) {
      rc_plan_row e;
      e.row = keys[sl];
      e.start = beg + occ_off;
      e.n = c;
      e.reserved = 0;
      rows[row_at++] = e;
      if (a.emit_long && c > (uint32_t)kPlanLongSeg) plan_register_long(a, e, side_b ? 1u : 0u);
      cells[sl] = occ_off;
      occ_off += c;
    } else {
      cells[sl] = kSingle;
    }
  }
  __syncthreads();
  if (wave != 0) return;

  // pass 2 (wave 0): positions into their row's slots, ascending; kHashBatch rounds of keys requested together
  uint8_t* single = (side_b || a.flags_done) ? nullptr : a.single_a;
  for (uint32_t jb = beg; jb < end; jb += 64 * kHashBatch) {
    uint32_t kk[kHashBatch], pp[kHashBatch];
#pragma unroll
    for (int q = 0; q < kHashBatch; ++q) {
      const uint32_t j = jb + q * 64 + lane;
      kk[q] = j < end ? a.w.lid32[j] : 0xFFFFFFFFu;
      pp[q] = j < end ? a.w.pos[j] : 0u;
    }
#pragma unroll
    for (int q = 0; q < kHashBatch; ++q) {
      if (jb + q * 64 >= end) break;  // wave-uniform
      const uint32_t lid = kk[q], p = pp[q];
      const bool valid = lid != 0xFFFFFFFFu;
      const uint64_t vm = __ballot(valid);
      if (vm == 0) continue;
      // hot group of the round's first id: one atomic for all of its lanes
      const int first = __ffsll((long long)vm) - 1;
      const uint32_t lid0 = (uint32_t)__shfl((int)lid, first, 64);
      const uint64_t grp = __ballot(valid && lid == lid0);
      const bool hot = __popcll(grp) >= kHotGroup;
      uint32_t slot = 0;
      if (hot) {
        uint32_t base = 0;
        if (lane == first) base = atomicAdd(&cells[hash_find(keys, lid0, lg)], (uint32_t)__popcll(grp));
        base = (uint32_t)__shfl((int)base, first, 64);
        if (valid && lid == lid0) slot = base + (uint32_t)__popcll(grp & lanes_below(lane));
      }
      const bool rest = valid && !(hot && lid == lid0);
      uint32_t old = 0, now = 1;
      if (rest) {
        const uint32_t h = hash_find(keys, lid, lg);
        old = atomicAdd(&cells[h], 1u);
        now = __hip_atomic_load(&cells[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        slot = old;
      }
      uint64_t pending = __ballot(rest && now != old + 1u);  // rows shared by several lanes of this round
      while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const uint32_t l0 = (uint32_t)__shfl((int)lid, leader, 64);
        const uint32_t now0 = (uint32_t)__shfl((int)now, leader, 64);
        const uint64_t same = __ballot(rest && lid == l0);
        if (rest && lid == l0) slot = now0 - (uint32_t)__popcll(same) + (uint32_t)__popcll(same & lanes_below(lane));
        pending &= ~same;
      }
      if (!valid) continue;
      if (slot & kSingle) {
        if (single) single[p] = 1;
      } else {
        a.occ[beg + slot] = p;
      }
    }
  }
}

// ---- 4n. narrow geometry: one WAVE per bucket of at most 32 ids, ballot ranks instead of LDS atomics -----------------
// The keys of a bucket are in position order; lanes of a round that hold the same id find each other with `shift` ballots
// (match_lanes), the lowest of them (the leader) moves the id's counter / cursor in a 32-entry per-wave LDS table by the
// size of the group -- distinct ids are distinct cells, so no atomics and nothing serialises on a hot row.
constexpr int kNarrowBatch = 8;   // rounds of 64 keys requested together

__global__ __launch_bounds__(kBucketThreads) void plan_bucket_narrow_kernel(PlanArgs a) {
  __shared__ uint32_t tabs[kBucketThreads / 64][32];
  constexpr uint32_t kSingle = 0x80000000u;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint32_t* tab = tabs[wave];
  const uint32_t bkt = blockIdx.x * (kBucketThreads / 64) + wave;
  if (bkt >= a.g.nb) return;   // wave-uniform; no workgroup barriers below
  const uint32_t beg = a.w.bucket_base[bkt], end = a.w.bucket_base[bkt + 1];
  if (beg == end) return;
  const int shift = a.g.shift;
  const uint32_t ids = 1u << shift;
  const bool side_b = bkt >= a.g.nb_a;
  const bool list_all = side_b || a.list_single_a != 0;
  if (lane < 32) tab[lane] = 0;

  // pass A: occurrences per id
  for (uint32_t j0 = beg; j0 < end; j0 += 64 * kNarrowBatch) {
    uint32_t k[kNarrowBatch];
#pragma unroll
    for (int q = 0; q < kNarrowBatch; ++q) {
      const uint32_t j = j0 + q * 64 + lane;
      k[q] = j < end ? (uint32_t)a.w.lid[j] : kNoLid;
    }
#pragma unroll
    for (int q = 0; q < kNarrowBatch; ++q) {
      if (j0 + q * 64 >= end) break;  // wave-uniform
      const bool valid = k[q] != kNoLid;
      const uint64_t m = match_lanes(k[q], shift, __ballot(valid));
      if (valid && (m & lanes_below(lane)) == 0) tab[k[q]] += (uint32_t)__popcll(m);   // group leader; distinct ids = distinct cells
    }
  }
  // counts -> row records + cursors (lane j < ids owns id j)
  const uint32_t c = (uint32_t)lane < ids ? tab[lane & 31] : 0u;
  const bool listed = c != 0 && (list_all || c >= 2);
  const uint32_t occ_incl = wave_inclusive_scan(listed ? c : 0u, lane);
  const uint64_t lm = __ballot(listed);
  const uint32_t n_listed = (uint32_t)__popcll(lm);
  uint32_t row_base = 0;
  if (lane == 0 && n_listed) row_base = atomicAdd(side_b ? a.n_rows_b : a.n_rows_a, n_listed);
  row_base = __shfl(row_base, 0, 64);
  const uint32_t row0 = (bkt - (side_b ? a.g.nb_a : 0u)) << shift;  // table-local id of the bucket's first row
  if ((uint32_t)lane < ids) {
    if (listed) {
      rc_plan_row e;
      e.row = row0 + lane;
      e.start = beg + occ_incl - c;
      e.n = c;
      e.reserved = 0;
      (side_b ? a.rows_b : a.rows_a)[row_base + (uint32_t)__popcll(lm & lanes_below(lane))] = e;
      if (a.emit_long && c > (uint32_t)kPlanLongSeg) plan_register_long(a, e, side_b ? 1u : 0u);
      tab[lane] = occ_incl - c;
    } else {
      tab[lane] = kSingle;
    }
  }

  // pass B: positions into their row's slots, ascending
  uint8_t* single = (side_b || a.flags_done) ? nullptr : a.single_a;
  for (uint32_t j0 = beg; j0 < end; j0 += 64 * kNarrowBatch) {
    uint32_t k[kNarrowBatch], p[kNarrowBatch];
#pragma unroll
    for (int q = 0; q < kNarrowBatch; ++q) {
      const uint32_t j = j0 + q * 64 + lane;
      k[q] = j < end ? (uint32_t)a.w.lid[j] : kNoLid;
      p[q] = j < end ? a.w.pos[j] : 0u;
    }
#pragma unroll
    for (int q = 0; q < kNarrowBatch; ++q) {
      if (j0 + q * 64 >= end) break;  // wave-uniform
      const bool valid = k[q] != kNoLid;
      const uint64_t m = match_lanes(k[q], shift, __ballot(valid));
      if (!valid) continue;
      const uint32_t cur = tab[k[q]];
      const uint32_t rank = (uint32_t)__popcll(m & lanes_below(lane));
      if (cur & kSingle) {
        if (single) single[p[q]] = 1;
      } else {
        a.occ[beg + cur + rank] = p[q];
        if (rank == 0) tab[k[q]] = cur + (uint32_t)__popcll(m);
      }
    }
  }
}

// ---- 4a. the bitmap of multi-occurrence rows of list a: what the fused BPRMF kernel needs from the plan.
// One workgroup per bucket: LDS count table from the 2-byte id stream (all four waves, order-free), then thread t
// packs the cells of ids 32 t .. 32 t + 31 into one word -- bit = 1 iff the row occurs at least twice -- and the
// bucket's words leave as ONE coalesced 1 KB store (8,192-id buckets).  Round 2 kept a flag byte per batch position:
// 3.4 M scattered byte stores per step, one 32-byte sector each (0.22 GB counted for 57 MB algorithmic), plus a
// second pass over 8-byte keys.  The bitmap of a 10 M-row table is 1.25 MB: it stays in every XCD's L2, and the
// fused kernel looks its candidates up there (fused_body.hpp).  Words of empty buckets are not written and never
// read (a lookup is always for an id of the batch).
// The row records and grouped positions (plan_bucket_kernel, whose ordered pass is one wave per bucket) are needed
// only by the updates AFTER the fused kernel, so the step runs them on a second stream (train_step.hip).
template <bool WIDE>
__global__ __launch_bounds__(kBucketThreads) void plan_bitmap_kernel(PlanArgs a) {
  extern __shared__ uint32_t tab[];
  const int tid = threadIdx.x;
  const uint32_t bkt = blockIdx.x;  // grid = nb_a: buckets of list a only
  const int shift = a.g.shift;
  const uint32_t ids = 1u << shift;
  const int per_shift = shift - 8;
  const uint32_t beg = a.w.bucket_base[bkt], end = a.w.bucket_base[bkt + 1];
  if (beg == end) return;
  if (WIDE != (end - beg >= kNarrowLimit || per_shift < 1)) return;
  const uint32_t words = (WIDE ? ids : ids / 2) + kBucketThreads;
  const BucketCells<WIDE> cells{tab, per_shift};
  for (uint32_t i = tid; i < words; i += kBucketThreads) tab[i] = 0;
  __syncthreads();
  bucket_count_pass<WIDE>(cells, a.w.lid, beg, end, tid);
  __syncthreads();
  uint32_t* out = a.bitmap_a + ((size_t)bkt << (shift - 5));
  for (uint32_t w = tid; w < (ids >> 5); w += kBucketThreads) {
    uint32_t bits = 0;
#pragma unroll 8
    for (uint32_t j = 0; j < 32; ++j) bits |= (cells.get(w * 32 + j) >= 2u ? 1u : 0u) << j;
    out[w] = bits;
  }
}

// front: histogram, tile offsets, stable partition (+ the multi-occurrence bitmap of list a when `bitmap` is set)
int plan_launch_front(const PlanArgs& a, bool bitmap, hipStream_t s) {
  const PlanGeom& g = a.g;
  const size_t hist_lds = (size_t)g.nb * sizeof(uint32_t);
  if (g.hashed) hipLaunchKernelGGL(plan_count_kernel<true>, dim3(plan_tile_grid(g.tiles)), dim3(kPlanThreads), hist_lds, s, a);
  else hipLaunchKernelGGL(plan_count_kernel<false>, dim3(plan_tile_grid(g.tiles)), dim3(kPlanThreads), hist_lds, s, a);
  RC_LAUNCH_CHECK();
  hipLaunchKernelGGL(plan_colscan_kernel, dim3((g.nb + 3) / 4), dim3(kBlock), 0, s, a);
  RC_LAUNCH_CHECK();
  const size_t nbp = (g.nb + 1) & ~(size_t)1;
  const size_t sc_lds = (nbp + kPlanTile + 2 * kPlanWaves) * sizeof(uint32_t) +
                        ((size_t)kPlanTile * (g.hashed ? 2 : 1) + kPlanWaves * nbp) * sizeof(uint16_t);
  if (g.hashed) hipLaunchKernelGGL(plan_scatter_kernel<true>, dim3(plan_tile_grid(g.tiles)), dim3(kPlanThreads), sc_lds, s, a);
  else hipLaunchKernelGGL(plan_scatter_kernel<false>, dim3(plan_tile_grid(g.tiles)), dim3(kPlanThreads), sc_lds, s, a);
  RC_LAUNCH_CHECK();
  if (bitmap && a.bitmap_a && g.nb_a > 0) {
    if (g.hashed || g.narrow)
      return fail(RC_ERR_UNSUPPORTED, "bucket plan: the multi-occurrence bitmap needs the wide id-range geometry");
    const size_t ids = (size_t)1 << g.shift;
    if (g.shift > 8) {
      hipLaunchKernelGGL(plan_bitmap_kernel<false>, dim3(g.nb_a), dim3(kBucketThreads), (ids / 2 + kBucketThreads + 16) * sizeof(uint32_t), s, a);
      RC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(plan_bitmap_kernel<true>, dim3(g.nb_a), dim3(kBucketThreads), (ids + kBucketThreads + 16) * sizeof(uint32_t), s, a);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}

// back: per bucket, row records + grouped positions (+ the flags unless a.flags_done)
int plan_launch_back(const PlanArgs& a, hipStream_t s) {
  const PlanGeom& g = a.g;
  if (g.hashed) {
    hipLaunchKernelGGL(plan_bucket_hash_kernel, dim3(g.nb), dim3(kBucketThreads), (2 * kPlanHashSlots + 16) * sizeof(uint32_t), s, a);
    RC_LAUNCH_CHECK();
    return RC_OK;
  }
  if (g.narrow) {
    hipLaunchKernelGGL(plan_bucket_narrow_kernel, dim3((g.nb + 3) / 4), dim3(kBucketThreads), 0, s, a);
    RC_LAUNCH_CHECK();
    return RC_OK;
  }
  const size_t ids = (size_t)1 << g.shift;
  if (g.shift > 8) {  // 16-bit cells: every bucket below 32,768 keys
    hipLaunchKernelGGL(plan_bucket_kernel<false>, dim3(g.nb), dim3(kBucketThreads), (ids / 2 + kBucketThreads + 16) * sizeof(uint32_t), s, a);
    RC_LAUNCH_CHECK();
  }
  // 32-bit cells: the remaining buckets (every workgroup of a bucket the other launch took returns at once)
  hipLaunchKernelGGL(plan_bucket_kernel<true>, dim3(g.nb), dim3(kBucketThreads), (ids + kBucketThreads + 16) * sizeof(uint32_t), s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

int plan_launch(const PlanArgs& a, hipStream_t s, hipEvent_t* ev_after_scatter) {
  RC_TRY(plan_launch_front(a, false, s));
  if (ev_after_scatter) RC_HIP(hipEventRecord(*ev_after_scatter, s));
  return plan_launch_back(a, s);
}

int plan_prepare() {
  // function attributes are per device: once per device of the process
  static bool done[64] = {};
  int dev = 0;
  RC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(RC_ERR_UNSUPPORTED, "bucket plan: device index %d", dev);
  if (done[dev]) return RC_OK;
  // the scatter kernel needs up to ~150 KB of dynamic LDS at 4,096 buckets
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(plan_scatter_kernel<false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(plan_scatter_kernel<true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(plan_bucket_hash_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  done[dev] = true;
  return RC_OK;
}

}  // namespace rc

using namespace rc;

extern "C" int rc_bucket_plan_supported(int64_t n_a, int64_t n_b, int64_t range_a, int64_t range_b) {
  return plan_geometry(n_a, n_b, range_a, range_b).ok;
}

extern "C" size_t rc_bucket_plan_workspace_bytes(int64_t n_a, int64_t n_b) {
  if (n_a < 0 || n_b < 0) return 0;
  return carve_plan_ws(nullptr, n_a + n_b).total;
}

extern "C" const uint32_t* rc_bucket_plan_status_ptr(const void* ws, int64_t n_a, int64_t n_b) {
  if (!ws || n_a < 0 || n_b < 0) return nullptr;
  return carve_plan_ws(const_cast<void*>(ws), n_a + n_b).counters + PC_STATUS;
}

extern "C" size_t rc_bucket_plan_flags_bytes(int64_t n_a) {
  if (n_a < 0) return 0;
  return align_up((size_t)n_a, kPlanTile);
}

extern "C" int rc_bucket_plan(const int64_t* ids_a, int64_t n_a, int64_t range_a, const int64_t* ids_b, int64_t n_b,
                              int64_t range_b, int list_single_a, uint8_t* single_a, rc_plan_row* rows_a,
                              uint32_t* n_rows_a, rc_plan_row* rows_b, uint32_t* n_rows_b, uint32_t* occ, void* ws,
                              size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(n_a >= 0 && n_b >= 0 && n_a + n_b < ((int64_t)1 << 31), "rc_bucket_plan: sizes out of range");
  hipStream_t s = as_stream(stream);
  if (n_a + n_b == 0) {     // (a plan of something zeroes the counts in its first kernel)
    if (n_rows_a) RC_HIP(hipMemsetAsync(n_rows_a, 0, sizeof(uint32_t), s));
    if (n_rows_b) RC_HIP(hipMemsetAsync(n_rows_b, 0, sizeof(uint32_t), s));
    return RC_OK;
  }
  RC_REQUIRE((n_a == 0 || (ids_a && rows_a && n_rows_a)) && (n_b == 0 || (ids_b && rows_b && n_rows_b)) && occ && ws,
             "rc_bucket_plan: null pointer");
  RC_REQUIRE(list_single_a || single_a || n_a == 0, "rc_bucket_plan: single_a is needed when single-occurrence rows are not listed");
  PlanArgs a;
  memset(&a, 0, sizeof(a));
  a.g = plan_geometry(n_a, n_b, range_a, range_b);
  if (!a.g.ok)
    return fail(RC_ERR_UNSUPPORTED, "rc_bucket_plan: id ranges %lld + %lld need more than %d buckets of %d ids",
                (long long)range_a, (long long)range_b, kPlanMaxBuckets, 1 << kPlanMaxShift);
  a.w = carve_plan_ws(ws, n_a + n_b);
  if (ws_bytes < a.w.total) return fail(RC_ERR_WORKSPACE, "rc_bucket_plan: workspace %zu < %zu", ws_bytes, a.w.total);
  RC_TRY(plan_prepare());
  a.ids_a = ids_a; a.ids_b = ids_b; a.n_a = (uint32_t)n_a; a.n = (uint32_t)(n_a + n_b);
  a.range_a = range_a; a.range_b = range_b;
  a.list_single_a = list_single_a; a.single_a = single_a;
  a.rows_a = rows_a; a.rows_b = rows_b; a.n_rows_a = n_rows_a; a.n_rows_b = n_rows_b; a.occ = occ;
  return plan_launch(a, s, nullptr);
}

extern "C" size_t rc_bucket_bitmap_bytes(int64_t range_a) {
  if (range_a < 1) return 0;
  return (size_t)((range_a + ((int64_t)1 << kPlanMaxShift) - 1) >> kPlanMaxShift) << (kPlanMaxShift - 3);
}

extern "C" int rc_bucket_multi_bitmap(const int64_t* ids_a, int64_t n_a, int64_t range_a, uint32_t* bitmap, void* ws,
                                      size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(n_a >= 0 && n_a < ((int64_t)1 << 31), "rc_bucket_multi_bitmap: sizes out of range");
  if (n_a == 0) return RC_OK;
  RC_REQUIRE(ids_a && bitmap && ws, "rc_bucket_multi_bitmap: null pointer");
  PlanArgs a;
  memset(&a, 0, sizeof(a));
  a.g = plan_geometry(n_a, 0, range_a, 0, 2);
  if (!a.g.ok || a.g.hashed || a.g.narrow)
    return fail(RC_ERR_UNSUPPORTED, "rc_bucket_multi_bitmap: id range %lld needs more than %d buckets of %d ids",
                (long long)range_a, kPlanMaxBuckets, 1 << kPlanMaxShift);
  a.w = carve_plan_ws(ws, n_a);
  if (ws_bytes < a.w.total) return fail(RC_ERR_WORKSPACE, "rc_bucket_multi_bitmap: workspace %zu < %zu", ws_bytes, a.w.total);
  RC_TRY(plan_prepare());
  a.ids_a = ids_a; a.n_a = (uint32_t)n_a; a.n = (uint32_t)n_a; a.range_a = range_a;
  a.bitmap_a = bitmap;
  return plan_launch_front(a, true, as_stream(stream));
}
