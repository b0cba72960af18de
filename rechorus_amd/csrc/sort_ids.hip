// sort_ids.hip -- stable sort of a batch's row ids, the first half of the atomic-free
// replacement for aten::embedding_dense_backward's index_add (which the reference reaches
// through loss.backward(), helpers/BaseRunner.py:205) where no bucket plan exists for the id
// list (very hot rows under a large batch, row widths the plan kernels do not instantiate).
//
// ids (int64, reference layout) are narrowed to uint32 keys on the fly (the first pass reads the
// int64 ids directly, no key-materialising pre-pass) and paired with their position; only
// ceil(log2 key_range) key bits are sorted.  Two id lists can be sorted in ONE call as a virtual
// concatenation (second list offset by `key_offset_b`): a BPRMF step in the sort pipeline sorts its
// B*(1+K) item ids and B user ids together, the user segment is the tail.
//
// Hand-written LSD radix sort, 8 bits per pass, three launches per pass (rounds 1-5 linked rocPRIM's
// device radix sort here):
//   digit_count    a workgroup counts the digits of its tile of 2,048 keys (LDS histogram, integer adds)
//   digit_offsets  per digit an exclusive scan of the tiles' counts; the digits' totals
//   digit_scatter  every key to base[digit] + tile offset + its rank among the tile's earlier keys of
//                  the same digit.  Ranks come from ballots (the lanes of a 64-key strip that share a
//                  digit find each other bit by bit) and per-wave running counts, strips in position
//                  order: the placement is a function of the input alone -- stable, no atomics in it.
// The passes ping-pong between the output arrays and a scratch pair so that the last one lands in the
// outputs.  Everything that consumes the sorted order is seg_update.hip.
#include "common.hpp"

namespace rc {

constexpr int kSortTile = 2048;                    // keys per workgroup: 256 threads x 8
constexpr int kSortPerThread = kSortTile / kBlock;
constexpr int kSortDigits = 256;
static_assert(kSortDigits == kBlock && kSortTile == kBlock * kSortPerThread && kSortPerThread == 8, "one thread per digit, eight strips of 64 per wave");

struct SortSrc {        // pass 0: keys = narrowed ids of the two lists, payload = position; later passes: the ping-pong arrays
  const int64_t* a;
  const int64_t* b;
  uint32_t na, off_b;
  const uint32_t* keys;
  const uint32_t* perm;
};

__device__ __forceinline__ uint32_t sort_key(const SortSrc& s, uint32_t i) {
  if (s.keys) return s.keys[i];
  return i < s.na ? (uint32_t)s.a[i] : s.off_b + (uint32_t)s.b[i - s.na];
}

// counts[digit][tile]
__global__ __launch_bounds__(kBlock) void digit_count_kernel(SortSrc s, uint32_t n, int shift, uint32_t n_tiles, uint32_t* __restrict__ counts) {
  __shared__ uint32_t hist[kSortDigits];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
#pragma unroll
  for (int k = 0; k < kSortPerThread; ++k) {
    const uint32_t i = base + k * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&hist[(sort_key(s, i) >> shift) & 0xFFu], 1u);      // integer LDS add: order-free
  }
  __syncthreads();
  counts[(size_t)threadIdx.x * n_tiles + blockIdx.x] = hist[threadIdx.x];
}

// one workgroup per digit: counts[digit][0 .. n_tiles) -> exclusive prefix in place, total[digit]
__global__ __launch_bounds__(kBlock) void digit_offsets_kernel(uint32_t* __restrict__ counts, uint32_t n_tiles, uint32_t* __restrict__ total) {
  __shared__ uint32_t wsum[kBlock / 64];
  uint32_t* row = counts + (size_t)blockIdx.x * n_tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 < n_tiles; t0 += kBlock) {
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t v = t < n_tiles ? row[t] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t before = carry, sum = 0;
    for (int q = 0; q < kBlock / 64; ++q) {
      if (q < wave) before += wsum[q];
      sum += wsum[q];
    }
    if (t < n_tiles) row[t] = before + x - v;
    carry += sum;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[blockIdx.x] = carry;
}

__global__ __launch_bounds__(kBlock) void digit_scatter_kernel(SortSrc s, uint32_t n, int shift, uint32_t n_tiles, const uint32_t* __restrict__ counts,
                                                               const uint32_t* __restrict__ total, uint32_t* __restrict__ keys_out,
                                                               uint32_t* __restrict__ perm_out) {
  __shared__ uint32_t base[kSortDigits];                     // where this tile's keys of a digit start in the output
  __shared__ uint32_t wcnt[kBlock / 64][kSortDigits];        // a wave's running count per digit
  __shared__ uint32_t wsum[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // exclusive scan of the 256 digit totals (one per thread)
  {
    const uint32_t v = total[threadIdx.x];
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t before = 0;
    for (int q = 0; q < wave; ++q) before += wsum[q];
    base[threadIdx.x] = before + x - v + counts[(size_t)threadIdx.x * n_tiles + blockIdx.x];
  }
#pragma unroll
  for (int q = 0; q < kBlock / 64; ++q) wcnt[q][threadIdx.x] = 0;
  __syncthreads();
  // a wave owns 512 consecutive positions of the tile as eight strips of 64: keys and payloads into registers
  const uint32_t wbase = blockIdx.x * kSortTile + wave * (kSortTile / (kBlock / 64));
  uint32_t key[kSortPerThread], val[kSortPerThread];
#pragma unroll
  for (int k = 0; k < kSortPerThread; ++k) {
    const uint32_t i = wbase + k * 64 + lane;
    key[k] = i < n ? sort_key(s, i) : 0xFFFFFFFFu;
    val[k] = i < n ? (s.perm ? s.perm[i] : i) : 0u;
  }
  // phase A: this wave's digit counts (lanes of a strip that share a digit: one of them adds their number)
  uint64_t same[kSortPerThread];
#pragma unroll
  for (int k = 0; k < kSortPerThread; ++k) {
    const uint32_t i = wbase + k * 64 + lane;
    const bool in = i < n;
    const uint32_t dg = (key[k] >> shift) & 0xFFu;
    uint64_t m = __ballot(in);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bal = __ballot((dg >> b) & 1u);
      m &= ((dg >> b) & 1u) ? bal : ~bal;
    }
    same[k] = in ? m : 0ull;
    if (in && (m & ((1ull << lane) - 1ull)) == 0ull) wcnt[wave][dg] += (uint32_t)__popcll(m);   // the strip's first lane of this digit; one wave, in order
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // the waves' counts -> where each wave's keys of a digit start inside the tile's run (thread = digit)
  {
    uint32_t run = 0;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) {
      const uint32_t c = wcnt[q][threadIdx.x];
      wcnt[q][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
  // phase B: place.  Strip by strip: running offset of (wave, digit) + rank among the strip's lanes of the same digit
#pragma unroll
  for (int k = 0; k < kSortPerThread; ++k) {
    const uint32_t i = wbase + k * 64 + lane;
    const bool in = i < n;
    const uint32_t dg = (key[k] >> shift) & 0xFFu;
    const uint64_t m = same[k];
    if (in) {
      const uint32_t at = base[dg] + wcnt[wave][dg] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      keys_out[at] = key[k];
      perm_out[at] = val[k];
    }
    __builtin_amdgcn_wave_barrier();       // every lane has read the running offsets of this strip
    if (in && (m & ((1ull << lane) - 1ull)) == 0ull) wcnt[wave][dg] += (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
  }
}

static int key_bits(int64_t key_range) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < key_range) ++bits;
  return bits;
}

static uint32_t sort_tiles(int64_t n) { return (uint32_t)((n + kSortTile - 1) / kSortTile); }

}  // namespace rc

using namespace rc;

extern "C" size_t rc_sort_workspace_bytes(int64_t n) {
  if (n <= 0) n = 1;
  // scratch keys + payloads, the digit counts of every tile, the digits' totals
  return 2 * align_up((size_t)n * sizeof(uint32_t), 256) + align_up((size_t)kSortDigits * sort_tiles(n) * sizeof(uint32_t), 256) +
         align_up(kSortDigits * sizeof(uint32_t), 256) + 256;
}

extern "C" int rc_sort_ids2(const int64_t* ids_a, int64_t n_a, const int64_t* ids_b, int64_t n_b,
                            int64_t key_offset_b, int64_t key_range, uint32_t* keys_out,
                            uint32_t* perm_out, void* ws, size_t ws_bytes, rc_stream_t stream) {
  const int64_t n = n_a + n_b;
  RC_REQUIRE(n_a >= 0 && n_b >= 0 && n < ((int64_t)1 << 31), "rc_sort_ids2: sizes out of range");
  if (n == 0) return RC_OK;
  RC_REQUIRE((n_a == 0 || ids_a) && (n_b == 0 || ids_b) && keys_out && perm_out && ws, "rc_sort_ids2: null pointer");
  RC_REQUIRE(key_range >= 1 && key_range <= ((int64_t)1 << 32) && key_offset_b >= 0 && key_offset_b < key_range,
             "rc_sort_ids2: key_range=%lld / key_offset_b=%lld out of range", (long long)key_range,
             (long long)key_offset_b);
  const size_t need = rc_sort_workspace_bytes(n);
  if (ws_bytes < need) return fail(RC_ERR_WORKSPACE, "rc_sort_ids2: workspace %zu < %zu", ws_bytes, need);
  hipStream_t s = as_stream(stream);
  Carver cv(ws);
  uint32_t* keys_tmp = cv.take<uint32_t>((size_t)n);
  uint32_t* perm_tmp = cv.take<uint32_t>((size_t)n);
  const uint32_t n_tiles = sort_tiles(n);
  uint32_t* counts = cv.take<uint32_t>((size_t)kSortDigits * n_tiles);
  uint32_t* total = cv.take<uint32_t>(kSortDigits);
  const int passes = (key_bits(key_range) + 7) / 8;
  SortSrc src;
  memset(&src, 0, sizeof(src));
  src.a = ids_a; src.b = ids_b; src.na = (uint32_t)n_a; src.off_b = (uint32_t)key_offset_b;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;        // the last pass lands in the outputs
    uint32_t* ko = to_out ? keys_out : keys_tmp;
    uint32_t* po = to_out ? perm_out : perm_tmp;
    hipLaunchKernelGGL(digit_count_kernel, dim3(n_tiles), dim3(kBlock), 0, s, src, (uint32_t)n, 8 * p, n_tiles, counts);
    RC_LAUNCH_CHECK();
    hipLaunchKernelGGL(digit_offsets_kernel, dim3(kSortDigits), dim3(kBlock), 0, s, counts, n_tiles, total);
    RC_LAUNCH_CHECK();
    hipLaunchKernelGGL(digit_scatter_kernel, dim3(n_tiles), dim3(kBlock), 0, s, src, (uint32_t)n, 8 * p, n_tiles, counts, total, ko, po);
    RC_LAUNCH_CHECK();
    memset(&src, 0, sizeof(src));
    src.keys = ko; src.perm = po;
  }
  return RC_OK;
}

extern "C" int rc_sort_ids(const int64_t* ids, int64_t n, int64_t n_rows, uint32_t* keys_out,
                           uint32_t* perm_out, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "rc_sort_ids: n=%lld out of range", (long long)n);
  if (n == 0) return RC_OK;
  RC_REQUIRE(ids && keys_out && perm_out && ws, "rc_sort_ids: null pointer");
  RC_REQUIRE(n_rows >= 1 && n_rows <= ((int64_t)1 << 32), "rc_sort_ids: n_rows=%lld out of range",
             (long long)n_rows);
  return rc_sort_ids2(ids, n, nullptr, 0, 0, n_rows, keys_out, perm_out, ws, ws_bytes, stream);
}
