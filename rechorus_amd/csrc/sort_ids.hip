// sort_ids.hip -- stable sort of a batch's row ids, the first half of the atomic-free
// replacement for aten::embedding_dense_backward's index_add (which the reference reaches
// through loss.backward(), helpers/BaseRunner.py:205).
//
// ids (int64, reference layout) are narrowed to uint32 keys on the fly (transform iterator: the
// first radix pass reads the int64 ids directly, no key-materialising pre-pass) and paired with
// their position (counting iterator); only ceil(log2 key_range) key bits are sorted.  Two id lists
// can be sorted in ONE call as a virtual concatenation (second list offset by `key_offset_b`):
// a BPRMF step sorts its B*(1+K) item ids and B user ids together, the user segment is the tail.
// The device-wide LSD radix sort is rocPRIM's (ROCm's header-only primitive library, stable);
// everything that consumes the sorted order is hand-written (seg_update.hip).
#include "common.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace rc {

struct ConcatKey {
  const int64_t* a;
  const int64_t* b;
  uint32_t na;
  uint32_t off_b;
  __host__ __device__ uint32_t operator()(uint32_t i) const {
    return i < na ? (uint32_t)a[i] : off_b + (uint32_t)b[i - na];
  }
};

using CountIt = rocprim::counting_iterator<uint32_t>;
using KeyIt = rocprim::transform_iterator<CountIt, ConcatKey, uint32_t>;

static int key_bits(int64_t key_range) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < key_range) ++bits;
  return bits;
}

// rocPRIM switches from merge sort (block sort + ~log2(n/1024) merge passes, two launches each) to
// onesweep radix passes at `merge_sort_limit` items, 1 M by default.  Ids are <= 24-bit keys (three 8-bit
// onesweep passes), and a BPRMF batch of 8,192 tuples x 100 candidates is 0.8 M items: measured there,
// merge sort is 20 launches / 132 us of a 419 us step.  Measured step times (tools/exp_sort_limit.sh),
// merge sort vs onesweep: 0.8 M items 0.381 vs 0.345 ms, 0.2 M items 0.214 vs 0.225 ms, 0.1 M items 0.160 vs
// 0.205 ms -> the switch sits at 512 K items.
#ifndef RC_MERGE_SORT_LIMIT
#define RC_MERGE_SORT_LIMIT (512 * 1024)
#endif
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::default_config, RC_MERGE_SORT_LIMIT>;

static hipError_t sort_call(void* temp, size_t& bytes, const ConcatKey& f, uint32_t* keys_out,
                            uint32_t* perm_out, size_t n, unsigned bits, hipStream_t s) {
  KeyIt keys_in(CountIt(0), f);
  CountIt vals_in(0);
  return rocprim::radix_sort_pairs<SortConfig>(temp, bytes, keys_in, keys_out, vals_in, perm_out, n, 0u, bits, s);
}

static size_t rocprim_temp_bytes(int64_t n) {
  size_t bytes = 0;
  ConcatKey f{nullptr, nullptr, 0, 0};
  hipError_t e = sort_call(nullptr, bytes, f, nullptr, nullptr, (size_t)n, 32u, (hipStream_t)0);
  if (e != hipSuccess) {  // e.g. no device: the size query needs the target architecture
    (void)hipGetLastError();
    fail(RC_ERR_HIP, "rocprim size query failed: %s", hipGetErrorString(e));
    return 0;
  }
  return bytes;
}

}  // namespace rc

using namespace rc;

extern "C" size_t rc_sort_workspace_bytes(int64_t n) {
  if (n <= 0) n = 1;
  return align_up(rocprim_temp_bytes(n), 256) + 256;
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
  ConcatKey f{ids_a, ids_b, (uint32_t)n_a, (uint32_t)key_offset_b};
  size_t temp_bytes = ws_bytes;
  RC_HIP(sort_call(ws, temp_bytes, f, keys_out, perm_out, (size_t)n, (unsigned)key_bits(key_range), as_stream(stream)));
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
