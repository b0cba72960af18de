// sort_ids.hip -- stable sort of a batch's row ids, the first half of the atomic-free
// replacement for aten::embedding_dense_backward's index_add (which the reference reaches
// through loss.backward(), helpers/BaseRunner.py:205).
//
// ids (int64, reference layout) are narrowed to uint32 keys and paired with their
// position; only ceil(log2 n_rows) key bits are sorted.  The device-wide LSD radix sort is
// rocPRIM's (ROCm's header-only primitive library, stable); everything that consumes the
// sorted order is hand-written (seg_update.hip).
#include "common.hpp"

#include <rocprim/device/device_radix_sort.hpp>

namespace rc {

__global__ __launch_bounds__(kBlock) void make_keys_kernel(const int64_t* __restrict__ ids,
                                                           int64_t n,
                                                           uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ vals) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    keys[i] = (uint32_t)ids[i];
    vals[i] = (uint32_t)i;
  }
}

static int key_bits(int64_t n_rows) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < n_rows) ++bits;
  return bits;
}

static size_t rocprim_temp_bytes(int64_t n) {
  size_t bytes = 0;
  uint32_t* nul = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)n, 0u,
                                           32u, (hipStream_t)0);
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
  return align_up((size_t)n * 4, 256) * 2 + align_up(rocprim_temp_bytes(n), 256) + 256;
}

extern "C" int rc_sort_ids(const int64_t* ids, int64_t n, int64_t n_rows, uint32_t* keys_out,
                           uint32_t* perm_out, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(ids && keys_out && perm_out && ws, "rc_sort_ids: null pointer");
  RC_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "rc_sort_ids: n=%lld out of range", (long long)n);
  RC_REQUIRE(n_rows >= 1 && n_rows <= ((int64_t)1 << 32), "rc_sort_ids: n_rows=%lld out of range",
             (long long)n_rows);
  if (n == 0) return RC_OK;
  if (ws_bytes < rc_sort_workspace_bytes(n))
    return fail(RC_ERR_WORKSPACE, "rc_sort_ids: workspace %zu < %zu", ws_bytes,
                rc_sort_workspace_bytes(n));
  hipStream_t s = as_stream(stream);
  Carver cv(ws);
  uint32_t* keys_in = cv.take<uint32_t>((size_t)n);
  uint32_t* vals_in = cv.take<uint32_t>((size_t)n);
  void* temp = cv.base + cv.off;
  size_t temp_bytes = ws_bytes - cv.off;
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(make_keys_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, ids, n, keys_in,
                     vals_in);
  RC_LAUNCH_CHECK();
  RC_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, perm_out,
                                   (size_t)n, 0u, (unsigned)key_bits(n_rows), s));
  return RC_OK;
}
