// library.hip -- ABI bookkeeping: version, thread-local error text, device probe; staging of a batch for a replayed step.
#include "common.hpp"

namespace rc {
char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace rc

extern "C" int rc_version(void) { return 1; }

extern "C" const char* rc_last_error_string(void) { return rc::last_error_buf(); }

extern "C" int rc_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return rc::fail(RC_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  return n;
}

// ---- a batch's id tensors into the static buffer of a captured step, one launch ----------------------------------------------
namespace rc {
__global__ __launch_bounds__(kBlock) void stage_batch_kernel(const int64_t* __restrict__ a, int64_t na, const int64_t* __restrict__ b,
                                                             int64_t nb, const int64_t* __restrict__ c, int64_t nc,
                                                             int64_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= na + nb + nc) return;
  dst[i] = i < na ? a[i] : (i < na + nb ? b[i - na] : c[i - na - nb]);
}
}  // namespace rc

extern "C" int rc_stage_batch(const int64_t* a, int64_t na, const int64_t* b, int64_t nb, const int64_t* c, int64_t nc,
                              int64_t* dst, rc_stream_t stream) {
  RC_REQUIRE(na >= 0 && nb >= 0 && nc >= 0 && (na == 0 || a) && (nb == 0 || b) && (nc == 0 || c), "rc_stage_batch: bad arguments");
  const int64_t n = na + nb + nc;
  if (n == 0) return RC_OK;
  RC_REQUIRE(dst != nullptr && n < ((int64_t)1 << 38), "rc_stage_batch: null destination / too large");
  hipLaunchKernelGGL(rc::stage_batch_kernel, dim3((unsigned)((n + rc::kBlock - 1) / rc::kBlock)), dim3(rc::kBlock), 0,
                     rc::as_stream(stream), a, na, b, nb, c, nc, dst);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
