// library.hip -- ABI bookkeeping: version, thread-local error text, device probe.
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
