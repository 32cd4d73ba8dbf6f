// Error reporting + version/device queries of libglamr_hip.so.
#include "common.hpp"

namespace glamr {
std::string& last_error_ref() {
  static thread_local std::string s;
  return s;
}
int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}
}  // namespace glamr

extern "C" int glamr_version(void) { return 100; }
extern "C" const char* glamr_last_error(void) { return glamr::last_error_ref().c_str(); }
extern "C" int glamr_device_info(int* cu_count, int* gfx_major_minor, size_t* hbm_bytes) {
  int dev = 0;
  GLAMR_HIP_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t p;
  GLAMR_HIP_CHECK(hipGetDeviceProperties(&p, dev));
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (gfx_major_minor) *gfx_major_minor = p.major * 10 + p.minor;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  return GLAMR_OK;
}
