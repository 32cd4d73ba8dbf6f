// Shared plumbing of libglamr_hip.so: error reporting, HIP call checking, small device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <string>
#include "../../include/glamr_hip.h"

namespace glamr {

std::string& last_error_ref();
int fail(int code, const char* fmt, ...);

#define GLAMR_HIP_CHECK(expr)                                                                                         \
  do {                                                                                                               \
    hipError_t _e = (expr);                                                                                          \
    if (_e != hipSuccess) return ::glamr::fail(GLAMR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                                              __FILE__, __LINE__);                                                   \
  } while (0)

#define GLAMR_REQUIRE(cond, ...)                                                                                      \
  do {                                                                                                               \
    if (!(cond)) return ::glamr::fail(GLAMR_E_INVALID, __VA_ARGS__);                                                  \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <typename T>
int upload(T** dst, const T* host, size_t n) {
  GLAMR_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(T)));
  GLAMR_HIP_CHECK(hipMemcpy(*dst, host, n * sizeof(T), hipMemcpyHostToDevice));
  return GLAMR_OK;
}

}  // namespace glamr
