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

// Kernels that sit on the pipeline's critical path (the LDS kernels between two optimiser stages: recurrence, the trajectory predictor's
// GEMMs, scene assembly, skinning) raise their waves' issue priority over the co-scheduled infiller of the other stream, whose one-wave
// workgroups share their SIMDs and have slack (user priority 0..3; the stage kernel itself runs at 3)
#ifndef GLAMR_NO_SETPRIO
#define GLAMR_CRITICAL_PATH_PRIO() __builtin_amdgcn_s_setprio(2)
#else
#define GLAMR_CRITICAL_PATH_PRIO() ((void)0)
#endif

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <typename T>
int upload(T** dst, const T* host, size_t n) {
  GLAMR_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(T)));
  GLAMR_HIP_CHECK(hipMemcpy(*dst, host, n * sizeof(T), hipMemcpyHostToDevice));
  return GLAMR_OK;
}

}  // namespace glamr
