// Workgroup runtime used by the templated per-scene / per-sequence algorithms on the device: thread id, barrier, block-wide sum,
// block-wide in-place prefix / suffix sum (wave shuffles + one LDS exchange per 64-lane wave).
#pragma once
#include <hip/hip_runtime.h>

namespace glamr {

struct DeviceRT {
  float* red;   // LDS: [16] wave partials + [1] carry
  __device__ __forceinline__ int tid() const { return threadIdx.x; }
  __device__ __forceinline__ int nthreads() const { return blockDim.x; }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ float reduce_sum(float v) const {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];
    return s;
  }
  // in-place inclusive prefix (or suffix) sum over a[i*stride], i in [0,n); ends with a barrier
  __device__ void scan(float* a, int n, int stride, bool reverse) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    float carry = 0.f;
    for (int base = 0; base < n; base += blockDim.x) {
      const int i = base + threadIdx.x;
      const size_t idx = (size_t)(reverse ? (n - 1 - i) : i) * stride;
      float x = (i < n) ? a[idx] : 0.f;
      for (int off = 1; off < 64; off <<= 1) {
        const float y = __shfl_up(x, off);
        if (lane >= off) x += y;
      }
      __syncthreads();
      if (lane == 63) red[wave] = x;
      __syncthreads();
      float pre = carry;
      for (int w = 0; w < wave; ++w) pre += red[w];
      float tot = 0.f;
      for (int w = 0; w < nw; ++w) tot += red[w];
      if (i < n) a[idx] = x + pre;
      carry += tot;
    }
    __syncthreads();
  }
};


}  // namespace glamr
