// DEVELOPMENT AID (GPU): DeviceRT::scan_multi (block_rt.hpp) against a sequential prefix / suffix sum, on integer-valued floats (every sum
// exact in fp32, so any summation order must give the same bits).  Build with -DGLAMR_SCAN_SHUFFLE for the shuffle variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../glamr_amd/csrc/block_rt.hpp"

template <bool LDS>
__global__ void probe(float* data, int nch, int n, int stride, int reverse, int chstride) {
  __shared__ float red[glamr::RT_RED_FLOATS];
  extern __shared__ __attribute__((aligned(16))) float glamr_dynamic_lds[];
  glamr::DeviceRT rt{red, nullptr};
  float* base = data;
  if (LDS) {
    for (int i = threadIdx.x; i < nch * chstride; i += blockDim.x) glamr_dynamic_lds[i] = data[i];
    __syncthreads();
    base = glamr_dynamic_lds;
  }
  float* ch[16];
  for (int c = 0; c < nch; ++c) ch[c] = base + (stride == 2 ? (c / 2) * chstride * 2 + (c & 1) : c * chstride);
  rt.template scan_multi<LDS>(ch, nch, n, stride, reverse != 0);
  __syncthreads();
  if (LDS) for (int i = threadIdx.x; i < nch * chstride; i += blockDim.x) data[i] = glamr_dynamic_lds[i];
}

int main() {
  int bad = 0, cases = 0;
  for (int lds = 0; lds < 2; ++lds)
    for (int n : {1, 2, 63, 64, 65, 100, 120, 128, 129, 192, 256, 260, 300, 511, 512, 700})
      for (int nch : {1, 2, 4})
        for (int stride : {1, 2})
          for (int rev = 0; rev < 2; ++rev) {
            if (stride == 2 && (nch & 1)) continue;
            int threads = (n + 63) / 64 * 64; if (threads > 512) threads = 512;
            const int chstride = n + 5;
            std::vector<float> h((size_t)nch * chstride * (stride == 2 ? 1 : 1) + 16, 0.f), ref;
            for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((int)((i * 2654435761u) >> 20) % 17 - 8);
            ref = h;
            for (int c = 0; c < nch; ++c) {
              float* p = ref.data() + (stride == 2 ? (c / 2) * chstride * 2 + (c & 1) : c * chstride);
              if (!rev) for (int i = 1; i < n; ++i) p[i * stride] += p[(i - 1) * stride];
              else for (int i = n - 2; i >= 0; --i) p[i * stride] += p[(i + 1) * stride];
            }
            float* d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            if (lds) hipLaunchKernelGGL(probe<true>, dim3(1), dim3(threads), h.size() * 4, 0, d, nch, n, stride, rev, chstride);
            else hipLaunchKernelGGL(probe<false>, dim3(1), dim3(threads), 0, 0, d, nch, n, stride, rev, chstride);
            std::vector<float> out(h.size());
            hipMemcpy(out.data(), d, h.size() * 4, hipMemcpyDeviceToHost); hipFree(d);
            int wrong = 0;
            for (size_t i = 0; i < h.size(); ++i) wrong += out[i] != ref[i];
            ++cases;
            if (wrong) { ++bad; printf("MISMATCH lds=%d n=%d nch=%d stride=%d reverse=%d: %d values\n", lds, n, nch, stride, rev, wrong); }
          }
  printf("%d cases, %d with mismatches\n", cases, bad);
  return bad != 0;
}
