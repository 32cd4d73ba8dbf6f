// DEVELOPMENT AID (GPU): DeviceRT::scan_multi (block_rt.hpp) against a sequential prefix / suffix sum, on integer-valued floats (every sum
// exact in fp32, so any summation order must give the same bits).  Build with -DGLAMR_SCAN_SHUFFLE for the shuffle variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../glamr_amd/csrc/block_rt.hpp"

template <bool LDS>
__global__ void probe(float* data, int nch, int n, int stride, int reverse, int chstride) {
  __shared__ __attribute__((aligned(16))) float red[glamr::RT_RED_FLOATS];
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

// DeviceRT::scan_regs<NW, NCH> (one element per thread, in registers, NW waves): exact on integer data, and BIT-IDENTICAL to scan_multi on
// arbitrary floats (same additions in the same order) -- for both wave-scan orders
template <int NW, int NCH>
__global__ void probe_regs(float* data, float* data_multi, int n, int reverse, int shuffle_order) {
  __shared__ __attribute__((aligned(16))) float red[glamr::RT_RED_FLOATS];
  glamr::DeviceRT rt{red, nullptr};
  float x[NCH];
  for (int c = 0; c < NCH; ++c) x[c] = (int)threadIdx.x < n ? data[c * 512 + threadIdx.x] : 0.f;
  rt.template scan_regs<NW, NCH>(x, reverse != 0, shuffle_order != 0);
  for (int c = 0; c < NCH; ++c) if ((int)threadIdx.x < n) data[c * 512 + threadIdx.x] = x[c];
  __syncthreads();
  float* ch[NCH];
  for (int c = 0; c < NCH; ++c) ch[c] = data_multi + c * 512;
  rt.template scan_multi<false>(ch, NCH, n, 1, reverse != 0, shuffle_order != 0);
}

template <int NW, int NCH>
static void run_regs(int& cases, int& bad) {
  for (int n : {NW * 64, NW * 64 - 1, NW * 64 - 37, (NW - 1) * 64 + 1})
    for (int rev = 0; rev < 2; ++rev)
      for (int shuf = 0; shuf < 2; ++shuf)
        for (int integer = 0; integer < 2; ++integer) {
          if (n < 1) continue;
          std::vector<float> h((size_t)NCH * 512, 0.f), ref;
          for (size_t i = 0; i < h.size(); ++i) {
            const unsigned r = (unsigned)(i * 2654435761u + 12345u * (n + rev));
            h[i] = integer ? (float)((int)(r >> 20) % 17 - 8) : ((float)(r >> 8) / 16777216.0f - 0.5f) * (1.0f + (float)(r & 255));
          }
          ref = h;
          for (int c = 0; c < NCH; ++c) {
            float* p = ref.data() + c * 512;
            if (!rev) for (int i = 1; i < n; ++i) p[i] += p[i - 1];
            else for (int i = n - 2; i >= 0; --i) p[i] += p[i + 1];
          }
          float *d, *dm;
          hipMalloc(&d, h.size() * 4); hipMalloc(&dm, h.size() * 4);
          hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dm, h.data(), h.size() * 4, hipMemcpyHostToDevice);
          hipLaunchKernelGGL((probe_regs<NW, NCH>), dim3(1), dim3(NW * 64), 0, 0, d, dm, n, rev, shuf);
          std::vector<float> out(h.size()), outm(h.size());
          hipMemcpy(out.data(), d, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(outm.data(), dm, h.size() * 4, hipMemcpyDeviceToHost);
          hipFree(d); hipFree(dm);
          int wrong = 0;
          for (int c = 0; c < NCH; ++c)
            for (int i = 0; i < n; ++i) {
              const size_t k = (size_t)c * 512 + i;
              if (integer) wrong += out[k] != ref[k];                               // exact sums: any order gives these bits
              wrong += memcmp(&out[k], &outm[k], 4) != 0;                           // registers == in-place scan, bit for bit
            }
          ++cases;
          if (wrong) { ++bad; printf("MISMATCH scan_regs NW=%d NCH=%d n=%d reverse=%d shuffle_order=%d integer=%d: %d values\n", NW, NCH, n, rev, shuf, integer, wrong); }
        }
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
  int rcases = 0, rbad = 0;
  run_regs<1, 1>(rcases, rbad); run_regs<2, 2>(rcases, rbad); run_regs<3, 1>(rcases, rbad); run_regs<4, 2>(rcases, rbad);
  run_regs<5, 1>(rcases, rbad); run_regs<5, 2>(rcases, rbad); run_regs<7, 2>(rcases, rbad); run_regs<8, 1>(rcases, rbad);
  printf("scan_regs: %d cases, %d with mismatches\n", rcases, rbad);
  bad += rbad;
  return bad != 0;
}
