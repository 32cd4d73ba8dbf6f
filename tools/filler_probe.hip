// Development probe (not product code): can an LDS-free, register-light matrix-core kernel run BESIDE resident workgroups of the
// optimiser stage (153 KB of LDS, five 256-register waves: one SIMD full, three half empty) and use their idle issue slots and the
// idle matrix pipes?  `filler<MPL, SPLIT>`: every wave streams 16-byte weight fragments from an L2-resident table and issues MPL
// v_mfma_f32_16x16x32_f16 per fragment; SPLIT adds the fp32 -> (hi, lo) fp16 operand split the priors' GEMMs do per loaded float.
// Built by tools/coresidency_probe.py:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/filler_probe.hip -o tools/_filler_probe.so
#include <hip/hip_runtime.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int MPL, bool SPLIT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void filler(const float* __restrict__ W, float* __restrict__ out, int iters, int nfrag, int lds_bytes) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * WAVES + (threadIdx.x >> 6);
  float4v acc[MPL];
#pragma unroll
  for (int m = 0; m < MPL; ++m) acc[m] = float4v{0.f, 0.f, 0.f, 0.f};
  half8 b[MPL];
#pragma unroll
  for (int m = 0; m < MPL; ++m)
#pragma unroll
    for (int k = 0; k < 8; ++k) b[m][k] = (_Float16)(0.001f * (lane + m + k));
  unsigned frag = (unsigned)(wave * 977) % (unsigned)nfrag;
  for (int it = 0; it < iters; ++it) {
    const float4v* p = reinterpret_cast<const float4v*>(W) + (size_t)frag * 128 + lane * 2;
    float4v x0 = p[0], x1 = p[1];
    frag = frag + 1 == (unsigned)nfrag ? 0u : frag + 1;
    half8 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (_Float16)x0[k];
      hi[4 + k] = (_Float16)x1[k];
    }
    if (SPLIT) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lo[k] = (_Float16)(x0[k] - (float)hi[k]);
        lo[4 + k] = (_Float16)(x1[k] - (float)hi[4 + k]);
      }
    }
#pragma unroll
    for (int m = 0; m < MPL; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, b[m], acc[m], 0, 0, 0);
      if (SPLIT && (m % 3) == 2) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, b[m], acc[m], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < MPL; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  if (lds_bytes > 0) {
    lds[threadIdx.x] = s;
    __syncthreads();
    s = lds[threadIdx.x ^ 1];
  }
  if (s == 12345.678f) out[wave * 64 + lane] = s;
}

extern "C" int filler_launch(int variant, int grid, const float* W, float* out, int iters, int nfrag, int lds_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0: hipLaunchKernelGGL((filler<8, false, 1>), dim3(grid), dim3(64), lds_bytes, st, W, out, iters, nfrag, lds_bytes); break;
    case 1: hipLaunchKernelGGL((filler<8, true, 1>), dim3(grid), dim3(64), lds_bytes, st, W, out, iters, nfrag, lds_bytes); break;
    case 2: hipLaunchKernelGGL((filler<16, true, 1>), dim3(grid), dim3(64), lds_bytes, st, W, out, iters, nfrag, lds_bytes); break;
    case 3: hipLaunchKernelGGL((filler<8, true, 4>), dim3(grid / 4), dim3(256), lds_bytes, st, W, out, iters, nfrag, lds_bytes); break;
    case 4: hipLaunchKernelGGL((filler<2, true, 1>), dim3(grid), dim3(64), lds_bytes, st, W, out, iters, nfrag, lds_bytes); break;
    case 5: hipLaunchKernelGGL((filler<8, true, 2>), dim3(grid / 2), dim3(128), lds_bytes, st, W, out, iters, nfrag, lds_bytes); break;
    case 6: hipLaunchKernelGGL((filler<8, true, 3>), dim3(grid / 3), dim3(192), lds_bytes, st, W, out, iters, nfrag, lds_bytes); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
