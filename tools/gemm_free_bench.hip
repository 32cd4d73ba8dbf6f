// Development aid (not product code): variants of gemm_free_kernel (nn_free.hpp) side by side -- tile shape (R x C 32-blocks), prefetch
// depth, register target.  Built by tools/gemm_free_bench.py together with api_common.cpp.
#include "../glamr_amd/csrc/nn_free.hpp"
using namespace glamr::nn;

extern "C" int gfb_launch(int variant, const float* X, int ldx, const unsigned short* Ws, size_t plane, const float* bias, float* Y, int ldy, int M, int N, int K,
                          void* stream) {
  GemmArgs a{X, nullptr, bias, nullptr, nullptr, Y, M, N, K, ldx, ldy, 0, 1, 0, ACT_NONE};
  a.Ws = Ws;
  a.ws_plane = plane;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 1: return launch_gemm_free_t<1, 2, 1, 4>(st, a);
    case 2: return launch_gemm_free_t<1, 2, 1, 4, 2>(st, a);
    case 3: return launch_gemm_free_t<1, 2, 1, 4, 3>(st, a);
    case 4: return launch_gemm_free_t<1, 4, 1, 2, 3>(st, a);
    case 5: return launch_gemm_free_t<1, 1, 2, 4, 3>(st, a);
    default: return -1;
  }
}
