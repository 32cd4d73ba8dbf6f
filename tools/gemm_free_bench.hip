// Development aid (not product code): variants of gemm_free_kernel (nn_free.hpp) side by side -- columns per wave (32 C), prefetch depth,
// register target; activations fragment-major or row-major.  Built by tools/gemm_free_bench.py together with api_common.cpp.
#include "../glamr_amd/csrc/nn_free.hpp"
using namespace glamr::nn;

extern "C" int gfb_launch(int variant, int frag, const float* X, int ldx, const unsigned short* Ws, size_t plane, const float* bias, float* Y, int ldy, int M, int N, int K,
                          void* stream) {
  GemmArgs a{X, nullptr, bias, nullptr, nullptr, Y, M, N, K, ldx, ldy, 0, 1, 0, ACT_NONE};
  a.Ws = Ws;
  a.ws_plane = plane;
  a.x_frag = a.y_frag = frag;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0: return launch_gemm_free_t<2, 1, 4>(st, a);      // the product's instance
    case 1: return launch_gemm_free_t<2, 2, 3>(st, a);
    case 2: return launch_gemm_free_t<4, 1, 2>(st, a);
    case 3: return launch_gemm_free_t<1, 2, 4>(st, a);
    default: return -1;
  }
}
