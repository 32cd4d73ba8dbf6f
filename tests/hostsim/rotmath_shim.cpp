// TEST INFRASTRUCTURE: exposes the scalar forward/backward pairs of glamr_amd/csrc/rotmath.hpp to Python (g++ build, host
// only) so they can be compared with torch autograd of the oracle.  Not part of the product library.
#include "../../glamr_amd/csrc/rotmath.hpp"
using namespace glamr::rm;

#define BATCH(name, NIN, NOUT, FWD, BWD)                                                            \
  extern "C" void name(int n, const float* x, const float* gout, float* out, float* gx) {          \
    for (int i = 0; i < n; ++i) {                                                                  \
      const float* xi = x + (size_t)i * NIN;                                                       \
      float* oi = out + (size_t)i * NOUT;                                                          \
      FWD;                                                                                         \
      float* gi = gx + (size_t)i * NIN;                                                            \
      for (int k = 0; k < NIN; ++k) gi[k] = 0.f;                                                   \
      const float* go = gout + (size_t)i * NOUT;                                                   \
      BWD;                                                                                         \
    }                                                                                              \
  }

BATCH(t_rot6d_to_rotmat, 6, 9, rot6d_to_rotmat(xi, oi), rot6d_to_rotmat_bwd(xi, go, gi))
BATCH(t_rotmat_to_quat, 9, 4, rotmat_to_quat(xi, oi), rotmat_to_quat_bwd(xi, go, gi))
BATCH(t_quat_to_aa, 4, 3, quat_to_aa(xi, oi), quat_to_aa_bwd(xi, go, gi))
BATCH(t_aa_to_quat, 3, 4, aa_to_quat(xi, oi), aa_to_quat_bwd(xi, go, gi))
BATCH(t_aa_to_rotmat_k, 3, 9, aa_to_rotmat_k(xi, oi), aa_to_rotmat_k_bwd(xi, go, gi))
BATCH(t_aa_to_rotmat_s, 3, 9, aa_to_rotmat_s(xi, oi), aa_to_rotmat_s_bwd(xi, go, gi))
BATCH(t_rotmat_to_aa, 9, 3, rotmat_to_aa(xi, oi), rotmat_to_aa_bwd(xi, go, gi))
BATCH(t_quat_mul, 8, 4, quat_mul(xi, xi + 4, oi), quat_mul_bwd(xi, xi + 4, go, gi, gi + 4))
BATCH(t_atan2s, 2, 1, oi[0] = atan2s(xi[0], xi[1]), atan2s_bwd(xi[0], xi[1], go[0], gi[0], gi[1]))
BATCH(t_normalize3, 3, 3, normalize3(xi, oi), normalize3_bwd(xi, go, gi))

// sine / cosine of the optimiser's heading angles (Cody-Waite + minimax polynomials): out = [sin, cos] per input
extern "C" void t_sincos(int n, const float* x, float* out) {
  for (int i = 0; i < n; ++i) sincos_(x[i], out[2 * i], out[2 * i + 1]);
}
