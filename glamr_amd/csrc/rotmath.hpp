// Scalar rotation / rigid-transform maths of the GLAMR optimiser with hand-derived reverse-mode derivatives.
//
// Each forward function reproduces the reference's PyTorch formulation (branches, clamps and epsilon placement included,
// SURVEY.md App. C 8/9) so that values AND gradients agree with autograd of the reference:
//   lib/utils/torch_transform.py   quat_mul :10-28, torch_safe_atan2 :63-67, rot6d_to_rotmat :220-227, inverse_transform :274-279
//   lib/utils/konia_transform.py   angle_axis_to_rotation_matrix :234-313, rotation_matrix_to_quaternion :349-443,
//                                  quaternion_to_angle_axis :560-630, angle_axis_to_quaternion :753-826, safe_zero_division :340-343
//   smplx.lbs.batch_rodrigues      (third-party; restated in oracle/smplx_lbs.py)
// Backward functions take the upstream gradient and ACCUMULATE (+=) into the input gradients.
// Conventions: quaternions (w,x,y,z); 3x3 matrices row-major float[9]; everything fp32.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define GLAMR_HD __host__ __device__ __forceinline__
#define GLAMR_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define GLAMR_HD inline
#define GLAMR_HD_NOINLINE __attribute__((noinline))
#endif

namespace glamr {
namespace rm {

// Reciprocal and square root as ONE hardware instruction (v_rcp_f32 / v_sqrt_f32, 1 ulp).  A plain `a / b` costs eight instructions
// on gfx950 even with approximate division enabled (frexp / ldexp range scaffolding for denormal operands); every denominator of this
// file and of the optimiser is a clamped norm, a depth, (sigma^2 + x^2) or sqrt(v) + 1e-8: normal-range numbers.  The CPU test
// runtime (tests/hostsim) keeps the exact operations.
#if defined(__HIP_DEVICE_COMPILE__) && defined(GLAMR_ROTMATH_IEEE)
// init.hip (once per sequence, not a hot loop): IEEE reciprocal / root and the library sine / cosine.  The initial camera poses decide which
// solution the optimiser of a sequence with a detection gap ends in -- the unmodified reference itself changes solution when they are
// perturbed by 1e-6 (tests/golden/full_glamr_dynamic_T300_family.npz) -- so init_data is computed as closely to the reference's CPU
// operators as fp32 allows (that translation unit is also built with -ffp-contract=off).
GLAMR_HD float rcp_(float x) { return 1.0f / x; }
GLAMR_HD float sqrt_(float x) { return sqrtf(x); }
#elif defined(__HIP_DEVICE_COMPILE__)
GLAMR_HD float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
GLAMR_HD float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
#elif defined(GLAMR_HOSTSIM_ULP_NOISE)
// CPU test runtime, robustness builds (tools/diverge_probe.py --flags -DGLAMR_HOSTSIM_ULP_NOISE): the 1-ulp hardware
// approximations modelled as a pseudo-random last-bit error, to show which results depend on them
inline float ulp_noise_(float r) {
  unsigned u; __builtin_memcpy(&u, &r, 4);
  const unsigned h = (u * 2654435761u) >> 29;               // 0..7
  if (h == 0) u += 1; else if (h == 1) u -= 1;              // a quarter of the results off by one ulp
  __builtin_memcpy(&r, &u, 4);
  return r;
}
GLAMR_HD float rcp_(float x) { return ulp_noise_(1.0f / x); }
GLAMR_HD float sqrt_(float x) { return ulp_noise_(sqrtf(x)); }
#else
GLAMR_HD float rcp_(float x) { return 1.0f / x; }
GLAMR_HD float sqrt_(float x) { return sqrtf(x); }
#endif

// IEEE operations for the ONE place where the reference's exact rounding decides what happens next: the Adam update.  The zero
// cameras of a detection gap wake up one frame per iteration under 1e10-sized gradients; their first steps are +-lr to the last bit
// and the Gram-Schmidt step of the 6D rotation then sees EXACTLY (anti)parallel columns.  With quotients / roots that are 1-2 ulp off
// the optimiser ends in a neighbouring solution (9 px away in 18 of 240 frames of BASELINE configs[1] with a detection gap;
// tools/diverge_probe.py), with IEEE results it follows the reference's trajectory to 1e-4 over 500 iterations.  Everything else in
// the iteration (norms, projections, residual weights) keeps the single-instruction approximations -- measured harmless.
//   div_:     v_rcp_f32 + the Newton / residual steps of the correctly rounded expansion (what `a / b` compiles to, minus the range
//             scaling and the special-value fix-up: Adam's denominators are sqrt(v) / c + 1e-8 and c, normal numbers)
//   sqrt_rn_: v_sqrt_f32 (1 ulp) corrected by the residuals of its two neighbours; exact for 0 and every normal argument
// The host runtime uses the C operators (IEEE by definition; built with -ffp-contract=off).
#if defined(__HIP_DEVICE_COMPILE__)
GLAMR_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// (the hardware instruction and the correction apart: callers that run several independent evaluations side by side -- grecon_algo.hpp
// adam_n -- issue all the v_sqrt / v_rcp first; div_ / sqrt_rn_ are the two halves in sequence)
GLAMR_HD float hw_rcp_(float d) { return __builtin_amdgcn_rcpf(d); }
GLAMR_HD float hw_sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
GLAMR_HD float div_fix_(float n, float d, float y) {
  const float e = __builtin_fmaf(-d, y, 1.0f);
  y = __builtin_fmaf(e, y, y);
  float q = n * y;
  float r = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(r, y, q);
}
GLAMR_HD float div_(float n, float d) { return div_fix_(n, d, hw_rcp_(d)); }
GLAMR_HD float sqrt_rn_fix_(float x, float s) {
  const float dn = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1), up = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
  const float rdn = __builtin_fmaf(-dn, s, x), rup = __builtin_fmaf(-up, s, x);
  float o = rdn <= 0.0f ? dn : s;
  o = rup > 0.0f ? up : o;
  return o;
}
GLAMR_HD float sqrt_rn_(float x) { return sqrt_rn_fix_(x, hw_sqrt_(x)); }
#elif defined(GLAMR_HOSTSIM_ULP_NOISE_ADAM)
GLAMR_HD float fma_(float a, float b, float c) { return fmaf(a, b, c); }
GLAMR_HD float div_(float a, float b) { return a * rcp_(b); }
GLAMR_HD float sqrt_rn_(float x) { return sqrt_(x); }
#else
GLAMR_HD float fma_(float a, float b, float c) { return fmaf(a, b, c); }
GLAMR_HD float div_(float a, float b) { return a / b; }
GLAMR_HD float sqrt_rn_(float x) { return sqrtf(x); }
#endif

// Sine and cosine of the same angle in ~30 instructions: Cody-Waite reduction to [-pi/4, pi/4] (pi/2 split in three floats, the
// products exact through fma -- good to |x| ~ 1e5, headings are prefix sums of a few hundred wrapped increments) and the cephes
// single-precision minimax polynomials (1 ulp on the reduced range).  The device library's sinf + cosf carry a Payne-Hanek
// large-argument path each: about 700 of the 4 600 instructions of an optimiser iteration went there.  Same code on the host runtime.
GLAMR_HD void sincos_(float x, float& s, float& c) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GLAMR_ROTMATH_IEEE)
  s = sinf(x); c = cosf(x);
  return;
#endif
  const float k = rintf(x * 0.6366197466850281f);
  float r = fmaf(k, -1.5707963705062866f, x);
  r = fmaf(k, 4.371138828673793e-08f, r);
  r = fmaf(k, 1.7151245100058819e-15f, r);
  const float z = r * r;
  const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
  const int q = (int)k;
  const float a = (q & 1) ? cp : sp, b = (q & 1) ? sp : cp;          // quadrant: swap ...
  s = (q & 2) ? -a : a;                                              // ... and change signs
  c = ((q + 1) & 2) ? -b : b;
}

// ---- elementary pieces ---------------------------------------------------------------------------------------------

// torch_safe_atan2: y += eps where |y| < eps and |x| < eps
GLAMR_HD float atan2s(float y, float x, float eps = 1e-6f) {
  if (fabsf(y) < eps && fabsf(x) < eps) y += eps;
  return atan2f(y, x);
}
GLAMR_HD void atan2s_bwd(float y, float x, float g, float& gy, float& gx, float eps = 1e-6f) {
  if (fabsf(y) < eps && fabsf(x) < eps) y += eps;
  const float r = rcp_(x * x + y * y);
  gy += g * x * r;
  gx += -g * y * r;
}

// safe_zero_division: den += eps where |den| < eps
GLAMR_HD float sdiv(float num, float den, float eps = 1e-6f) {
  if (fabsf(den) < eps) den += eps;
  return num * rcp_(den);
}
GLAMR_HD void sdiv_bwd(float num, float den, float g, float& gnum, float& gden, float eps = 1e-6f) {
  if (fabsf(den) < eps) den += eps;
  const float r = rcp_(den);
  gnum += g * r;
  gden += -g * num * r * r;
}

// sqrt(clamp_min(a, eps)); torch's clamp passes the gradient where a >= eps
GLAMR_HD float sqrt_clamped(float a, float eps) { return sqrt_(fmaxf(a, eps)); }
GLAMR_HD float sqrt_clamped_bwd(float a, float eps, float g) { return (a >= eps) ? g * 0.5f * rcp_(sqrt_(a)) : 0.0f; }

// normalize(): x / clamp(||x||, min=eps)   (lib/utils/torch_transform.py:6-7)
GLAMR_HD void normalize3(const float x[3], float out[3], float eps = 1e-9f) {
  const float n = sqrt_(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  const float r = rcp_(fmaxf(n, eps));
  out[0] = x[0] * r; out[1] = x[1] * r; out[2] = x[2] * r;
}
GLAMR_HD void normalize3_bwd(const float x[3], const float g[3], float gx[3], float eps = 1e-9f) {
  const float n = sqrt_(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  const float r = rcp_(fmaxf(n, eps));
  float gd = -(g[0] * x[0] + g[1] * x[1] + g[2] * x[2]) * r * r;        // d out / d d
  for (int i = 0; i < 3; ++i) gx[i] += g[i] * r;
  if (n >= eps && n > 0.0f) {                                            // clamp passes, norm backward = x / n (0 at 0)
    const float rn = gd * rcp_(n);
    for (int i = 0; i < 3; ++i) gx[i] += rn * x[i];
  }
}

GLAMR_HD void cross3(const float a[3], const float b[3], float c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

GLAMR_HD void mat3_mul(const float A[9], const float B[9], float C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
// C = A B :  gA += gC B^T,  gB += A^T gC
GLAMR_HD void mat3_mul_bwd(const float A[9], const float B[9], const float gC[9], float gA[9], float gB[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float sa = 0.f, sb = 0.f;
      for (int k = 0; k < 3; ++k) {
        sa += gC[i * 3 + k] * B[j * 3 + k];
        sb += A[k * 3 + i] * gC[k * 3 + j];
      }
      if (gA) gA[i * 3 + j] += sa;
      if (gB) gB[i * 3 + j] += sb;
    }
}
GLAMR_HD void mat3_vec(const float A[9], const float v[3], float o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = A[i * 3 + 0] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
GLAMR_HD void mat3T_vec(const float A[9], const float v[3], float o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = A[0 * 3 + i] * v[0] + A[1 * 3 + i] * v[1] + A[2 * 3 + i] * v[2];
}

// ---- 6D <-> rotation matrix ---------------------------------------------------------------------------------------

// rot6d_to_rotmat: Gram-Schmidt; output columns are b1, b2, b3
GLAMR_HD void rot6d_to_rotmat(const float d6[6], float R[9]) {
  float b1[3], b2[3], b3[3], u[3];
  normalize3(d6, b1);
  const float dot = b1[0] * d6[3] + b1[1] * d6[4] + b1[2] * d6[5];
  for (int i = 0; i < 3; ++i) u[i] = d6[3 + i] - dot * b1[i];
  normalize3(u, b2);
  cross3(b1, b2, b3);
  for (int r = 0; r < 3; ++r) { R[r * 3 + 0] = b1[r]; R[r * 3 + 1] = b2[r]; R[r * 3 + 2] = b3[r]; }
}
GLAMR_HD void rot6d_to_rotmat_bwd(const float d6[6], const float gR[9], float gd6[6]) {
  float b1[3], b2[3], u[3];
  normalize3(d6, b1);
  const float dot = b1[0] * d6[3] + b1[1] * d6[4] + b1[2] * d6[5];
  for (int i = 0; i < 3; ++i) u[i] = d6[3 + i] - dot * b1[i];
  normalize3(u, b2);
  float gb1[3], gb2[3], gb3[3], t[3];
  for (int r = 0; r < 3; ++r) { gb1[r] = gR[r * 3 + 0]; gb2[r] = gR[r * 3 + 1]; gb3[r] = gR[r * 3 + 2]; }
  cross3(b2, gb3, t);                              // b3 = b1 x b2
  for (int i = 0; i < 3; ++i) gb1[i] += t[i];
  cross3(gb3, b1, t);
  for (int i = 0; i < 3; ++i) gb2[i] += t[i];
  float gu[3] = {0.f, 0.f, 0.f};
  normalize3_bwd(u, gb2, gu);
  const float gdot = -(gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2]);
  float ga2[3];
  for (int i = 0; i < 3; ++i) {
    ga2[i] = gu[i] + gdot * b1[i];
    gb1[i] += -dot * gu[i] + gdot * d6[3 + i];
  }
  float ga1[3] = {0.f, 0.f, 0.f};
  normalize3_bwd(d6, gb1, ga1);
  for (int i = 0; i < 3; ++i) { gd6[i] += ga1[i]; gd6[3 + i] += ga2[i]; }
}

// ---- rotation matrix -> quaternion (kornia, four candidates selected by `where`) ---------------------------------

GLAMR_HD int rotmat_to_quat_branch(const float m[9]) {
  const float tr = m[0] + m[4] + m[8];
  if (tr > 0.0f) return 0;
  if (m[0] > m[4] && m[0] > m[8]) return 1;
  return (m[4] > m[8]) ? 2 : 3;
}
// per branch: arg of the sqrt, and the three (numerator, sign) pairs in quaternion slot order
GLAMR_HD void rotmat_to_quat(const float m[9], float q[4], float eps = 1e-6f) {
  const int br = rotmat_to_quat_branch(m);
  const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
  if (br == 0) {
    const float sq = sqrt_clamped(m00 + m11 + m22 + 1.0f, eps) * 2.0f;
    q[0] = 0.25f * sq; q[1] = sdiv(m21 - m12, sq); q[2] = sdiv(m02 - m20, sq); q[3] = sdiv(m10 - m01, sq);
  } else if (br == 1) {
    const float sq = sqrt_clamped(1.0f + m00 - m11 - m22, eps) * 2.0f;
    q[0] = sdiv(m21 - m12, sq); q[1] = 0.25f * sq; q[2] = sdiv(m01 + m10, sq); q[3] = sdiv(m02 + m20, sq);
  } else if (br == 2) {
    const float sq = sqrt_clamped(1.0f + m11 - m00 - m22, eps) * 2.0f;
    q[0] = sdiv(m02 - m20, sq); q[1] = sdiv(m01 + m10, sq); q[2] = 0.25f * sq; q[3] = sdiv(m12 + m21, sq);
  } else {
    const float sq = sqrt_clamped(1.0f + m22 - m00 - m11, eps) * 2.0f;
    q[0] = sdiv(m10 - m01, sq); q[1] = sdiv(m02 + m20, sq); q[2] = sdiv(m12 + m21, sq); q[3] = 0.25f * sq;
  }
}
GLAMR_HD void rotmat_to_quat_bwd(const float m[9], const float gq[4], float gm[9], float eps = 1e-6f) {
  const int br = rotmat_to_quat_branch(m);
  // diagonal signs of the sqrt argument, slot of the 0.25*sq term, and for the other three slots: (i, j, sign) -> m_i + sign*m_j
  //           br0                 br1                 br2                 br3
  const float sd[4][3] = {{1, 1, 1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}};
  const int slot_sq[4] = {0, 1, 2, 3};
  // numerators as (index a, index b, sign): value = m[a] + sign * m[b]
  const int na[4][3] = {{7, 2, 3}, {7, 1, 2}, {2, 1, 5}, {3, 2, 5}};
  const int nb[4][3] = {{5, 6, 1}, {5, 3, 6}, {6, 3, 7}, {1, 6, 7}};
  const float ns[4][3] = {{-1, -1, -1}, {-1, 1, 1}, {-1, 1, 1}, {-1, 1, 1}};
  const int nslot[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
  const float arg = 1.0f + sd[br][0] * m[0] + sd[br][1] * m[4] + sd[br][2] * m[8];
  const float sq = sqrt_clamped(arg, eps) * 2.0f;
  float gsq = 0.25f * gq[slot_sq[br]];
  for (int k = 0; k < 3; ++k) {
    const float num = m[na[br][k]] + ns[br][k] * m[nb[br][k]];
    float gnum = 0.f;
    sdiv_bwd(num, sq, gq[nslot[br][k]], gnum, gsq);
    gm[na[br][k]] += gnum;
    gm[nb[br][k]] += ns[br][k] * gnum;
  }
  const float garg = sqrt_clamped_bwd(arg, eps, 2.0f * gsq);
  gm[0] += sd[br][0] * garg; gm[4] += sd[br][1] * garg; gm[8] += sd[br][2] * garg;
}

// ---- quaternion product (reference's 9-multiplication arrangement) -------------------------------------------------

GLAMR_HD void quat_mul(const float a[4], const float b[4], float o[4]) {
  const float w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3], w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  const float ww = (z1 + x1) * (x2 + y2);
  const float yy = (w1 - y1) * (w2 + z2);
  const float zz = (w1 + y1) * (w2 - z2);
  const float xx = ww + yy + zz;
  const float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  o[0] = qq - ww + (z1 - y1) * (y2 - z2);
  o[1] = qq - xx + (x1 + w1) * (x2 + w2);
  o[2] = qq - yy + (w1 - x1) * (y2 + z2);
  o[3] = qq - zz + (z1 + y1) * (w2 - x2);
}
GLAMR_HD void quat_mul_plain(const float a[4], const float b[4], float o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
// c = a (x) b :  ga += gc (x) conj(b),  gb += conj(a) (x) gc   (gradients treated as quaternions)
GLAMR_HD void quat_mul_bwd(const float a[4], const float b[4], const float gc[4], float* ga, float* gb) {
  if (ga) {
    const float bc[4] = {b[0], -b[1], -b[2], -b[3]};
    float t[4];
    quat_mul_plain(gc, bc, t);
    for (int i = 0; i < 4; ++i) ga[i] += t[i];
  }
  if (gb) {
    const float ac[4] = {a[0], -a[1], -a[2], -a[3]};
    float t[4];
    quat_mul_plain(ac, gc, t);
    for (int i = 0; i < 4; ++i) gb[i] += t[i];
  }
}

// ---- quaternion <-> axis-angle (kornia) ----------------------------------------------------------------------------

GLAMR_HD void quat_to_aa(const float q[4], float aa[3], float eps = 1e-6f) {
  const float c = q[0];
  const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const float s = sqrt_clamped(s2, eps);
  const float tt = 2.0f * ((c < 0.0f) ? atan2s(-s, -c) : atan2s(s, c));
  const float k = (s2 > 0.0f) ? sdiv(tt, s, eps) : 2.0f;
  aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
}
GLAMR_HD void quat_to_aa_bwd(const float q[4], const float g[3], float gq[4], float eps = 1e-6f) {
  const float c = q[0];
  const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const float s = sqrt_clamped(s2, eps);
  const bool neg = c < 0.0f;
  const float tt = 2.0f * (neg ? atan2s(-s, -c) : atan2s(s, c));
  const float k = (s2 > 0.0f) ? sdiv(tt, s, eps) : 2.0f;
  const float gk = g[0] * q[1] + g[1] * q[2] + g[2] * q[3];
  for (int i = 0; i < 3; ++i) gq[1 + i] += g[i] * k;
  if (s2 > 0.0f) {
    float gtt = 0.f, gs = 0.f;
    sdiv_bwd(tt, s, gk, gtt, gs, eps);
    float gy = 0.f, gx = 0.f;
    if (neg) {
      atan2s_bwd(-s, -c, 2.0f * gtt, gy, gx);
      gs += -gy;
      gq[0] += -gx;
    } else {
      atan2s_bwd(s, c, 2.0f * gtt, gy, gx);
      gs += gy;
      gq[0] += gx;
    }
    const float gs2 = sqrt_clamped_bwd(s2, eps, gs);
    for (int i = 0; i < 3; ++i) gq[1 + i] += 2.0f * q[1 + i] * gs2;
  }
}

GLAMR_HD void aa_to_quat(const float aa[3], float q[4], float eps = 1e-6f) {
  const float th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  const float th = sqrt_clamped(th2, eps);
  const float half = th * 0.5f;
  const bool pos = th2 > 0.0f;
  float sh, ch;
  sincos_(half, sh, ch);
  const float k = pos ? sdiv(sh, th, eps) : 0.5f;
  q[0] = pos ? ch : 1.0f;
  q[1] = aa[0] * k; q[2] = aa[1] * k; q[3] = aa[2] * k;
}
GLAMR_HD void aa_to_quat_bwd(const float aa[3], const float gq[4], float gaa[3], float eps = 1e-6f) {
  const float th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  const float th = sqrt_clamped(th2, eps);
  const float half = th * 0.5f;
  const bool pos = th2 > 0.0f;
  float sh, ch;
  sincos_(half, sh, ch);
  const float k = pos ? sdiv(sh, th, eps) : 0.5f;
  for (int i = 0; i < 3; ++i) gaa[i] += gq[1 + i] * k;
  if (pos) {
    const float gk = gq[1] * aa[0] + gq[2] * aa[1] + gq[3] * aa[2];
    float gsh = 0.f, gth = 0.f;
    sdiv_bwd(sh, th, gk, gsh, gth, eps);
    const float ghalf = gsh * ch - gq[0] * sh;
    gth += 0.5f * ghalf;
    const float gth2 = sqrt_clamped_bwd(th2, eps, gth);
    for (int i = 0; i < 3; ++i) gaa[i] += 2.0f * aa[i] * gth2;
  }
}

// ---- axis-angle -> rotation matrix, kornia variant (Taylor switch at theta^2 <= 1e-6, divides by theta + 1e-6) -----

GLAMR_HD void aa_to_rotmat_k(const float aa[3], float R[9]) {
  const float th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2 > 1e-6f) {
    const float th = sqrtf(th2);
    const float inv = 1.0f / (th + 1e-6f);
    const float wx = aa[0] * inv, wy = aa[1] * inv, wz = aa[2] * inv;
    float c, s;
    sincos_(th, s, c);
    const float k = 1.0f - c;
    R[0] = c + wx * wx * k;       R[1] = wx * wy * k - wz * s;  R[2] = wy * s + wx * wz * k;
    R[3] = wz * s + wx * wy * k;  R[4] = c + wy * wy * k;       R[5] = -wx * s + wy * wz * k;
    R[6] = -wy * s + wx * wz * k; R[7] = wx * s + wy * wz * k;  R[8] = c + wz * wz * k;
  } else {
    R[0] = 1.f;     R[1] = -aa[2]; R[2] = aa[1];
    R[3] = aa[2];   R[4] = 1.f;    R[5] = -aa[0];
    R[6] = -aa[1];  R[7] = aa[0];  R[8] = 1.f;
  }
}
GLAMR_HD void aa_to_rotmat_k_bwd(const float aa[3], const float g[9], float gaa[3]) {
  const float th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2 > 1e-6f) {
    const float th = sqrtf(th2);
    const float den = th + 1e-6f, inv = 1.0f / den;
    const float w[3] = {aa[0] * inv, aa[1] * inv, aa[2] * inv};
    float c, s;
    sincos_(th, s, c);
    const float k = 1.0f - c;
    float gww = 0.f;                                  // sum_ij g_ij w_i w_j
    float gw[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        gww += g[i * 3 + j] * w[i] * w[j];
        gw[i] += k * (g[i * 3 + j] + g[j * 3 + i]) * w[j];
      }
    const float gc = (g[0] + g[4] + g[8]) - gww;
    const float gs = -w[2] * g[1] + w[1] * g[2] + w[2] * g[3] - w[0] * g[5] - w[1] * g[6] + w[0] * g[7];
    gw[0] += s * (g[7] - g[5]);
    gw[1] += s * (g[2] - g[6]);
    gw[2] += s * (g[3] - g[1]);
    float gth = -gc * s + gs * c;
    gth += -(gw[0] * aa[0] + gw[1] * aa[1] + gw[2] * aa[2]) * inv * inv;
    const float gth2 = gth / (2.0f * th);
    for (int i = 0; i < 3; ++i) gaa[i] += gw[i] * inv + 2.0f * aa[i] * gth2;
  } else {
    gaa[0] += g[7] - g[5];
    gaa[1] += g[2] - g[6];
    gaa[2] += g[3] - g[1];
  }
}

// ---- axis-angle -> rotation matrix, smplx variant (angle = || r + 1e-8 ||) ------------------------------------------

GLAMR_HD void aa_to_rotmat_s(const float r[3], float R[9]) {
  const float ax = r[0] + 1e-8f, ay = r[1] + 1e-8f, az = r[2] + 1e-8f;
  const float angle = sqrtf(ax * ax + ay * ay + az * az);
  const float inv = 1.0f / angle;
  const float x = r[0] * inv, y = r[1] * inv, z = r[2] * inv;
  float s, cc;
  sincos_(angle, s, cc);
  const float c1 = 1.0f - cc;
  R[0] = 1.0f + c1 * (-(y * y + z * z)); R[1] = s * (-z) + c1 * (x * y);        R[2] = s * (y) + c1 * (x * z);
  R[3] = s * (z) + c1 * (x * y);         R[4] = 1.0f + c1 * (-(x * x + z * z)); R[5] = s * (-x) + c1 * (y * z);
  R[6] = s * (-y) + c1 * (x * z);        R[7] = s * (x) + c1 * (y * z);         R[8] = 1.0f + c1 * (-(x * x + y * y));
}
GLAMR_HD void aa_to_rotmat_s_bwd(const float r[3], const float gR[9], float gr[3]) {
  const float ax = r[0] + 1e-8f, ay = r[1] + 1e-8f, az = r[2] + 1e-8f;
  const float angle = sqrtf(ax * ax + ay * ay + az * az), inv = 1.0f / angle;
  const float x = r[0] * inv, y = r[1] * inv, z = r[2] * inv;
  float s, c;
  sincos_(angle, s, c);
  const float c1 = 1.0f - c;
  const float K[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  const float K2[9] = {-(y * y + z * z), x * y, x * z, x * y, -(x * x + z * z), y * z, x * z, y * z, -(x * x + y * y)};
  float g_s = 0.f, g_c1 = 0.f;
  for (int e = 0; e < 9; ++e) { g_s += gR[e] * K[e]; g_c1 += gR[e] * K2[e]; }
  const float gx = s * (gR[7] - gR[5]) + c1 * (y * (gR[1] + gR[3]) + z * (gR[2] + gR[6]) - 2.f * x * (gR[4] + gR[8]));
  const float gy = s * (gR[2] - gR[6]) + c1 * (x * (gR[1] + gR[3]) + z * (gR[5] + gR[7]) - 2.f * y * (gR[0] + gR[8]));
  const float gz = s * (gR[3] - gR[1]) + c1 * (x * (gR[2] + gR[6]) + y * (gR[5] + gR[7]) - 2.f * z * (gR[0] + gR[4]));
  const float g_angle = g_s * c + g_c1 * s - (gx * r[0] + gy * r[1] + gz * r[2]) * inv * inv;
  gr[0] += gx * inv + g_angle * ax * inv;
  gr[1] += gy * inv + g_angle * ay * inv;
  gr[2] += gz * inv + g_angle * az * inv;
}

// ---- composites ----------------------------------------------------------------------------------------------------

// rotation_matrix_to_angle_axis = quat_to_aa(rotmat_to_quat(.))
GLAMR_HD void rotmat_to_aa(const float R[9], float aa[3]) {
  float q[4];
  rotmat_to_quat(R, q);
  quat_to_aa(q, aa);
}
GLAMR_HD void rotmat_to_aa_bwd(const float R[9], const float g[3], float gR[9]) {
  float q[4], gq[4] = {0.f, 0.f, 0.f, 0.f};
  rotmat_to_quat(R, q);
  quat_to_aa_bwd(q, g, gq);
  rotmat_to_quat_bwd(R, gq, gR);
}

// ---- quaternion helpers of the trajectory predictor's training-mode inputs (lib/utils/torch_transform.py) -----------------------
GLAMR_HD void quat_conj(const float q[4], float o[4]) { o[0] = q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = -q[3]; }      // :31-35
// quat_apply :38-45
GLAMR_HD void quat_rotate(const float q[4], const float v[3], float o[3]) {
  float t[3], u[3];
  cross3(q + 1, v, t);
  for (int i = 0; i < 3; ++i) t[i] *= 2.0f;
  cross3(q + 1, t, u);
  for (int i = 0; i < 3; ++i) o[i] = v[i] + q[0] * t[i] + u[i];
}
// get_heading :172-177 = 2 atan2s(z, w);  get_heading_q :180-185 = normalize(w, 0, 0, z)
GLAMR_HD float quat_heading(const float q[4]) { return 2.0f * atan2s(q[3], q[0]); }
GLAMR_HD void quat_heading_q(const float q[4], float o[4]) {
  const float n = sqrtf(q[0] * q[0] + q[3] * q[3]);
  const float c = fmaxf(n, 1e-9f);
  o[0] = q[0] / c; o[1] = 0.f; o[2] = 0.f; o[3] = q[3] / c;
}
// quaternion_to_rotation_matrix (konia_transform.py:470-555): normalised with eps 1e-12 first
GLAMR_HD void quat_to_rotmat(const float q[4], float R[9]) {
  const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
  const float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0f - (txx + tyy);
}

// heading quaternion of angle theta: angle_axis_to_quaternion((0, 0, theta))
GLAMR_HD void heading_quat(float theta, float q[4]) {
  const float aa[3] = {0.f, 0.f, theta};
  aa_to_quat(aa, q);
}
GLAMR_HD float heading_quat_bwd(float theta, const float gq[4]) {
  const float aa[3] = {0.f, 0.f, theta};
  float g[3] = {0.f, 0.f, 0.f};
  aa_to_quat_bwd(aa, gq, g);
  return g[2];
}

}  // namespace rm
}  // namespace glamr
