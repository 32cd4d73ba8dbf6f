// Evaluator pieces on the device (global_recon/utils/evaluator.py:202-327): the 17-joint H36M regression from the skinned
// vertices (lib/models/smpl.py:29 JOINT_REGRESSOR_H36M; evaluator.py:262-263 `torch.matmul(J_regressor, vertices)`), the Procrustes alignment of
// every frame's joints onto the ground truth (lib/utils/torch_transform.py:282-345, batch_compute_similarity_transform_torch: 3 x 3 SVDs) and the
// per-chunk heading alignment of a trajectory (evaluator.py:202-216 get_aligned_orient_trans over traj_pred/utils/traj_utils.py:97-107
// convert_traj_world2heading).  All three are HBM / latency bound elementwise-per-frame work: one workgroup (regression) or one thread per frame.
#include "common.hpp"
#include "rotmath.hpp"

namespace glamr {
namespace {

// joints[b][j][:] = sum_v Jreg[j][v] verts[b][v][:]   -- one workgroup per frame, NJ <= 32
constexpr int REG_MAX_J = 32;
__global__ __launch_bounds__(256) void regress_joints_kernel(int B, int V, int NJ, const float* verts, const float* Jreg, float* out) {
  __shared__ float red[4][REG_MAX_J * 3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* vb = verts + (size_t)b * V * 3;
  for (int j0 = 0; j0 < NJ; j0 += 8) {                 // eight joints per pass over the frame's vertices: 24 accumulators per thread
    float acc[8][3];
    for (int j = 0; j < 8; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
    for (int v = tid; v < V; v += 256) {
      const float x = vb[v * 3 + 0], y = vb[v * 3 + 1], z = vb[v * 3 + 2];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float w = (j0 + j < NJ) ? Jreg[(size_t)(j0 + j) * V + v] : 0.f;
        acc[j][0] = fmaf(w, x, acc[j][0]); acc[j][1] = fmaf(w, y, acc[j][1]); acc[j][2] = fmaf(w, z, acc[j][2]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float s = acc[j][c];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) red[wave][(j0 + j) * 3 + c] = s;
      }
  }
  __syncthreads();
  for (int i = tid; i < NJ * 3; i += 256) out[(size_t)b * NJ * 3 + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// eigen-decomposition of a symmetric 3 x 3 matrix by cyclic Jacobi rotations (double): A = V diag(w) V^T
__device__ void jacobi3(double A[3][3], double V[3][3], double w[3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {      // A <- A J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {      // A <- J^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}
__device__ double det3(const double M[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) + M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

// Per frame: the similarity transform (scale, R, t) that maps S1 onto S2 in the least-squares sense, applied to S1.  The rotation is
// R = V Z U^T of the SVD K = U S V^T, K = X1 X2^T, Z = diag(1, 1, sign det(U V^T)) -- computed here from the eigen-decomposition of K^T K
// (V, S^2) and U = K V S^-1; R does not depend on the sign / order conventions of an SVD routine.  Double precision, one thread per frame.
__global__ __launch_bounds__(64) void procrustes_kernel(int n, int J, const float* S1, const float* S2, float* out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float* a = S1 + (size_t)i * J * 3;
  const float* b = S2 + (size_t)i * J * 3;
  double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
  for (int j = 0; j < J; ++j) for (int c = 0; c < 3; ++c) { mu1[c] += a[j * 3 + c]; mu2[c] += b[j * 3 + c]; }
  for (int c = 0; c < 3; ++c) { mu1[c] /= J; mu2[c] /= J; }
  double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
  for (int j = 0; j < J; ++j) {
    double x1[3], x2[3];
    for (int c = 0; c < 3; ++c) { x1[c] = a[j * 3 + c] - mu1[c]; x2[c] = b[j * 3 + c] - mu2[c]; var1 += x1[c] * x1[c]; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K[r][c] += x1[r] * x2[c];
  }
  double A[3][3], V[3][3], w[3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[r][c] = K[0][r] * K[0][c] + K[1][r] * K[1][c] + K[2][r] * K[2][c];      // K^T K
  jacobi3(A, V, w);
  // order the singular values: largest first (columns of V permuted alike)
  int idx[3] = {0, 1, 2};
  for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) if (w[idx[q]] > w[idx[p]]) { const int t = idx[p]; idx[p] = idx[q]; idx[q] = t; }
  double Vs[3][3], U[3][3], sv[3];
  for (int k = 0; k < 3; ++k) { sv[k] = sqrt(fmax(w[idx[k]], 0.0)); for (int r = 0; r < 3; ++r) Vs[r][k] = V[r][idx[k]]; }
  for (int k = 0; k < 2; ++k) {
    double u[3] = {0, 0, 0}, nrm = 0.0;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) u[r] += K[r][c] * Vs[c][k]; nrm += u[r] * u[r]; }
    nrm = sqrt(nrm);
    for (int r = 0; r < 3; ++r) U[r][k] = nrm > 0.0 ? u[r] / nrm : (r == k ? 1.0 : 0.0);
  }
  // the third left vector: K v3 / s3 when s3 is well above rounding, else +-(u1 x u2) -- either sign gives the same R (Z absorbs it)
  {
    double u[3] = {0, 0, 0}, nrm = 0.0;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) u[r] += K[r][c] * Vs[c][2]; nrm += u[r] * u[r]; }
    nrm = sqrt(nrm);
    const double cr[3] = {U[1][0] * U[2][1] - U[2][0] * U[1][1], U[2][0] * U[0][1] - U[0][0] * U[2][1], U[0][0] * U[1][1] - U[1][0] * U[0][1]};
    if (nrm > 1e-12 * (sv[0] + 1e-300)) for (int r = 0; r < 3; ++r) U[r][2] = u[r] / nrm;
    else for (int r = 0; r < 3; ++r) U[r][2] = cr[r];
  }
  const double z = (det3(U) * det3(Vs)) < 0.0 ? -1.0 : 1.0;      // sign det(U V^T)
  double R[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r][c] = Vs[r][0] * U[c][0] + Vs[r][1] * U[c][1] + z * Vs[r][2] * U[c][2];      // V Z U^T
  double tr = 0.0;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) tr += R[r][c] * K[c][r];
  const double scale = tr / var1;
  double t[3];
  for (int r = 0; r < 3; ++r) t[r] = mu2[r] - scale * (R[r][0] * mu1[0] + R[r][1] * mu1[1] + R[r][2] * mu1[2]);
  for (int j = 0; j < J; ++j)
    for (int r = 0; r < 3; ++r)
      out[((size_t)i * J + j) * 3 + r] = (float)(scale * (R[r][0] * a[j * 3 + 0] + R[r][1] * a[j * 3 + 1] + R[r][2] * a[j * 3 + 2]) + t[r]);
}

// evaluator.py:202-216: frame t belongs to chunk i = t / F; the chunk is expressed in the heading frame of ITS first frame -- frame i F - 1 for
// i > 0 (chunks overlap by one frame), frame 0 for the first one -- and moved so that this frame's xy is the origin (traj_utils.py:97-107 with
// apply_base_orient_after).  One thread per frame.
__global__ __launch_bounds__(64) void heading_align_kernel(int n, int F, const float* orient_aa, const float* trans, float* out_aa, float* out_trans, float* out_q) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n) return;
  const int ci = t / F, r = ci > 0 ? ci * F - 1 : 0;
  const float base[4] = {0.5f, 0.5f, 0.5f, 0.5f};
  float bc[4], q[4], qr[4], nb[4], nbr[4], hq[4], ih[4], oh[4], o[4];
  rm::quat_conj(base, bc);
  rm::aa_to_quat(orient_aa + (size_t)t * 3, q);
  rm::aa_to_quat(orient_aa + (size_t)r * 3, qr);
  rm::quat_mul(q, bc, nb);
  rm::quat_mul(qr, bc, nbr);
  rm::quat_heading_q(nbr, hq);
  rm::quat_conj(hq, ih);
  rm::quat_mul(ih, nb, oh);
  rm::quat_mul(oh, base, o);
  const float local[3] = {trans[(size_t)t * 3 + 0] - trans[(size_t)r * 3 + 0], trans[(size_t)t * 3 + 1] - trans[(size_t)r * 3 + 1], trans[(size_t)t * 3 + 2]};
  float th[3], aa[3];
  rm::quat_rotate(ih, local, th);
  rm::quat_to_aa(o, aa);
  for (int k = 0; k < 3; ++k) { out_trans[(size_t)t * 3 + k] = th[k]; out_aa[(size_t)t * 3 + k] = aa[k]; }
  if (out_q) for (int k = 0; k < 4; ++k) out_q[(size_t)t * 4 + k] = o[k];
}

}  // namespace
}  // namespace glamr

using namespace glamr;

extern "C" int glamr_eval_regress_joints(int B, int V, int n_joints, const float* verts, const float* regressor, float* joints, void* stream) {
  GLAMR_REQUIRE(verts && regressor && joints, "null argument");
  GLAMR_REQUIRE(B > 0 && V > 0 && n_joints > 0 && n_joints <= REG_MAX_J, "need B, V > 0 and 1 <= n_joints <= %d", REG_MAX_J);
  hipLaunchKernelGGL(regress_joints_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), B, V, n_joints, verts, regressor, joints);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

extern "C" int glamr_eval_procrustes(int n, int n_joints, const float* S1, const float* S2, float* S1_aligned, void* stream) {
  GLAMR_REQUIRE(S1 && S2 && S1_aligned, "null argument");
  GLAMR_REQUIRE(n > 0 && n_joints >= 3, "need n > 0 and at least 3 joints");
  hipLaunchKernelGGL(procrustes_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), n, n_joints, S1, S2, S1_aligned);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

extern "C" int glamr_eval_heading_align(int n, int align_freq, const float* orient_aa, const float* trans, float* aligned_orient_aa, float* aligned_trans,
                                        float* aligned_orient_q, void* stream) {
  GLAMR_REQUIRE(orient_aa && trans && aligned_orient_aa && aligned_trans, "null argument");
  GLAMR_REQUIRE(n > 0 && align_freq > 0, "need n > 0 and align_freq > 0");
  hipLaunchKernelGGL(heading_align_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), n, align_freq, orient_aa, trans, aligned_orient_aa,
                     aligned_trans, aligned_orient_q);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}
