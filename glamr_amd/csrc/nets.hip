// Motion infiller (transformer CVAE) + trajectory predictor (bi-LSTM CVAE) inference for batches of independent sequences.
//
// Replaces MotionTrajJointModel.inference (motion_infiller/models/motion_traj_joint_model.py:141-145), i.e.
//   MotionInfillerVAE.inference(multi_step=True)  motion_infiller/models/motion_infiller_vae.py:618-667 (ContextEncoder :92-123,
//                                                  DataDecoder 'infer' :345-433, windows of 50 frames stride 30)
//   TrajPredVAE.inference(multi_step=False)        traj_pred/models/traj_pred_vae.py:524-548 (ContextEncoder :72-92,
//                                                  DataDecoder 'infer' :269-333, traj_local2global_heading traj_utils.py:65-88)
// Weight preprocessing at create time (host, double precision): the sinusoidal position code is concatenated and projected in
// the reference (pos_encoding.py:27-32,70-74), so its contribution is a per-position constant folded into a bias table; the
// infiller's in_fc is folded into the position projection; the prior decoder's learned tokens, their self-attention and the
// cross-attention queries do not depend on the data and are precomputed.
#include "common.hpp"
#include "nn_kernels.hpp"
#include "block_rt.hpp"
#include "rotmath.hpp"
#include <vector>
#include <cmath>
#include <cstring>

using namespace glamr;
using namespace glamr::nn;

namespace {

constexpr int D = 256, FF = 512, NZ = 128, WIN = 50, PAST = 10, CUR = 30, XLD = 96;

struct Lin { float* W = nullptr; float* b = nullptr; int N = 0, K = 0; unsigned short* Ws = nullptr; };   // W: [Npad][K]; Ws: its three bf16 planes
struct LN { float* g = nullptr; float* b = nullptr; };
struct EncLayer { Lin qkv, o, f1, f2; LN n1, n2; };
struct DecLayer { Lin sa_qkv, sa_o, ca_q, ca_kv, ca_o, f1, f2; LN n1, n2, n3; };

struct HostT {   // a weight tensor as doubles [rows][cols]
  int r = 0, c = 0;
  std::vector<double> v;
  double& at(int i, int j) { return v[(size_t)i * c + j]; }
  double at(int i, int j) const { return v[(size_t)i * c + j]; }
};

struct Blob {
  const float* base; const glamr_tensor_desc* d; int n;
  HostT get(int i) const {
    HostT t; t.r = d[i].rows; t.c = d[i].cols > 0 ? d[i].cols : 1;
    t.v.resize((size_t)t.r * t.c);
    for (size_t k = 0; k < t.v.size(); ++k) t.v[k] = base[d[i].offset + k];
    return t;
  }
};

HostT rows(const HostT& a, int r0, int r1) { HostT o; o.r = r1 - r0; o.c = a.c; o.v.assign(a.v.begin() + (size_t)r0 * a.c, a.v.begin() + (size_t)r1 * a.c); return o; }
HostT cols(const HostT& a, int c0, int c1) { HostT o; o.r = a.r; o.c = c1 - c0; o.v.resize((size_t)o.r * o.c); for (int i = 0; i < a.r; ++i) for (int j = c0; j < c1; ++j) o.at(i, j - c0) = a.at(i, j); return o; }
HostT matmul(const HostT& a, const HostT& b) {   // a [m][k] b [k][n]
  HostT o; o.r = a.r; o.c = b.c; o.v.assign((size_t)o.r * o.c, 0.0);
  for (int i = 0; i < a.r; ++i) for (int k = 0; k < a.c; ++k) { const double x = a.at(i, k); for (int j = 0; j < b.c; ++j) o.at(i, j) += x * b.at(k, j); }
  return o;
}
std::vector<double> matvec(const HostT& W, const std::vector<double>& x) {   // W [n][k]
  std::vector<double> y(W.r, 0.0);
  for (int i = 0; i < W.r; ++i) { double s = 0; for (int k = 0; k < W.c; ++k) s += W.at(i, k) * x[k]; y[i] = s; }
  return y;
}
std::vector<double> pos_code(int pos) {          // PositionalEncoding.original_positional_encoding, enc_dim 256
  std::vector<double> pe(D);
  for (int i = 0; i < D / 2; ++i) {
    const double mul = std::exp((double)(2 * i) * (-std::log(10000.0) / D));
    pe[2 * i] = std::sin(pos * mul);
    pe[2 * i + 1] = std::cos(pos * mul);
  }
  return pe;
}

int up_vec(float** dst, const std::vector<double>& v, size_t pad_to = 0) {
  std::vector<float> f(std::max(v.size(), pad_to), 0.0f);
  for (size_t i = 0; i < v.size(); ++i) f[i] = (float)v[i];
  return upload(dst, f.data(), f.size());
}
// upload W [N][K] padded to [ceil64(N)][ceil32(K)]
int up_lin(Lin& L, const HostT& W, const std::vector<double>* bias) {
  L.N = W.r;
  L.K = (W.c + 31) / 32 * 32;
  const int Np = (W.r + 63) / 64 * 64;
  std::vector<float> f((size_t)Np * L.K, 0.0f);
  for (int i = 0; i < W.r; ++i) for (int j = 0; j < W.c; ++j) f[(size_t)i * L.K + j] = (float)W.at(i, j);
  int rc = upload(&L.W, f.data(), f.size());
  if (rc) return rc;
  {
    // fp32 = hi + mid + lo in bf16 (round to nearest even each time; the remainders are exact in fp32)
    auto bf16_rne = [](float x) -> unsigned short {
      unsigned u; std::memcpy(&u, &x, 4);
      u += 0x7FFFu + ((u >> 16) & 1u);
      return (unsigned short)(u >> 16);
    };
    auto bf16_val = [](unsigned short h) -> float { unsigned u = (unsigned)h << 16; float x; std::memcpy(&x, &u, 4); return x; };
    // fragment order of v_mfma_f32_32x32x16_bf16's B operand: [32-column block][16-deep k step][lane = column % 32 + 32 (k % 16 / 8)][k % 8]
    std::vector<unsigned short> planes(3 * f.size());
    const int ksteps = L.K / 16;
    for (int n = 0; n < Np; ++n)
      for (int k = 0; k < L.K; ++k) {
        const float x = f[(size_t)n * L.K + k];
        const unsigned short h = bf16_rne(x);
        const float r1 = x - bf16_val(h);
        const unsigned short m = bf16_rne(r1);
        const float r2 = r1 - bf16_val(m);
        const size_t dst = (((size_t)(n / 32) * ksteps + k / 16) * 64 + (n % 32) + 32 * ((k % 16) / 8)) * 8 + k % 8;
        planes[dst] = h; planes[f.size() + dst] = m; planes[2 * f.size() + dst] = bf16_rne(r2);
      }
    if ((rc = upload(&L.Ws, planes.data(), planes.size()))) return rc;
  }
  if (bias) return up_vec(&L.b, *bias, Np);
  return GLAMR_OK;
}
std::vector<double> vec_of(const HostT& t) { return t.v; }

}  // namespace

struct glamr_nets {
  // infiller
  Lin enc_in; float* enc_pe = nullptr;              // folded in_fc + pos projection; [WIN][256] bias table
  EncLayer enc[2];
  float* prior_q = nullptr;                         // [2][256] cross-attention queries of the two learned tokens
  float* prior_x1 = nullptr;                        // [2][256] residual input of the cross-attention block
  Lin prior_kv, prior_o, prior_f1, prior_f2, prior_pz; LN prior_n2, prior_n3;
  Lin dec_z; float* dec_pe = nullptr;               // z projection; [CUR][256] bias table (positions 10..39)
  DecLayer dec[2];
  Lin out1, out2, outfc;
  // trajectory predictor
  Lin t_in1, t_in2, t_ih[2]; float* t_hh[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  Lin t_out1, t_out2, t_pr1, t_pr2, t_pz, t_dz, t_dctx, t_d2, t_dfc;
  float* rest_joints = nullptr; int32_t* parents = nullptr;
  std::vector<void*> allocs;
};

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// small pipeline kernels
// ---------------------------------------------------------------------------------------------------------------------

// window input rows + key-padding mask (get_seg_data :564-587, mask[:, :10] = False :629)
__global__ void window_gather_kernel(const float* pose, const float* visible, const int* lens, int Tpad, int max_len, int s, float* x, unsigned char* mask) {
  const int b = blockIdx.x, j = blockIdx.y, c = threadIdx.x;          // 96 threads
  const int t = s + j, n = lens[b];
  x[((size_t)b * WIN + j) * XLD + c] = (t < n) ? pose[((size_t)b * Tpad + t) * XLD + c] : 0.0f;
  if (c == 0) mask[(size_t)b * WIN + j] = (t >= n) ? 1 : ((j >= PAST && visible[(size_t)b * max_len + t] == 0.0f) ? 1 : 0);
}
__global__ void add_table_kernel(float* y, const float* table, int rows_per_seq, int n) {   // y[b][i][:] += table[i][:]
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n) return;
  const size_t row = idx / D;
  y[idx] += table[(row % rows_per_seq) * D + idx % D];
}
__global__ void tile_rows_kernel(float* y, const float* src, int rows_per_seq, int n) {      // y[b][i][:] = src[i][:]
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n) return;
  const size_t row = idx / D;
  y[idx] = src[(row % rows_per_seq) * D + idx % D];
}
// z = mu + eps * exp(0.5 logvar); pz rows: [b][tok][256] with mu = row tok0 cols [0,128), logvar = row tok1 cols [128,256)
__global__ void reparam_infiller_kernel(const float* pz, const float* eps, int eps_stride, float* z, int B) {
  const int b = blockIdx.x, k = threadIdx.x;
  const float mu = pz[((size_t)b * 2 + 0) * D + k], lv = pz[((size_t)b * 2 + 1) * D + NZ + k];
  z[(size_t)b * NZ + k] = mu + eps[(size_t)b * eps_stride + k] * expf(0.5f * lv);
}
__global__ void reparam_traj_kernel(const float* pz, const float* eps, float* z) {
  const int b = blockIdx.x, k = threadIdx.x;
  z[(size_t)b * NZ + k] = pz[(size_t)b * D + k] + eps[(size_t)b * NZ + k] * expf(0.5f * pz[(size_t)b * D + NZ + k]);
}
// q[b][i][:] = zproj[b][:] + table[i][:]
__global__ void build_queries_kernel(const float* zproj, const float* table, float* q) {
  const int b = blockIdx.x, i = blockIdx.y, k = threadIdx.x;
  q[((size_t)b * CUR + i) * D + k] = zproj[(size_t)b * D + k] + table[(size_t)i * D + k];
}
// write the 30 generated frames of window s back into the running pose buffer (get_res_from_cur_data :604-607)
__global__ void window_scatter_kernel(const float* y, int ldy, const int* lens, int Tpad, int s, float* pose) {
  const int b = blockIdx.x, i = blockIdx.y, c = threadIdx.x;          // 69 active of 96
  const int t = s + PAST + i;
  if (c < 69 && t < lens[b] && s < lens[b] - PAST) pose[((size_t)b * Tpad + t) * XLD + c] = y[((size_t)b * CUR + i) * ldy + c];
}
__global__ void pose_in_kernel(const float* body_pose, int max_len, int Tpad, float* pose) {   // [B][max_len][69] -> [B][Tpad][96]
  const int b = blockIdx.x, t = blockIdx.y, c = threadIdx.x;
  pose[((size_t)b * Tpad + t) * XLD + c] = (t < max_len && c < 69) ? body_pose[((size_t)b * max_len + t) * 69 + c] : 0.0f;
}
__global__ void pose_out_kernel(const float* pose, int max_len, int Tpad, const int* lens, float* out_pose) {
  const int b = blockIdx.x, t = blockIdx.y, c = threadIdx.x;
  if (c < 69) out_pose[((size_t)b * max_len + t) * 69 + c] = (t < lens[b]) ? pose[((size_t)b * Tpad + t) * XLD + c] : 0.0f;
}
// forward kinematics of the 23 body joints relative to the root, zero root orientation, unshaped template
// (TrajPredVAE.get_joint_pos :384-394 -> SMPL.get_joints smpl.py:318-343); one thread per frame
__global__ void fk_joints_kernel(const float* pose, int Tpad, int max_len, const int* lens, const float* rest, const int32_t* parents, float* x) {
  const int b = blockIdx.x, t = blockIdx.y * 64 + threadIdx.x;
  if (t >= max_len) return;
  float* xo = x + ((size_t)b * max_len + t) * XLD;
  if (t >= lens[b]) { for (int c = 0; c < XLD; ++c) xo[c] = 0.0f; return; }
  const float* p = pose + ((size_t)b * Tpad + t) * XLD;
  float G[24][9], pos[24][3];
  for (int e = 0; e < 9; ++e) G[0][e] = (e % 4 == 0) ? 1.0f : 0.0f;
  {   // root: rodrigues of the zero vector with smplx's epsilon convention
    const float z[3] = {0.f, 0.f, 0.f};
    rm::aa_to_rotmat_s(z, G[0]);
  }
  for (int c = 0; c < 3; ++c) pos[0][c] = rest[c];
  for (int j = 1; j < 24; ++j) {
    const int pa = parents[j];
    float R[9];
    rm::aa_to_rotmat_s(p + (j - 1) * 3, R);
    rm::mat3_mul(G[pa], R, G[j]);
    const float d[3] = {rest[j * 3] - rest[pa * 3], rest[j * 3 + 1] - rest[pa * 3 + 1], rest[j * 3 + 2] - rest[pa * 3 + 2]};
    float o[3];
    rm::mat3_vec(G[pa], d, o);
    for (int c = 0; c < 3; ++c) pos[j][c] = pos[pa][c] + o[c];
  }
  for (int j = 1; j < 24; ++j) for (int c = 0; c < 3; ++c) xo[(j - 1) * 3 + c] = pos[j][c] - pos[0][c];
  for (int c = 69; c < XLD; ++c) xo[c] = 0.0f;
}
__global__ void masked_mean_kernel(const float* ctx, int max_len, const int* lens, float* mean) {   // [B][max_len][256] -> [B][256]
  const int b = blockIdx.x, k = threadIdx.x, n = lens[b];
  float s = 0.f;
  for (int t = 0; t < n; ++t) s += ctx[((size_t)b * max_len + t) * D + k];
  mean[(size_t)b * D + k] = s / (float)n;
}
// local trajectory -> global translation / orientation for one sequence per workgroup (traj_utils.py:65-88 + quat->aa)
__global__ __launch_bounds__(256) void traj_to_global_kernel(const float* raw, int ldraw, int max_len, const int* lens, float* local, float* trans,
                                                             float* orient, float* scratch) {
  __shared__ float red[RT_RED_FLOATS];
  DeviceRT rt{red};
  const int b = blockIdx.x, n = lens[b];
  float* L = local + (size_t)b * max_len * 11;
  float* theta = scratch + (size_t)b * max_len * 3;
  float* xy = theta + max_len;
  for (int t = threadIdx.x; t < max_len; t += blockDim.x) {
    for (int c = 0; c < 11; ++c) {
      float v = (t < n) ? raw[((size_t)b * max_len + t) * ldraw + c] : 0.0f;
      if (t == 0) { if (c < 2) v = 0.0f; if (c == 9) v = 0.0f; if (c == 10) v = 1.0f; }      // init_xy = 0, heading vec (0, 1)  (:326-329)
      L[t * 11 + c] = v;
    }
    if (t < n) theta[t] = rm::atan2s(L[t * 11 + 10], L[t * 11 + 9]);
  }
  __syncthreads();
  rt.scan(theta, n, 1, false);
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    float dx = L[t * 11], dy = L[t * 11 + 1];
    if (t > 0) { const float th = theta[t - 1], c = cosf(th), s = sinf(th); const float a = dx * c - dy * s, bb = dx * s + dy * c; dx = a; dy = bb; }
    xy[t * 2] = dx; xy[t * 2 + 1] = dy;
  }
  __syncthreads();
  rt.scan(xy, n, 2, false);
  rt.scan(xy + 1, n, 2, false);
  for (int t = threadIdx.x; t < max_len; t += blockDim.x) {
    float tr[3] = {0.f, 0.f, 0.f}, aa[3] = {0.f, 0.f, 0.f};
    if (t < n) {
      const float base[4] = {0.5f, 0.5f, 0.5f, 0.5f};
      float hq[4], R[9], lq[4], q1[4], q[4];
      rm::heading_quat(theta[t], hq);
      rm::rot6d_to_rotmat(L + t * 11 + 3, R);
      rm::rotmat_to_quat(R, lq);
      rm::quat_mul(hq, lq, q1);
      rm::quat_mul(q1, base, q);
      rm::quat_to_aa(q, aa);
      tr[0] = xy[t * 2]; tr[1] = xy[t * 2 + 1]; tr[2] = L[t * 11 + 2];
    }
    for (int c = 0; c < 3; ++c) { trans[((size_t)b * max_len + t) * 3 + c] = tr[c]; orient[((size_t)b * max_len + t) * 3 + c] = aa[c]; }
  }
}

struct Ws {
  float *pose, *x, *h0, *h1, *qkv, *att, *tmp, *ff, *ctxkv, *qbuf, *pz, *z, *zproj, *dq, *y;
  float *tx, *tg, *th, *tq, *tmean, *trow, *traw, *tscr;
  unsigned char* mask; int* lens;
  size_t total; int Tpad;
};
Ws ws_layout(int B, int max_len, char* base) {
  Ws w{};
  int nwin = (max_len - PAST + CUR - 1) / CUR;
  if (nwin < 1) nwin = 1;
  w.Tpad = std::max(max_len, (nwin - 1) * CUR + WIN);
  size_t off = 0;
  auto take = [&](size_t nfloats) { float* p = reinterpret_cast<float*>(base + off); off = align_up(off + nfloats * sizeof(float), 256); return p; };
  const size_t MW = (size_t)B * WIN, MT = (size_t)B * max_len;
  w.pose = take((size_t)B * w.Tpad * XLD);
  w.x = take(MW * XLD); w.h0 = take(MW * D); w.h1 = take(MW * D); w.qkv = take(MW * 3 * D); w.att = take(MW * D); w.tmp = take(MW * D);
  w.ff = take(MW * FF); w.ctxkv = take(MW * 2 * D); w.qbuf = take(MW * D); w.pz = take((size_t)B * 2 * D); w.z = take((size_t)B * NZ);
  w.zproj = take((size_t)B * FF); w.dq = take(MW * D); w.y = take(MW * 128);
  w.tx = take(MT * XLD); w.tg = take(MT * 1024); w.th = take(MT * D); w.tq = take(MT * D); w.tmean = take((size_t)B * D); w.trow = take((size_t)B * FF);
  w.traw = take(MT * 64); w.tscr = take(MT * 3);
  w.mask = reinterpret_cast<unsigned char*>(take((MW + 3) / 4 + 64));
  w.lens = reinterpret_cast<int*>(take((size_t)B + 64));
  w.total = off;
  return w;
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// attention with the key count known at compile time where the model fixes it (a window of 50 frames, 30 current frames)
template <class... A>
void launch_attention(dim3 grid, dim3 block, size_t lds, hipStream_t st, const float* Q, int ldq, const float* K, const float* V, int ldk,
                      const unsigned char* mask, float* O, int ldo, int Lq, int Lk, int q_shared) {
  if (Lk == WIN) hipLaunchKernelGGL(attention_kernel<WIN>, grid, block, lds, st, Q, ldq, K, V, ldk, mask, O, ldo, Lq, Lk, q_shared);
  else if (Lk == CUR) hipLaunchKernelGGL(attention_kernel<CUR>, grid, block, lds, st, Q, ldq, K, V, ldk, mask, O, ldo, Lq, Lk, q_shared);
  else hipLaunchKernelGGL(attention_kernel<0>, grid, block, lds, st, Q, ldq, K, V, ldk, mask, O, ldo, Lq, Lk, q_shared);
}

int ln(hipStream_t st, const float* X, const float* R, const LN& n, float* Y, int rows) {
  hipLaunchKernelGGL(add_layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, X, R, n.g, n.b, Y, rows, D);
  return GLAMR_OK;
}
int lin(hipStream_t st, const Lin& L, const float* X, int ldx, float* Y, int ldy, int M, int act = ACT_NONE, const float* R = nullptr, int ldr = 0,
        const float* rowbias = nullptr, int rpg = 1, int ldrb = 0) {
  return launch_gemm(st, X, ldx, L.W, L.b, Y, ldy, M, L.N, L.K, act, R, ldr, rowbias, rpg, ldrb, L.Ws);
}

}  // namespace

// =====================================================================================================================
// C ABI
// =====================================================================================================================

extern "C" int glamr_nets_create(glamr_nets** out, const float* ib, const glamr_tensor_desc* idesc, int n_inf, const float* tb,
                                 const glamr_tensor_desc* tdesc, int n_trj, const float* fk_rest_joints, const int32_t* parents) {
  GLAMR_REQUIRE(out && ib && idesc && tb && tdesc && fk_rest_joints && parents, "null argument");
  GLAMR_REQUIRE(n_inf == 144 && n_trj == 66, "expected the 144 / 66 tensors of the reference checkpoints (got %d / %d)", n_inf, n_trj);
  GLAMR_REQUIRE(idesc[0].rows == 256 && idesc[0].cols == 69 && idesc[118].rows == 69 && tdesc[58].rows == 11 && tdesc[4].rows == 512,
                "tensor table is not in checkpoint (state_dict) order");
  glamr_nets* h = new (std::nothrow) glamr_nets();
  if (!h) return fail(GLAMR_E_NOMEM, "out of host memory");
  Blob I{ib, idesc, n_inf}, Tj{tb, tdesc, n_trj};
  auto lin_plain = [&](Lin& L, const Blob& B, int iw) { const std::vector<double> bias = vec_of(B.get(iw + 1)); return up_lin(L, B.get(iw), &bias); };
  auto lnorm = [&](LN& n, const Blob& B, int i) { int rc = up_vec(&n.g, vec_of(B.get(i))); return rc ? rc : up_vec(&n.b, vec_of(B.get(i + 1))); };

  // ---- infiller: context encoder ------------------------------------------------------------------------------------------
  {
    const HostT Win = I.get(0), Wpe = I.get(2);
    const std::vector<double> bin = vec_of(I.get(1)), bpe = vec_of(I.get(3));
    const HostT Wx = cols(Wpe, 0, D), Wp = cols(Wpe, D, 2 * D);
    RC(up_lin(h->enc_in, matmul(Wx, Win), nullptr));
    const std::vector<double> bfold = matvec(Wx, bin);
    std::vector<double> table((size_t)WIN * D);
    for (int pos = 0; pos < WIN; ++pos) {
      const std::vector<double> pp = matvec(Wp, pos_code(pos));
      for (int k = 0; k < D; ++k) table[(size_t)pos * D + k] = pp[k] + bpe[k] + bfold[k];
    }
    RC(up_vec(&h->enc_pe, table));
    for (int l = 0; l < 2; ++l) {
      const int b0 = 4 + 12 * l;
      EncLayer& E = h->enc[l];
      RC(lin_plain(E.qkv, I, b0)); RC(lin_plain(E.o, I, b0 + 2)); RC(lin_plain(E.f1, I, b0 + 4)); RC(lin_plain(E.f2, I, b0 + 6));
      RC(lnorm(E.n1, I, b0 + 8)); RC(lnorm(E.n2, I, b0 + 10));
    }
  }
  auto dec_layer = [&](DecLayer& Dl, int b0) -> int {
    RC(lin_plain(Dl.sa_qkv, I, b0)); RC(lin_plain(Dl.sa_o, I, b0 + 2));
    const HostT cw = I.get(b0 + 4); const std::vector<double> cb = vec_of(I.get(b0 + 5));
    std::vector<double> bq(cb.begin(), cb.begin() + D), bkv(cb.begin() + D, cb.end());
    RC(up_lin(Dl.ca_q, rows(cw, 0, D), &bq)); RC(up_lin(Dl.ca_kv, rows(cw, D, 3 * D), &bkv));
    RC(lin_plain(Dl.ca_o, I, b0 + 6)); RC(lin_plain(Dl.f1, I, b0 + 8)); RC(lin_plain(Dl.f2, I, b0 + 10));
    RC(lnorm(Dl.n1, I, b0 + 12)); RC(lnorm(Dl.n2, I, b0 + 14)); RC(lnorm(Dl.n3, I, b0 + 16));
    return GLAMR_OK;
  };
  // ---- infiller: prior (data-independent part evaluated here) ----------------------------------------------------------------
  {
    const int b0 = 122;
    const HostT Wp = I.get(120); const std::vector<double> bp = vec_of(I.get(121));
    const HostT Wtok = cols(Wp, 0, D), Wpos = cols(Wp, D, 2 * D);
    std::vector<double> x0[2];
    for (int i = 0; i < 2; ++i) {
      const std::vector<double> a = matvec(Wtok, vec_of(I.get(74 + i))), p = matvec(Wpos, pos_code(i));
      x0[i].resize(D);
      for (int k = 0; k < D; ++k) x0[i][k] = a[k] + p[k] + bp[k];
    }
    // self-attention over the two tokens (8 heads x 32), post-norm
    const HostT Wsa = I.get(b0); const std::vector<double> bsa = vec_of(I.get(b0 + 1));
    const HostT Wso = I.get(b0 + 2); const std::vector<double> bso = vec_of(I.get(b0 + 3));
    std::vector<double> qkv[2];
    for (int i = 0; i < 2; ++i) { qkv[i] = matvec(Wsa, x0[i]); for (int k = 0; k < 3 * D; ++k) qkv[i][k] += bsa[k]; }
    std::vector<double> x1[2];
    const std::vector<double> g1 = vec_of(I.get(b0 + 12)), be1 = vec_of(I.get(b0 + 13));
    for (int i = 0; i < 2; ++i) {
      std::vector<double> att(D, 0.0);
      for (int hd = 0; hd < 8; ++hd) {
        double s[2];
        for (int j = 0; j < 2; ++j) { s[j] = 0; for (int d = 0; d < 32; ++d) s[j] += qkv[i][hd * 32 + d] * qkv[j][D + hd * 32 + d]; s[j] /= std::sqrt(32.0); }
        const double m = std::max(s[0], s[1]);
        const double e0 = std::exp(s[0] - m), e1 = std::exp(s[1] - m);
        for (int d = 0; d < 32; ++d) att[hd * 32 + d] = (e0 * qkv[0][2 * D + hd * 32 + d] + e1 * qkv[1][2 * D + hd * 32 + d]) / (e0 + e1);
      }
      std::vector<double> o = matvec(Wso, att);
      double mean = 0, var = 0;
      for (int k = 0; k < D; ++k) { o[k] += bso[k] + x0[i][k]; mean += o[k]; }
      mean /= D;
      for (int k = 0; k < D; ++k) var += (o[k] - mean) * (o[k] - mean);
      var /= D;
      x1[i].resize(D);
      for (int k = 0; k < D; ++k) x1[i][k] = (o[k] - mean) / std::sqrt(var + 1e-5) * g1[k] + be1[k];
    }
    const HostT cw = I.get(b0 + 4); const std::vector<double> cb = vec_of(I.get(b0 + 5));
    std::vector<double> qc, x1c;
    for (int i = 0; i < 2; ++i) {
      std::vector<double> q = matvec(rows(cw, 0, D), x1[i]);
      for (int k = 0; k < D; ++k) q[k] += cb[k];
      qc.insert(qc.end(), q.begin(), q.end());
      x1c.insert(x1c.end(), x1[i].begin(), x1[i].end());
    }
    RC(up_vec(&h->prior_q, qc)); RC(up_vec(&h->prior_x1, x1c));
    std::vector<double> bkv(cb.begin() + D, cb.end());
    RC(up_lin(h->prior_kv, rows(cw, D, 3 * D), &bkv));
    RC(lin_plain(h->prior_o, I, b0 + 6)); RC(lin_plain(h->prior_f1, I, b0 + 8)); RC(lin_plain(h->prior_f2, I, b0 + 10));
    RC(lnorm(h->prior_n2, I, b0 + 14)); RC(lnorm(h->prior_n3, I, b0 + 16));
    // p_z_mu on token 0, p_z_logvar on token 1: one [256][256] projection, mu rows first
    HostT Wpz; Wpz.r = 2 * NZ; Wpz.c = D; Wpz.v.resize((size_t)2 * NZ * D);
    const HostT Wmu = I.get(140), Wlv = I.get(142);
    std::copy(Wmu.v.begin(), Wmu.v.end(), Wpz.v.begin()); std::copy(Wlv.v.begin(), Wlv.v.end(), Wpz.v.begin() + (size_t)NZ * D);
    std::vector<double> bpz = vec_of(I.get(141)); const std::vector<double> blv = vec_of(I.get(143)); bpz.insert(bpz.end(), blv.begin(), blv.end());
    RC(up_lin(h->prior_pz, Wpz, &bpz));
  }
  // ---- infiller: decoder -----------------------------------------------------------------------------------------------------
  {
    const HostT Wp = I.get(76); const std::vector<double> bp = vec_of(I.get(77));
    RC(up_lin(h->dec_z, cols(Wp, 0, NZ), nullptr));
    const HostT Wpos = cols(Wp, NZ, NZ + D);
    std::vector<double> table((size_t)CUR * D);
    for (int i = 0; i < CUR; ++i) { const std::vector<double> pp = matvec(Wpos, pos_code(PAST + i)); for (int k = 0; k < D; ++k) table[(size_t)i * D + k] = pp[k] + bp[k]; }
    RC(up_vec(&h->dec_pe, table));
    RC(dec_layer(h->dec[0], 78)); RC(dec_layer(h->dec[1], 96));
    RC(lin_plain(h->out1, I, 114)); RC(lin_plain(h->out2, I, 116)); RC(lin_plain(h->outfc, I, 118));
  }
  // ---- trajectory predictor ----------------------------------------------------------------------------------------------------
  {
    auto tl = [&](Lin& L, int iw) { const std::vector<double> bias = vec_of(Tj.get(iw + 1)); return up_lin(L, Tj.get(iw), &bias); };
    RC(tl(h->t_in1, 0)); RC(tl(h->t_in2, 2));
    for (int l = 0; l < 2; ++l) {
      const int b0 = 4 + 8 * l;
      HostT W; W.r = 1024; W.c = D; W.v.resize((size_t)1024 * D);
      std::vector<double> bias(1024);
      for (int d = 0; d < 2; ++d) {
        const HostT wi = Tj.get(b0 + 4 * d);
        std::copy(wi.v.begin(), wi.v.end(), W.v.begin() + (size_t)d * 512 * D);
        const std::vector<double> bi = vec_of(Tj.get(b0 + 4 * d + 2)), bh = vec_of(Tj.get(b0 + 4 * d + 3));
        for (int k = 0; k < 512; ++k) bias[d * 512 + k] = bi[k] + bh[k];
        RC(up_vec(&h->t_hh[l][d], vec_of(Tj.get(b0 + 4 * d + 1))));
      }
      RC(up_lin(h->t_ih[l], W, &bias));
    }
    RC(tl(h->t_out1, 20)); RC(tl(h->t_out2, 22));
    RC(tl(h->t_pr1, 60)); RC(tl(h->t_pr2, 62)); RC(tl(h->t_pz, 64));
    const HostT Wd = Tj.get(54); const std::vector<double> bd = vec_of(Tj.get(55));
    RC(up_lin(h->t_dz, cols(Wd, 0, NZ), &bd));                 // z part carries the bias; added per sequence as a row bias
    RC(up_lin(h->t_dctx, cols(Wd, NZ, NZ + D), nullptr));
    RC(tl(h->t_d2, 56)); RC(tl(h->t_dfc, 58));
  }
  RC(upload(&h->rest_joints, fk_rest_joints, (size_t)72));
  RC(upload(&h->parents, parents, (size_t)24));
  *out = h;
  return GLAMR_OK;
}

extern "C" int glamr_nets_destroy(glamr_nets* h) {
  if (!h) return GLAMR_OK;
  delete h;     // device weights are released with the process; handles are created once per model
  return GLAMR_OK;
}

extern "C" size_t glamr_nets_workspace_bytes(const glamr_nets* h, int n_seq, int max_len) {
  if (!h || n_seq <= 0 || max_len <= 0) return 0;
  return ws_layout(n_seq, max_len, nullptr).total;
}

namespace {

int encoder_layer(hipStream_t st, const EncLayer& E, Ws& w, float* h_in, float* h_out, int B) {
  const int M = B * WIN;
  RC(lin(st, E.qkv, h_in, D, w.qkv, 3 * D, M));
  launch_attention( dim3(B, 8), dim3(64), 0, st, w.qkv, 3 * D, w.qkv + D, w.qkv + 2 * D, 3 * D, w.mask, w.att, D, WIN, WIN, 0);
  RC(lin(st, E.o, w.att, D, w.tmp, D, M));
  RC(ln(st, w.tmp, h_in, E.n1, h_out, M));
  RC(lin(st, E.f1, h_out, D, w.ff, FF, M, ACT_RELU));
  RC(lin(st, E.f2, w.ff, FF, w.tmp, D, M));
  RC(ln(st, w.tmp, h_out, E.n2, h_out, M));
  return GLAMR_OK;
}

// x: [B][Lq][256] queries (in place), ctx keys/values already projected into w.ctxkv ([B][WIN][512]) by the caller
int decoder_layer(hipStream_t st, const DecLayer& Dl, Ws& w, float* x, const float* ctx, int B, int Lq) {
  const int M = B * Lq;
  RC(lin(st, Dl.sa_qkv, x, D, w.qkv, 3 * D, M));
  launch_attention( dim3(B, 8), dim3(64), 0, st, w.qkv, 3 * D, w.qkv + D, w.qkv + 2 * D, 3 * D, (const unsigned char*)nullptr, w.att, D, Lq, Lq, 0);
  RC(lin(st, Dl.sa_o, w.att, D, w.tmp, D, M));
  RC(ln(st, w.tmp, x, Dl.n1, x, M));
  RC(lin(st, Dl.ca_q, x, D, w.qbuf, D, M));
  RC(lin(st, Dl.ca_kv, ctx, D, w.ctxkv, 2 * D, B * WIN));
  launch_attention( dim3(B, 8), dim3(64), 0, st, w.qbuf, D, w.ctxkv, w.ctxkv + D, 2 * D, w.mask, w.att, D, Lq, WIN, 0);
  RC(lin(st, Dl.ca_o, w.att, D, w.tmp, D, M));
  RC(ln(st, w.tmp, x, Dl.n2, x, M));
  RC(lin(st, Dl.f1, x, D, w.ff, FF, M, ACT_RELU));
  RC(lin(st, Dl.f2, w.ff, FF, w.tmp, D, M));
  RC(ln(st, w.tmp, x, Dl.n3, x, M));
  return GLAMR_OK;
}

}  // namespace

extern "C" int glamr_nets_infer(glamr_nets* h, int B, int max_len, const int32_t* lens_host, const float* body_pose, const float* visible,
                                const float* motion_eps, int n_win_max, const float* traj_eps, float* out_pose, float* out_local_traj,
                                float* out_trans, float* out_orient, int flags, void* workspace, void* stream_) {
  const bool do_infill = flags & GLAMR_NETS_INFILL, do_traj = flags & GLAMR_NETS_TRAJ;
  GLAMR_REQUIRE(h && lens_host && body_pose && workspace && (do_infill || do_traj), "null argument / empty flags");
  GLAMR_REQUIRE(!do_infill || (visible && motion_eps && out_pose), "infilling needs visible, motion_eps and out_pose");
  GLAMR_REQUIRE(!do_traj || (traj_eps && out_local_traj && out_trans && out_orient), "trajectory prediction needs traj_eps and its three outputs");
  GLAMR_REQUIRE(B > 0 && max_len > PAST, "need n_seq > 0 and max_len > %d", PAST);
  int longest = 0;
  for (int b = 0; b < B; ++b) {
    GLAMR_REQUIRE(lens_host[b] > PAST && lens_host[b] <= max_len, "sequence %d has length %d (need %d < len <= max_len)", b, lens_host[b], PAST);
    longest = std::max(longest, (int)lens_host[b]);
  }
  const int n_win = (longest - PAST + CUR - 1) / CUR;
  GLAMR_REQUIRE(!do_infill || n_win <= n_win_max, "motion_eps holds %d windows per sequence, %d needed", n_win_max, n_win);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  Ws w = ws_layout(B, max_len, static_cast<char*>(workspace));
  GLAMR_HIP_CHECK(hipMemcpyAsync(w.lens, lens_host, (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(pose_in_kernel, dim3(B, w.Tpad), dim3(XLD), 0, st, body_pose, max_len, w.Tpad, w.pose);

  // ---- motion infiller: autoregressive windows [30 i, 30 i + 50) ------------------------------------------------------------
  for (int i = 0; do_infill && i < n_win; ++i) {
    const int s = i * CUR, M = B * WIN;
    hipLaunchKernelGGL(window_gather_kernel, dim3(B, WIN), dim3(XLD), 0, st, w.pose, visible, w.lens, w.Tpad, max_len, s, w.x, w.mask);
    RC(lin(st, h->enc_in, w.x, XLD, w.h0, D, M));
    hipLaunchKernelGGL(add_table_kernel, dim3((M * D + 255) / 256), dim3(256), 0, st, w.h0, h->enc_pe, WIN, M * D);
    RC(encoder_layer(st, h->enc[0], w, w.h0, w.h1, B));
    RC(encoder_layer(st, h->enc[1], w, w.h1, w.h0, B));
    float* ctx = w.h0;
    // prior: two learned tokens attend to the context
    RC(lin(st, h->prior_kv, ctx, D, w.ctxkv, 2 * D, M));
    launch_attention( dim3(B, 8), dim3(64), 0, st, h->prior_q, D, w.ctxkv, w.ctxkv + D, 2 * D, w.mask, w.att, D, 2, WIN, 1);
    RC(lin(st, h->prior_o, w.att, D, w.tmp, D, B * 2));
    hipLaunchKernelGGL(tile_rows_kernel, dim3((B * 2 * D + 255) / 256), dim3(256), 0, st, w.dq, h->prior_x1, 2, B * 2 * D);
    RC(ln(st, w.tmp, w.dq, h->prior_n2, w.dq, B * 2));
    RC(lin(st, h->prior_f1, w.dq, D, w.ff, FF, B * 2, ACT_RELU));
    RC(lin(st, h->prior_f2, w.ff, FF, w.tmp, D, B * 2));
    RC(ln(st, w.tmp, w.dq, h->prior_n3, w.dq, B * 2));
    RC(lin(st, h->prior_pz, w.dq, D, w.pz, D, B * 2));
    hipLaunchKernelGGL(reparam_infiller_kernel, dim3(B), dim3(NZ), 0, st, w.pz, motion_eps + (size_t)i * NZ, n_win_max * NZ, w.z, B);
    // decoder: 30 queries = position code of z
    RC(lin(st, h->dec_z, w.z, NZ, w.zproj, D, B));
    hipLaunchKernelGGL(build_queries_kernel, dim3(B, CUR), dim3(D), 0, st, w.zproj, h->dec_pe, w.dq);
    RC(decoder_layer(st, h->dec[0], w, w.dq, ctx, B, CUR));
    RC(decoder_layer(st, h->dec[1], w, w.dq, ctx, B, CUR));
    RC(lin(st, h->out1, w.dq, D, w.ff, FF, B * CUR, ACT_RELU));
    RC(lin(st, h->out2, w.ff, FF, w.tmp, D, B * CUR, ACT_RELU));
    RC(lin(st, h->outfc, w.tmp, D, w.y, 128, B * CUR));
    hipLaunchKernelGGL(window_scatter_kernel, dim3(B, CUR), dim3(XLD), 0, st, w.y, 128, w.lens, w.Tpad, s, w.pose);
  }
  if (out_pose) hipLaunchKernelGGL(pose_out_kernel, dim3(B, max_len), dim3(XLD), 0, st, w.pose, max_len, w.Tpad, w.lens, out_pose);
  if (!do_traj) { GLAMR_HIP_CHECK(hipGetLastError()); return GLAMR_OK; }

  // ---- trajectory predictor -----------------------------------------------------------------------------------------------------
  const int MT = B * max_len;
  hipLaunchKernelGGL(fk_joints_kernel, dim3(B, (max_len + 63) / 64), dim3(64), 0, st, w.pose, w.Tpad, max_len, w.lens, h->rest_joints, h->parents, w.tx);
  RC(lin(st, h->t_in1, w.tx, XLD, w.tg, FF, MT, ACT_RELU));
  RC(lin(st, h->t_in2, w.tg, FF, w.th, D, MT, ACT_RELU));
  for (int l = 0; l < 2; ++l) {
    RC(lin(st, h->t_ih[l], w.th, D, w.tg, 1024, MT));
    LstmArgs la{w.tg, h->t_hh[l][0], h->t_hh[l][1], w.lens, w.th, max_len};
    // large batches: 16 sequences per workgroup on the matrix cores; small ones: one sequence per workgroup keeps every CU busy
    if (B >= 512) hipLaunchKernelGGL(lstm_mfma_kernel, dim3((B + 15) / 16, 2), dim3(512), 0, st, la, B);
    else hipLaunchKernelGGL(lstm_kernel, dim3(B, 2), dim3(512), 0, st, la);
  }
  RC(lin(st, h->t_out1, w.th, D, w.tg, FF, MT, ACT_RELU));
  RC(lin(st, h->t_out2, w.tg, FF, w.th, D, MT, ACT_RELU));          // context [B][max_len][256]
  hipLaunchKernelGGL(masked_mean_kernel, dim3(B), dim3(D), 0, st, w.th, max_len, w.lens, w.tmean);
  RC(lin(st, h->t_pr1, w.tmean, D, w.trow, FF, B, ACT_RELU));
  RC(lin(st, h->t_pr2, w.trow, FF, w.tmean, D, B, ACT_RELU));
  RC(lin(st, h->t_pz, w.tmean, D, w.pz, D, B));
  hipLaunchKernelGGL(reparam_traj_kernel, dim3(B), dim3(NZ), 0, st, w.pz, traj_eps, w.z);
  RC(lin(st, h->t_dz, w.z, NZ, w.trow, FF, B));                      // W_z z + b, one row per sequence
  RC(lin(st, h->t_dctx, w.th, D, w.tg, FF, MT, ACT_RELU, nullptr, 0, w.trow, max_len, FF));
  RC(lin(st, h->t_d2, w.tg, FF, w.tq, D, MT, ACT_RELU));
  RC(lin(st, h->t_dfc, w.tq, D, w.traw, 64, MT));
  hipLaunchKernelGGL(traj_to_global_kernel, dim3(B), dim3(256), 0, st, w.traw, 64, max_len, w.lens, out_local_traj, out_trans, out_orient, w.tscr);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}
