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
#include "nn_free.hpp"
#include "block_rt.hpp"
#include "rotmath.hpp"
#include <vector>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

using namespace glamr;
using namespace glamr::nn;

namespace {

constexpr int D = 256, FF = 512, NZ = 128, WIN = 50, PAST = 10, CUR = 30, XLD = 96;

struct Lin { float* W = nullptr; float* b = nullptr; int N = 0, K = 0; unsigned short* Ws = nullptr;   // W: [Npad][K]; Ws: its two fp16 planes
             double rowabs = 0, babs = 0, wabs = 0; };      // max_j sum_k |W_jk|, max |b_j|, max |W_jk|: the range analysis of glamr_nets_create
struct LN { float* g = nullptr; float* b = nullptr; double bound = 0; };      // bound: max_k |gamma_k| sqrt(255) + |beta_k| -- no LayerNorm output exceeds it
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

// every device allocation of a handle is recorded while glamr_nets_create runs, so that glamr_nets_destroy can release it
thread_local std::vector<void*>* tl_allocs = nullptr;
template <class T> int upload_t(T** dst, const T* host, size_t n) {
  const int rc = upload(dst, host, n);
  if (!rc && tl_allocs) tl_allocs->push_back(static_cast<void*>(*dst));
  return rc;
}

int up_vec(float** dst, const std::vector<double>& v, size_t pad_to = 0) {
  std::vector<float> f(std::max(v.size(), pad_to), 0.0f);
  for (size_t i = 0; i < v.size(); ++i) f[i] = (float)v[i];
  return upload_t(dst, f.data(), f.size());
}
// W [Np][K] (Np a multiple of 64, K of 32) as two fp16 planes in the fragment order the split-fp16 kernels fetch
std::vector<unsigned short> split_planes(const std::vector<float>& f, int Np, int K) {
  // fp32 ~ hi + lo in fp16 (round to nearest even; the remainder is exact in fp32), see nn_kernels.hpp
  auto f16_bits = [](float x) -> unsigned short { const _Float16 h = (_Float16)x; unsigned short u; std::memcpy(&u, &h, 2); return u; };
  auto f16_val = [](float x) -> float { return (float)(_Float16)x; };
  // fragment order of v_mfma_f32_32x32x16_f16's B operand: [32-column block][16-deep k step][lane = column % 32 + 32 (k % 16 / 8)][k % 8]
  std::vector<unsigned short> planes(2 * f.size());
  const int ksteps = K / 16;
  for (int n = 0; n < Np; ++n)
    for (int k = 0; k < K; ++k) {
      const float x = f[(size_t)n * K + k];
      const float r1 = x - f16_val(x);
      const size_t dst = (((size_t)(n / 32) * ksteps + k / 16) * 64 + (n % 32) + 32 * ((k % 16) / 8)) * 8 + k % 8;
      planes[dst] = f16_bits(x); planes[f.size() + dst] = f16_bits(r1);
    }
  return planes;
}

// upload W [N][K] padded to [ceil64(N)][ceil32(K)]
int up_lin(Lin& L, const HostT& W, const std::vector<double>* bias) {
  L.N = W.r;
  L.K = (W.c + 31) / 32 * 32;
  L.rowabs = L.babs = L.wabs = 0;
  for (int i = 0; i < W.r; ++i) {
    double sum = 0;
    for (int j = 0; j < W.c; ++j) { const double a = std::fabs(W.at(i, j)); sum += a; L.wabs = std::max(L.wabs, a); }
    L.rowabs = std::max(L.rowabs, sum);
  }
  if (bias) for (double b : *bias) L.babs = std::max(L.babs, std::fabs(b));
  const int Np = (W.r + 63) / 64 * 64;
  std::vector<float> f((size_t)Np * L.K, 0.0f);
  for (int i = 0; i < W.r; ++i) for (int j = 0; j < W.c; ++j) f[(size_t)i * L.K + j] = (float)W.at(i, j);
  int rc = upload_t(&L.W, f.data(), f.size());
  if (rc) return rc;
  {
    const std::vector<unsigned short> planes = split_planes(f, Np, L.K);
    if ((rc = upload_t(&L.Ws, planes.data(), planes.size()))) return rc;
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
  // infiller posterior encoder (DataEncoder :126-249; forward(data) / recon only)
  Lin qe_in; float* qe_table = nullptr;             // folded in_fc + pos projection; [32][256]: rows 0,1 = the projected tokens, 2..31 = positions
  DecLayer qe[2];
  Lin qe_pz;                                        // q_z_mu on token 0, q_z_logvar on token 1
  // trajectory predictor
  Lin t_in1, t_in2, t_ih[2]; float* t_hh[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  Lin t_out1, t_out2, t_pr1, t_pr2, t_pz, t_dz, t_dctx, t_d2, t_dfc;
  // trajectory posterior encoder (DataEncoder traj_pred_vae.py:95-199; forward(data) / recon only)
  Lin te_in1, te_in2, te_ih[2]; float* te_hh[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  Lin te_out1, te_out2, te_f1, te_f2, te_qz;
  float* rest_joints = nullptr; int32_t* parents = nullptr;
  std::vector<void*> allocs;
  // glamr_nets_infer launch sequences captured as HIP graphs, keyed by batch geometry + every buffer address of the call
  struct GraphKey {
    int v[5]; const void* p[9];
    bool operator<(const GraphKey& o) const { return std::memcmp(this, &o, sizeof(GraphKey)) < 0; }
  };
  struct GraphEntry { hipGraphExec_t exec = nullptr; int seen = 0; uint64_t last_use = 0; };
  std::map<GraphKey, GraphEntry> graphs;          // at most GRAPH_CACHE_MAX entries, least recently used evicted (its executable destroyed)
  uint64_t graph_clock = 0;
  std::mutex graph_mu;
  // length tables of calls recorded into a CALLER's capture: carved from one pinned slab the handle owns (allocated at creation -- a
  // capturing thread must not allocate -- and alive until glamr_nets_destroy), so the copy node of the caller's graph has a source that
  // outlives the call
  int32_t* capture_lens = nullptr;
  size_t capture_lens_used = 0;
  // Range analysis (glamr_nets_create): worst-case magnitude of any value the fp16-split kernels convert, from the weights alone.  Above
  // fp16's range the handle runs the plain fp32 kernels everywhere (fp32_only): slower, never wrong.
  bool fp32_only = false;
  double worst_activation = 0, worst_weight = 0;
  std::map<const Lin*, Lin> lin_T;      // transposed weights of the layers the infiller's backward multiplies with (nets_tape.inc), made on first use
};
constexpr size_t GRAPH_CACHE_MAX = 24, CAPTURE_LENS_INTS = 256 * 1024;

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// small pipeline kernels
// ---------------------------------------------------------------------------------------------------------------------

__global__ void copy_ints_kernel(int* dst, const int32_t* src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
// window input rows + key-padding mask (get_seg_data :564-587, mask[:, :10] = False :629)
__global__ void window_gather_kernel(const float* pose, const float* visible, const int* lens, int Tpad, int max_len, int s, float* x, unsigned char* mask) {
  const int b = blockIdx.x, j = blockIdx.y;
  const int t = s + j, n = lens[b];
  for (int c = threadIdx.x; c < XLD; c += blockDim.x) x[((size_t)b * WIN + j) * XLD + c] = (t < n) ? pose[((size_t)b * Tpad + t) * XLD + c] : 0.0f;
  if (threadIdx.x == 0) mask[(size_t)b * WIN + j] = (t >= n) ? 1 : ((j >= PAST && visible[(size_t)b * max_len + t] == 0.0f) ? 1 : 0);
}
__global__ void tile_rows_kernel(float* y, const float* src, int rows_per_seq, int n, int frag = 0) {      // y[b][i][:] = src[i][:]
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)blockIdx.x * 256 + 256 && idx < (size_t)n; idx += blockDim.x) {
    const size_t row = idx / D;
    y[frag ? x32_off((int)row, (int)(idx % D), D) : idx] = src[(row % rows_per_seq) * D + idx % D];
  }
}
// z = mu + eps * exp(0.5 logvar); pz rows: [b][tok][256] with mu = row tok0 cols [0,128), logvar = row tok1 cols [128,256)
__global__ void reparam_infiller_kernel(const float* pz, const float* eps, int eps_stride, float* z, int B) {
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < NZ; k += blockDim.x) {
    const float mu = pz[((size_t)b * 2 + 0) * D + k], lv = pz[((size_t)b * 2 + 1) * D + NZ + k];
    z[(size_t)b * NZ + k] = mu + eps[(size_t)b * eps_stride + k] * expf(0.5f * lv);
  }
}
__global__ void reparam_traj_kernel(const float* pz, const float* eps, float* z) {
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < NZ; k += blockDim.x) z[(size_t)b * NZ + k] = pz[(size_t)b * D + k] + eps[(size_t)b * NZ + k] * expf(0.5f * pz[(size_t)b * D + NZ + k]);
}
// q[b][i][:] = zproj[b][:] + table[i][:]
__global__ void build_queries_kernel(const float* zproj, const float* table, float* q, int frag = 0) {
  const int b = blockIdx.x, i = blockIdx.y;
  for (int k = threadIdx.x; k < D; k += blockDim.x)
    q[frag ? x32_off(b * CUR + i, k, D) : ((size_t)b * CUR + i) * D + k] = zproj[(size_t)b * D + k] + table[(size_t)i * D + k];
}
// write the 30 generated frames of window s back into the running pose buffer (get_res_from_cur_data :604-607)
__global__ void window_scatter_kernel(const float* y, int ldy, const int* lens, int Tpad, int s, float* pose) {
  const int b = blockIdx.x, i = blockIdx.y;
  const int t = s + PAST + i;
  if (t < lens[b] && s < lens[b] - PAST)
    for (int c = threadIdx.x; c < 69; c += blockDim.x) pose[((size_t)b * Tpad + t) * XLD + c] = y[((size_t)b * CUR + i) * ldy + c];
}
__global__ void pose_in_kernel(const float* body_pose, int max_len, int Tpad, float* pose) {   // [B][max_len][69] -> [B][Tpad][96]
  const int b = blockIdx.x, t = blockIdx.y;
  for (int c = threadIdx.x; c < XLD; c += blockDim.x) pose[((size_t)b * Tpad + t) * XLD + c] = (t < max_len && c < 69) ? body_pose[((size_t)b * max_len + t) * 69 + c] : 0.0f;
}
__global__ void pose_out_kernel(const float* pose, int max_len, int Tpad, const int* lens, float* out_pose) {
  const int b = blockIdx.x, t = blockIdx.y;
  for (int c = threadIdx.x; c < 69; c += blockDim.x) out_pose[((size_t)b * max_len + t) * 69 + c] = (t < lens[b]) ? pose[((size_t)b * Tpad + t) * XLD + c] : 0.0f;
}
// forward kinematics of the 23 body joints relative to the root, zero root orientation, unshaped template
// (TrajPredVAE.get_joint_pos :384-394 -> SMPL.get_joints smpl.py:318-343); a thread per (frame, joint), the chain one tree level at a
// time through LDS (a thread per frame walking all 24 joints with its matrices in scratch memory took 0.33 ms per 1024 x 300 frames)
constexpr int FK_FRAMES = 10;      // 10 x 24 = 240 of 256 threads
__global__ __launch_bounds__(256) void fk_joints_kernel(const float* pose, int Tpad, int max_len, const int* lens, const float* rest, const int32_t* parents, float* x) {
  __shared__ float sG[FK_FRAMES][24][9], sP[FK_FRAMES][24][3];
  __shared__ int sLev[24];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int fl = tid / 24, j = tid % 24;
  const int t = blockIdx.y * FK_FRAMES + fl;
  if (tid < 24) {
    int lev = 0;
    for (int a = parents[tid]; a >= 0; a = parents[a]) ++lev;
    sLev[tid] = lev;
  }
  __syncthreads();
  int nlev = 0;
  for (int k = 0; k < 24; ++k) nlev = sLev[k] > nlev ? sLev[k] : nlev;
  const bool active = fl < FK_FRAMES && t < max_len && t < lens[b];
  const int pa = parents[j], lev = sLev[j];
  float R[9];
  if (active) {
    if (j == 0) { const float z[3] = {0.f, 0.f, 0.f}; rm::aa_to_rotmat_s(z, R); }      // root: rodrigues of the zero vector with smplx's epsilon convention
    else rm::aa_to_rotmat_s(pose + ((size_t)b * Tpad + t) * XLD + (j - 1) * 3, R);
    if (j == 0) {
      for (int e = 0; e < 9; ++e) sG[fl][0][e] = R[e];
      for (int c = 0; c < 3; ++c) sP[fl][0][c] = rest[c];
    }
  }
  __syncthreads();
  for (int L = 1; L <= nlev; ++L) {
    if (active && lev == L) {
      float G[9], o[3];
      rm::mat3_mul(sG[fl][pa], R, G);
      const float d[3] = {rest[j * 3] - rest[pa * 3], rest[j * 3 + 1] - rest[pa * 3 + 1], rest[j * 3 + 2] - rest[pa * 3 + 2]};
      rm::mat3_vec(sG[fl][pa], d, o);
      for (int e = 0; e < 9; ++e) sG[fl][j][e] = G[e];
      for (int c = 0; c < 3; ++c) sP[fl][j][c] = sP[fl][pa][c] + o[c];
    }
    __syncthreads();
  }
  if (fl >= FK_FRAMES || t >= max_len) return;
  float* xo = x + ((size_t)b * max_len + t) * XLD;
  if (!active) { for (int c = j * 4; c < j * 4 + 4; ++c) xo[c] = 0.0f; return; }      // 24 threads x 4 columns = the 96 of a padded row
  if (j > 0) for (int c = 0; c < 3; ++c) xo[(j - 1) * 3 + c] = sP[fl][j][c] - sP[fl][0][c];
  else for (int c = 69; c < XLD; ++c) xo[c] = 0.0f;
}
// The same chain without LDS (co-schedulable, see nn_free.hpp): two frames per wave, lane = (frame, joint); a joint fetches its parent's
// transform with twelve lane exchanges per tree level instead of reading it from LDS.  Same arithmetic in the same order.
__global__ __launch_bounds__(64) void fk_joints_free_kernel(const float* pose, int Tpad, int max_len, const int* lens, const float* rest, const int32_t* parents, float* x) {
  const int b = blockIdx.x, lane = threadIdx.x, fl = lane >> 5, j = lane & 31;
  const int t = blockIdx.y * 2 + fl;
  const bool joint = j < 24;
  const bool active = joint && t < max_len && t < lens[b];
  const int pa = joint ? parents[j] : -1;
  int lev = 0;
  if (joint) for (int a = pa; a >= 0; a = parents[a]) ++lev;
  int nlev = lev;
  for (int off = 32; off > 0; off >>= 1) nlev = max(nlev, __shfl_xor(nlev, off));
  float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, G[9], P[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
  if (active) {
    if (j == 0) { const float z[3] = {0.f, 0.f, 0.f}; rm::aa_to_rotmat_s(z, R); }
    else rm::aa_to_rotmat_s(pose + ((size_t)b * Tpad + t) * XLD + (j - 1) * 3, R);
    if (j == 0) for (int c = 0; c < 3; ++c) P[c] = rest[c];
    else for (int c = 0; c < 3; ++c) d[c] = rest[j * 3 + c] - rest[pa * 3 + c];
  }
  for (int e = 0; e < 9; ++e) G[e] = R[e];
  const int src = (lane & 32) + max(pa, 0);
  for (int L = 1; L <= nlev; ++L) {
    float pg[9], pp[3];
    for (int e = 0; e < 9; ++e) pg[e] = __shfl(G[e], src);
    for (int c = 0; c < 3; ++c) pp[c] = __shfl(P[c], src);
    if (active && lev == L) {
      float o[3];
      rm::mat3_mul(pg, R, G);
      rm::mat3_vec(pg, d, o);
      for (int c = 0; c < 3; ++c) P[c] = pp[c] + o[c];
    }
  }
  const int root = lane & 32;
  float p0[3];
  for (int c = 0; c < 3; ++c) p0[c] = __shfl(P[c], root);
  if (!joint || t >= max_len) return;
  float* xo = x + ((size_t)b * max_len + t) * XLD;
  if (!active) { for (int c = j * 4; c < j * 4 + 4; ++c) xo[c] = 0.0f; return; }
  if (j > 0) for (int c = 0; c < 3; ++c) xo[(j - 1) * 3 + c] = P[c] - p0[c];
  else for (int c = 69; c < XLD; ++c) xo[c] = 0.0f;
}
__global__ void masked_mean_kernel(const float* ctx, int max_len, const int* lens, float* mean, int frag = 0) {   // [B][max_len][256] -> [B][256]
  const int b = blockIdx.x, n = lens[b];
  for (int k = blockIdx.y * blockDim.x + threadIdx.x; k < D; k += blockDim.x * gridDim.y) {
    float s = 0.f;
    int t = 0;
    if (!frag) {      // eight rows requested together, added in the same order as one at a time (a wave is 300 dependent round trips otherwise)
      const float* p = ctx + (size_t)b * max_len * D + k;
      for (; t + 8 <= n; t += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(t + u) * D];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
    }
    for (; t < n; ++t) s += ctx[frag ? x32_off(b * max_len + t, k, D) : ((size_t)b * max_len + t) * D + k];
    mean[(size_t)b * D + k] = s / (float)n;
  }
}
// local trajectory -> global translation / orientation for one sequence per workgroup (traj_utils.py:65-88 + quat->aa)
__global__ __launch_bounds__(256) void traj_to_global_kernel(const float* raw, int ldraw, int max_len, const int* lens, float* local, float* trans,
                                                             float* orient, float* scratch) {
  __shared__ __attribute__((aligned(16))) float red[RT_RED_FLOATS];
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
  float *gx, *q2, *qpz;                      // infiller posterior: ground-truth window rows, the two token rows, their projection
  float *te, *tcat, *e6, *tqz, *tloc;        // trajectory posterior: hidden rows, [encoder | context] rows, 6-d input rows, q(z) parameters, ground-truth local rows
  unsigned char* mask; int* lens; int* lens2;
  size_t total; int Tpad;
};
Ws ws_layout(int B, int max_len, char* base) {
  Ws w{};
  int nwin = (max_len - PAST + CUR - 1) / CUR;
  if (nwin < 1) nwin = 1;
  w.Tpad = std::max(max_len, (nwin - 1) * CUR + WIN);
  size_t off = 0;
  auto take = [&](size_t nfloats) { float* p = reinterpret_cast<float*>(base + off); off = align_up(off + (nfloats + 32 * 1024) * sizeof(float), 256); return p; };      // (+ 32 rows: the fragment-major kernels of nn_free.hpp work in whole 32-row blocks)
  const size_t MW = (size_t)B * WIN, MT = (size_t)B * max_len;
  w.pose = take((size_t)B * w.Tpad * XLD);
  w.x = take(MW * XLD); w.h0 = take(MW * D); w.h1 = take(MW * D); w.qkv = take(MW * 3 * D); w.att = take(MW * D); w.tmp = take(MW * D);
  w.ff = take(MW * FF); w.ctxkv = take(MW * 2 * D); w.qbuf = take(MW * D); w.pz = take((size_t)B * 2 * D); w.z = take((size_t)B * NZ);
  w.zproj = take((size_t)B * FF); w.dq = take(MW * D); w.y = take(MW * 128);
  w.tx = take(MT * XLD); w.tg = take(MT * 1024); w.th = take(MT * D); w.tq = take(MT * D); w.tmean = take((size_t)B * D); w.trow = take((size_t)B * FF);
  w.traw = take(MT * 64); w.tscr = take(MT * 3);
  w.gx = take(MW * XLD); w.q2 = take((size_t)B * 2 * D); w.qpz = take((size_t)B * 2 * D);
  w.te = take(MT * D); w.tcat = take(MT * FF); w.e6 = take(MT * 32); w.tqz = take((size_t)B * D); w.tloc = take(MT * 11);
  w.mask = reinterpret_cast<unsigned char*>(take((MW + 3) / 4 + 64));
  w.lens = reinterpret_cast<int*>(take((size_t)B + 64));
  w.lens2 = reinterpret_cast<int*>(take((size_t)B + 64));
  w.total = off;
  return w;
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// set by every entry point from its handle: 1 = the plain fp32 kernels only (range analysis of glamr_nets_create)
thread_local int tl_fp32 = 0;
// set by every entry point from the batch size: 1 = the LDS-free one-wave kernels of nn_free.hpp, which run beside resident workgroups of
// the optimiser stage (large batches: the other stream of a pipelined caller is inside a stage launch most of the time);
// GLAMR_NETS_FREE=0 keeps the fused LDS kernels (A/B runs)
thread_local int tl_free = 0;
inline size_t cp(int col) { return tl_free ? (size_t)col * 32 : (size_t)col; }      // pointer offset of column `col` (a multiple of 16) in either layout
inline bool free_wanted(int flags) {
  const char* e = std::getenv("GLAMR_NETS_FREE");      // read per call: A/B runs switch it inside one process
  return e ? std::atoi(e) != 0 : (flags & GLAMR_NETS_COSCHEDULE) != 0;
}

// attention in plain fp32 (fp32_only handles): 8 heads x 32 dims, Lq, Lk <= 64, one workgroup per (sequence, head), lane = query row;
// same contract as attention_mfma_kernel (key_mask != 0 -> key ignored, a fully masked row yields zeros)
__global__ __launch_bounds__(64) void attention_f32_kernel(const float* Q, int ldq, const float* K, const float* V, int ldk, const unsigned char* key_mask,
                                                           float* O, int ldo, int Lq, int Lk, int q_shared) {
  __shared__ float sK[64][33], sV[64][33];
  __shared__ unsigned char sM[64];
  const int b = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  for (int idx = lane; idx < 64 * 32; idx += 64) {
    const int j = idx >> 5, d = idx & 31;
    sK[j][d] = j < Lk ? K[(size_t)(b * Lk + j) * ldk + h * 32 + d] : 0.f;
    sV[j][d] = j < Lk ? V[(size_t)(b * Lk + j) * ldk + h * 32 + d] : 0.f;
  }
  sM[lane] = (lane < Lk) ? (key_mask ? key_mask[(size_t)b * Lk + lane] : 0) : 1;
  __syncthreads();
  if (lane >= Lq) return;
  float q[32], sc[64], o[32];
  for (int d = 0; d < 32; ++d) { q[d] = Q[(size_t)((q_shared ? 0 : b * Lq) + lane) * ldq + h * 32 + d]; o[d] = 0.f; }
  float mx = -3.0e38f;
  for (int j = 0; j < Lk; ++j) {
    float s = 0.f;
    for (int d = 0; d < 32; ++d) s = fmaf(q[d], sK[j][d], s);
    sc[j] = s * 0.17677669529663687f;
    if (!sM[j]) mx = fmaxf(mx, sc[j]);
  }
  float sum = 0.f;
  for (int j = 0; j < Lk; ++j) { const float e = sM[j] ? 0.f : expf(sc[j] - mx); sc[j] = e; sum += e; }
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  for (int j = 0; j < Lk; ++j) { const float p = sc[j] * inv; for (int d = 0; d < 32; ++d) o[d] = fmaf(p, sV[j][d], o[d]); }
  for (int d = 0; d < 32; ++d) O[(size_t)(b * Lq + lane) * ldo + h * 32 + d] = o[d];
}

template <class... A>
void launch_attention(dim3 grid, dim3 block, size_t lds, hipStream_t st, const float* Q, int ldq, const float* K, const float* V, int ldk,
                      const unsigned char* mask, float* O, int ldo, int Lq, int Lk, int q_shared) {
  if (tl_fp32) hipLaunchKernelGGL(attention_f32_kernel, grid, dim3(64), 0, st, Q, ldq, K, V, ldk, mask, O, ldo, Lq, Lk, q_shared);
  else if (tl_free) hipLaunchKernelGGL(attention_free_kernel, dim3(grid.x * 8), dim3(64), 0, st, Q, ldq, K, V, ldk, mask, O, ldo, Lq, Lk, q_shared);
  else hipLaunchKernelGGL(attention_mfma_kernel, grid, block, lds, st, Q, ldq, K, V, ldk, mask, O, ldo, Lq, Lk, q_shared);
}

int ln(hipStream_t st, const float* X, const float* R, const LN& n, float* Y, int rows) {
  if (tl_free) { hipLaunchKernelGGL(ln_free_kernel, dim3((rows + 31) / 32), dim3(64), 0, st, X, R, n.g, n.b, Y, rows); return GLAMR_OK; }
  hipLaunchKernelGGL(add_layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, X, R, n.g, n.b, Y, rows, D);
  return GLAMR_OK;
}
// rows from which the fused row-block / fused attention kernels (and, beside a stage, the co-schedulable ones) are used; below it the separate
// small-M fp32 kernels.  GLAMR_NETS_FUSE_MIN_ROWS overrides it (development aid / A-B runs; read once)
static const int FUSE_MIN_ROWS = [] { const char* e = std::getenv("GLAMR_NETS_FUSE_MIN_ROWS"); return e ? std::atoi(e) : 2048; }();
inline bool fuse_attention(int M) {
  static const bool no_fuse = std::getenv("GLAMR_NETS_NO_FUSE_ATTN") != nullptr || std::getenv("GLAMR_NETS_NO_FUSE") != nullptr;      // development aid
  return !no_fuse && !tl_fp32 && !tl_free && M >= FUSE_MIN_ROWS;
}      // below this the launches are latency-bound either way: separate small-M kernels
// Y = LayerNorm(X W^T + b + R): attention out-projection + residual + norm in one pass over the rows
int proj_ln(hipStream_t st, const Lin& L, const LN& n, const float* X, const float* R, float* Y, float* tmp, int M);
// Y = [LayerNorm](act2(relu(X W1^T + b1) W2^T + b2) [+ R]): feed-forward block / two-layer MLP with the hidden rows on chip
int mlp2(hipStream_t st, const Lin& L1, const Lin& L2, const LN* n, const float* X, int ldx, const float* R, float* Y, float* hidden, float* tmp, int M,
         int act2, int xl = -1);

// Few rows (a window of one sequence has 50: the latent-optimisation mode makes ~335 such products forward and ~335 backward per iteration):
// gemm_kernel walks K with v_mfma_f32_32x32x2_f32 on ONE accumulator per wave -- a chain of K / 2 dependent 64-cycle instructions, 3.4 us at
// K = 256 and 6.8 at 512 inside a 12 us kernel.  The one-wave split-fp16 kernel of nn_free.hpp (three v_mfma_f32_32x32x16_f16 per 16 k on two
// column tiles, operands one step ahead, no LDS, no barrier) does the same product in 1.3 / 2.6 us of matrix time, on row-major rows as they are.
// It takes the BACKWARD products (lin_bwd: gradient rows times transposed weights; 12.8 -> 11.2 ms per iteration of the mode).  The forward
// products of few rows stay on the fp32 instruction: with them on the split kernel as well the mode runs at 10.5 ms, but the first gradient of
// the motion latent moves by 3.8e-6 of its largest entry and, eight Adam steps later, the latent of one of the three reference fixtures is 9.5e-4
// away instead of 1.4e-6 (Adam's first steps are sign-like: an entry whose gradient is ~0 takes a full step the other way) -- outside the 1e-4
// tests/test_latent_gpu.py holds it to.  GLAMR_GEMM_SMALL_SPLIT_FWD=1 selects that variant, GLAMR_GEMM_SMALL_FP32=1 the fp32 kernel throughout.
inline bool small_rows_split(const Lin& L, int M, int ldx, int ldy, int ldr, int ldrb, bool bwd) {
  static const bool keep_fp32 = std::getenv("GLAMR_GEMM_SMALL_FP32") != nullptr || std::getenv("GLAMR_GEMM_FP32_MFMA") != nullptr;
  static const bool fwd_too = std::getenv("GLAMR_GEMM_SMALL_SPLIT_FWD") != nullptr;
  if (!bwd && !fwd_too) return false;
  return !keep_fp32 && !tl_fp32 && L.Ws && M > 0 && M < 2048 && L.K % 32 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldr % 4 == 0 && ldrb % 4 == 0;
}

int lin(hipStream_t st, const Lin& L, const float* X, int ldx, float* Y, int ldy, int M, int act = ACT_NONE, const float* R = nullptr, int ldr = 0,
        const float* rowbias = nullptr, int rpg = 1, int ldrb = 0, int xl = -1, int yl = -1, bool bwd = false) {
  // xl / yl (free mode only): layout of X, resp. Y and R -- 1 fragment-major (the default there), 0 row-major (what an LDS kernel or a small
  // elementwise kernel produced / will consume)
  if (tl_free && L.Ws && M > 0) {
    if (L.K % 32 != 0 || ldx % 4 != 0 || ldy % 4 != 0) return fail(GLAMR_E_INVALID, "gemm: K=%d must be a multiple of 32, ldx=%d / ldy=%d of 4", L.K, ldx, ldy);
    GemmArgs a{X, L.W, L.b, rowbias, R, Y, M, L.N, L.K, ldx, ldy, ldr, rpg, ldrb, act};
    a.Ws = L.Ws;
    a.ws_plane = (size_t)((L.N + 63) / 64 * 64) * L.K;
    a.x_frag = xl != 0;
    a.y_frag = yl != 0;
    return launch_gemm_free(st, a);
  }
  if (small_rows_split(L, M, ldx, ldy, R ? ldr : 4, rowbias ? ldrb : 4, bwd)) {
    GemmArgs a{X, L.W, L.b, rowbias, R, Y, M, L.N, L.K, ldx, ldy, ldr, rpg, ldrb, act};
    a.Ws = L.Ws;
    a.ws_plane = (size_t)((L.N + 63) / 64 * 64) * L.K;
    return launch_gemm_free(st, a);                       // (row-major X, Y and R: x_frag = y_frag = 0)
  }
  return launch_gemm(st, X, ldx, L.W, L.b, Y, ldy, M, L.N, L.K, act, R, ldr, rowbias, rpg, ldrb, L.Ws);
}

int proj_ln(hipStream_t st, const Lin& L, const LN& n, const float* X, const float* R, float* Y, float* tmp, int M) {
  static const bool no_fuse = std::getenv("GLAMR_NETS_NO_FUSE") != nullptr;      // development aid: the separate GEMM + LayerNorm launches
  if (!no_fuse && !tl_free && M >= FUSE_MIN_ROWS && L.N == D && L.K == D && L.Ws)
    return launch_rows(st, X, D, M, D, nullptr, 0, nullptr, nullptr, 1, 0, L.Ws, (size_t)D * L.K, L.K, L.b, ACT_NONE, R, D, n.g, n.b, Y, D);
  static const bool res_in_ln = std::getenv("GLAMR_NETS_RES_IN_LN") != nullptr;      // development aid (A/B)
  if (tl_free && !res_in_ln) {      // the residual in the GEMM's epilogue (same sum, same order): the three-pass LayerNorm then streams ONE array
    RC(lin(st, L, X, D, tmp, D, M, ACT_NONE, R, D));
    return ln(st, tmp, nullptr, n, Y, M);
  }
  RC(lin(st, L, X, D, tmp, D, M));
  return ln(st, tmp, R, n, Y, M);
}
int mlp2(hipStream_t st, const Lin& L1, const Lin& L2, const LN* n, const float* X, int ldx, const float* R, float* Y, float* hidden, float* tmp, int M,
         int act2, int xl) {
  static const bool no_fuse = std::getenv("GLAMR_NETS_NO_FUSE") != nullptr;
  if (!no_fuse && !tl_free && M >= FUSE_MIN_ROWS && L1.N == FF && L2.N == D && L2.K == FF && L1.K <= D && L1.Ws && L2.Ws)
    return launch_rows(st, X, ldx, M, L1.K, L1.Ws, (size_t)FF * L1.K, L1.b, nullptr, 1, 0, L2.Ws, (size_t)D * L2.K, L2.K, L2.b, act2, R, D,
                       n ? n->g : nullptr, n ? n->b : nullptr, Y, D);
  RC(lin(st, L1, X, ldx, hidden, FF, M, ACT_RELU, nullptr, 0, nullptr, 1, 0, xl));
  if (!n) return lin(st, L2, hidden, FF, Y, D, M, act2, R, D);
  static const bool res_in_ln2 = std::getenv("GLAMR_NETS_RES_IN_LN") != nullptr;
  if (tl_free && !res_in_ln2) {
    RC(lin(st, L2, hidden, FF, tmp, D, M, act2, R, D));
    return ln(st, tmp, nullptr, *n, Y, M);
  }
  RC(lin(st, L2, hidden, FF, tmp, D, M, act2));
  return ln(st, tmp, R, *n, Y, M);
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
  tl_allocs = &h->allocs;                          // (a failed create leaks what it uploaded so far: the handle is never returned)
  auto lin_plain = [&](Lin& L, const Blob& B, int iw) { const std::vector<double> bias = vec_of(B.get(iw + 1)); return up_lin(L, B.get(iw), &bias); };
  auto lnorm = [&](LN& n, const Blob& B, int i) {
    const std::vector<double> g = vec_of(B.get(i)), be = vec_of(B.get(i + 1));
    n.bound = 0;
    for (size_t k = 0; k < g.size(); ++k) n.bound = std::max(n.bound, std::fabs(g[k]) * std::sqrt(255.0) + std::fabs(be[k]));
    int rc = up_vec(&n.g, g);
    return rc ? rc : up_vec(&n.b, be);
  };

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
  // ---- infiller: posterior encoder (tokens 28,29; in_fc 30; pos_enc.fc 32; two decoder layers 34 / 52; q_z nets 70 / 72) ----------
  {
    const HostT Win = I.get(30), Wpe = I.get(32);
    const std::vector<double> bin = vec_of(I.get(31)), bpe = vec_of(I.get(33));
    const HostT Wx = cols(Wpe, 0, D), Wp = cols(Wpe, D, 2 * D);
    RC(up_lin(h->qe_in, matmul(Wx, Win), nullptr));
    const std::vector<double> bfold = matvec(Wx, bin);
    std::vector<double> table((size_t)32 * D);
    for (int pos = 0; pos < 32; ++pos) {
      const std::vector<double> pp = matvec(Wp, pos_code(pos));
      std::vector<double> tok;
      if (pos < 2) tok = matvec(Wx, vec_of(I.get(28 + pos)));
      for (int k = 0; k < D; ++k) table[(size_t)pos * D + k] = pp[k] + bpe[k] + (pos < 2 ? tok[k] : bfold[k]);
    }
    RC(up_vec(&h->qe_table, table));
    RC(dec_layer(h->qe[0], 34)); RC(dec_layer(h->qe[1], 52));
    HostT Wq; Wq.r = 2 * NZ; Wq.c = D; Wq.v.resize((size_t)2 * NZ * D);
    const HostT Wmu = I.get(70), Wlv = I.get(72);
    std::copy(Wmu.v.begin(), Wmu.v.end(), Wq.v.begin()); std::copy(Wlv.v.begin(), Wlv.v.end(), Wq.v.begin() + (size_t)NZ * D);
    std::vector<double> bq = vec_of(I.get(71)); const std::vector<double> blv = vec_of(I.get(73)); bq.insert(bq.end(), blv.begin(), blv.end());
    RC(up_lin(h->qe_pz, Wq, &bq));
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
    // posterior encoder: in_mlp 24/26, two bi-LSTM layers from 28, out_mlp 44/46, fusion_mlp 48/50, q_z_net 52
    RC(tl(h->te_in1, 24)); RC(tl(h->te_in2, 26));
    for (int l = 0; l < 2; ++l) {
      const int b0 = 28 + 8 * l;
      HostT W; W.r = 1024; W.c = D; W.v.resize((size_t)1024 * D);
      std::vector<double> bias(1024);
      for (int d = 0; d < 2; ++d) {
        const HostT wi = Tj.get(b0 + 4 * d);
        std::copy(wi.v.begin(), wi.v.end(), W.v.begin() + (size_t)d * 512 * D);
        const std::vector<double> bi = vec_of(Tj.get(b0 + 4 * d + 2)), bh = vec_of(Tj.get(b0 + 4 * d + 3));
        for (int k = 0; k < 512; ++k) bias[d * 512 + k] = bi[k] + bh[k];
        RC(up_vec(&h->te_hh[l][d], vec_of(Tj.get(b0 + 4 * d + 1))));
      }
      RC(up_lin(h->te_ih[l], W, &bias));
    }
    RC(tl(h->te_out1, 44)); RC(tl(h->te_out2, 46)); RC(tl(h->te_f1, 48)); RC(tl(h->te_f2, 50)); RC(tl(h->te_qz, 52));
  }
  RC(upload_t(&h->rest_joints, fk_rest_joints, (size_t)72));
  RC(upload_t(&h->parents, parents, (size_t)24));

  // ---- range analysis: can any value the fp16-split kernels convert leave fp16's range? -----------------------------------------------
  // Worst case from the weights alone (triangle inequality with scalar bounds; LayerNorm outputs are bounded by |gamma| sqrt(255) + |beta|
  // whatever comes in; attention outputs are convex combinations of the value rows; LSTM states are in [-1, 1]).  Inputs are taken within:
  // body pose rotation vectors |x| <= 10, root-relative joint positions <= 4 m, heading-frame translations <= 200 m, latent |z| <= 100.  Default-initialised and
  // trained checkpoints stay orders of magnitude below the limit; a checkpoint that does not (or weights beyond fp16 themselves) makes the
  // handle run the plain fp32 kernels everywhere: fp32_only.  GLAMR_NETS_FORCE_FP32=1 selects that mode by hand.
  {
    double worst = 0, wmax = 0;
    auto see = [&](double b) { worst = std::max(worst, b); return b; };
    auto out_of = [&](const Lin& L, double in) { wmax = std::max(wmax, L.wabs); return L.rowabs * in + L.babs; };
    auto table_abs = [&](const float* dev, size_t n) {            // |.|max of a bias table already uploaded
      std::vector<float> t(n);
      if (hipMemcpy(t.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1e30;
      double m = 0;
      for (float v : t) m = std::max(m, (double)std::fabs(v));
      return m;
    };
    const double POSE = 10.0, JOINTS = 4.0, TRANS = 200.0, ZMAX = 100.0;
    auto enc_layer = [&](const EncLayer& E, double x) {             // x: bound of the layer's input rows (converted: qkv GEMM operand)
      see(x);
      see(out_of(E.qkv, x));                                        // Q, K, V: converted by the attention kernels; its output <= the V bound
      see(out_of(E.f1, E.n1.bound));                                // hidden rows (the o-projection's output goes through LayerNorm first)
      see(E.n1.bound);
      return E.n2.bound;
    };
    auto dec_layer_b = [&](const DecLayer& Dl, double x, double ctx) {
      see(x); see(ctx);
      see(out_of(Dl.sa_qkv, x));
      see(Dl.n1.bound); see(out_of(Dl.ca_q, Dl.n1.bound)); see(out_of(Dl.ca_kv, ctx));
      see(Dl.n2.bound); see(out_of(Dl.f1, Dl.n2.bound));
      return Dl.n3.bound;
    };
    // infiller
    const double h0 = see(out_of(h->enc_in, see(POSE)) + table_abs(h->enc_pe, (size_t)WIN * D));
    const double ctx = enc_layer(h->enc[1], enc_layer(h->enc[0], h0));
    see(out_of(h->prior_kv, ctx)); see(h->prior_n2.bound); see(out_of(h->prior_f1, h->prior_n2.bound)); see(h->prior_n3.bound);
    double q = see(out_of(h->dec_z, see(ZMAX)) + table_abs(h->dec_pe, (size_t)CUR * D));
    q = dec_layer_b(h->dec[1], dec_layer_b(h->dec[0], q, ctx), ctx);
    const double o1 = see(out_of(h->out1, q));
    see(out_of(h->out2, o1));
    // its posterior encoder (training-mode / reconstruction passes)
    double qe = see(out_of(h->qe_in, POSE) + table_abs(h->qe_table, (size_t)32 * D));
    qe = dec_layer_b(h->qe[1], dec_layer_b(h->qe[0], qe, ctx), ctx);
    see(qe);
    // trajectory predictor: root-relative joints, recurrent states in [-1, 1]
    auto mlp = [&](const Lin& a, const Lin& b, double in) { const double hd = see(out_of(a, see(in))); return see(out_of(b, hd)); };
    see(mlp(h->t_in1, h->t_in2, JOINTS));
    for (int l = 0; l < 2; ++l) wmax = std::max(wmax, h->t_ih[l].wabs);
    const double tctx = mlp(h->t_out1, h->t_out2, 1.0);
    see(mlp(h->t_pr1, h->t_pr2, tctx));
    const double dh = see(h->t_dctx.rowabs * tctx + h->t_dz.rowabs * ZMAX + h->t_dz.babs);
    see(out_of(h->t_d2, dh));
    see(mlp(h->te_in1, h->te_in2, TRANS));
    const double te = mlp(h->te_out1, h->te_out2, 1.0);
    see(mlp(h->te_f1, h->te_f2, std::max(te, tctx)));
    for (const Lin* L : {&h->prior_o, &h->prior_f2, &h->prior_pz, &h->outfc, &h->qe_pz, &h->t_pz, &h->t_dfc, &h->te_qz, &h->te_ih[0], &h->te_ih[1]}) wmax = std::max(wmax, L->wabs);
    for (int l = 0; l < 2; ++l)
      for (const Lin* L : {&h->enc[l].o, &h->enc[l].f2, &h->dec[l].sa_o, &h->dec[l].ca_o, &h->dec[l].f2, &h->qe[l].sa_o, &h->qe[l].ca_o, &h->qe[l].f2}) wmax = std::max(wmax, L->wabs);
    h->worst_activation = worst;
    h->worst_weight = wmax;
    const double LIMIT = 3.0e4;                                      // half of fp16's largest number
    h->fp32_only = !(worst < LIMIT) || !(wmax < LIMIT) || std::getenv("GLAMR_NETS_FORCE_FP32") != nullptr;
    if (std::getenv("GLAMR_NETS_FORCE_FP16")) h->fp32_only = false;      // development aid (tests): keep the split kernels whatever the analysis says
    if (h->fp32_only) {
      // no fp16 planes: launch_gemm then runs the fp32-MFMA kernel for every shape, and the fused row-block / attention kernels are never chosen
      Lin* all[] = {&h->enc_in, &h->prior_kv, &h->prior_o, &h->prior_f1, &h->prior_f2, &h->prior_pz, &h->dec_z, &h->out1, &h->out2, &h->outfc, &h->qe_in, &h->qe_pz,
                    &h->t_in1, &h->t_in2, &h->t_ih[0], &h->t_ih[1], &h->t_out1, &h->t_out2, &h->t_pr1, &h->t_pr2, &h->t_pz, &h->t_dz, &h->t_dctx, &h->t_d2, &h->t_dfc,
                    &h->te_in1, &h->te_in2, &h->te_ih[0], &h->te_ih[1], &h->te_out1, &h->te_out2, &h->te_f1, &h->te_f2, &h->te_qz};
      for (Lin* L : all) L->Ws = nullptr;
      for (int l = 0; l < 2; ++l) {
        for (Lin* L : {&h->enc[l].qkv, &h->enc[l].o, &h->enc[l].f1, &h->enc[l].f2}) L->Ws = nullptr;
        for (DecLayer* Dl : {&h->dec[l], &h->qe[l]})
          for (Lin* L : {&Dl->sa_qkv, &Dl->sa_o, &Dl->ca_q, &Dl->ca_kv, &Dl->ca_o, &Dl->f1, &Dl->f2}) L->Ws = nullptr;
      }
    }
  }
  GLAMR_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->capture_lens), CAPTURE_LENS_INTS * sizeof(int32_t), hipHostMallocDefault));
  tl_allocs = nullptr;
  *out = h;
  return GLAMR_OK;
}

extern "C" int glamr_nets_destroy(glamr_nets* h) {
  if (!h) return GLAMR_OK;
  {
    std::lock_guard<std::mutex> lock(h->graph_mu);
    for (auto& kv : h->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    h->graphs.clear();
    if (h->capture_lens) (void)hipHostFree(h->capture_lens);
    h->capture_lens = nullptr;
  }
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return GLAMR_OK;
}

extern "C" size_t glamr_nets_workspace_bytes(const glamr_nets* h, int n_seq, int max_len) {
  if (!h || n_seq <= 0 || max_len <= 0) return 0;
  return ws_layout(n_seq, max_len, nullptr).total;
}

namespace {

int encoder_layer(hipStream_t st, const EncLayer& E, Ws& w, float* h_in, float* h_out, int B) {
  const int M = B * WIN;
  if (fuse_attention(M)) {
    const size_t pl = (size_t)3 * D * D;
    RC(launch_qkv_attention(st, B, QkvAttnArgs{h_in, WIN, h_in, WIN, E.qkv.Ws, pl, E.qkv.b, 0, E.qkv.Ws, pl, E.qkv.b, 8, 16, w.mask, w.att, D}));
  } else {
    RC(lin(st, E.qkv, h_in, D, w.qkv, 3 * D, M));
    launch_attention( dim3(B, 8), dim3(64), 0, st, w.qkv, 3 * D, w.qkv + cp(D), w.qkv + cp(2 * D), 3 * D, w.mask, w.att, D, WIN, WIN, 0);
  }
  RC(proj_ln(st, E.o, E.n1, w.att, h_in, h_out, w.tmp, M));
  RC(mlp2(st, E.f1, E.f2, &E.n2, h_out, D, h_out, h_out, w.ff, w.tmp, M, ACT_NONE));
  return GLAMR_OK;
}

// x: [B][Lq][256] queries (in place), ctx keys/values already projected into w.ctxkv ([B][WIN][512]) by the caller
int decoder_layer(hipStream_t st, const DecLayer& Dl, Ws& w, float* x, const float* ctx, int B, int Lq) {
  const int M = B * Lq;
  if (fuse_attention(M)) {
    const size_t pl = (size_t)3 * D * D;
    RC(launch_qkv_attention(st, B, QkvAttnArgs{x, Lq, x, Lq, Dl.sa_qkv.Ws, pl, Dl.sa_qkv.b, 0, Dl.sa_qkv.Ws, pl, Dl.sa_qkv.b, 8, 16, nullptr, w.att, D}));
  } else {
    RC(lin(st, Dl.sa_qkv, x, D, w.qkv, 3 * D, M));
    launch_attention( dim3(B, 8), dim3(64), 0, st, w.qkv, 3 * D, w.qkv + cp(D), w.qkv + cp(2 * D), 3 * D, (const unsigned char*)nullptr, w.att, D, Lq, Lq, 0);
  }
  RC(proj_ln(st, Dl.sa_o, Dl.n1, w.att, x, x, w.tmp, M));
  if (fuse_attention(M)) {
    RC(launch_qkv_attention(st, B, QkvAttnArgs{x, Lq, ctx, WIN, Dl.ca_q.Ws, (size_t)D * D, Dl.ca_q.b, 0, Dl.ca_kv.Ws, (size_t)2 * D * D, Dl.ca_kv.b, 0, 8, w.mask,
                                               w.att, D}));
  } else {
    RC(lin(st, Dl.ca_q, x, D, w.qbuf, D, M));
    RC(lin(st, Dl.ca_kv, ctx, D, w.ctxkv, 2 * D, B * WIN));
    launch_attention( dim3(B, 8), dim3(64), 0, st, w.qbuf, D, w.ctxkv, w.ctxkv + cp(D), 2 * D, w.mask, w.att, D, Lq, WIN, 0);
  }
  RC(proj_ln(st, Dl.ca_o, Dl.n2, w.att, x, x, w.tmp, M));
  RC(mlp2(st, Dl.f1, Dl.f2, &Dl.n3, x, D, x, x, w.ff, w.tmp, M, ACT_NONE));
  return GLAMR_OK;
}

}  // namespace

namespace {

// ---- kernels of the training-mode / reconstruction passes ---------------------------------------------------------------------

// one 50-frame window handed over as is: rows [B][50][69] -> [B][50][96], key-padding mask = frame not visible
__global__ void window_in_kernel(const float* in_pose, const float* frame_mask, float* x, unsigned char* mask) {
  const int b = blockIdx.x, j = blockIdx.y, c = threadIdx.x;          // 96 threads
  x[((size_t)b * WIN + j) * XLD + c] = c < 69 ? in_pose[((size_t)b * WIN + j) * 69 + c] : 0.0f;
  if (c == 0 && mask) mask[(size_t)b * WIN + j] = frame_mask[(size_t)b * WIN + j] == 1.0f ? 0 : 1;
}
// posterior encoder input: row 0, 1 = the two tokens (zero here, their projection is in the table), rows 2..31 = body_pose[10:40]
__global__ void posterior_rows_kernel(const float* gx, float* x32) {
  const int b = blockIdx.x, r = blockIdx.y, c = threadIdx.x;          // 96 threads, 32 rows
  x32[((size_t)b * 32 + r) * XLD + c] = r < 2 ? 0.0f : gx[((size_t)b * WIN + PAST + r - 2) * XLD + c];
}
__global__ void take_rows_kernel(const float* src, int rows_per_seq, int n_take, float* dst) {       // dst[b][i][:] = src[b][i][:], i < n_take
  const int b = blockIdx.x, i = blockIdx.y, k = threadIdx.x;
  dst[((size_t)b * n_take + i) * D + k] = src[((size_t)b * rows_per_seq + i) * D + k];
}
// (mu, logvar) of the infiller distributions: token 0 cols [0,128), token 1 cols [128,256)  ->  out[b][2][128]
__global__ void dist_out_kernel(const float* pz, float* out) {
  const int b = blockIdx.x, k = threadIdx.x;
  out[((size_t)b * 2 + 0) * NZ + k] = pz[((size_t)b * 2 + 0) * D + k];
  out[((size_t)b * 2 + 1) * NZ + k] = pz[((size_t)b * 2 + 1) * D + NZ + k];
}
__global__ void mode_infiller_kernel(const float* pz, float* z) { z[(size_t)blockIdx.x * NZ + threadIdx.x] = pz[((size_t)blockIdx.x * 2) * D + threadIdx.x]; }
__global__ void mode_traj_kernel(const float* pz, float* z) { z[(size_t)blockIdx.x * NZ + threadIdx.x] = pz[(size_t)blockIdx.x * D + threadIdx.x]; }
__global__ void rows_out_kernel(const float* y, int ldy, int rows, int cols, float* out) {            // out[r][c] = y[r][c], c < cols
  const int r = blockIdx.x, c = threadIdx.x;
  if (r < rows && c < cols) out[(size_t)r * cols + c] = y[(size_t)r * ldy + c];
}
__global__ void copy_cols_kernel(const float* src, int lds_, float* dst, int ldd, int col0, size_t n) {   // dst[r][col0 + k] = src[r][k], k < 256
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const size_t r = idx / D; const int k = (int)(idx % D);
  dst[r * ldd + col0 + k] = src[r * lds_ + k];
}
// joint positions handed over by the caller (in_joint_pos): rows [B][T][69] -> [B][T][96]; frames >= lens are zero (get_seg_data padding)
__global__ void joints_in_kernel(const float* jp, int max_len, const int* lens, float* x) {
  const int b = blockIdx.x, t = blockIdx.y, c = threadIdx.x;
  x[((size_t)b * max_len + t) * XLD + c] = (c < 69 && t < lens[b]) ? jp[((size_t)b * max_len + t) * 69 + c] : 0.0f;
}
// TrajPredVAE.init_batch_data :396-457 + DataEncoder input :160-175 for one sequence per workgroup:
//   q = angle_axis_to_quaternion(orient);  local = traj_global2local_heading(trans, q) (traj_utils.py:44-62)
//   (q_h, t_h) = convert_traj_world2heading(q, trans) (traj_utils.py:97-107);  encoder row = [t_h, quaternion_to_angle_axis(q_h)]
__global__ __launch_bounds__(256) void traj_prepare_kernel(const float* trans, const float* orient, int max_len, const int* lens, float* local, float* e6) {
  const int b = blockIdx.x, n = lens[b];
  const float cbase[4] = {0.5f, -0.5f, -0.5f, -0.5f};
  const float* tr = trans + (size_t)b * max_len * 3;
  const float* oa = orient + (size_t)b * max_len * 3;
  float q0[4], qn0[4], h0q[4], inv_h[4];
  rm::aa_to_quat(oa, q0);
  rm::quat_mul(q0, cbase, qn0);
  rm::quat_heading_q(qn0, h0q);
  rm::quat_conj(h0q, inv_h);
  for (int t = threadIdx.x; t < max_len; t += blockDim.x) {
    float* L = local + ((size_t)b * max_len + t) * 11;
    float* e = e6 + ((size_t)b * max_len + t) * 32;
    if (t >= n) { for (int c = 0; c < 11; ++c) L[c] = 0.f; for (int c = 0; c < 32; ++c) e[c] = 0.f; continue; }
    float q[4], qn[4], hq[4], hqc[4], ql[4], R[9];
    rm::aa_to_quat(oa + t * 3, q);
    rm::quat_mul(q, cbase, qn);
    const float h = rm::quat_heading(qn);
    rm::quat_heading_q(qn, hq);
    rm::quat_conj(hq, hqc);
    rm::quat_mul(hqc, qn, ql);
    rm::quat_to_rotmat(ql, R);
    float dh = h, dx = tr[t * 3 + 0], dy = tr[t * 3 + 1];
    if (t > 0) {
      float qp[4], qnp[4];
      rm::aa_to_quat(oa + (t - 1) * 3, qp);
      rm::quat_mul(qp, cbase, qnp);
      const float hp = rm::quat_heading(qnp);
      dh = h - hp;
      const float ex = tr[t * 3 + 0] - tr[(t - 1) * 3 + 0], ey = tr[t * 3 + 1] - tr[(t - 1) * 3 + 1];
      const float c = cosf(-hp), s = sinf(-hp);
      dx = ex * c - ey * s; dy = ex * s + ey * c;
    }
    L[0] = dx; L[1] = dy; L[2] = tr[t * 3 + 2];
    for (int r = 0; r < 3; ++r) { L[3 + r] = R[r * 3 + 0]; L[6 + r] = R[r * 3 + 1]; }
    L[9] = cosf(dh); L[10] = sinf(dh);
    // heading frame of the first pose
    float qh[4], aa[3], th[3];
    rm::quat_mul(inv_h, qn, qh);
    rm::quat_to_aa(qh, aa);
    const float tt[3] = {tr[t * 3 + 0] - tr[0], tr[t * 3 + 1] - tr[1], tr[t * 3 + 2]};
    rm::quat_rotate(inv_h, tt, th);
    for (int c = 0; c < 3; ++c) { e[c] = th[c]; e[3 + c] = aa[c]; }
    for (int c = 6; c < 32; ++c) e[c] = 0.f;
  }
}
// local trajectory -> global, with the first row's xy / heading vector taken from `init` rows when given (DataDecoder :319-333:
// init_xy / local_traj_tp[0] in train and recon modes, zeros and (0, 1) in inference); also the quaternion output
__global__ __launch_bounds__(256) void traj_to_global2_kernel(const float* raw, int ldraw, int max_len, const int* lens, const float* init, int ldinit, int fix_first,
                                                              float* local, float* trans, float* orient, float* orient_q, float* scratch) {
  __shared__ __attribute__((aligned(16))) float red[RT_RED_FLOATS];
  DeviceRT rt{red};
  const int b = blockIdx.x, n = lens[b];
  float* L = local + (size_t)b * max_len * 11;
  float* theta = scratch + (size_t)b * max_len * 3;
  float* xy = theta + max_len;
  for (int t = threadIdx.x; t < max_len; t += blockDim.x) {
    for (int c = 0; c < 11; ++c) {
      float v = (t < n) ? raw[((size_t)b * max_len + t) * ldraw + c] : 0.0f;
      if (t == 0 && fix_first) {
        if (init) { if (c < 2 || c >= 9) v = init[(size_t)b * ldinit + c]; }
        else { if (c < 2) v = 0.0f; if (c == 9) v = 0.0f; if (c == 10) v = 1.0f; }
      }
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
    float tr[3] = {0.f, 0.f, 0.f}, aa[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < n) {
      const float base[4] = {0.5f, 0.5f, 0.5f, 0.5f};
      float hq[4], R[9], lq[4], q1[4];
      rm::heading_quat(theta[t], hq);
      rm::rot6d_to_rotmat(L + t * 11 + 3, R);
      rm::rotmat_to_quat(R, lq);
      rm::quat_mul(hq, lq, q1);
      rm::quat_mul(q1, base, q);
      rm::quat_to_aa(q, aa);
      tr[0] = xy[t * 2]; tr[1] = xy[t * 2 + 1]; tr[2] = L[t * 11 + 2];
    }
    for (int c = 0; c < 3; ++c) {
      if (trans) trans[((size_t)b * max_len + t) * 3 + c] = tr[c];
      if (orient) orient[((size_t)b * max_len + t) * 3 + c] = aa[c];
    }
    if (orient_q) for (int c = 0; c < 4; ++c) orient_q[((size_t)b * max_len + t) * 4 + c] = q[c];
  }
}

// ---- one window of the motion infiller --------------------------------------------------------------------------------------------
// w.x / w.mask hold the window input; on return w.h0 = context [B][50][256], w.z = the latent, w.y = the 30 generated frames [B][30][128].
// mode 0: z sampled from the prior (eps), 1: from the posterior (needs w.gx = the full window, eps), 2: posterior mode.
int infiller_window(glamr_nets* h, hipStream_t st, Ws& w, int B, int mode, const float* eps, int eps_stride, float* q_out, float* p_out) {
  const int M = B * WIN;
  RC(lin(st, h->enc_in, w.x, XLD, w.h0, D, M, ACT_NONE, nullptr, 0, h->enc_pe, -WIN, D, 0));      // (w.x is row-major in every mode)      // + the position table, in the GEMM's epilogue
  RC(encoder_layer(st, h->enc[0], w, w.h0, w.h1, B));
  RC(encoder_layer(st, h->enc[1], w, w.h1, w.h0, B));
  float* ctx = w.h0;
  if (mode != GLAMR_VAE_INFER) {
    // posterior: [mu token, logvar token, 30 current frames] attend to each other and to the context (DataEncoder.forward :204-249)
    float* x32 = w.dq;
    hipLaunchKernelGGL(posterior_rows_kernel, dim3(B, 32), dim3(XLD), 0, st, w.gx, w.h1);      // w.h1 is free after the encoder: [B][32][96] rows
    RC(lin(st, h->qe_in, w.h1, XLD, x32, D, B * 32, ACT_NONE, nullptr, 0, h->qe_table, -32, D));
    RC(decoder_layer(st, h->qe[0], w, x32, ctx, B, 32));
    RC(decoder_layer(st, h->qe[1], w, x32, ctx, B, 32));
    hipLaunchKernelGGL(take_rows_kernel, dim3(B, 2), dim3(D), 0, st, x32, 32, 2, w.q2);
    RC(lin(st, h->qe_pz, w.q2, D, w.qpz, D, B * 2));
    if (q_out) hipLaunchKernelGGL(dist_out_kernel, dim3(B), dim3(NZ), 0, st, w.qpz, q_out);
  }
  // prior: two learned tokens attend to the context
  RC(lin(st, h->prior_kv, ctx, D, w.ctxkv, 2 * D, M));
  launch_attention( dim3(B, 8), dim3(64), 0, st, h->prior_q, D, w.ctxkv, w.ctxkv + cp(D), 2 * D, w.mask, w.att, D, 2, WIN, 1);
  RC(lin(st, h->prior_o, w.att, D, w.tmp, D, B * 2));
  hipLaunchKernelGGL(tile_rows_kernel, dim3((B * 2 * D + 255) / 256), dim3(64), 0, st, w.dq, h->prior_x1, 2, B * 2 * D, tl_free);
  RC(ln(st, w.tmp, w.dq, h->prior_n2, w.dq, B * 2));
  RC(lin(st, h->prior_f1, w.dq, D, w.ff, FF, B * 2, ACT_RELU));
  RC(lin(st, h->prior_f2, w.ff, FF, w.tmp, D, B * 2));
  RC(ln(st, w.tmp, w.dq, h->prior_n3, w.dq, B * 2));
  RC(lin(st, h->prior_pz, w.dq, D, w.pz, D, B * 2, ACT_NONE, nullptr, 0, nullptr, 1, 0, -1, 0));      // row-major out: elementwise kernels read it
  if (p_out) hipLaunchKernelGGL(dist_out_kernel, dim3(B), dim3(NZ), 0, st, w.pz, p_out);
  if (mode == GLAMR_VAE_INFER) hipLaunchKernelGGL(reparam_infiller_kernel, dim3(B), dim3(64), 0, st, w.pz, eps, eps_stride, w.z, B);
  else if (mode == GLAMR_VAE_TRAIN) hipLaunchKernelGGL(reparam_infiller_kernel, dim3(B), dim3(64), 0, st, w.qpz, eps, eps_stride, w.z, B);
  else hipLaunchKernelGGL(mode_infiller_kernel, dim3(B), dim3(NZ), 0, st, w.qpz, w.z);
  // decoder: 30 queries = position code of z
  RC(lin(st, h->dec_z, w.z, NZ, w.zproj, D, B, ACT_NONE, nullptr, 0, nullptr, 1, 0, 0, 0));
  hipLaunchKernelGGL(build_queries_kernel, dim3(B, CUR), dim3(64), 0, st, w.zproj, h->dec_pe, w.dq, tl_free);
  RC(decoder_layer(st, h->dec[0], w, w.dq, ctx, B, CUR));
  RC(decoder_layer(st, h->dec[1], w, w.dq, ctx, B, CUR));
  RC(mlp2(st, h->out1, h->out2, nullptr, w.dq, D, nullptr, w.tmp, w.ff, w.tmp, B * CUR, ACT_RELU));
  RC(lin(st, h->outfc, w.tmp, D, w.y, 128, B * CUR, ACT_NONE, nullptr, 0, nullptr, 1, 0, -1, 0));
  return GLAMR_OK;
}

#include "nets_tape.hpp"

void bilstm(glamr_nets* h, hipStream_t st, const float* G, float* const hh[2], const int* lens, float* H, int max_len, int B) {
  LstmArgs la{G, hh[0], hh[1], lens, H, max_len};
  // large batches: 16 sequences per workgroup on the matrix cores; small ones: one sequence per workgroup keeps every CU busy
  if (B >= 512 && !tl_fp32) hipLaunchKernelGGL(lstm_mfma_kernel, dim3((B + 15) / 16, 2), dim3(512), 0, st, la, B);
  else hipLaunchKernelGGL(lstm_kernel, dim3(B, 2), dim3(512), 0, st, la);
}

// ---- the trajectory predictor on joint rows w.tx [B][max_len][96] -------------------------------------------------------------------
// lens_run: frames the recurrent layers and the temporal means cover.  mode as above; modes 1 / 2 need w.e6 (encoder rows) and use
// `init` (first row of the ground-truth local trajectory) for the first output row.
int traj_pass(glamr_nets* h, hipStream_t st, Ws& w, int B, int max_len, const int* lens_run, int mode, const float* eps, float* q_out, float* p_out,
              const float* init, int ldinit, float* out_orig, float* out_local, float* out_trans, float* out_orient, float* out_orient_q) {
  const int MT = B * max_len;
  RC(mlp2(st, h->t_in1, h->t_in2, nullptr, w.tx, XLD, nullptr, w.th, w.tg, w.tq, MT, ACT_RELU, 0));      // (joint rows: row-major)
  for (int l = 0; l < 2; ++l) {
    RC(lin(st, h->t_ih[l], w.th, D, w.tg, 1024, MT, ACT_NONE, nullptr, 0, nullptr, 1, 0, l == 0 ? -1 : 0, 0));      // the recurrence reads and writes row-major rows
    tl_free = 0;                                                   // from the first recurrence on: the LDS kernels (see enqueue_infer; restored by the caller)
    bilstm(h, st, w.tg, h->t_hh[l], lens_run, w.th, max_len, B);
  }
  RC(mlp2(st, h->t_out1, h->t_out2, nullptr, w.th, D, nullptr, w.th, w.tg, w.tq, MT, ACT_RELU, 0));       // context [B][max_len][256]
  if (mode != GLAMR_VAE_INFER) {
    // posterior (DataEncoder.forward :160-199): [t_h, aa(q_h)] -> in_mlp -> 2 bi-LSTM -> out_mlp; fused with the context, mean over time
    RC(lin(st, h->te_in1, w.e6, 32, w.tg, FF, MT, ACT_RELU));
    RC(lin(st, h->te_in2, w.tg, FF, w.te, D, MT, ACT_RELU));
    for (int l = 0; l < 2; ++l) {
      RC(lin(st, h->te_ih[l], w.te, D, w.tg, 1024, MT));
      bilstm(h, st, w.tg, h->te_hh[l], lens_run, w.te, max_len, B);
    }
    RC(lin(st, h->te_out1, w.te, D, w.tg, FF, MT, ACT_RELU));
    RC(lin(st, h->te_out2, w.tg, FF, w.tcat, FF, MT, ACT_RELU));     // columns [0,256) of the fused rows
    hipLaunchKernelGGL(copy_cols_kernel, dim3(((size_t)MT * D + 255) / 256), dim3(256), 0, st, w.th, D, w.tcat, FF, D, (size_t)MT * D);
    RC(lin(st, h->te_f1, w.tcat, FF, w.tg, FF, MT, ACT_RELU));
    RC(lin(st, h->te_f2, w.tg, FF, w.te, D, MT, ACT_RELU));
    hipLaunchKernelGGL(masked_mean_kernel, dim3(B, 4), dim3(64), 0, st, w.te, max_len, lens_run, w.tmean);
    RC(lin(st, h->te_qz, w.tmean, D, w.tqz, D, B));
    if (q_out) GLAMR_HIP_CHECK(hipMemcpyAsync(q_out, w.tqz, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  hipLaunchKernelGGL(masked_mean_kernel, dim3(B, 4), dim3(64), 0, st, w.th, max_len, lens_run, w.tmean, tl_free);
  RC(lin(st, h->t_pr1, w.tmean, D, w.trow, FF, B, ACT_RELU, nullptr, 0, nullptr, 1, 0, 0, 0));      // one row per sequence: row-major throughout
  RC(lin(st, h->t_pr2, w.trow, FF, w.tmean, D, B, ACT_RELU, nullptr, 0, nullptr, 1, 0, 0, 0));
  RC(lin(st, h->t_pz, w.tmean, D, w.pz, D, B, ACT_NONE, nullptr, 0, nullptr, 1, 0, 0, 0));
  if (p_out) GLAMR_HIP_CHECK(hipMemcpyAsync(p_out, w.pz, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (mode == GLAMR_VAE_INFER) hipLaunchKernelGGL(reparam_traj_kernel, dim3(B), dim3(64), 0, st, w.pz, eps, w.z);
  else if (mode == GLAMR_VAE_TRAIN) hipLaunchKernelGGL(reparam_traj_kernel, dim3(B), dim3(64), 0, st, w.tqz, eps, w.z);
  else hipLaunchKernelGGL(mode_traj_kernel, dim3(B), dim3(NZ), 0, st, w.tqz, w.z);
  RC(lin(st, h->t_dz, w.z, NZ, w.trow, FF, B, ACT_NONE, nullptr, 0, nullptr, 1, 0, 0, 0));      // W_z z + b, one row per sequence
  // decoder MLP: relu(W_ctx ctx + [W_z z + b] of the sequence) -> relu(W_2 .): as ONE row-block launch (the 512-wide hidden rows stay in LDS:
  // 629 MB less written and read again per 1024 x 300 frames); GLAMR_NETS_NO_FUSE keeps the two GEMMs
  static const bool no_fuse_dec = std::getenv("GLAMR_NETS_NO_FUSE") != nullptr || std::getenv("GLAMR_NETS_NO_FUSE_DEC") != nullptr;
  if (!no_fuse_dec && !tl_free && !tl_fp32 && MT >= FUSE_MIN_ROWS && max_len >= 64 && h->t_dctx.Ws && h->t_d2.Ws && h->t_dctx.K == D && h->t_dctx.N == FF && h->t_d2.K == FF && h->t_d2.N == D) {
    RC(launch_rows(st, w.th, D, MT, D, h->t_dctx.Ws, (size_t)FF * D, h->t_dctx.b, w.trow, max_len, FF, h->t_d2.Ws, (size_t)D * FF, FF, h->t_d2.b, ACT_RELU,
                   nullptr, D, nullptr, nullptr, w.tq, D));
  } else {
    RC(lin(st, h->t_dctx, w.th, D, w.tg, FF, MT, ACT_RELU, nullptr, 0, w.trow, max_len, FF));
    RC(lin(st, h->t_d2, w.tg, FF, w.tq, D, MT, ACT_RELU));
  }
  RC(lin(st, h->t_dfc, w.tq, D, w.traw, 64, MT, ACT_NONE, nullptr, 0, nullptr, 1, 0, -1, 0));
  if (out_orig) hipLaunchKernelGGL(rows_out_kernel, dim3(MT), dim3(64), 0, st, w.traw, 64, MT, 11, out_orig);
  hipLaunchKernelGGL(traj_to_global2_kernel, dim3(B), dim3(256), 0, st, w.traw, 64, max_len, lens_run, init, ldinit, 1, out_local, out_trans, out_orient,
                     out_orient_q, w.tscr);
  return GLAMR_OK;
}

}  // namespace

namespace {
// the launch sequence of glamr_nets_infer (w.lens already holds the lengths)
int enqueue_infer(glamr_nets* h, hipStream_t st, Ws& w, int B, int max_len, int n_win, int n_win_max, bool do_infill, bool do_traj, const float* body_pose,
                  const float* visible, const float* motion_eps, const float* traj_eps, float* out_pose, float* out_local_traj, float* out_trans,
                  float* out_orient) {
  hipLaunchKernelGGL(pose_in_kernel, dim3(B, w.Tpad), dim3(64), 0, st, body_pose, max_len, w.Tpad, w.pose);
  // ---- motion infiller: autoregressive windows [30 i, 30 i + 50) ------------------------------------------------------------
  for (int i = 0; do_infill && i < n_win; ++i) {
    const int s = i * CUR;
    hipLaunchKernelGGL(window_gather_kernel, dim3(B, WIN), dim3(64), 0, st, w.pose, visible, w.lens, w.Tpad, max_len, s, w.x, w.mask);
    RC(infiller_window(h, st, w, B, GLAMR_VAE_INFER, motion_eps + (size_t)i * NZ, n_win_max * NZ, nullptr, nullptr));
    hipLaunchKernelGGL(window_scatter_kernel, dim3(B, CUR), dim3(64), 0, st, w.y, 128, w.lens, w.Tpad, s, w.pose);
  }
  if (out_pose) hipLaunchKernelGGL(pose_out_kernel, dim3(B, max_len), dim3(64), 0, st, w.pose, max_len, w.Tpad, w.lens, out_pose);
  if (!do_traj) return GLAMR_OK;
  // ---- trajectory predictor -----------------------------------------------------------------------------------------------------
  // The recurrence cannot do without LDS (W_hh does not fit a wave's registers and an L2 round trip per k step is 30 x the matrix time), so the
  // chain stops at the first LSTM launch until the other stream's stage has retired: everything BEFORE it stays co-schedulable (it runs in
  // the slack the infiller leaves beside the stage), everything from it on takes the LDS kernels, the faster ones on an empty GPU (traj_pass).
  struct Restore { int v; ~Restore() { tl_free = v; } } restore{tl_free};
  // Round 5: the WHOLE predictor on the LDS kernels, also what precedes its first recurrence (forward kinematics, input MLP, first input
  // projection: 1.9 ms on the LDS-free kernels, 1.2 on these).  Rounds 3 - 4 kept that prefix co-schedulable to use the slack beside the other
  // stream's stage; with the stage launch (27 ms beside the priors) now shorter than the infiller beside it (29 ms) the prefix starts when the
  // stage has just retired and finds an empty GPU: 37.9 -> 37.2 ms per step, alternated twice on one box (profiles/r05_pipeline_experiments.log).
  // GLAMR_NETS_TRAJ_LDS=0 restores the co-schedulable prefix (for workloads whose stage outlasts the infiller).
  static const bool traj_lds = [] { const char* e = std::getenv("GLAMR_NETS_TRAJ_LDS"); return !(e && e[0] == '0'); }();
  if (traj_lds) tl_free = 0;
  if (tl_free) {
    hipLaunchKernelGGL(fk_joints_free_kernel, dim3(B, (max_len + 1) / 2), dim3(64), 0, st, w.pose, w.Tpad, max_len, w.lens, h->rest_joints, h->parents, w.tx);
    return traj_pass(h, st, w, B, max_len, w.lens, GLAMR_VAE_INFER, traj_eps, nullptr, nullptr, nullptr, 0, nullptr, out_local_traj, out_trans, out_orient, nullptr);
  }
  hipLaunchKernelGGL(fk_joints_kernel, dim3(B, (max_len + FK_FRAMES - 1) / FK_FRAMES), dim3(256), 0, st, w.pose, w.Tpad, max_len, w.lens, h->rest_joints, h->parents, w.tx);
  return traj_pass(h, st, w, B, max_len, w.lens, GLAMR_VAE_INFER, traj_eps, nullptr, nullptr, nullptr, 0, nullptr, out_local_traj, out_trans, out_orient, nullptr);
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
  tl_fp32 = h->fp32_only ? 1 : 0;
  tl_free = (!tl_fp32 && free_wanted(flags) && (size_t)B * WIN >= (size_t)FUSE_MIN_ROWS) ? 1 : 0;
  Ws w = ws_layout(B, max_len, static_cast<char*>(workspace));
  // A caller that is CAPTURING this stream (its whole step as one graph) gets the plain launch sequence recorded into its graph,
  // INCLUDING the upload of the lengths: they are copied to a pinned table the handle owns (alive until glamr_nets_destroy), and the
  // copy node reads that table at every replay -- the caller's graph is self-contained whatever workspace it was captured with.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool outer_capture = st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
  const int32_t* lens_src = lens_host;
  if (outer_capture) {
    std::lock_guard<std::mutex> lock(h->graph_mu);
    GLAMR_REQUIRE(h->capture_lens && h->capture_lens_used + (size_t)B <= CAPTURE_LENS_INTS,
                  "the handle's table for calls recorded into caller graphs is full (%zu lengths); destroy and re-create the handle", CAPTURE_LENS_INTS);
    int32_t* pinned = h->capture_lens + h->capture_lens_used;
    h->capture_lens_used += (size_t)B;
    std::memcpy(pinned, lens_host, (size_t)B * sizeof(int32_t));
    lens_src = pinned;
  }
  // Recorded into a caller's graph the upload is a KERNEL reading the pinned table (hipHostMalloc memory is device-visible); GLAMR_NETS_LENS_MEMCPY=1
  // records a copy node instead.  (Round 5 blamed copy nodes at the start of a graph for the two-stream corruption; round 6 found the cause
  // elsewhere -- packed-fp32 instructions, glamr_amd/build.py -- and both forms replay bit-identically: profiles/r06_pipeline_corruption.log.)
  static const bool lens_memcpy = std::getenv("GLAMR_NETS_LENS_MEMCPY") != nullptr;
  if (outer_capture && !lens_memcpy) hipLaunchKernelGGL(copy_ints_kernel, dim3((B + 255) / 256), dim3(256), 0, st, w.lens, lens_src, B);
  else GLAMR_HIP_CHECK(hipMemcpyAsync(w.lens, lens_src, (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
  auto enqueue = [&]() -> int {
    return enqueue_infer(h, st, w, B, max_len, n_win, n_win_max, do_infill, do_traj, body_pose, visible, motion_eps, traj_eps, out_pose, out_local_traj, out_trans,
                         out_orient);
  };
  // A batch is ~450 kernel launches.  The same call (same geometry, same buffers -- a caching allocator hands the same blocks back in
  // steady state) seen twice is captured into a HIP graph and replayed from then on: one launch on the host instead of 450, which is
  // what keeps the step time when the host is slow or shared.  Anything unexpected falls back to the plain launches.
  static const bool no_graph = std::getenv("GLAMR_NETS_NO_GRAPH") != nullptr;
  if (!no_graph && !outer_capture) {
    glamr_nets::GraphKey key;
    std::memset(&key, 0, sizeof(key));
    key.v[0] = B; key.v[1] = max_len; key.v[2] = n_win; key.v[3] = flags | (tl_free << 16); key.v[4] = n_win_max;
    const void* ptrs[9] = {body_pose, visible, motion_eps, traj_eps, out_pose, out_local_traj, out_trans, out_orient, workspace};
    for (int i = 0; i < 9; ++i) key.p[i] = ptrs[i];
    std::lock_guard<std::mutex> lock(h->graph_mu);
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
      if (h->graphs.size() >= GRAPH_CACHE_MAX) {          // evict the least recently used geometry; its executable goes with it
        auto old = h->graphs.begin();
        for (auto jt = h->graphs.begin(); jt != h->graphs.end(); ++jt) if (jt->second.last_use < old->second.last_use) old = jt;
        if (old->second.exec) {
          (void)hipDeviceSynchronize();                    // a launch of it may still be in flight; evictions are rare (a 25th geometry)
          (void)hipGraphExecDestroy(old->second.exec);
        }
        h->graphs.erase(old);
      }
      it = h->graphs.emplace(key, glamr_nets::GraphEntry()).first;
    }
    {
      glamr_nets::GraphEntry& e = it->second;
      e.last_use = ++h->graph_clock;
      if (e.exec) {
        GLAMR_HIP_CHECK(hipGraphLaunch(e.exec, st));
        return GLAMR_OK;
      }
      ++e.seen;
      if ((e.seen == 2 || (e.seen == 1 && (flags & GLAMR_NETS_PERSISTENT))) && st != nullptr) {      // (the legacy default stream cannot be captured)
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
          (void)hipGetLastError();                       // not capturable: clear the error, keep the plain launches for this key
        } else {
          const int rc = enqueue();
          const hipError_t e_in = hipPeekAtLastError();
          hipGraph_t g = nullptr;
          const hipError_t ec = hipStreamEndCapture(st, &g);
          if (std::getenv("GLAMR_DEBUG_GRAPH")) std::fprintf(stderr, "graph capture: rc=%d in-capture error=%s end=%s B=%d max_len=%d flags=%d\n", rc, hipGetErrorString(e_in), hipGetErrorString(ec), B, max_len, flags);
          hipGraphExec_t exec = nullptr;
          if (rc == GLAMR_OK && ec == hipSuccess && g && hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess) {
            (void)hipGraphDestroy(g);
            e.exec = exec;
            GLAMR_HIP_CHECK(hipGraphLaunch(e.exec, st));
            return GLAMR_OK;
          }
          if (g) (void)hipGraphDestroy(g);
          (void)hipGetLastError();                 // capture failed: nothing was launched; run the plain sequence below and never retry this key
        }
      }
    }
  }
  return enqueue();
}

extern "C" int glamr_nets_infiller_window(glamr_nets* h, int B, int mode, const glamr_infiller_io* io, void* workspace, void* stream_) {
  GLAMR_REQUIRE(h && io && workspace && B > 0, "null argument");
  GLAMR_REQUIRE(mode == GLAMR_VAE_INFER || mode == GLAMR_VAE_TRAIN || mode == GLAMR_VAE_RECON, "mode must be GLAMR_VAE_INFER / TRAIN / RECON");
  GLAMR_REQUIRE(io->in_body_pose && io->frame_mask && io->out_body_pose, "in_body_pose, frame_mask and out_body_pose are required");
  GLAMR_REQUIRE(mode == GLAMR_VAE_INFER || io->body_pose, "the posterior encoder (train / recon) needs body_pose");
  GLAMR_REQUIRE(mode == GLAMR_VAE_RECON || io->eps, "sampling (infer / train) needs eps");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  tl_fp32 = h->fp32_only ? 1 : 0;
  tl_free = 0;
  Ws w = ws_layout(B, WIN, static_cast<char*>(workspace));
  hipLaunchKernelGGL(window_in_kernel, dim3(B, WIN), dim3(XLD), 0, st, io->in_body_pose, io->frame_mask, w.x, w.mask);
  if (mode != GLAMR_VAE_INFER) hipLaunchKernelGGL(window_in_kernel, dim3(B, WIN), dim3(XLD), 0, st, io->body_pose, io->frame_mask, w.gx, (unsigned char*)nullptr);
  RC(infiller_window(h, st, w, B, mode, io->eps, NZ, io->q_z, io->p_z));
  if (io->context) GLAMR_HIP_CHECK(hipMemcpyAsync(io->context, w.h0, (size_t)B * WIN * D * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (io->z) GLAMR_HIP_CHECK(hipMemcpyAsync(io->z, w.z, (size_t)B * NZ * sizeof(float), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(rows_out_kernel, dim3(B * CUR), dim3(128), 0, st, w.y, 128, B * CUR, 69, io->out_body_pose);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

extern "C" int glamr_nets_traj_clip(glamr_nets* h, int B, int T, int mode, const glamr_traj_io* io, void* workspace, void* stream_) {
  GLAMR_REQUIRE(h && io && workspace && B > 0 && T > 1, "null argument / empty clip");
  GLAMR_REQUIRE(mode == GLAMR_VAE_INFER || mode == GLAMR_VAE_TRAIN || mode == GLAMR_VAE_RECON, "mode must be GLAMR_VAE_INFER / TRAIN / RECON");
  GLAMR_REQUIRE(io->in_body_pose || io->in_joint_pos, "in_body_pose (joints by forward kinematics) or in_joint_pos is required");
  GLAMR_REQUIRE(mode == GLAMR_VAE_INFER || (io->trans && io->orient), "the posterior encoder (train / recon) needs trans and orient");
  GLAMR_REQUIRE(mode == GLAMR_VAE_RECON || io->eps, "sampling (infer / train) needs eps");
  GLAMR_REQUIRE(io->out_local_traj, "out_local_traj is required");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  tl_fp32 = h->fp32_only ? 1 : 0;
  tl_free = 0;
  Ws w = ws_layout(B, T, static_cast<char*>(workspace));
  std::vector<int> run(B, T), valid(B, T);
  if (io->valid_len > 0 && io->valid_len < T) std::fill(valid.begin(), valid.end(), io->valid_len);      // zero-padded chunk (get_seg_data)
  GLAMR_HIP_CHECK(hipMemcpyAsync(w.lens, run.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
  GLAMR_HIP_CHECK(hipMemcpyAsync(w.lens2, valid.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
  if (io->in_joint_pos) {
    hipLaunchKernelGGL(joints_in_kernel, dim3(B, T), dim3(XLD), 0, st, io->in_joint_pos, T, w.lens2, w.tx);
  } else {
    hipLaunchKernelGGL(pose_in_kernel, dim3(B, w.Tpad), dim3(64), 0, st, io->in_body_pose, T, w.Tpad, w.pose);
    hipLaunchKernelGGL(fk_joints_kernel, dim3(B, (T + FK_FRAMES - 1) / FK_FRAMES), dim3(256), 0, st, w.pose, w.Tpad, T, w.lens2, h->rest_joints, h->parents, w.tx);
  }
  const float* init = nullptr;
  if (io->trans && io->orient) {
    hipLaunchKernelGGL(traj_prepare_kernel, dim3(B), dim3(256), 0, st, io->trans, io->orient, T, w.lens, w.tloc, w.e6);
    if (io->local_traj) GLAMR_HIP_CHECK(hipMemcpyAsync(io->local_traj, w.tloc, (size_t)B * T * 11 * sizeof(float), hipMemcpyDeviceToDevice, st));
    init = w.tloc;            // DataDecoder :322-324: the first row's xy and heading vector come from local_traj_tp when it exists
  }
  int ldinit = T * 11;
  if (io->init_row) { init = io->init_row; ldinit = 11; }      // :319-321 takes precedence
  RC(traj_pass(h, st, w, B, T, w.lens, mode, io->eps, io->q_z, io->p_z, init, ldinit, io->out_orig_local_traj, io->out_local_traj, io->out_trans,
               io->out_orient, io->out_orient_q));
  if (io->z) GLAMR_HIP_CHECK(hipMemcpyAsync(io->z, w.z, (size_t)B * NZ * sizeof(float), hipMemcpyDeviceToDevice, st));
  GLAMR_HIP_CHECK(hipStreamSynchronize(st));          // `run` / `valid` are pageable host buffers
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

extern "C" int glamr_traj_local_to_global(int B, int T, const float* local_traj, float* out_trans, float* out_orient, float* out_orient_q, void* workspace,
                                          void* stream_) {
  GLAMR_REQUIRE(local_traj && workspace && B > 0 && T > 0 && (out_trans || out_orient || out_orient_q), "null argument");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  // workspace: [B] lengths, [B][T][11] rows, [B][T][3] scan buffers  (glamr_traj_local_to_global_workspace_bytes)
  int* lens = static_cast<int*>(workspace);
  float* rows = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)B * sizeof(int), 256));
  float* scr = rows + (size_t)B * T * 11;
  std::vector<int> run(B, T);
  GLAMR_HIP_CHECK(hipMemcpyAsync(lens, run.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(traj_to_global2_kernel, dim3(B), dim3(256), 0, st, local_traj, 11, T, lens, (const float*)nullptr, 0, 0, rows, out_trans, out_orient,
                     out_orient_q, scr);
  GLAMR_HIP_CHECK(hipStreamSynchronize(st));
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

extern "C" size_t glamr_traj_local_to_global_workspace_bytes(int B, int T) {
  if (B <= 0 || T <= 0) return 0;
  return align_up((size_t)B * sizeof(int), 256) + (size_t)B * T * 14 * sizeof(float);
}

// ---- taped infiller: forward with every activation kept + backward w.r.t. the latent draws (latent-optimisation mode) -----------------
extern "C" size_t glamr_nets_tape_bytes(const glamr_nets* h, int n_seq, int max_len) {
  if (!h || n_seq <= 0 || max_len <= PAST) return 0;
  return tape_layout(n_seq, max_len, nullptr).total;
}

namespace {
int tape_windows(const int32_t* lens_host, int B, int max_len, int* n_win) {
  int longest = 0;
  for (int b = 0; b < B; ++b) {
    GLAMR_REQUIRE(lens_host[b] > PAST && lens_host[b] <= max_len, "sequence %d has length %d (need %d < len <= max_len)", b, lens_host[b], PAST);
    longest = std::max(longest, (int)lens_host[b]);
  }
  *n_win = (longest - PAST + CUR - 1) / CUR;
  return GLAMR_OK;
}
}  // namespace

extern "C" int glamr_nets_infill_taped(glamr_nets* h, int B, int max_len, const int32_t* lens_host, const float* body_pose, const float* visible,
                                       const float* motion_eps, int n_win_max, float* out_pose, void* tape_, void* stream_) {
  GLAMR_REQUIRE(h && lens_host && body_pose && visible && motion_eps && out_pose && tape_, "null argument");
  GLAMR_REQUIRE(B > 0 && max_len > PAST, "need n_seq > 0 and max_len > %d", PAST);
  int n_win = 0;
  RC(tape_windows(lens_host, B, max_len, &n_win));
  GLAMR_REQUIRE(n_win <= n_win_max, "motion_eps holds %d windows per sequence, %d needed", n_win_max, n_win);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  tl_fp32 = h->fp32_only ? 1 : 0;
  tl_free = 0;
  Tape t = tape_layout(B, max_len, static_cast<char*>(tape_));
  // (a caller capturing this stream -- the latent-optimisation mode replays its iteration as a HIP graph -- gets the lengths uploaded from
  // the handle's pinned table, as glamr_nets_infer does: a copy node must not read pageable memory that is gone at replay)
  const int32_t* lens_src = lens_host;
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive) {
      std::lock_guard<std::mutex> lock(h->graph_mu);
      GLAMR_REQUIRE(h->capture_lens && h->capture_lens_used + (size_t)B <= CAPTURE_LENS_INTS,
                    "the handle's table for calls recorded into caller graphs is full (%zu lengths); destroy and re-create the handle", CAPTURE_LENS_INTS);
      int32_t* pinned = h->capture_lens + h->capture_lens_used;
      h->capture_lens_used += (size_t)B;
      std::memcpy(pinned, lens_host, (size_t)B * sizeof(int32_t));
      lens_src = pinned;
    }
  }
  GLAMR_HIP_CHECK(hipMemcpyAsync(t.lens, lens_src, (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(pose_in_kernel, dim3(B, t.Tpad), dim3(64), 0, st, body_pose, max_len, t.Tpad, t.pose);
  for (int i = 0; i < n_win; ++i) {
    WinTape& w = t.win[i];
    const int s = i * CUR;
    hipLaunchKernelGGL(window_gather_kernel, dim3(B, WIN), dim3(64), 0, st, t.pose, visible, t.lens, t.Tpad, max_len, s, w.x, w.mask);
    RC(taped_window(h, st, w, B, motion_eps + (size_t)i * NZ, n_win_max * NZ));
    hipLaunchKernelGGL(window_scatter_kernel, dim3(B, CUR), dim3(64), 0, st, w.y, 128, t.lens, t.Tpad, s, t.pose);
  }
  hipLaunchKernelGGL(pose_out_kernel, dim3(B, max_len), dim3(64), 0, st, t.pose, max_len, t.Tpad, t.lens, out_pose);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

extern "C" int glamr_nets_infill_backward(glamr_nets* h, int B, int max_len, const int32_t* lens_host, const float* motion_eps, int n_win_max,
                                          const float* g_out_pose, float* g_motion_eps, void* tape_, void* stream_) {
  GLAMR_REQUIRE(h && lens_host && motion_eps && g_out_pose && g_motion_eps && tape_, "null argument");
  GLAMR_REQUIRE(B > 0 && max_len > PAST, "need n_seq > 0 and max_len > %d", PAST);
  int n_win = 0;
  RC(tape_windows(lens_host, B, max_len, &n_win));
  GLAMR_REQUIRE(n_win <= n_win_max, "motion_eps holds %d windows per sequence, %d needed", n_win_max, n_win);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  tl_fp32 = h->fp32_only ? 1 : 0;
  tl_free = 0;
  Tape t = tape_layout(B, max_len, static_cast<char*>(tape_));
  GLAMR_HIP_CHECK(hipMemsetAsync(t.pose + t.values, 0, t.values * sizeof(float), st));            // every gradient starts at zero
  GLAMR_HIP_CHECK(hipMemsetAsync(g_motion_eps, 0, (size_t)B * n_win_max * NZ * sizeof(float), st));
  TapeCtx c{h, st, &t};
  hipLaunchKernelGGL(pose_out_bwd_kernel, dim3(B, t.Tpad), dim3(XLD), 0, st, g_out_pose, max_len, t.Tpad, t.lens, GR(t, t.pose));
  for (int i = n_win - 1; i >= 0; --i) {
    const WinTape& w = t.win[i];
    const int s = i * CUR;
    hipLaunchKernelGGL(window_scatter_bwd_kernel, dim3(B, CUR), dim3(XLD), 0, st, GR(t, t.pose), t.lens, t.Tpad, s, GR(t, w.y), 128);
    RC(taped_window_bwd(c, w, B, motion_eps + (size_t)i * NZ, n_win_max * NZ, g_motion_eps + (size_t)i * NZ, n_win_max * NZ));
    hipLaunchKernelGGL(window_gather_bwd_kernel, dim3(B, WIN), dim3(XLD), 0, st, GR(t, w.x), t.lens, t.Tpad, s, GR(t, t.pose));
  }
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

// 0 = fp32-grade products on the fp16 matrix cores (two-plane operands), 1 = plain fp32 kernels only (the range analysis of
// glamr_nets_create found a value that could leave fp16's range, or GLAMR_NETS_FORCE_FP32).  worst_case[0] = largest possible magnitude of
// a converted activation, worst_case[1] = largest weight (either may be NULL).
extern "C" int glamr_nets_precision(const glamr_nets* h, double* worst_case) {
  if (!h) return -1;
  if (worst_case) { worst_case[0] = h->worst_activation; worst_case[1] = h->worst_weight; }
  return h->fp32_only ? 1 : 0;
}
