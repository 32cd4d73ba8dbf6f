// Taped forward and explicit backward of the motion infiller: gradients of the generated body pose w.r.t. the latent draws
// (`in_motion_latent`), for the latent-optimisation mode of the global optimiser (global_recon_model.py:43-44,155-158,434-437: the
// infiller runs inside the Adam loop and `motion_latent` is a parameter; the trajectory predictor's output is detached by
// get_pred_trajectory_base :396, so it needs no backward).  What torch autograd does through MotionInfillerVAE.inference_multi_step
// (motion_infiller_vae.py:618-632): every window's context encoder, prior, reparameterisation and decoder, and the autoregression --
// a window's output frames are the next window's past frames (:604-607).
//
// Included by nets.hip INSIDE its anonymous namespace (not a stand-alone header).  The forward below is the small-batch launch sequence of infiller_window (separate
// GEMM / attention / LayerNorm kernels, all fp32 MFMA) with every activation of every window kept in a tape arena; the backward walks
// the windows in reverse and each layer in reverse.  Linear layers: dX = dY W with the TRANSPOSED weight through the same GEMM kernel
// (transposes are made once per handle, the first time a tape is asked for); weights are constants (no weight gradients).

// ---- backward kernels ---------------------------------------------------------------------------------------------------------------
__global__ void acc_kernel(float* dst, const float* src, size_t n) {                       // dst += src
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
__global__ void relu_mask_kernel(float* g, const float* y, size_t n) {                     // g *= (y > 0)
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && !(y[i] > 0.0f)) g[i] = 0.0f;
}
// y = LayerNorm(x + r) * gamma + beta over 256 columns:  dsum = rstd (gy - mean(gy) - xhat mean(gy xhat)),  gy = dy gamma;  dx += dsum, dr += dsum
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* dY, const float* X, const float* R, const float* gamma, float* dX, float* dR, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[4], gy[4];
  for (int k = 0; k < 4; ++k) {
    const size_t i = (size_t)row * D + lane * 4 + k;
    v[k] = X[i] + (R ? R[i] : 0.0f);
    gy[k] = dY[i] * gamma[lane * 4 + k];
  }
  float s = v[0] + v[1] + v[2] + v[3];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s * (1.0f / 256.0f);
  float q = 0.f;
  for (int k = 0; k < 4; ++k) { v[k] -= mean; q += v[k] * v[k]; }
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
  float a = 0.f, c = 0.f;
  for (int k = 0; k < 4; ++k) { v[k] *= rstd; a += gy[k]; c += gy[k] * v[k]; }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); c += __shfl_xor(c, off); }
  a *= (1.0f / 256.0f); c *= (1.0f / 256.0f);
  for (int k = 0; k < 4; ++k) {
    const size_t i = (size_t)row * D + lane * 4 + k;
    const float d = rstd * (gy[k] - a - v[k] * c);
    dX[i] += d;
    if (dR) dR[i] += d;
  }
}
// Multi-head attention backward, 8 heads x 32 dims, Lq, Lk <= 64, one workgroup (64 threads) per (sequence, head); probabilities are
// recomputed.  O = P V, P = softmax(Q K^T / sqrt(32)) over the unmasked keys:
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - rowsum(dP o P)),  dQ = dS K / sqrt(32),  dK = dS^T Q / sqrt(32)        (accumulated into dQ / dK / dV)
// q_shared: the queries are constants shared by all sequences (the prior's learned tokens): no dQ.
// 256 threads per (sequence, head): thread (i, g) = (row, quarter).  Every output element is formed by the SAME chain of operations as when one
// lane did a whole row (round 3: 98 us per call, 70 calls per iteration of the latent-optimisation mode = a third of it): the rows' scores and
// dO V^T are spread over the quarters by key (j = g mod 4), the softmax of a row stays with one thread, dQ / dK / dV are spread by channel
// (8 of the 32 per thread).
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* Q, int ldq, const float* K, const float* V, int ldk, const unsigned char* key_mask,
                                                            const float* dO, int ldo, float* dQ, float* dK, float* dV, int Lq, int Lk, int q_shared) {
  // (rows of 36 floats: 16-byte aligned, so the channel loops below read four values per LDS instruction -- most of them broadcasts of one key row)
  __shared__ __attribute__((aligned(16))) float sQ[64][36], sK[64][36], sV[64][36], sdO[64][36];
  __shared__ float sP[64][65], sdS[64][65];
  __shared__ float sPart[64][4];
  __shared__ unsigned char sM[64];
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, i = tid & 63, g = tid >> 6;
  const float scale = 0.17677669529663687f;      // 1 / sqrt(32)
  for (int idx = tid; idx < 64 * 32; idx += 256) {
    const int r = idx >> 5, d = idx & 31;
    sQ[r][d] = r < Lq ? Q[(size_t)((q_shared ? 0 : b * Lq) + r) * ldq + h * 32 + d] : 0.f;
    sdO[r][d] = r < Lq ? dO[(size_t)(b * Lq + r) * ldo + h * 32 + d] : 0.f;
    sK[r][d] = r < Lk ? K[(size_t)(b * Lk + r) * ldk + h * 32 + d] : 0.f;
    sV[r][d] = r < Lk ? V[(size_t)(b * Lk + r) * ldk + h * 32 + d] : 0.f;
  }
  if (tid < 64) sM[tid] = (tid < Lk) ? (key_mask ? key_mask[(size_t)b * Lk + tid] : 0) : 1;
  __syncthreads();
  // scaled scores and dP = dO V^T of row i, keys j = g, g + 4, ...
  if (i < Lq) {
    float q[32], o[32];
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sQ[i][d4 * 4]), b4 = *reinterpret_cast<const f32x4*>(&sdO[i][d4 * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { q[d4 * 4 + e] = a4[e]; o[d4 * 4 + e] = b4[e]; }
    }
    for (int j = g; j < Lk; j += 4) {
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < 8; ++d4) {
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(&sK[j][d4 * 4]), v4 = *reinterpret_cast<const f32x4*>(&sV[j][d4 * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc = fmaf(q[d4 * 4 + e], k4[e], sc); dp = fmaf(o[d4 * 4 + e], v4[e], dp); }
      }
      sP[i][j] = sc * scale;
      sdS[i][j] = dp;
    }
  } else {
    for (int j = g; j < 64; j += 4) { sP[i][j] = 0.f; sdS[i][j] = 0.f; }
  }
  // softmax over the unmasked keys and dS = P o (dP - rowsum(dP o P)).  Round 3 - 5: one thread per row, four dependent loops over the keys with an LDS
  // round trip (and an expf) per trip -- 8 of the kernel's 29 us, 70 calls per iteration of the latent-optimisation mode.  Round 6: what has no order
  // (the row maximum) or no neighbour (the exponentials, the final products) is spread over the row's four threads by key, as the scores were; the two
  // SUMS stay with one thread per row, added key by key in the order they always were -- every output bit as before.
  if (i < Lq) {
    float pm = -3.0e38f;
    for (int j = g; j < Lk; j += 4) if (!sM[j]) pm = fmaxf(pm, sP[i][j]);      // (own writes of the loop above: no barrier needed)
    sPart[i][g] = pm;
  }
  __syncthreads();
  if (i < Lq) {
    const float mx = fmaxf(fmaxf(sPart[i][0], sPart[i][1]), fmaxf(sPart[i][2], sPart[i][3]));
    for (int j = g; j < Lk; j += 4) sP[i][j] = sM[j] ? 0.f : expf(sP[i][j] - mx);
  }
  __syncthreads();
  if (g == 0 && i < Lq) {
    // (four keys' values requested together, then added one after the other: the order of the sums is the keys' order)
    float sum = 0.f;
    int j = 0;
    for (; j + 4 <= Lk; j += 4) {
      const float e0 = sP[i][j], e1 = sP[i][j + 1], e2 = sP[i][j + 2], e3 = sP[i][j + 3];
      sum += e0; sum += e1; sum += e2; sum += e3;
    }
    for (; j < Lk; ++j) sum += sP[i][j];
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    float dot = 0.f;
    for (j = 0; j + 4 <= Lk; j += 4) {
      const float e0 = sP[i][j], e1 = sP[i][j + 1], e2 = sP[i][j + 2], e3 = sP[i][j + 3];
      const float d0_ = sdS[i][j], d1_ = sdS[i][j + 1], d2_ = sdS[i][j + 2], d3_ = sdS[i][j + 3];
      dot = fmaf(d0_, e0 * inv, dot); dot = fmaf(d1_, e1 * inv, dot); dot = fmaf(d2_, e2 * inv, dot); dot = fmaf(d3_, e3 * inv, dot);
    }
    for (; j < Lk; ++j) dot = fmaf(sdS[i][j], sP[i][j] * inv, dot);
    sPart[i][0] = inv;
    sPart[i][1] = dot;
  }
  __syncthreads();
  if (i < Lq) {
    const float inv = sPart[i][0], dot = sPart[i][1];
    for (int j = g; j < Lk; j += 4) { const float pr = sP[i][j] * inv; sP[i][j] = pr; sdS[i][j] = pr * (sdS[i][j] - dot); }
  }
  __syncthreads();
  const int d0 = g * 8;
  if (i < Lq && dQ && !q_shared) {               // dQ = dS K / sqrt(32): row i, channels d0 .. d0 + 7
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < Lk; ++j) {
      const float ds = sdS[i][j];
      const f32x4 k0 = *reinterpret_cast<const f32x4*>(&sK[j][d0]), k1 = *reinterpret_cast<const f32x4*>(&sK[j][d0 + 4]);
#pragma unroll
      for (int d = 0; d < 4; ++d) { acc[d] = fmaf(ds, k0[d], acc[d]); acc[4 + d] = fmaf(ds, k1[d], acc[4 + d]); }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) dQ[(size_t)(b * Lq + i) * ldq + h * 32 + d0 + d] += acc[d] * scale;
  }
  if (i < Lk) {                                  // dK = dS^T Q / sqrt(32), dV = P^T dO: key row j = i, channels d0 .. d0 + 7
    const int j = i;
    float gk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int qr = 0; qr < Lq; ++qr) {
      const float ds = sdS[qr][j], pp = sP[qr][j];
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(&sQ[qr][d0]), q1 = *reinterpret_cast<const f32x4*>(&sQ[qr][d0 + 4]);
      const f32x4 o0 = *reinterpret_cast<const f32x4*>(&sdO[qr][d0]), o1 = *reinterpret_cast<const f32x4*>(&sdO[qr][d0 + 4]);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        gk[d] = fmaf(ds, q0[d], gk[d]); gk[4 + d] = fmaf(ds, q1[d], gk[4 + d]);
        gv[d] = fmaf(pp, o0[d], gv[d]); gv[4 + d] = fmaf(pp, o1[d], gv[4 + d]);
      }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      dK[(size_t)(b * Lk + j) * ldk + h * 32 + d0 + d] += gk[d] * scale;
      dV[(size_t)(b * Lk + j) * ldk + h * 32 + d0 + d] += gv[d];
    }
  }
}
// z = mu + eps exp(0.5 logvar): d mu = dz, d logvar = dz eps 0.5 exp(0.5 logvar), d eps = dz exp(0.5 logvar)   (pz rows as reparam_infiller_kernel)
__global__ void reparam_infiller_bwd_kernel(const float* pz, const float* eps, int eps_stride, const float* dz, float* dpz, float* deps, int deps_stride) {
  const int b = blockIdx.x, k = threadIdx.x;
  const float lv = pz[((size_t)b * 2 + 1) * D + NZ + k], sd = expf(0.5f * lv), g = dz[(size_t)b * NZ + k];
  dpz[((size_t)b * 2 + 0) * D + k] += g;
  dpz[((size_t)b * 2 + 1) * D + NZ + k] += g * eps[(size_t)b * eps_stride + k] * 0.5f * sd;
  deps[(size_t)b * deps_stride + k] += g * sd;
}
__global__ void build_queries_bwd_kernel(const float* dq, float* dzproj) {                   // dzproj[b][k] += sum_i dq[b][i][k]
  const int b = blockIdx.x, k = threadIdx.x;
  float acc = 0.f;
  for (int i = 0; i < CUR; ++i) acc += dq[((size_t)b * CUR + i) * D + k];
  dzproj[(size_t)b * D + k] += acc;
}
// gradient of the running pose buffer -> gradient of window s's 30 output rows (the frames window_scatter_kernel wrote), consumed
__global__ void window_scatter_bwd_kernel(float* gpose, const int* lens, int Tpad, int s, float* dy, int ldy) {
  const int b = blockIdx.x, i = blockIdx.y, c = threadIdx.x;
  const int t = s + PAST + i;
  float g = 0.f;
  if (c < 69 && t < lens[b] && s < lens[b] - PAST) { g = gpose[((size_t)b * Tpad + t) * XLD + c]; gpose[((size_t)b * Tpad + t) * XLD + c] = 0.f; }
  if (c < ldy) dy[((size_t)b * CUR + i) * ldy + c] = g;
}
__global__ void window_gather_bwd_kernel(const float* dx, const int* lens, int Tpad, int s, float* gpose) {
  const int b = blockIdx.x, j = blockIdx.y, c = threadIdx.x;
  const int t = s + j;
  if (t < lens[b]) gpose[((size_t)b * Tpad + t) * XLD + c] += dx[((size_t)b * WIN + j) * XLD + c];
}
__global__ void pose_out_bwd_kernel(const float* g_out, int max_len, int Tpad, const int* lens, float* gpose) {
  const int b = blockIdx.x, t = blockIdx.y, c = threadIdx.x;
  gpose[((size_t)b * Tpad + t) * XLD + c] = (c < 69 && t < max_len && t < lens[b]) ? g_out[((size_t)b * max_len + t) * 69 + c] : 0.0f;
}

// ---- tape layout --------------------------------------------------------------------------------------------------------------------
struct EncTape { float *qkv, *att, *tmp1, *mid, *ff, *tmp2, *out; };
struct DecTape { float *qkv, *att_s, *tmp_s, *xa, *qbuf, *ctxkv, *att_c, *tmp_c, *xb, *ff, *tmp_f, *xc; };
struct WinTape {
  float *x, *h0; EncTape enc[2];
  float *p_ctxkv, *p_att, *p_tmp1, *p_x1, *p_a, *p_ff, *p_tmp2, *p_b, *pz, *z;
  float *zproj, *q0; DecTape dec[2];
  float *o1, *o2, *y;
  unsigned char* mask;
};
struct Tape {
  float* pose; int* lens; int Tpad, n_win; size_t values, total;      // `values` floats of activations, then the same layout again for their gradients
  std::vector<WinTape> win;
};
// the gradient of an activation lives at the same offset of the second half of the arena
inline float* GR(const Tape& t, float* v) { return v + t.values; }

Tape tape_layout(int B, int max_len, char* base) {
  Tape t;
  t.n_win = std::max(1, (max_len - PAST + CUR - 1) / CUR);
  t.Tpad = std::max(max_len, (t.n_win - 1) * CUR + WIN);
  float* fb = reinterpret_cast<float*>(base);
  size_t off = 0;
  auto take = [&](size_t n) { float* p = fb + off; off += (n + 63) / 64 * 64; return p; };
  const size_t MW = (size_t)B * WIN, MC = (size_t)B * CUR, M2 = (size_t)B * 2;
  t.pose = take((size_t)B * t.Tpad * XLD);
  t.win.resize(t.n_win);
  for (WinTape& w : t.win) {
    w.x = take(MW * XLD); w.h0 = take(MW * D);
    for (EncTape& e : w.enc) { e.qkv = take(MW * 3 * D); e.att = take(MW * D); e.tmp1 = take(MW * D); e.mid = take(MW * D); e.ff = take(MW * FF); e.tmp2 = take(MW * D); e.out = take(MW * D); }
    w.p_ctxkv = take(MW * 2 * D); w.p_att = take(M2 * D); w.p_tmp1 = take(M2 * D); w.p_x1 = take(M2 * D); w.p_a = take(M2 * D); w.p_ff = take(M2 * FF);
    w.p_tmp2 = take(M2 * D); w.p_b = take(M2 * D); w.pz = take(M2 * D); w.z = take((size_t)B * NZ);
    w.zproj = take((size_t)B * D); w.q0 = take(MC * D);
    for (DecTape& d : w.dec) {
      d.qkv = take(MC * 3 * D); d.att_s = take(MC * D); d.tmp_s = take(MC * D); d.xa = take(MC * D); d.qbuf = take(MC * D); d.ctxkv = take(MW * 2 * D);
      d.att_c = take(MC * D); d.tmp_c = take(MC * D); d.xb = take(MC * D); d.ff = take(MC * FF); d.tmp_f = take(MC * D); d.xc = take(MC * D);
    }
    w.o1 = take(MC * FF); w.o2 = take(MC * D); w.y = take(MC * 128);
  }
  t.values = off;
  // after the two halves: the key-padding masks and the lengths
  size_t tail = 2 * off * sizeof(float);
  for (WinTape& w : t.win) { w.mask = reinterpret_cast<unsigned char*>(base + tail); tail += align_up(MW, 256); }
  t.lens = reinterpret_cast<int*>(base + tail);
  tail += align_up((size_t)B * sizeof(int), 256);
  t.total = tail;
  return t;
}

// transposed copies of the linear layers the backward multiplies with (made once per handle)
const Lin* transposed(glamr_nets* h, const Lin& L) {
  std::lock_guard<std::mutex> lock(h->graph_mu);
  auto it = h->lin_T.find(&L);
  if (it != h->lin_T.end()) return &it->second;
  const int Np = (L.N + 63) / 64 * 64;
  std::vector<float> W((size_t)Np * L.K);
  if (hipMemcpy(W.data(), L.W, W.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
  Lin T;
  T.N = L.K;                                   // outputs of the transposed layer = inputs of the layer (padded K: the extra rows are zero)
  T.K = (L.N + 31) / 32 * 32;
  const int Tp = (T.N + 63) / 64 * 64;
  std::vector<float> Wt((size_t)Tp * T.K, 0.0f);
  for (int n = 0; n < L.N; ++n) for (int k = 0; k < L.K; ++k) Wt[(size_t)k * T.K + n] = W[(size_t)n * L.K + k];
  if (hipMalloc(reinterpret_cast<void**>(&T.W), Wt.size() * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMemcpy(T.W, Wt.data(), Wt.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  h->allocs.push_back(T.W);
  if (L.Ws) {      // the layer is cleared for the split-fp16 kernels (range analysis of glamr_nets_create): so is its transpose, entry for entry
    const std::vector<unsigned short> planes = split_planes(Wt, Tp, T.K);
    if (hipMalloc(reinterpret_cast<void**>(&T.Ws), planes.size() * sizeof(unsigned short)) != hipSuccess) return nullptr;
    if (hipMemcpy(T.Ws, planes.data(), planes.size() * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    h->allocs.push_back(T.Ws);
  }
  return &h->lin_T.emplace(&L, T).first->second;
}

struct TapeCtx { glamr_nets* h; hipStream_t st; const Tape* t; };
inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

// dX += act'(.) dY W  (and dR += dY for a residual added after the activation).  dY is modified in place when the layer has a ReLU.
int lin_bwd(const TapeCtx& c, const Lin& L, float* dY, int ldy, const float* Y, int act, float* dX, int ldx, int M, float* dR = nullptr) {
  if (dR) hipLaunchKernelGGL(acc_kernel, dim3(blocks((size_t)M * ldy)), dim3(256), 0, c.st, dR, dY, (size_t)M * ldy);      // (ldy == ldr == the row width)
  if (act == ACT_RELU) hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks((size_t)M * ldy)), dim3(256), 0, c.st, dY, Y, (size_t)M * ldy);
  if (!dX) return GLAMR_OK;
  const Lin* T = transposed(c.h, L);
  if (!T) return fail(GLAMR_E_HIP, "could not build the transposed weights of a layer");
  if (T->K > ldy) return fail(GLAMR_E_INVALID, "lin_bwd: gradient rows of %d floats, %d needed", ldy, T->K);
  return lin(c.st, *T, dY, ldy, dX, ldx, M, ACT_NONE, dX, ldx, nullptr, 1, 0, -1, -1, true);      // accumulates through the residual input (few rows: the one-wave split-fp16 kernel)
}
int ln_bwd(const TapeCtx& c, const LN& n, const float* dY, const float* X, const float* R, float* dX, float* dR, int rows) {
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, c.st, dY, X, R, n.g, dX, dR, rows);
  return GLAMR_OK;
}
void attn_bwd(const TapeCtx& c, const float* Q, int ldq, const float* K, const float* V, int ldk, const unsigned char* mask, const float* dO, float* dQ, float* dK,
              float* dV, int B, int Lq, int Lk, int q_shared) {
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(B, 8), dim3(256), 0, c.st, Q, ldq, K, V, ldk, mask, dO, D, dQ, dK, dV, Lq, Lk, q_shared);
}

// ---- forward of one window, every activation kept (mirrors infiller_window / encoder_layer / decoder_layer, small-batch path) -------
int taped_window(glamr_nets* h, hipStream_t st, WinTape& w, int B, const float* eps, int eps_stride) {
  const int M = B * WIN, MC = B * CUR;
  RC(lin(st, h->enc_in, w.x, XLD, w.h0, D, M, ACT_NONE, nullptr, 0, h->enc_pe, -WIN, D));
  const float* hin = w.h0;
  for (int l = 0; l < 2; ++l) {
    const EncLayer& E = h->enc[l];
    EncTape& e = w.enc[l];
    RC(lin(st, E.qkv, hin, D, e.qkv, 3 * D, M));
    launch_attention(dim3(B, 8), dim3(64), 0, st, e.qkv, 3 * D, e.qkv + D, e.qkv + 2 * D, 3 * D, w.mask, e.att, D, WIN, WIN, 0);
    RC(lin(st, E.o, e.att, D, e.tmp1, D, M));
    RC(ln(st, e.tmp1, hin, E.n1, e.mid, M));
    RC(lin(st, E.f1, e.mid, D, e.ff, FF, M, ACT_RELU));
    RC(lin(st, E.f2, e.ff, FF, e.tmp2, D, M));
    RC(ln(st, e.tmp2, e.mid, E.n2, e.out, M));
    hin = e.out;
  }
  const float* ctx = hin;
  // prior
  RC(lin(st, h->prior_kv, ctx, D, w.p_ctxkv, 2 * D, M));
  launch_attention(dim3(B, 8), dim3(64), 0, st, h->prior_q, D, w.p_ctxkv, w.p_ctxkv + D, 2 * D, w.mask, w.p_att, D, 2, WIN, 1);
  RC(lin(st, h->prior_o, w.p_att, D, w.p_tmp1, D, B * 2));
  hipLaunchKernelGGL(tile_rows_kernel, dim3((B * 2 * D + 255) / 256), dim3(64), 0, st, w.p_x1, h->prior_x1, 2, B * 2 * D);
  RC(ln(st, w.p_tmp1, w.p_x1, h->prior_n2, w.p_a, B * 2));
  RC(lin(st, h->prior_f1, w.p_a, D, w.p_ff, FF, B * 2, ACT_RELU));
  RC(lin(st, h->prior_f2, w.p_ff, FF, w.p_tmp2, D, B * 2));
  RC(ln(st, w.p_tmp2, w.p_a, h->prior_n3, w.p_b, B * 2));
  RC(lin(st, h->prior_pz, w.p_b, D, w.pz, D, B * 2));
  hipLaunchKernelGGL(reparam_infiller_kernel, dim3(B), dim3(64), 0, st, w.pz, eps, eps_stride, w.z, B);
  // decoder
  RC(lin(st, h->dec_z, w.z, NZ, w.zproj, D, B));
  hipLaunchKernelGGL(build_queries_kernel, dim3(B, CUR), dim3(64), 0, st, w.zproj, h->dec_pe, w.q0);
  const float* xin = w.q0;
  for (int l = 0; l < 2; ++l) {
    const DecLayer& Dl = h->dec[l];
    DecTape& d = w.dec[l];
    RC(lin(st, Dl.sa_qkv, xin, D, d.qkv, 3 * D, MC));
    launch_attention(dim3(B, 8), dim3(64), 0, st, d.qkv, 3 * D, d.qkv + D, d.qkv + 2 * D, 3 * D, (const unsigned char*)nullptr, d.att_s, D, CUR, CUR, 0);
    RC(lin(st, Dl.sa_o, d.att_s, D, d.tmp_s, D, MC));
    RC(ln(st, d.tmp_s, xin, Dl.n1, d.xa, MC));
    RC(lin(st, Dl.ca_q, d.xa, D, d.qbuf, D, MC));
    RC(lin(st, Dl.ca_kv, ctx, D, d.ctxkv, 2 * D, M));
    launch_attention(dim3(B, 8), dim3(64), 0, st, d.qbuf, D, d.ctxkv, d.ctxkv + D, 2 * D, w.mask, d.att_c, D, CUR, WIN, 0);
    RC(lin(st, Dl.ca_o, d.att_c, D, d.tmp_c, D, MC));
    RC(ln(st, d.tmp_c, d.xa, Dl.n2, d.xb, MC));
    RC(lin(st, Dl.f1, d.xb, D, d.ff, FF, MC, ACT_RELU));
    RC(lin(st, Dl.f2, d.ff, FF, d.tmp_f, D, MC));
    RC(ln(st, d.tmp_f, d.xb, Dl.n3, d.xc, MC));
    xin = d.xc;
  }
  RC(lin(st, h->out1, xin, D, w.o1, FF, MC, ACT_RELU));
  RC(lin(st, h->out2, w.o1, FF, w.o2, D, MC, ACT_RELU));
  RC(lin(st, h->outfc, w.o2, D, w.y, 128, MC));
  return GLAMR_OK;
}

// ---- backward of one window: GR(w.y) holds dL/dy on entry; on return GR(w.x) holds dL/dx and deps has received dL/d eps -----------------
int taped_window_bwd(const TapeCtx& c, const WinTape& w, int B, const float* eps, int eps_stride, float* deps, int deps_stride) {
  glamr_nets* h = c.h;
  const Tape& t = *c.t;
  const int M = B * WIN, MC = B * CUR;
  auto G = [&](float* v) { return GR(t, v); };
  const float* ctx = w.enc[1].out;
  float* dctx = G(w.enc[1].out);
  RC(lin_bwd(c, h->outfc, G(w.y), 128, nullptr, ACT_NONE, G(w.o2), D, MC));
  RC(lin_bwd(c, h->out2, G(w.o2), D, w.o2, ACT_RELU, G(w.o1), FF, MC));
  const float* xin1 = w.dec[0].xc;
  RC(lin_bwd(c, h->out1, G(w.o1), FF, w.o1, ACT_RELU, G(w.dec[1].xc), D, MC));
  for (int l = 1; l >= 0; --l) {
    const DecLayer& Dl = h->dec[l];
    const DecTape& d = w.dec[l];
    float* xin = l == 0 ? w.q0 : const_cast<float*>(xin1);
    RC(ln_bwd(c, Dl.n3, G(d.xc), d.tmp_f, d.xb, G(d.tmp_f), G(d.xb), MC));
    RC(lin_bwd(c, Dl.f2, G(d.tmp_f), D, nullptr, ACT_NONE, G(d.ff), FF, MC));
    RC(lin_bwd(c, Dl.f1, G(d.ff), FF, d.ff, ACT_RELU, G(d.xb), D, MC));
    RC(ln_bwd(c, Dl.n2, G(d.xb), d.tmp_c, d.xa, G(d.tmp_c), G(d.xa), MC));
    RC(lin_bwd(c, Dl.ca_o, G(d.tmp_c), D, nullptr, ACT_NONE, G(d.att_c), D, MC));
    attn_bwd(c, d.qbuf, D, d.ctxkv, d.ctxkv + D, 2 * D, w.mask, G(d.att_c), G(d.qbuf), G(d.ctxkv), G(d.ctxkv) + D, B, CUR, WIN, 0);
    RC(lin_bwd(c, Dl.ca_kv, G(d.ctxkv), 2 * D, nullptr, ACT_NONE, dctx, D, M));
    RC(lin_bwd(c, Dl.ca_q, G(d.qbuf), D, nullptr, ACT_NONE, G(d.xa), D, MC));
    RC(ln_bwd(c, Dl.n1, G(d.xa), d.tmp_s, xin, G(d.tmp_s), G(xin), MC));
    RC(lin_bwd(c, Dl.sa_o, G(d.tmp_s), D, nullptr, ACT_NONE, G(d.att_s), D, MC));
    attn_bwd(c, d.qkv, 3 * D, d.qkv + D, d.qkv + 2 * D, 3 * D, nullptr, G(d.att_s), G(d.qkv), G(d.qkv) + D, G(d.qkv) + 2 * D, B, CUR, CUR, 0);
    RC(lin_bwd(c, Dl.sa_qkv, G(d.qkv), 3 * D, nullptr, ACT_NONE, G(xin), D, MC));
  }
  hipLaunchKernelGGL(build_queries_bwd_kernel, dim3(B), dim3(D), 0, c.st, G(w.q0), G(w.zproj));
  RC(lin_bwd(c, h->dec_z, G(w.zproj), D, nullptr, ACT_NONE, G(w.z), NZ, B));
  hipLaunchKernelGGL(reparam_infiller_bwd_kernel, dim3(B), dim3(NZ), 0, c.st, w.pz, eps, eps_stride, G(w.z), G(w.pz), deps, deps_stride);
  // prior
  RC(lin_bwd(c, h->prior_pz, G(w.pz), D, nullptr, ACT_NONE, G(w.p_b), D, B * 2));
  RC(ln_bwd(c, h->prior_n3, G(w.p_b), w.p_tmp2, w.p_a, G(w.p_tmp2), G(w.p_a), B * 2));
  RC(lin_bwd(c, h->prior_f2, G(w.p_tmp2), D, nullptr, ACT_NONE, G(w.p_ff), FF, B * 2));
  RC(lin_bwd(c, h->prior_f1, G(w.p_ff), FF, w.p_ff, ACT_RELU, G(w.p_a), D, B * 2));
  RC(ln_bwd(c, h->prior_n2, G(w.p_a), w.p_tmp1, w.p_x1, G(w.p_tmp1), nullptr, B * 2));            // the residual input is the constant token rows
  RC(lin_bwd(c, h->prior_o, G(w.p_tmp1), D, nullptr, ACT_NONE, G(w.p_att), D, B * 2));
  attn_bwd(c, h->prior_q, D, w.p_ctxkv, w.p_ctxkv + D, 2 * D, w.mask, G(w.p_att), nullptr, G(w.p_ctxkv), G(w.p_ctxkv) + D, B, 2, WIN, 1);
  RC(lin_bwd(c, h->prior_kv, G(w.p_ctxkv), 2 * D, nullptr, ACT_NONE, dctx, D, M));
  // context encoder
  for (int l = 1; l >= 0; --l) {
    const EncLayer& E = h->enc[l];
    const EncTape& e = w.enc[l];
    float* hin = l == 0 ? w.h0 : w.enc[0].out;
    RC(ln_bwd(c, E.n2, G(e.out), e.tmp2, e.mid, G(e.tmp2), G(e.mid), M));
    RC(lin_bwd(c, E.f2, G(e.tmp2), D, nullptr, ACT_NONE, G(e.ff), FF, M));
    RC(lin_bwd(c, E.f1, G(e.ff), FF, e.ff, ACT_RELU, G(e.mid), D, M));
    RC(ln_bwd(c, E.n1, G(e.mid), e.tmp1, hin, G(e.tmp1), G(hin), M));
    RC(lin_bwd(c, E.o, G(e.tmp1), D, nullptr, ACT_NONE, G(e.att), D, M));
    attn_bwd(c, e.qkv, 3 * D, e.qkv + D, e.qkv + 2 * D, 3 * D, w.mask, G(e.att), G(e.qkv), G(e.qkv) + D, G(e.qkv) + 2 * D, B, WIN, WIN, 0);
    RC(lin_bwd(c, E.qkv, G(e.qkv), 3 * D, nullptr, ACT_NONE, G(hin), D, M));
  }
  (void)ctx;
  RC(lin_bwd(c, h->enc_in, G(w.h0), D, nullptr, ACT_NONE, G(w.x), XLD, M));
  return GLAMR_OK;
}
