// One scene of GLAMR's global optimisation: forward residuals, hand-written reverse pass and Adam, for all iterations of a stage.
//
// Replaces GlobalReconOptimizer.forward / get_pred_trajectory_base / compute_loss / optimize_main
// (global_recon/models/global_recon_model.py:394-570) and the residuals in global_recon/models/loss_func.py for every branch
// the shipped configs reach.  The SMPL call inside the loop is replaced by the rigid
// identity  joints = R_smplx(orient_world) * j_local + trans_world  with j_local cached once per sequence (App. B step 8).
//
// The algorithm is written against a tiny "block runtime" RT (thread id, barrier, block reduction, block prefix sum) so the
// same source is instantiated by the gfx950 kernel (grecon.hip: one workgroup per scene, threads over frames) and by the
// single-threaded host runtime the CPU test-suite uses to check gradients without a GPU (tests/hostsim).  Every loop over
// frames is `for (t = rt.tid(); t < T; t += rt.nthreads())`; phases are separated by rt.sync().
//
// Per-frame dataflow (person p, existing frames e in [0,n), video frames t in [0,T)):
//   A  L[e] = prior[e] + deltas;  dtheta[e] = atan2s(L.sin, L.cos)                    -> scan  theta = cumsum(dtheta)
//   B  d[e] = Rot2D(theta[e-1]) L[e].xy                                             -> scan  xy = cumsum(d)
//   C  q = hq(theta) * R2q(6d->R(L.rot)) * base;  orient_base = q2aa(q);  world_dheading;  d6 = aa2R_k(orient_world)[:, :2]
//   D  camera (own 6D/trans parameters | averaged from persons + residuals | constant)
//   E  cam-frame orientation, joints, projection, all residuals and their per-frame gradients (neighbours read, own written)
//   G  camera gradients -> camera parameters (Adam) or back into the persons' transforms
//   H  reverse of C;  suffix scans of g_xy and g_theta reverse B and A;  Adam on every trajectory variable.
#pragma once
#include <type_traits>
#include "rotmath.hpp"
#include "../../include/glamr_hip.h"

// phase timing (development builds only: -DGLAMR_PHASE_TIMING; tools/grecon_phases.py)
#ifdef GLAMR_PHASE_TIMING
#define GLAMR_MARK_BEGIN(rt) (rt).mark_begin()
#define GLAMR_MARK(rt, k) (rt).template mark<k>()
#define GLAMR_MARK_END(rt) (rt).mark_end()
#else
#define GLAMR_MARK_BEGIN(rt)
#define GLAMR_MARK(rt, k)
#define GLAMR_MARK_END(rt)
#endif

#ifndef GLAMR_HOIST_FRAME_CONSTS
#define GLAMR_HOIST_FRAME_CONSTS 0
#endif
// Persons per scene: 8 in the instances the library normally launches (the scene description sits in static LDS next to a 153 KB arena: no
// room for more), 32 in the second copy of this header that csrc/grecon_wide.hip compiles into namespace grecon_wide for larger scenes.
#ifndef GLAMR_MAX_PERSONS
#define GLAMR_MAX_PERSONS 8
#endif
#ifndef GLAMR_GRECON_NS
#define GLAMR_GRECON_NS grecon
#endif
namespace glamr {
namespace GLAMR_GRECON_NS {
constexpr int MAXP = GLAMR_MAX_PERSONS;

constexpr int NJ = 26;
#ifndef GLAMR_KP_DEPTH
#define GLAMR_KP_DEPTH 3      // keypoint rows requested ahead of the one being processed (grecon_algo.hpp phase E; GLAMR_KP_GROUP=0 builds only)
#endif
#ifndef GLAMR_KP_FOLD
// 1 (default since round 6): a keypoint row costs 43 instead of 52 VALU instructions (joint_nb: fused chains started from q, the robust term's constant
// factors folded into the table's weight column, out-of-range rows through a zero weight) -- stage launch 21.7 -> 20.9 ms (profiles/r06_stage_ab.log).
// The kernel is issue-bound on the SIMD that carries two of a scene's five waves, so its time follows the instruction count.  0 = rounds 4-5's arithmetic.
#define GLAMR_KP_FOLD 1
#endif
#ifndef GLAMR_KP_GROUP
#define GLAMR_KP_GROUP 3      // workspace keypoint rows per group (requested a whole group ahead, processed without branches); 0 = the one-row ring
#endif
constexpr float FPS = 30.0f;

struct PersonConst {
  int fr_start, fr_end;               // exist_frames = [fr_start, fr_end)
  const float* vis;                   // [T]
  int* vis_rank;                      // [T] index among visible frames, -1 if invisible (filled by setup_tables)
  const float* j_local;               // [T][NJ][3]
  const float* kp_2d;                 // [T][NJ][2]
  const float* kp_score;              // [T][NJ]
  const float* cam_K;                 // [T][9]
  const float* prior;                 // [T][11] traj_local_pred, rows [0,n)
  const float* orient_cam;            // [T][3]
  const float* base_orient;           // [T][3]
  const float* base_trans;            // [T][3]
  const float* person2cam;            // [T][12]
  const float* dheading_mask;         // [T] (row e, e >= 1) or null = all zero
  int frozen;                         // person owned by another rank: world pose given in base_orient / base_trans (glamr_scene_batch.frozen)
};

struct PersonState {
  float* p;                           // parameter block of this person (glamr_param_layout offsets): the batch array, or its on-chip copy
  float* p_g;                         // the batch array itself
  float* m; float* v; float* g;       // Adam moments, gradient (same layout)
  float* theta; float* xy;            // [T], [T][2] scan buffers
  float* csn;                         // [T][2] cos, sin of theta[t] (read by the neighbouring frame in phases B and I)
  float* d6;                          // [T][6] first two columns c1, c2 of the world rotation (the third is c1 x c2)
  float* tw;                          // [T][3] world translation
  float* g_d6; float* g_tw;           // [T][6], [T][3] gradients of the above (the third column's gradient folded onto c1, c2)
  float* orient_world; float* trans_world;  // [T][3] outputs, written by the last evaluation
  float* g_theta; float* g_xy;        // [T], [T][2]
  float* kp_2d_pred; float* orient_cam_in_world;   // outputs
  float* g_j_local;                   // [T][NJ][3] output of the last evaluation: dL/d j_local, or null (glamr_scene_batch.g_j_local)
  float* kp_wsum;                     // [NJ] sum over visible frames of thresholded score^2
  float* Lc;                          // [12][T] cached trajectory row of frame t (dx dy z r6[6] h), written in phase A
  float* kpc;                         // [njc][6][T] compact keypoint data of the scored joints: j_local(3) target(2) weight(1)
  float* kpc_ws;                      // same table in the workspace, for the joints jj >= njc_fast that do not fit on chip
  int njc_fast;
  int njc; int jidx[NJ];              // joints whose residual weight is non-zero somewhere
  float* h_prior;                     // [T] row e: heading angle of the prior row, atan2s(sin, cos)  (constant per stage)
  float* oc6;                         // [T][6] first two columns of aa2R_k(orient_cam) (cam_traj_rot target, constant)
  float *in_vis, *in_cam_K, *in_prior, *in_base_orient, *in_base_trans, *in_person2cam, *in_dmask;   // workspace copies of the stage-constant
                                      // inputs the loop reads (constant-layout instances read them at constant offsets)
};

struct Scene {
  int P, T;
  const glamr_param_layout* lay;
  const glamr_stage_desc* st;
  PersonConst pc[MAXP];
  PersonState ps[MAXP];
  const float* rel_cam;               // [P][P][T][12] or null
  int* pair_first;                    // [MAXP][MAXP] first co-visible frame or -1 (setup_tables)
  int* fill_src;                      // [T] frame whose averaged camera a frame without persons inherits (:493-498)
  int* n_vis_persons;                 // [T] fr_num_persons
  float* cam_pose;                    // [T][12]  in/out
  float* cam_inv;                     // [T][12]  camera-to-world of the current evaluation (neighbours read it)
  float* cam_t;                       // [T][3]   translation column of cam_pose (world-to-camera)
  float* g_cam; float* g_caminv;      // [T][12]
  float* g_avg;                       // [T][12] (cam-from-person)
  float* cp; float* cm; float* cv; float* cg;   // scene-level parameter block (camera), moments, gradient
  float* cp_g;                        // the batch array; cp is it, or (on-chip Adam state) a COMPACT 9 T copy: camera parameters when the
                                      // stage optimises them, else the camera residuals shifted down by 9 T
  float* losses;                      // [GLAMR_NUM_LOSSES]
  float* loss_history;                // [niters][GLAMR_NUM_LOSSES] the reported values of every iteration, or null (glamr_scene_batch.loss_history)
  int store_grad;
  int rel_stride_p, rel_stride_t;     // padded person count / frame count of rel_cam
  float* fast_free; size_t fast_left; // unused tail of the on-chip arena (claimed by setup for kpc when it fits)
  int TM;                             // frame count the arena / workspace arrays are laid out for (their strides); >= the batch's max_len
  float* ws;                          // this scene's workspace slice
  const float* adam_tab;              // [niters][2] per-iteration (-lr / (1 - beta1^t), sqrt(1 - beta2^t)) formed on the host, or null
};

// ---- wave-uniform views ------------------------------------------------------------------------------------------------------
// The scene description lives in LDS; a plain read leaves every pointer in two VECTOR registers per lane, and the optimiser hoists
// dozens of them out of the iteration loop.  Everything in it is uniform over the workgroup, so the device build routes the hot
// loop's copies through readfirstlane: they end up in scalar registers.
// They also come out of LDS as GENERIC pointers (flat_load / flat_store with a 64-bit address per lane); the views say which memory
// each array lives in -- workspace and batch arrays are global, the exchange arrays are LDS when the arena is on chip (FAST) --
// so the loop runs on global_load with a scalar base and ds_read.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ T* uni(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
// (the empty asm keeps the optimiser from folding the cast pair away, and pins the value to scalar registers)
template <class T> __device__ __forceinline__ T* glob(T* p) {
  __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)uni(p);
  asm("" : "+s"(g));
  return (T*)g;
}
template <int FAST, class T> __device__ __forceinline__ T* fastp(T* p) {
  if (FAST != 0) {
    __attribute__((address_space(3))) T* g = (__attribute__((address_space(3))) T*)uni(p);
    asm("" : "+s"(g));
    return (T*)g;
  }
  return glob(p);
}
// arrays of the second group are on chip only with a full arena (FAST == 1 or 3); with the lite arena (FAST == 2) they are global
template <int FAST, class T> __device__ __forceinline__ T* fastp2(T* p) { return (FAST == 1 || FAST == 3) ? fastp<1>(p) : glob(p); }
// ... and the own-frame hand-over arrays between the residual phase and the reverse pass (world translation, the two adjoints) also with the MID
// arena (FAST == 4: lite + those three, for scenes of several persons whose full arena does not fit)
template <int FAST, class T> __device__ __forceinline__ T* fastp3(T* p) { return (FAST == 1 || FAST == 3 || FAST == 4) ? fastp<1>(p) : glob(p); }
#else
template <class T> inline T* uni(T* p) { return p; }
inline int uni(int x) { return x; }
template <class T> inline T* glob(T* p) { return p; }
template <int FAST, class T> inline T* fastp(T* p) { return p; }
template <int FAST, class T> inline T* fastp2(T* p) { return p; }
template <int FAST, class T> inline T* fastp3(T* p) { return p; }
#endif

// Frame-loop condition.  The device build also tells the compiler that a frame index is a small non-negative number (the launcher
// rejects sequences longer than GLAMR_GRECON_MAX_FRAMES): `base[t * 6 + k]` then compiles to a global access with the SCALAR base and
// one 32-bit lane offset instead of a 64-bit address pair per array (v_ashrrev + v_lshl_add_u64 and two registers each).
constexpr int GLAMR_GRECON_MAX_FRAMES = 32768;
GLAMR_HD bool frame_in(int t, int T) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(t >= 0);
  __builtin_assume(T <= GLAMR_GRECON_MAX_FRAMES);
#endif
  return t < T;
}

struct PersonView {      // what the iteration loop needs of PersonConst + PersonState, uniform
  int fr_start, fr_end, njc, njc_fast, frozen;      // (a frozen person's existing range is EMPTY here: every frame takes the given pose)
  const float *vis, *j_local, *kp_2d, *kp_score, *cam_K, *prior, *base_orient, *base_trans, *person2cam, *dheading_mask, *h_prior, *oc6, *kp_wsum;
  const int* vis_rank;
  float *p, *m, *v, *g, *theta, *csn, *xy, *d6, *tw, *g_d6, *g_tw, *orient_world, *trans_world, *g_theta, *g_xy, *kp_2d_pred, *orient_cam_in_world, *Lc, *g_j_local;
  const float *kpc, *kpc_ws;
};
struct SceneView {
  const float* rel_cam; const int* pair_first; const int* fill_src; const int* n_vis_persons;
  float *cam_pose, *cam_inv, *cam_t, *g_cam, *g_caminv, *g_avg, *cp, *cpg, *cm, *cv, *cg, *losses;
  int store_grad, rel_stride_p, rel_stride_t, TM;
};
// ---- small helpers ---------------------------------------------------------------------------------------------------

GLAMR_HD float gmof(float x, float sigma2) { return sigma2 * x * x / (sigma2 + x * x); }
GLAMR_HD float gmof_d(float x, float sigma2) { const float r = rm::rcp_(sigma2 + x * x); return 2.0f * sigma2 * sigma2 * x * r * r; }

struct AdamCoef {
  float neg_step; float bc2_sqrt;    // -lr / (1 - beta1^t),  sqrt(1 - beta2^t)   (both formed in double, as Python does)
  float inv_bc2_sqrt;                // RN(1 / bc2_sqrt): see adam()
  GLAMR_HD void finish() { inv_bc2_sqrt = rm::div_(1.0f, bc2_sqrt); }
};
constexpr int ADAM_TAB_MAX = 4096;
// torch/optim/adam.py: bias_correction = 1 - beta ** step (Python float pow), step_size = lr / bias_correction1,
// bias_correction2_sqrt = bias_correction2 ** 0.5; both reach the fp32 kernels as scalars rounded from double.  Host only (libm pow).
inline void adam_coef_host(double lr, int step, float out[2]) {
  const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
  out[0] = (float)(-(lr / bc1));
  out[1] = (float)pow(bc2, 0.5);
}

// torch.optim.Adam, single-tensor path (betas 0.9/0.999, eps 1e-8, no weight decay), in the operation ORDER of its CPU kernels:
//   exp_avg.lerp_(grad, 1 - beta1)                          -> fma(w, g - m, m)               (vectorised lerp is an fmadd)
//   exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)  -> fma((1 - beta2) g, g, beta2 v)
//   denom = exp_avg_sq.sqrt() / sqrt(bias_correction2) + eps;   param.addcdiv_(exp_avg, denom, value=-step_size) -> p + (a m) / denom
// with 1 - beta formed in DOUBLE and rounded to fp32 (0.1f, 0.001f -- not 1.0f - 0.9f).  Checked bit for bit against torch.optim.Adam
// on the CPU runtime (tests/test_rotmath_grads.py); torch's own sqrt is 1 ulp off for 0.7 % of its arguments, which no one can follow.
GLAMR_HD void adam(float& p, float& m, float& v, float g, const AdamCoef& c) {
  m = rm::fma_(0.1f, g - m, m);
  v = rm::fma_(0.001f * g, g, v * 0.999f);
  // sqrt(v) / bc2_sqrt: the divisor is the same for every parameter of an iteration, so its correctly rounded reciprocal y is formed once
  // (AdamCoef::finish) and the correctly rounded quotient costs three operations per parameter instead of a reciprocal and seven:
  //     q = RN(a y),  r = a - b q (exact in an fma),  RN(q + r y) = RN(a / b)       (Markstein 1990; no overflow / underflow here: a is 0
  // or >= 3.7e-23, 0.03 < b <= 1).  Checked against IEEE division bit for bit (tests/test_adam_exact.py: host runtime and device).
  const float a = rm::sqrt_rn_(v);
  const float q = a * c.inv_bc2_sqrt;
  const float denom = rm::fma_(rm::fma_(-c.bc2_sqrt, q, a), c.inv_bc2_sqrt, q) + 1e-8f;
  p = p + rm::div_(c.neg_step * m, denom);
}

// N updates of one frame.  GLAMR_ADAM_INTERLEAVE (development aid, OFF): stage by stage across the N parameters, the scheduler barred from moving
// anything across a stage boundary, so that every instruction of the ~30-long dependent chain of an update (through a square root and a
// reciprocal) has N - 1 independent ones between itself and its consumer.  Measured on the MI355X (profiles/r05_stage_ab.log): 11.15 against
// 10.8 us per iteration for the plain loop -- the scheduling barriers also pin the loads and stores around the updates, which costs more than
// the shorter chains give.  Operation for operation what adam() does either way: same bits (tests/test_adam_exact.py, tools/stage_bits.py).
#if defined(GLAMR_ADAM_INTERLEAVE) && GLAMR_ADAM_INTERLEAVE == 2      // the staged form WITHOUT scheduling barriers between the stages
#define GLAMR_ADAM_STAGE_FENCE() ((void)0)
#else
#define GLAMR_ADAM_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
template <int N>
GLAMR_HD void adam_n(float (&P)[N], float (&M)[N], float (&V)[N], const float (&g)[N], const AdamCoef& c) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GLAMR_ADAM_INTERLEAVE)
  float s[N], den[N], num[N], y[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    M[k] = rm::fma_(0.1f, g[k] - M[k], M[k]);
    V[k] = rm::fma_(0.001f * g[k], g[k], V[k] * 0.999f);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) s[k] = rm::hw_sqrt_(V[k]);
  GLAMR_ADAM_STAGE_FENCE();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float a = rm::sqrt_rn_fix_(V[k], s[k]);
    const float q = a * c.inv_bc2_sqrt;
    den[k] = rm::fma_(rm::fma_(-c.bc2_sqrt, q, a), c.inv_bc2_sqrt, q) + 1e-8f;
    num[k] = c.neg_step * M[k];
  }
  GLAMR_ADAM_STAGE_FENCE();
#pragma unroll
  for (int k = 0; k < N; ++k) y[k] = rm::hw_rcp_(den[k]);
  GLAMR_ADAM_STAGE_FENCE();
#pragma unroll
  for (int k = 0; k < N; ++k) P[k] = P[k] + rm::div_fix_(num[k], den[k], y[k]);
#else
  for (int k = 0; k < N; ++k) adam(P[k], M[k], V[k], g[k], c);
#endif
}

GLAMR_HD void invert34(const float M[12], float O[12]) {       // [R|t] -> [R^T | -R^T t]   (inverse_transform)
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) O[i * 4 + j] = M[j * 4 + i];
    O[i * 4 + 3] = -(M[0 * 4 + 3] * M[0 * 4 + i] + M[1 * 4 + 3] * M[1 * 4 + i] + M[2 * 4 + 3] * M[2 * 4 + i]);
  }
}
// O = invert(M):  gM += d(O)/d(M)^T gO
GLAMR_HD void invert34_bwd(const float M[12], const float gO[12], float gM[12]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) gM[j * 4 + i] += gO[i * 4 + j];
  for (int i = 0; i < 3; ++i) {            // O_i3 = -sum_k t_k R_ki
    const float g = gO[i * 4 + 3];
    for (int k = 0; k < 3; ++k) {
      gM[k * 4 + 3] += -g * M[k * 4 + i];
      gM[k * 4 + i] += -g * M[k * 4 + 3];
    }
  }
}
GLAMR_HD void get_R(const float M[12], float R[9]) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] = M[i * 4 + j]; }
// C = A B for 3x4 rigid transforms (implicit last row 0 0 0 1)
GLAMR_HD void mul34(const float A[12], const float B[12], float C[12]) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) C[i * 4 + j] = A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j] + A[i * 4 + 2] * B[2 * 4 + j];
    C[i * 4 + 3] += A[i * 4 + 3];
  }
}
GLAMR_HD void mul34_bwd(const float A[12], const float B[12], const float gC[12], float* gA, float* gB) {
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      if (gA) { float s = 0.f; for (int j = 0; j < 4; ++j) s += gC[i * 4 + j] * B[k * 4 + j]; gA[i * 4 + k] += s; }
    }
  if (gA) for (int i = 0; i < 3; ++i) gA[i * 4 + 3] += gC[i * 4 + 3];
  if (gB)
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 4; ++j) { float s = 0.f; for (int i = 0; i < 3; ++i) s += A[i * 4 + k] * gC[i * 4 + j]; gB[k * 4 + j] += s; }
}

// person trajectory row L[e] from the prior and the optimisation deltas (get_pred_trajectory_base :394-419)
struct LocalRow { float dx, dy, z, r6[6], h; };   // h = heading angle after adding the delta; (cos h, sin h) replaces cols 9,10
GLAMR_HD LocalRow local_row(const PersonView& c, const glamr_param_layout& l, int e) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(e >= 0);
#endif
  const float* pr = c.prior + (size_t)e * 11;
  const float* P = c.p;
  // Every load is issued before the first use and none sits behind a branch.  Frame 0 reads other parameters than the rest (local_xy /
  // local_heading instead of its rows of local_dxy / local_dheading): selected INDICES, not selected code -- the wave that holds frame 0 used
  // to walk both sides of the branch, three dependent trips to the workspace on the critical path of every iteration.  Same operations per
  // frame as before (the heading of frame 0 is h_prior + p, of the others h_prior + p * mask).
  const bool first = e == 0;
  const int ixy = first ? l.local_xy : l.local_dxy + e * 2;
  const int ih = first ? l.local_heading : l.local_dheading + e;
  float prv[9], p6[6];
  for (int k = 0; k < 9; ++k) prv[k] = pr[k];
  const float hp = c.h_prior[e];
  const float mk = c.dheading_mask ? c.dheading_mask[e] : 0.0f;
  const float px = P[ixy], py = P[ixy + 1], ph = P[ih], pz = P[l.local_z + e];
  for (int k = 0; k < 6; ++k) p6[k] = P[l.local_rot + e * 6 + k];
  LocalRow L;
  L.dx = prv[0] + px;
  L.dy = prv[1] + py;
  const float h0 = hp + ph, h1 = hp + (c.dheading_mask ? ph * mk : 0.0f);
  L.h = first ? h0 : h1;
  L.z = prv[2] + pz;
  for (int k = 0; k < 6; ++k) L.r6[k] = prv[3 + k] + p6[k];
  return L;
}

// the same row from values the thread keeps in registers for the whole launch (FrameRegs): same operations in the same order
GLAMR_HD LocalRow local_row_regs(const float (&pr)[9], float h_prior, float dmask, bool has_dmask, const float* P, const glamr_param_layout& l, int e) {
  LocalRow L;
  if (e == 0) {
    L.dx = pr[0] + P[l.local_xy + 0];
    L.dy = pr[1] + P[l.local_xy + 1];
    L.h = h_prior + P[l.local_heading];
  } else {
    L.dx = pr[0] + P[l.local_dxy + e * 2 + 0];
    L.dy = pr[1] + P[l.local_dxy + e * 2 + 1];
    L.h = h_prior + (has_dmask ? P[l.local_dheading + e] * dmask : 0.0f);
  }
  L.z = pr[2] + P[l.local_z + e];
  for (int k = 0; k < 6; ++k) L.r6[k] = pr[3 + k] + P[l.local_rot + e * 6 + k];
  return L;
}

GLAMR_HD void store_row(float* Lc, int TM, int t, const LocalRow& L) {
  Lc[0 * TM + t] = L.dx; Lc[1 * TM + t] = L.dy; Lc[2 * TM + t] = L.z;
  for (int k = 0; k < 6; ++k) Lc[(3 + k) * TM + t] = L.r6[k];
  Lc[9 * TM + t] = L.h;
}
GLAMR_HD LocalRow load_row(const float* Lc, int TM, int t) {
  LocalRow L;
  L.dx = Lc[0 * TM + t]; L.dy = Lc[1 * TM + t]; L.z = Lc[2 * TM + t];
  for (int k = 0; k < 6; ++k) L.r6[k] = Lc[(3 + k) * TM + t];
  L.h = Lc[9 * TM + t];
  return L;
}

// Adam on n consecutive parameters: all loads first (one exposed memory latency instead of n), then the arithmetic, then the stores
template <int N>
GLAMR_HD void adam_block(float* p, float* m, float* v, float* gstore, int base, const float (&g)[N], const AdamCoef& c) {
  float P[N], M[N], V[N];
  for (int k = 0; k < N; ++k) { P[k] = p[base + k]; M[k] = m[base + k]; V[k] = v[base + k]; }
  adam_n(P, M, V, g, c);
  for (int k = 0; k < N; ++k) { p[base + k] = P[k]; m[base + k] = M[k]; v[base + k] = V[k]; }
  if (gstore) for (int k = 0; k < N; ++k) gstore[base + k] = g[k];
}

// Adam on N consecutive parameters whose state is fetched EARLY (top of a phase) and consumed late: with one or two waves per SIMD
// nothing else hides the workspace latency, so every phase issues all its loads first.
template <int N>
struct AdamRegs {
  float P[N], M[N], V[N];
  GLAMR_HD void load(const float* p, const float* m, const float* v, int base) {
    for (int k = 0; k < N; ++k) { P[k] = p[base + k]; M[k] = m[base + k]; V[k] = v[base + k]; }
  }
  GLAMR_HD void zero() { for (int k = 0; k < N; ++k) { P[k] = 0.f; M[k] = 0.f; V[k] = 0.f; } }
  GLAMR_HD void store(float* p, float* m, float* v, float* gstore, int base, const float (&g)[N]) const {
    for (int k = 0; k < N; ++k) { p[base + k] = P[k]; m[base + k] = M[k]; v[base + k] = V[k]; }
    if (gstore) for (int k = 0; k < N; ++k) gstore[base + k] = g[k];
  }
  GLAMR_HD void step_store(float* p, float* m, float* v, float* gstore, int base, const float (&g)[N], const AdamCoef& c) {
    adam_n(P, M, V, g, c);
    store(p, m, v, gstore, base, g);
  }
};
// the updates of two / three parameter groups of a frame as ONE interleaved pass (adam_n); a group that is not updated takes part with
// whatever its registers hold (zeros: AdamRegs::zero) and is simply not stored by the caller
template <int A, int B>
GLAMR_HD void adam_step2(AdamRegs<A>& a, const float (&ga)[A], AdamRegs<B>& b, const float (&gb)[B], const AdamCoef& c) {
  float P[A + B], M[A + B], V[A + B], g[A + B];
  for (int k = 0; k < A; ++k) { P[k] = a.P[k]; M[k] = a.M[k]; V[k] = a.V[k]; g[k] = ga[k]; }
  for (int k = 0; k < B; ++k) { P[A + k] = b.P[k]; M[A + k] = b.M[k]; V[A + k] = b.V[k]; g[A + k] = gb[k]; }
  adam_n(P, M, V, g, c);
  for (int k = 0; k < A; ++k) { a.P[k] = P[k]; a.M[k] = M[k]; a.V[k] = V[k]; }
  for (int k = 0; k < B; ++k) { b.P[k] = P[A + k]; b.M[k] = M[A + k]; b.V[k] = V[A + k]; }
}
template <int A, int B, int C>
GLAMR_HD void adam_step3(AdamRegs<A>& a, const float (&ga)[A], AdamRegs<B>& b, const float (&gb)[B], AdamRegs<C>& cc, const float (&gc)[C], const AdamCoef& c) {
  float P[A + B + C], M[A + B + C], V[A + B + C], g[A + B + C];
  for (int k = 0; k < A; ++k) { P[k] = a.P[k]; M[k] = a.M[k]; V[k] = a.V[k]; g[k] = ga[k]; }
  for (int k = 0; k < B; ++k) { P[A + k] = b.P[k]; M[A + k] = b.M[k]; V[A + k] = b.V[k]; g[A + k] = gb[k]; }
  for (int k = 0; k < C; ++k) { P[A + B + k] = cc.P[k]; M[A + B + k] = cc.M[k]; V[A + B + k] = cc.V[k]; g[A + B + k] = gc[k]; }
  adam_n(P, M, V, g, c);
  for (int k = 0; k < A; ++k) { a.P[k] = P[k]; a.M[k] = M[k]; a.V[k] = V[k]; }
  for (int k = 0; k < B; ++k) { b.P[k] = P[A + k]; b.M[k] = M[A + k]; b.V[k] = V[A + k]; }
  for (int k = 0; k < C; ++k) { cc.P[k] = P[A + B + k]; cc.M[k] = M[A + B + k]; cc.V[k] = V[A + B + k]; }
}

// World orientation as a rotation MATRIX.  The reference composes quaternions and goes through axis-angle at every step
// (traj_local2global_heading: hq(theta) (x) R2q(6d->R(r6)) (x) (.5,.5,.5,.5), :459-465 world heading offset, then angle_axis ->
// rotation matrix again for the joints, :517 and the losses); as rotations these are
//     R_w = Rz(theta + world_dheading) . [b2 b3 b1],      (b1, b2, b3) = columns of 6d->R(r6),
// because (.5,.5,.5,.5) is the cyclic permutation x->y->z->x.  Working on the matrix removes every atan2/sqrt/sin/cos of the
// conversions from the iteration; values and gradients agree with the chain up to rounding (all Jacobians along the chain map
// onto the tangent space of SO(3), where R(aa(R)) is the identity).  Only the first two columns (c1, c2) are kept: c3 = c1 x c2.
GLAMR_HD void rotz2(float cs, float sn, const float a[3], float o[3]) { o[0] = cs * a[0] - sn * a[1]; o[1] = sn * a[0] + cs * a[1]; o[2] = a[2]; }
GLAMR_HD void rotz2T(float cs, float sn, const float a[3], float o[3]) { o[0] = cs * a[0] + sn * a[1]; o[1] = -sn * a[0] + cs * a[1]; o[2] = a[2]; }
GLAMR_HD void cols_to_R(const float* d6, float R[9]) {
  const float c1[3] = {d6[0], d6[1], d6[2]}, c2[3] = {d6[3], d6[4], d6[5]};
  float c3[3];
  rm::cross3(c1, c2, c3);
  for (int r = 0; r < 3; ++r) { R[r * 3 + 0] = c1[r]; R[r * 3 + 1] = c2[r]; R[r * 3 + 2] = c3[r]; }
}
// gradient of a full 3x3 matrix -> gradient of (c1, c2) with c3 = c1 x c2 folded in
GLAMR_HD void fold_R_grad(const float* d6, const float gR[9], float g6[6]) {
  const float c1[3] = {d6[0], d6[1], d6[2]}, c2[3] = {d6[3], d6[4], d6[5]};
  const float g3[3] = {gR[2], gR[5], gR[8]};
  float a[3], b[3];
  rm::cross3(c2, g3, a);          // d(c1 x c2)/dc1 ^T g3
  rm::cross3(g3, c1, b);          // d(c1 x c2)/dc2 ^T g3
  for (int r = 0; r < 3; ++r) { g6[r] = gR[r * 3 + 0] + a[r]; g6[3 + r] = gR[r * 3 + 1] + b[r]; }
}


// ---- layout of the parameter vector and of the per-scene workspace ------------------------------------------------------

GLAMR_HD constexpr glamr_param_layout make_layout(int max_persons, int max_len) {
  glamr_param_layout l{};
  const int T = max_len;
  l.cam_rot6d = 0;
  l.cam_trans = l.cam_rot6d + 6 * T;
  l.cam_inv_rot_res = l.cam_trans + 3 * T;
  l.cam_inv_trans_res = l.cam_inv_rot_res + 6 * T;
  l.person0 = l.cam_inv_trans_res + 3 * T;
  l.local_xy = 0;
  l.local_heading = 2;
  l.local_dxy = 4;
  l.local_dheading = l.local_dxy + 2 * T;
  l.local_z = l.local_dheading + T;
  l.local_rot = l.local_z + T;
  l.world_dheading = l.local_rot + 6 * T;
  l.person_stride = l.world_dheading + T;
  l.scene_stride = l.person0 + max_persons * l.person_stride;
  return l;
}
GLAMR_HD void param_layout(int max_persons, int max_len, glamr_param_layout& l) { l = make_layout(max_persons, max_len); }

// Where every per-scene array lives: float offsets into the on-chip arena (lds) or into the scene's workspace slice.  One function
// for the binder (run time), the size queries, and -- with a frame count known at compile time -- the constant addresses of the
// constant-layout instances.  `fast`: an arena exists; `fast_mode` 1 = full (+ parameters and Adam moments of single-person scenes),
// 2 = lite, 3 = full with the Adam state in the workspace (a smaller arena: several workgroups of short sequences share a CU),
// 4 = mid: lite + world translation and the two adjoint hand-over arrays (26 instead of 14 floats per person-frame: what 3 - 4 persons x 300 frames
// leave room for; round 6).
struct ArrOff { unsigned off; bool lds; };
struct PersonOff {
  ArrOff m_ws, v_ws, m, v, p, g_ws, theta, csn, xy, d6, tw, g_d6, g_tw, g_theta, g_xy, Lc, kpc_ws, vis_rank, kp_wsum, h_prior, oc6;
  ArrOff in_vis, in_cam_K, in_prior, in_base_orient, in_base_trans, in_person2cam, in_dmask;      // stage-constant inputs, copied next to the rest
};
struct SceneOff {
  ArrOff cm_ws, cv_ws, cm, cv, cp, cg_ws, cam_inv, cam_t, g_cam, g_caminv, g_avg, fill_src, n_vis_persons, pair_first;
  PersonOff ps[MAXP];
  unsigned ws_end, fast_end;
  bool adam_fast;
};
GLAMR_HD constexpr SceneOff scene_offsets(int max_persons, int max_len, bool fast, int fast_mode) {
  SceneOff o{};
  const glamr_param_layout l = make_layout(max_persons, max_len);
  const unsigned TM = (unsigned)max_len, person0 = (unsigned)l.person0, pstride = (unsigned)l.person_stride;
  unsigned w = 0, f = 0;
  // arrays other threads read (neighbouring frames, prefix sums) go to the arena when there is one; the second group (own-frame
  // hand-over arrays) only to the full arena; parameters and Adam moments only for single-person scenes with the full arena
  const bool full = fast && (fast_mode == 1 || fast_mode == 3);
  o.adam_fast = fast && fast_mode == 1 && max_persons == 1;
  const bool af = o.adam_fast;
  auto take = [&](unsigned n) { ArrOff r{w, false}; w += n; return r; };
  auto lds = [&](unsigned n) { ArrOff r{f, true}; f += n; return r; };
  auto takef = [&](unsigned n) { return fast ? lds(n) : take(n); };
  auto takef2 = [&](unsigned n) { return full ? lds(n) : take(n); };
  const bool mid = fast && fast_mode == 4;      // lite + the three hand-over arrays a frame's own thread reads back after a barrier (12 floats per person-frame)
  auto takef3 = [&](unsigned n) { return (full || mid) ? lds(n) : take(n); };
  // the moments of the camera block use a COMPACT index: camera parameters i, camera residuals i - 9 T
  o.cm_ws = take(person0); o.cv_ws = take(person0);
  o.cm = af ? lds(9 * TM) : o.cm_ws; o.cv = af ? lds(9 * TM) : o.cv_ws;
  if (af) o.cp = lds(9 * TM);
  o.cg_ws = take(person0);
  o.cam_inv = takef(12 * TM); o.cam_t = takef(3 * TM); o.g_cam = take(12 * TM); o.g_caminv = take(12 * TM); o.g_avg = take(12 * TM);
  o.fill_src = take(TM); o.n_vis_persons = take(TM); o.pair_first = take(MAXP * MAXP);
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {        // (fully unrolled: o.ps[p] is then addressed statically and the struct stays in registers)
    if (p >= max_persons) break;
    PersonOff& s = o.ps[p];
    s.m_ws = take(pstride); s.v_ws = take(pstride);
    s.m = af ? lds(pstride) : s.m_ws; s.v = af ? lds(pstride) : s.v_ws;
    if (af) s.p = lds(pstride);
    s.g_ws = take(pstride);
    s.theta = takef(TM); s.csn = takef(2 * TM); s.xy = takef(2 * TM); s.d6 = takef(6 * TM); s.tw = takef3(3 * TM); s.g_d6 = takef3(6 * TM); s.g_tw = takef3(3 * TM);
    s.g_theta = takef(TM); s.g_xy = takef(2 * TM); s.Lc = takef2(12 * TM);
    s.kpc_ws = take(26 * 6 * TM);
    s.vis_rank = take(TM); s.kp_wsum = take(32); s.h_prior = take(TM); s.oc6 = take(6 * TM);
    s.in_vis = take(TM); s.in_cam_K = take(9 * TM); s.in_prior = take(11 * TM); s.in_base_orient = take(3 * TM); s.in_base_trans = take(3 * TM);
    s.in_person2cam = take(12 * TM); s.in_dmask = take(TM);
  }
  o.ws_end = w; o.fast_end = f;
  return o;
}
// on-chip arena: mode 1 (full) = every exchange / hand-over array; mode 2 (lite) = only the arrays other threads read (prefix sums,
// cos/sin, world rotation columns, camera) -- 14 instead of 38 floats per person-frame, for scenes with many persons or frames.
// Single-person scenes with the full arena also keep the parameters and the Adam moments p, m, v on chip (read and written by every
// iteration: the 58 KB an iteration used to write through to memory; they go back to the batch array once, when the stage ends): the
// camera block compacted to 9 T (the camera parameters and the camera residuals are never optimised together and share it) + the
// person block (11 T + 4), three times.
GLAMR_HD size_t scene_fast_floats(int max_persons, int max_len, int mode = 1) { return scene_offsets(max_persons, max_len, true, mode).fast_end; }
GLAMR_HD size_t scene_workspace_floats(int max_persons, int max_len) { return scene_offsets(max_persons, max_len, false, 0).ws_end; }
// Constant-layout instances: single-person scenes of 257..304 frames (BASELINE configs[1]: 300) lay their arena and workspace out for
// 304 frames whatever the batch's padded length, so that every array address in the iteration loop is a compile-time constant
// (304 and not 320: the arena then still has room for two rows of the keypoint table)
constexpr int GLAMR_CONST_LAYOUT_FRAMES = 304;
GLAMR_HD constexpr int layout_frames(int max_persons, int max_len) {
  return (max_persons == 1 && max_len > 256 && max_len <= GLAMR_CONST_LAYOUT_FRAMES) ? GLAMR_CONST_LAYOUT_FRAMES : max_len;
}

// ---- view builders --------------------------------------------------------------------------------------------------------
// TMC > 0 (constant-layout instances: one person, full arena, Adam state on chip): every arena array is the arena's base -- a link-time
// constant -- plus a compile-time offset, every workspace array the workspace base plus a compile-time offset, and the stage-constant
// inputs are read from their workspace copies: nothing is fetched from the scene description and no pointer occupies a register.
template <int FAST, bool AF = false, int TMC = 0, class RT>
GLAMR_HD PersonView person_view(RT& rt, const Scene& sc, int p) {
  const PersonConst& c = sc.pc[p];
  const PersonState& s = sc.ps[p];
  PersonView w;
  w.frozen = TMC > 0 ? 0 : uni(c.frozen);
  w.fr_start = w.frozen ? 0 : uni(c.fr_start); w.fr_end = w.frozen ? 0 : uni(c.fr_end); w.njc = uni(s.njc); w.njc_fast = uni(s.njc_fast);
  w.j_local = glob(c.j_local); w.kp_2d = glob(c.kp_2d); w.kp_score = glob(c.kp_score);      // last evaluation only
  w.orient_world = glob(s.orient_world); w.trans_world = glob(s.trans_world); w.kp_2d_pred = glob(s.kp_2d_pred); w.orient_cam_in_world = glob(s.orient_cam_in_world);
  w.g_j_local = glob(s.g_j_local);
  if constexpr (TMC > 0) {
    constexpr SceneOff o = scene_offsets(1, TMC, true, 1);
    constexpr PersonOff q = o.ps[0];
    float* const A = rt.arena();
    float* const W = rt.workspace();
    w.vis = W + q.in_vis.off; w.cam_K = W + q.in_cam_K.off; w.prior = W + q.in_prior.off; w.base_orient = W + q.in_base_orient.off;
    w.base_trans = W + q.in_base_trans.off; w.person2cam = W + q.in_person2cam.off; w.dheading_mask = W + q.in_dmask.off;
    w.h_prior = W + q.h_prior.off; w.oc6 = W + q.oc6.off; w.kp_wsum = W + q.kp_wsum.off; w.vis_rank = reinterpret_cast<const int*>(W + q.vis_rank.off);
    w.g = nullptr;                                                        // gradients are not recorded by these instances
    w.p = A + q.p.off; w.m = A + q.m.off; w.v = A + q.v.off;
    w.theta = A + q.theta.off; w.csn = A + q.csn.off; w.xy = A + q.xy.off; w.d6 = A + q.d6.off; w.g_theta = A + q.g_theta.off; w.g_xy = A + q.g_xy.off;
    w.tw = A + q.tw.off; w.g_d6 = A + q.g_d6.off; w.g_tw = A + q.g_tw.off; w.Lc = A + q.Lc.off;
    w.kpc = A + o.fast_end; w.kpc_ws = W + q.kpc_ws.off;
    return w;
  } else {
    w.vis = glob(c.vis); w.cam_K = glob(c.cam_K); w.prior = glob(c.prior);
    w.base_orient = glob(c.base_orient); w.base_trans = glob(c.base_trans); w.person2cam = glob(c.person2cam); w.dheading_mask = glob(c.dheading_mask);
    w.h_prior = glob(s.h_prior); w.oc6 = glob(s.oc6); w.kp_wsum = glob(s.kp_wsum); w.vis_rank = glob(c.vis_rank);
    w.g = glob(s.g);
    // parameters + Adam moments: in the on-chip arena for single-person scenes with the full arena (AF), else batch array / workspace
    if (AF) { w.p = fastp<1>(s.p); w.m = fastp<1>(s.m); w.v = fastp<1>(s.v); } else { w.p = glob(s.p); w.m = glob(s.m); w.v = glob(s.v); }
    // first group (prefix sums, neighbour reads): on chip with either arena; second group (own-frame hand-over arrays): full arena only
    w.theta = fastp<FAST>(s.theta); w.csn = fastp<FAST>(s.csn); w.xy = fastp<FAST>(s.xy); w.d6 = fastp<FAST>(s.d6); w.g_theta = fastp<FAST>(s.g_theta); w.g_xy = fastp<FAST>(s.g_xy);
    w.tw = fastp3<FAST>(s.tw); w.g_d6 = fastp3<FAST>(s.g_d6); w.g_tw = fastp3<FAST>(s.g_tw); w.Lc = fastp2<FAST>(s.Lc);
    w.kpc = fastp<FAST>(s.kpc); w.kpc_ws = glob(s.kpc_ws);      // kpc is only dereferenced for the njc_fast joints that are on chip
    return w;
  }
}
template <int FAST, bool AF = false, int TMC = 0, class RT>
GLAMR_HD SceneView scene_view(RT& rt, const Scene& sc) {
  SceneView w;
  w.rel_cam = glob(sc.rel_cam); w.cam_pose = glob(sc.cam_pose); w.cpg = glob(sc.cp_g); w.losses = glob(sc.losses);
  w.rel_stride_p = uni(sc.rel_stride_p); w.rel_stride_t = uni(sc.rel_stride_t);
  if constexpr (TMC > 0) {
    constexpr SceneOff o = scene_offsets(1, TMC, true, 1);
    float* const A = rt.arena();
    float* const W = rt.workspace();
    w.pair_first = reinterpret_cast<const int*>(W + o.pair_first.off); w.fill_src = reinterpret_cast<const int*>(W + o.fill_src.off);
    w.n_vis_persons = reinterpret_cast<const int*>(W + o.n_vis_persons.off);
    w.cam_inv = A + o.cam_inv.off; w.cam_t = A + o.cam_t.off; w.g_cam = W + o.g_cam.off; w.g_caminv = W + o.g_caminv.off; w.g_avg = W + o.g_avg.off;
    w.cp = A + o.cp.off; w.cm = A + o.cm.off; w.cv = A + o.cv.off;
    w.cg = nullptr; w.store_grad = 0; w.TM = TMC;
    return w;
  } else {
    w.pair_first = glob(sc.pair_first); w.fill_src = glob(sc.fill_src); w.n_vis_persons = glob(sc.n_vis_persons);
    w.cam_inv = fastp<FAST>(sc.cam_inv); w.cam_t = fastp<FAST>(sc.cam_t); w.g_cam = glob(sc.g_cam); w.g_caminv = glob(sc.g_caminv); w.g_avg = glob(sc.g_avg);
    w.cg = glob(sc.cg);
    if (AF) { w.cp = fastp<1>(sc.cp); w.cm = fastp<1>(sc.cm); w.cv = fastp<1>(sc.cv); } else { w.cp = glob(sc.cp); w.cm = glob(sc.cm); w.cv = glob(sc.cv); }
    w.store_grad = uni(sc.store_grad); w.TM = uni(sc.TM);
    return w;
  }
}

// Binds scene `si` of the batch to pointers (no computation).  `ws` = this scene's workspace slice.
GLAMR_HD void assemble_scene(const glamr_scene_batch& b, const glamr_param_layout& l, const glamr_stage_desc* st, int si,
                             int n_persons, int seq_len, float* ws, float* grads_out, Scene& sc, float* fast = nullptr,
                             size_t fast_floats = 0, int fast_mode = 1, int layout_len = 0) {
  const size_t TM = (size_t)b.max_len;                                   // strides of the batch arrays
  const int TML = layout_len > 0 ? layout_len : b.max_len;               // strides of the arena / workspace arrays
  const SceneOff o = scene_offsets(b.max_persons, TML, fast != nullptr, fast_mode);
  auto at = [&](const ArrOff& a) { return (a.lds ? fast : ws) + a.off; };
  auto ati = [&](const ArrOff& a) { return reinterpret_cast<int*>(at(a)); };
  sc.P = n_persons; sc.T = seq_len; sc.lay = &l; sc.st = st;
  sc.rel_cam = b.rel_transform_cam ? b.rel_transform_cam + (size_t)si * b.max_persons * b.max_persons * TM * 12 : nullptr;
  sc.cam_pose = b.cam_pose + (size_t)si * TM * 12;
  sc.cp_g = b.params + (size_t)si * l.scene_stride;
  sc.cp = o.adam_fast ? at(o.cp) : sc.cp_g;
  sc.losses = b.losses + (size_t)si * GLAMR_NUM_LOSSES;
  sc.loss_history = (b.loss_history && st->niters > 0) ? b.loss_history + (size_t)si * st->niters * GLAMR_NUM_LOSSES : nullptr;
  sc.store_grad = grads_out != nullptr;
  sc.cm = at(o.cm); sc.cv = at(o.cv);
  sc.cg = grads_out ? grads_out + (size_t)si * l.scene_stride : at(o.cg_ws);
  sc.cam_inv = at(o.cam_inv); sc.cam_t = at(o.cam_t); sc.g_cam = at(o.g_cam); sc.g_caminv = at(o.g_caminv); sc.g_avg = at(o.g_avg);
  sc.fill_src = ati(o.fill_src); sc.n_vis_persons = ati(o.n_vis_persons); sc.pair_first = ati(o.pair_first);
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    if (p >= b.max_persons) break;
    const size_t slot = (size_t)si * b.max_persons + p;
    const PersonOff& q = o.ps[p];
    PersonConst& c = sc.pc[p];
    PersonState& s = sc.ps[p];
    s.p_g = sc.cp_g + l.person0 + (size_t)p * l.person_stride;
    s.p = o.adam_fast ? at(q.p) : s.p_g;
    s.m = at(q.m); s.v = at(q.v);
    s.g = grads_out ? sc.cg + l.person0 + (size_t)p * l.person_stride : at(q.g_ws);
    s.theta = at(q.theta); s.csn = at(q.csn); s.xy = at(q.xy); s.d6 = at(q.d6); s.tw = at(q.tw); s.g_d6 = at(q.g_d6); s.g_tw = at(q.g_tw);
    s.g_theta = at(q.g_theta); s.g_xy = at(q.g_xy); s.Lc = at(q.Lc);
    s.kpc_ws = at(q.kpc_ws); s.kpc = s.kpc_ws; s.njc = 0; s.njc_fast = 0;
    c.vis_rank = ati(q.vis_rank);
    s.kp_wsum = at(q.kp_wsum);
    s.h_prior = at(q.h_prior); s.oc6 = at(q.oc6);
    s.in_vis = at(q.in_vis); s.in_cam_K = at(q.in_cam_K); s.in_prior = at(q.in_prior); s.in_base_orient = at(q.in_base_orient);
    s.in_base_trans = at(q.in_base_trans); s.in_person2cam = at(q.in_person2cam); s.in_dmask = at(q.in_dmask);
    c.fr_start = b.fr_start[slot]; c.fr_end = b.fr_end[slot];
    c.vis = b.vis + slot * TM;
    c.j_local = b.j_local + slot * TM * NJ * 3;
    c.kp_2d = b.kp_2d + slot * TM * NJ * 2;
    c.kp_score = b.kp_score + slot * TM * NJ;
    c.cam_K = b.cam_K + slot * TM * 9;
    c.prior = b.traj_local_pred + slot * TM * 11;
    c.orient_cam = b.orient_cam + slot * TM * 3;
    c.base_orient = b.base_orient + slot * TM * 3;
    c.base_trans = b.base_trans + slot * TM * 3;
    c.person2cam = b.person2cam + slot * TM * 12;
    c.dheading_mask = b.dheading_mask ? b.dheading_mask + slot * TM : nullptr;
    c.frozen = (b.frozen && b.max_persons > 1) ? (b.frozen[slot] != 0) : 0;
    s.orient_world = b.orient_world + slot * TM * 3;
    s.trans_world = b.trans_world + slot * TM * 3;
    s.kp_2d_pred = b.kp_2d_pred + slot * TM * NJ * 2;
    s.orient_cam_in_world = b.orient_cam_in_world + slot * TM * 3;
    s.g_j_local = b.g_j_local ? b.g_j_local + slot * TM * NJ * 3 : nullptr;
  }
  sc.fast_free = fast ? fast + o.fast_end : nullptr;
  sc.fast_left = fast ? fast_floats - (size_t)o.fast_end : 0;
  sc.TM = TML;
  sc.ws = ws;
  sc.adam_tab = nullptr;
  // rel_transform_cam is indexed with the padded person count
  sc.rel_stride_p = b.max_persons;
  sc.rel_stride_t = (int)TM;
}

// Integer tables derived from the visibility masks; once per stage.
template <class RT>
GLAMR_HD void setup_tables(RT& rt, Scene& sc) {
  const int T = sc.T, P = sc.P;
  for (int t = rt.tid(); frame_in(t, T); t += rt.nthreads()) {
    int n = 0;
    for (int p = 0; p < P; ++p) n += sc.pc[p].vis[t] != 0.f ? 1 : 0;
    sc.n_vis_persons[t] = n;
  }
  // index among the visible frames = inclusive prefix count - 1 (a block scan over a scratch row: the per-thread walk over all earlier
  // frames it replaces was 300 dependent memory round trips for the last thread -- most of a forward-only launch's 0.65 ms)
  for (int p = 0; p < P; ++p) {
    float* cnt = sc.ps[p].g_theta;                      // free until the first evaluation; [T]
    for (int t = rt.tid(); frame_in(t, T); t += rt.nthreads()) cnt[t] = sc.pc[p].vis[t] != 0.f ? 1.0f : 0.0f;
    rt.sync();
    rt.scan(cnt, T, 1, false);
    for (int t = rt.tid(); frame_in(t, T); t += rt.nthreads()) sc.pc[p].vis_rank[t] = sc.pc[p].vis[t] != 0.f ? (int)cnt[t] - 1 : -1;
    rt.sync();
  }
  for (int i = rt.tid(); i < P * P; i += rt.nthreads()) {
    const int a = i / P, b = i % P;
    int first = -1;
    for (int t = 0; t < T && first < 0; ++t) if (sc.pc[a].vis[t] != 0.f && sc.pc[b].vis[t] != 0.f) first = t;
    sc.pair_first[a * MAXP + b] = first;
  }
  rt.sync();
  for (int t = rt.tid(); frame_in(t, T); t += rt.nthreads()) {
    int src = t;
    while (src >= 0 && sc.n_vis_persons[src] == 0) --src;
    if (src < 0) { src = 0; while (src < T - 1 && sc.n_vis_persons[src] == 0) ++src; }
    sc.fill_src[t] = src;
  }
  rt.sync();
}

// ---- the per-scene driver --------------------------------------------------------------------------------------------

// loss_func.py:248-271 for person p at frame t (both ordered pairs it takes part in).  Returns the unweighted sum of squares of the
// pairs (p, o) and adds the gradient w.r.t. p's world transform [Rk | tw].
template <int FAST, class RT>
GLAMR_HD float rel_transform_term(RT& rt, const Scene& sc, const SceneView& sh, const glamr_stage_desc& st, int P, int p, int t, const float* Rk,
                                           const float* tw, float w_rel, float* gRk, float* g_tw) {
  float value = 0.f;
    float Ti[12];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Ti[i * 4 + j] = Rk[i * 3 + j]; Ti[i * 4 + 3] = tw[i]; }
    float Tiinv[12];
    invert34(Ti, Tiinv);
    float gTi[12];
    for (int k = 0; k < 12; ++k) gTi[k] = 0.f;
    for (int o = 0; o < P; ++o) {
      if (o == p) continue;
      // Everything the pair reads from memory is requested HERE, ahead of the per-lane visibility test: the other person's visibility, world
      // rotation columns and translation, and both directions' 3 x 4 targets (constants of the launch that do not fit the arena: L2).  Asked for where
      // they are used -- behind the test, then inside the direction loop -- every pair waited three memory round trips in a row: 25 of the 109 us of a
      // 4-person iteration (tools/grecon_phases.py).  Same arithmetic, same order.
      const PersonView vo = person_view<FAST>(rt, sc, o);
      const float vis_o = vo.vis[t];
      float d6o[6], two[3], tgt2[2][12];
      for (int k = 0; k < 6; ++k) d6o[k] = vo.d6[t * 6 + k];
      for (int k = 0; k < 3; ++k) two[k] = vo.tw[t * 3 + k];
      for (int dir = 0; dir < 2; ++dir) {
        const int a = dir == 0 ? p : o, b = dir == 0 ? o : p;
        const float* target = sh.rel_cam + (((size_t)a * sh.rel_stride_p + b) * sh.rel_stride_t + t) * 12;
        for (int k = 0; k < 12; ++k) tgt2[dir][k] = target[k];
      }
      if (vis_o == 0.f) continue;
      float To[12], Ro[9];
      cols_to_R(d6o, Ro);
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) To[i * 4 + j] = Ro[i * 3 + j]; To[i * 4 + 3] = two[i]; }
      float Toinv[12];
      invert34(To, Toinv);
      // pair (p, o): rel = inv(T_p) T_o ; pair (o, p): rel = inv(T_o) T_p.  This thread owns T_p's gradient of both.
#pragma unroll
      for (int dir = 0; dir < 2; ++dir) {
        const int a = dir == 0 ? p : o, b = dir == 0 ? o : p;
        const float* target = tgt2[dir];
        const float fw = (sh.pair_first[a * MAXP + b] == t) ? st.first_frame_weight[GLAMR_LOSS_REL_TRANSFORM] : 1.0f;
        float rel[12];
        if (dir == 0) mul34(Tiinv, To, rel); else mul34(Toinv, Ti, rel);
        float grel[12];
        for (int k = 0; k < 12; ++k) grel[k] = 0.f;
        for (int i = 0; i < 3; ++i) {
          for (int j = 0; j < 2; ++j) {
            const float d = (target[i * 4 + j] - rel[i * 4 + j]) * fw;
            if (dir == 0) value += d * d;
            grel[i * 4 + j] = -2.0f * d * fw * w_rel;
          }
          const float d = (target[i * 4 + 3] - rel[i * 4 + 3]) * fw;
          if (dir == 0) value += d * d * st.rel_trans_weight;
          grel[i * 4 + 3] = -2.0f * d * fw * w_rel * st.rel_trans_weight;
        }
        if (w_rel == 0.f) continue;
        if (dir == 0) {
          float gInv[12];
          for (int k = 0; k < 12; ++k) gInv[k] = 0.f;
          mul34_bwd(Tiinv, To, grel, gInv, nullptr);
          invert34_bwd(Ti, gInv, gTi);
        } else {
          mul34_bwd(Toinv, Ti, grel, nullptr, gTi);
        }
      }
    }
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) gRk[i * 3 + j] += gTi[i * 4 + j]; g_tw[i] += gTi[i * 4 + 3]; }

  return value;
}

// SINGLE: the scene has exactly one person.  The person loops and the relative-transform term then disappear at compile time, and
// with them every register spill of the general instance (264 VGPRs -> 0; 39 -> 26 us per iteration on the headline config).
// CAM: how the stage treats the camera, when known at launch: 1 = optimised per frame, 2 = one optimised camera shared by all frames
// (flag_fixed_cam), 0 = anything (constant, or derived from the persons).  Each value removes the other modes' code -- and their
// registers -- from the instance: the general multi-person instance spills 214 VGPRs, its CAM = 1 version 58.
// 3 = a CONSTANT camera: neither optimised nor derived from the persons -- the first stage of the multi-person configurations (variables
// local_xy / local_heading only) and every forward-only pass; without the camera's gradient, its regularisers' reverse pass and the averaging over
// the persons the several-person instances drop from 163 / 114 spilled registers (lite / full arena) to 0 / 3.
inline int camera_mode(const glamr_stage_desc& st) {
  if (!(st.var_mask & GLAMR_VAR_CAM)) return (st.flags & GLAMR_FLAG_CAM_FROM_PERSON) ? 0 : 3;
  return (st.flags & GLAMR_FLAG_FIXED_CAM) ? 2 : 1;
}

// per-iteration observer of the CPU test runtime (tests/hostsim records the parameter trajectory); runtimes without a `trace`
// member -- the device runtime -- compile to nothing
template <class RT> GLAMR_HD auto trace_hook(RT& rt, int it, Scene& sc, int) -> decltype(rt.trace(it, sc), void()) { rt.trace(it, sc); }
template <class RT> GLAMR_HD void trace_hook(RT&, int, Scene&, long) {}

// person parameter block from one layout to another, segment by segment (layouts of different frame counts: the shorter segment)
template <class RT>
GLAMR_HD void copy_person_block(RT& rt, float* dst, const glamr_param_layout& ld, const float* src, const glamr_param_layout& ls) {
  const int so[8] = {ls.local_xy, ls.local_heading, ls.local_dxy, ls.local_dheading, ls.local_z, ls.local_rot, ls.world_dheading, ls.person_stride};
  const int dof[8] = {ld.local_xy, ld.local_heading, ld.local_dxy, ld.local_dheading, ld.local_z, ld.local_rot, ld.world_dheading, ld.person_stride};
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int ns = so[k + 1] - so[k], nd = dof[k + 1] - dof[k], n = ns < nd ? ns : nd;
    for (int i = rt.tid(); i < n; i += rt.nthreads()) dst[dof[k] + i] = src[so[k] + i];
  }
}

// TMC > 0: constant-layout instance (see the view builders): arena / workspace arrays laid out for TMC frames, on-chip parameter blocks in
// the parameter layout of TMC frames (`lo`; the batch arrays keep the batch's own layout `l`).
template <int TMC> struct OnChipLayout { static constexpr glamr_param_layout value = make_layout(1, TMC > 0 ? TMC : 2); };
template <int FAST, bool SINGLE, int CAM, int TMC = 0, class RT>
GLAMR_HD void run_scene(RT& rt, Scene& sc, const glamr_stage_desc& st, const glamr_param_layout& l) {
  constexpr bool AF = FAST == 1 && SINGLE;          // parameters + Adam moments on chip (assemble_scene: adam_fast)
  static_assert(TMC == 0 || AF, "constant layouts are for single-person scenes with the full arena");
  const glamr_param_layout& lo = TMC > 0 ? OnChipLayout<TMC>::value : l;      // layout of the on-chip parameter / moment blocks
  // One-thread-per-frame instances with a compile-time geometry (TMC > 0: one person, TMC frames of layout, (TMC + 63) / 64 waves) carry a
  // frame's scan operands and results, its trajectory row and the cos / sin of its heading in REGISTERS from phase to phase: the scans take and
  // return registers (DeviceRT::scan_regs), `theta`, `xy` and `g_theta` are never stored, and phases B and C do not re-read what phase A wrote.
  // GLAMR_REG_MASK (development aid) selects the pieces: 1 heading prefix sum, 2 the reverse phases' sums, 4 planar position, 8 the trajectory
  // row, 16 cos / sin of the heading, 32 the camera parameters requested before the barrier that precedes phase D.  Default 23: pieces 8 and
  // 32 are measured (the launch another 0.5 % shorter) and OFF, because they change results in the last bit: with a frame's 6D rotation
  // arriving in registers instead of from LDS the compiler fuses a different product of rot6d_to_rotmat's `x0 x0 + x1 x1 + x2 x2` and
  // `b1 . a2` (which operand of a commutative node comes first is not stable under such edits), and the optimisation of a sequence with a
  // detection gap is chaotic in exactly those bits.  Every enabled piece leaves all outputs of all stages bit-identical to
  // round 4's kernel (tools/stage_bits.py, seven cases, profiles/r05_stage_bits.log).
#ifndef GLAMR_REG_MASK
#define GLAMR_REG_MASK 23
#endif
  constexpr bool REG_ANY = TMC > 0 && RT::one_thread_per_frame;
  constexpr bool REG = REG_ANY && (GLAMR_REG_MASK & 1);          // heading prefix sum
  constexpr bool REGB = REG_ANY && (GLAMR_REG_MASK & 2);
  constexpr bool REG_XY = REG_ANY && (GLAMR_REG_MASK & 4);       // planar position prefix sum
  constexpr bool REG_L = REG_ANY && (GLAMR_REG_MASK & 8);        // trajectory row
  constexpr bool REG_CS = REG_ANY && (GLAMR_REG_MASK & 16);      // cos / sin of the heading
  constexpr bool REG_CAM = REG_ANY && (GLAMR_REG_MASK & 32);     // camera parameters requested early
  constexpr int NWC = TMC > 0 ? (TMC + 63) / 64 : 1;
  // Uniform scalars of the scene description the loop needs (single-person instances): the description lives in LDS, which every phase
  // writes, so the compiler must re-read them -- a ds_read, a wait and two v_readfirstlane at the top of every phase -- unless they are
  // copied out once.  Filled in after the stage setup (njc / njc_fast are computed there).
  struct PersonScalars { int fr_start, fr_end, njc, njc_fast; } hs{0, 0, 0, 0};
  bool hs_set = false;
  auto pv = [&](int p) {
    PersonView w = person_view<FAST, AF, TMC>(rt, sc, p);
    if (SINGLE && hs_set) { w.fr_start = hs.fr_start; w.fr_end = hs.fr_end; w.njc = hs.njc; w.njc_fast = hs.njc_fast; }
    return w;
  };
  // Frame loops of the full-arena instances make ONE pass (the launcher only selects them when every frame has its own thread): with
  // a step the compiler can see is larger than any sequence the loop is an `if`, and no per-array 64-bit induction pointers stay
  // live across the whole body (two registers each, a few dozen arrays).
  const int fstep = ((FAST == 1 || FAST == 3) && RT::one_thread_per_frame) ? (1 << 20) : rt.nthreads();
  const int T = sc.T, P = SINGLE ? 1 : sc.P;
  const bool var_cam = CAM == 3 ? false : (CAM != 0 ? true : (bool)(st.var_mask & GLAMR_VAR_CAM));
  const bool fixed_cam = CAM == 2 ? true : ((CAM == 1 || CAM == 3) ? false : (bool)(st.flags & GLAMR_FLAG_FIXED_CAM));      // (only read where var_cam holds)
  const bool cam_from_person = CAM != 0 ? false : (!var_cam && (st.flags & GLAMR_FLAG_CAM_FROM_PERSON));
  const bool has_wd = (st.flags & GLAMR_FLAG_HAS_WORLD_DHEADING) || (st.var_mask & GLAMR_VAR_WORLD_DHEADING);
  auto on = [&](int id) { return (st.loss_mask >> id) & 1u; };
  auto active = [&](int id) { return ((st.loss_mask >> id) & 1u) && !((st.monitor_mask >> id) & 1u); };
  auto ffo = [&](int id) { return (st.first_frame_only_mask >> id) & 1u; };

  // A forward pass that is only there for the world poses (GLAMR_FLAG_POSES_ONLY: init_data's pass before init_cam_pose(all_frames), the first
  // launch of every iteration of the launch-by-launch schedules) stops after phase D: it needs neither the visibility tables nor the
  // normalisers nor the per-joint score sums -- 0.40 -> 0.2 ms of a launch that sits on the pipeline's critical chain (round 5).
  const bool poses_only = st.niters == 0 && (st.flags & GLAMR_FLAG_POSES_ONLY) && !cam_from_person;
  // The keypoint row's arithmetic with folded constants (GLAMR_KP_FOLD, joint_nb) -- except where the camera rides on the person (cfg glamr_3dpw):
  // there the first Adam steps are lr x the SIGN of gradients that are rounding noise, the reference's sign is reproduced by rounds 4-5's operation
  // order (0.002 px after 15 steps against 0.20 px with any other: tests/grecon_common.py KSTEP_TOL), and that order is kept, as for the scans.
  const bool kp_fold = GLAMR_KP_FOLD != 0 && !cam_from_person;
  // GLAMR_FLAG_KEEP_TABLES (launch-by-launch schedules, every gradient launch of a stage but its first, on the SAME workspace): what the stage set-up
  // leaves in the workspace and no iteration touches -- visibility tables, per-joint score sums, the orientation targets, the workspace rows of the
  // keypoint table -- is still there.  Honoured by the several-person run-time-layout instances only (see NR below: the others keep their code).
  constexpr bool KT = !SINGLE && TMC == 0;
  bool keep_tables = false;
  if constexpr (KT) keep_tables = !poses_only && st.niters > 0 && (st.flags & GLAMR_FLAG_KEEP_TABLES);
  if constexpr (KT) { if (!poses_only && !keep_tables) setup_tables(rt, sc); } else { if (!poses_only) setup_tables(rt, sc); }
  // ---- stage setup: normalisers, Adam state, camera parameters from the current camera (get_parameter :596-606) ----------
  float n_vis_total = 0.f, n_exist = 0.f, n_exist_m1 = 0.f;
  for (int p = 0; p < P && !poses_only; ++p) {
    float c = 0.f;
    for (int t = rt.tid(); frame_in(t, T); t += fstep) c += sc.pc[p].vis[t];
    n_vis_total += rt.reduce_sum(c);
    const int n = sc.pc[p].fr_end - sc.pc[p].fr_start;
    n_exist += (float)n;
    n_exist_m1 += (float)(n - 1);
  }
  for (int i = rt.tid(); i < l.person0; i += rt.nthreads()) sc.cg[i] = 0.f;
  for (int i = rt.tid(); i < lo.cam_inv_rot_res; i += rt.nthreads()) { sc.cm[i] = 0.f; sc.cv[i] = 0.f; }      // moments: compact 9 T
  for (int p = 0; p < P; ++p) {
    for (int i = rt.tid(); i < l.person_stride; i += rt.nthreads()) sc.ps[p].g[i] = 0.f;
    for (int i = rt.tid(); i < lo.person_stride; i += rt.nthreads()) { sc.ps[p].m[i] = 0.f; sc.ps[p].v[i] = 0.f; }
  }
  const int rs = AF ? lo.cam_inv_rot_res : 0;          // index shift of the camera residuals in the on-chip (compact) camera block
  const int Tb = l.cam_trans / 6;                      // frames of the batch layout
  if (AF) {
    // parameters on chip: the person block as it is; the camera block = the residuals unless the stage optimises the camera itself (then
    // it is initialised from the camera poses right below)
    copy_person_block(rt, sc.ps[0].p, lo, sc.ps[0].p_g, l);
    if (!var_cam) {
      for (int i = rt.tid(); i < 6 * Tb; i += rt.nthreads()) sc.cp[i] = sc.cp_g[l.cam_inv_rot_res + i];
      for (int i = rt.tid(); i < 3 * Tb; i += rt.nthreads()) sc.cp[lo.cam_trans + i] = sc.cp_g[l.cam_inv_trans_res + i];
    }
  }
  if (TMC > 0) {
    // stage-constant inputs next to the rest of the workspace: the loop reads them at constant offsets (a missing heading mask = zeros)
    const PersonConst& c = sc.pc[0];
    PersonState& s = sc.ps[0];
    for (int t = rt.tid(); frame_in(t, T); t += fstep) {
      s.in_vis[t] = c.vis[t];
      s.in_dmask[t] = c.dheading_mask ? c.dheading_mask[t] : 0.f;
      for (int k = 0; k < 9; ++k) s.in_cam_K[t * 9 + k] = c.cam_K[(size_t)t * 9 + k];
      for (int k = 0; k < 11; ++k) s.in_prior[t * 11 + k] = c.prior[(size_t)t * 11 + k];
      for (int k = 0; k < 3; ++k) { s.in_base_orient[t * 3 + k] = c.base_orient[t * 3 + k]; s.in_base_trans[t * 3 + k] = c.base_trans[t * 3 + k]; }
      for (int k = 0; k < 12; ++k) s.in_person2cam[t * 12 + k] = c.person2cam[(size_t)t * 12 + k];
    }
  }
  if (var_cam && !(st.flags & GLAMR_FLAG_KEEP_CAM_PARAMS)) {
    const int rows = fixed_cam ? 1 : T;
    for (int t = rt.tid(); t < rows; t += rt.nthreads()) {
      const float* M = sc.cam_pose + (size_t)t * 12;
      for (int r = 0; r < 3; ++r) {
        sc.cp[lo.cam_rot6d + t * 6 + r] = M[r * 4 + 0];
        sc.cp[lo.cam_rot6d + t * 6 + 3 + r] = M[r * 4 + 1];
        sc.cp[lo.cam_trans + t * 3 + r] = M[r * 4 + 3];
      }
    }
  } else if (var_cam && AF) {
    // a stage continued launch by launch: the camera parameters of the batch array are current -- into the on-chip block
    const int rows = fixed_cam ? 1 : T;
    for (int t = rt.tid(); t < rows; t += rt.nthreads()) {
      for (int k = 0; k < 6; ++k) sc.cp[lo.cam_rot6d + t * 6 + k] = sc.cp_g[l.cam_rot6d + t * 6 + k];
      for (int k = 0; k < 3; ++k) sc.cp[lo.cam_trans + t * 3 + k] = sc.cp_g[l.cam_trans + t * 3 + k];
    }
  }
  for (int p = 0; p < P && !(KT && keep_tables); ++p)
    for (int t = rt.tid(); frame_in(t, T); t += fstep) {
      const PersonConst& c = sc.pc[p];
      if (t < c.fr_end - c.fr_start) sc.ps[p].h_prior[t] = rm::atan2s(c.prior[(size_t)t * 11 + 10], c.prior[(size_t)t * 11 + 9]);
      float Rb[9];
      rm::aa_to_rotmat_k(c.orient_cam + t * 3, Rb);
      for (int r = 0; r < 3; ++r) { sc.ps[p].oc6[t * 6 + r] = Rb[r * 3 + 0]; sc.ps[p].oc6[t * 6 + 3 + r] = Rb[r * 3 + 1]; }
    }
  // per-joint sum over visible frames of thresholded score^2 (first_frame_only broadcasting of kp_2d, loss_func.py:27-33)
  for (int p = 0; p < P && !poses_only && !(KT && keep_tables); ++p) {
    float cj[NJ];
    for (int j = 0; j < NJ; ++j) cj[j] = 0.f;
    for (int t = rt.tid(); frame_in(t, T); t += fstep) {
      if (sc.pc[p].vis[t] == 0.f) continue;
      for (int j = 0; j < NJ; ++j) {
        const float sj = sc.pc[p].kp_score[(size_t)t * NJ + j];
        if (sj >= st.kp_min_conf) cj[j] += sj * sj;
      }
    }
    rt.reduce_sum_n(cj);                               // all 26 sums with one exchange (26 separate block reductions were 52 barriers)
    if (rt.tid() == 0) for (int j = 0; j < NJ; ++j) sc.ps[p].kp_wsum[j] = cj[j];
  }
  rt.sync();
  // scored joints and their compact per-frame table (joint position, 2-D target, residual weight), on chip when it fits
  if (rt.tid() == 0)
    for (int p = 0; p < P; ++p) {
      PersonState& s = sc.ps[p];
      s.njc = 0;
      // (a forward-only launch, niters == 0, evaluates all 26 joints from the full arrays once: it needs no table)
      if (on(GLAMR_LOSS_KP_2D) && st.niters > 0) for (int j = 0; j < NJ; ++j) if (s.kp_wsum[j] > 0.f) s.jidx[s.njc++] = j;
      const size_t per_joint = (size_t)6 * sc.TM;
      s.njc_fast = 0;
      if (sc.fast_free) {
        const size_t fit = sc.fast_left / per_joint;
        s.njc_fast = (int)(fit < (size_t)s.njc ? fit : (size_t)s.njc);
        s.kpc = sc.fast_free; sc.fast_free += s.njc_fast * per_joint; sc.fast_left -= s.njc_fast * per_joint;
      }
    }
  rt.sync();
  for (int p = 0; p < P; ++p) {
    const PersonConst& c = sc.pc[p];
    PersonState& s = sc.ps[p];
    const bool kp_first = ffo(GLAMR_LOSS_KP_2D);
    for (int t = rt.tid(); frame_in(t, T); t += fstep) {
      const bool kp_frame = c.vis[t] != 0.f && (!kp_first || c.vis_rank[t] == 0);
      const float rank_w = (c.vis_rank[t] >= 0 && c.vis_rank[t] < 10) ? st.first_frame_weight[GLAMR_LOSS_KP_2D] : 1.0f;
      for (int jj = 0; jj < s.njc; ++jj) {
        if (KT && keep_tables && jj >= s.njc_fast) break;      // (the workspace rows are still there; the on-chip rows are rebuilt with the arena)
        const int j = s.jidx[jj];
        float w = 0.f;
        if (kp_frame) {
          if (kp_first) w = s.kp_wsum[j] * rank_w;
          else { const float sj = c.kp_score[(size_t)t * NJ + j]; const float q = sj < st.kp_min_conf ? 0.f : sj; w = q * q * rank_w; }
        }
        float* o = (jj < s.njc_fast ? s.kpc : s.kpc_ws) + (size_t)jj * 6 * sc.TM + t;
        for (int k = 0; k < 3; ++k) o[k * sc.TM] = c.j_local[((size_t)t * NJ + j) * 3 + k];
        o[3 * sc.TM] = c.kp_2d[((size_t)t * NJ + j) * 2 + 0];
        o[4 * sc.TM] = c.kp_2d[((size_t)t * NJ + j) * 2 + 1];
        // kp_fold: the table's weight column carries the constant factor of the robust term's derivative as well, 2 sigma^4 = 2e8 (only
        // iterations that are not the last read the table: the last evaluation, which reports the loss VALUE, walks the full arrays)
        o[5 * sc.TM] = kp_fold ? w * 2e8f : w;
      }
    }
  }
  rt.sync();

  // Per-frame constants of the iteration kept in REGISTERS for the whole launch (constant-layout instances: one frame per thread, one
  // person): the intrinsics, the camera-frame orientation target, the prior row, visibility.  The loop used to fetch these 110 B per
  // frame from their workspace copies every iteration -- a quarter of the 122 KB a scene re-reads per iteration, and what pushed the 32
  // scenes resident on an XCD past its 4 MB of L2 (29 KB per scene-iteration fetched from beyond it: profiles/r03_pmc_stage_kernel.json).
  constexpr bool HOIST = GLAMR_HOIST_FRAME_CONSTS && TMC > 0 && RT::one_thread_per_frame;
  struct FrameRegs { float K[9], tgt[6], prior[9], h_prior, dmask, vis; int rank; } fr;
  if (HOIST) {
    const int t = rt.tid();
    if (frame_in(t, T)) {
      const PersonView c = pv(0);
      for (int k = 0; k < 9; ++k) fr.K[k] = c.cam_K[(size_t)t * 9 + k];
      for (int k = 0; k < 6; ++k) fr.tgt[k] = c.oc6[t * 6 + k];
      fr.vis = c.vis[t];
      fr.rank = c.vis_rank[t];
      const int e = t - c.fr_start;
      const bool ex = t >= c.fr_start && t < c.fr_end;
      for (int k = 0; k < 9; ++k) fr.prior[k] = ex ? c.prior[(size_t)e * 11 + k] : 0.f;
      fr.h_prior = ex ? c.h_prior[e] : 0.f;
      fr.dmask = (ex && c.dheading_mask) ? c.dheading_mask[e] : 0.f;
    }
  }

  if (SINGLE) {
    const PersonView w0 = person_view<FAST, AF, TMC>(rt, sc, 0);
    hs.fr_start = w0.fr_start; hs.fr_end = w0.fr_end; hs.njc = w0.njc; hs.njc_fast = w0.njc_fast;
    hs_set = true;
  }
  const float* const adam_tab = uni(sc.adam_tab);      // (read once: see PersonScalars)
  const int niters = st.niters;
  double b1p = 1.0, b2p = 1.0;
  int n_done = 0;
  const int n_eval = niters > 0 ? niters : 1;
  const SceneView sh = scene_view<FAST, AF, TMC>(rt, sc);
  // One evaluation (+ update).  Instantiated twice: the iterations that only update (no reported values, no outputs) and the LAST
  // evaluation, which also writes the outputs and reduces the loss values -- keeping that code (axis-angle conversions, all 26
  // joints, 13 accumulators) out of the hot instance keeps its registers out of it too.
  auto evaluate = [&](auto last_tag) {
    constexpr bool last = decltype(last_tag)::value;
    const bool update = niters > 0;
    b1p *= 0.9; b2p *= 0.999;
    AdamCoef ac;
    const float* tab = adam_tab;
    if (tab) {                                     // the host's table: Python's own arithmetic (pow in double)
      ac.neg_step = glob(tab)[2 * n_done]; ac.bc2_sqrt = glob(tab)[2 * n_done + 1];
    } else {                                       // stages longer than the table: running products (same values to ~1 ulp of a double)
      ac.neg_step = (float)(-(st.lr / (1.0 - b1p)));
      ac.bc2_sqrt = (float)sqrt(1.0 - b2p);
    }
    ac.finish();
    ++n_done;
    float lsum[GLAMR_NUM_LOSSES];
    for (int i = 0; i < GLAMR_NUM_LOSSES; ++i) lsum[i] = 0.f;
    float kp_dist_cnt = 0.f;
    GLAMR_MARK_BEGIN(rt);

    // ---- A: heading increments (arrays are indexed by VIDEO frame t and zero outside the person's existing range, so the
    //         prefix sums run over [0,T) and element t is always owned by thread t mod nthreads) ------------------------------
    // The camera rides on the person (cfg glamr_3dpw): the world trajectory is a GAUGE there, the reference's own gradient of the planar
    // position is below the rounding of what it is summed from, and Adam's first steps are lr x its SIGN -- which the summation order of the
    // stage's prefix / suffix sums decides.  These instances keep the Hillis-Steele order of rounds 1-3 for all four scans (it reproduces the
    // reference's 15-step state to 0.002 px; the DPP order 0.20 px, and so does DPP with only the planar-position gradient's scan in the old
    // order: profiles/r05_gputest_a.log); every other instance runs the DPP scans.
    const bool scan_shuffle = cam_from_person;
    // REG instances: this frame's row, heading prefix sum with its cos / sin, and planar position, carried in registers (lanes beyond T: zeros)
    LocalRow Lr{};
    float th_r[1] = {0.f}, cs_r = 1.f, sn_r = 0.f, xy_r[2] = {0.f, 0.f};
    for (int t = rt.tid(); frame_in(t, T); t += fstep)
      for (int p = 0; p < P; ++p) {
        const PersonView c = pv(p);
        float v = 0.f;
        if (t >= c.fr_start && t < c.fr_end) {
          const LocalRow L = HOIST ? local_row_regs(fr.prior, fr.h_prior, fr.dmask, c.dheading_mask != nullptr, c.p, lo, t - c.fr_start)
                                   : local_row(c, lo, t - c.fr_start);
          store_row(pv(p).Lc, sh.TM, t, L);
          if (REG_L) Lr = L;
          // atan2(sin h, cos h) of the reference (:401-405) only wraps h into (-pi, pi]: done arithmetically
          v = L.h - 6.28318530717958647692f * rintf(L.h * 0.15915494309189533577f);
        }
        if (REG) th_r[0] = v; else pv(p).theta[t] = v;
      }
    if constexpr (REG) {
      rt.template scan_regs<NWC, 1>(th_r, false, scan_shuffle);
    } else
#ifdef GLAMR_GRECON_WIDE      // absolute_heading (:59,283,421): the per-frame headings are not summed up (traj_local2global_heading(local_heading=False))
    if (st.flags & GLAMR_FLAG_ABSOLUTE_HEADING) rt.sync(); else
#endif
    {
      float* ch[8];
      if constexpr (MAXP <= 8) {
        for (int p = 0; p < P; ++p) ch[p] = pv(p).theta;
        rt.template scan_multi<(FAST != 0)>(ch, P, T, 1, false, scan_shuffle);
      } else {
        for (int p0 = 0; p0 < P; p0 += 8) {      // (the scan scratch holds 16 channels: 8 persons a call)
          const int np = P - p0 < 8 ? P - p0 : 8;
          for (int p = 0; p < np; ++p) ch[p] = pv(p0 + p).theta;
          rt.template scan_multi<(FAST != 0)>(ch, np, T, 1, false, scan_shuffle);
        }
      }
    }
    // own element of the prefix sum is final: its cos / sin serve phases B, C and I (this frame's and the next frame's)
    for (int t = rt.tid(); frame_in(t, T); t += fstep)
      for (int p = 0; p < P; ++p) {
        float sn, cs;
        rm::sincos_(REG ? th_r[0] : pv(p).theta[t], sn, cs);
        pv(p).csn[t * 2 + 0] = cs;
        pv(p).csn[t * 2 + 1] = sn;
        if (REG_CS) { cs_r = cs; sn_r = sn; }
      }
    rt.sync();
    GLAMR_MARK(rt, 0);
    // ---- B: planar displacement in world axes -------------------------------------------------------------------------
    for (int t = rt.tid(); frame_in(t, T); t += fstep)
      for (int p = 0; p < P; ++p) {
        const PersonView c = pv(p);
        float dx = 0.f, dy = 0.f;
        if (t >= c.fr_start && t < c.fr_end) {
          const int e = t - c.fr_start;
          const LocalRow L = REG_L ? Lr : load_row(pv(p).Lc, sh.TM, t);
          dx = L.dx; dy = L.dy;
          // (neighbours' values are fetched with a clamped index, outside the per-lane condition: a load behind a condition cannot be
          // issued early or batched with others, and costs its own round trip -- here and in phases E3, E4 and I)
          const int tp = e > 0 ? t - 1 : t;
          const float cs = pv(p).csn[tp * 2 + 0], sn = pv(p).csn[tp * 2 + 1];
          if (e > 0) {
            dx = L.dx * cs - L.dy * sn;
            dy = L.dx * sn + L.dy * cs;
          }
        }
        if (REG_XY) { xy_r[0] = dx; xy_r[1] = dy; } else {
          pv(p).xy[t * 2 + 0] = dx;
          pv(p).xy[t * 2 + 1] = dy;
        }
      }
    // REG instances with an own camera per frame: this frame's camera parameters (written by this thread, in phase E5 of the previous
    // iteration) are requested before the scan's barrier instead of in phase D, where nothing covered their latency
    constexpr bool CAMPRE = REG_CAM && CAM == 1;
    float cam9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (CAMPRE)
      for (int t = rt.tid(); frame_in(t, T); t += fstep) {
        for (int k = 0; k < 6; ++k) cam9[k] = sh.cp[lo.cam_rot6d + t * 6 + k];
        for (int k = 0; k < 3; ++k) cam9[6 + k] = sh.cp[lo.cam_trans + t * 3 + k];
      }
    if constexpr (REG_XY) {
      rt.template scan_regs<NWC, 2>(xy_r, false, scan_shuffle);
    } else {
      float* ch[16];
      if constexpr (MAXP <= 8) {
        for (int p = 0; p < P; ++p) { ch[2 * p] = pv(p).xy; ch[2 * p + 1] = pv(p).xy + 1; }
        rt.template scan_multi<(FAST != 0)>(ch, 2 * P, T, 2, false, scan_shuffle);
      } else {
        for (int p0 = 0; p0 < P; p0 += 8) {
          const int np = P - p0 < 8 ? P - p0 : 8;
          for (int p = 0; p < np; ++p) { ch[2 * p] = pv(p0 + p).xy; ch[2 * p + 1] = pv(p0 + p).xy + 1; }
          rt.template scan_multi<(FAST != 0)>(ch, 2 * np, T, 2, false, scan_shuffle);
        }
      }
    }
    GLAMR_MARK(rt, 1);
    // ---- C: world orientation / translation (own elements of theta / xy only: no barrier needed) ----------------------------
    for (int t = rt.tid(); frame_in(t, T); t += fstep) {
      for (int p = 0; p < P; ++p) {
        const PersonView c = pv(p);
        const PersonView& s = c;
        float c1[3], c2[3], tb[3], phi = 0.f;
        if (t >= c.fr_start && t < c.fr_end) {
          const LocalRow L = REG_L ? Lr : load_row(s.Lc, sh.TM, t);
          float Rl[9];
          rm::rot6d_to_rotmat(L.r6, Rl);
          for (int k = 0; k < 3; ++k) { c1[k] = Rl[k * 3 + 1]; c2[k] = Rl[k * 3 + 2]; }
          phi = REG ? th_r[0] : s.theta[t];
          tb[0] = REG_XY ? xy_r[0] : s.xy[t * 2 + 0]; tb[1] = REG_XY ? xy_r[1] : s.xy[t * 2 + 1]; tb[2] = L.z;
        } else {
          float Rb[9];
          rm::aa_to_rotmat_k(c.base_orient + t * 3, Rb);
          for (int k = 0; k < 3; ++k) { c1[k] = Rb[k * 3 + 0]; c2[k] = Rb[k * 3 + 1]; tb[k] = c.base_trans[t * 3 + k]; }
        }
        float sn = 0.f, cs = 1.f;
        const bool frozen = !SINGLE && c.frozen;            // the given pose already carries its owner's world heading offset
        if (has_wd && !frozen) { phi += s.p[lo.world_dheading + t]; rm::sincos_(phi, sn, cs); }
        else if (t >= c.fr_start && t < c.fr_end) { cs = REG_CS ? cs_r : s.csn[t * 2 + 0]; sn = REG_CS ? sn_r : s.csn[t * 2 + 1]; }
        float w1[3], w2[3];
        rotz2(cs, sn, c1, w1);
        rotz2(cs, sn, c2, w2);
        s.Lc[10 * sh.TM + t] = cs;
        s.Lc[11 * sh.TM + t] = sn;
        for (int k = 0; k < 3; ++k) {
          s.d6[t * 6 + k] = w1[k];
          s.d6[t * 6 + 3 + k] = w2[k];
          s.tw[t * 3 + k] = tb[k];
        }
        if (last && !frozen) {
          float Rw[9], ow[3];
          cols_to_R(s.d6 + t * 6, Rw);
          rm::rotmat_to_aa(Rw, ow);
          for (int k = 0; k < 3; ++k) { s.orient_world[t * 3 + k] = ow[k]; s.trans_world[t * 3 + k] = tb[k]; }
        }
      }
      // ---- D: camera of this frame, unless it is derived from the persons (needs other frames' transforms) -----------------
      if (!cam_from_person) {
        float M[12], Mi[12];
        if (var_cam) {
          const int row = fixed_cam ? 0 : t;
          float R[9];
          rm::rot6d_to_rotmat(CAMPRE ? cam9 : sh.cp + lo.cam_rot6d + row * 6, R);
          for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i * 3 + j]; M[i * 4 + 3] = CAMPRE ? cam9[6 + i] : sh.cp[lo.cam_trans + row * 3 + i]; }
        } else {
          for (int k = 0; k < 12; ++k) M[k] = sh.cam_pose[(size_t)t * 12 + k];
        }
        invert34(M, Mi);
        for (int k = 0; k < 12; ++k) sh.cam_inv[(size_t)t * 12 + k] = Mi[k];
        for (int k = 0; k < 3; ++k) sh.cam_t[t * 3 + k] = M[k * 4 + 3];
        if (last && var_cam) for (int k = 0; k < 12; ++k) sh.cam_pose[(size_t)t * 12 + k] = M[k];      // in/out array: the last forward's camera
      }
    }
    rt.sync();
    GLAMR_MARK(rt, 2);
    if (cam_from_person) {
      for (int t = rt.tid(); frame_in(t, T); t += fstep) {
        float M[12], Mi[12];
        const int src = sh.fill_src[t];
        float avg[12];
        for (int k = 0; k < 12; ++k) avg[k] = 0.f;
        for (int p = 0; p < P; ++p) {
          if (pv(p).vis[src] == 0.f) continue;
          float Tw[12], Rk[9], C[12];
          cols_to_R(pv(p).d6 + src * 6, Rk);
          for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Tw[i * 4 + j] = Rk[i * 3 + j]; Tw[i * 4 + 3] = pv(p).tw[src * 3 + i]; }
          mul34(Tw, pv(p).person2cam + (size_t)src * 12, C);
          for (int k = 0; k < 12; ++k) avg[k] += C[k];
        }
        const float inv_n = 1.0f / (float)sh.n_vis_persons[src];
        for (int k = 0; k < 12; ++k) avg[k] = avg[k] * inv_n;   // sum(...) / num_persons  (:492)
        float r6[6];
        for (int r = 0; r < 3; ++r) { r6[r] = avg[r * 4 + 0]; r6[3 + r] = avg[r * 4 + 1]; }
        if (sh.n_vis_persons[t] == 0) for (int k = 0; k < 6; ++k) r6[k] += sh.cp[lo.cam_inv_rot_res - rs + t * 6 + k];
        float R[9];
        rm::rot6d_to_rotmat(r6, R);
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Mi[i * 4 + j] = R[i * 3 + j]; Mi[i * 4 + 3] = avg[i * 4 + 3] + sh.cp[lo.cam_inv_trans_res - rs + t * 3 + i]; }
        invert34(Mi, M);
        for (int k = 0; k < 12; ++k) sh.cam_inv[(size_t)t * 12 + k] = Mi[k];
        for (int k = 0; k < 3; ++k) sh.cam_t[t * 3 + k] = M[k * 4 + 3];
        if (last) for (int k = 0; k < 12; ++k) sh.cam_pose[(size_t)t * 12 + k] = M[k];
      }
      rt.sync();
    }
    GLAMR_MARK(rt, 3);
    // a forward pass that is only there for the world poses (init_data's pass before init_cam_pose(all_frames) :243-246): done
    if (last && niters == 0 && (st.flags & GLAMR_FLAG_POSES_ONLY)) return;
    // ---- E: residuals and per-frame gradients -----------------------------------------------------------------------------
    const float w_kp = active(GLAMR_LOSS_KP_2D) ? st.loss_weight[GLAMR_LOSS_KP_2D] / n_vis_total : 0.f;
    const float n_ctr = ffo(GLAMR_LOSS_CAM_TRAJ_ROT) ? (float)P : n_vis_total;
    const float w_ctr = active(GLAMR_LOSS_CAM_TRAJ_ROT) ? st.loss_weight[GLAMR_LOSS_CAM_TRAJ_ROT] / n_ctr : 0.f;
    const float n_trs = (float)(P * (T - 1));
    const float w_trs = active(GLAMR_LOSS_TRAJ_ROT_SMOOTHNESS) ? st.loss_weight[GLAMR_LOSS_TRAJ_ROT_SMOOTHNESS] / n_trs : 0.f;
    const bool cam_terms = !(st.flags & GLAMR_FLAG_NO_CAMERA_TERMS);      // false: the camera-only residuals are another rank's
    const float w_crs = (cam_terms && active(GLAMR_LOSS_CAM_INV_ROT_SMOOTHNESS)) ? st.loss_weight[GLAMR_LOSS_CAM_INV_ROT_SMOOTHNESS] / (float)(T - 1) : 0.f;
    const float w_cos = (cam_terms && active(GLAMR_LOSS_CAM_ORIGIN_SMOOTHNESS)) ? st.loss_weight[GLAMR_LOSS_CAM_ORIGIN_SMOOTHNESS] / (float)(T - 1) : 0.f;
    const float n_up = ffo(GLAMR_LOSS_CAM_UP_REG) ? 1.0f : (float)T;
    const float w_up = (cam_terms && active(GLAMR_LOSS_CAM_UP_REG)) ? st.loss_weight[GLAMR_LOSS_CAM_UP_REG] / n_up : 0.f;
    const float n_rel = (float)(P * (P - 1) * T);
    const float w_rel = (active(GLAMR_LOSS_REL_TRANSFORM) && P > 1) ? st.loss_weight[GLAMR_LOSS_REL_TRANSFORM] / n_rel : 0.f;
    const float min_conf = st.kp_min_conf;

    float gfix[9];
    for (int k = 0; k < 9; ++k) gfix[k] = 0.f;
    for (int t = rt.tid(); frame_in(t, T); t += fstep) {
      float gC[12], gCi[12];
      for (int k = 0; k < 12; ++k) { gC[k] = 0.f; gCi[k] = 0.f; }
      // everything this frame needs from the workspace is requested up front
      const bool own_cam = update && var_cam && !fixed_cam && !cam_from_person;
      // world-to-camera [Rc | tc] from the on-chip copies: Rc is the transpose of the camera-to-world rotation
      const float* Mi = sh.cam_inv + (size_t)t * 12;
      float M[12], Rc[9];
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) M[i * 4 + j] = Mi[j * 4 + i]; M[i * 4 + 3] = sh.cam_t[t * 3 + i]; }
      get_R(M, Rc);
      for (int p = 0; p < P; ++p) {
        const PersonView c = pv(p);
        const PersonView& s = c;
        if (!SINGLE && c.frozen) {
          for (int k = 0; k < 6; ++k) s.g_d6[t * 6 + k] = 0.f;
          for (int k = 0; k < 3; ++k) s.g_tw[t * 3 + k] = 0.f;
          continue;
        }
        const float* tw = s.tw + t * 3;
        float K[9], tgt[6];
        for (int k = 0; k < 9; ++k) K[k] = HOIST ? fr.K[k] : c.cam_K[(size_t)t * 9 + k];
        for (int k = 0; k < 6; ++k) tgt[k] = HOIST ? fr.tgt[k] : s.oc6[t * 6 + k];
        const float vis_t = HOIST ? fr.vis : c.vis[t];
        const int rank_t = HOIST ? fr.rank : c.vis_rank[t];
        float g_tw[3] = {0, 0, 0};
        float gRk[9];
        for (int k = 0; k < 9; ++k) gRk[k] = 0.f;
        float Rk[9];
        cols_to_R(s.d6 + t * 6, Rk);
        const bool visible = vis_t != 0.f;
        // orientation seen from the camera: transform_rot(cam_pose, orient_world)  (:512)
        float Mk[9];
        rm::mat3_mul(Rc, Rk, Mk);
        if (last) {
          float ociw[3];
          rm::rotmat_to_aa(Mk, ociw);
          for (int k = 0; k < 3; ++k) s.orient_cam_in_world[t * 3 + k] = ociw[k];
        }
        if (on(GLAMR_LOSS_CAM_TRAJ_ROT) && visible && (!ffo(GLAMR_LOSS_CAM_TRAJ_ROT) || rank_t == 0)) {
          const float* Ra = Mk;             // = aa2R(R2aa(Mk)) of the reference
          const float fw = (!ffo(GLAMR_LOSS_CAM_TRAJ_ROT) && rank_t == 0) ? st.first_frame_weight[GLAMR_LOSS_CAM_TRAJ_ROT] : 1.0f;
          float gRa[9];
          for (int k = 0; k < 9; ++k) gRa[k] = 0.f;
          for (int r = 0; r < 3; ++r)
            for (int col = 0; col < 2; ++col) {
              const float d = (tgt[col * 3 + r] - Ra[r * 3 + col]) * fw;
              if (last) lsum[GLAMR_LOSS_CAM_TRAJ_ROT] += d * d;
              gRa[r * 3 + col] = -2.0f * d * fw * w_ctr;
            }
          if (w_ctr != 0.f) {
            const float* gMk = gRa;
            float gRc[9];
            for (int k = 0; k < 9; ++k) gRc[k] = 0.f;
            rm::mat3_mul_bwd(Rc, Rk, gMk, gRc, gRk);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gC[i * 4 + j] += gRc[i * 3 + j];
          }
        }
        GLAMR_MARK(rt, 9);
        // joints, projection, 2-D keypoint residual  (:517-528, loss_func.py:15-57)
        const bool kp_first = ffo(GLAMR_LOSS_KP_2D);
        const bool need_kp = last || (w_kp != 0.f && visible && (!kp_first || rank_t == 0));
        if (need_kp) {
          const float* Rs = Rk;             // smplx Rodrigues of the same rotation
          float* gRs = gRk;
          const float rank_w = (rank_t >= 0 && rank_t < 10) ? st.first_frame_weight[GLAMR_LOSS_KP_2D] : 1.0f;
          const bool kp_frame = visible && (!kp_first || rank_t == 0);
          // The homogeneous image point is affine in the cached joint: h = K (Rc (Rs jl + tw) + tc) = H jl + q with
          // H = K Rc Rs and q = K (Rc tw + tc) formed once per frame; the joints accumulate dL/dH and dL/dq, which are pushed back
          // onto the camera, the orientation and the translation after the loop (same sums as the joint-by-joint chain rule).
          float A[9], H[9], q[3], gH[9], gq[3];
          rm::mat3_mul(K, Rc, A);
          rm::mat3_mul(A, Rs, H);
          for (int i = 0; i < 3; ++i) {
            q[i] = A[i * 3 + 0] * tw[0] + A[i * 3 + 1] * tw[1] + A[i * 3 + 2] * tw[2] + (K[i * 3 + 0] * M[3] + K[i * 3 + 1] * M[7] + K[i * 3 + 2] * M[11]);
            gq[i] = 0.f;
          }
          for (int k = 0; k < 9; ++k) gH[k] = 0.f;
          // one joint: project, accumulate the residual value and (weight > 0) its gradient
          auto joint = [&](const float jl[3], float kx, float ky, float wj, float sc_raw, int jout) {
            const float hx = H[0] * jl[0] + H[1] * jl[1] + H[2] * jl[2] + q[0];
            const float hy = H[3] * jl[0] + H[4] * jl[1] + H[5] * jl[2] + q[1];
            const float hz = H[6] * jl[0] + H[7] * jl[1] + H[8] * jl[2] + q[2] + 1e-8f;
            const float ihz = rm::rcp_(hz);
            const float u = hx * ihz, v = hy * ihz;
            const float du = u - kx, dv = v - ky;
            if (jout >= 0) {
              s.kp_2d_pred[((size_t)t * NJ + jout) * 2 + 0] = u;
              s.kp_2d_pred[((size_t)t * NJ + jout) * 2 + 1] = v;
              if (on(GLAMR_LOSS_KP_2D_DIST) && sc_raw > min_conf && (!ffo(GLAMR_LOSS_KP_2D_DIST) || t == 0)) {
                if (last) lsum[GLAMR_LOSS_KP_2D_DIST] += sqrtf(du * du + dv * dv);
                if (last) kp_dist_cnt += 1.0f;
              }
            }
            if (wj == 0.f) return;
            if (last) lsum[GLAMR_LOSS_KP_2D] += (gmof(du, 1e4f) + gmof(dv, 1e4f)) * wj;
            if (w_kp == 0.f) return;
            const float gu = gmof_d(du, 1e4f) * wj * w_kp, gv = gmof_d(dv, 1e4f) * wj * w_kp;
            const float gh[3] = {gu * ihz, gv * ihz, -(gu * u + gv * v) * ihz};          // u = hx / hz, v = hy / hz
            for (int i = 0; i < 3; ++i) {
              for (int k = 0; k < 3; ++k) gH[i * 3 + k] += gh[i] * jl[k];
              gq[i] += gh[i];
            }
            if (last && jout >= 0 && s.g_j_local)                                        // h = H jl + q: dL/d jl = H^T gh
              for (int k = 0; k < 3; ++k) s.g_j_local[((size_t)t * NJ + jout) * 3 + k] = H[0 * 3 + k] * gh[0] + H[1 * 3 + k] * gh[1] + H[2 * 3 + k] * gh[2];
          };
          // dL/dH, dL/dq -> camera [Rc | tc], world rotation Rs, world translation tw
          auto push_back = [&]() {
            float gA[9], gRc[9];
            for (int k = 0; k < 9; ++k) { gA[k] = 0.f; gRc[k] = 0.f; }
            rm::mat3_mul_bwd(A, Rs, gH, gA, gRs);                                     // H = A Rs
            for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) gA[i * 3 + k] += gq[i] * tw[k];      // q = A tw + K tc
            float gt[3], gc[3];
            rm::mat3T_vec(A, gq, gt);
            rm::mat3T_vec(K, gq, gc);
            rm::mat3_mul_bwd(K, Rc, gA, nullptr, gRc);                                // A = K Rc
            for (int i = 0; i < 3; ++i) {
              g_tw[i] += gt[i];
              for (int k = 0; k < 3; ++k) gC[i * 4 + k] += gRc[i * 3 + k];
              gC[i * 4 + 3] += gc[i];
            }
          };
          if (last) {
            if (s.g_j_local) for (int k = 0; k < NJ * 3; ++k) s.g_j_local[(size_t)t * NJ * 3 + k] = 0.f;
            // all 26 joints from the full arrays: the projections are an output (kp_2d_pred) and feed the monitor term
            for (int j = 0; j < NJ; ++j) {
              const float sc_raw = c.kp_score[(size_t)t * NJ + j];
              float wj = 0.f;
              if (kp_frame && on(GLAMR_LOSS_KP_2D)) {
                // with first_frame_only the first visible frame's residual is multiplied by the sum of score^2 over ALL visible
                // frames (broadcast quirk, loss_func.py:27-33)
                if (kp_first) wj = s.kp_wsum[j] * rank_w;
                else { const float scj = sc_raw < min_conf ? 0.f : sc_raw; wj = scj * scj * rank_w; }
              }
              const float* jl = c.j_local + ((size_t)t * NJ + j) * 3;
              joint(jl, c.kp_2d[((size_t)t * NJ + j) * 2 + 0], c.kp_2d[((size_t)t * NJ + j) * 2 + 1], wj, sc_raw, j);
            }
          } else {
            // only the joints that carry a score (14 of 26 for HybrIK input), from the compact table
            // the next joint's six values are fetched while the current one is processed (the rows that do not fit on chip come
            // from the workspace: hundreds of cycles each with one or two waves per SIMD to hide them)
            const int njc = s.njc, nf = s.njc_fast, TMs = sh.TM;
            // A ring of KP_DEPTH joints in flight: a row from the workspace takes 300-400 ns to arrive and a joint ~120 ns to process
            constexpr int KD = GLAMR_KP_DEPTH;
            float nx[KD][6];
            // (two loops, one per memory: a pointer selected between the arena and the workspace would be a generic one -- flat loads
            // with 64-bit per-lane addresses that wait on both memory counters)
            auto fetch_chip = [&](int jj, float (&dst)[6]) {
              const float* o = s.kpc + (size_t)jj * 6 * TMs + t;
              for (int k = 0; k < 6; ++k) dst[k] = o[k * TMs];
            };
            auto fetch_ws = [&](int jj, float (&dst)[6]) {
              const float* o = s.kpc_ws + (size_t)jj * 6 * TMs + t;
              for (int k = 0; k < 6; ++k) dst[k] = o[k * TMs];
            };
            auto run = [&](int lo, int hi, auto fetch) {
#pragma unroll
              for (int d = 0; d < KD; ++d) if (lo + d < hi) fetch(lo + d, nx[d]);
              for (int jj = lo; jj < hi; jj += KD) {
#pragma unroll
                for (int d = 0; d < KD; ++d) {
                  if (jj + d >= hi) break;
                  const float cur[6] = {nx[d][0], nx[d][1], nx[d][2], nx[d][3], nx[d][4], nx[d][5]};
                  if (jj + d + KD < hi) fetch(jj + d + KD, nx[d]);
                  if (cur[5] == 0.f) continue;
                  joint(cur, cur[3], cur[4], cur[5], 0.f, -1);
                }
              }
            };
#if GLAMR_KP_GROUP > 0
            // Workspace rows in GROUPS, the next group requested before the current one is processed, and no branch inside a group: around
            // a loop back edge (or any merge) the compiler cannot count the loads in flight and waits for ALL of them -- with the one-row
            // ring above every workspace row cost a full memory round trip (0.3 us); a whole group of arithmetic now covers it.
            constexpr int KG = GLAMR_KP_GROUP;
            (void)run;
            if (w_kp != 0.f) {
              // Every multiply-add of a row is written out (fused where fma_ says so, nowhere else): the compiler contracts a * b + c at its
              // own discretion PER COPY of this code, and the copies (on-chip rows, the two group buffers) must agree to the bit -- which
              // rows fit the arena must not change the result (test_shared_cu_arena_gives_the_same_values).
              auto joint_nb = [&](const float (&cur)[6], bool in_range) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
                if (kp_fold) {
                // 43 instead of 52 operations per joint: the homogeneous point as three fused chains started from q, the constant factor 2e8 in the
                // table's weight column, rows outside the range through a zero WEIGHT (one select) instead of three selected products
                const float wgt = (in_range ? cur[5] : 0.f) * w_kp;
                const float hx = rm::fma_(H[2], cur[2], rm::fma_(H[1], cur[1], rm::fma_(H[0], cur[0], q[0])));
                const float hy = rm::fma_(H[5], cur[2], rm::fma_(H[4], cur[1], rm::fma_(H[3], cur[0], q[1])));
                const float hz = rm::fma_(H[8], cur[2], rm::fma_(H[7], cur[1], rm::fma_(H[6], cur[0], q[2]))) + 1e-8f;
                const float ihz = rm::rcp_(hz);
                const float u = hx * ihz, v = hy * ihz;
                const float du = u - cur[3], dv = v - cur[4];
                const float ru = rm::rcp_(rm::fma_(du, du, 1e4f)), rv = rm::rcp_(rm::fma_(dv, dv, 1e4f));
                const float gu = ((du * ru) * ru) * wgt, gv = ((dv * rv) * rv) * wgt;
                const float g2 = -rm::fma_(gu, u, gv * v);
                const float gh[3] = {gu * ihz, gv * ihz, g2 * ihz};
                for (int i = 0; i < 3; ++i) {
                  for (int k = 0; k < 3; ++k) gH[i * 3 + k] = rm::fma_(gh[i], cur[k], gH[i * 3 + k]);
                  gq[i] = gq[i] + gh[i];
                }
                return;
                }
                const float hx = rm::fma_(H[2], cur[2], rm::fma_(H[1], cur[1], H[0] * cur[0])) + q[0];
                const float hy = rm::fma_(H[5], cur[2], rm::fma_(H[4], cur[1], H[3] * cur[0])) + q[1];
                const float hz = (rm::fma_(H[8], cur[2], rm::fma_(H[7], cur[1], H[6] * cur[0])) + q[2]) + 1e-8f;
                const float ihz = rm::rcp_(hz);
                const float u = hx * ihz, v = hy * ihz;
                const float du = u - cur[3], dv = v - cur[4];
                const float ru = rm::rcp_(rm::fma_(du, du, 1e4f)), rv = rm::rcp_(rm::fma_(dv, dv, 1e4f));      // gmof_d(x, 1e4) = 2e8 x / (1e4 + x^2)^2
                const float gu = ((((2e8f * du) * ru) * ru) * cur[5]) * w_kp, gv = ((((2e8f * dv) * rv) * rv) * cur[5]) * w_kp;
                const bool live = in_range && cur[5] != 0.f;          // a select, not a branch: rows without weight add exact zeros
                const float g2 = -rm::fma_(gu, u, gv * v);
                const float gh[3] = {live ? gu * ihz : 0.f, live ? gv * ihz : 0.f, live ? g2 * ihz : 0.f};
                for (int i = 0; i < 3; ++i) {
                  for (int k = 0; k < 3; ++k) gH[i * 3 + k] = rm::fma_(gh[i], cur[k], gH[i * 3 + k]);
                  gq[i] = gq[i] + gh[i];
                }
              };
              // on-chip rows ONE AHEAD (two buffers, the loop unrolled twice): a row's six ds_reads are in flight while the previous row is
              // processed -- fetched and consumed in the same trip, every row exposed an LDS round trip (12 rows x ~100 cycles per iteration)
              float ca[6], cb[6];
              if (nf > 0) fetch_chip(0, ca);
              for (int jj = 0; jj < nf; jj += 2) {
                if (jj + 1 < nf) fetch_chip(jj + 1, cb);
                joint_nb(ca, true);
                if (jj + 1 >= nf) break;
                if (jj + 2 < nf) fetch_chip(jj + 2, ca);
                joint_nb(cb, true);
              }
              float ga[KG][6], gb[KG][6];
              auto fetch_group = [&](int base, float (&b)[KG][6]) {
#pragma unroll
                for (int d = 0; d < KG; ++d) fetch_ws(base + d < njc ? base + d : njc - 1, b[d]);
              };
              auto do_group = [&](int base, float (&b)[KG][6]) {
#pragma unroll
                for (int d = 0; d < KG; ++d) joint_nb(b[d], base + d < njc);
              };
              int base = nf;
              if (base < njc) fetch_group(base, ga);
              while (base < njc) {
                if (base + KG < njc) fetch_group(base + KG, gb);
                do_group(base, ga);
                base += KG;
                if (base >= njc) break;
                if (base + KG < njc) fetch_group(base + KG, ga);
                do_group(base, gb);
                base += KG;
              }
            }
#else
            run(0, nf, fetch_chip);
            run(nf, njc, fetch_ws);
#endif
          }
          if (w_kp != 0.f && kp_frame) push_back();
        }
        GLAMR_MARK(rt, 10);
        // smoothness of the world orientation in 6D  (loss_func.py:117-132)
        if (on(GLAMR_LOSS_TRAJ_ROT_SMOOTHNESS)) {
          const float* d0 = s.d6 + t * 6;
          float gd6[6] = {0, 0, 0, 0, 0, 0};
          const int tn = t + 1 < T ? t + 1 : t, tp = t > 0 ? t - 1 : t;
          float dn[6], dp[6];
          for (int k = 0; k < 6; ++k) { dn[k] = s.d6[tn * 6 + k]; dp[k] = s.d6[tp * 6 + k]; }
          if (t + 1 < T) for (int k = 0; k < 6; ++k) { const float v = (dn[k] - d0[k]) * FPS; if (last) lsum[GLAMR_LOSS_TRAJ_ROT_SMOOTHNESS] += v * v; gd6[k] -= 2.0f * FPS * v * w_trs; }
          if (t > 0) for (int k = 0; k < 6; ++k) { const float v = (d0[k] - dp[k]) * FPS; gd6[k] += 2.0f * FPS * v * w_trs; }
          for (int r = 0; r < 3; ++r) { gRk[r * 3 + 0] += gd6[r]; gRk[r * 3 + 1] += gd6[3 + r]; }
        }
        float g6[6];
        fold_R_grad(s.d6 + t * 6, gRk, g6);
        for (int k = 0; k < 6; ++k) s.g_d6[t * 6 + k] = g6[k];
        for (int k = 0; k < 3; ++k) s.g_tw[t * 3 + k] = g_tw[k];
      }
      GLAMR_MARK(rt, 11);
      // the camera's parameter state is requested here: the smoothness terms below cover the latency
      AdamRegs<6> a_rot;
      AdamRegs<3> a_tr;
      if (own_cam) { a_rot.load(sh.cp, sh.cm, sh.cv, lo.cam_rot6d + t * 6); a_tr.load(sh.cp, sh.cm, sh.cv, lo.cam_trans + t * 3); }
      // camera-only terms on the camera-to-world transform  (loss_func.py:76-114)
      // the neighbouring frames' camera-to-world transforms: two whole rows up front (clamped at the ends) instead of eighteen conditional
      // element reads -- each of those was a branch of its own with an LDS round trip inside (0.75 of this phase's 1.1 us)
      float Cn[12], Cp[12], Mo[12];      // (Mo: this frame's own row, in registers -- `Mi[..]` inside a condition is again a load behind a branch)
      if (on(GLAMR_LOSS_CAM_INV_ROT_SMOOTHNESS) || on(GLAMR_LOSS_CAM_ORIGIN_SMOOTHNESS)) {
        const int tn = t + 1 < T ? t + 1 : t, tp = t > 0 ? t - 1 : t;
        for (int k = 0; k < 12; ++k) { Cn[k] = sh.cam_inv[(size_t)tn * 12 + k]; Cp[k] = sh.cam_inv[(size_t)tp * 12 + k]; Mo[k] = Mi[k]; }
      }
      if (on(GLAMR_LOSS_CAM_INV_ROT_SMOOTHNESS)) {
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 2; ++j) {
            if (t + 1 < T) { const float v = (Mo[i * 4 + j] - Cn[i * 4 + j]) * FPS; if (last) lsum[GLAMR_LOSS_CAM_INV_ROT_SMOOTHNESS] += v * v; gCi[i * 4 + j] += 2.0f * FPS * v * w_crs; }
            if (t > 0) { const float v = (Cp[i * 4 + j] - Mo[i * 4 + j]) * FPS; gCi[i * 4 + j] -= 2.0f * FPS * v * w_crs; }
          }
      }
      if (on(GLAMR_LOSS_CAM_ORIGIN_SMOOTHNESS)) {
        for (int i = 0; i < 3; ++i) {
          if (t + 1 < T) { const float v = (Cn[i * 4 + 3] - Mo[i * 4 + 3]) * FPS; if (last) lsum[GLAMR_LOSS_CAM_ORIGIN_SMOOTHNESS] += v * v; gCi[i * 4 + 3] -= 2.0f * FPS * v * w_cos; }
          if (t > 0) { const float v = (Mo[i * 4 + 3] - Cp[i * 4 + 3]) * FPS; gCi[i * 4 + 3] += 2.0f * FPS * v * w_cos; }
        }
      }
      if (on(GLAMR_LOSS_CAM_UP_REG) && (!ffo(GLAMR_LOSS_CAM_UP_REG) || t == 0)) {
        const float fw = t < 10 ? st.first_frame_weight[GLAMR_LOSS_CAM_UP_REG] : 1.0f;
        if (last) lsum[GLAMR_LOSS_CAM_UP_REG] += Mi[2 * 4 + 1] * fw;
        gCi[2 * 4 + 1] += fw * w_up;
      }
      GLAMR_MARK(rt, 12);
      if (cam_from_person) {
        for (int k = 0; k < 12; ++k) { sh.g_cam[(size_t)t * 12 + k] = gC[k]; sh.g_caminv[(size_t)t * 12 + k] = gCi[k]; }
      } else if (update && var_cam) {
        // ---- G (own camera parameters): this frame's gradient is complete, no other thread contributes -------------------
        const int row = fixed_cam ? 0 : t;
        invert34_bwd(M, gCi, gC);
        float gR[9], g6[6] = {0, 0, 0, 0, 0, 0};
        get_R(gC, gR);
        if (fixed_cam) {
          rm::rot6d_to_rotmat_bwd(sh.cp + lo.cam_rot6d + row * 6, gR, g6);
          for (int k = 0; k < 6; ++k) gfix[k] += g6[k];
          for (int k = 0; k < 3; ++k) gfix[6 + k] += gC[k * 4 + 3];
        } else {
          rm::rot6d_to_rotmat_bwd(a_rot.P, gR, g6);
          const float gt3[3] = {gC[3], gC[7], gC[11]};
          adam_step2(a_rot, g6, a_tr, gt3, ac);
          a_rot.store(sh.cp, sh.cm, sh.cv, sh.store_grad ? sh.cg : nullptr, lo.cam_rot6d + t * 6, g6);
          a_tr.store(sh.cp, sh.cm, sh.cv, sh.store_grad ? sh.cg : nullptr, lo.cam_trans + t * 3, gt3);
        }
      }
      if (last && on(GLAMR_LOSS_CAM_INV_TRANS_RES_REG))
        for (int k = 0; k < 3; ++k) {      // (with the camera optimised the on-chip block holds the camera, the residuals stay in the batch array)
          const float r = (var_cam ? sh.cpg[l.cam_inv_trans_res + t * 3 + k] : sh.cp[lo.cam_inv_trans_res - rs + t * 3 + k]) * FPS;
          if (last) lsum[GLAMR_LOSS_CAM_INV_TRANS_RES_REG] += r * r;
        }
    }
    // relative transform between persons on co-visible frames (loss_func.py:248-271), in a loop of its own: its ~100 live values
    // are then not allocated alongside the keypoint / camera terms above.  Own frame only -> no barrier; the fold is linear.
    if (!SINGLE && on(GLAMR_LOSS_REL_TRANSFORM) && P > 1)
      for (int t = rt.tid(); frame_in(t, T); t += fstep)
        for (int p = 0; p < P; ++p) {
          const PersonView s = pv(p);
          if (s.vis[t] == 0.f || s.frozen) continue;
          float Rk[9], gRk[9], g_tw[3] = {0, 0, 0}, g6[6];
          for (int k = 0; k < 9; ++k) gRk[k] = 0.f;
          cols_to_R(s.d6 + t * 6, Rk);
          const float v = rel_transform_term<FAST>(rt, sc, sh, st, P, p, t, Rk, s.tw + t * 3, w_rel, gRk, g_tw);
          if (last) lsum[GLAMR_LOSS_REL_TRANSFORM] += v;
          fold_R_grad(s.d6 + t * 6, gRk, g6);
          for (int k = 0; k < 6; ++k) s.g_d6[t * 6 + k] += g6[k];
          for (int k = 0; k < 3; ++k) s.g_tw[t * 3 + k] += g_tw[k];
        }
    GLAMR_MARK(rt, 4);
    if (update && var_cam && fixed_cam) {
      rt.reduce_sum_n(gfix);                             // all 9 sums with one exchange (nine separate block reductions were 18 barriers per iteration)
      // one LANE per parameter (every thread holds all nine sums): thread 0 doing them one after the other was nine dependent round trips to the
      // parameter block -- global memory for scenes of several persons: 11 us of a 93 us iteration of BASELINE configs[3]
      for (int k = rt.tid(); k < 9; k += rt.nthreads()) {
        float gk = gfix[0];
#pragma unroll
        for (int q = 1; q < 9; ++q) gk = (k == q) ? gfix[q] : gk;
        const int i = k < 6 ? lo.cam_rot6d + k : lo.cam_trans + (k - 6);
        if (sh.store_grad) sh.cg[i] = gk;
        adam(sh.cp[i], sh.cm[i], sh.cv[i], gk, ac);
      }
    } else if (update && cam_from_person) {
      // ---- G (camera derived from the persons): gradient of every frame's averaged transform, folded onto its source frame ----
      rt.sync();
      for (int t = rt.tid(); frame_in(t, T); t += fstep) {
        float gMi[12];
        for (int k = 0; k < 12; ++k) gMi[k] = sh.g_caminv[(size_t)t * 12 + k];
        invert34_bwd(sh.cam_inv + (size_t)t * 12, sh.g_cam + (size_t)t * 12, gMi);     // cam_pose = invert(cam_inv)
        // cam_inv = [6d->R(r6) | avg_t + res]; r6 = avg cols (+ res on empty frames)
        const int src = sh.fill_src[t];
        float avg6[6];
        {
          float avg[12];
          for (int k = 0; k < 12; ++k) avg[k] = 0.f;
          for (int p = 0; p < P; ++p) {
            if (pv(p).vis[src] == 0.f) continue;
            float Tw[12], Rk[9], C[12];
            cols_to_R(pv(p).d6 + src * 6, Rk);
            for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Tw[i * 4 + j] = Rk[i * 3 + j]; Tw[i * 4 + 3] = pv(p).tw[src * 3 + i]; }
            mul34(Tw, pv(p).person2cam + (size_t)src * 12, C);
            for (int k = 0; k < 12; ++k) avg[k] += C[k];
          }
          const float inv_n = 1.0f / (float)sh.n_vis_persons[src];
          for (int r = 0; r < 3; ++r) { avg6[r] = avg[r * 4 + 0] * inv_n; avg6[3 + r] = avg[r * 4 + 1] * inv_n; }
          if (sh.n_vis_persons[t] == 0) for (int k = 0; k < 6; ++k) avg6[k] += sh.cp[lo.cam_inv_rot_res - rs + t * 6 + k];
        }
        float gR[9], g6[6] = {0, 0, 0, 0, 0, 0};
        get_R(gMi, gR);
        rm::rot6d_to_rotmat_bwd(avg6, gR, g6);
        float* ga = sh.g_avg + (size_t)t * 12;
        for (int r = 0; r < 3; ++r) { ga[r * 4 + 0] = g6[r]; ga[r * 4 + 1] = g6[3 + r]; ga[r * 4 + 2] = 0.f; ga[r * 4 + 3] = gMi[r * 4 + 3]; }
        float g_tres[3] = {gMi[3], gMi[7], gMi[11]};
        if (cam_terms && active(GLAMR_LOSS_CAM_INV_TRANS_RES_REG)) {
          const float wreg = st.loss_weight[GLAMR_LOSS_CAM_INV_TRANS_RES_REG] / (float)T;
          for (int k = 0; k < 3; ++k) g_tres[k] += 2.0f * FPS * FPS * sh.cp[lo.cam_inv_trans_res - rs + t * 3 + k] * wreg;
        }
        for (int k = 0; k < 3; ++k) { const int i = lo.cam_inv_trans_res + t * 3 + k, j = i - lo.cam_inv_rot_res; if (sh.store_grad) sh.cg[i] = g_tres[k]; adam(sh.cp[i - rs], sh.cm[j], sh.cv[j], g_tres[k], ac); }
        if (sh.n_vis_persons[t] == 0)
          for (int k = 0; k < 6; ++k) { const int i = lo.cam_inv_rot_res + t * 6 + k, j = i - lo.cam_inv_rot_res; if (sh.store_grad) sh.cg[i] = g6[k]; adam(sh.cp[i - rs], sh.cm[j], sh.cv[j], g6[k], ac); }
      }
      rt.sync();
      for (int t = rt.tid(); frame_in(t, T); t += fstep) {
        if (sh.n_vis_persons[t] == 0) continue;                 // only frames with persons are sources
        float ga[12];
        for (int k = 0; k < 12; ++k) ga[k] = 0.f;
        for (int u = 0; u < T; ++u) {
          if (sh.fill_src[u] != t) continue;
          for (int k = 0; k < 12; ++k) ga[k] += sh.g_avg[(size_t)u * 12 + k];
        }
        const float inv_n = 1.0f / (float)sh.n_vis_persons[t];
        for (int k = 0; k < 12; ++k) ga[k] *= inv_n;
        for (int p = 0; p < P; ++p) {
          if (pv(p).vis[t] == 0.f) continue;
          float Tw[12], Rk[9], gTw[12];
          for (int k = 0; k < 12; ++k) gTw[k] = 0.f;
          cols_to_R(pv(p).d6 + t * 6, Rk);
          for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Tw[i * 4 + j] = Rk[i * 3 + j]; Tw[i * 4 + 3] = pv(p).tw[t * 3 + i]; }
          mul34_bwd(Tw, pv(p).person2cam + (size_t)t * 12, ga, gTw, nullptr);
          float gRk[9], g6[6];
          get_R(gTw, gRk);
          fold_R_grad(pv(p).d6 + t * 6, gRk, g6);
          for (int k = 0; k < 6; ++k) pv(p).g_d6[t * 6 + k] += g6[k];
          for (int k = 0; k < 3; ++k) pv(p).g_tw[t * 3 + k] += gTw[k * 4 + 3];
        }
      }
      // the fold writes g_d6 / g_tw of frame t from thread t only: the owner continues without a barrier
    } else if (update && !var_cam && cam_terms && active(GLAMR_LOSS_CAM_INV_TRANS_RES_REG)) {
      // camera neither optimised nor derived from the persons: the residual only feels its own regulariser
      const float wreg = st.loss_weight[GLAMR_LOSS_CAM_INV_TRANS_RES_REG] / (float)T;
      for (int t = rt.tid(); frame_in(t, T); t += fstep)
        for (int k = 0; k < 3; ++k) {
          const int i = lo.cam_inv_trans_res + t * 3 + k;
          const float g = 2.0f * FPS * FPS * sh.cp[i - rs] * wreg;
          if (sh.store_grad) sh.cg[i] = g;
          adam(sh.cp[i - rs], sh.cm[i - lo.cam_inv_rot_res], sh.cv[i - lo.cam_inv_rot_res], g, ac);
        }
    }
    GLAMR_MARK(rt, 5);
    // ---- H: reverse of the orientation chain; direct parameter gradients (own frame only) -----------------------------------
    const float w_rot = active(GLAMR_LOSS_LOCAL_ROT_REG) ? st.loss_weight[GLAMR_LOSS_LOCAL_ROT_REG] / n_exist : 0.f;
    const float w_z = active(GLAMR_LOSS_LOCAL_Z_REG) ? st.loss_weight[GLAMR_LOSS_LOCAL_Z_REG] / n_exist : 0.f;
    const float w_dxy = active(GLAMR_LOSS_LOCAL_DXY_REG) ? st.loss_weight[GLAMR_LOSS_LOCAL_DXY_REG] / n_exist_m1 : 0.f;
    const float w_dh = active(GLAMR_LOSS_LOCAL_DHEADING_REG_NEW) ? st.loss_weight[GLAMR_LOSS_LOCAL_DHEADING_REG_NEW] / n_exist_m1 : 0.f;
    float gth_r[1] = {0.f}, gxy_r[2] = {0.f, 0.f};      // REGB instances: this frame's heading / planar-position gradient (scan operands and results)
    for (int t = rt.tid(); frame_in(t, T); t += fstep)
      for (int p = 0; p < P; ++p) {
        const PersonView c = pv(p);
        const PersonView& s = c;
        const bool ex = t >= c.fr_start && t < c.fr_end;
        const int e = t - c.fr_start;
        // parameter state of this frame, requested before anything is computed
        const bool upd_wd = has_wd && update && (st.var_mask & GLAMR_VAR_WORLD_DHEADING);
        const bool upd_rot = ex && update && (st.var_mask & GLAMR_VAR_LOCAL_ROT);
        const bool upd_z = ex && update && (st.var_mask & GLAMR_VAR_LOCAL_Z);
        AdamRegs<1> a_wd;
        AdamRegs<6> a_rot;
        AdamRegs<1> a_z;
        a_wd.zero(); a_rot.zero(); a_z.zero();      // (a group that is not updated rides along in the common pass below with zeros)
        if (upd_wd) a_wd.load(s.p, s.m, s.v, lo.world_dheading + t);
        if (upd_rot) a_rot.load(s.p, s.m, s.v, lo.local_rot + e * 6); else if (ex) for (int k = 0; k < 6; ++k) a_rot.P[k] = s.p[lo.local_rot + e * 6 + k];
        if (upd_z) a_z.load(s.p, s.m, s.v, lo.local_z + e); else if (ex) a_z.P[0] = s.p[lo.local_z + e];
        // R_w = Rz(phi) [b2 b3 | .]:  d/dphi = J R_w with J = [[0,-1,0],[1,0,0],[0,0,0]]
        const float* d6 = s.d6 + t * 6;
        const float* g6 = s.g_d6 + t * 6;
        const float gphi = (g6[1] * d6[0] - g6[0] * d6[1]) + (g6[4] * d6[3] - g6[3] * d6[4]);
        const float gw[1] = {gphi};
        if (!ex) {
          if (upd_wd) a_wd.step_store(s.p, s.m, s.v, sh.store_grad ? s.g : nullptr, lo.world_dheading + t, gw, ac);
          if (!REGB) { s.g_theta[t] = 0.f; s.g_xy[t * 2 + 0] = 0.f; s.g_xy[t * 2 + 1] = 0.f; }
          continue;
        }
        const LocalRow L = load_row(s.Lc, sh.TM, t);
        const float cs = s.Lc[10 * sh.TM + t], sn = s.Lc[11 * sh.TM + t];
        const float gth = gphi;
        float gr6[6] = {0, 0, 0, 0, 0, 0};
        {
          const float g1[3] = {g6[0], g6[1], g6[2]}, g2[3] = {g6[3], g6[4], g6[5]};
          float gb2[3], gb3[3], gRl[9];
          rotz2T(cs, sn, g1, gb2);
          rotz2T(cs, sn, g2, gb3);
          for (int r = 0; r < 3; ++r) { gRl[r * 3 + 0] = 0.f; gRl[r * 3 + 1] = gb2[r]; gRl[r * 3 + 2] = gb3[r]; }
          rm::rot6d_to_rotmat_bwd(L.r6, gRl, gr6);
        }
        if (REGB) { gth_r[0] = gth; gxy_r[0] = s.g_tw[t * 3 + 0]; gxy_r[1] = s.g_tw[t * 3 + 1]; } else {
          s.g_theta[t] = gth;
          s.g_xy[t * 2 + 0] = s.g_tw[t * 3 + 0];
          s.g_xy[t * 2 + 1] = s.g_tw[t * 3 + 1];
        }
        // local_rot / local_z: gradient is final here (+ regularisers loss_func.py:189-237); the frame's eight updates (world heading
        // offset, local rotation, height) run as one interleaved pass
        {
          float g6r[6];
          for (int k = 0; k < 6; ++k) {
            const float r = a_rot.P[k] * FPS;
            if (on(GLAMR_LOSS_LOCAL_ROT_REG)) if (last) lsum[GLAMR_LOSS_LOCAL_ROT_REG] += r * r;
            g6r[k] = gr6[k] + 2.0f * FPS * r * w_rot;
          }
          const float r = a_z.P[0] * FPS;
          if (on(GLAMR_LOSS_LOCAL_Z_REG)) if (last) lsum[GLAMR_LOSS_LOCAL_Z_REG] += r * r;
          const float gz[1] = {s.g_tw[t * 3 + 2] + 2.0f * FPS * r * w_z};
          GLAMR_MARK(rt, 13);
          if (upd_wd || upd_rot || upd_z) adam_step3(a_wd, gw, a_rot, g6r, a_z, gz, ac);
          if (upd_wd) a_wd.store(s.p, s.m, s.v, sh.store_grad ? s.g : nullptr, lo.world_dheading + t, gw);
          if (upd_rot) a_rot.store(s.p, s.m, s.v, sh.store_grad ? s.g : nullptr, lo.local_rot + e * 6, g6r);
          if (upd_z) a_z.store(s.p, s.m, s.v, sh.store_grad ? s.g : nullptr, lo.local_z + e, gz);
        }
      }
    GLAMR_MARK(rt, 14);
    if (update) {
      if constexpr (REGB) {
        rt.template scan_regs<NWC, 2>(gxy_r, true, scan_shuffle);
        for (int t = rt.tid(); frame_in(t, T); t += fstep) { pv(0).g_xy[t * 2 + 0] = gxy_r[0]; pv(0).g_xy[t * 2 + 1] = gxy_r[1]; }      // (frame t + 1's is read by frame t)
      } else {
        float* ch[16];
        if constexpr (MAXP <= 8) {
          for (int p = 0; p < P; ++p) { ch[2 * p] = pv(p).g_xy; ch[2 * p + 1] = pv(p).g_xy + 1; }
          rt.template scan_multi<(FAST != 0)>(ch, 2 * P, T, 2, true, scan_shuffle);
        } else {
          for (int p0 = 0; p0 < P; p0 += 8) {
            const int np = P - p0 < 8 ? P - p0 : 8;
            for (int p = 0; p < np; ++p) { ch[2 * p] = pv(p0 + p).g_xy; ch[2 * p + 1] = pv(p0 + p).g_xy + 1; }
            rt.template scan_multi<(FAST != 0)>(ch, 2 * np, T, 2, true, scan_shuffle);
          }
        }
      }
      rt.sync();
      GLAMR_MARK(rt, 6);
      // ---- I: reverse of B ----------------------------------------------------------------------------------------------
      for (int t = rt.tid(); frame_in(t, T); t += fstep)
        for (int p = 0; p < P; ++p) {
          const PersonView c = pv(p);
          const PersonView& s = c;
          if (t < c.fr_start || t >= c.fr_end) continue;
          const int e = t - c.fr_start, n = c.fr_end - c.fr_start;
          // frame 0 updates local_xy where the others update their row of local_dxy: a selected INDEX, one code path (the wave that holds
          // frame 0 used to walk a branch of its own with two more updates straight on the arrays)
          const bool first = e == 0;
          const int ixy = first ? lo.local_xy : lo.local_dxy + e * 2;
          const bool upd_dxy = first ? (bool)(st.var_mask & GLAMR_VAR_LOCAL_XY) : (bool)(st.var_mask & GLAMR_VAR_LOCAL_DXY);
          AdamRegs<2> a_dxy;
          if (upd_dxy) a_dxy.load(s.p, s.m, s.v, ixy); else for (int k = 0; k < 2; ++k) a_dxy.P[k] = s.p[ixy + k];
          // contribution of d[e+1] = Rot(theta[e]) L[e+1].xy to g_theta[e]   (every neighbour value fetched up front, clamped index)
          const int tn = e + 1 < n ? t + 1 : t, tp = e > 0 ? t - 1 : t;
          struct { float dx, dy; } Ln = {s.Lc[0 * sh.TM + tn], s.Lc[1 * sh.TM + tn]};
          const float cs0 = s.csn[t * 2 + 0], sn0 = s.csn[t * 2 + 1];
          const float gdx = s.g_xy[tn * 2 + 0], gdy = s.g_xy[tn * 2 + 1];
          const float csp = s.csn[tp * 2 + 0], snp = s.csn[tp * 2 + 1];
          if (e + 1 < n) {
            const float cs = cs0, sn = sn0;
            if (REGB) gth_r[0] += gdx * (-Ln.dx * sn - Ln.dy * cs) + gdy * (Ln.dx * cs - Ln.dy * sn);
            else s.g_theta[t] += gdx * (-Ln.dx * sn - Ln.dy * cs) + gdy * (Ln.dx * cs - Ln.dy * sn);
          }
          float gx = REGB ? gxy_r[0] : s.g_xy[t * 2 + 0], gy = REGB ? gxy_r[1] : s.g_xy[t * 2 + 1];
          if (e > 0) {
            const float cs = csp, sn = snp;
            const float a = gx * cs + gy * sn, b = -gx * sn + gy * cs;
            gx = a; gy = b;
          }
          const float g[2] = {gx, gy};
          {
            float g2[2];
            for (int k = 0; k < 2; ++k) {
              const float r = a_dxy.P[k] * FPS;
              if (on(GLAMR_LOSS_LOCAL_DXY_REG)) if (last && !first) lsum[GLAMR_LOSS_LOCAL_DXY_REG] += r * r;
              const float greg = g[k] + 2.0f * FPS * r * w_dxy;
              g2[k] = first ? g[k] : greg;                       // (local_xy has no regulariser)
            }
            if (upd_dxy) a_dxy.step_store(s.p, s.m, s.v, sh.store_grad ? s.g : nullptr, ixy, g2, ac);
          }
        }
      // REGB instances: the state of this frame's heading parameter (own data) is requested before the scan's barrier, not after it
      AdamRegs<1> a_hpre;
      a_hpre.zero();
      float dmask_pre = 0.f;      // (and its row of the heading mask: a load from the workspace that sat right in front of its use)
      if (REGB)
        for (int t = rt.tid(); frame_in(t, T); t += fstep) {
          const PersonView c = pv(0);
          if (t < c.fr_start || t >= c.fr_end) continue;
          const int e = t - c.fr_start;
          const int i = e == 0 ? lo.local_heading : lo.local_dheading + e;
          if (e == 0 ? (bool)(st.var_mask & GLAMR_VAR_LOCAL_HEADING) : (bool)(st.var_mask & GLAMR_VAR_LOCAL_DHEADING)) a_hpre.load(c.p, c.m, c.v, i);
          else a_hpre.P[0] = c.p[i];
          if (c.dheading_mask) dmask_pre = c.dheading_mask[e];
        }
      if constexpr (REGB) {
        rt.template scan_regs<NWC, 1>(gth_r, true, scan_shuffle);
      } else
#ifdef GLAMR_GRECON_WIDE      // (absolute_heading: a frame's heading gradient is its own)
      if (st.flags & GLAMR_FLAG_ABSOLUTE_HEADING) rt.sync(); else
#endif
      {
        float* ch[8];
        if constexpr (MAXP <= 8) {
          for (int p = 0; p < P; ++p) ch[p] = pv(p).g_theta;
          rt.template scan_multi<(FAST != 0)>(ch, P, T, 1, true, scan_shuffle);
        } else {
          for (int p0 = 0; p0 < P; p0 += 8) {
            const int np = P - p0 < 8 ? P - p0 : 8;
            for (int p = 0; p < np; ++p) ch[p] = pv(p0 + p).g_theta;
            rt.template scan_multi<(FAST != 0)>(ch, np, T, 1, true, scan_shuffle);
          }
        }
      }
      GLAMR_MARK(rt, 7);
      // ---- J: reverse of A (own element of the suffix sum) ------------------------------------------------------------------
      for (int t = rt.tid(); frame_in(t, T); t += fstep)
        for (int p = 0; p < P; ++p) {
          const PersonView c = pv(p);
          const PersonView& s = c;
          if (t < c.fr_start || t >= c.fr_end) continue;
          const int e = t - c.fr_start;
          const float gh = REGB ? gth_r[0] : s.g_theta[t];
          // frame 0 updates local_heading where the others update their row of local_dheading: selected index, one update (see phase I)
          const bool first = e == 0;
          const int i = first ? lo.local_heading : lo.local_dheading + e;
          const bool upd = first ? (bool)(st.var_mask & GLAMR_VAR_LOCAL_HEADING) : (bool)(st.var_mask & GLAMR_VAR_LOCAL_DHEADING);
          AdamRegs<1> a_h = a_hpre;
          if (!REGB) { if (upd) a_h.load(s.p, s.m, s.v, i); else a_h.P[0] = s.p[i]; }
          float gj[1] = {gh};
          if (!first) {
            const float v = a_h.P[0];
            float sv, cv;
            rm::sincos_(v, sv, cv);
            if (on(GLAMR_LOSS_LOCAL_DHEADING_REG_NEW)) { const float a = (cv - 1.0f) * FPS, b = sv * FPS; if (last) lsum[GLAMR_LOSS_LOCAL_DHEADING_REG_NEW] += a * a + b * b; }
            const float mk = REGB ? dmask_pre : (c.dheading_mask ? c.dheading_mask[e] : 0.0f);
            const float g = (c.dheading_mask ? gh * mk : 0.0f) + 2.0f * FPS * FPS * ((cv - 1.0f) * (-sv) + sv * cv) * w_dh;
            gj[0] = g;
          }
          if (upd) a_h.step_store(s.p, s.m, s.v, sh.store_grad ? s.g : nullptr, i, gj, ac);
        }
    }
    // regulariser values that do not depend on being optimised (reported every evaluation)
    if (last) {
      for (int p = 0; p < P && !update; ++p) {
        const PersonView s = pv(p);
        const int n = pv(p).fr_end - pv(p).fr_start;
        for (int e = rt.tid() + 1; e < n; e += rt.nthreads()) {
          if (on(GLAMR_LOSS_LOCAL_DXY_REG)) for (int k = 0; k < 2; ++k) { const float r = s.p[lo.local_dxy + e * 2 + k] * FPS; if (last) lsum[GLAMR_LOSS_LOCAL_DXY_REG] += r * r; }
          if (on(GLAMR_LOSS_LOCAL_DHEADING_REG_NEW)) { float sv, cv; rm::sincos_(s.p[lo.local_dheading + e], sv, cv); const float a = (cv - 1.0f) * FPS, b = sv * FPS; if (last) lsum[GLAMR_LOSS_LOCAL_DHEADING_REG_NEW] += a * a + b * b; }
        }
      }
      // block-reduce and normalise the reported (unweighted) loss values
      float tot[GLAMR_NUM_LOSSES];
      for (int i = 0; i < GLAMR_NUM_LOSSES; ++i) tot[i] = rt.reduce_sum(lsum[i]);
      const float cnt = rt.reduce_sum(kp_dist_cnt);
      if (rt.tid() == 0) {
        float* o = sh.losses;
        o[GLAMR_LOSS_KP_2D] = tot[GLAMR_LOSS_KP_2D] / n_vis_total;
        o[GLAMR_LOSS_KP_2D_DIST] = tot[GLAMR_LOSS_KP_2D_DIST] / cnt;
        o[GLAMR_LOSS_REL_TRANSFORM] = P > 1 ? tot[GLAMR_LOSS_REL_TRANSFORM] / n_rel : 0.f;
        o[GLAMR_LOSS_CAM_TRAJ_ROT] = tot[GLAMR_LOSS_CAM_TRAJ_ROT] / n_ctr;
        o[GLAMR_LOSS_TRAJ_ROT_SMOOTHNESS] = tot[GLAMR_LOSS_TRAJ_ROT_SMOOTHNESS] / n_trs;
        o[GLAMR_LOSS_LOCAL_DXY_REG] = tot[GLAMR_LOSS_LOCAL_DXY_REG] / n_exist_m1;
        o[GLAMR_LOSS_LOCAL_DHEADING_REG_NEW] = tot[GLAMR_LOSS_LOCAL_DHEADING_REG_NEW] / n_exist_m1;
        o[GLAMR_LOSS_LOCAL_ROT_REG] = tot[GLAMR_LOSS_LOCAL_ROT_REG] / n_exist;
        o[GLAMR_LOSS_LOCAL_Z_REG] = tot[GLAMR_LOSS_LOCAL_Z_REG] / n_exist;
        o[GLAMR_LOSS_CAM_INV_TRANS_RES_REG] = tot[GLAMR_LOSS_CAM_INV_TRANS_RES_REG] / (float)T;
        o[GLAMR_LOSS_CAM_INV_ROT_SMOOTHNESS] = tot[GLAMR_LOSS_CAM_INV_ROT_SMOOTHNESS] / (float)(T - 1);
        o[GLAMR_LOSS_CAM_ORIGIN_SMOOTHNESS] = tot[GLAMR_LOSS_CAM_ORIGIN_SMOOTHNESS] / (float)(T - 1);
        o[GLAMR_LOSS_CAM_UP_REG] = tot[GLAMR_LOSS_CAM_UP_REG] / n_up;
      }
    }
    GLAMR_MARK(rt, 8);
  };
  // per-iteration loss log (write_logs, :564,646-659): every iteration runs the reporting evaluation and thread 0 copies the values it just
  // wrote.  Only the plain instances carry this loop (the launcher sends a batch with a history there): the arena instances stay as they are.
  float* const hist = FAST == 0 ? uni(sc.loss_history) : nullptr;
  auto keep_losses = [&](int it) {
    if (FAST == 0 && hist && rt.tid() == 0) for (int k = 0; k < GLAMR_NUM_LOSSES; ++k) glob(hist)[(size_t)it * GLAMR_NUM_LOSSES + k] = sh.losses[k];
  };
  // GLAMR_FLAG_NO_REPORT (launch-by-launch schedules: the gradient launch of every iteration but a stage's last): the final evaluation is an
  // update-only one as well -- no outputs, no loss values, gradients (and grads_out) as always.  Same two instantiations, one call site each.
  // (honoured by the instances for scenes of several persons with run-time layouts -- what the person-sharded schedule launches; compiled out of the
  // single-person instances, whose loops keep the code they were tuned with: with it the constant-layout instance measured 10.67 instead of 10.50 us)
  constexpr bool NR = !SINGLE && TMC == 0;
  if constexpr (NR) {
    const bool no_report = niters > 0 && (st.flags & GLAMR_FLAG_NO_REPORT) && !(FAST == 0 && hist);
    const int n_plain = no_report ? n_eval : n_eval - 1;
    if (FAST == 0 && hist) {
      for (int it = 0; it + 1 < n_eval; ++it) { evaluate(std::true_type{}); keep_losses(it); trace_hook(rt, it, sc, 0); }
    } else {
      for (int it = 0; it < n_plain; ++it) { evaluate(std::false_type{}); trace_hook(rt, it, sc, 0); }
    }
    if (!no_report) {
      evaluate(std::true_type{});
      keep_losses(n_eval - 1);
      trace_hook(rt, n_eval - 1, sc, 0);
    }
  } else {
    if (FAST == 0 && hist) {
      for (int it = 0; it + 1 < n_eval; ++it) { evaluate(std::true_type{}); keep_losses(it); trace_hook(rt, it, sc, 0); }
    } else {
      for (int it = 0; it + 1 < n_eval; ++it) { evaluate(std::false_type{}); trace_hook(rt, it, sc, 0); }
    }
    evaluate(std::true_type{});
    keep_losses(n_eval - 1);
    trace_hook(rt, n_eval - 1, sc, 0);
  }
  rt.sync();
  if (AF) {      // the on-chip parameters go back to the batch array, once
    copy_person_block(rt, sc.ps[0].p_g, l, sc.ps[0].p, lo);
    if (var_cam) {
      const int rows = fixed_cam ? 1 : T;
      for (int t = rt.tid(); t < rows; t += rt.nthreads()) {
        for (int k = 0; k < 6; ++k) sc.cp_g[l.cam_rot6d + t * 6 + k] = sc.cp[lo.cam_rot6d + t * 6 + k];
        for (int k = 0; k < 3; ++k) sc.cp_g[l.cam_trans + t * 3 + k] = sc.cp[lo.cam_trans + t * 3 + k];
      }
    } else {
      for (int i = rt.tid(); i < 6 * Tb; i += rt.nthreads()) sc.cp_g[l.cam_inv_rot_res + i] = sc.cp[i];
      for (int i = rt.tid(); i < 3 * Tb; i += rt.nthreads()) sc.cp_g[l.cam_inv_trans_res + i] = sc.cp[lo.cam_trans + i];
    }
    rt.sync();
  }
  GLAMR_MARK_END(rt);
}

}  // namespace GLAMR_GRECON_NS
}  // namespace glamr
