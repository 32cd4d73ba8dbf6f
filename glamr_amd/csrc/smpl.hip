// SMPL forward kinematics + linear blend skinning for gfx950 (MI355X).
//
// Replaces lib/models/smpl.py:274-343 and the smplx.lbs it calls.  Pipeline per call:
//   smpl_prep_kernel    per frame: Rodrigues (smplx variant), pose/shape feature row, kinematic chain (level-parallel),
//                       skinning transforms A_j, posed chain joints
//   smpl_lbs_kernel     f32-MFMA tiles: v_posed = [betas | R-I | 1] . [shapedirs | posedirs | v_template] (K=218, three
//                       coordinate planes sharing the frame operand), T = W . A (K=24), verts = T . [v_posed; 1];
//                       in-register regression of the extra joints, picked-vertex capture, optional vertex write-out
//                       through an LDS transpose (coalesced 384-B rows)
//   smpl_finish_kernel  reduces the per-tile regression partials, assembles the mapped joints, re-anchors them
// Calls that want VERTICES run the LBS kernel twice: first over the joints tileset (3 tiles: the picked vertices and the virtual vertices of
// the extra-regressed joints) -> joints and the re-anchoring pivot (output joint 0, smpl.py:312: for body26fk a REGRESSED joint, i.e. a
// function of every skinned vertex), then over the full mesh with the regression switched off and verts = (verts - pivot) * scale + trans
// applied where a vertex tile leaves the accumulators (round 3 re-read and re-wrote all vertices in a separate pass for that, and wrote /
// re-read 200 MB of per-tile regression partials at B = 19 200).
// Operands of the fp16-plane instances that come from global memory (feature rows, joint transforms) are stored FRAGMENT-MAJOR by
// smpl_prep_kernel: [frame tile][plane][k step][lane][8 halves], so a wave's B-operand fetch is one contiguous 1 KB read (row-major rows had
// every lane of a load in its own cache line: 64 lines per instruction, and the loop waited on each of them) and is issued PF k steps ahead.
//
// MFMA tile orientation: rows (A operand) = vertices of the tile, cols (B operand) = frames.  With
// v_mfma_f32_32x32x2_f32, lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the reduction index k is
// consumed in two interleaved halves (lanes 0-31 walk k = 0..KH-1, lanes 32-63 walk k = KH..2KH-1) so every lane
// reads CONTIGUOUS k and operands can be fetched 16 bytes at a time.  Accumulator register r of lane l is
// element (row = (r & 3) + 8 (r >> 2) + 4 (l >> 5), col = l & 31).
#include "common.hpp"
#include "rotmath.hpp"
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <set>

namespace glamr {

constexpr int NJ = 24;            // chain joints
constexpr int KH = 112;           // reduction entries per lane-half
constexpr int KTOT = 2 * KH;      // 224 >= 10 + 207 + 1
constexpr int KSTRIDE = 228;      // LDS row stride in floats: 228 mod 64 = 36 -> conflict-free ds_read_b128 over 16 rows
constexpr int TILE_V = 32;
constexpr int TILE_F = 32;
constexpr int MAX_EXTRA = 9;
constexpr int MAX_PICKED = 32;
constexpr int OUT_STRIDE = 100;   // LDS transpose row stride (floats) for the vertex write-out: 400-B rows keep the 48-B runs of a lane 16-B aligned,
                                  // and 100 mod 64 = 36 spreads the 16 frames of a pass over all banks (conflict-free ds_write_b128)
constexpr int K_ONE = 217;        // feature index that carries the constant 1 (multiplies v_template)
constexpr int KSH = 232;          // halves per LDS row of one fp16 plane of the direction matrix (224 + 8: 16-byte rows, conflict-free b128 reads)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// fragment-major addresses (in halves) of the fp16 planes the F16 LBS instances fetch from global memory.  Lane l = 32 half + col of a wave
// working on frame tile ft supplies, at k step m, the 8 consecutive k values  112 half + 8 m ..  (feature rows; 14 steps)  resp.
// 16 m + 8 half ..  (joint transforms, K = 24 padded to 32; 2 steps) of frame 32 ft + col.
constexpr int FEAT_STEPS = 14;                        // KH / 8
constexpr int FEAT_TILE_H = 2 * FEAT_STEPS * 64 * 8;  // halves per frame tile (= 32 frames x 2 planes x KTOT)
constexpr int ASK_TILE_H = 12 * 2 * 2 * 64 * 8;       // halves per frame tile (= 32 frames x 12 entries x 2 planes x 32)
__host__ __device__ __forceinline__ size_t feat_h_off(int b, int plane, int k) {
  const int ft = b >> 5, col = b & 31, half = k / 112, kk = k - half * 112;
  return (size_t)ft * FEAT_TILE_H + (size_t)(((plane * FEAT_STEPS + (kk >> 3)) * 2 + half) * 32 + col) * 8 + (kk & 7);
}
__host__ __device__ __forceinline__ size_t askin_h_off(int b, int e, int plane, int k) {
  const int ft = b >> 5, col = b & 31;
  return (size_t)ft * ASK_TILE_H + (size_t)(((((e * 2 + plane) * 2 + (k >> 4)) * 2 + ((k >> 3) & 1)) * 32) + col) * 8 + (k & 7);
}

}  // namespace glamr

// One tiling of a vertex set for the LBS kernel: 32-vertex tiles with their direction matrices, skinning weights, extra-joint
// regressor columns and the picked vertices that fall into each tile.
struct glamr_tileset {
  int n_verts, n_tiles, Vpad;
  float* dirs_tiled;      // [n_tiles][3][32][KSTRIDE]
  unsigned short* dirs_h; // [n_tiles][3][2 planes][32][KSH] the same matrix as two fp16 planes (hi, lo), or null
  float* w_tiled;         // [Vpad][24]
  float* jx_used;         // [n_extra_used][Vpad]
  int32_t* pick_row;      // [n_picked] row inside the tile
  int32_t* tile_pick_start;  // [n_tiles+1] CSR over picks sorted by tile
  int32_t* tile_pick_ids;    // [n_picked] pick slot ids sorted by tile
};

struct glamr_smpl {
  int V, num_betas, n_extra, n_picked, n_out, n_levels;
  int n_extra_used;
  glamr_tileset full;      // every vertex (vertices requested)
  glamr_tileset joints;    // joints only: the picked vertices + 24 VIRTUAL vertices per used extra-regressed joint (see glamr_smpl_create)
  // device constants
  float* j_template;      // [24][3]
  float* j_shapedirs;     // [24][3][num_betas]
  int32_t* parents;       // [24]
  int32_t* level;         // [24]
  int32_t* joint_map;     // [n_out]
  int32_t* extra_slot;    // [MAX_EXTRA] -> slot in jx_used or -1
};

namespace glamr {

// ---------------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------------

// smplx batch_rodrigues: angle = || r + 1e-8 ||, R = I + sin K + (1 - cos) K^2 with K from r / angle.
__device__ __forceinline__ void rodrigues_smplx(const float r[3], float R[9]) {
  const float ax = r[0] + 1e-8f, ay = r[1] + 1e-8f, az = r[2] + 1e-8f;
  const float angle = sqrtf(ax * ax + ay * ay + az * az);
  const float inv = 1.0f / angle;
  const float x = r[0] * inv, y = r[1] * inv, z = r[2] * inv;
  // (rm::sincos_: Cody-Waite reduction + minimax polynomials, 1.5 ulp, ~30 instructions for both; the device library's sinf + cosf carry a
  // Payne-Hanek large-argument path each -- the prep kernel evaluates 24 of these per frame)
  float s, cs;
  rm::sincos_(angle, s, cs);
  const float c1 = 1.0f - cs;
  // K = [[0,-z,y],[z,0,-x],[-y,x,0]],  K^2 = [[-(y^2+z^2), xy, xz],[xy, -(x^2+z^2), yz],[xz, yz, -(x^2+y^2)]]
  R[0] = 1.0f + c1 * (-(y * y + z * z));
  R[1] = s * (-z) + c1 * (x * y);
  R[2] = s * (y) + c1 * (x * z);
  R[3] = s * (z) + c1 * (x * y);
  R[4] = 1.0f + c1 * (-(x * x + z * z));
  R[5] = s * (-x) + c1 * (y * z);
  R[6] = s * (-y) + c1 * (x * z);
  R[7] = s * (x) + c1 * (y * z);
  R[8] = 1.0f + c1 * (-(x * x + y * y));
}


// backward of rodrigues_smplx: g_r = (dR/dr)^T gR
__device__ __forceinline__ void rodrigues_smplx_bwd(const float r[3], const float gR[9], float* g_r) {
  const float ax = r[0] + 1e-8f, ay = r[1] + 1e-8f, az = r[2] + 1e-8f;
  const float angle = sqrtf(ax * ax + ay * ay + az * az), inv = 1.0f / angle;
  const float x = r[0] * inv, y = r[1] * inv, z = r[2] * inv;
  float s, c;
  rm::sincos_(angle, s, c);
  const float c1 = 1.0f - c;
  // R = I + s K + c1 K2
  const float K[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  const float K2[9] = {-(y * y + z * z), x * y, x * z, x * y, -(x * x + z * z), y * z, x * z, y * z, -(x * x + y * y)};
  float g_s = 0.f, g_c1 = 0.f;
  for (int e = 0; e < 9; ++e) { g_s += gR[e] * K[e]; g_c1 += gR[e] * K2[e]; }
  // d/dx, d/dy, d/dz of (s K + c1 K2)
  const float gx = s * (gR[7] - gR[5]) + c1 * (y * (gR[1] + gR[3]) + z * (gR[2] + gR[6]) - 2.f * x * (gR[4] + gR[8]));
  const float gy = s * (gR[2] - gR[6]) + c1 * (x * (gR[1] + gR[3]) + z * (gR[5] + gR[7]) - 2.f * y * (gR[0] + gR[8]));
  const float gz = s * (gR[3] - gR[1]) + c1 * (x * (gR[2] + gR[6]) + y * (gR[5] + gR[7]) - 2.f * z * (gR[0] + gR[4]));
  const float g_angle = g_s * c + g_c1 * s;          // d s/d angle = cos, d c1/d angle = sin
  // x = r0 / angle etc.; angle = || r + eps ||
  const float g_angle_tot = g_angle - (gx * r[0] + gy * r[1] + gz * r[2]) * inv * inv;
  g_r[0] = gx * inv + g_angle_tot * ax * inv;
  g_r[1] = gy * inv + g_angle_tot * ay * inv;
  g_r[2] = gz * inv + g_angle_tot * az * inv;
}

// ---------------------------------------------------------------------------------------------------------------------
// prep: one thread per (frame, joint); FRAMES_PER_BLOCK frames per 256-thread block
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PREP_FRAMES = 8;   // 8 * 24 = 192 active threads of 256

struct PrepArgs {
  int B, Bpad, num_betas, n_levels, use_shape;      // frames [B, Bpad) are the padding of the last 32-frame tile: their operand rows are written as zeros
  const float* pose;        // (B,72), or (B,69) = the body pose alone when body_only (the global orientation is then zero)
  int body_only;
  const float* betas;       // (B,num_betas) or null
  const float* j_template;  // (24,3)
  const float* j_shapedirs; // (24,3,num_betas)
  const int32_t* parents;
  const int32_t* level;
  float* feat;              // (Bpad, KTOT) or null
  unsigned short* feat_h;   // the same rows as two fp16 planes, fragment-major (feat_h_off), or null
  float* askin;             // (Bpad, 12, 24) or null
  unsigned short* askin_h;  // the same transforms as two fp16 planes, K padded to 32 with zeros, fragment-major (askin_h_off), or null
  float* chain_joints;      // (B, 24, 3)
};

// The fp16 planes leave through LDS: a block's 8 frames are 8 consecutive columns of one 32-frame tile, so for every (plane, k step, lane
// half) -- and (entry, plane, k step, lane half) of the joint transforms -- they form ONE 128-byte run of the fragment-major arrays; written
// straight from the (frame, joint) threads these were 2-byte stores 16 bytes apart.
constexpr int PREP_FEAT_SEGS = 2 * FEAT_STEPS * 2;          // (plane, step, half)
constexpr int PREP_ASK_SEGS = 12 * 2 * 2 * 2;               // (entry, plane, step, half)

__global__ __launch_bounds__(256) void smpl_prep_kernel(PrepArgs a) {
  GLAMR_CRITICAL_PATH_PRIO();
  __shared__ float sG[PREP_FRAMES][NJ][12];   // global transform of each joint: 3x3 rotation | translation
  __shared__ float sJ[PREP_FRAMES][NJ][3];    // rest joints
  __shared__ __attribute__((aligned(16))) _Float16 sFeatH[PREP_FEAT_SEGS][PREP_FRAMES * 8];
  __shared__ __attribute__((aligned(16))) _Float16 sAskH[PREP_ASK_SEGS][PREP_FRAMES * 8];
  const int tid = threadIdx.x;
  const int fl = tid / NJ, j = tid % NJ;
  const int b = blockIdx.x * PREP_FRAMES + fl;
  const bool active = (fl < PREP_FRAMES) && (b < a.B);
  const bool padding = (fl < PREP_FRAMES) && !active && (b < a.Bpad);      // rows the padded MFMA tiles read: zeros
  const bool want_feat = a.feat != nullptr || a.feat_h != nullptr;
  const bool want_askin = a.askin != nullptr || a.askin_h != nullptr;
  float R[9];
  float Jr[3] = {0.f, 0.f, 0.f};
  if (active || padding) {
    if (active) {
      const float* pr = a.body_only ? a.pose + (size_t)b * 69 + (j - 1) * 3 : a.pose + (size_t)b * 72 + j * 3;
      float r[3] = {0.f, 0.f, 0.f};
      if (!a.body_only || j > 0) { r[0] = pr[0]; r[1] = pr[1]; r[2] = pr[2]; }
      rodrigues_smplx(r, R);
      for (int c = 0; c < 3; ++c) {
        float v = a.j_template[j * 3 + c];
        if (a.use_shape)
          for (int l = 0; l < a.num_betas; ++l) v = fmaf(a.j_shapedirs[(j * 3 + c) * a.num_betas + l], a.betas[(size_t)b * a.num_betas + l], v);
        Jr[c] = v;
        sJ[fl][j][c] = v;
      }
    }
    if (want_feat) {
      float* f = a.feat ? a.feat + (size_t)b * KTOT : nullptr;
      auto put = [&](int k, float v) {
        if (padding) v = 0.0f;
        if (f) f[k] = v;
        if (a.feat_h) {
          const int half = k / 112, kk = k - half * 112, m = kk >> 3, i = kk & 7;
          const _Float16 hi = (_Float16)v;
          sFeatH[(0 * FEAT_STEPS + m) * 2 + half][fl * 8 + i] = hi;
          sFeatH[(1 * FEAT_STEPS + m) * 2 + half][fl * 8 + i] = (_Float16)(v - (float)hi);
        }
      };
      if (j > 0) {
        for (int e = 0; e < 9; ++e) put(10 + (j - 1) * 9 + e, padding ? 0.0f : R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f));
      } else {
        for (int l = 0; l < 10; ++l) put(l, (!padding && l < a.num_betas) ? a.betas[(size_t)b * a.num_betas + l] : 0.0f);
        put(K_ONE, 1.0f);
        for (int k = K_ONE + 1; k < KTOT; ++k) put(k, 0.0f);
      }
    }
  }
  __syncthreads();
  // kinematic chain, one tree level at a time (SMPL: 9 levels)
  const int par = active ? a.parents[j] : -1;
  const int lev = active ? a.level[j] : -1;
  if (active && lev == 0) {
    for (int e = 0; e < 9; ++e) sG[fl][j][(e / 3) * 4 + (e % 3)] = R[e];
    for (int c = 0; c < 3; ++c) sG[fl][j][c * 4 + 3] = Jr[c];
  }
  __syncthreads();
  for (int L = 1; L < a.n_levels; ++L) {
    if (active && lev == L) {
      const float* Gp = sG[fl][par];
      float t[3] = {Jr[0] - sJ[fl][par][0], Jr[1] - sJ[fl][par][1], Jr[2] - sJ[fl][par][2]};
      float G[12];
      for (int r0 = 0; r0 < 3; ++r0) {
        for (int c = 0; c < 3; ++c) G[r0 * 4 + c] = Gp[r0 * 4 + 0] * R[0 * 3 + c] + Gp[r0 * 4 + 1] * R[1 * 3 + c] + Gp[r0 * 4 + 2] * R[2 * 3 + c];
        G[r0 * 4 + 3] = Gp[r0 * 4 + 0] * t[0] + Gp[r0 * 4 + 1] * t[1] + Gp[r0 * 4 + 2] * t[2] + Gp[r0 * 4 + 3];
      }
      for (int e = 0; e < 12; ++e) sG[fl][j][e] = G[e];
    }
    __syncthreads();
  }
  if (active || padding) {
    const float* G = sG[fl][j];
    if (active) for (int c = 0; c < 3; ++c) a.chain_joints[((size_t)b * NJ + j) * 3 + c] = G[c * 4 + 3];
    if (want_askin) {
      // relative transform: A = [G_R | G_t - G_R J]
      float* A = a.askin ? a.askin + (size_t)b * 12 * NJ : nullptr;
      auto put = [&](int e, float v) {
        if (padding) v = 0.0f;
        if (A) A[e * NJ + j] = v;
        if (a.askin_h) {
          const _Float16 hi = (_Float16)v;
          const int m = j >> 4, half = (j >> 3) & 1, i = j & 7;
          sAskH[(((e * 2 + 0) * 2 + m) * 2) + half][fl * 8 + i] = hi;
          sAskH[(((e * 2 + 1) * 2 + m) * 2) + half][fl * 8 + i] = (_Float16)(v - (float)hi);
          if (j < 8) { sAskH[(((e * 2 + 0) * 2 + 1) * 2) + 1][fl * 8 + j] = (_Float16)0.f; sAskH[(((e * 2 + 1) * 2 + 1) * 2) + 1][fl * 8 + j] = (_Float16)0.f; }      // K padding: k = 24 + j
        }
      };
      for (int r0 = 0; r0 < 3; ++r0) {
        for (int c = 0; c < 3; ++c) put(r0 * 4 + c, padding ? 0.0f : G[r0 * 4 + c]);
        put(r0 * 4 + 3, padding ? 0.0f : G[r0 * 4 + 3] - (G[r0 * 4 + 0] * Jr[0] + G[r0 * 4 + 1] * Jr[1] + G[r0 * 4 + 2] * Jr[2]));
      }
    }
  }
  if (a.feat_h || a.askin_h) {
    // (frames beyond Bpad of the last block: their columns belong to no tile -- Bpad is a multiple of 32 and of PREP_FRAMES, so a block is
    // either entirely inside the padded range or not launched)
    __syncthreads();
    const int b0 = blockIdx.x * PREP_FRAMES, ft = b0 >> 5, c0 = b0 & 31;
    const int nfeat = a.feat_h ? PREP_FEAT_SEGS * 8 : 0, nask = a.askin_h ? PREP_ASK_SEGS * 8 : 0;
    for (int i = tid; i < nfeat + nask; i += 256) {
      if (i < nfeat) {
        const int seg = i >> 3, piece = i & 7;
        uint4* dst = reinterpret_cast<uint4*>(a.feat_h + (size_t)ft * FEAT_TILE_H + (size_t)seg * 256 + c0 * 8) + piece;
        *dst = reinterpret_cast<const uint4*>(&sFeatH[seg][0])[piece];
      } else {
        const int q = i - nfeat, seg = q >> 3, piece = q & 7;
        uint4* dst = reinterpret_cast<uint4*>(a.askin_h + (size_t)ft * ASK_TILE_H + (size_t)seg * 256 + c0 * 8) + piece;
        *dst = reinterpret_cast<const uint4*>(&sAskH[seg][0])[piece];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// main LBS kernel
// ---------------------------------------------------------------------------------------------------------------------
struct LbsArgs {
  int B, V, n_tiles, n_ftiles, n_extra_used, n_picked;
  const float* dirs_tiled;
  const float* w_tiled;
  const float* jx_used;
  const int32_t* tile_pick_start;
  const int32_t* tile_pick_ids;
  const int32_t* pick_row;
  const float* feat;        // (Bpad, KTOT)
  const unsigned short* dirs_h;   // fp16 planes of dirs_tiled (F16 instances)
  const unsigned short* feat_h;   // fp16 planes of feat (F16 instances)
  const unsigned short* askin_h;  // fp16 planes of askin (F16 instances)
  const float* askin;       // (Bpad, 12, 24)
  float* verts;             // (B, V, 3) or null
  float* picked;            // (B, n_picked, 3)
  float* partial;           // (n_tiles, Bpad, n_extra_used, 3)
  int Bpad;
  // re-anchoring of the vertices where they leave the accumulators (smpl.py:309-315): v = (v - pivot) * scale + trans; pivot null = none
  const float* pivot;       // (B,3) un-anchored output joint 0, from the joints pass
  const float* root_trans;  // (B,3)
  const float* root_scale;  // (B) or null
  int skip_picks;           // the joints pass already captured the picked vertices
#ifdef GLAMR_SMPL_EXPERIMENT
  int exp;                  // development builds: ablation switches (tools/smpl_ablate.py)
#endif
};
#ifdef GLAMR_SMPL_EXPERIMENT
#define GLAMR_SMPL_EXP(a, bit) (((a).exp & (bit)) != 0)
#else
#define GLAMR_SMPL_EXP(a, bit) false
#endif

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// F16: the blend-shape product (K = 218, 70 % of the matrix work) on the fp16 matrix cores with both operands as two fp16 planes (hi + lo,
// three v_mfma_f32_32x32x16_f16 per 16-deep k step: fp32-grade, 2^-22 of the operands) instead of 112 v_mfma_f32_32x32x2_f32 per
// coordinate; planes are built once (directions at model creation, feature rows in smpl_prep_kernel).  Skinning stays fp32.
// The F16 instances run EIGHT waves per workgroup (two per SIMD: the direction tile fills the CU's LDS, so a second wave on the SIMD is
// the only thing that hides a wave's operand fetches); their vertex transpose buffers hold half a frame tile and are used twice.
template <int NE, bool F16 = false>
__global__ __launch_bounds__(F16 ? 512 : 256) void smpl_lbs_kernel(LbsArgs a) {
  GLAMR_CRITICAL_PATH_PRIO();
  constexpr int NW = F16 ? 8 : 4, NT = NW * 64;               // waves, threads
  constexpr int OF = F16 ? TILE_F / 2 : TILE_F;                // frames per transpose pass
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sDirs = smem;                                   // [3][32][KSTRIDE]
  float* sOut = smem + (F16 ? 3 * 2 * TILE_V * KSH / 2 : 3 * TILE_V * KSTRIDE);      // [4 waves][32 frames][OUT_STRIDE]  (only when verts != null)
  // workgroup -> vertex tile: workgroups go to the 8 XCDs round robin by their linear id (x fastest; 216 tiles: the phase is the same in every y
  // row), so tiles x and x + 1 wrote the two ends of a shared 128-byte line of every frame row from two different L2s.  XCD i takes a CONTIGUOUS
  // range of tiles instead: neighbouring 384-byte runs of a frame row meet in one L2 (vertex pass at B = 19 200: 1.13 -> 1.06 ms, same bits).
  const int xq = a.n_tiles >> 3, xr = a.n_tiles & 7, xi = blockIdx.x & 7;
  const int tile = xi * xq + min(xi, xr) + (blockIdx.x >> 3);
  const int v0 = tile * TILE_V;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;

  // stage this tile's direction matrix (contiguous 3*32*KSTRIDE floats, or 3*2*32*KSH halves) into LDS
  if (F16) {
    const uint4* src = reinterpret_cast<const uint4*>(a.dirs_h + (size_t)tile * 3 * 2 * TILE_V * KSH);
    uint4* dst = reinterpret_cast<uint4*>(sDirs);
    // 5568 sixteen-byte pieces over 512 threads: all eleven loads of a thread in flight before its first LDS write (one memory round trip
    // per workgroup instead of eleven -- nothing else runs on the CU while its only workgroup stages)
    constexpr int NPIECE = 3 * 2 * TILE_V * KSH / 8, NFULL = NPIECE / NT;
    uint4 st[NFULL], tail = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NFULL; ++k) st[k] = src[tid + k * NT];
    const bool has_tail = tid + NFULL * NT < NPIECE;
    if (has_tail) tail = src[tid + NFULL * NT];
#pragma unroll
    for (int k = 0; k < NFULL; ++k) dst[tid + k * NT] = st[k];
    if (has_tail) dst[tid + NFULL * NT] = tail;
  } else {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.dirs_tiled + (size_t)tile * 3 * TILE_V * KSTRIDE);
    f32x4* dst = reinterpret_cast<f32x4*>(sDirs);
    for (int i = tid; i < 3 * TILE_V * KSTRIDE / 4; i += NT) dst[i] = src[i];
  }
  // skinning weights of my A-operand row (vertex v0 + col), my k-half: 12 values
  float wreg[12];
  {
    const float* w = a.w_tiled + (size_t)(v0 + col) * NJ + half * 12;
#pragma unroll
    for (int m = 0; m < 12; ++m) wreg[m] = w[m];
  }
  // F16: the same weights as fp16 planes in the k order of the 16-deep steps (step m, lane half h: joints 16 m + 8 h .. + 7, zero from 24 on)
  f16x8 wh_[2], wl_[2];
  if (F16) {
    const float* w = a.w_tiled + (size_t)(v0 + col) * NJ;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 16 * m + 8 * half + i;
        const float x = k < NJ ? w[k] : 0.0f;
        const _Float16 hi = (_Float16)x;
        wh_[m][i] = hi;
        wl_[m][i] = (_Float16)(x - (float)hi);
      }
  }
  // extra-joint regressor weights for the 16 accumulator rows this lane owns
  // (registers in the 4-wave instances; the 8-wave F16 instances live on 256 registers and read them from LDS at the point of use)
  __shared__ float sJx[(NE > 0 ? NE : 1) * TILE_V];
  float jx[NE > 0 ? NE : 1][16];
  if (F16) {
    for (int i = tid; i < NE * TILE_V; i += NT) sJx[i] = a.jx_used[(size_t)(i / TILE_V) * (a.n_tiles * TILE_V) + v0 + (i % TILE_V)];
  } else {
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) jx[e][r] = a.jx_used[(size_t)e * (a.n_tiles * TILE_V) + v0 + acc_row(r, half)];
  }
  __syncthreads();

  const float* myDirs = sDirs + (size_t)col * KSTRIDE + half * KH;

  for (int ft = blockIdx.y * NW + wave; ft < a.n_ftiles; ft += NW * gridDim.y) {
    const int b = ft * TILE_F + col;            // Bpad is a multiple of 32: always a readable row
    const float* frow = a.feat + (size_t)b * KTOT + half * KH;
    f32x16 px = {0}, py = {0}, pz = {0};
    if (F16) {
      // lane half h supplies k = 112 h + 8 m .. + 7 to k step m, on both operands alike.  The feature planes are fragment-major: the
      // fetch of (plane, step) is ONE contiguous 1 KB read per wave, requested PF steps before its MFMAs
      const unsigned short* myH = reinterpret_cast<const unsigned short*>(sDirs) + (size_t)col * KSH + half * KH;
      const uint4* fq = reinterpret_cast<const uint4*>(a.feat_h) + (GLAMR_SMPL_EXP(a, 8) ? (size_t)wave : (size_t)ft) * (FEAT_TILE_H / 8) + lane;
      constexpr int PF = 3, RING = PF + 1;
      uint4 qh[RING], ql[RING];
      uint4 ah[2][3], al[2][3];          // direction fragments (LDS), one k step ahead as well
      auto lds_frag = [&](int m, int buf) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          ah[buf][c] = *reinterpret_cast<const uint4*>(myH + (size_t)(c * 2 + 0) * TILE_V * KSH + 8 * m);
          al[buf][c] = *reinterpret_cast<const uint4*>(myH + (size_t)(c * 2 + 1) * TILE_V * KSH + 8 * m);
        }
      };
#pragma unroll
      for (int m = 0; m < PF; ++m) { qh[m] = fq[(0 * FEAT_STEPS + m) * 64]; ql[m] = fq[(1 * FEAT_STEPS + m) * 64]; }
      lds_frag(0, 0);
#pragma unroll
      for (int m = 0; m < FEAT_STEPS; ++m) {
        if (m + PF < FEAT_STEPS) {
          qh[(m + PF) % RING] = fq[(0 * FEAT_STEPS + m + PF) * 64];
          ql[(m + PF) % RING] = fq[(1 * FEAT_STEPS + m + PF) * 64];
        }
        if (m + 1 < FEAT_STEPS) lds_frag(m + 1, (m + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);            // (left alone the scheduler sinks the loads to just before their use)
        const f16x8 bh = __builtin_bit_cast(f16x8, qh[m % RING]);
        const f16x8 bl = __builtin_bit_cast(f16x8, ql[m % RING]);
        const int cur = m & 1;
        // small products first; consecutive MFMAs go to different accumulators
        px = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[cur][0]), bh, px, 0, 0, 0);
        py = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[cur][1]), bh, py, 0, 0, 0);
        pz = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[cur][2]), bh, pz, 0, 0, 0);
        px = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[cur][0]), bl, px, 0, 0, 0);
        py = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[cur][1]), bl, py, 0, 0, 0);
        pz = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[cur][2]), bl, pz, 0, 0, 0);
        px = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[cur][0]), bh, px, 0, 0, 0);
        py = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[cur][1]), bh, py, 0, 0, 0);
        pz = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[cur][2]), bh, pz, 0, 0, 0);
      }
    } else
#pragma unroll 2
    for (int m = 0; m < KH / 4; ++m) {
      const f32x4 fb = *reinterpret_cast<const f32x4*>(frow + 4 * m);
      const f32x4 ax = *reinterpret_cast<const f32x4*>(myDirs + 0 * TILE_V * KSTRIDE + 4 * m);
      const f32x4 ay = *reinterpret_cast<const f32x4*>(myDirs + 1 * TILE_V * KSTRIDE + 4 * m);
      const f32x4 az = *reinterpret_cast<const f32x4*>(myDirs + 2 * TILE_V * KSTRIDE + 4 * m);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        px = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[s], fb[s], px, 0, 0, 0);
        py = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[s], fb[s], py, 0, 0, 0);
        pz = __builtin_amdgcn_mfma_f32_32x32x2f32(az[s], fb[s], pz, 0, 0, 0);
      }
    }
    // skinning: out_r = T_r0 px + T_r1 py + T_r2 pz + T_r3,  T_e[v, b] = sum_j W[v, j] A_e[b, j]
    const float* arow = a.askin + (size_t)b * 12 * NJ + half * 12;
    f32x16 out[3];
    const bool frame_ok = b < a.B;
    // the frame's re-anchoring constants (lane = frame) are requested HERE, a skinning phase ahead of their use: asked for where the vertices leave
    // the accumulators, every frame tile waited a memory round trip for them (s_waitcnt vmcnt(0) between the last MFMA and the first subtraction).
    // (a fused v * scale + (trans - pivot * scale) and a nested-fma form of the skinning sum were measured as well: 100 fewer VALU operations per frame
    // tile, no faster -- the epilogue's arithmetic is not what the tile waits for -- and not kept: the results keep their bits)
    float anc_pv[3] = {0.f, 0.f, 0.f}, anc_tr[3] = {0.f, 0.f, 0.f}, anc_sc = 1.0f;
    if (a.pivot && frame_ok) {
      anc_sc = a.root_scale ? a.root_scale[b] : 1.0f;
#pragma unroll
      for (int r = 0; r < 3; ++r) { anc_pv[r] = a.pivot[(size_t)b * 3 + r]; anc_tr[r] = a.root_trans[(size_t)b * 3 + r]; }
    }
    if (F16) {
      // joint transforms, fragment-major: entry e = 4 r + c, (plane, k step m) -> one contiguous 1 KB read per wave.  Twelve phases
      // (r, m, pair of c) of four fetches and six MFMAs each; the fetches of a phase are requested before the MFMAs of the one before it
      // (a ring of three sets, two phases ahead, costs 16 more registers: the epilogue then spills addresses, and a scratch reload waits on
      // vmcnt(0) -- on every vertex store still in flight; the ablations say the operand fetches are 0.2 of 1.5 ms, the epilogue 0.75)
      const uint4* aq = reinterpret_cast<const uint4*>(a.askin_h) + (GLAMR_SMPL_EXP(a, 4) ? (size_t)wave : (size_t)ft) * (ASK_TILE_H / 8) + lane;
      uint4 qa[2][2][2];      // [ring][c of the pair][plane]
      auto fetch = [&](int ph, int buf) {
        const int r = ph >> 2, m = (ph >> 1) & 1, cp = ph & 1;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) qa[buf][ci][pl] = aq[((((r * 4 + 2 * cp + ci) * 2 + pl) * 2 + m)) * 64];
      };
      fetch(0, 0);
      f32x16 T[4];
#pragma unroll
      for (int ph = 0; ph < 12; ++ph) {
        const int r = ph >> 2, m = (ph >> 1) & 1, cp = ph & 1, buf = ph & 1;
        if (ph + 1 < 12) fetch(ph + 1, (ph + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int c0 = 2 * cp, c1 = 2 * cp + 1;
        if (m == 0) { T[c0] = (f32x16){0}; T[c1] = (f32x16){0}; }
        T[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl_[m], __builtin_bit_cast(f16x8, qa[buf][0][0]), T[c0], 0, 0, 0);
        T[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl_[m], __builtin_bit_cast(f16x8, qa[buf][1][0]), T[c1], 0, 0, 0);
        T[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[m], __builtin_bit_cast(f16x8, qa[buf][0][1]), T[c0], 0, 0, 0);
        T[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[m], __builtin_bit_cast(f16x8, qa[buf][1][1]), T[c1], 0, 0, 0);
        T[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[m], __builtin_bit_cast(f16x8, qa[buf][0][0]), T[c0], 0, 0, 0);
        T[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[m], __builtin_bit_cast(f16x8, qa[buf][1][0]), T[c1], 0, 0, 0);
        if (m == 1 && cp == 1) out[r] = T[0] * px + T[1] * py + T[2] * pz + T[3];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        f32x16 T[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          T[c] = (f32x16){0};
          const float* ae = arow + (r * 4 + c) * NJ;
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(ae), a1 = *reinterpret_cast<const f32x4*>(ae + 4),
                      a2 = *reinterpret_cast<const f32x4*>(ae + 8);
#pragma unroll
          for (int m = 0; m < 4; ++m) T[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[m], a0[m], T[c], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 4; ++m) T[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 + m], a1[m], T[c], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 4; ++m) T[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[8 + m], a2[m], T[c], 0, 0, 0);
        }
        out[r] = T[0] * px + T[1] * py + T[2] * pz + T[3];
      }
    }
    if (a.pivot && frame_ok) {      // re-anchor on output joint 0 (lane = frame: three pivots, three translations, one scale per lane)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) out[r][q] = (out[r][q] - anc_pv[r]) * anc_sc + anc_tr[r];
    }
    // extra-joint regression partial sums: sum over this tile's 32 vertices (16 rows here + 16 in the other half)
    if (NE > 0) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float s = 0.f;
#pragma unroll
          for (int q = 0; q < 16; ++q) s = fmaf(F16 ? sJx[e * TILE_V + acc_row(q, half)] : jx[e][q], out[r][q], s);
          s += __shfl_xor(s, 32);
          if (half == 0) a.partial[(((size_t)tile * a.Bpad + b) * NE + e) * 3 + r] = s;
        }
      }
    }
    // picked vertices of this tile (rare: 21 picks over 216 tiles)
    const int pick0 = a.skip_picks ? 0 : a.tile_pick_start[tile], pick1 = a.skip_picks ? 0 : a.tile_pick_start[tile + 1];
    for (int p = pick0; p < pick1; ++p) {
      const int slot = a.tile_pick_ids[p];
      const int row = a.pick_row[slot];
      if (frame_ok) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (acc_row(q, half) == row) {
            float* dst = a.picked + ((size_t)b * a.n_picked + slot) * 3;
            dst[0] = out[0][q];
            dst[1] = out[1][q];
            dst[2] = out[2][q];
          }
      }
    }
    // vertex write-out through an LDS transpose: [frame][vertex row][xyz] so each frame's 96 floats are contiguous.  A lane's 16 accumulator
    // rows are four runs of four consecutive vertices = 12 consecutive floats each: three ds_write_b128 per run; the rows leave as 16-byte
    // pieces, two frame rows per store instruction
    if (a.verts && !GLAMR_SMPL_EXP(a, 2)) {
      float* so = sOut + (size_t)wave * OF * OUT_STRIDE;
      const int nvalid = GLAMR_SMPL_EXP(a, 1) ? 0 : min(TILE_V, a.V - v0) * 3;
      const bool wide = (a.V & 1) == 0;
#pragma unroll
      for (int pass = 0; pass < TILE_F / OF; ++pass) {
        if (col / OF == pass) {
          const int fc = col - pass * OF;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4* dst = reinterpret_cast<f32x4*>(so + fc * OUT_STRIDE + (8 * g + 4 * half) * 3);
            dst[0] = (f32x4){out[0][4 * g + 0], out[1][4 * g + 0], out[2][4 * g + 0], out[0][4 * g + 1]};
            dst[1] = (f32x4){out[1][4 * g + 1], out[2][4 * g + 1], out[0][4 * g + 2], out[1][4 * g + 2]};
            dst[2] = (f32x4){out[2][4 * g + 2], out[0][4 * g + 3], out[1][4 * g + 3], out[2][4 * g + 3]};
          }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (wide) {
          // TWO frame rows (24 sixteen-byte pieces each) per step: lanes 0 .. 23 the even row, lanes 32 .. 55 the odd one.  The LDS address is
          // one register + an immediate, the global address advances by two frame strides -- no per-piece index arithmetic for the compiler to
          // hoist out of the frame loop and spill (a scratch reload waits on vmcnt(0), i.e. on every vertex store still in flight).  A row starts
          // on an 8-byte boundary (V even), so every other row's pieces are 8-byte aligned only: global memory takes unaligned dwordx4.
          const int sub = lane >> 5, pc = lane & 31;
          const bool mine = pc < 24 && 4 * pc < nvalid;
          const int nhere = nvalid - 4 * pc;                       // floats of this piece inside the tile's valid range (>= 4: whole piece)
          const float* src = so + sub * OUT_STRIDE + 4 * (pc < 24 ? pc : 0);
          const int bb0 = ft * TILE_F + pass * OF + sub;
          float* dst = a.verts + ((size_t)bb0 * a.V + v0) * 3 + 4 * (pc < 24 ? pc : 0);
          const size_t fstride2 = (size_t)a.V * 6;
#pragma unroll
          for (int h = 0; h < OF / 2; h += 4) {
            f32x4 piece[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) piece[k] = *reinterpret_cast<const f32x4*>(src + 2 * (h + k) * OUT_STRIDE);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float* d = dst + (size_t)(h + k) * fstride2;
              if (mine && bb0 + 2 * (h + k) < a.B) {
                // (non-temporal: 1.6 GB of vertices per 19 200 frames stream through the L2s, nobody reads them back here -- 1.06 -> 0.97 ms)
                if (nhere >= 4) __builtin_nontemporal_store(piece[k], reinterpret_cast<f32x4*>(d));
                else { d[0] = piece[k][0]; if (nhere > 1) d[1] = piece[k][1]; if (nhere > 2) d[2] = piece[k][2]; }
              }
            }
          }
        } else {
          for (int idx = lane; idx < OF * 96; idx += 64) {
            const int f = idx / 96, c = idx - f * 96;
            const int bb = ft * TILE_F + pass * OF + f;
            if (bb < a.B && c < nvalid) a.verts[((size_t)bb * a.V + v0) * 3 + c] = so[f * OUT_STRIDE + c];
          }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// finish: assemble mapped joints, re-anchor
// ---------------------------------------------------------------------------------------------------------------------
struct FinishArgs {
  int B, Bpad, n_tiles, n_extra_used, n_picked, n_out, orig_joints;
  const float* chain_joints;   // (B,24,3)
  const float* picked;         // (B,n_picked,3)
  const float* partial;        // (n_tiles,Bpad,n_extra_used,3)
  const int32_t* joint_map;
  const int32_t* extra_slot;
  const float* root_trans;     // (B,3) or null
  const float* root_scale;     // (B) or null
  float* joints;               // (B,n_out,3)
  float* pivot;                // (B,3) un-anchored position of joint 0 (for the vertex pass)
};

// one WAVE per frame, four frames per workgroup (a workgroup per frame was 307 200 launches of 64 threads for a 1024 x 300 batch: 0.27 ms)
__global__ __launch_bounds__(256) void smpl_finish_kernel(FinishArgs a) {
  GLAMR_CRITICAL_PATH_PRIO();
  __shared__ float sExtra[4][MAX_EXTRA * 3];
  __shared__ float sPivot[4][3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  const bool ok = b < a.B;
  const int nred = a.n_extra_used * 3;
  // deterministic reduction over tiles: lane l sums tiles l, l+64, ... then a shuffle tree -- the sums of a frame side by side: one
  // sum after the other was nred dependent chains of a load and six shuffles each (0.44 ms per 307 200 frames; the additions and their order
  // are the same)
  constexpr int NRED_MAX = MAX_EXTRA * 3;
  float s[NRED_MAX];
#pragma unroll
  for (int q = 0; q < NRED_MAX; ++q) s[q] = 0.f;
  if (ok)
    for (int t = lane; t < a.n_tiles; t += 64) {
      const float* p = a.partial + ((size_t)t * a.Bpad + b) * nred;
#pragma unroll
      for (int q = 0; q < NRED_MAX; ++q) if (q < nred) s[q] += p[q];
    }
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int q = 0; q < NRED_MAX; ++q) if (q < nred) s[q] += __shfl_xor(s[q], off);
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < NRED_MAX; ++q) if (q < nred) sExtra[wave][q] = s[q];
  }
  __syncthreads();
  const int n_out = a.orig_joints ? NJ : a.n_out;
  float val[3] = {0.f, 0.f, 0.f};
  if (ok && lane < n_out) {
    const int src = a.orig_joints ? lane : a.joint_map[lane];
    for (int c = 0; c < 3; ++c) {
      if (src < NJ) val[c] = a.chain_joints[((size_t)b * NJ + src) * 3 + c];
      else if (src < NJ + a.n_picked) val[c] = a.picked[((size_t)b * a.n_picked + (src - NJ)) * 3 + c];
      else val[c] = sExtra[wave][a.extra_slot[src - NJ - a.n_picked] * 3 + c];
    }
    if (lane == 0) for (int c = 0; c < 3; ++c) sPivot[wave][c] = val[c];
  }
  __syncthreads();
  if (!ok) return;
  if (lane < n_out) {
    if (a.root_trans) {
      const float sc = a.root_scale ? a.root_scale[b] : 1.0f;
      for (int c = 0; c < 3; ++c) val[c] = (val[c] - sPivot[wave][c]) * sc + a.root_trans[(size_t)b * 3 + c];
    }
    for (int c = 0; c < 3; ++c) a.joints[((size_t)b * n_out + lane) * 3 + c] = val[c];
  }
  if (lane < 3 && a.pivot) a.pivot[(size_t)b * 3 + lane] = sPivot[wave][lane];
}

// The same for calls whose regression partials come from at most four tiles (the joints tileset: 3): one THREAD per (frame, output joint), eight
// frames per workgroup, the few partials of a regressed joint added in the order the wave reduction above adds them -- (t0 + t2) + (t1 + t3).
// A wave per frame was 307 200 waves of three working lanes and six shuffles for a 1024 x 300 batch: 0.38 ms for 0.28 GB.
__global__ __launch_bounds__(256) void smpl_finish_small_kernel(FinishArgs a) {
  GLAMR_CRITICAL_PATH_PRIO();
  __shared__ float sPivot[8][3];
  const int fl = threadIdx.x >> 5, j = threadIdx.x & 31;
  const int b = blockIdx.x * 8 + fl;
  const int n_out = a.orig_joints ? NJ : a.n_out;
  const bool ok = b < a.B && j < n_out;
  float val[3] = {0.f, 0.f, 0.f};
  if (ok) {
    const int src = a.orig_joints ? j : a.joint_map[j];
    if (src < NJ) {
      for (int c = 0; c < 3; ++c) val[c] = a.chain_joints[((size_t)b * NJ + src) * 3 + c];
    } else if (src < NJ + a.n_picked) {
      for (int c = 0; c < 3; ++c) val[c] = a.picked[((size_t)b * a.n_picked + (src - NJ)) * 3 + c];
    } else {
      const int slot = a.extra_slot[src - NJ - a.n_picked], nred = a.n_extra_used * 3;
      for (int c = 0; c < 3; ++c) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 4; ++k) if (k < a.n_tiles) t[k] = 0.f + a.partial[((size_t)k * a.Bpad + b) * nred + slot * 3 + c];
        val[c] = (t[0] + t[2]) + (t[1] + t[3]);
      }
    }
    if (j == 0) for (int c = 0; c < 3; ++c) sPivot[fl][c] = val[c];
  }
  __syncthreads();
  if (!ok) return;
  const float pv[3] = {sPivot[fl][0], sPivot[fl][1], sPivot[fl][2]};
  if (a.root_trans) {
    const float sc = a.root_scale ? a.root_scale[b] : 1.0f;
    for (int c = 0; c < 3; ++c) val[c] = (val[c] - pv[c]) * sc + a.root_trans[(size_t)b * 3 + c];
  }
  for (int c = 0; c < 3; ++c) a.joints[((size_t)b * n_out + j) * 3 + c] = val[c];
  if (j < 3 && a.pivot) a.pivot[(size_t)b * 3 + j] = pv[j];
}

__global__ __launch_bounds__(256) void smpl_anchor_kernel(int B, int V, const float* pivot, const float* root_trans,
                                                          const float* root_scale, float* verts) {
  const int b = blockIdx.y;
  const float sc = root_scale ? root_scale[b] : 1.0f;
  const float p[3] = {pivot[b * 3], pivot[b * 3 + 1], pivot[b * 3 + 2]};
  const float t[3] = {root_trans[b * 3], root_trans[b * 3 + 1], root_trans[b * 3 + 2]};
  float* row = verts + (size_t)b * V * 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < V * 3; i += gridDim.x * 256) {
    const int c = i % 3;
    row[i] = (row[i] - p[c]) * sc + t[c];
  }
}

// FK-only re-anchoring (get_joints): joints = (joints - joints[0]) * scale + root_trans
__global__ void smpl_fk_anchor_kernel(int B, const float* root_trans, const float* root_scale, float* joints) {
  const int b = blockIdx.x, j = threadIdx.x;
  __shared__ float p[3];
  float v[3];
  for (int c = 0; c < 3; ++c) v[c] = joints[((size_t)b * NJ + j) * 3 + c];
  if (j == 0) for (int c = 0; c < 3; ++c) p[c] = v[c];
  __syncthreads();
  const float sc = root_scale ? root_scale[b] : 1.0f;
  for (int c = 0; c < 3; ++c) joints[((size_t)b * NJ + j) * 3 + c] = (v[c] - p[c]) * sc + root_trans[(size_t)b * 3 + c];
}

// ---------------------------------------------------------------------------------------------------------------------
// backward w.r.t. root orientation / translation / scale through the rigid identity
//   y = s R (x_local - pivot_local) + t   =>   dL/dR = [sum g (y - t)^T] R / 1,   dL/dt = sum g,   dL/ds = sum g.(y - t) / s
// followed by the backward of the smplx Rodrigues formula.
// ---------------------------------------------------------------------------------------------------------------------
struct BwdArgs {
  int B, V, n_out;
  const float* pose; const float* root_trans; const float* root_scale;
  const float* verts; const float* joints; const float* g_verts; const float* g_joints;
  float* g_orient; float* g_trans; float* g_scale;
};

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void smpl_backward_root_kernel(BwdArgs a) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool anchored = a.root_trans != nullptr;
  float t[3] = {0.f, 0.f, 0.f};
  if (anchored) for (int c = 0; c < 3; ++c) t[c] = a.root_trans[(size_t)b * 3 + c];
  // when not anchored the rigid motion is about the (shaped) root joint, which the outputs do not carry: unsupported here
  float M[9] = {0}, gs[3] = {0}, gdot = 0.f;
  auto accum = [&](const float* y, const float* g) {
    const float d[3] = {y[0] - t[0], y[1] - t[1], y[2] - t[2]};
    for (int i = 0; i < 3; ++i) {
      gs[i] += g[i];
      gdot += g[i] * d[i];
      for (int k = 0; k < 3; ++k) M[i * 3 + k] += g[i] * d[k];
    }
  };
  if (a.g_verts)
    for (int v = tid; v < a.V; v += 256) accum(a.verts + ((size_t)b * a.V + v) * 3, a.g_verts + ((size_t)b * a.V + v) * 3);
  if (a.g_joints)
    for (int j = tid; j < a.n_out; j += 256) accum(a.joints + ((size_t)b * a.n_out + j) * 3, a.g_joints + ((size_t)b * a.n_out + j) * 3);
  for (int e = 0; e < 9; ++e) M[e] = block_sum_256(M[e], red);
  for (int e = 0; e < 3; ++e) gs[e] = block_sum_256(gs[e], red);
  gdot = block_sum_256(gdot, red);
  if (tid != 0) return;
  const float sc = a.root_scale ? a.root_scale[b] : 1.0f;
  if (a.g_trans) for (int c = 0; c < 3; ++c) a.g_trans[(size_t)b * 3 + c] = gs[c];
  if (a.g_scale) a.g_scale[b] = gdot / sc;
  // dL/dR = M R   (since (x_local - pivot) = R^T (y - t) / s  and  dy/dR = s (x_local - pivot)^T)
  const float r[3] = {a.pose[(size_t)b * 72], a.pose[(size_t)b * 72 + 1], a.pose[(size_t)b * 72 + 2]};
  float R[9];
  rodrigues_smplx(r, R);
  float gR[9];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) gR[i * 3 + k] = M[i * 3 + 0] * R[0 * 3 + k] + M[i * 3 + 1] * R[1 * 3 + k] + M[i * 3 + 2] * R[2 * 3 + k];
  rodrigues_smplx_bwd(r, gR, a.g_orient + (size_t)b * 3);
}


// ---------------------------------------------------------------------------------------------------------------------
// GENERAL backward: gradients w.r.t. the whole pose (global orientation + 23 body joints), the shape coefficients, root translation and
// scale, from gradients of the joints and / or the vertices (torch autograd through lib/models/smpl.py:289-316 + smplx.lbs).  Needed when
// the body pose itself is a function of optimisation variables (the latent-optimisation mode, global_recon_model.py:434-437).
//   anchor   y = (x - x_pivot) s + t                       -> g_x = s g_y,  g_pivot = -s sum g_y,  g_t = sum g_y,  g_s = sum g_y . (y - t) / s
//   vertex   x_v = T_v [v_posed_v; 1],  T_v = sum_k W_vk A_k -> g_vposed = T_R^T g,  g_T = g (x) [v_posed; 1],  g_A_k = sum_v W_vk g_T_v
//   blend    v_posed = dirs . [betas | vec(R_k - I) | 1]    -> g_feat = dirs^T g_vposed
//   chain    G_k = G_parent [R_k | J_k - J_parent],  A_k = [G_R | G_t - G_R J_k],  chain joint k = G_t          (reverse order)
//   rest     J = J_template + J_shapedirs betas             -> g_betas += J_shapedirs^T g_J
// Per (vertex tile, frame) partial sums are written out and reduced in a FIXED order (no atomics: results are reproducible).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BW_PART = 218 + 12 * NJ;        // g_feat[0..217] | g_A[12][24]

struct BwdGenArgs {
  int B, Bpad, V, Vpad, n_tiles, n_extra_used, n_picked, n_out, orig_joints, num_betas;
  const float* pose; const float* betas; const float* root_trans; const float* root_scale;
  const float* verts; const float* joints;         // forward outputs (needed for g_scale only)
  const float* g_verts; const float* g_joints;
  const int32_t* joint_map; const int32_t* extra_slot; const int32_t* parents;
  const float* j_template; const float* j_shapedirs;
  const float* dirs_tiled; const float* w_tiled; const float* jx_used;
  const int32_t* tile_pick_start; const int32_t* tile_pick_ids; const int32_t* pick_row;
  const float* feat; const float* askin;           // recomputed by smpl_prep_kernel
  float* gj54;                                     // (B, 24 + MAX_PICKED + MAX_EXTRA, 3): gradient of [chain | picked | extra-regressed] joints
  float* partial;                                  // (n_tiles, Bpad, BW_PART)
  float* g_pose; float* g_betas; float* g_trans; float* g_scale;
};
constexpr int GJ_STRIDE = NJ + MAX_PICKED + MAX_EXTRA;

// one workgroup per frame: gradient of the mapped joints -> gradient of the 54-joint set, anchor terms
__global__ __launch_bounds__(256) void smpl_bwd_anchor_kernel(BwdGenArgs a) {
  __shared__ float red[4];
  __shared__ float sg[GJ_STRIDE * 3];
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool anchored = a.root_trans != nullptr;
  const float sc = (anchored && a.root_scale) ? a.root_scale[b] : 1.0f;
  const int n_out = a.orig_joints ? NJ : a.n_out;
  for (int i = tid; i < GJ_STRIDE * 3; i += 256) sg[i] = 0.f;
  float gs[3] = {0.f, 0.f, 0.f}, gdot = 0.f;
  float t[3] = {0.f, 0.f, 0.f};
  if (anchored) for (int c = 0; c < 3; ++c) t[c] = a.root_trans[(size_t)b * 3 + c];
  if (a.g_verts)
    for (int v = tid; v < a.V; v += 256)
      for (int c = 0; c < 3; ++c) {
        const float g = a.g_verts[((size_t)b * a.V + v) * 3 + c];
        gs[c] += g;
        if (a.g_scale && a.verts) gdot += g * (a.verts[((size_t)b * a.V + v) * 3 + c] - t[c]);
      }
  if (a.g_joints && tid < n_out)
    for (int c = 0; c < 3; ++c) {
      const float g = a.g_joints[((size_t)b * n_out + tid) * 3 + c];
      gs[c] += g;
      if (a.g_scale && a.joints) gdot += g * (a.joints[((size_t)b * n_out + tid) * 3 + c] - t[c]);
    }
  for (int c = 0; c < 3; ++c) gs[c] = block_sum_256(gs[c], red);
  gdot = block_sum_256(gdot, red);
  __syncthreads();
  // scatter through the joint map (a source joint may be mapped more than once: one thread walks the outputs)
  if (tid == 0) {
    for (int j = 0; j < n_out && a.g_joints; ++j) {
      const int src = a.orig_joints ? j : a.joint_map[j];
      const int slot = src < NJ ? src : (src < NJ + a.n_picked ? NJ + (src - NJ) : NJ + MAX_PICKED + a.extra_slot[src - NJ - a.n_picked]);
      for (int c = 0; c < 3; ++c) sg[slot * 3 + c] += sc * a.g_joints[((size_t)b * n_out + j) * 3 + c];
    }
    if (anchored) {      // the pivot is output joint 0 before re-anchoring
      const int src = a.orig_joints ? 0 : a.joint_map[0];
      const int slot = src < NJ ? src : (src < NJ + a.n_picked ? NJ + (src - NJ) : NJ + MAX_PICKED + a.extra_slot[src - NJ - a.n_picked]);
      for (int c = 0; c < 3; ++c) sg[slot * 3 + c] -= sc * gs[c];
      if (a.g_trans) for (int c = 0; c < 3; ++c) a.g_trans[(size_t)b * 3 + c] = gs[c];
      if (a.g_scale) a.g_scale[b] = gdot / sc;
    }
  }
  __syncthreads();
  for (int i = tid; i < GJ_STRIDE * 3; i += 256) a.gj54[(size_t)b * GJ_STRIDE * 3 + i] = sg[i];
}

// one workgroup per (vertex tile, frame)
__global__ __launch_bounds__(256) void smpl_bwd_tile_kernel(BwdGenArgs a) {
  __shared__ float sfeat[KTOT];
  __shared__ float sA[12 * NJ];
  __shared__ float sgv[TILE_V][3], svp[TILE_V][3], sgvp[TILE_V][3], sgT[TILE_V][12];
  const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  for (int k = tid; k < KTOT; k += 256) sfeat[k] = a.feat[(size_t)b * KTOT + k];
  for (int k = tid; k < 12 * NJ; k += 256) sA[k] = a.askin[(size_t)b * 12 * NJ + k];
  const float sc = (a.root_trans && a.root_scale) ? a.root_scale[b] : 1.0f;
  const float* gj = a.gj54 + (size_t)b * GJ_STRIDE * 3;
  if (tid < TILE_V * 3) {
    const int v = tid / 3, c = tid % 3, vg = tile * TILE_V + v;
    float g = 0.f;
    if (a.g_verts && vg < a.V) g = sc * a.g_verts[((size_t)b * a.V + vg) * 3 + c];
    for (int x = 0; x < a.n_extra_used; ++x) g += a.jx_used[(size_t)x * a.Vpad + vg] * gj[(NJ + MAX_PICKED + x) * 3 + c];
    for (int q = a.tile_pick_start[tile]; q < a.tile_pick_start[tile + 1]; ++q) {
      const int id = a.tile_pick_ids[q];
      if (a.pick_row[id] == v) g += gj[(NJ + id) * 3 + c];
    }
    sgv[v][c] = g;
  }
  __syncthreads();
  const float* dirs = a.dirs_tiled + (size_t)tile * 3 * TILE_V * KSTRIDE;
  if (tid < TILE_V * 3) {      // v_posed of the tile (recomputed: the forward does not keep it)
    const int v = tid % TILE_V, c = tid / TILE_V;
    const float* row = dirs + ((size_t)c * TILE_V + v) * KSTRIDE;
    float acc = 0.f;
    for (int k = 0; k <= K_ONE; ++k) acc = fmaf(row[k], sfeat[k], acc);
    svp[v][c] = acc;
  }
  __syncthreads();
  if (tid < TILE_V) {
    const int v = tid, vg = tile * TILE_V + v;
    float T[12];
    for (int e = 0; e < 12; ++e) {
      float acc = 0.f;
      for (int j = 0; j < NJ; ++j) acc = fmaf(a.w_tiled[(size_t)vg * NJ + j], sA[e * NJ + j], acc);
      T[e] = acc;
    }
    const float g[3] = {sgv[v][0], sgv[v][1], sgv[v][2]};
    for (int c = 0; c < 3; ++c) sgvp[v][c] = T[0 * 4 + c] * g[0] + T[1 * 4 + c] * g[1] + T[2 * 4 + c] * g[2];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) sgT[v][r * 4 + c] = g[r] * svp[v][c];
      sgT[v][r * 4 + 3] = g[r];
    }
  }
  __syncthreads();
  float* out = a.partial + ((size_t)tile * a.Bpad + b) * BW_PART;
  for (int idx = tid; idx < BW_PART; idx += 256) {
    float acc = 0.f;
    if (idx <= K_ONE) {
      for (int c = 0; c < 3; ++c)
        for (int v = 0; v < TILE_V; ++v) acc = fmaf(dirs[((size_t)c * TILE_V + v) * KSTRIDE + idx], sgvp[v][c], acc);
    } else {
      const int e = (idx - 218) / NJ, j = (idx - 218) % NJ;
      for (int v = 0; v < TILE_V; ++v) acc = fmaf(a.w_tiled[(size_t)(tile * TILE_V + v) * NJ + j], sgT[v][e], acc);
    }
    out[idx] = acc;
  }
}

// one workgroup per frame: reduce the tile partials, then the kinematic chain in reverse
__global__ __launch_bounds__(256) void smpl_bwd_chain_kernel(BwdGenArgs a) {
  __shared__ float sp[BW_PART];
  __shared__ float sR[NJ][9], sGR[NJ][9], sGt[NJ][3], sJ[NJ][3], gGR[NJ][9], gGt[NJ][3], gJ[NJ][3], gpose[NJ][3];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int idx = tid; idx < BW_PART; idx += 256) {
    float acc = 0.f;
    for (int t = 0; t < a.n_tiles; ++t) acc += a.partial[((size_t)t * a.Bpad + b) * BW_PART + idx];
    sp[idx] = acc;
  }
  if (tid < NJ) {
    const int j = tid;
    const float r[3] = {a.pose[(size_t)b * 72 + j * 3], a.pose[(size_t)b * 72 + j * 3 + 1], a.pose[(size_t)b * 72 + j * 3 + 2]};
    rodrigues_smplx(r, sR[j]);
    for (int c = 0; c < 3; ++c) {
      float v = a.j_template[j * 3 + c];
      for (int l = 0; l < a.num_betas; ++l) v = fmaf(a.j_shapedirs[(j * 3 + c) * a.num_betas + l], a.betas[(size_t)b * a.num_betas + l], v);
      sJ[j][c] = v;
    }
  }
  __syncthreads();
  if (tid != 0) return;
  const float* gj = a.gj54 + (size_t)b * GJ_STRIDE * 3;
  // forward chain (parents precede their children in the SMPL tree)
  for (int k = 0; k < NJ; ++k) {
    const int p = a.parents[k];
    if (p < 0) { for (int e = 0; e < 9; ++e) sGR[k][e] = sR[k][e]; for (int c = 0; c < 3; ++c) sGt[k][c] = sJ[k][c]; continue; }
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) sGR[k][r * 3 + c] = sGR[p][r * 3 + 0] * sR[k][0 * 3 + c] + sGR[p][r * 3 + 1] * sR[k][1 * 3 + c] + sGR[p][r * 3 + 2] * sR[k][2 * 3 + c];
      sGt[k][r] = sGR[p][r * 3 + 0] * (sJ[k][0] - sJ[p][0]) + sGR[p][r * 3 + 1] * (sJ[k][1] - sJ[p][1]) + sGR[p][r * 3 + 2] * (sJ[k][2] - sJ[p][2]) + sGt[p][r];
    }
  }
  // A_k = [G_R | G_t - G_R J_k],  chain joint k = G_t
  for (int k = 0; k < NJ; ++k) {
    float gAt[3];
    for (int r = 0; r < 3; ++r) gAt[r] = sp[218 + (r * 4 + 3) * NJ + k];
    for (int r = 0; r < 3; ++r) {
      gGt[k][r] = gj[k * 3 + r] + gAt[r];
      for (int c = 0; c < 3; ++c) gGR[k][r * 3 + c] = sp[218 + (r * 4 + c) * NJ + k] - gAt[r] * sJ[k][c];
    }
    for (int c = 0; c < 3; ++c) gJ[k][c] = -(sGR[k][0 * 3 + c] * gAt[0] + sGR[k][1 * 3 + c] * gAt[1] + sGR[k][2 * 3 + c] * gAt[2]);
  }
  for (int k = NJ - 1; k >= 0; --k) {
    const int p = a.parents[k];
    float gR[9];
    if (p < 0) {
      for (int e = 0; e < 9; ++e) gR[e] = gGR[k][e];
      for (int c = 0; c < 3; ++c) gJ[k][c] += gGt[k][c];
    } else {
      const float d[3] = {sJ[k][0] - sJ[p][0], sJ[k][1] - sJ[p][1], sJ[k][2] - sJ[p][2]};
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          gR[r * 3 + c] = sGR[p][0 * 3 + r] * gGR[k][0 * 3 + c] + sGR[p][1 * 3 + r] * gGR[k][1 * 3 + c] + sGR[p][2 * 3 + r] * gGR[k][2 * 3 + c];
          gGR[p][r * 3 + c] += gGR[k][r * 3 + 0] * sR[k][c * 3 + 0] + gGR[k][r * 3 + 1] * sR[k][c * 3 + 1] + gGR[k][r * 3 + 2] * sR[k][c * 3 + 2] + gGt[k][r] * d[c];
        }
      for (int c = 0; c < 3; ++c) {
        gGt[p][c] += gGt[k][c];
        const float gd = sGR[p][0 * 3 + c] * gGt[k][0] + sGR[p][1 * 3 + c] * gGt[k][1] + sGR[p][2 * 3 + c] * gGt[k][2];
        gJ[k][c] += gd;
        gJ[p][c] -= gd;
      }
      for (int e = 0; e < 9; ++e) gR[e] += sp[10 + (k - 1) * 9 + e];      // pose blend shapes: feature = vec(R_k - I), k >= 1
    }
    const float r[3] = {a.pose[(size_t)b * 72 + k * 3], a.pose[(size_t)b * 72 + k * 3 + 1], a.pose[(size_t)b * 72 + k * 3 + 2]};
    rodrigues_smplx_bwd(r, gR, gpose[k]);
  }
  if (a.g_pose) for (int k = 0; k < NJ; ++k) for (int c = 0; c < 3; ++c) a.g_pose[(size_t)b * 72 + k * 3 + c] = gpose[k][c];
  if (a.g_betas)
    for (int l = 0; l < a.num_betas; ++l) {
      float acc = sp[l];
      for (int k = 0; k < NJ; ++k) for (int c = 0; c < 3; ++c) acc = fmaf(a.j_shapedirs[(k * 3 + c) * a.num_betas + l], gJ[k][c], acc);
      a.g_betas[(size_t)b * a.num_betas + l] = acc;
    }
}

}  // namespace glamr

// =====================================================================================================================
// C ABI
// =====================================================================================================================
using namespace glamr;

extern "C" int glamr_smpl_create(glamr_smpl** out, int V, int num_betas, const float* v_template, const float* shapedirs,
                                 const float* posedirs, const float* J_regressor, const float* lbs_weights,
                                 const float* J_regressor_extra, int n_extra, const int32_t* parents,
                                 const int32_t* extra_vertex_ids, int n_picked, const int32_t* joint_map, int n_out) {
  GLAMR_REQUIRE(out && v_template && shapedirs && posedirs && J_regressor && lbs_weights && parents && joint_map, "null argument");
  GLAMR_REQUIRE(V > 0 && num_betas > 0 && num_betas <= 10, "unsupported V=%d num_betas=%d (<=10)", V, num_betas);
  GLAMR_REQUIRE(n_extra >= 0 && n_extra <= MAX_EXTRA && n_picked >= 0 && n_picked <= MAX_PICKED && n_out > 0 && n_out <= 64,
                "unsupported n_extra=%d n_picked=%d n_out=%d", n_extra, n_picked, n_out);
  glamr_smpl* h = new (std::nothrow) glamr_smpl();
  if (!h) return fail(GLAMR_E_NOMEM, "out of host memory");
  std::memset(h, 0, sizeof(*h));
  h->V = V; h->num_betas = num_betas; h->n_extra = n_extra; h->n_picked = n_picked; h->n_out = n_out;

  // tree levels
  std::vector<int32_t> level(NJ, 0);
  int n_levels = 1;
  for (int j = 0; j < NJ; ++j) {
    if (parents[j] >= 0) {
      GLAMR_REQUIRE(parents[j] < j, "parents must be topologically ordered");
      level[j] = level[parents[j]] + 1;
    }
    n_levels = std::max(n_levels, level[j] + 1);
  }
  h->n_levels = n_levels;

  // which extra-regressed joints does the joint map reference?
  std::vector<int32_t> extra_slot(MAX_EXTRA, -1);
  int n_used = 0;
  for (int i = 0; i < n_out; ++i) {
    const int src = joint_map[i];
    GLAMR_REQUIRE(src >= 0 && src < NJ + n_picked + n_extra, "joint_map[%d]=%d out of range", i, src);
    if (src >= NJ + n_picked && extra_slot[src - NJ - n_picked] < 0) extra_slot[src - NJ - n_picked] = n_used++;
  }
  h->n_extra_used = n_used;
  GLAMR_REQUIRE(n_used == 0 || J_regressor_extra, "joint_map references extra joints but J_regressor_extra is null");

  // tilings: [tile][plane][row][KSTRIDE]; k: 0..9 shapedirs, 10..216 posedirs, 217 v_template
  // A (possibly virtual) vertex of a tiling: its blend directions over the feature vector [10 betas | 207 pose features | 1], its
  // skinning weights and its weight in every used extra-joint regressor.
  struct VirtVert { float d[3][218]; float w[NJ]; float jx[MAX_EXTRA]; };
  auto real_vertex = [&](int v, bool with_regressors) {
    VirtVert q;
    std::memset(&q, 0, sizeof(q));
    for (int r = 0; r < 3; ++r) {
      for (int l = 0; l < num_betas; ++l) q.d[r][l] = shapedirs[((size_t)v * 3 + r) * num_betas + l];
      for (int k = 0; k < 207; ++k) q.d[r][10 + k] = posedirs[(size_t)k * V * 3 + (size_t)v * 3 + r];
      q.d[r][217] = v_template[(size_t)v * 3 + r];
    }
    for (int j = 0; j < NJ; ++j) q.w[j] = lbs_weights[(size_t)v * NJ + j];
    if (with_regressors)
      for (int e = 0; e < n_extra; ++e)
        if (extra_slot[e] >= 0) q.jx[extra_slot[e]] = J_regressor_extra[(size_t)e * V + v];
    return q;
  };
  // pick_pos[p]: index in `vv` of picked vertex p
  auto build_tileset = [&](glamr_tileset* ts, const std::vector<VirtVert>& vv, const std::vector<int32_t>& pick_pos) -> int {
    const int n = (int)vv.size();
    ts->n_verts = n;
    ts->n_tiles = (n + TILE_V - 1) / TILE_V;
    ts->Vpad = ts->n_tiles * TILE_V;
    std::vector<float> dirs((size_t)ts->n_tiles * 3 * TILE_V * KSTRIDE, 0.0f);
    std::vector<float> w((size_t)ts->Vpad * NJ, 0.0f);
    std::vector<float> jx((size_t)std::max(1, n_used) * ts->Vpad, 0.0f);
    for (int i = 0; i < n; ++i) {
      const int tile = i / TILE_V, row = i % TILE_V;
      for (int r = 0; r < 3; ++r) {
        float* d = &dirs[(((size_t)tile * 3 + r) * TILE_V + row) * KSTRIDE];
        for (int k = 0; k < 218; ++k) d[k] = vv[i].d[r][k];          // d[K_ONE = 217] carries the template
      }
      for (int j = 0; j < NJ; ++j) w[(size_t)i * NJ + j] = vv[i].w[j];
      for (int e = 0; e < n_used; ++e) jx[(size_t)e * ts->Vpad + i] = vv[i].jx[e];
    }
    // picked vertices sorted by tile (CSR)
    std::vector<int32_t> pick_row(std::max(1, n_picked)), tstart(ts->n_tiles + 1, 0), tids(std::max(1, n_picked));
    for (int p = 0; p < n_picked; ++p) {
      pick_row[p] = pick_pos[p] % TILE_V;
      tstart[pick_pos[p] / TILE_V + 1]++;
    }
    for (int t = 0; t < ts->n_tiles; ++t) tstart[t + 1] += tstart[t];
    {
      std::vector<int32_t> cur(tstart.begin(), tstart.end() - 1);
      for (int p = 0; p < n_picked; ++p) tids[cur[pick_pos[p] / TILE_V]++] = p;
    }
    int rc;
    if ((rc = upload(&ts->dirs_tiled, dirs.data(), dirs.size()))) return rc;
    {
      // the same matrix as two fp16 planes: x = hi + lo to 2^-22 |x| (round to nearest, the remainder is exact in fp32)
      std::vector<unsigned short> planes((size_t)ts->n_tiles * 3 * 2 * TILE_V * KSH, 0);
      auto bits = [](float x) { const _Float16 h = (_Float16)x; unsigned short u; std::memcpy(&u, &h, 2); return u; };
      for (int t = 0; t < ts->n_tiles; ++t)
        for (int r = 0; r < 3; ++r)
          for (int row = 0; row < TILE_V; ++row)
            for (int k = 0; k < KTOT; ++k) {
              const float x = dirs[(((size_t)t * 3 + r) * TILE_V + row) * KSTRIDE + k];
              const float hi = (float)(_Float16)x;
              planes[((((size_t)t * 3 + r) * 2 + 0) * TILE_V + row) * KSH + k] = bits(x);
              planes[((((size_t)t * 3 + r) * 2 + 1) * TILE_V + row) * KSH + k] = bits(x - hi);
            }
      if ((rc = upload(&ts->dirs_h, planes.data(), planes.size()))) return rc;
    }
    if ((rc = upload(&ts->w_tiled, w.data(), w.size()))) return rc;
    if ((rc = upload(&ts->jx_used, jx.data(), jx.size()))) return rc;
    if ((rc = upload(&ts->pick_row, pick_row.data(), pick_row.size()))) return rc;
    if ((rc = upload(&ts->tile_pick_start, tstart.data(), tstart.size()))) return rc;
    if ((rc = upload(&ts->tile_pick_ids, tids.data(), tids.size()))) return rc;
    return GLAMR_OK;
  };
  for (int p = 0; p < n_picked; ++p)
    GLAMR_REQUIRE(extra_vertex_ids[p] >= 0 && extra_vertex_ids[p] < V, "extra_vertex_ids[%d] out of range", p);
  int rc;
  {
    std::vector<VirtVert> all((size_t)V);
    std::vector<int32_t> pick_pos(std::max(1, n_picked), 0);
    for (int v = 0; v < V; ++v) all[v] = real_vertex(v, true);
    for (int p = 0; p < n_picked; ++p) pick_pos[p] = extra_vertex_ids[p];
    if ((rc = build_tileset(&h->full, all, pick_pos))) return rc;
  }
  {
    // Joints only.  A regressed joint is  sum_v Jx_v T_v vposed_v  with T_v = sum_k W_vk A_k, i.e.  sum_k A_k . (sum_v Jx_v W_vk vposed_v):
    // the inner sum is linear in the feature vector, so per (joint, chain joint k) ONE virtual vertex with the weighted-average blend
    // directions, skinning weight s_k = sum_v Jx_v W_vk on bone k and regressor weight 1 gives the same joint -- 24 virtual vertices
    // per regressed joint instead of the hundreds in its support.  Picked vertices stay real.
    std::vector<VirtVert> vv;
    std::vector<int32_t> pick_pos(std::max(1, n_picked), 0);
    for (int p = 0; p < n_picked; ++p) { pick_pos[p] = (int)vv.size(); vv.push_back(real_vertex(extra_vertex_ids[p], false)); }
    std::vector<double> acc(3 * 218);
    for (int e = 0; e < n_extra; ++e) {
      if (extra_slot[e] < 0) continue;
      for (int k = 0; k < NJ; ++k) {
        double sk = 0.0;
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int v = 0; v < V; ++v) {
          const double c = (double)J_regressor_extra[(size_t)e * V + v] * (double)lbs_weights[(size_t)v * NJ + k];
          if (c == 0.0) continue;
          sk += c;
          for (int r = 0; r < 3; ++r) {
            double* a = &acc[r * 218];
            for (int l = 0; l < num_betas; ++l) a[l] += c * shapedirs[((size_t)v * 3 + r) * num_betas + l];
            for (int q = 0; q < 207; ++q) a[10 + q] += c * posedirs[(size_t)q * V * 3 + (size_t)v * 3 + r];
            a[217] += c * v_template[(size_t)v * 3 + r];
          }
        }
        if (sk == 0.0) continue;
        VirtVert q;
        std::memset(&q, 0, sizeof(q));
        for (int r = 0; r < 3; ++r) for (int i = 0; i < 218; ++i) q.d[r][i] = (float)(acc[r * 218 + i] / sk);
        q.w[k] = (float)sk;
        q.jx[extra_slot[e]] = 1.0f;
        vv.push_back(q);
      }
    }
    if (!vv.empty() && (rc = build_tileset(&h->joints, vv, pick_pos))) return rc;
  }
  // rest joints and their shape derivatives (J = J_regressor (v_template + shapedirs beta) is linear in beta)
  std::vector<float> jt(NJ * 3), js((size_t)NJ * 3 * num_betas);
  for (int j = 0; j < NJ; ++j)
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
      for (int v = 0; v < V; ++v) s += (double)J_regressor[(size_t)j * V + v] * v_template[(size_t)v * 3 + c];
      jt[j * 3 + c] = (float)s;
      for (int l = 0; l < num_betas; ++l) {
        double q = 0.0;
        for (int v = 0; v < V; ++v) q += (double)J_regressor[(size_t)j * V + v] * shapedirs[((size_t)v * 3 + c) * num_betas + l];
        js[((size_t)j * 3 + c) * num_betas + l] = (float)q;
      }
    }

  if ((rc = upload(&h->j_template, jt.data(), jt.size()))) return rc;
  if ((rc = upload(&h->j_shapedirs, js.data(), js.size()))) return rc;
  if ((rc = upload(&h->parents, parents, (size_t)NJ))) return rc;
  if ((rc = upload(&h->level, level.data(), level.size()))) return rc;
  if ((rc = upload(&h->joint_map, joint_map, (size_t)n_out))) return rc;
  if ((rc = upload(&h->extra_slot, extra_slot.data(), extra_slot.size()))) return rc;
  *out = h;
  return GLAMR_OK;
}

extern "C" int glamr_smpl_destroy(glamr_smpl* h) {
  if (!h) return GLAMR_OK;
  void* ptrs[] = {h->full.dirs_h, h->joints.dirs_h, h->j_template, h->j_shapedirs, h->parents, h->level, h->joint_map, h->extra_slot,
                  h->full.dirs_tiled, h->full.w_tiled, h->full.jx_used, h->full.pick_row, h->full.tile_pick_start, h->full.tile_pick_ids,
                  h->joints.dirs_tiled, h->joints.w_tiled, h->joints.jx_used, h->joints.pick_row, h->joints.tile_pick_start,
                  h->joints.tile_pick_ids};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete h;
  return GLAMR_OK;
}

namespace {
struct SmplWs { float *feat, *feat_h, *askin, *askin_h, *chain, *picked, *partial, *pivot; size_t total; int Bpad; };
SmplWs smpl_ws_layout(const glamr_smpl* h, int B, char* base) {
  SmplWs w{};
  w.Bpad = (B + TILE_F - 1) / TILE_F * TILE_F;
  size_t off = 0;
  auto take = [&](size_t nfloats) { float* p = reinterpret_cast<float*>(base + off); off = align_up(off + nfloats * sizeof(float), 256); return p; };
  w.feat = take((size_t)w.Bpad * KTOT);
  w.feat_h = take((size_t)w.Bpad * KTOT);      // two fp16 planes of the same rows
  w.askin = take((size_t)w.Bpad * 12 * NJ);
  w.askin_h = take((size_t)w.Bpad * 12 * 32);     // two fp16 planes, K padded to 32
  w.chain = take((size_t)w.Bpad * NJ * 3);
  w.picked = take((size_t)w.Bpad * std::max(1, h->n_picked) * 3);
  // regression partials of the pass that produces the joints: the 3-tile joints tileset when it exists, else the full mesh
  w.partial = take((size_t)(h->joints.n_tiles > 0 ? h->joints.n_tiles : h->full.n_tiles) * w.Bpad * std::max(1, h->n_extra_used) * 3);
  w.pivot = take((size_t)w.Bpad * 3);
  w.total = off;
  return w;
}
}  // namespace

// Frame tiles are split over gridDim.y.  The split decides how evenly the frame tiles fall on the (waves x chunks) wave slots of a vertex tile,
// how full the last round of workgroups leaves the 256 CUs (one workgroup per CU: the direction tile fills its LDS), and how often the 87 KB
// direction tile is re-staged (about a third of a tile's time).  Cost model, in units of one (vertex tile, frame tile) pair:
//   rounds x (0.35 + frame tiles per wave).
// 216 vertex tiles x 5 chunks = 1080 workgroups were 4.2 rounds of 15 tiles -- a fifth round at 22 % occupancy; 15 chunks = 12.7 rounds of 5.
// The 3-tile joints pass of a 300-frame call gets 2 chunks (one tile per wave instead of two in sequence on three CUs).
static int lbs_frame_chunks(int n_tiles, int n_ftiles, int nw) {
  const int cus = 256;
  int best = 1;
  double best_cost = 1e300;
  for (int gy = 1; gy <= std::max(1, (n_ftiles + nw - 1) / nw); ++gy) {
    const int rounds = (n_tiles * gy + cus - 1) / cus;
    const int per_wave = (n_ftiles + nw * gy - 1) / (nw * gy);
    const double cost = rounds * (0.35 + per_wave);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = gy; }
  }
  return best;
}

extern "C" size_t glamr_smpl_workspace_bytes(const glamr_smpl* h, int B) {
  if (!h || B <= 0) return 0;
  return smpl_ws_layout(h, B, nullptr).total;
}

extern "C" int glamr_smpl_forward(glamr_smpl* h, int B, const float* pose, const float* betas, const float* root_trans,
                                  const float* root_scale, float* verts, float* joints, int flags, void* workspace, void* stream_) {
  GLAMR_REQUIRE(h && pose && betas && joints && workspace, "null argument");
  GLAMR_REQUIRE(B > 0, "B must be positive");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SmplWs w = smpl_ws_layout(h, B, static_cast<char*>(workspace));
  const bool orig = (flags & GLAMR_SMPL_ORIG_JOINTS) != 0;
  // blend shapes and skinning transforms on the fp16 matrix cores (two-plane operands: fp32-grade products); GLAMR_SMPL_FP32_BLEND=1
  // selects the fp32-MFMA instances (and then the planes of the feature rows / joint transforms are not written)
  const bool planes = std::getenv("GLAMR_SMPL_FP32_BLEND") == nullptr && h->full.dirs_h != nullptr && (h->joints.n_tiles == 0 || h->joints.dirs_h != nullptr);
  // (rows [B, Bpad) of the operand arrays, read by the padded MFMA tiles, are written as zeros by the prep kernel)
  // (the fp16-plane instances read nothing but the planes: the fp32 rows are only written for the fp32-MFMA instances)
  PrepArgs pa{B, w.Bpad, h->num_betas, h->n_levels, 1, pose, (flags & GLAMR_SMPL_BODY_POSE_ONLY) ? 1 : 0, betas, h->j_template, h->j_shapedirs, h->parents, h->level, planes ? nullptr : w.feat,
              planes ? reinterpret_cast<unsigned short*>(w.feat_h) : nullptr, planes ? nullptr : w.askin, planes ? reinterpret_cast<unsigned short*>(w.askin_h) : nullptr, w.chain};
  hipLaunchKernelGGL(smpl_prep_kernel, dim3((w.Bpad + PREP_FRAMES - 1) / PREP_FRAMES), dim3(256), 0, stream, pa);
  const int n_ftiles = w.Bpad / TILE_F;
  const bool f16 = planes;
  const int nw = f16 ? 8 : 4;
  const size_t dirs_bytes = f16 ? (size_t)3 * 2 * TILE_V * KSH * sizeof(unsigned short) : (size_t)3 * TILE_V * KSTRIDE * sizeof(float);
  // one pass of the LBS kernel over a tileset: `ne` regressed joints accumulated (0: none), vertices written when `vout`, re-anchored when `pivot`
  auto lbs_pass = [&](const glamr_tileset& ts, int ne, float* vout, const float* pivot, bool skip_picks) -> int {
    LbsArgs la{B, h->V, ts.n_tiles, n_ftiles, ne, h->n_picked, ts.dirs_tiled, ts.w_tiled, ts.jx_used,
               ts.tile_pick_start, ts.tile_pick_ids, ts.pick_row, w.feat, ts.dirs_h, reinterpret_cast<const unsigned short*>(w.feat_h),
               reinterpret_cast<const unsigned short*>(w.askin_h), w.askin, vout, w.picked, w.partial, w.Bpad,
               pivot, root_trans, root_scale, skip_picks ? 1 : 0
#ifdef GLAMR_SMPL_EXPERIMENT
               , std::getenv("GLAMR_SMPL_EXP") ? std::atoi(std::getenv("GLAMR_SMPL_EXP")) : 0
#endif
    };
    const size_t lds = dirs_bytes + (vout ? (size_t)nw * (f16 ? TILE_F / 2 : TILE_F) * OUT_STRIDE * sizeof(float) : 0);
    // frame tiles are split over gridDim.y so that a launch has ~4 workgroups per CU even with few vertex tiles; every workgroup
    // re-stages its 87 KB direction tile, so a chunk keeps >= 8 frame tiles per wave
    const int gy = lbs_frame_chunks(ts.n_tiles, n_ftiles, nw);
    auto launch = [&](auto kern) -> int {
      {      // once per instance and process: the largest arena an instance can ask for (direction tile + eight half-tile transposes)
        static std::mutex amu;
        static std::set<const void*> raised;
        std::lock_guard<std::mutex> lock(amu);
        if (raised.insert(reinterpret_cast<const void*>(kern)).second)
          GLAMR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      }
      hipLaunchKernelGGL(kern, dim3(ts.n_tiles, gy), dim3(nw * 64), lds, stream, la);
      return GLAMR_OK;
    };
    switch (ne) {
      case 0: return f16 ? launch(smpl_lbs_kernel<0, true>) : launch(smpl_lbs_kernel<0>);
      case 1: return f16 ? launch(smpl_lbs_kernel<1, true>) : launch(smpl_lbs_kernel<1>);
      case 2: return f16 ? launch(smpl_lbs_kernel<2, true>) : launch(smpl_lbs_kernel<2>);
      case 3: return f16 ? launch(smpl_lbs_kernel<3, true>) : launch(smpl_lbs_kernel<3>);
      case 4: return f16 ? launch(smpl_lbs_kernel<4, true>) : launch(smpl_lbs_kernel<4>);
      default: return fail(GLAMR_E_UNSUPPORTED, "joint_map references %d extra-regressed joints (max 4 supported)", ne);
    }
  };
  int rc = GLAMR_OK;
  // joints: the picked vertices and the virtual vertices of the regressed joints (3 tiles) when that tileset exists, else the full mesh
  const bool two_pass = verts && h->joints.n_tiles > 0;
  const glamr_tileset& jts = h->joints.n_tiles > 0 ? h->joints : h->full;
  const bool need_tiles = !orig || (verts && !two_pass);      // (orig_joints: chain joints only, nothing to regress or pick)
  if (need_tiles && (rc = lbs_pass(jts, h->n_extra_used, two_pass ? nullptr : verts, nullptr, false))) return rc;
  FinishArgs fa{B, w.Bpad, (need_tiles && !orig) ? jts.n_tiles : 0, h->n_extra_used, h->n_picked, h->n_out, orig ? 1 : 0,
                w.chain, w.picked, w.partial, h->joint_map, h->extra_slot, root_trans, root_scale, joints, w.pivot};
  if (fa.n_tiles <= 4 && fa.n_out <= 32) hipLaunchKernelGGL(smpl_finish_small_kernel, dim3((B + 7) / 8), dim3(256), 0, stream, fa);
  else hipLaunchKernelGGL(smpl_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, fa);
  if (two_pass) {
    // the full mesh: no regression, no picks, vertices re-anchored on the pivot the finish kernel just wrote
    if ((rc = lbs_pass(h->full, 0, verts, root_trans ? w.pivot : nullptr, true))) return rc;
  } else if (verts && root_trans) {
    hipLaunchKernelGGL(smpl_anchor_kernel, dim3(8, B), dim3(256), 0, stream, B, h->V, w.pivot, root_trans, root_scale, verts);
  }
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

extern "C" int glamr_smpl_fk(glamr_smpl* h, int B, const float* pose, const float* root_trans, const float* root_scale,
                             float* joints, void* stream_) {
  GLAMR_REQUIRE(h && pose && joints, "null argument");
  GLAMR_REQUIRE(B > 0, "B must be positive");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PrepArgs pa{B, B, h->num_betas, h->n_levels, 0, pose, 0, nullptr, h->j_template, h->j_shapedirs, h->parents, h->level, nullptr, nullptr, nullptr, nullptr, joints};
  hipLaunchKernelGGL(smpl_prep_kernel, dim3((B + PREP_FRAMES - 1) / PREP_FRAMES), dim3(256), 0, stream, pa);
  if (root_trans) hipLaunchKernelGGL(smpl_fk_anchor_kernel, dim3(B), dim3(NJ), 0, stream, B, root_trans, root_scale, joints);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

extern "C" int glamr_smpl_backward_root(glamr_smpl* h, int B, const float* pose, const float* root_trans, const float* root_scale,
                                        const float* verts, const float* joints, const float* g_verts, const float* g_joints,
                                        float* g_orient, float* g_trans, float* g_scale, int flags, void* stream_) {
  GLAMR_REQUIRE(h && pose && g_orient, "null argument");
  GLAMR_REQUIRE(root_trans, "backward_root needs the re-anchored forward (root_trans != NULL): without it the rotation pivot "
                            "(the shaped root joint) is not recoverable from the outputs");
  GLAMR_REQUIRE((!g_verts || verts) && (!g_joints || joints), "gradient given without the matching forward output");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  BwdArgs ba{B, h->V, (flags & GLAMR_SMPL_ORIG_JOINTS) ? NJ : h->n_out, pose, root_trans, root_scale, verts, joints, g_verts, g_joints,
             g_orient, g_trans, g_scale};
  hipLaunchKernelGGL(smpl_backward_root_kernel, dim3(B), dim3(256), 0, stream, ba);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

namespace {
struct SmplBwdWs { float *feat, *askin, *chain, *gj54, *partial; size_t total; int Bpad; };
SmplBwdWs smpl_bwd_ws_layout(const glamr_smpl* h, int B, int with_verts, char* base) {
  SmplBwdWs w{};
  w.Bpad = (B + TILE_F - 1) / TILE_F * TILE_F;
  size_t off = 0;
  auto take = [&](size_t nfloats) { float* p = reinterpret_cast<float*>(base + off); off = align_up(off + nfloats * sizeof(float), 256); return p; };
  const glamr_tileset& ts = (!with_verts && h->joints.n_tiles > 0) ? h->joints : h->full;
  w.feat = take((size_t)w.Bpad * KTOT);
  w.askin = take((size_t)w.Bpad * 12 * NJ);
  w.chain = take((size_t)w.Bpad * NJ * 3);
  w.gj54 = take((size_t)w.Bpad * GJ_STRIDE * 3);
  w.partial = take((size_t)ts.n_tiles * w.Bpad * BW_PART);
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t glamr_smpl_backward_workspace_bytes(const glamr_smpl* h, int B, int with_vertex_gradient) {
  if (!h || B <= 0) return 0;
  return smpl_bwd_ws_layout(h, B, with_vertex_gradient, nullptr).total;
}

extern "C" int glamr_smpl_backward(glamr_smpl* h, int B, const float* pose, const float* betas, const float* root_trans, const float* root_scale,
                                   const float* verts, const float* joints, const float* g_verts, const float* g_joints, float* g_pose,
                                   float* g_betas, float* g_trans, float* g_scale, int flags, void* workspace, void* stream_) {
  GLAMR_REQUIRE(h && pose && betas && workspace && (g_verts || g_joints) && (g_pose || g_betas), "null argument");
  GLAMR_REQUIRE(B > 0, "B must be positive");
  GLAMR_REQUIRE(!g_scale || (root_trans && (!g_verts || verts) && (!g_joints || joints)),
                "g_scale needs the re-anchored forward and its outputs (verts / joints)");
  GLAMR_REQUIRE(!(g_trans || g_scale) || root_trans, "g_trans / g_scale only exist for the re-anchored forward (root_trans != NULL)");
  GLAMR_REQUIRE(h->n_picked <= MAX_PICKED, "more than %d picked vertices", MAX_PICKED);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool orig = (flags & GLAMR_SMPL_ORIG_JOINTS) != 0;
  const int with_verts = g_verts != nullptr;
  SmplBwdWs w = smpl_bwd_ws_layout(h, B, with_verts, static_cast<char*>(workspace));
  const glamr_tileset& ts = (!with_verts && h->joints.n_tiles > 0) ? h->joints : h->full;
  PrepArgs pa{B, B, h->num_betas, h->n_levels, 1, pose, 0, betas, h->j_template, h->j_shapedirs, h->parents, h->level, w.feat, nullptr, w.askin, nullptr, w.chain};
  hipLaunchKernelGGL(smpl_prep_kernel, dim3((B + PREP_FRAMES - 1) / PREP_FRAMES), dim3(256), 0, stream, pa);
  BwdGenArgs a{B, w.Bpad, h->V, ts.Vpad, ts.n_tiles, h->n_extra_used, h->n_picked, h->n_out, orig ? 1 : 0, h->num_betas,
               pose, betas, root_trans, root_scale, verts, joints, g_verts, g_joints, h->joint_map, h->extra_slot, h->parents,
               h->j_template, h->j_shapedirs, ts.dirs_tiled, ts.w_tiled, ts.jx_used, ts.tile_pick_start, ts.tile_pick_ids, ts.pick_row,
               w.feat, w.askin, w.gj54, w.partial, g_pose, g_betas, g_trans, g_scale};
  hipLaunchKernelGGL(smpl_bwd_anchor_kernel, dim3(B), dim3(256), 0, stream, a);
  // (orig_joints without a vertex gradient: only the chain carries gradient, the tile pass has nothing to add)
  const bool tiles = with_verts || !orig;
  if (tiles) hipLaunchKernelGGL(smpl_bwd_tile_kernel, dim3(ts.n_tiles, B), dim3(256), 0, stream, a);
  else a.n_tiles = 0;
  hipLaunchKernelGGL(smpl_bwd_chain_kernel, dim3(B), dim3(256), 0, stream, a);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}
