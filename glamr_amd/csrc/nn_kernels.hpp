// Building-block kernels of the two motion priors on gfx950: f32-MFMA linear layers with fused epilogues, masked multi-head
// attention for <= 64 tokens, residual + LayerNorm, and a register-resident LSTM recurrence.
//
// All activations are fp32 row-major [rows][ld] with ld a multiple of 4 (16-byte rows) so tiles move as dwordx4.
#pragma once
#include <cstdlib>
#include <cstdio>
#include <type_traits>
#include "common.hpp"

namespace glamr {
namespace nn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GT_M = 64, GT_N = 64, GT_K = 32, GT_LD = 36;   // LDS row stride 36 floats: conflict-free ds_read_b128 over 16 rows

enum Act { ACT_NONE = 0, ACT_RELU = 1 };

// Y[M,N] = act(X[M,K] W[N,K]^T + bias[N] + (rowbias ? rowbias[row / rows_per_group][N] : 0)) + (R ? R[M,N] : 0)
// (rows_per_group < 0: a per-POSITION table instead, rowbias[row % -rows_per_group][N] -- the folded position codes of a window's tokens)
//   X: ldx >= K (K multiple of 32, zero padded), W: [Npad][K] with Npad a multiple of 64 (zero rows), Y/R: ldy.
// Tile 64x64 per 256-thread workgroup, each wave owns a 32x32 accumulator (v_mfma_f32_32x32x2_f32); both operands are staged
// through LDS in 32-deep K chunks; lanes 0-31 / 32-63 consume the low / high 16 k of a chunk so every lane reads contiguous k.
struct GemmArgs {
  const float* X; const float* W; const float* bias; const float* rowbias; const float* R; float* Y;
  int M, N, K, ldx, ldy, ldr, rows_per_group, ldrb, act;
  const unsigned short* Ws = nullptr;     // the weights split into two fp16 planes in MFMA fragment order (see gemm_split_kernel)
  size_t ws_plane = 0;                    // elements per plane
  int x_frag = 0, y_frag = 0;             // gemm_free_kernel only (nn_free.hpp): X, resp. Y and R, in fragment-major order
};

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
  // (two LDS buffers: one barrier per 32-deep chunk; the chunks' global loads run GT_PF chunks ahead in registers.  Round 3 loaded a chunk,
  // waited, stored it and computed, with two barriers: a call with 64 rows -- the latent-optimisation mode makes 670 per iteration -- spent
  // 8 memory latencies on K = 256: 14.8 us)
  __shared__ __attribute__((aligned(16))) float sA[2][GT_M * GT_LD];
  __shared__ __attribute__((aligned(16))) float sB[2][GT_N * GT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int col = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * GT_M, n0 = blockIdx.x * GT_N;
  f32x16 acc = {0};
  // staging map: 512 float4 per operand tile, two per thread
  const int r0 = tid >> 3, c4 = (tid & 7) * 4;          // rows r0 and r0 + 32, k offset c4
  constexpr int GT_PF = 4;
  const int nchunks = a.K / GT_K;
  f32x4 va[GT_PF][2], vb[GT_PF][2];
  const float* xa[2];
  const float* xb[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = r0 + q * 32;
    xa[q] = a.X + (size_t)min(m0 + r, a.M - 1) * a.ldx + c4;
    xb[q] = a.W + (size_t)(n0 + r) * a.K + c4;
  }
  auto fetch = [&](int c, int slot) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      va[slot][q] = *reinterpret_cast<const f32x4*>(xa[q] + c * GT_K);
      vb[slot][q] = *reinterpret_cast<const f32x4*>(xb[q] + c * GT_K);
    }
  };
#pragma unroll
  for (int u = 0; u < GT_PF; ++u) if (u < nchunks) fetch(u, u);
  for (int c0 = 0; c0 < nchunks; c0 += GT_PF) {
#pragma unroll
    for (int u = 0; u < GT_PF; ++u) {
      const int c = c0 + u;
      if (c >= nchunks) break;
      float* bA = sA[c & 1];
      float* bB = sB[c & 1];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<f32x4*>(bA + (r0 + q * 32) * GT_LD + c4) = va[u][q];
        *reinterpret_cast<f32x4*>(bB + (r0 + q * 32) * GT_LD + c4) = vb[u][q];
      }
      __syncthreads();
      if (c + GT_PF < nchunks) fetch(c + GT_PF, u);
      const float* pa = bA + (wm * 32 + col) * GT_LD + half * 16;
      const float* pb = bB + (wn * 32 + col) * GT_LD + half * 16;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(pa + s4 * 4);
        const f32x4 w = *reinterpret_cast<const f32x4*>(pb + s4 * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[s], w[s], acc, 0, 0, 0);
      }
    }
  }
  // epilogue: lane owns column n0 + wn*32 + col, rows (r & 3) + 8 (r >> 2) + 4 half
  const int n = n0 + wn * 32 + col;
  if (n >= a.N) return;
  const float b = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (m >= a.M) continue;
    float v = acc[r] + b;
    if (a.rowbias) v += a.rowbias[(size_t)(a.rows_per_group > 0 ? m / a.rows_per_group : m % (-a.rows_per_group)) * a.ldrb + n];
    if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
    if (a.R) v += a.R[(size_t)m * a.ldr + n];
    a.Y[(size_t)m * a.ldy + n] = v;
  }
}

// The same contract for FEW rows (M < 2048, what launch_gemm does not hand to the split-fp16 kernel: a 50-frame window of one sequence -- a lone sequence's priors, the taped forward of the
// latent-optimisation mode: ~330 such products per pass).  gemm_kernel's wave walks K with one accumulator of v_mfma_f32_32x32x2_f32: K / 2 dependent
// instructions of 64 cycles each, 3.4 us at K = 256 and 6.8 at 512, on 4 - 12 workgroups of a 256-CU chip.  Here a workgroup owns a 32 x 32 tile, each
// wave a 16 x 16 accumulator walked with v_mfma_f32_16x16x4_f32 (K / 4 dependent instructions of ~40 cycles: 1.1 / 2.1 us) and four times as many
// workgroups share the rows.  Products and sums stay fp32 (the split-fp16 kernels move a lone sequence's results out of the single-sequence parity bounds).
typedef float f32x4acc __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs a) {
  constexpr int TS = 32;
  __shared__ __attribute__((aligned(16))) float sA[2][TS * GT_LD];
  __shared__ __attribute__((aligned(16))) float sB[2][TS * GT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
  f32x4acc acc = {0.f, 0.f, 0.f, 0.f};
  // staging map: 256 float4 per operand tile and chunk, one per thread
  const int r0 = tid >> 3, c4 = (tid & 7) * 4;
  constexpr int PF = 4;
  const int nchunks = a.K / GT_K;
  f32x4 va[PF], vb[PF];
  const float* xa = a.X + (size_t)min(m0 + r0, a.M - 1) * a.ldx + c4;
  const float* xb = a.W + (size_t)(n0 + r0) * a.K + c4;      // (W is padded to a multiple of 64 rows)
  auto fetch = [&](int c, int slot) {
    va[slot] = *reinterpret_cast<const f32x4*>(xa + c * GT_K);
    vb[slot] = *reinterpret_cast<const f32x4*>(xb + c * GT_K);
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) if (u < nchunks) fetch(u, u);
  for (int c0 = 0; c0 < nchunks; c0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int c = c0 + u;
      if (c >= nchunks) break;
      float* bA = sA[c & 1];
      float* bB = sB[c & 1];
      *reinterpret_cast<f32x4*>(bA + r0 * GT_LD + c4) = va[u];
      *reinterpret_cast<f32x4*>(bB + r0 * GT_LD + c4) = vb[u];
      __syncthreads();
      if (c + PF < nchunks) fetch(c + PF, u);
      // lane group g consumes k = 8 g .. 8 g + 7 of the chunk, one value per instruction (the instruction's four k slots are the four groups)
      const float* pa = bA + (wm * 16 + r16) * GT_LD + g * 8;
      const float* pb = bB + (wn * 16 + r16) * GT_LD + g * 8;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(pa), x1 = *reinterpret_cast<const f32x4*>(pa + 4);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(pb), w1 = *reinterpret_cast<const f32x4*>(pb + 4);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[s], w0[s], acc, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[s], w1[s], acc, 0, 0, 0);
    }
  }
  // epilogue: lane owns column n0 + 16 wn + r16, rows 16 wm + 4 g + r
  const int n = n0 + wn * 16 + r16;
  if (n >= a.N) return;
  const float b = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + wm * 16 + 4 * g + r;
    if (m >= a.M) continue;
    float v = acc[r] + b;
    if (a.rowbias) v += a.rowbias[(size_t)(a.rows_per_group > 0 ? m / a.rows_per_group : m % (-a.rows_per_group)) * a.ldrb + n];
    if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
    if (a.R) v += a.R[(size_t)m * a.ldr + n];
    a.Y[(size_t)m * a.ldy + n] = v;
  }
}

// ---- fp32 GEMM on the 16-bit matrix cores ----------------------------------------------------------------------------------------
// An fp32 number is, to 2^-22 of its size, the sum of two fp16 numbers (11 + 11 mantissa bits): x = hi + lo, hi = fp16(x), lo = fp16(x - hi)
// (the remainder is exact in fp32).  The product of two such numbers, dropping lo*lo (2^-22 of it), is  hi*hi + hi*lo + lo*hi:  THREE
// v_mfma_f32_32x32x16_f16 (fp32 accumulation) per k step.  Round 1 split into three bf16 planes (8 + 8 + 8 bits) and needed SIX
// products for the same 2^-24: half the matrix-core work now, for an error of 3 x 2^-22 = 7e-7 per product against the 1e-4 the
// priors' outputs are held to (tests/test_nets_gpu.py, unchanged).  Range: fp16 holds 6e-8 ... 65504; the networks' activations are
// LayerNorm / ReLU outputs of O(1), their weights O(0.05), and a low part that falls below fp16's normal range (|x| < 0.12) keeps an
// ABSOLUTE error of 3e-8 -- nothing against the tolerance.  Weights are split once when the model is created; activations are split
// while their tile is staged into LDS.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NPL = 2;          // planes
constexpr int BS_ROW = 80;      // bytes per LDS row of one plane: 32 halves + 16 pad -> conflict-free ds_read_b128 over 16 rows

__device__ __forceinline__ void split2(f32x2 x, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(x, f16x2);
  const f32x2 r1 = x - __builtin_convertvector(h, f32x2);          // exact
  const f16x2 l = __builtin_convertvector(r1, f16x2);
  hi = __builtin_bit_cast(unsigned, h); lo = __builtin_bit_cast(unsigned, l);
}

// Workgroup tile 128 x (64 WN): 4 waves as 2 x 2, each holding 2 x WN accumulator tiles; K chunks of 32.  Only the activations go
// through LDS (two fp16 planes, rows padded to 80 bytes: conflict-free ds_read_b128); the weight planes are stored in MFMA
// fragment order ([plane][32-column block][16-deep k step][lane][8 halves]) and every wave fetches its operands straight from L2 with
// one coalesced 1 KB load each, one k step ahead.  (Measured alternatives on the B=256 step: both operands through LDS 63 TFLOP/s --
// LDS bandwidth; no LDS at all 77 -- L1 bandwidth; this one 85; the plain fp32-MFMA kernel 76.)
template <int WN>
__global__ __launch_bounds__(256) void gemm_split_kernel(GemmArgs a) {
  GLAMR_CRITICAL_PATH_PRIO();
  constexpr int TN = 64 * WN;
  __shared__ __attribute__((aligned(16))) unsigned char sA[NPL][128 * BS_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, kh = lane >> 5;
  // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin by their linear id, and each XCD has its own L2.  All column
  // blocks of one 128-row block of the activations run on the SAME XCD (linear id -> xcd = id % 8 -> row block 8 * group + xcd), so the
  // tile of X is fetched into one L2 once instead of once per column block.
  const int ncb = gridDim.x, lin = blockIdx.y * ncb + blockIdx.x;
  const int grp = lin / (8 * ncb), within = lin % (8 * ncb);
  const int rb = grp * 8 + (within & 7), cb = within >> 3;
  if (rb * 128 >= a.M) return;
  const int m0 = rb * 128, n0 = cb * TN;
  const int ksteps = a.K / 16;
  f32x16 acc[2][WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x16){0};
  const uint4* wfrag[NPL][WN];
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const size_t nb = (size_t)(n0 / 32 + wn * WN + j);
      wfrag[p][j] = reinterpret_cast<const uint4*>(a.Ws + p * a.ws_plane) + (nb * ksteps) * 64 + lane;
    }
  f32x4 va[4];
  auto fetch_a = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = tid + q * 256, row = f >> 3, c4 = (f & 7) * 4;
      va[q] = *reinterpret_cast<const f32x4*>(a.X + (size_t)min(m0 + row, a.M - 1) * a.ldx + k0 + c4);
    }
  };
  uint4 vb[2][NPL][WN];                             // two k steps in flight
  auto fetch_b = [&](int buf, int kstep) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int j = 0; j < WN; ++j) vb[buf][p][j] = wfrag[p][j][(size_t)kstep * 64];
  };
  fetch_a(0);
  fetch_b(0, 0);
  for (int k0 = 0; k0 < a.K; k0 += 32) {
    __syncthreads();                                 // everybody is done reading the previous chunk
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = tid + q * 256, row = f >> 3, c4 = (f & 7) * 4;
      unsigned h0, l0, h1, l1;
      split2((f32x2){va[q][0], va[q][1]}, h0, l0);
      split2((f32x2){va[q][2], va[q][3]}, h1, l1);
      const int off = row * BS_ROW + c4 * 2;
      *reinterpret_cast<uint2*>(&sA[0][off]) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(&sA[1][off]) = make_uint2(l0, l1);
    }
    __syncthreads();
    if (k0 + 32 < a.K) fetch_a(k0 + 32);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kstep = k0 / 16 + ks;
      if (kstep + 1 < ksteps) fetch_b(ks ^ 1, kstep + 1);
      const int kb = ks * 32 + kh * 16;            // byte offset of this lane's 8 k values
      f16x8 xa[NPL][2], xb[NPL][WN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) xa[p][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(&sA[p][(wm * 64 + i * 32 + r) * BS_ROW + kb]));
#pragma unroll
        for (int j = 0; j < WN; ++j) xb[p][j] = __builtin_bit_cast(f16x8, vb[ks][p][j]);
      }
      // three products per accumulator, the two small ones first; consecutive MFMAs go to DIFFERENT accumulators (a dependent MFMA
      // waits for the previous one to drain)
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[PA[term]][i], xb[PB[term]][j], acc[i][j], 0, 0, 0);
    }
  }
  // epilogue: lane owns column n, rows (q & 3) + 8 (q >> 2) + 4 kh of each accumulator tile.
  // Round 5: a full 128-row tile leaves through straight-line code, one copy per combination of the three uniform options.  The general loop
  // below decides four uniform conditions PER ELEMENT (64 elements per lane: ~250 scalar branches, an integer division per element for the
  // per-row bias, 64-bit address chains -- ~6 000 of the kernel's 6 500 instructions against ~1 100 executed in the k loop: the matrix pipe sat
  // at 14 - 22 %); it stays for the last, partial row block.  Same operations per element in the same order.
  const bool full_tile = m0 + 128 <= a.M;
  if (full_tile) {
    // one straight-line copy per combination of the three uniform options (per-row bias, ReLU, residual), chosen once
    auto emit = [&](auto has_rb, auto relu, auto has_r) {
      const size_t ld = (size_t)a.ldy;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = n0 + (wn * WN + j) * 32 + r;
        if (n >= a.N) continue;                      // (per lane, decided once per column tile)
        const float b = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int mb = m0 + (wm * 2 + i) * 32 + 4 * kh;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int m = mb + (q & 3) + 8 * (q >> 2);
            float v = acc[i][j][q] + b;
            if (decltype(has_rb)::value) v += a.rowbias[(size_t)(a.rows_per_group > 0 ? m / a.rows_per_group : m % (-a.rows_per_group)) * a.ldrb + n];
            if (decltype(relu)::value) v = fmaxf(v, 0.0f);
            if (decltype(has_r)::value) v += a.R[(size_t)m * a.ldr + n];
            a.Y[(size_t)m * ld + n] = v;
          }
        }
      }
    };
    using T = std::true_type; using F = std::false_type;
    const int sel = (a.rowbias ? 4 : 0) | (a.act == ACT_RELU ? 2 : 0) | (a.R ? 1 : 0);
    switch (sel) {
      case 0: emit(F{}, F{}, F{}); break;
      case 1: emit(F{}, F{}, T{}); break;
      case 2: emit(F{}, T{}, F{}); break;
      case 3: emit(F{}, T{}, T{}); break;
      case 4: emit(T{}, F{}, F{}); break;
      case 5: emit(T{}, F{}, T{}); break;
      case 6: emit(T{}, T{}, F{}); break;
      default: emit(T{}, T{}, T{}); break;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = n0 + (wn * WN + j) * 32 + r;
    if (n >= a.N) continue;
    const float b = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = m0 + (wm * 2 + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
        if (m >= a.M) continue;
        float v = acc[i][j][q] + b;
        if (a.rowbias) v += a.rowbias[(size_t)(a.rows_per_group > 0 ? m / a.rows_per_group : m % (-a.rows_per_group)) * a.ldrb + n];
        if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
        if (a.R) v += a.R[(size_t)m * a.ldr + n];
        a.Y[(size_t)m * a.ldy + n] = v;
      }
    }
  }
}

// Y[row] = LayerNorm(X[row] (+ R[row])) * gamma + beta over D = 256 columns, eps = 1e-5; one wave per row.
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* X, const float* R, const float* gamma, const float* beta, float* Y,
                                                            int rows, int ld) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(X + (size_t)row * ld + lane * 4);
  if (R) { const f32x4 r = *reinterpret_cast<const f32x4*>(R + (size_t)row * ld + lane * 4); v += r; }
  float s = v[0] + v[1] + v[2] + v[3];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s * (1.0f / 256.0f);
  const f32x4 d = v - mean;
  float q = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 4), be = *reinterpret_cast<const f32x4*>(beta + lane * 4);
  *reinterpret_cast<f32x4*>(Y + (size_t)row * ld + lane * 4) = d * rstd * g + be;
}

// ---- fused row-block layers: activations stay on chip between the GEMMs of a block ------------------------------------------------
// The priors' linear layers are K = 256 / 512 deep: as separate GEMMs each is an HBM round trip of its activations (52 MB in, 52-157 MB
// out per 51 200 rows) and the matrix pipe idles 80 % of the time.  This kernel keeps a block of 64 rows in LDS through a whole
// sub-block of a transformer / MLP layer, with the full 256-wide output row in ONE workgroup so LayerNorm can follow in the epilogue:
//   ONE layer :  Y = [LN]( act(X W2^T + b2) + R )                                   e.g. attention out-projection + residual + LayerNorm
//   TWO layers:  H = relu(X W1^T + b1 [+ row bias]),  Y = [LN]( act(H W2^T + b2) + R )   feed-forward block (hidden 512 in two 256-column halves,
//                                                                                     never leaving LDS), the MLPs of the trajectory predictor
// X tile and hidden tile live in LDS as two fp16 planes (hi, lo; rows padded by 16 B: conflict-free ds_read_b128), i.e. directly as MFMA A
// operands -- no per-k-step conversion; weights come from L2 in fragment order as in gemm_split_kernel, one k step ahead.  8 waves, each
// owning 2 row tiles x 1 column tile of the 64 x 256 output (and of each hidden half).  LayerNorm arithmetic = add_layernorm_kernel.
struct RowsArgs {
  const float* X; int ldx, M, K1;                     // input rows [M][K1] (K1 multiple of 32, <= 256 with two layers, <= 512 with one)
  const unsigned short* W1s; size_t w1_plane;         // layer 1 weight planes (N = 512, K = K1), null for ONE layer
  const float* b1; const float* rowbias; int rpg, ldrb;
  const unsigned short* W2s; size_t w2_plane; int K2; // layer 2 (or the only layer): N = 256, K = K2 (512 after a hidden layer, else K1)
  const float* b2; int act2;
  const float* R; int ldr;                            // residual rows (added after act2) or null
  const float* gamma; const float* beta;              // LayerNorm over the 256 outputs, or null
  float* Y; int ldy;
};

// KS1 > 0: the k steps of the first GEMM phase are known at compile time (16 for the 256-wide transformer layers): every phase is then
// straight-line code -- around a loop back edge the compiler cannot count the weight fragments in flight and waits for all of them at
// the loop head, which defeats the three-steps-ahead prefetch.
// RT: 32-row tiles per workgroup.  2 = 64 rows (135 KB of LDS with two layers: ONE workgroup per CU, its staging and epilogue overlap nobody's
// MFMAs); 1 = 32 rows at half the LDS and <= 128 registers: TWO workgroups per CU, each wave then re-reads the weight fragments for half the
// rows (twice the L2 traffic per row, still a fraction of its bandwidth).
template <bool TWO, int KS1 = 0, int RT = 2>
__global__ __launch_bounds__(512, RT == 1 ? 4 : 1) void rows_fused_kernel(RowsArgs a) {      // (second argument: waves per SIMD)
  GLAMR_CRITICAL_PATH_PRIO();
  constexpr int RB = 32 * RT;                                       // rows per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NJ = 1;                                             // 8 waves: one 32-column tile each (x 2 row tiles)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, kg = lane >> 5;
  const int m0 = blockIdx.x * RB;
  const int K1 = a.K1, XS = (K1 + 8) * 2;                         // bytes per row of one X plane
  constexpr int HS = (256 + 8) * 2;                               // bytes per row of one hidden plane
  unsigned char* sX = smem;                                       // [2][64][XS]
  unsigned char* sH = smem + 2 * RB * XS;                         // [2][RB][HS] (two layers only)
  // the fp32 output rows [64][260] reuse the hidden planes (two layers) or the X planes (one layer, K1 = 256: 67.6 KB -> two workgroups per CU)
  float* sOut = reinterpret_cast<float*>(TWO ? sH : sX);
  // ---- X tile -> planes ---------------------------------------------------------------------------------------------------------
  for (int f = tid; f < RB * (K1 / 4); f += 512) {
    const int row = f / (K1 / 4), c4 = (f % (K1 / 4)) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.X + (size_t)min(m0 + row, a.M - 1) * a.ldx + c4);
    unsigned h0, l0, h1, l1;
    split2((f32x2){v[0], v[1]}, h0, l0);
    split2((f32x2){v[2], v[3]}, h1, l1);
    *reinterpret_cast<uint2*>(sX + row * XS + c4 * 2) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(sX + RB * XS + row * XS + c4 * 2) = make_uint2(l0, l1);
  }
  __syncthreads();
  // one GEMM phase: acc[i][j] += A(rows 32 i.., K) * W(cols of this wave's tiles j, k range) with A planes in LDS
  // KSC > 0: compile-time number of k steps (fully unrolled, no branches); KSC == 0: `ksteps_here` at run time
  auto phase = [&](auto ksc_tag, const unsigned char* sA, int AS, int ksteps_here, const unsigned short* Ws, size_t plane, int ksteps_w, int kstep0, int nb0, f32x16 (&acc)[RT][NJ]) {
    constexpr int KSC = decltype(ksc_tag)::value;
    const uint4* wf[2][NJ];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < NJ; ++j) wf[p][j] = reinterpret_cast<const uint4*>(Ws + p * plane) + ((size_t)(nb0 + j) * ksteps_w + kstep0) * 64 + lane;
    // weight fragments THREE k steps ahead (a wave has one other wave on its SIMD to hide an L2 round trip behind: 3 x 6 MFMAs do)
    uint4 vb[4][2][NJ];
    auto step = [&](int kstep, int u) {
      f16x8 xa[2][RT], xb[2][NJ];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < RT; ++i) xa[p][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sA + p * RB * AS + (i * 32 + c) * AS + kstep * 32 + kg * 16));
#pragma unroll
        for (int j = 0; j < NJ; ++j) xb[p][j] = __builtin_bit_cast(f16x8, vb[u][p][j]);
      }
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[PA[term]][i], xb[PB[term]][j], acc[i][j], 0, 0, 0);
    };
    if constexpr (KSC > 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int j = 0; j < NJ; ++j) if (d < KSC) vb[d][p][j] = wf[p][j][(size_t)d * 64];
#pragma unroll
      for (int kstep = 0; kstep < KSC; ++kstep) {
        if (kstep + 3 < KSC) {
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < NJ; ++j) vb[(kstep + 3) & 3][p][j] = wf[p][j][(size_t)(kstep + 3) * 64];
        }
        step(kstep, kstep & 3);
      }
    } else {
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int j = 0; j < NJ; ++j) if (d < ksteps_here) vb[d][p][j] = wf[p][j][(size_t)d * 64];
      for (int ks = 0; ks < ksteps_here; ks += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kstep = ks + u;
          if (kstep >= ksteps_here) break;
          if (kstep + 3 < ksteps_here) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
              for (int j = 0; j < NJ; ++j) vb[(u + 3) & 3][p][j] = wf[p][j][(size_t)(kstep + 3) * 64];
          }
          step(kstep, u);
        }
      }
    }
  };
  using ks1_t = std::integral_constant<int, KS1>;
  using ks16_t = std::integral_constant<int, 16>;
  f32x16 acc2[RT][NJ];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc2[i][j] = (f32x16){0};
  if (TWO) {
    for (int hh = 0; hh < 2; ++hh) {
      f32x16 acc1[RT][NJ];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc1[i][j] = (f32x16){0};
      phase(ks1_t{}, sX, XS, K1 / 16, a.W1s, a.w1_plane, K1 / 16, 0, hh * 8 + wave * NJ, acc1);
      if (hh) __syncthreads();                                    // everybody is done reading the previous hidden half
      // hidden half -> planes: lane owns column n, rows (q & 3) + 8 (q >> 2) + 4 kg of each tile.  (Row bias per group of rpg rows -- the decoder
      // of the trajectory predictor, one row per sequence: a block of RB rows spans at most two groups when rpg >= RB, so a row's group is a
      // comparison, not a division per element; launch_rows refuses a row bias with rpg < RB.)
      const int rb_g0 = m0 / a.rpg, rb_next = (rb_g0 + 1) * a.rpg;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nloc = (wave * NJ + j) * 32 + c, n = hh * 256 + nloc;
        const float b = a.b1 ? a.b1[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int row = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
            float v = acc1[i][j][q] + b;
            if (a.rowbias) v += a.rowbias[(size_t)(rb_g0 + (min(m0 + row, a.M - 1) >= rb_next ? 1 : 0)) * a.ldrb + n];
            v = fmaxf(v, 0.0f);
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            *reinterpret_cast<_Float16*>(sH + row * HS + nloc * 2) = h;
            *reinterpret_cast<_Float16*>(sH + RB * HS + row * HS + nloc * 2) = l;
          }
      }
      __syncthreads();
      phase(ks16_t{}, sH, HS, 16, a.W2s, a.w2_plane, a.K2 / 16, hh * 16, wave * NJ, acc2);
    }
  } else {
    phase(ks1_t{}, sX, XS, K1 / 16, a.W2s, a.w2_plane, a.K2 / 16, 0, wave * NJ, acc2);
  }
  __syncthreads();                                                // sH is free: output rows go there
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = (wave * NJ + j) * 32 + c;
    const float b = a.b2 ? a.b2[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
        float v = acc2[i][j][q] + b;
        if (a.act2 == ACT_RELU) v = fmaxf(v, 0.0f);
        sOut[row * 260 + n] = v;
      }
  }
  __syncthreads();
  // rows out: one wave per row, 4 columns per lane (coalesced 1 KB rows); residual and LayerNorm as add_layernorm_kernel
  for (int row = wave; row < RB; row += 8) {
    const int m = m0 + row;
    if (m >= a.M) break;
    f32x4 v = *reinterpret_cast<const f32x4*>(sOut + row * 260 + lane * 4);
    if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + (size_t)m * a.ldr + lane * 4);
    if (a.gamma) {
      float s_ = v[0] + v[1] + v[2] + v[3];
      for (int off = 32; off > 0; off >>= 1) s_ += __shfl_xor(s_, off);
      const float mean = s_ * (1.0f / 256.0f);
      const f32x4 d = v - mean;
      float q_ = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
      for (int off = 32; off > 0; off >>= 1) q_ += __shfl_xor(q_, off);
      const float rstd = 1.0f / sqrtf(q_ * (1.0f / 256.0f) + 1e-5f);
      const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + lane * 4), be = *reinterpret_cast<const f32x4*>(a.beta + lane * 4);
      v = d * rstd * g + be;
    }
    *reinterpret_cast<f32x4*>(a.Y + (size_t)m * a.ldy + lane * 4) = v;
  }
}

inline size_t rows_fused_lds(int K1, bool two, int rb = 64) {
  const size_t x = (size_t)2 * rb * (K1 + 8) * 2, h = (size_t)2 * rb * (256 + 8) * 2;
  return two ? x + h : (x > (size_t)rb * 260 * 4 ? x : (size_t)rb * 260 * 4);
}

// Multi-head attention for short sequences on the matrix cores: 8 heads x 32 dims, Lq, Lk <= 64; ONE wave per (sequence, head).
//   S^T = K Q^T / sqrt(32)   A = K rows (m = key), B = Q rows (n = query)        -> lane owns a QUERY column, its 16 registers per tile are keys
//   softmax over the keys    = over the lane's own registers + ONE exchange with lane ^ 32 (the other half of the rows)
//   O^T = V^T P^T            A = V columns (m = dim, from LDS), B = P^T straight from the score registers: the MFMA sums over its 16 k slots
//                            in any order, so the k slots are DEFINED as the keys the score registers already hold (slot j of half g =
//                            key 32 t + 16 u + (j & 3) + 8 (j >> 2) + 4 g) and V is gathered from LDS in that order -- no transposition
// fp32 operands as two fp16 planes, three v_mfma_f32_32x32x16_f16 per product (see the GEMM above): 48 MFMAs per head instead of
// ~50 x 50 x 64 scalar FMAs per query lane.  Q rows at Q[(b*Lq + i)*ldq + h*32] (q_shared: the same Lq rows for every sequence), K / V rows
// at K[(b*Lk + j)*ldk + h*32]; key_mask[b*Lk + j] != 0 -> key j is ignored (PyTorch key_padding_mask); a fully masked row yields zeros.
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)x[i];
    hi[i] = h;
    lo[i] = (_Float16)(x[i] - (float)h);
  }
}
__device__ __forceinline__ f32x16 mfma3(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}

__global__ __launch_bounds__(64) void attention_mfma_kernel(const float* Q, int ldq, const float* K, const float* V, int ldk, const unsigned char* key_mask,
                                                            float* O, int ldo, int Lq, int Lk, int q_shared) {
  __shared__ __attribute__((aligned(16))) float sV[64][36];      // rows padded to 36 floats: float4 stores, conflict-free column reads
  __shared__ unsigned char sM[64];
  const int b = blockIdx.x, h = blockIdx.y, lane = threadIdx.x, c = lane & 31, kg = lane >> 5;
  for (int idx = lane; idx < 64 * 8; idx += 64) {
    const int j = idx >> 3, d4 = (idx & 7) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < Lk) v = *reinterpret_cast<const f32x4*>(V + (size_t)(b * Lk + j) * ldk + h * 32 + d4);
    *reinterpret_cast<f32x4*>(&sV[j][d4]) = v;
  }
  sM[lane] = (lane < Lk) ? (key_mask ? key_mask[(size_t)b * Lk + lane] : 0) : 1;
  const bool two_k = Lk > 32, two_q = Lq > 32;
  // operand fragments: 8 consecutive dims (16 s + 8 kg ...) of row (32 tile + c)
  auto row_frag = [&](const float* base, int ld, int row, int nrows, int s, float scale, f16x8& hi, f16x8& lo) {
    float x[8];
    if (row < nrows) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + h * 32 + 16 * s + 8 * kg);
      const f32x4 w = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + h * 32 + 16 * s + 8 * kg + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[i] = u[i] * scale; x[4 + i] = w[i] * scale; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = 0.f;
    }
    split8(x, hi, lo);
  };
  const float* Kb = K + (size_t)b * Lk * ldk;
  const float* Qb = Q + (q_shared ? (size_t)0 : (size_t)b * Lq * ldq);
  f16x8 kh_[2][2], kl_[2][2], qh_[2][2], ql_[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (t == 0 || two_k) row_frag(Kb, ldk, 32 * t + c, Lk, s, 1.0f, kh_[t][s], kl_[t][s]);
      if (t == 0 || two_q) row_frag(Qb, ldq, 32 * t + c, Lq, s, 0.17677669529663687f, qh_[t][s], ql_[t][s]);
    }
  f32x16 sc[2][2];                                  // [key tile][query tile]
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      sc[kt][qt] = (f32x16){0};
      if ((kt == 0 || two_k) && (qt == 0 || two_q)) {
#pragma unroll
        for (int s = 0; s < 2; ++s) sc[kt][qt] = mfma3(kh_[kt][s], kl_[kt][s], qh_[qt][s], ql_[qt][s], sc[kt][qt]);
      }
    }
  __syncthreads();                                  // sV / sM visible
  // softmax over the keys of each query column
  float inv[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    inv[qt] = 0.f;
    if (qt == 1 && !two_q) continue;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !two_k) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * kg;
        const float v = sM[key] ? -INFINITY : sc[kt][qt][r];
        sc[kt][qt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float den = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !two_k) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = (mx > -INFINITY) ? __expf(sc[kt][qt][r] - mx) : 0.f;
        sc[kt][qt][r] = p;
        den += p;
      }
    }
    den += __shfl_xor(den, 32);
    inv[qt] = den > 0.f ? 1.0f / den : 0.f;
  }
  // O^T = V^T P^T
  f32x16 oc[2] = {(f32x16){0}, (f32x16){0}};
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    if (kt == 1 && !two_k) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float vx[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vx[j] = sV[32 * kt + 16 * u + (j & 3) + 8 * (j >> 2) + 4 * kg][c];
      f16x8 vh, vl;
      split8(vx, vh, vl);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        if (qt == 1 && !two_q) continue;
        float px[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) px[j] = sc[kt][qt][8 * u + j];
        f16x8 ph, pl;
        split8(px, ph, pl);
        oc[qt] = mfma3(vh, vl, ph, pl, oc[qt]);
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = 32 * qt + c;
    if ((qt == 1 && !two_q) || query >= Lq) continue;
    float* op = O + (size_t)(b * Lq + query) * ldo + h * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {oc[qt][4 * g] * inv[qt], oc[qt][4 * g + 1] * inv[qt], oc[qt][4 * g + 2] * inv[qt], oc[qt][4 * g + 3] * inv[qt]};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * kg) = v;
    }
  }
}

// ---- QKV projection + attention in one pass: Q, K, V never exist in memory ---------------------------------------------------------
// One workgroup per sequence, one wave per head (8 x 32).  The rows of the sequence sit in LDS as fp16 planes (MFMA operands).  The
// projections are computed in the ORIENTATION the attention MFMAs consume, so nothing is transposed through LDS:
//   Q^T_h = W_q,h X^T  (A = weight fragment: lane = dim, B = row fragment: lane = token)  -> lane owns a TOKEN, its 16 registers are dims
//   K^T_h likewise.  S^T = K Q^T contracts over the 32 dims in any order, so the k slots of that MFMA are DEFINED as the dims the
//   registers hold (slot j of step u, half g = dim 16 u + (j & 3) + 8 (j >> 2) + 4 g) -- for both operands alike.
//   V_h = X W_v,h^T      (A = row fragment: lane = token, B = weight fragment: lane = dim)  -> lane owns a DIM, registers are tokens in
//   exactly the key order the probabilities come out of the softmax in: the A operand of O^T = V^T P^T as it stands.
// Self-attention: Xq = Xkv.  Cross-attention: queries from the decoder rows, keys / values from the context rows.  Saves the write and
// re-read of the (3 x 256)-wide QKV rows (157 + 157 MB per 51 200 rows) and a launch per attention block.
struct QkvAttnArgs {
  const float* Xq; int Lq;                       // [B][Lq][256] rows the queries are projected from
  const float* Xkv; int Lk;                      // [B][Lk][256] rows the keys / values are projected from (== Xq for self-attention)
  const unsigned short* Wq; size_t wq_plane; const float* bq; int q_nb0;       // fragment-ordered planes holding the query projection; its first 32-column block
  const unsigned short* Wkv; size_t wkv_plane; const float* bkv; int k_nb0, v_nb0;
  const unsigned char* mask;                     // [B][Lk] key-padding mask or null
  float* O; int ldo;                             // [B][Lq][256]
};

#ifndef GLAMR_QKV_WAVES
#define GLAMR_QKV_WAVES 1      // waves per SIMD the register allocation aims at (development knob)
#endif
__global__ __launch_bounds__(512, GLAMR_QKV_WAVES) void qkv_attention_kernel(QkvAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int XS = (256 + 8) * 2;                               // bytes per row of one plane
  constexpr int KS = 16;                                          // k steps of the projections (K = 256)
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, c = lane & 31, kg = lane >> 5;
  const int b = blockIdx.x, Lq = a.Lq, Lk = a.Lk;
  const bool self = a.Xq == a.Xkv;
  unsigned char* sKV = smem;                                      // [2][64][XS]
  unsigned char* sQ = self ? sKV : smem + 2 * 64 * XS;            // [2][64][XS] (rows >= L are zero)
  __shared__ unsigned char sM[64];
  auto stage = [&](const float* X, int L, unsigned char* dst) {
    for (int f = tid; f < 64 * 64; f += 512) {
      const int row = f >> 6, c4 = (f & 63) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < L) v = *reinterpret_cast<const f32x4*>(X + ((size_t)b * L + row) * 256 + c4);
      unsigned h0, l0, h1, l1;
      split2((f32x2){v[0], v[1]}, h0, l0);
      split2((f32x2){v[2], v[3]}, h1, l1);
      *reinterpret_cast<uint2*>(dst + row * XS + c4 * 2) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(dst + 64 * XS + row * XS + c4 * 2) = make_uint2(l0, l1);
    }
  };
  stage(a.Xkv, Lk, sKV);
  if (!self) stage(a.Xq, Lq, sQ);
  if (tid < 64) sM[tid] = (tid < Lk) ? (a.mask ? a.mask[(size_t)b * Lk + tid] : 0) : 1;
  __syncthreads();
  const bool two_q = Lq > 32, two_k = Lk > 32;
  // row fragment (8 consecutive features of row 32 t + c, k step ks) and weight fragment (column block nb, k step ks)
  auto rowf = [&](const unsigned char* sX, int t, int ks, f16x8& hi, f16x8& lo) {
    hi = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sX + (t * 32 + c) * XS + ks * 32 + kg * 16));
    lo = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sX + 64 * XS + (t * 32 + c) * XS + ks * 32 + kg * 16));
  };
  auto wf = [&](const unsigned short* W, size_t plane, int nb, int ks, f16x8& hi, f16x8& lo) {
    const uint4* p0 = reinterpret_cast<const uint4*>(W) + ((size_t)nb * KS + ks) * 64 + lane;
    const uint4* p1 = reinterpret_cast<const uint4*>(W + plane) + ((size_t)nb * KS + ks) * 64 + lane;
    hi = __builtin_bit_cast(f16x8, *p0);
    lo = __builtin_bit_cast(f16x8, *p1);
  };
  // transposed projection of BOTH 32-token tiles in one pass over the weights: lane = token, registers = dims of head h (+ bias, x scale),
  // as split k-step fragments.  One weight fragment feeds the two tiles (half the L2 requests of a pass per tile) and the two
  // accumulators are independent MFMA chains (a dependent MFMA waits for the previous one to drain).
  auto proj_t = [&](const unsigned char* sX, bool two, const unsigned short* W, size_t plane, int nb, const float* bias, float scale, f16x8 (&fh)[2][2], f16x8 (&fl)[2][2]) {
    f32x16 acc[2] = {(f32x16){0}, (f32x16){0}};
    f16x8 wh[4], wl[4];                              // weight fragments three k steps ahead (an L2 round trip is a few MFMAs long)
#pragma unroll
    for (int d = 0; d < 3; ++d) wf(W, plane, nb, d, wh[d], wl[d]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 3 < KS) wf(W, plane, nb, ks + 3, wh[(ks + 3) & 3], wl[(ks + 3) & 3]);
      f16x8 xh[2], xl[2];
      rowf(sX, 0, ks, xh[0], xl[0]);
      if (two) rowf(sX, 1, ks, xh[1], xl[1]);
      // small products first, the two tiles interleaved
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks & 3], xh[0], acc[0], 0, 0, 0);
      if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks & 3], xh[1], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 3], xl[0], acc[0], 0, 0, 0);
      if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 3], xl[1], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 3], xh[0], acc[0], 0, 0, 0);
      if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks & 3], xh[1], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 1 && !two) continue;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * u + j, dim = (r & 3) + 8 * (r >> 2) + 4 * kg;
          x[j] = (acc[t][r] + (bias ? bias[nb * 32 + dim] : 0.0f)) * scale;
        }
        split8(x, fh[t][u], fl[t][u]);
      }
    }
  };
  f16x8 qh_[2][2], ql_[2][2], kh_[2][2], kl_[2][2];
#ifndef GLAMR_QKV_NO_FUSE_QK
  if (self) {
    // self-attention: queries and keys are projections of the SAME rows -- one pass, the row fragments read once for four chains
    f32x16 acc[2][2] = {{(f32x16){0}, (f32x16){0}}, {(f32x16){0}, (f32x16){0}}};       // [q / k][tile]
    f16x8 wh[2][4], wl[2][4];
    const int nbq = a.q_nb0 + h, nbk = a.k_nb0 + h;
#pragma unroll
    for (int d = 0; d < 3; ++d) { wf(a.Wq, a.wq_plane, nbq, d, wh[0][d], wl[0][d]); wf(a.Wkv, a.wkv_plane, nbk, d, wh[1][d], wl[1][d]); }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 3 < KS) { wf(a.Wq, a.wq_plane, nbq, ks + 3, wh[0][(ks + 3) & 3], wl[0][(ks + 3) & 3]); wf(a.Wkv, a.wkv_plane, nbk, ks + 3, wh[1][(ks + 3) & 3], wl[1][(ks + 3) & 3]); }
      f16x8 xh[2], xl[2];
      rowf(sKV, 0, ks, xh[0], xl[0]);
      if (two_k) rowf(sKV, 1, ks, xh[1], xl[1]);
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (t == 1 && !two_k) continue;
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? wl[m][ks & 3] : wh[m][ks & 3], term == 1 ? xl[t] : xh[t], acc[m][t], 0, 0, 0);
          }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two_k) continue;
        const float* bias = m ? a.bkv : a.bq;
        const int nb = m ? nbk : nbq;
        const float scale = m ? 1.0f : 0.17677669529663687f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * u + j, dim = (r & 3) + 8 * (r >> 2) + 4 * kg;
            x[j] = (acc[m][t][r] + (bias ? bias[nb * 32 + dim] : 0.0f)) * scale;
          }
          if (m) split8(x, kh_[t][u], kl_[t][u]); else split8(x, qh_[t][u], ql_[t][u]);
        }
      }
  } else
#endif
  {
    proj_t(sQ, two_q, a.Wq, a.wq_plane, a.q_nb0 + h, a.bq, 0.17677669529663687f, qh_, ql_);
    proj_t(sKV, two_k, a.Wkv, a.wkv_plane, a.k_nb0 + h, a.bkv, 1.0f, kh_, kl_);
  }
  f32x16 sc[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      sc[kt][qt] = (f32x16){0};
      if ((kt == 0 || two_k) && (qt == 0 || two_q)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) sc[kt][qt] = mfma3(kh_[kt][u], kl_[kt][u], qh_[qt][u], ql_[qt][u], sc[kt][qt]);
      }
    }
  float inv[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    inv[qt] = 0.f;
    if (qt == 1 && !two_q) continue;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !two_k) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * kg;
        const float v = sM[key] ? -INFINITY : sc[kt][qt][r];
        sc[kt][qt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float den = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !two_k) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = (mx > -INFINITY) ? __expf(sc[kt][qt][r] - mx) : 0.f;
        sc[kt][qt][r] = p;
        den += p;
      }
    }
    den += __shfl_xor(den, 32);
    inv[qt] = den > 0.f ? 1.0f / den : 0.f;
  }
  // values of head h, one 32-key tile at a time: lane = dim, registers = keys; straight into O^T = V^T P^T
  f32x16 oc[2] = {(f32x16){0}, (f32x16){0}};
  const float bv = a.bkv ? a.bkv[(a.v_nb0 + h) * 32 + c] : 0.0f;
  f32x16 va[2] = {(f32x16){0}, (f32x16){0}};
  {
    f16x8 wh[4], wl[4];
#pragma unroll
    for (int d = 0; d < 3; ++d) wf(a.Wkv, a.wkv_plane, a.v_nb0 + h, d, wh[d], wl[d]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 3 < KS) wf(a.Wkv, a.wkv_plane, a.v_nb0 + h, ks + 3, wh[(ks + 3) & 3], wl[(ks + 3) & 3]);
      f16x8 xh[2], xl[2];
      rowf(sKV, 0, ks, xh[0], xl[0]);
      if (two_k) rowf(sKV, 1, ks, xh[1], xl[1]);
      va[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[0], wh[ks & 3], va[0], 0, 0, 0);
      if (two_k) va[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[1], wh[ks & 3], va[1], 0, 0, 0);
      va[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[0], wl[ks & 3], va[0], 0, 0, 0);
      if (two_k) va[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[1], wl[ks & 3], va[1], 0, 0, 0);
      va[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[0], wh[ks & 3], va[0], 0, 0, 0);
      if (two_k) va[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[1], wh[ks & 3], va[1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    if (kt == 1 && !two_k) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float vx[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vx[j] = va[kt][8 * u + j] + bv;
      f16x8 vh, vl;
      split8(vx, vh, vl);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        if (qt == 1 && !two_q) continue;
        float px[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) px[j] = sc[kt][qt][8 * u + j];
        f16x8 ph, pl;
        split8(px, ph, pl);
        oc[qt] = mfma3(vh, vl, ph, pl, oc[qt]);
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = 32 * qt + c;
    if ((qt == 1 && !two_q) || query >= Lq) continue;
    float* op = a.O + ((size_t)b * Lq + query) * a.ldo + h * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {oc[qt][4 * g] * inv[qt], oc[qt][4 * g + 1] * inv[qt], oc[qt][4 * g + 2] * inv[qt], oc[qt][4 * g + 3] * inv[qt]};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * kg) = v;
    }
  }
}

inline int launch_qkv_attention(hipStream_t st, int B, const QkvAttnArgs& a) {
  const bool self = a.Xq == a.Xkv;
  const size_t lds = (size_t)(self ? 1 : 2) * 2 * 64 * (256 + 8) * 2;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess)
      return fail(GLAMR_E_HIP, "hipFuncSetAttribute(qkv_attention_kernel) failed");
    attr_done = true;
  }
  hipLaunchKernelGGL(qkv_attention_kernel, dim3(B), dim3(512), lds, st, a);
  return GLAMR_OK;
}

// LSTM recurrence (nn.LSTMCell semantics, gate order i f g o, hidden 128).  The input projections
// G[t] = W_ih x_t + b_ih + b_hh are precomputed by a GEMM; this kernel adds W_hh h_{t-1} and applies the cell.
// One 512-thread workgroup per (sequence, direction): thread r keeps row r of W_hh (128 floats) in registers.
struct LstmArgs {
  const float* G;        // [n_seq][max_len][1024]: cols [0,512) forward gates, [512,1024) backward gates
  const float* Whh_f;    // [512][128]
  const float* Whh_b;
  const int* lens;       // [n_seq]
  float* H;              // [n_seq][max_len][256]: cols [0,128) forward h_t, [128,256) backward h_t
  int max_len;
};

__global__ __launch_bounds__(512) void lstm_kernel(LstmArgs a) {
  __shared__ __attribute__((aligned(16))) float sh[128];
  __shared__ float sg[512];
  const int b = blockIdx.x, dir = blockIdx.y, r = threadIdx.x;
  const int n = a.lens[b];
  const float* Wrow = (dir ? a.Whh_b : a.Whh_f) + (size_t)r * 128;
  float w[128];
#pragma unroll
  for (int k = 0; k < 128; k += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(Wrow + k);
    w[k] = v[0]; w[k + 1] = v[1]; w[k + 2] = v[2]; w[k + 3] = v[3];
  }
  float c = 0.f;
  if (r < 128) sh[r] = 0.f;
  __syncthreads();
  const float* Gb = a.G + (size_t)b * a.max_len * 1024 + dir * 512 + r;
  float* Hb = a.H + (size_t)b * a.max_len * 256 + dir * 128;
  // (the gate pre-activation of the NEXT step is requested while this step runs: asked for at the top of its own step, every step began with a
  // memory round trip -- 300 of them per launch, two launches per call of a lone sequence's trajectory predictor)
  float gnext = n > 0 ? Gb[(size_t)(dir ? n - 1 : 0) * 1024] : 0.f;
  for (int s = 0; s < n; ++s) {
    const int t = dir ? (n - 1 - s) : s;
    float acc = gnext;
    if (s + 1 < n) gnext = Gb[(size_t)(dir ? n - 2 - s : s + 1) * 1024];
#pragma unroll
    for (int k = 0; k < 128; k += 4) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(sh + k);
      acc = fmaf(w[k], hv[0], acc); acc = fmaf(w[k + 1], hv[1], acc); acc = fmaf(w[k + 2], hv[2], acc); acc = fmaf(w[k + 3], hv[3], acc);
    }
    sg[r] = acc;
    __syncthreads();
    if (r < 128) {
      const float ig = 1.0f / (1.0f + expf(-sg[r]));
      const float fg = 1.0f / (1.0f + expf(-sg[128 + r]));
      const float gg = tanhf(sg[256 + r]);
      const float og = 1.0f / (1.0f + expf(-sg[384 + r]));
      c = fg * c + ig * gg;
      const float hnew = og * tanhf(c);
      sh[r] = hnew;
      Hb[(size_t)t * 256 + r] = hnew;
    }
    __syncthreads();
  }
}

// The same recurrence for large batches, on the matrix cores: one 512-thread workgroup advances 16 sequences of one direction
// together.  Per step, gates[16 x 512] = G[t] + h[16 x 128] . W_hh^T on the fp16 matrix cores with both operands split in two fp16
// planes (hi + lo, three products per accumulator: fp32-grade results, see gemm_split_kernel) as v_mfma_f32_16x16x32_f16: wave w owns
// hidden units [16 w, 16 w + 16) and therefore the four gate tiles {w, 8 + w, 16 + w, 24 + w}; its W_hh fragments (4 gates x 4 k steps
// x 2 planes x 8 halves) stay in 128 registers for the whole launch; h lives in LDS ALREADY SPLIT (the lane that produces a value
// writes its two halves), double-buffered, rows padded to 136 halves: the A operand of a step is 8 ds_read_b128 (lane = sequence
// n, k quarter q supplies k = 32 q + 8 j .. + 7 to the j-th k step -- the same permutation of k on both operands).  The cell update is
// lane-local: lane (n, q) holds the four gates of hidden unit 16 w + n for sequences 4 q .. 4 q + 3.  One barrier per step.
// 48 MFMAs per wave-step instead of the 128 v_mfma_f32_16x16x4_f32 of the fp32 version (3.4 of 5.6 us per step were matrix time on
// the ONE CU a workgroup owns); sigmoid / tanh through v_exp_f32 + v_rcp_f32 (1e-7 absolute, the parity bound of the priors is 1e-4).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)); }
__global__ __launch_bounds__(512) void lstm_mfma_kernel(LstmArgs a, int n_seq) {
  GLAMR_CRITICAL_PATH_PRIO();
  constexpr int HS = 136;                                                   // halves per LDS row
  __shared__ __attribute__((aligned(16))) _Float16 sh[2][2][16][HS];        // [buffer][plane][sequence][unit]
  const int dir = blockIdx.y, s0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int u0 = wave * 16;
  const float* Whh = dir ? a.Whh_b : a.Whh_f;
  f16x8 wh[4][4], wl[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* row = Whh + (size_t)(g * 128 + u0 + n) * 128 + q * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(row + j * 8), v1 = *reinterpret_cast<const f32x4*>(row + j * 8 + 4);
      const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      split8(x, wh[g][j], wl[g][j]);
    }
  }
  int len[4], maxlen = 0;
  const float* Gb[4];
  float* Hb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int seq = s0 + 4 * q + r;
    len[r] = seq < n_seq ? a.lens[seq] : 0;
    Gb[r] = a.G + (size_t)min(seq, n_seq - 1) * a.max_len * 1024 + dir * 512 + u0 + n;
    Hb[r] = a.H + (size_t)min(seq, n_seq - 1) * a.max_len * 256 + dir * 128 + u0 + n;
  }
  for (int m = 0; m < 16; ++m) { const int sq = s0 + m; if (sq < n_seq) maxlen = max(maxlen, a.lens[sq]); }
  for (int i = tid; i < 2 * 2 * 16 * HS; i += 512) (&sh[0][0][0][0])[i] = (_Float16)0.f;
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  float gnext[4][4];
  auto fetch_g = [&](int s) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = s < len[r];
      const int t = dir ? (len[r] - 1 - s) : s;
#pragma unroll
      for (int g = 0; g < 4; ++g) gnext[g][r] = act ? Gb[r][(size_t)t * 1024 + g * 128] : 0.f;
    }
  };
  fetch_g(0);
  __syncthreads();
  for (int s = 0; s < maxlen; ++s) {
    const int cur = s & 1;
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){gnext[g][0], gnext[g][1], gnext[g][2], gnext[g][3]};
    if (s + 1 < maxlen) fetch_g(s + 1);
    f16x8 ah[4], al[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ah[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(&sh[cur][0][n][q * 32 + j * 8]));
      al[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(&sh[cur][1][n][q * 32 + j * 8]));
    }
    // the two small products first; consecutive MFMAs go to different accumulators (a dependent MFMA waits for the previous one)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[j], wh[g][j], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[j], wl[g][j], acc[g], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[j], wh[g][j], acc[g], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * q + r;
      float hnew = 0.f;                                     // a finished sequence's state is never read again
      if (s < len[r]) {
        const float ig = fast_sigmoid(acc[0][r]);
        const float fg = fast_sigmoid(acc[1][r]);
        const float gg = fast_tanh(acc[2][r]);
        const float og = fast_sigmoid(acc[3][r]);
        c[r] = fg * c[r] + ig * gg;
        hnew = og * fast_tanh(c[r]);
        const int t = dir ? (len[r] - 1 - s) : s;
        Hb[r][(size_t)t * 256] = hnew;
      }
      const _Float16 hh = (_Float16)hnew;
      sh[cur ^ 1][0][m][u0 + n] = hh;
      sh[cur ^ 1][1][m][u0 + n] = (_Float16)(hnew - (float)hh);
    }
    __syncthreads();
  }
}

inline int launch_gemm(hipStream_t st, const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N, int K,
                       int act = ACT_NONE, const float* R = nullptr, int ldr = 0, const float* rowbias = nullptr, int rows_per_group = 1, int ldrb = 0,
                       const unsigned short* Ws = nullptr) {
  if (M <= 0) return GLAMR_OK;
  if (K % GT_K != 0 || ldx % 4 != 0) return fail(GLAMR_E_INVALID, "gemm: K=%d must be a multiple of %d and ldx=%d of 4", K, GT_K, ldx);
  static const bool log_shapes = std::getenv("GLAMR_GEMM_LOG") != nullptr;      // development aid (tools/gemm_profile.py)
  if (log_shapes) std::fprintf(stderr, "GEMM %d %d %d\n", M, N, K);
  GemmArgs a{X, W, bias, rowbias, R, Y, M, N, K, ldx, ldy, ldr, rows_per_group, ldrb, act};
  const int npad = (N + GT_N - 1) / GT_N * GT_N;
  static const bool no_split = std::getenv("GLAMR_GEMM_FP32_MFMA") != nullptr;      // development aid: force the plain fp32 kernel
  if (Ws && M >= 2048 && !no_split) {
    // tall activations: the split-bf16 kernel (small M is launch / latency bound either way)
    a.Ws = Ws;
    a.ws_plane = (size_t)npad * K;
    const bool narrow_only = std::getenv("GLAMR_GEMM_NARROW") != nullptr;      // development aid: 128x64 tiles only (116 registers)
    if (!narrow_only && npad % 128 == 0 && (size_t)(M / 128) * (npad / 128) >= 512)
      hipLaunchKernelGGL((gemm_split_kernel<2>), dim3(npad / 128, ((M + 127) / 128 + 7) / 8 * 8), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((gemm_split_kernel<1>), dim3(npad / 64, ((M + 127) / 128 + 7) / 8 * 8), dim3(256), 0, st, a);
    return GLAMR_OK;
  }
  // (measured and dropped, round 4: the K loop of a tile over two groups of four waves for grids of a handful of workgroups -- 13.3 against 12.4 us
  // per call at M = 50: these calls are launch + latency, not the chain of fp32 MFMAs)
  static const bool no_small = [] { const char* e = std::getenv("GLAMR_GEMM_SMALL16"); return e && e[0] == '0'; }();      // development aid: gemm_kernel for few rows as well
  if (!no_small) {      // (every product that is not on the split-fp16 kernels: a sequence alone and the same sequence inside a small batch then agree to the bit)
    hipLaunchKernelGGL(gemm_small_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, st, a);
    return GLAMR_OK;
  }
  hipLaunchKernelGGL(gemm_kernel, dim3((N + GT_N - 1) / GT_N, (M + GT_M - 1) / GT_M), dim3(256), 0, st, a);
  return GLAMR_OK;
}

// ONE layer (W1s == nullptr) or TWO layers on row blocks, see rows_fused_kernel.  Output width is 256.
inline int launch_rows(hipStream_t st, const float* X, int ldx, int M, int K1, const unsigned short* W1s, size_t w1_plane, const float* b1,
                       const float* rowbias, int rpg, int ldrb, const unsigned short* W2s, size_t w2_plane, int K2, const float* b2, int act2,
                       const float* R, int ldr, const float* gamma, const float* beta, float* Y, int ldy) {
  if (M <= 0) return GLAMR_OK;
  if (K1 % 32 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || (W1s && (K1 > 256 || K2 != 512)) || (!W1s && (K1 > 256 || K2 != K1)))
    return fail(GLAMR_E_INVALID, "fused rows: unsupported shape K1=%d K2=%d", K1, K2);
  RowsArgs a{X, ldx, M, K1, W1s, w1_plane, b1, rowbias, rpg, ldrb, W2s, w2_plane, K2, b2, act2, R, ldr, gamma, beta, Y, ldy};
  // 32-row blocks: two workgroups per CU (measured on 1024 x 300 frames, both priors: 20.6 ms against 21.7 with 64-row blocks);
  // GLAMR_ROWS_RT=2 selects the 64-row instances (A/B runs)
  static const int rt_env = std::getenv("GLAMR_ROWS_RT") ? std::atoi(std::getenv("GLAMR_ROWS_RT")) : 0;
  static const int rt_one = std::getenv("GLAMR_ROWS_RT_ONE") ? std::atoi(std::getenv("GLAMR_ROWS_RT_ONE")) : 0;      // (one-layer blocks only)
  const int rt_sel = (!W1s && (rt_one == 1 || rt_one == 2)) ? rt_one : rt_env;
  const int rt = rt_sel == 1 || rt_sel == 2 ? rt_sel : 1;
  const int rb = 32 * rt;
  if (rowbias && (rpg < rb || !W1s)) return fail(GLAMR_E_INVALID, "fused rows: a row bias needs two layers and groups of at least %d rows (got %d)", rb, rpg);
  const size_t lds = rows_fused_lds(K1, W1s != nullptr, rb);
  static bool attr_done = false;
  if (!attr_done) {
    const void* kerns[] = {reinterpret_cast<const void*>(rows_fused_kernel<true>), reinterpret_cast<const void*>(rows_fused_kernel<false>),
                           reinterpret_cast<const void*>(rows_fused_kernel<true, 16>), reinterpret_cast<const void*>(rows_fused_kernel<false, 16>),
                           reinterpret_cast<const void*>(rows_fused_kernel<true, 0, 1>), reinterpret_cast<const void*>(rows_fused_kernel<false, 0, 1>),
                           reinterpret_cast<const void*>(rows_fused_kernel<true, 16, 1>), reinterpret_cast<const void*>(rows_fused_kernel<false, 16, 1>)};
    for (const void* k : kerns)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess) return fail(GLAMR_E_HIP, "hipFuncSetAttribute(rows_fused_kernel) failed");
    attr_done = true;
  }
  // K1 = 256 (every transformer block): the instances whose phases are straight-line code
  const bool k256 = K1 == 256;
  const dim3 grid((M + rb - 1) / rb), block(512);
  if (rt == 1) {
    if (W1s) { if (k256) hipLaunchKernelGGL((rows_fused_kernel<true, 16, 1>), grid, block, lds, st, a); else hipLaunchKernelGGL((rows_fused_kernel<true, 0, 1>), grid, block, lds, st, a); }
    else { if (k256) hipLaunchKernelGGL((rows_fused_kernel<false, 16, 1>), grid, block, lds, st, a); else hipLaunchKernelGGL((rows_fused_kernel<false, 0, 1>), grid, block, lds, st, a); }
  } else {
    if (W1s) { if (k256) hipLaunchKernelGGL((rows_fused_kernel<true, 16>), grid, block, lds, st, a); else hipLaunchKernelGGL(rows_fused_kernel<true>, grid, block, lds, st, a); }
    else { if (k256) hipLaunchKernelGGL((rows_fused_kernel<false, 16>), grid, block, lds, st, a); else hipLaunchKernelGGL(rows_fused_kernel<false>, grid, block, lds, st, a); }
  }
  return GLAMR_OK;
}

}  // namespace nn
}  // namespace glamr
