// The priors' building blocks as kernels that run BESIDE resident workgroups of the optimiser stage.
//
// A stage workgroup (grecon.hip) holds 153 of a CU's 160 KB of LDS and five 256-register waves: SIMD 0 is full, SIMDs 1-3 are half
// empty and issue half of the time, the matrix pipes idle.  Measured with tools/coresidency_probe.py: the hardware places another
// kernel's workgroup into that space only if it (a) allocates NO LDS at all (1 KB is already refused), (b) has at most three waves (a
// fourth would need SIMD 0), (c) needs <= 128 registers per wave for two waves per free SIMD -- and only short launches interleave (a
// long-running kernel owns the SIMDs before the next stage workgroup arrives), and only with moderate L2 traffic (the stage re-reads 29 KB
// per scene-iteration from L2).  A train of such launches then runs at ~65 % of its stand-alone rate while the stage launch slows by
// ~10-20 %, against 0 % overlap for anything with LDS.  The fused row-block / QKV-attention kernels of nn_kernels.hpp are built around
// 130-140 KB of LDS, so every workgroup of theirs waits for a stage workgroup to retire: priors and stage ADD UP.  The kernels here give
// up on-chip fusion (activations round-trip through L2 / MALL between the GEMMs of a layer: alone they are the SLOWER ones) and get the
// stage's shadow in return.  Callers ask for them per call (GLAMR_NETS_COSCHEDULE); see DESIGN.md section 3 for the measurements.
//
// Conventions as in nn_kernels.hpp -- fp32 values as two fp16 planes, three v_mfma_f32_32x32x16_f16 per k step, weights in the fragment
// order of glamr_nets_create (`Lin::Ws`) -- except for the memory layout of the activations (fragment-major, below).  Every GEMM is
// computed TRANSPOSED,
//   Y^T = W X^T :  A = weight fragment (lane = output column), B = row fragment straight from global memory (lane = row),
// so that a lane's accumulator registers are 4 CONSECUTIVE output columns of ONE row (16-byte stores, no transposition through LDS).
#pragma once
#include <type_traits>
#include "nn_kernels.hpp"

namespace glamr {
namespace nn {

// Fragment-major ("X32") activations: a matrix [rows][ld] (ld a multiple of 16, rows padded to 32) is stored as
//   [row block of 32][16-column step][lane = row % 32 + 32 (col % 16 / 8)][col % 8]
// i.e. every 32 x 16 tile is the B operand of v_mfma_f32_32x32x16_f16 as it stands (before the fp16 split): a wave's operand fetch is ONE
// contiguous 2 KB read and its accumulator tile goes out as contiguous 1 KB writes.  With row-major rows the same kernels touch 32 cache
// lines per load / store instruction (16 or 32 useful bytes in each) and run 1.6x slower -- the L1's transaction rate, not bandwidth, was
// the limit (tools/gemm_free_bench.py).  Only the kernels of this file read or write the layout.
__host__ __device__ inline size_t x32_off(int row, int col, int ld) {
  return (((size_t)(row >> 5) * (ld >> 4) + (col >> 4)) * 64 + (row & 31) + 32 * ((col >> 3) & 1)) * 8 + (col & 7);
}

// Y = act(X W^T + bias + rowbias) + R, see GemmArgs (x_frag / y_frag: X, resp. Y and R, fragment-major).  One wave per 32 rows x (32 C)
// columns; the workgroup id -> tile map keeps all column blocks of a row block on one XCD (its L2 holds the X tile once).  KS: k steps of
// 16 at compile time (0: run time); PD: operands that many k steps ahead.  MAXW: waves per SIMD the register allocation aims at.
// (Measured and dropped: 2- and 3-wave workgroups marching through the same weight fragments with a barrier per k step, for L1 hits --
// no change; 64 x 64 / 64 x 128 tiles -- faster alone, slower beside the stage, where one 230-register wave per SIMD hides nothing.)
// The value of lane ^ 32 (the other half of the wave) through v_permlane32_swap_b32, a register-file exchange: __shfl_xor(x, 32) is
// ds_bpermute_b32, a trip through the CU's LDS pipeline -- which the kernels of this file share with another kernel's workgroups by design.
__device__ __forceinline__ float lane_xor32(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

template <int KS, int C, int PD, int MAXW>
__global__ __launch_bounds__(64, MAXW) void gemm_free_kernel(GemmArgs a) {
  const int lane = threadIdx.x, c = lane & 31, kg = lane >> 5;
  const int ncb = (a.N + 32 * C - 1) / (32 * C);
  const int lin = blockIdx.x, grp = lin / (8 * ncb), within = lin % (8 * ncb);
  const int rb = grp * 8 + (within & 7), cb = within >> 3;
  if (rb * 32 >= a.M) return;
  const int m0 = rb * 32, n0 = cb * 32 * C;
  const int ksteps = KS ? KS : a.K / 16;
  const int row = m0 + c;
  // per k step: 8 consecutive floats per lane; fragment-major: 2 KB per step, row-major: 64 B of each of the 32 rows
  const float* xp = a.x_frag ? a.X + ((size_t)rb * (a.ldx >> 4) * 64 + lane) * 8 : a.X + (size_t)min(row, a.M - 1) * a.ldx + 8 * kg;
  const int xstep = a.x_frag ? 512 : 16;
  const int ntile = (a.N + 63) / 64 * 2;                // 32-column tiles the planes hold (rows padded to 64)
  const uint4* wp[2][C];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < C; ++j) wp[p][j] = reinterpret_cast<const uint4*>(a.Ws + p * a.ws_plane) + ((size_t)min(cb * C + j, ntile - 1) * ksteps) * 64 + lane;
  f32x16 acc[C];
#pragma unroll
  for (int j = 0; j < C; ++j) acc[j] = (f32x16){0};
  f32x4 xr[PD + 1][2];
  uint4 wr[PD + 1][2][C];
  auto fetch = [&](int ks, int slot) {
    xr[slot][0] = *reinterpret_cast<const f32x4*>(xp + (size_t)ks * xstep);
    xr[slot][1] = *reinterpret_cast<const f32x4*>(xp + (size_t)ks * xstep + 4);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < C; ++j) wr[slot][p][j] = wp[p][j][(size_t)ks * 64];
  };
  auto step = [&](int slot) {
    const float x[8] = {xr[slot][0][0], xr[slot][0][1], xr[slot][0][2], xr[slot][0][3], xr[slot][1][0], xr[slot][1][1], xr[slot][1][2], xr[slot][1][3]};
    f16x8 xh, xl;
#ifdef GLAMR_GEMM_NOSPLIT      // development aid (tools/gemm_free_bench.py): what the operand split costs beside the stage -- wrong results
    xh = __builtin_bit_cast(f16x8, xr[slot][0]); xl = __builtin_bit_cast(f16x8, xr[slot][1]);
#else
    split8(x, xh, xl);
#endif
    // the two small products first; consecutive MFMAs go to different accumulators
#pragma unroll
    for (int j = 0; j < C; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wr[slot][1][j]), xh, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < C; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wr[slot][0][j]), xl, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < C; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wr[slot][0][j]), xh, acc[j], 0, 0, 0);
  };
  if constexpr (KS > 0) {
#pragma unroll
    for (int d = 0; d < PD; ++d) if (d < KS) fetch(d, d);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // the fetches stay AHEAD of the step: without the fences the scheduler sinks every load to just before its use (s_waitcnt vmcnt(0)
      // in front of most MFMAs: one k step of memory latency per k step)
      if (ks + PD < KS) fetch(ks + PD, (ks + PD) % (PD + 1));
      __builtin_amdgcn_sched_barrier(0);
      step(ks % (PD + 1));
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int d = 0; d < PD; ++d) if (d < ksteps) fetch(d, d);
    for (int k3 = 0; k3 < ksteps; k3 += PD + 1) {
#pragma unroll
      for (int u = 0; u <= PD; ++u) {
        const int ks = k3 + u;
        if (ks >= ksteps) break;
        if (ks + PD < ksteps) fetch(ks + PD, (u + PD) % (PD + 1));
        __builtin_amdgcn_sched_barrier(0);
        step(u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // epilogue: register q of tile j = output column n0 + 32 j + 8 (q >> 2) + 4 kg + (q & 3) of row m0 + c
  if (row >= a.M) return;
  const int rbrow = a.rowbias ? (a.rows_per_group > 0 ? row / a.rows_per_group : row % (-a.rows_per_group)) : 0;
  // Column tiles that lie whole inside N (or any tile of a fragment-major output, whose rows are padded) leave through STRAIGHT-LINE code, one
  // copy per combination of the uniform options (bias, per-row bias, ReLU, residual, output layout) chosen once: the general loop below decides
  // them per group of four columns -- 183 branches, ~1 400 of the kernel's 2 270 instructions.  (Measured twice in round 5: while the optimiser
  // stage's chain was the pipeline's critical one a faster infiller made the step SLOWER -- it takes more from the stage beside it -- and this
  // was not kept; since the skinning moved ahead of the predictor the infiller's chain is the critical one by 2.6 ms:
  // profiles/r05_pipeline_experiments.log.)
  if (n0 + 32 * C <= a.N || a.y_frag) {
    auto emit = [&](auto has_b, auto has_rb, auto relu, auto has_r, auto frag) {
#pragma unroll
      for (int j = 0; j < C; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = n0 + 32 * j + 8 * i + 4 * kg;
          if (n >= a.N) continue;                      // (fragment-major outputs only: N is a multiple of 4 there)
          f32x4 v = {acc[j][4 * i], acc[j][4 * i + 1], acc[j][4 * i + 2], acc[j][4 * i + 3]};
          if (decltype(has_b)::value) v += *reinterpret_cast<const f32x4*>(a.bias + n);
          if (decltype(has_rb)::value) v += *reinterpret_cast<const f32x4*>(a.rowbias + (size_t)rbrow * a.ldrb + n);
          if (decltype(relu)::value) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
          if (decltype(has_r)::value) v += *reinterpret_cast<const f32x4*>(a.R + (decltype(frag)::value ? x32_off(row, n, a.ldr) : (size_t)row * a.ldr + n));
          *reinterpret_cast<f32x4*>(a.Y + (decltype(frag)::value ? x32_off(row, n, a.ldy) : (size_t)row * a.ldy + n)) = v;
        }
    };
    using T = std::true_type; using F = std::false_type;
    auto pick_frag = [&](auto has_b, auto has_rb, auto relu, auto has_r) { if (a.y_frag) emit(has_b, has_rb, relu, has_r, T{}); else emit(has_b, has_rb, relu, has_r, F{}); };
    auto pick_r = [&](auto has_b, auto has_rb, auto relu) { if (a.R) pick_frag(has_b, has_rb, relu, T{}); else pick_frag(has_b, has_rb, relu, F{}); };
    auto pick_relu = [&](auto has_b, auto has_rb) { if (a.act == ACT_RELU) pick_r(has_b, has_rb, T{}); else pick_r(has_b, has_rb, F{}); };
    auto pick_rb = [&](auto has_b) { if (a.rowbias) pick_relu(has_b, T{}); else pick_relu(has_b, F{}); };
    if (a.bias) pick_rb(T{}); else pick_rb(F{});
    return;
  }
#pragma unroll
  for (int j = 0; j < C; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + 32 * j + 8 * i + 4 * kg;
      if (n >= a.N) continue;
      f32x4 v = {acc[j][4 * i], acc[j][4 * i + 1], acc[j][4 * i + 2], acc[j][4 * i + 3]};
      if (n + 3 < a.N || a.y_frag) {                      // (fragment-major rows are padded: the zero columns of the padded weights go out too)
        if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
        if (a.rowbias) v += *reinterpret_cast<const f32x4*>(a.rowbias + (size_t)rbrow * a.ldrb + n);
        if (a.act == ACT_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + (a.y_frag ? x32_off(row, n, a.ldr) : (size_t)row * a.ldr + n));
        *reinterpret_cast<f32x4*>(a.Y + (a.y_frag ? x32_off(row, n, a.ldy) : (size_t)row * a.ldy + n)) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= a.N) break;
          float s = v[e] + (a.bias ? a.bias[n + e] : 0.f);
          if (a.rowbias) s += a.rowbias[(size_t)rbrow * a.ldrb + n + e];
          if (a.act == ACT_RELU) s = fmaxf(s, 0.f);
          if (a.R) s += a.R[(size_t)row * a.ldr + n + e];
          a.Y[(size_t)row * a.ldy + n + e] = s;
        }
      }
    }
}

template <int C, int PD, int MAXW>
inline int launch_gemm_free_t(hipStream_t st, const GemmArgs& a) {
  const int ncb = (a.N + 32 * C - 1) / (32 * C), nrb = (a.M + 31) / 32;
  const dim3 grid((unsigned)((nrb + 7) / 8 * 8 * ncb)), block(64);
  switch (a.K / 16) {
    case 2: hipLaunchKernelGGL((gemm_free_kernel<2, C, PD, MAXW>), grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL((gemm_free_kernel<6, C, PD, MAXW>), grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL((gemm_free_kernel<8, C, PD, MAXW>), grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL((gemm_free_kernel<16, C, PD, MAXW>), grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL((gemm_free_kernel<32, C, PD, MAXW>), grid, block, 0, st, a); break;
    case 48: hipLaunchKernelGGL((gemm_free_kernel<48, C, PD, MAXW>), grid, block, 0, st, a); break;      // (K = 768: the backward of a QKV projection)
    default: hipLaunchKernelGGL((gemm_free_kernel<0, C, PD, MAXW>), grid, block, 0, st, a); break;
  }
  return GLAMR_OK;
}
inline int launch_gemm_free(hipStream_t st, const GemmArgs& a) {
  if ((a.x_frag && a.ldx % 16 != 0) || (a.y_frag && (a.ldy % 16 != 0 || (a.R && a.ldr % 16 != 0) || a.ldy < (a.N + 3) / 4 * 4)))
    return fail(GLAMR_E_INVALID, "fragment-major GEMM: ldx=%d / ldy=%d / ldr=%d must be multiples of 16", a.ldx, a.ldy, a.ldr);
#ifndef GLAMR_FREE_GEMM_CFG
#define GLAMR_FREE_GEMM_CFG 2, 1, 4      // development aid: column tiles per wave, k steps of operands ahead, waves per SIMD the allocation aims at
#endif
  // few rows (the backward products of the taped infiller: 50 rows on an otherwise idle chip): ONE column tile per wave -- twice the waves, half the
  // chain of MFMAs in each (every output element is the same sum in the same order: bits unchanged)
  if (a.M <= 128 && !a.x_frag && !a.y_frag) return launch_gemm_free_t<1, 1, 4>(st, a);
  return launch_gemm_free_t<GLAMR_FREE_GEMM_CFG>(st, a);
}

// Y[row] = LayerNorm(X[row] (+ R[row])) over 256 columns (add_layernorm_kernel's two-pass arithmetic), all three fragment-major; one wave per
// 32-row block, lane (row, half) owns 128 of its row's values -- three streaming passes over the 32 KB block instead of 128 registers
__global__ __launch_bounds__(64, 4) void ln_free_kernel(const float* X, const float* R, const float* gamma, const float* beta, float* Y, int rows) {
  const int lane = threadIdx.x, kg = lane >> 5;
  const size_t base = ((size_t)blockIdx.x * 16 * 64 + lane) * 8;
  auto load = [&](int s, f32x4& u, f32x4& w) {
    u = *reinterpret_cast<const f32x4*>(X + base + (size_t)s * 512);
    w = *reinterpret_cast<const f32x4*>(X + base + (size_t)s * 512 + 4);
    if (R) { u += *reinterpret_cast<const f32x4*>(R + base + (size_t)s * 512); w += *reinterpret_cast<const f32x4*>(R + base + (size_t)s * 512 + 4); }
  };
  float sum = 0.f;
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    f32x4 u, w;
    load(s, u, w);
    sum += ((u[0] + u[1]) + (u[2] + u[3])) + ((w[0] + w[1]) + (w[2] + w[3]));
  }
  sum += lane_xor32(sum);
  const float mean = sum * (1.0f / 256.0f);
  float sq = 0.f;
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    f32x4 u, w;
    load(s, u, w);
    u -= mean; w -= mean;
    sq += ((u[0] * u[0] + u[1] * u[1]) + (u[2] * u[2] + u[3] * u[3])) + ((w[0] * w[0] + w[1] * w[1]) + (w[2] * w[2] + w[3] * w[3]));
  }
  sq += lane_xor32(sq);
  const float rstd = 1.0f / sqrtf(sq * (1.0f / 256.0f) + 1e-5f);
  if ((int)blockIdx.x * 32 + (lane & 31) >= rows) return;
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    f32x4 u, w;
    load(s, u, w);
    const int col = 16 * s + 8 * kg;
    u = (u - mean) * rstd * *reinterpret_cast<const f32x4*>(gamma + col) + *reinterpret_cast<const f32x4*>(beta + col);
    w = (w - mean) * rstd * *reinterpret_cast<const f32x4*>(gamma + col + 4) + *reinterpret_cast<const f32x4*>(beta + col + 4);
    *reinterpret_cast<f32x4*>(Y + base + (size_t)s * 512) = u;
    *reinterpret_cast<f32x4*>(Y + base + (size_t)s * 512 + 4) = w;
  }
}

// attention_mfma_kernel without LDS: the value rows are gathered from global memory in the key order the probabilities sit in, the key
// mask is one ballot.  Same arithmetic in the same order.  K, V, O fragment-major; Q too unless q_shared (the prior's two token queries:
// one row-major table for every sequence).
__global__ __launch_bounds__(64, 4) void attention_free_kernel(const float* Q, int ldq, const float* K, const float* V, int ldk, const unsigned char* key_mask,
                                                               float* O, int ldo, int Lq, int Lk, int q_shared) {
  const int b = blockIdx.x >> 3, h = blockIdx.x & 7, lane = threadIdx.x, c = lane & 31, kg = lane >> 5;
  const unsigned char mk = (lane < Lk) ? (key_mask ? key_mask[(size_t)b * Lk + lane] : 0) : 1;
  const unsigned long long masked = __ballot(mk != 0);
  const bool two_k = Lk > 32, two_q = Lq > 32;
  auto row_frag = [&](const float* base, int ld, int grow, bool frag, bool valid, int s, float scale, f16x8& hi, f16x8& lo) {
    float x[8];
    if (valid) {
      const float* p = base + (frag ? x32_off(grow, h * 32 + 16 * s + 8 * kg, ld) : (size_t)grow * ld + h * 32 + 16 * s + 8 * kg);
      const f32x4 u = *reinterpret_cast<const f32x4*>(p), w = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[i] = u[i] * scale; x[4 + i] = w[i] * scale; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = 0.f;
    }
    split8(x, hi, lo);
  };
  f16x8 kh_[2][2], kl_[2][2], qh_[2][2], ql_[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (t == 0 || two_k) row_frag(K, ldk, b * Lk + 32 * t + c, true, 32 * t + c < Lk, s, 1.0f, kh_[t][s], kl_[t][s]);
      if (t == 0 || two_q) row_frag(Q, ldq, (q_shared ? 0 : b * Lq) + 32 * t + c, !q_shared, 32 * t + c < Lq, s, 0.17677669529663687f, qh_[t][s], ql_[t][s]);
    }
  f32x16 sc[2][2];
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
        const float v = ((masked >> key) & 1ull) ? -INFINITY : sc[kt][qt][r];
        sc[kt][qt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, lane_xor32(mx));
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
    den += lane_xor32(den);
    inv[qt] = den > 0.f ? 1.0f / den : 0.f;
  }
  f32x16 oc[2] = {(f32x16){0}, (f32x16){0}};
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    if (kt == 1 && !two_k) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float vx[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = 32 * kt + 16 * u + (j & 3) + 8 * (j >> 2) + 4 * kg;
        vx[j] = key < Lk ? V[x32_off(b * Lk + key, h * 32 + c, ldk)] : 0.f;
      }
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
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {oc[qt][4 * g] * inv[qt], oc[qt][4 * g + 1] * inv[qt], oc[qt][4 * g + 2] * inv[qt], oc[qt][4 * g + 3] * inv[qt]};
      *reinterpret_cast<f32x4*>(O + x32_off(b * Lq + query, h * 32 + 8 * g + 4 * kg, ldo)) = v;
    }
  }
}

}  // namespace nn
}  // namespace glamr
