// Workgroup runtime used by the templated per-scene / per-sequence algorithms on the device: thread id, barrier, block-wide sum,
// block-wide in-place prefix / suffix sum (wave shuffles + one LDS exchange per 64-lane wave).
#pragma once
#include <hip/hip_runtime.h>

namespace glamr {

// Wave scans through DPP row shifts (2: prefix and suffix sums, 1: prefix sums only, 0 / GLAMR_SCAN_SHUFFLE: the ds_bpermute shuffles of rounds
// 1-3).  Exact on integer data for every length, channel count, stride and direction (tools/scan_probe.hip, tests/test_scan_gpu.py); the
// optimiser stage is 4.8 % shorter with it (profiles/r04_scan_dpp_ab.log).
#if !defined(GLAMR_SCAN_DPP) && !defined(GLAMR_SCAN_SHUFFLE)
#define GLAMR_SCAN_DPP 2
#endif
#if defined(GLAMR_SCAN_DPP) && GLAMR_SCAN_DPP == 0
#undef GLAMR_SCAN_DPP
#endif
constexpr int RT_MAX_CH = 16;
constexpr int RT_SCAN_FLOATS = 2 * RT_MAX_CH * 16;    // two generations x channels x waves
constexpr int RT_RED_FLOATS = RT_SCAN_FLOATS + 16;     // + a region of its own for reduce_sum (scans may follow it without a barrier)

#ifdef GLAMR_PHASE_TIMING
__device__ unsigned long long g_phase_ticks[16];     // 100 MHz ticks per phase, summed over the iterations of workgroup 0
#endif

struct DeviceRT {
  float* red;   // LDS scratch of RT_RED_FLOATS floats
  float* ws = nullptr;   // this workgroup's workspace slice (constant-layout instances address their arrays relative to it)
  int gen = 0;
  static constexpr bool one_thread_per_frame = true;     // the launcher gives full-arena instances a thread per frame
  // base of the dynamic LDS arena: a link-time constant, so `arena() + compile-time offset` needs no register
  static __device__ __forceinline__ float* arena() {
    extern __shared__ __attribute__((aligned(16))) float glamr_dynamic_lds[];
    return glamr_dynamic_lds;
  }
  __device__ __forceinline__ float* workspace() const {      // uniform, pinned to scalar registers: loads take it as their scalar base
    const unsigned long long v = reinterpret_cast<unsigned long long>(ws);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    __attribute__((address_space(1))) float* g = (__attribute__((address_space(1))) float*)(((unsigned long long)hi << 32) | lo);
    asm("" : "+s"(g));
    return (float*)g;
  }
#ifdef GLAMR_PHASE_TIMING
  unsigned long long t_last = 0, acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  __device__ void mark_begin() { t_last = wall_clock64(); }
  template <int K> __device__ void mark() { const unsigned long long n = wall_clock64(); acc[K] += n - t_last; t_last = n; }
#ifndef GLAMR_PHASE_TIMING_THREAD
#define GLAMR_PHASE_TIMING_THREAD 0      // whose clock is reported: thread 0's wave shares its SIMD with the fifth wave of a 300-frame scene, thread 64's has a SIMD to itself
#endif
  __device__ void mark_end() { if (blockIdx.x == 0 && threadIdx.x == GLAMR_PHASE_TIMING_THREAD) for (int k = 0; k < 16; ++k) g_phase_ticks[k] = acc[k]; }
#endif  // scan_multi alternates between two halves of `red` so consecutive calls need no barrier in between
  // In-place inclusive prefix (reverse: suffix) sums of up to RT_MAX_CH arrays ch[c][i*stride], i in [0,n).  Element i is read and
  // written by thread i mod blockDim only, so a caller whose next phase touches only its own elements needs no barrier after it;
  // one barrier per chunk of blockDim elements happens inside.
  // LDS: the arrays are known to be in LDS (ds_read / ds_write instead of flat accesses), else in global memory
#ifdef GLAMR_SCAN_DPP
  template <int CTRL> static __device__ __forceinline__ float dpp0(float v) {      // lane's DPP source, 0 where it falls outside the row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
  }
  static __device__ __forceinline__ float lane_value(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
  }
  // inclusive prefix (reverse: suffix) sum over the 64 lanes of a wave
  static __device__ __forceinline__ float wave_scan_dpp(float v, bool reverse) {
    const int row = (threadIdx.x & 63) >> 4;
    if (!reverse) {
      v += dpp0<0x111>(v); v += dpp0<0x112>(v); v += dpp0<0x114>(v); v += dpp0<0x118>(v);      // row_shr:1,2,4,8
      const float t0 = lane_value(v, 15), t1 = lane_value(v, 31), t2 = lane_value(v, 47);      // row totals
      const float s1 = t0, s2 = t0 + t1, s3 = s2 + t2;
      float add = 0.f;                                   // selects, not branches (the nested conditional compiled to two exec-mask branches)
      add = row >= 1 ? s1 : add; add = row >= 2 ? s2 : add; add = row >= 3 ? s3 : add;
      v += add;
    } else {
      v += dpp0<0x101>(v); v += dpp0<0x102>(v); v += dpp0<0x104>(v); v += dpp0<0x108>(v);      // row_shl:1,2,4,8
      const float t1 = lane_value(v, 16), t2 = lane_value(v, 32), t3 = lane_value(v, 48);
      const float s2 = t3, s1 = t3 + t2, s0 = s1 + t1;
      float add = 0.f;
      add = row <= 2 ? s2 : add; add = row <= 1 ? s1 : add; add = row <= 0 ? s0 : add;
      v += add;
    }
    return v;
  }
#endif
  // inclusive prefix / suffix sum over the lanes of a wave in the summation order the build selects (`shuffle_order`: the Hillis-Steele
  // order of the ds_bpermute scans whatever the build -- see scan_multi's last argument)
  static __device__ __forceinline__ float wave_scan(float v, bool reverse, bool shuffle_order) {
    const int lane = threadIdx.x & 63;
#ifdef GLAMR_SCAN_DPP
    if (!shuffle_order && (!reverse || GLAMR_SCAN_DPP >= 2)) return wave_scan_dpp(v, reverse);
#endif
    if (!reverse) { for (int off = 1; off < 64; off <<= 1) { const float y = __shfl_up(v, off); if (lane >= off) v += y; } }
    else { for (int off = 1; off < 64; off <<= 1) { const float y = __shfl_down(v, off); if (lane + off < 64) v += y; } }
    return v;
  }
  // The same sums for values held in REGISTERS, one element per thread, a workgroup of exactly NW waves (instances that give every frame its
  // own thread and know their geometry at compile time): x[c] <- inclusive prefix (suffix) sum over the workgroup of channel c.  One barrier;
  // the NW wave totals of a channel are fetched with vector reads issued together (scan_multi walks them one dependent LDS round trip at a
  // time: 5 x ~100 cycles per channel on the iteration's critical path).  Same additions in the same order as scan_multi: same bits.
  template <int NW, int NCH>
  __device__ __forceinline__ void scan_regs(float (&x)[NCH], bool reverse, bool shuffle_order = false) {
    static_assert(NW >= 1 && NW <= 8 && NCH <= RT_MAX_CH, "scan_regs: geometry");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* r = red + (gen & 1) * (RT_MAX_CH * 16);
    gen++;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      x[c] = wave_scan(x[c], reverse, shuffle_order);
      if (lane == (reverse ? 0 : 63)) r[c * 16 + wave] = x[c];
    }
    __syncthreads();
    float t[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(r + c * 16);
      t[c][0] = a.x; t[c][1] = a.y; t[c][2] = a.z; t[c][3] = a.w;
      if (NW > 4) {
        const float4 b = *reinterpret_cast<const float4*>(r + c * 16 + 4);
        t[c][4] = b.x; t[c][5] = b.y; t[c][6] = b.z; t[c][7] = b.w;
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float pre = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) if (reverse ? (w > wave) : (w < wave)) pre += t[c][w];
      x[c] = x[c] + pre;
    }
  }
  // shuffle_order: this scan's wave part in the Hillis-Steele order of the ds_bpermute scans (rounds 1-3) even in a DPP build
  template <bool LDS = false>
  __device__ void scan_multi(float* const* ch, int nch, int n, int stride, bool reverse, bool shuffle_order = false) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    const int nchunks = (n + blockDim.x - 1) / blockDim.x;
    float carry[RT_MAX_CH];
    for (int c = 0; c < nch; ++c) carry[c] = 0.f;
    for (int k = 0; k < nchunks; ++k) {
      const int base = (reverse ? (nchunks - 1 - k) : k) * blockDim.x;
      const int i = base + threadIdx.x;                 // thread i mod blockDim always handles element i, in both directions
      const size_t idx = (size_t)i * stride;
      float* r = red + (gen & 1) * (RT_MAX_CH * 16);
      gen++;
      float x[RT_MAX_CH];
      for (int c = 0; c < nch; ++c) {
        float v = (i < n) ? elem<LDS>(ch[c], idx) : 0.f;
        // the wave scan through DPP row shifts inside the 16-lane rows (out-of-row sources read 0: bound_ctrl) and v_readlane of the three
        // row totals, instead of six ds_bpermute round trips.  GLAMR_SCAN_DPP=1: prefix sums only; 2: suffix sums as well
        v = wave_scan(v, reverse, shuffle_order);
        if (lane == (reverse ? 0 : 63)) r[c * 16 + wave] = v;
        x[c] = v;
      }
#if !defined(GLAMR_EXP_NOSYNC) || GLAMR_EXP_NOSYNC < 2
      __syncthreads();
#endif
      // the wave totals of every channel are fetched with two vector reads per channel, all issued before the first is used (a workgroup has at
      // most 8 waves): walking them one ds_read at a time was up to 8 dependent LDS round trips per channel.  Same additions in the same order.
      for (int c = 0; c < nch; ++c) {
        const float4 ta = *reinterpret_cast<const float4*>(r + c * 16), tb = *reinterpret_cast<const float4*>(r + c * 16 + 4);
        const float t8[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
        float pre = carry[c], tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          if (w >= nw) break;
          const float t = t8[w];
          tot += t;
          if (reverse ? (w > wave) : (w < wave)) pre += t;
        }
        if (i < n) elem<LDS>(ch[c], idx) = x[c] + pre;
        carry[c] += tot;
      }
    }
  }
  template <bool LDS> static __device__ __forceinline__ float& elem(float* base, size_t idx) {
    if (LDS) return *(float*)((__attribute__((address_space(3))) float*)base + idx);
    return *(float*)((__attribute__((address_space(1))) float*)base + idx);
  }
  __device__ __forceinline__ int tid() const { return threadIdx.x; }
  __device__ __forceinline__ int nthreads() const { return blockDim.x; }
#ifdef GLAMR_EXP_NOSYNC      // development aid (WRONG results): what the workgroup barriers of the iteration cost -- 1: rt.sync() only, 2: the scans' as well
  __device__ __forceinline__ void sync() const {}
#else
  __device__ __forceinline__ void sync() const { __syncthreads(); }
#endif
  __device__ float reduce_sum(float v) const {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    float* r = red + RT_SCAN_FLOATS;
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) s += r[w];
    return s;
  }
  // N block-wide sums at once (two barriers in all instead of two per value): v[k] <- sum over the workgroup of v[k]
  template <int N>
  __device__ void reduce_sum_n(float (&v)[N]) const {
    static_assert(N * 16 <= RT_SCAN_FLOATS, "reduce_sum_n: scratch too small");
#pragma unroll
    for (int k = 0; k < N; ++k)
      for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
    __syncthreads();                                   // the scan halves of `red` may still be read by a preceding scan
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
      for (int k = 0; k < N; ++k) red[k * 16 + wave] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += red[k * 16 + w];
      v[k] = s;
    }
    __syncthreads();                                   // before anybody overwrites `red` again
  }
  // in-place inclusive prefix (or suffix) sum over a[i*stride], i in [0,n); ends with a barrier
  __device__ void scan(float* a, int n, int stride, bool reverse) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    float carry = 0.f;
    for (int base = 0; base < n; base += blockDim.x) {
      const int i = base + threadIdx.x;
      const size_t idx = (size_t)(reverse ? (n - 1 - i) : i) * stride;
      float x = (i < n) ? a[idx] : 0.f;
      for (int off = 1; off < 64; off <<= 1) {
        const float y = __shfl_up(x, off);
        if (lane >= off) x += y;
      }
      __syncthreads();
      if (lane == 63) red[wave] = x;
      __syncthreads();
      float pre = carry;
      for (int w = 0; w < wave; ++w) pre += red[w];
      float tot = 0.f;
      for (int w = 0; w < nw; ++w) tot += red[w];
      if (i < n) a[idx] = x + pre;
      carry += tot;
    }
    __syncthreads();
  }
};


}  // namespace glamr
