// Stand-alone reproducer (no torch) of the corruption found in the two-stream pipeline: a kernel whose waves execute
// PACKED fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) on stream A, a kernel that runs chains of DEPENDENT
// v_mfma_f32_32x32x16_f16 in one-wave workgroups without LDS on stream B.  The victim's results are compared bit for bit with the same kernel
// run alone (victims 0, 1) or checked inside the kernel against the unpacked instruction on the same operands (victim 2).
// Build WITHOUT the library's -target-feature -packed-fp32-ops (the point is to have the packed instructions):
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/race_repro.hip glamr_amd/csrc/api_common.cpp -o tools/_race_repro
//   run:   tools/_race_repro [iterations] [frames] [victim] [trigger] [graphs: 0 / 1]
//     victim:  0 smpl_prep_kernel (the library's), 1 its kinematic chain alone, 2 v_pk_fma_f32 against v_fma_f32 on register operands (self-checking)
//     trigger: 0 gemm_free_kernel<6, 2, 1, 4> (independent accumulators), 1 none, 2 plain loads / stores, 3 attention_free_kernel (the library's),
//              4 chains of three dependent MFMAs on one accumulator, 5 the same number of MFMAs over four accumulators in turn
//   (LD_LIBRARY_PATH=<torch>/lib runs it on the HIP runtime PyTorch ships instead of /opt/rocm's)
#include "../glamr_amd/csrc/smpl.hip"
#include "../glamr_amd/csrc/nn_free.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(2); } } while (0)

__global__ void spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// a trigger without matrix instructions: one wave per workgroup, 100 registers of streaming loads
__global__ __launch_bounds__(64, 4) void stream_kernel(const float4* x, float4* y, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 * 8 + threadIdx.x;
  float4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = i + (size_t)k * 64 < n ? x[i + (size_t)k * 64] : float4{0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 8; ++k) if (i + (size_t)k * 64 < n) y[i + (size_t)k * 64] = float4{v[k].x * 2.f, v[k].y + 1.f, v[k].z, v[k].w};
}

typedef float f2 __attribute__((ext_vector_type(2)));
// victim 2: c <- a * c + b, once with v_pk_fma_f32 on a register pair and once with two v_fma_f32, 4096 times; mismatching lanes are counted
__global__ __launch_bounds__(256) void pk_selfcheck_kernel(unsigned* n_bad, unsigned* bad_lanes, int iters) {
  const int lane = threadIdx.x & 63;
  const float a0 = 0.99f - 1e-4f * (float)lane, a1 = 0.98f + 2e-4f * (float)lane, b0 = 0.01f * (float)(1 + (blockIdx.x & 7)), b1 = 0.02f;
  f2 a = {a0, a1}, b = {b0, b1}, c = {1.0f, 2.0f};
  float r0 = 1.0f, r1 = 2.0f;
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(c) : "v"(a), "v"(b));
    asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(r0) : "v"(a0), "v"(b0));
    asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(r1) : "v"(a1), "v"(b1));
    if (__float_as_uint(c.x) != __float_as_uint(r0) || __float_as_uint(c.y) != __float_as_uint(r1)) {
      atomicAdd(n_bad, 1u);
      atomicAdd(bad_lanes + lane, 1u);
      c.x = r0; c.y = r1;
    }
  }
  if (c.x == 12345.f) n_bad[1] = 1;
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// triggers 4 / 5: one wave per workgroup, no LDS; DEP: every MFMA accumulates into the result of the one before it
template <bool DEP>
__global__ __launch_bounds__(64, 4) void mfma_chain_kernel(float* out, int iters) {
  const int lane = threadIdx.x;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (float)((lane + i) & 15)); b[i] = (_Float16)(0.02f * (float)((lane * 3 + i) & 15)); }
  f16v acc[4] = {(f16v){0}, (f16v){0}, (f16v){0}, (f16v){0}};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      f16v& t = acc[DEP ? (k / 3) : (k & 3)];
      t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, t, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  out[(size_t)blockIdx.x * 64 + lane] = s;
}

// the kinematic chain of smpl_prep_kernel alone: same thread map (8 frames x 24 joints of 256 threads), same LDS footprint
__global__ __launch_bounds__(256) void chain_only_kernel(int B, const float* pose, const int32_t* parents, const int32_t* level, int n_levels, float* chain) {
  __shared__ float sG[8][24][12];
  __shared__ float sJ[8][24][3];
  __shared__ float pad[(30976 - 8 * 24 * 15 * 4) / 4];
  const int tid = threadIdx.x, fl = tid / 24, j = tid % 24, b = blockIdx.x * 8 + fl;
  const bool active = fl < 8 && b < B;
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Jr[3] = {0.f, 0.f, 0.f};
  if (active) {
    float r[3] = {0.f, 0.f, 0.f};
    if (j > 0) for (int c = 0; c < 3; ++c) r[c] = pose[(size_t)b * 69 + (j - 1) * 3 + c];
    glamr::rodrigues_smplx(r, R);
    for (int c = 0; c < 3; ++c) { Jr[c] = 0.01f * (float)(j * 3 + c) + r[c]; sJ[fl][j][c] = Jr[c]; }
    if (tid == 0) pad[0] = r[0];
  }
  __syncthreads();
  const int par = active ? parents[j] : -1, lev = active ? level[j] : -1;
  if (active && lev == 0) {
    for (int e = 0; e < 9; ++e) sG[fl][j][(e / 3) * 4 + (e % 3)] = R[e];
    for (int c = 0; c < 3; ++c) sG[fl][j][c * 4 + 3] = Jr[c];
  }
  __syncthreads();
  for (int L = 1; L < n_levels; ++L) {
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
  if (active) for (int c = 0; c < 3; ++c) chain[((size_t)b * 24 + j) * 3 + c] = sG[fl][j][c * 4 + 3];
  if (tid == 255 && pad[0] == 123.f) chain[0] = pad[1];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 40;
  const int B = argc > 2 ? std::atoi(argv[2]) : 307200;
  const int victim = argc > 3 ? std::atoi(argv[3]) : 0, trigger = argc > 4 ? std::atoi(argv[4]) : 0, graphs = argc > 5 ? std::atoi(argv[5]) : 0;
  { int rv = 0; CK(hipRuntimeGetVersion(&rv)); std::printf("HIP runtime %d, %s launches\n", rv, graphs ? "graph" : "plain"); }
  const int Bpad = (B + 31) / 32 * 32;
  static const int parents_h[24] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};
  int level_h[24], n_levels = 1;
  level_h[0] = 0;
  for (int j = 1; j < 24; ++j) { level_h[j] = level_h[parents_h[j]] + 1; n_levels = std::max(n_levels, level_h[j] + 1); }
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-0.3f, 0.3f);
  std::vector<float> pose_h((size_t)B * 69), betas_h((size_t)B * 10), jt(72), jsd(720);
  for (auto& x : pose_h) x = u(rng);
  for (auto& x : betas_h) x = u(rng);
  for (auto& x : jt) x = u(rng);
  for (auto& x : jsd) x = 0.1f * u(rng);
  float *pose, *betas, *jtd, *jsdd, *feat_h, *askin_h, *chain, *ref;
  int32_t *parents, *level;
  CK(hipMalloc(&pose, pose_h.size() * 4)); CK(hipMalloc(&betas, betas_h.size() * 4)); CK(hipMalloc(&jtd, 72 * 4)); CK(hipMalloc(&jsdd, 720 * 4));
  CK(hipMalloc(&parents, 96)); CK(hipMalloc(&level, 96));
  CK(hipMalloc(&feat_h, (size_t)Bpad * 224 * 4)); CK(hipMalloc(&askin_h, (size_t)Bpad * 384 * 4));
  CK(hipMalloc(&chain, (size_t)B * 72 * 4)); CK(hipMalloc(&ref, (size_t)B * 72 * 4));
  CK(hipMemcpy(pose, pose_h.data(), pose_h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(betas, betas_h.data(), betas_h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(jtd, jt.data(), 288, hipMemcpyHostToDevice)); CK(hipMemcpy(jsdd, jsd.data(), 2880, hipMemcpyHostToDevice));
  CK(hipMemcpy(parents, parents_h, 96, hipMemcpyHostToDevice)); CK(hipMemcpy(level, level_h, 96, hipMemcpyHostToDevice));
  // the infiller's first GEMM on 1024 sequences: 51 200 rows x 96 -> 256, row-major in, fragment-major out, per-row bias
  const int M = 51200, N = 256, K = 96;
  float *X, *rowbias, *Y;
  unsigned short* Ws;
  CK(hipMalloc(&X, (size_t)M * K * 4)); CK(hipMalloc(&rowbias, 50 * 256 * 4)); CK(hipMalloc(&Y, (size_t)M * N * 4)); CK(hipMalloc(&Ws, (size_t)2 * 256 * K * 2));
  { std::vector<float> h((size_t)M * K); for (auto& x : h) x = u(rng); CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  { std::vector<float> h(50 * 256); for (auto& x : h) x = u(rng); CK(hipMemcpy(rowbias, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  { std::vector<_Float16> h((size_t)2 * 256 * K); for (auto& x : h) x = (_Float16)u(rng); CK(hipMemcpy(Ws, h.data(), h.size() * 2, hipMemcpyHostToDevice)); }
  glamr::nn::GemmArgs ga{X, nullptr, nullptr, rowbias, nullptr, Y, M, N, K, K, N, 0, -50, 256, 0};
  ga.Ws = Ws; ga.ws_plane = (size_t)256 * K; ga.x_frag = 0; ga.y_frag = 1;
  const dim3 ggrid((unsigned)((M / 32 + 7) / 8 * 8 * 4));
  // the first attention of the infiller's first window: 1024 sequences x 8 heads, 50 x 50, fragment-major Q | K | V rows
  float *qkv, *att, *mf_out;
  unsigned char* kmask;
  unsigned *n_bad_d, *bad_lanes_d;
  CK(hipMalloc(&qkv, (size_t)(M + 32) * 768 * 4)); CK(hipMalloc(&att, (size_t)(M + 32) * 256 * 4)); CK(hipMalloc(&kmask, M)); CK(hipMemset(kmask, 0, M));
  CK(hipMalloc(&mf_out, (size_t)8192 * 64 * 4)); CK(hipMalloc(&n_bad_d, 8)); CK(hipMalloc(&bad_lanes_d, 256));
  { std::vector<float> h((size_t)(M + 32) * 768); for (auto& x : h) x = u(rng); CK(hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  glamr::PrepArgs pa{B, Bpad, 10, n_levels, 1, pose, 1, betas, jtd, jsdd, parents, level, nullptr, reinterpret_cast<unsigned short*>(feat_h), nullptr,
                     reinterpret_cast<unsigned short*>(askin_h), chain};
  auto launch_victim = [&](hipStream_t st) {
    if (victim == 2) { hipLaunchKernelGGL(pk_selfcheck_kernel, dim3(256 * 5 * 8), dim3(256), 0, st, n_bad_d, bad_lanes_d, 4096); return; }
    if (victim == 0) hipLaunchKernelGGL(glamr::smpl_prep_kernel, dim3(Bpad / 8), dim3(256), 0, st, pa);
    else hipLaunchKernelGGL(chain_only_kernel, dim3((B + 7) / 8), dim3(256), 0, st, B, pose, parents, level, n_levels, chain);
  };
  CK(hipMemset(n_bad_d, 0, 8)); CK(hipMemset(bad_lanes_d, 0, 256));
  launch_victim(sa);
  CK(hipStreamSynchronize(sa));
  if (victim == 2) { unsigned nb = 0; CK(hipMemcpy(&nb, n_bad_d, 4, hipMemcpyDeviceToHost)); std::printf("victim alone: %u packed results differ from the unpacked ones\n", nb); }
  CK(hipMemcpy(ref, chain, (size_t)B * 72 * 4, hipMemcpyDeviceToDevice));
  std::vector<float> ref_h((size_t)B * 72), got_h((size_t)B * 72);
  CK(hipMemcpy(ref_h.data(), ref, ref_h.size() * 4, hipMemcpyDeviceToHost));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int n_bad_runs = 0;
  auto launch_trigger = [&](unsigned long long delay_us) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sb, delay_us * 100ull);
    for (int k = 0; k < 12; ++k) {
      if (trigger == 0) hipLaunchKernelGGL((glamr::nn::gemm_free_kernel<6, 2, 1, 4>), ggrid, dim3(64), 0, sb, ga);
      else if (trigger == 3) hipLaunchKernelGGL(glamr::nn::attention_free_kernel, dim3(1024 * 8), dim3(64), 0, sb, qkv, 768, qkv + 256 * 32, qkv + 512 * 32, 768, kmask, att, 256, 50, 50, 0);
      else if (trigger == 4) hipLaunchKernelGGL(mfma_chain_kernel<true>, dim3(8192), dim3(64), 0, sb, mf_out, 200);
      else if (trigger == 5) hipLaunchKernelGGL(mfma_chain_kernel<false>, dim3(8192), dim3(64), 0, sb, mf_out, 200);
      else hipLaunchKernelGGL(stream_kernel, dim3((unsigned)(((size_t)M * N / 4 + 511) / 512)), dim3(64), 0, sb, reinterpret_cast<const float4*>(Y), reinterpret_cast<float4*>(feat_h), (size_t)M * N / 4);
    }
  };
  hipGraphExec_t gv = nullptr, gt[12] = {};
  if (graphs) {
    hipGraph_t g;
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    launch_victim(sa);
    CK(hipStreamEndCapture(sa, &g)); CK(hipGraphInstantiate(&gv, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
    for (int d = 0; d < 12 && trigger != 1; ++d) {
      CK(hipStreamBeginCapture(sb, hipStreamCaptureModeThreadLocal));
      launch_trigger(30 + 40 * d);
      CK(hipStreamEndCapture(sb, &g)); CK(hipGraphInstantiate(&gt[d], g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
    }
  }
  for (int it = 0; it < iters; ++it) {
    CK(hipMemsetAsync(n_bad_d, 0, 8, sa)); CK(hipMemsetAsync(bad_lanes_d, 0, 256, sa));
    CK(hipMemsetAsync(chain, 0xFF, (size_t)B * 72 * 4, sa));
    CK(hipStreamSynchronize(sa));
    const unsigned long long delay_us = 30 + 40 * (it % 12);
    if (trigger != 1) { if (graphs) CK(hipGraphLaunch(gt[it % 12], sb)); else launch_trigger(delay_us); }
    CK(hipEventRecord(e0, sa));
    if (graphs) CK(hipGraphLaunch(gv, sa)); else launch_victim(sa);
    CK(hipEventRecord(e1, sa));
    CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (victim == 2) {
      unsigned nb[2], lanes[64];
      CK(hipMemcpy(nb, n_bad_d, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(lanes, bad_lanes_d, 256, hipMemcpyDeviceToHost));
      std::printf("run %d (other stream %llu us after the launch): victim %.3f ms, %u packed results differ from the unpacked ones; lanes:", it, delay_us, ms, nb[0]);
      for (int l = 0; l < 64; ++l) if (lanes[l]) std::printf(" %d:%u", l, lanes[l]);
      std::printf("\n");
      n_bad_runs += nb[0] > 0;
      continue;
    }
    CK(hipMemcpy(got_h.data(), chain, got_h.size() * 4, hipMemcpyDeviceToHost));
    int bad_frames = 0;
    for (int f = 0; f < B; ++f) {
      if (std::memcmp(&got_h[(size_t)f * 72], &ref_h[(size_t)f * 72], 288) == 0) continue;
      if (bad_frames < 6) {
        std::printf("   run %d: frame %d (block %d, frame-in-block %d) joints wrong:", it, f, f / 8, f % 8);
        for (int j = 0; j < 24; ++j) if (std::memcmp(&got_h[(size_t)f * 72 + j * 3], &ref_h[(size_t)f * 72 + j * 3], 12) != 0) std::printf(" %d(tid %d, wave %d lane %d)", j, (f % 8) * 24 + j, ((f % 8) * 24 + j) >> 6, ((f % 8) * 24 + j) & 63);
        std::printf("\n");
      }
      ++bad_frames;
    }
    std::printf("run %d (other stream's GEMMs %llu us after the launch): victim %.3f ms, %d of %d frames differ from the kernel alone\n", it, delay_us, ms, bad_frames, B);
    n_bad_runs += bad_frames > 0;
  }
  std::printf("SUMMARY: %d of %d runs beside the other stream differ from the kernel alone\n", n_bad_runs, iters);
  return n_bad_runs ? 1 : 0;
}
