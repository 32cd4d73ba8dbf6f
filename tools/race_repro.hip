// Stand-alone reproducer (no torch, no graphs) of the corruption seen in the two-stream pipeline: smpl_prep_kernel (the library's own, included
// below) runs on stream A while stream B starts a train of the infiller's first GEMM (gemm_free_kernel<6, 2, 1, 4>, also the library's own).
// The chain joints of every run are compared bit for bit with a run of smpl_prep_kernel ALONE on the same inputs.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/race_repro.hip glamr_amd/csrc/api_common.cpp -o tools/_race_repro
//   run:   tools/_race_repro [iterations] [frames] [victim: 0 smpl_prep_kernel, 1 the reduced kernel below] [trigger: 0 gemm_free, 1 none, 2 plain loads] [graphs: 0 / 1]
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
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  glamr::PrepArgs pa{B, Bpad, 10, n_levels, 1, pose, 1, betas, jtd, jsdd, parents, level, nullptr, reinterpret_cast<unsigned short*>(feat_h), nullptr,
                     reinterpret_cast<unsigned short*>(askin_h), chain};
  auto launch_victim = [&](hipStream_t st) {
    if (victim == 0) hipLaunchKernelGGL(glamr::smpl_prep_kernel, dim3(Bpad / 8), dim3(256), 0, st, pa);
    else hipLaunchKernelGGL(chain_only_kernel, dim3((B + 7) / 8), dim3(256), 0, st, B, pose, parents, level, n_levels, chain);
  };
  launch_victim(sa);
  CK(hipStreamSynchronize(sa));
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
