// gfx950 kernel + C ABI of the fused global optimiser (algorithm: grecon_algo.hpp).
//
// Launch geometry: ONE workgroup per scene (a sequence with its persons and camera), threads over frames.  A scene's whole
// optimisation state is ~0.1-0.2 MB and stays L2-resident; iterations are separated by workgroup barriers only, so a stage of
// 500 iterations is a single launch with no grid-wide synchronisation and no host round trip.  Many scenes run concurrently,
// one per CU (blockIdx -> scene; with >= 256 scenes every CU is busy; XCD placement is irrelevant because scenes never
// communicate).
#include "common.hpp"
#include "grecon_algo.hpp"
#include "block_rt.hpp"
#include <map>
#include <memory>
#include <set>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

namespace glamr {
namespace GLAMR_GRECON_NS {

#ifndef GLAMR_GRECON_MAX_THREADS
#define GLAMR_GRECON_MAX_THREADS 512
#endif
constexpr int MAX_THREADS = GLAMR_GRECON_MAX_THREADS;
static_assert(MAX_THREADS <= 512, "DeviceRT::scan_multi / scan_regs fetch the wave totals of a workgroup as two float4 reads: at most 8 waves (block_rt.hpp)");
#ifndef GLAMR_GRECON_WAVES_PER_EU
#define GLAMR_GRECON_WAVES_PER_EU 2
#endif
#ifndef GLAMR_GRECON_LDS_KB
#ifdef GLAMR_GRECON_WIDE
#define GLAMR_GRECON_LDS_KB 96      // (the description of 32 persons takes ~25 KB of static LDS)
#else
#define GLAMR_GRECON_LDS_KB 153
#endif
#endif

struct KernelArgs {
  glamr_scene_batch b;
  glamr_stage_desc st;
  glamr_param_layout lay;
  float* workspace;
  unsigned long long* stamps;          // {earliest workgroup start, latest workgroup end} of this launch, 100 MHz ticks
  size_t ws_floats_per_scene;
  float* grads_out;
  const float* adam_tab;               // per-iteration Adam scalars formed on the host (workspace header) or null
  int use_lds;
  unsigned fast_floats;
  int layout_len;      // frames the arena / workspace arrays are laid out for (0 = the batch's max_len)
};

template <int FAST, bool SINGLE, int CAM, int TMC = 0>
__global__ __launch_bounds__(MAX_THREADS, GLAMR_GRECON_WAVES_PER_EU) void grecon_stage_kernel(KernelArgs a) {
  __shared__ __attribute__((aligned(16))) float red[RT_RED_FLOATS];
  __shared__ Scene sc;
  __shared__ glamr_stage_desc s_st;        // the scene keeps POINTERS to these: they must live in LDS, not in a thread's private copy
  __shared__ glamr_param_layout s_lay;
  extern __shared__ __attribute__((aligned(16))) float arena[];     // prefix-sum buffers and neighbour-read arrays, when they fit
  const int si = blockIdx.x;
  // The stage is the pipeline's critical path; the priors' co-schedulable kernels of the other stream (nn_free.hpp) share these SIMDs
  // and have slack: this kernel's waves win the issue arbitration against them (user priority 0..3, the others stay at 0)
#ifndef GLAMR_GRECON_PRIO
#define GLAMR_GRECON_PRIO 3
#endif
#if !defined(GLAMR_GRECON_NO_SETPRIO) && GLAMR_GRECON_PRIO > 0
  __builtin_amdgcn_s_setprio(GLAMR_GRECON_PRIO);
#endif
  if (threadIdx.x == 0) {
    atomicMin(a.stamps, (unsigned long long)wall_clock64());
    s_st = a.st;
    s_lay = a.lay;
    assemble_scene(a.b, s_lay, &s_st, si, a.b.n_persons[si], a.b.seq_len[si], a.workspace + (size_t)si * a.ws_floats_per_scene, a.grads_out, sc,
                   a.use_lds ? arena : nullptr, a.fast_floats, a.use_lds, a.layout_len);
    sc.adam_tab = a.adam_tab;
  }
  __syncthreads();
  glamr::DeviceRT rt{red, a.workspace + (size_t)si * a.ws_floats_per_scene};
  run_scene<FAST, SINGLE, CAM, TMC>(rt, sc, a.st, a.lay);
  if (threadIdx.x == 0) atomicMax(a.stamps + 1, (unsigned long long)wall_clock64());
}

}  // namespace GLAMR_GRECON_NS
}  // namespace glamr

using namespace glamr;
using namespace glamr::GLAMR_GRECON_NS;

// Scenes of more than 8 persons (up to GLAMR_GRECON_MAX_PERSONS = 32) run on the instances of csrc/grecon_wide.hip: this file compiled a
// second time with the person arrays of the scene description four times as long (too large for the static LDS the 153 KB arena of the
// instances below leaves), parameters and exchange arrays in the workspace or the lite arena.
constexpr int WIDE_MAX_PERSONS = 32;
#ifdef GLAMR_GRECON_WIDE
#define glamr_grecon_workspace_bytes glamr_grecon_workspace_bytes_wide_
#define glamr_grecon_run_stage glamr_grecon_run_stage_wide_
static_assert(MAXP == WIDE_MAX_PERSONS, "grecon_wide.hip sets GLAMR_MAX_PERSONS");
#else
extern "C" size_t glamr_grecon_workspace_bytes_wide_(int n_scenes, int max_persons, int max_len);
extern "C" int glamr_grecon_run_stage_wide_(const glamr_scene_batch* batch, const glamr_stage_desc* stage, float* grads_out, void* workspace, void* stream_);

extern "C" int glamr_grecon_param_layout(int max_persons, int max_len, glamr_param_layout* out) {
  GLAMR_REQUIRE(out && max_persons >= 1 && max_persons <= WIDE_MAX_PERSONS && max_len >= 2, "bad arguments (1 <= max_persons <= %d, max_len >= 2)", WIDE_MAX_PERSONS);
  param_layout(max_persons, max_len, *out);
  return GLAMR_OK;
}
#endif

// workspace header: the launch's clock stamps
constexpr size_t GLAMR_GRECON_WS_HEADER = 256;

namespace {
#ifndef GLAMR_GRECON_WIDE
// completion event of the last stage launch per workspace: glamr_grecon_last_launch_ns waits for THAT launch only, not for the device
std::mutex g_ws_mu;
// (an EVENT, not the stream handle: the caller may destroy its stream; an event outlives it.  Shared ownership: a thread that waits for a launch
// holds a reference, NOT the table's mutex -- launches on other workspaces, streams and threads go on while it waits)
struct LaunchEvent {
  hipEvent_t ev = nullptr;
  ~LaunchEvent() { if (ev) (void)hipEventDestroy(ev); }
};
std::map<const void*, std::shared_ptr<LaunchEvent>> g_ws_event;
void forget_workspaces_locked() { g_ws_event.clear(); }
std::shared_ptr<LaunchEvent> new_launch_event() {
  auto e = std::make_shared<LaunchEvent>();
  if (hipEventCreateWithFlags(&e->ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); e->ev = nullptr; return nullptr; }
  return e;
}
void record_launch(const void* workspace, hipStream_t stream);
#endif
int current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}
}  // namespace

extern "C" size_t glamr_grecon_workspace_bytes(int n_scenes, int max_persons, int max_len) {
  if (n_scenes <= 0 || max_persons < 1 || max_persons > WIDE_MAX_PERSONS || max_len < 2) return 0;
#ifndef GLAMR_GRECON_WIDE
  if (max_persons > MAXP) return glamr_grecon_workspace_bytes_wide_(n_scenes, max_persons, max_len);
  {      // a stage with a flag only the wide instances know (GLAMR_FLAG_ABSOLUTE_HEADING) runs there with ITS workspace layout: the larger of the two
    const size_t wide = glamr_grecon_workspace_bytes_wide_(n_scenes, max_persons, max_len);
    const size_t own = GLAMR_GRECON_WS_HEADER + (size_t)n_scenes * align_up(scene_workspace_floats(max_persons, layout_frames(max_persons, max_len)), 64) * sizeof(float);
    return wide > own ? wide : own;
  }
#endif
  return GLAMR_GRECON_WS_HEADER + (size_t)n_scenes * align_up(scene_workspace_floats(max_persons, layout_frames(max_persons, max_len)), 64) * sizeof(float);
}

extern "C" int glamr_grecon_run_stage(const glamr_scene_batch* batch, const glamr_stage_desc* stage, float* grads_out,
                                      void* workspace, void* stream_) {
  GLAMR_REQUIRE(batch && stage && workspace, "null argument");
  GLAMR_REQUIRE(batch->n_scenes > 0 && batch->max_persons >= 1 && batch->max_persons <= WIDE_MAX_PERSONS && batch->max_len >= 2,
                "bad batch geometry: n_scenes=%d max_persons=%d (at most %d) max_len=%d", batch->n_scenes, batch->max_persons, WIDE_MAX_PERSONS, batch->max_len);
#ifndef GLAMR_GRECON_WIDE
  if (batch->max_persons > MAXP || (stage && (stage->flags & GLAMR_FLAG_ABSOLUTE_HEADING))) {
    const int rc = glamr_grecon_run_stage_wide_(batch, stage, grads_out, workspace, stream_);
    if (rc == GLAMR_OK) record_launch(workspace, static_cast<hipStream_t>(stream_));
    return rc;
  }
#endif
  GLAMR_REQUIRE(batch->n_joints == NJ, "n_joints must be %d", NJ);
  GLAMR_REQUIRE(batch->max_len <= GLAMR_GRECON_MAX_FRAMES, "max_len=%d exceeds %d frames", batch->max_len, GLAMR_GRECON_MAX_FRAMES);
  GLAMR_REQUIRE(batch->n_persons && batch->seq_len && batch->fr_start && batch->fr_end && batch->vis && batch->j_local && batch->kp_2d &&
                    batch->kp_score && batch->cam_K && batch->traj_local_pred && batch->orient_cam && batch->base_orient &&
                    batch->base_trans && batch->person2cam && batch->cam_pose && batch->params && batch->losses && batch->orient_world &&
                    batch->trans_world && batch->kp_2d_pred && batch->orient_cam_in_world,
                "a required array of glamr_scene_batch is NULL");
  GLAMR_REQUIRE(batch->max_persons == 1 || batch->rel_transform_cam || !((stage->loss_mask >> GLAMR_LOSS_REL_TRANSFORM) & 1u),
                "rel_transform loss needs rel_transform_cam for multi-person scenes");
  GLAMR_REQUIRE(stage->niters >= 0, "niters must be >= 0");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  KernelArgs ka;
  ka.b = *batch;
  ka.st = *stage;
  param_layout(batch->max_persons, batch->max_len, ka.lay);
  // workspace header: the launch's own clock stamps (read back with glamr_grecon_last_launch_ns)
  ka.stamps = static_cast<unsigned long long*>(workspace);
  GLAMR_HIP_CHECK(hipMemsetAsync(workspace, 0xFF, 8, stream));
  GLAMR_HIP_CHECK(hipMemsetAsync(static_cast<char*>(workspace) + 8, 0, 8, stream));
  ka.workspace = reinterpret_cast<float*>(static_cast<char*>(workspace) + GLAMR_GRECON_WS_HEADER);
  // Adam's step size and second-moment correction per iteration, in the reference's own arithmetic (Python doubles, libm pow): a
  // running product on the device would differ in the last bit of the fp32 scalar now and then, and the update must not (rotmath.hpp).
  // One table per (device, lr, niters), uploaded the first time a stage with that schedule runs and kept for the life of the process:
  // no per-launch host->device copy (a pageable copy on the launch stream would stall a pipelined host).
  ka.adam_tab = nullptr;
  if (stage->niters > 0 && stage->niters <= ADAM_TAB_MAX) {
    static std::mutex mu;
    static std::map<std::tuple<int, double, int>, float*> tables;
    int devid = 0;
    GLAMR_HIP_CHECK(hipGetDevice(&devid));
    std::lock_guard<std::mutex> lock(mu);
    float*& dtab = tables[std::make_tuple(devid, stage->lr, (int)stage->niters)];
    if (!dtab) {
      std::vector<float> tab(2 * (size_t)stage->niters);
      for (int i = 0; i < stage->niters; ++i) adam_coef_host(stage->lr, i + 1, &tab[2 * (size_t)i]);
      GLAMR_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dtab), tab.size() * sizeof(float)));
      GLAMR_HIP_CHECK(hipMemcpy(dtab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    ka.adam_tab = dtab;
  }
  ka.ws_floats_per_scene = align_up(scene_workspace_floats(batch->max_persons, layout_frames(batch->max_persons, batch->max_len)), 64);
  ka.grads_out = grads_out;
  int threads = (batch->max_len + 63) / 64 * 64;
  if (threads > MAX_THREADS) threads = MAX_THREADS;
  if (const char* e = std::getenv("GLAMR_GRECON_THREADS_RT")) {      // development aid: fewer threads than frames = several passes per thread
    const int cap = std::atoi(e);
    if (cap >= 64 && cap < threads) threads = cap / 64 * 64;
  }
  // on-chip arena: prefix-sum / neighbour-exchange arrays first, then as much of the compact keypoint table as fits
  constexpr size_t LDS_MAX = GLAMR_GRECON_LDS_KB * 1024;
  size_t LDS_BUDGET = LDS_MAX;
  // Occupancy: a workgroup's waves hold 256 registers each, so a CU (4 SIMDs x 512 registers) takes 8 / waves workgroups -- if their arenas fit its
  // 160 KB together.  When the batch has more scenes than the chip has CUs, the arena is capped at that share (the keypoint table
  // overflows into the workspace): 1024 scenes of 256 frames 40.1 -> 27.5 ms per 500 iterations.  A 300-frame scene has 5 waves: one per CU.
  const int devid_launch = current_device();
  const int n_cus = [devid_launch] {      // per device: a process may drive more than one GPU
    static std::mutex cmu;
    static std::map<int, int> cus;
    std::lock_guard<std::mutex> lock(cmu);
    int& n = cus[devid_launch];
    if (n <= 0 && (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, devid_launch) != hipSuccess || n <= 0)) n = 256;
    return n;
  }();
  const int wgs_per_cu = 4 * GLAMR_GRECON_WAVES_PER_EU / (threads / 64);
  // (the exchange arrays alone -- arena mode 3: parameters and Adam moments then stay in the workspace; keeping them on chip as well,
  // mode 1, needs 113 instead of 53 floats per frame and is what a scene that has its CU to itself gets)
  const size_t full_arena = scene_fast_floats(batch->max_persons, batch->max_len, 3) * sizeof(float);
  if (wgs_per_cu > 1 && batch->n_scenes > n_cus) {
    // static LDS per workgroup: the scene description (8 persons ~4 KB, 32 persons ~25 KB in the wide build), the scan scratch, the
    // stage descriptor and the layout -- taken from the types, so the share is right for both builds
    const size_t static_lds = sizeof(Scene) + RT_RED_FLOATS * sizeof(float) + sizeof(glamr_stage_desc) + sizeof(glamr_param_layout) + 256;
    const size_t share = (size_t)160 * 1024 / wgs_per_cu - static_lds;
    if (full_arena <= share && batch->max_len <= threads) LDS_BUDGET = share;
  }
  if (const char* e = std::getenv("GLAMR_GRECON_LDS_KB_RT")) {      // development aid (tools/overlap_probe.py)
    const size_t v = (size_t)std::atoi(e) * 1024;
    LDS_BUDGET = v < 16384 ? (size_t)16384 : (v > LDS_MAX ? LDS_MAX : v);
  }
  // constant-layout instance (every array address of the loop a compile-time constant): one person, 257..304 frames, a thread per
  // frame, the full arena of the 304-frame layout fits, and the caller does not ask for the gradient record
  const int lay_len = layout_frames(batch->max_persons, batch->max_len);
  const bool no_const_layout = std::getenv("GLAMR_GRECON_NO_CONST_LAYOUT") != nullptr;      // development aid / A-B tests (read per launch)
  const bool const_layout = lay_len == GLAMR_CONST_LAYOUT_FRAMES && !grads_out && !no_const_layout && batch->max_len <= threads && !(batch->loss_history && stage->niters > 0) &&
                            scene_fast_floats(1, lay_len, 1) * sizeof(float) <= LDS_BUDGET;
  ka.layout_len = const_layout ? lay_len : 0;
  const int arena_len = const_layout ? lay_len : batch->max_len;
  const size_t full = scene_fast_floats(batch->max_persons, arena_len, 1) * sizeof(float);
  const size_t lite = scene_fast_floats(batch->max_persons, batch->max_len, 2) * sizeof(float);
  // 1 full arena + Adam state of single-person scenes, 3 full arena (both: single-pass instances, a thread per frame), 2 lite arena,
  // 0 everything in the workspace
  const bool per_frame = batch->max_len <= threads;
  // (4 = mid, scenes of several persons only: BASELINE configs[3]'s 4 x 300 frames take 143 of the 153 KB; `GLAMR_GRECON_NO_MID_ARENA` keeps them on the lite arena)
  const size_t mid = scene_fast_floats(batch->max_persons, batch->max_len, 4) * sizeof(float);
  const bool no_mid = std::getenv("GLAMR_GRECON_NO_MID_ARENA") != nullptr;      // development aid / A-B tests (read per launch)
  ka.use_lds = (full <= LDS_BUDGET && per_frame) ? 1 : ((full_arena <= LDS_BUDGET && per_frame) ? 3 : ((mid <= LDS_BUDGET && batch->max_persons > 1 && !no_mid) ? 4 : (lite <= LDS_BUDGET ? 2 : 0)));
  // a launch that records the per-iteration loss history (glamr_scene_batch.loss_history) runs on the plain instance: the reporting evaluation
  // every iteration is a diagnostic mode, and the arena instances' loops stay free of it
  if (batch->loss_history && stage->niters > 0) ka.use_lds = 0;
  const size_t base = ka.use_lds == 1 ? full : (ka.use_lds == 3 ? full_arena : (ka.use_lds == 4 ? mid : lite));
  const size_t want = base + (size_t)NJ * 6 * batch->max_persons * arena_len * sizeof(float);
  const size_t dyn = ka.use_lds ? (want < LDS_BUDGET ? want : LDS_BUDGET) : 0;
  ka.fast_floats = (unsigned)(dyn / sizeof(float));
  // SINGLE needs every scene of the batch to hold exactly one person: max_persons == 1 guarantees it
  [[maybe_unused]] const bool single = batch->max_persons == 1;
  auto launch = [&](auto kern, size_t lds) -> int {
    if (lds) {      // once per (device, instance) and process (a driver call per launch is host time on every step)
      static std::mutex amu;
      static std::set<std::pair<int, const void*>> raised;
      std::lock_guard<std::mutex> lock(amu);
      if (raised.insert(std::make_pair(devid_launch, reinterpret_cast<const void*>(kern))).second)
        GLAMR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
    }
    hipLaunchKernelGGL(kern, dim3(batch->n_scenes), dim3(threads), lds, stream, ka);
    return GLAMR_OK;
  };
  int rc;
#ifdef GLAMR_GRECON_WIDE
  if (ka.use_lds == 1 || ka.use_lds == 3 || ka.use_lds == 4) {      // (the full arena of > 8 persons never fits; kept from being instantiated -- the mid arena likewise)
    ka.use_lds = lite <= LDS_BUDGET ? 2 : 0;
    const size_t want2 = lite + (size_t)NJ * 6 * batch->max_persons * arena_len * sizeof(float);
    ka.fast_floats = (unsigned)((ka.use_lds ? (want2 < LDS_BUDGET ? want2 : LDS_BUDGET) : 0) / sizeof(float));
  }
  const size_t dynw = (size_t)ka.fast_floats * sizeof(float);
  rc = ka.use_lds == 2 ? launch(grecon_stage_kernel<2, false, 0>, dynw) : launch(grecon_stage_kernel<0, false, 0>, 0);
#else
  const int cam = camera_mode(*stage);      // per-camera-mode instances: 24.1 vs 26.1 us per iteration (1 person), 61.8 vs 66.2 (2 persons, main stage)
  constexpr int CL = GLAMR_CONST_LAYOUT_FRAMES;
  if (ka.use_lds == 1 && single && const_layout)
    rc = cam == 1 ? launch(grecon_stage_kernel<1, true, 1, CL>, dyn) : cam == 2 ? launch(grecon_stage_kernel<1, true, 2, CL>, dyn) : launch(grecon_stage_kernel<1, true, 0, CL>, dyn);
  else if (ka.use_lds == 1 && single)
    rc = cam == 1 ? launch(grecon_stage_kernel<1, true, 1>, dyn) : cam == 2 ? launch(grecon_stage_kernel<1, true, 2>, dyn) : launch(grecon_stage_kernel<1, true, 0>, dyn);
  else if (ka.use_lds == 3 && single)
    rc = cam == 1 ? launch(grecon_stage_kernel<3, true, 1>, dyn) : cam == 2 ? launch(grecon_stage_kernel<3, true, 2>, dyn) : launch(grecon_stage_kernel<3, true, 0>, dyn);
  else if (ka.use_lds == 1 || ka.use_lds == 3)      // several persons: the two full-arena modes are the same layout
    rc = cam == 1 ? launch(grecon_stage_kernel<1, false, 1>, dyn) : cam == 2 ? launch(grecon_stage_kernel<1, false, 2>, dyn) : cam == 3 ? launch(grecon_stage_kernel<1, false, 3>, dyn) : launch(grecon_stage_kernel<1, false, 0>, dyn);
  else if (ka.use_lds == 2 && single)
    rc = cam == 1 ? launch(grecon_stage_kernel<2, true, 1>, dyn) : cam == 2 ? launch(grecon_stage_kernel<2, true, 2>, dyn) : launch(grecon_stage_kernel<2, true, 0>, dyn);
  else if (ka.use_lds == 4)
    rc = cam == 1 ? launch(grecon_stage_kernel<4, false, 1>, dyn) : cam == 2 ? launch(grecon_stage_kernel<4, false, 2>, dyn) : cam == 3 ? launch(grecon_stage_kernel<4, false, 3>, dyn) : launch(grecon_stage_kernel<4, false, 0>, dyn);
  else if (ka.use_lds == 2)
    rc = cam == 1 ? launch(grecon_stage_kernel<2, false, 1>, dyn) : cam == 2 ? launch(grecon_stage_kernel<2, false, 2>, dyn) : cam == 3 ? launch(grecon_stage_kernel<2, false, 3>, dyn) : launch(grecon_stage_kernel<2, false, 0>, dyn);
  else
    rc = launch(grecon_stage_kernel<0, false, 0>, 0);
#endif
  if (rc) return rc;
  GLAMR_HIP_CHECK(hipGetLastError());
#ifndef GLAMR_GRECON_WIDE
  record_launch(workspace, stream);
#endif
  return GLAMR_OK;
}

#ifndef GLAMR_GRECON_WIDE
namespace {
void record_launch(const void* workspace, hipStream_t stream) {
  {
    // completion event of this launch, per workspace (not under stream capture: a captured launch has no completion of its own, and
    // glamr_grecon_last_launch_ns then falls back to a device-wide wait)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    std::lock_guard<std::mutex> lock(g_ws_mu);
    if (cap == hipStreamCaptureStatusNone) {
      if (g_ws_event.size() > 4096) forget_workspaces_locked();      // callers that never ask for the stamps
      auto it = g_ws_event.find(workspace);
      // an event somebody is waiting on keeps standing for the launch it was recorded after: this launch gets a new one
      if (it != g_ws_event.end() && it->second.use_count() > 1) { g_ws_event.erase(it); it = g_ws_event.end(); }
      if (it == g_ws_event.end()) {
        if (auto e = new_launch_event()) it = g_ws_event.emplace(workspace, std::move(e)).first;
      }
      if (it != g_ws_event.end() && hipEventRecord(it->second->ev, stream) != hipSuccess) { (void)hipGetLastError(); g_ws_event.erase(it); }
    } else {
      g_ws_event.erase(workspace);
    }
  }
}
}  // namespace

// torch.optim.Adam on a flat parameter vector, with the optimiser's own update function: the entry point the parity tests use to
// check the device arithmetic bit for bit against torch.optim.Adam (and a plain fused Adam for callers that keep their own loop)
__global__ void adam_step_kernel(int n, float* p, float* m, float* v, const float* g, glamr::grecon::AdamCoef c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) glamr::grecon::adam(p[i], m[i], v[i], g[i], c);
}
extern "C" int glamr_adam_step(int n, float* params, float* exp_avg, float* exp_avg_sq, const float* grad, double lr, int step, void* stream_) {
  GLAMR_REQUIRE(n >= 0 && (n == 0 || (params && exp_avg && exp_avg_sq && grad)), "null argument");
  GLAMR_REQUIRE(step >= 1 && lr > 0.0, "step must be >= 1 (1-based, as torch counts) and lr > 0");
  if (n == 0) return GLAMR_OK;
  float tab[2];
  adam_coef_host(lr, step, tab);
  AdamCoef c{tab[0], tab[1], 0.0f};
  c.inv_bc2_sqrt = 1.0f / tab[1];            // (host: IEEE division)
  hipLaunchKernelGGL(adam_step_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), n, params, exp_avg, exp_avg_sq, grad, c);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

// The same step with its number ON THE DEVICE: loops that are replayed as HIP graphs (the latent-optimisation mode's iteration) cannot carry
// a per-iteration scalar in a kernel argument.  `coef` holds two floats per step, made on the host (Python's arithmetic: libm pow in double).
__global__ void adam_step_indexed_kernel(int n, float* p, float* m, float* v, const float* g, const float* coef, const int32_t* step_index) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = *step_index;
  glamr::grecon::AdamCoef c{coef[2 * k], coef[2 * k + 1], 0.0f};
  c.finish();
  if (i < n) glamr::grecon::adam(p[i], m[i], v[i], g[i], c);
}
__global__ void counter_add_kernel(int32_t* c, int value) { *c += value; }

extern "C" int glamr_adam_coef_table(double lr, int n_steps, float* out_host) {
  GLAMR_REQUIRE(out_host && n_steps > 0 && lr > 0.0, "bad argument");
  for (int k = 0; k < n_steps; ++k) adam_coef_host(lr, k + 1, out_host + 2 * k);
  return GLAMR_OK;
}
extern "C" int glamr_adam_step_indexed(int n, float* params, float* exp_avg, float* exp_avg_sq, const float* grad, const float* coef_table,
                                       const int32_t* step_index, void* stream_) {
  GLAMR_REQUIRE(n >= 0 && (n == 0 || (params && exp_avg && exp_avg_sq && grad)) && coef_table && step_index, "null argument");
  if (n == 0) return GLAMR_OK;
  hipLaunchKernelGGL(adam_step_indexed_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), n, params, exp_avg, exp_avg_sq, grad,
                     coef_table, step_index);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}
extern "C" int glamr_counter_add(int32_t* counter, int value, void* stream_) {
  GLAMR_REQUIRE(counter, "null argument");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_), counter, value);
  GLAMR_HIP_CHECK(hipGetLastError());
  return GLAMR_OK;
}

#ifdef GLAMR_PHASE_TIMING
// development builds only: per-phase time of workgroup 0 in the last stage launch, in 10 ns ticks
extern "C" int glamr_debug_phase_ticks(unsigned long long* out16) {
  GLAMR_HIP_CHECK(hipDeviceSynchronize());
  GLAMR_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(glamr::g_phase_ticks), 16 * sizeof(unsigned long long)));
  return GLAMR_OK;
}
#endif

extern "C" int glamr_grecon_last_launch_ns(const void* workspace, double* ns) {
  GLAMR_REQUIRE(workspace && ns, "null argument");
  unsigned long long st[2];
  // the stamps are read on the stream the launch ran on: this waits for the work of THAT stream, not for the device (a pipelined caller
  // keeps its other streams running).  A workspace this library has not seen a launch on falls back to a device-wide wait.
  // The launch's own completion event: waits for that launch, not for the device.  The table's mutex only covers the look-up; the wait holds a
  // reference to the event, so another thread's launch on the same workspace (record_launch) or the table's clean-up cannot destroy it under this
  // thread, and launches on any workspace proceed while it waits.
  bool waited = false;
  {
    std::shared_ptr<LaunchEvent> e;
    {
      std::lock_guard<std::mutex> lock(g_ws_mu);
      auto it = g_ws_event.find(workspace);
      if (it != g_ws_event.end()) e = it->second;
    }
    if (e) waited = hipEventSynchronize(e->ev) == hipSuccess;
  }
  if (!waited) {
    (void)hipGetLastError();
    GLAMR_HIP_CHECK(hipDeviceSynchronize());
  }
  // the 16 bytes come back on a stream of their own (non-blocking: a plain hipMemcpy goes through the null stream and would wait for every
  // blocking stream of the process)
  {
    static std::mutex cmu;
    static std::map<int, hipStream_t> copy_streams;
    std::lock_guard<std::mutex> lock(cmu);
    hipStream_t& cs = copy_streams[current_device()];
    if (!cs) GLAMR_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    GLAMR_HIP_CHECK(hipMemcpyAsync(st, workspace, sizeof(st), hipMemcpyDeviceToHost, cs));
    GLAMR_HIP_CHECK(hipStreamSynchronize(cs));
  }
  GLAMR_REQUIRE(st[0] != ~0ull, "no stage launch has completed on this workspace");
  *ns = st[1] > st[0] ? (double)(st[1] - st[0]) * 10.0 : 0.0;
  return GLAMR_OK;
}
#endif  // !GLAMR_GRECON_WIDE
