// TEST INFRASTRUCTURE: instantiates the per-scene optimiser algorithm (glamr_amd/csrc/grecon_algo.hpp) with a single-threaded
// host runtime so the CPU test-suite can check its gradients and control flow against the oracle without a GPU.
// Same argument structs as the C ABI, but every pointer is a HOST pointer.  Never loaded by the product.
#include "../../glamr_amd/csrc/grecon_algo.hpp"
#include <vector>
using namespace glamr::grecon;

struct HostRT {
  static constexpr bool one_thread_per_frame = false;
  float* arena_ = nullptr; float* ws_ = nullptr;      // only the constant-layout instances ask for them
  float* arena() const { return arena_; }
  float* workspace() const { return ws_; }
  int tid() const { return 0; }
  int nthreads() const { return 1; }
  void sync() const {}
  float reduce_sum(float v) const { return v; }
  template <int N> void reduce_sum_n(float (&)[N]) const {}
  template <bool LDS = false>
  void scan_multi(float* const* ch, int nch, int n, int stride, bool reverse, bool /*shuffle_order*/ = false) const {
    for (int c = 0; c < nch; ++c) scan(ch[c], n, stride, reverse);
  }
  void scan(float* a, int n, int stride, bool reverse) const {
    if (!reverse) { for (int i = 1; i < n; ++i) a[(size_t)i * stride] += a[(size_t)(i - 1) * stride]; }
    else { for (int i = n - 2; i >= 0; --i) a[(size_t)i * stride] += a[(size_t)(i + 1) * stride]; }
  }
};

// HostRT + a per-iteration record of the scene's parameter vector and gradient (scene 0): the trajectory tests/tools compare with
// the reference's Adam trajectory iteration by iteration
struct TraceRT : HostRT {
  float* params_out = nullptr;   // [niters][scene_stride]
  float* grads_out = nullptr;    // [niters][scene_stride] (needs store_grad) or null
  int stride = 0;
  void trace(int it, Scene& sc) {
    if (params_out) for (int i = 0; i < stride; ++i) params_out[(size_t)it * stride + i] = sc.cp[i];
    if (grads_out && sc.store_grad) for (int i = 0; i < stride; ++i) grads_out[(size_t)it * stride + i] = sc.cg[i];
  }
};

// absolute_heading is only compiled into the instances of csrc/grecon_wide.hip: this runtime would silently accumulate the headings -- every
// entry point that takes a stage descriptor refuses it (return code 2; the callers assert 0)
static bool unsupported(const glamr_stage_desc* st) { return (st->flags & GLAMR_FLAG_ABSOLUTE_HEADING) != 0; }

template <class RT>
static void run_one(RT& rt, const glamr_scene_batch* b, const glamr_stage_desc* st, const glamr_param_layout& l, Scene& sc, int si) {
  std::vector<float> tab(2 * (size_t)(st->niters > 0 ? st->niters : 1));
  for (int i = 0; i * 2 < (int)tab.size(); ++i) adam_coef_host(st->lr, i + 1, &tab[2 * i]);
  sc.adam_tab = st->niters <= ADAM_TAB_MAX ? tab.data() : nullptr;
  // the same instance the device launch would pick (single-person / camera-mode specialisations)
  const bool single = b->n_persons[si] == 1;
  switch (camera_mode(*st)) {
    case 1: if (single) run_scene<0, true, 1>(rt, sc, *st, l); else run_scene<0, false, 1>(rt, sc, *st, l); break;
    case 2: if (single) run_scene<0, true, 2>(rt, sc, *st, l); else run_scene<0, false, 2>(rt, sc, *st, l); break;
    case 3: if (single) run_scene<0, true, 0>(rt, sc, *st, l); else run_scene<0, false, 3>(rt, sc, *st, l); break;      // (constant camera: its own instance for several persons, as on the device)
    default: if (single) run_scene<0, true, 0>(rt, sc, *st, l); else run_scene<0, false, 0>(rt, sc, *st, l); break;
  }
}

// scene 0 only, with the trajectory recorded
extern "C" int hostsim_grecon_trace_stage(const glamr_scene_batch* b, const glamr_stage_desc* st, float* grads_scratch, float* params_trace, float* grads_trace) {
  if (unsupported(st)) return 2;
  glamr_param_layout l;
  param_layout(b->max_persons, b->max_len, l);
  std::vector<float> ws(scene_workspace_floats(b->max_persons, b->max_len));
  TraceRT rt;
  rt.params_out = params_trace; rt.grads_out = grads_trace; rt.stride = l.scene_stride;
  Scene sc;
  assemble_scene(*b, l, st, 0, b->n_persons[0], b->seq_len[0], ws.data(), grads_scratch, sc);
  run_one(rt, b, st, l, sc, 0);
  return 0;
}

extern "C" int hostsim_grecon_param_layout(int max_persons, int max_len, glamr_param_layout* out) {
  param_layout(max_persons, max_len, *out);
  return 0;
}

extern "C" int hostsim_grecon_run_stage(const glamr_scene_batch* b, const glamr_stage_desc* st, float* grads_out) {
  if (unsupported(st)) return 2;
  glamr_param_layout l;
  param_layout(b->max_persons, b->max_len, l);
  std::vector<float> ws(scene_workspace_floats(b->max_persons, b->max_len));
  HostRT rt;
  for (int si = 0; si < b->n_scenes; ++si) {
    Scene sc;
    assemble_scene(*b, l, st, si, b->n_persons[si], b->seq_len[si], ws.data(), grads_out, sc);
    run_one(rt, b, st, l, sc, si);
  }
  return 0;
}

// The full-arena single-person instances (parameters + Adam moments in the arena) with the arena in host memory; layout_len > 0 = the
// constant-layout instance (arena / workspace laid out for GLAMR_CONST_LAYOUT_FRAMES frames, on-chip parameter blocks in that layout).
// Must give the results of hostsim_grecon_run_stage to the bit: same arithmetic, different addresses.
extern "C" int hostsim_grecon_run_stage_arena(const glamr_scene_batch* b, const glamr_stage_desc* st, int const_layout) {
  if (unsupported(st)) return 2;
  if (b->max_persons != 1) return -1;
  constexpr int CL = GLAMR_CONST_LAYOUT_FRAMES;
  if (const_layout && layout_frames(1, b->max_len) != CL) return -2;
  const int lay_len = const_layout ? CL : b->max_len;
  glamr_param_layout l;
  param_layout(1, b->max_len, l);
  std::vector<float> ws(scene_workspace_floats(1, lay_len));
  const size_t fast_floats = scene_fast_floats(1, lay_len, 1) + (size_t)2 * 6 * lay_len;      // room for two keypoint rows "on chip"
  std::vector<float> arena(fast_floats);
  std::vector<float> tab(2 * (size_t)(st->niters > 0 ? st->niters : 1));
  for (int i = 0; i < st->niters; ++i) adam_coef_host(st->lr, i + 1, &tab[2 * (size_t)i]);
  HostRT rt;
  rt.arena_ = arena.data(); rt.ws_ = ws.data();
  for (int si = 0; si < b->n_scenes; ++si) {
    Scene sc;
    assemble_scene(*b, l, st, si, 1, b->seq_len[si], ws.data(), nullptr, sc, arena.data(), fast_floats, 1, const_layout ? CL : 0);
    sc.adam_tab = st->niters > 0 ? tab.data() : nullptr;
    const int cam = camera_mode(*st);
    if (const_layout) {
      if (cam == 1) run_scene<1, true, 1, CL>(rt, sc, *st, l); else if (cam == 2) run_scene<1, true, 2, CL>(rt, sc, *st, l); else run_scene<1, true, 0, CL>(rt, sc, *st, l);
    } else {
      if (cam == 1) run_scene<1, true, 1>(rt, sc, *st, l); else if (cam == 2) run_scene<1, true, 2>(rt, sc, *st, l); else run_scene<1, true, 0>(rt, sc, *st, l);
    }
  }
  return 0;
}

// The full-arena single-person instance (parameters + Adam moments in the "on-chip" arena) WITH the gradient record: what a stage driven
// launch by launch from outside runs on the device for one-person scenes (latent-optimisation mode, GLAMR_FLAG_KEEP_CAM_PARAMS).
extern "C" int hostsim_grecon_run_stage_arena_grads(const glamr_scene_batch* b, const glamr_stage_desc* st, float* grads_out) {
  if (unsupported(st)) return 2;
  if (b->max_persons != 1) return -1;
  glamr_param_layout l;
  param_layout(1, b->max_len, l);
  std::vector<float> ws(scene_workspace_floats(1, b->max_len));
  const size_t fast_floats = scene_fast_floats(1, b->max_len, 1) + (size_t)2 * 6 * b->max_len;
  std::vector<float> arena(fast_floats);
  std::vector<float> tab(2 * (size_t)(st->niters > 0 ? st->niters : 1));
  for (int i = 0; i < st->niters; ++i) adam_coef_host(st->lr, i + 1, &tab[2 * (size_t)i]);
  HostRT rt;
  rt.arena_ = arena.data(); rt.ws_ = ws.data();
  for (int si = 0; si < b->n_scenes; ++si) {
    Scene sc;
    assemble_scene(*b, l, st, si, 1, b->seq_len[si], ws.data(), grads_out, sc, arena.data(), fast_floats, 1, 0);
    sc.adam_tab = st->niters > 0 ? tab.data() : nullptr;
    const int cam = camera_mode(*st);
    if (cam == 1) run_scene<1, true, 1>(rt, sc, *st, l); else if (cam == 2) run_scene<1, true, 2>(rt, sc, *st, l); else run_scene<1, true, 0>(rt, sc, *st, l);
  }
  return 0;
}

// the optimiser's Adam update on a flat vector (compared bit for bit with torch.optim.Adam in tests/test_adam_exact.py)
extern "C" void hostsim_adam_step(int n, float* p, float* m, float* v, const float* g, double lr, int step) {
  float tab[2];
  adam_coef_host(lr, step, tab);
  AdamCoef c{tab[0], tab[1], 0.0f};
  c.finish();
  for (int i = 0; i < n; ++i) adam(p[i], m[i], v[i], g[i], c);
}
