// TEST INFRASTRUCTURE: instantiates the per-scene optimiser algorithm (glamr_amd/csrc/grecon_algo.hpp) with a single-threaded
// host runtime so the CPU test-suite can check its gradients and control flow against the oracle without a GPU.
// Same argument structs as the C ABI, but every pointer is a HOST pointer.  Never loaded by the product.
#include "../../glamr_amd/csrc/grecon_algo.hpp"
#include <vector>
using namespace glamr::grecon;

struct HostRT {
  int tid() const { return 0; }
  int nthreads() const { return 1; }
  void sync() const {}
  float reduce_sum(float v) const { return v; }
  template <bool LDS = false>
  void scan_multi(float* const* ch, int nch, int n, int stride, bool reverse) const {
    for (int c = 0; c < nch; ++c) scan(ch[c], n, stride, reverse);
  }
  void scan(float* a, int n, int stride, bool reverse) const {
    if (!reverse) { for (int i = 1; i < n; ++i) a[(size_t)i * stride] += a[(size_t)(i - 1) * stride]; }
    else { for (int i = n - 2; i >= 0; --i) a[(size_t)i * stride] += a[(size_t)(i + 1) * stride]; }
  }
};

extern "C" int hostsim_grecon_param_layout(int max_persons, int max_len, glamr_param_layout* out) {
  param_layout(max_persons, max_len, *out);
  return 0;
}

extern "C" int hostsim_grecon_run_stage(const glamr_scene_batch* b, const glamr_stage_desc* st, float* grads_out) {
  glamr_param_layout l;
  param_layout(b->max_persons, b->max_len, l);
  std::vector<float> ws(scene_workspace_floats(b->max_persons, b->max_len));
  HostRT rt;
  for (int si = 0; si < b->n_scenes; ++si) {
    Scene sc;
    assemble_scene(*b, l, st, si, b->n_persons[si], b->seq_len[si], ws.data(), grads_out, sc);
    // the same instance the device launch would pick (single-person / camera-mode specialisations)
    const bool single = b->n_persons[si] == 1;
    switch (camera_mode(*st)) {
      case 1: if (single) run_scene<0, true, 1>(rt, sc, *st, l); else run_scene<0, false, 1>(rt, sc, *st, l); break;
      case 2: if (single) run_scene<0, true, 2>(rt, sc, *st, l); else run_scene<0, false, 2>(rt, sc, *st, l); break;
      default: if (single) run_scene<0, true, 0>(rt, sc, *st, l); else run_scene<0, false, 0>(rt, sc, *st, l); break;
    }
  }
  return 0;
}
