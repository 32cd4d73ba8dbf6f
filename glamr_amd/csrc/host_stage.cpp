// Host side of the input boundary: scatters the per-detection HybrIK arrays of a batch (pose_est/hybrik_demo/demo.py:317-354: one row per
// DETECTED frame) to their video-frame rows in the pinned staging arrays the device `init_data` reads (glamr_raw_batch).  Plain
// memcpy work -- 1.1 KB per detected frame, 0.34 GB for 1024 sequences of 300 frames -- that the Python host did with numpy block
// copies on one thread (the GIL serialises a thread pool of small copies): here the persons are split over a few threads.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>
#include "common.hpp"

namespace {

struct PersonDesc {      // one row of the int64 table the caller passes (glamr_hip.h)
  int64_t exist, exist_is_f64, n_frames, n_det, rot, betas, trans, K, kp, kp_stride;      // kp_stride: floats per detection row of kp_2d (2 x its keypoints, >= 48)
};

inline bool detected(const PersonDesc& d, int t) {
  return d.exist_is_f64 ? reinterpret_cast<const double*>(d.exist)[t] != 0.0 : reinterpret_cast<const float*>(d.exist)[t] != 0.f;
}

// 0 = ok, 1 = no detection at all, 2 = the number of detected frames differs from the number of rows
int scatter_one(const PersonDesc& d, int k, int T, const glamr_host_staging& h, int32_t* seq_len, int32_t* exist_len) {
  if (!d.exist) { seq_len[k] = 0; return 0; }          // empty person slot
  const int n_fr = (int)d.n_frames, nv = (int)d.n_det;
  float* ex = h.exist + (size_t)k * T;
  int first = -1, last = -1, count = 0;
  for (int t = 0; t < n_fr; ++t) {
    const bool on = detected(d, t);
    ex[t] = d.exist_is_f64 ? (float)reinterpret_cast<const double*>(d.exist)[t] : reinterpret_cast<const float*>(d.exist)[t];
    if (on) { if (first < 0) first = t; last = t; ++count; }
  }
  seq_len[k] = n_fr;
  if (count == 0) return 1;
  if (count != nv) return 2;
  exist_len[k] = last + 1 - first;
  const float* rot = reinterpret_cast<const float*>(d.rot);
  const float* betas = reinterpret_cast<const float*>(d.betas);
  const float* trans = reinterpret_cast<const float*>(d.trans);
  const float* K = reinterpret_cast<const float*>(d.K);
  const float* kp = reinterpret_cast<const float*>(d.kp);
  int det = 0;                                      // detection row of the run's first frame
  for (int t = first; t <= last;) {
    if (!detected(d, t)) { ++t; continue; }
    int e = t;
    while (e <= last && detected(d, e)) ++e;        // run of detections [t, e)
    const size_t n = (size_t)(e - t), row = (size_t)k * T + t;
    std::memcpy(h.rot + row * 216, rot + (size_t)det * 216, n * 216 * sizeof(float));
    std::memcpy(h.betas + row * 10, betas + (size_t)det * 10, n * 10 * sizeof(float));
    std::memcpy(h.trans + row * 3, trans + (size_t)det * 3, n * 3 * sizeof(float));
    std::memcpy(h.K + row * 9, K + (size_t)det * 9, n * 9 * sizeof(float));
    const size_t ks = (size_t)d.kp_stride;
    for (size_t i = 0; i < n; ++i) std::memcpy(h.kp + (row + i) * 48, kp + ((size_t)det + i) * ks, 48 * sizeof(float));      // the first 24 keypoints (29 from HybrIK)
    det += (int)n;
    t = e;
  }
  return 0;
}

}  // namespace

extern "C" int glamr_host_scatter(int n_persons, const int64_t* table, int max_len, const glamr_host_staging* staging, int32_t* seq_len,
                                  int32_t* exist_len, int threads) {
  using namespace glamr;
  GLAMR_REQUIRE(n_persons >= 0 && (n_persons == 0 || (table && staging && seq_len && exist_len)), "null argument");
  GLAMR_REQUIRE(max_len >= 2 && threads >= 1, "max_len must be >= 2 and threads >= 1");
  GLAMR_REQUIRE(n_persons == 0 || (staging->exist && staging->rot && staging->betas && staging->trans && staging->K && staging->kp),
                "a staging array is NULL");
  const PersonDesc* desc = reinterpret_cast<const PersonDesc*>(table);
  for (int k = 0; k < n_persons; ++k) {
    if (!desc[k].exist) continue;                      // empty person slot (a scene with fewer persons than the batch maximum)
    GLAMR_REQUIRE(desc[k].n_frames >= 1 && desc[k].n_frames <= max_len, "person %d: %lld frames do not fit max_len=%d", k, (long long)desc[k].n_frames, max_len);
    GLAMR_REQUIRE(desc[k].rot && desc[k].betas && desc[k].trans && desc[k].K && desc[k].kp, "person %d: a source array is NULL", k);
    GLAMR_REQUIRE(desc[k].kp_stride >= 48, "person %d: kp_2d rows hold %lld values, at least 24 keypoints x 2 are needed", k, (long long)desc[k].kp_stride);
  }
  const int nthr = std::max(1, std::min(threads, n_persons / 32 + 1));
  std::vector<int> status((size_t)std::max(n_persons, 1), 0);
  auto work = [&](int lo, int hi) { for (int k = lo; k < hi; ++k) status[k] = scatter_one(desc[k], k, max_len, *staging, seq_len, exist_len); };
  if (nthr == 1) {
    work(0, n_persons);
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < nthr; ++i) pool.emplace_back(work, (int)((int64_t)n_persons * i / nthr), (int)((int64_t)n_persons * (i + 1) / nthr));
    for (auto& t : pool) t.join();
  }
  for (int k = 0; k < n_persons; ++k) {
    GLAMR_REQUIRE(status[k] != 1, "person %d has no detected frame", k);
    GLAMR_REQUIRE(status[k] != 2, "person %d: bboxes_dict['exist'] marks a different number of frames than the arrays have rows (%lld)", k, (long long)desc[k].n_det);
  }
  return GLAMR_OK;
}
