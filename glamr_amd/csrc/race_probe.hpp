// DEBUG BUILDS ONLY (-DGLAMR_RACE_PROBE; tools/race_probe.py): instrumentation that separates the two readings of the two-stream pipeline's
// corruption ("the pipeline race") in one run.  Compiled out of the shipped library.
//   kind 1  a (frame, joint) thread of smpl_prep_kernel re-reads its OWN sG / sJ rows after the level loop and finds something else than it wrote
//           -> LDS contents changed under a live workgroup
//   kind 2  the pose triple a thread read at kernel start differs from what the same address holds at kernel end (cached re-load and
//           system-scope re-load) -> the consumer ran before the producer's stores were visible
//   kind 3  at kernel start a cached load and a system-scope load of the same address disagree -> a stale line in this XCD's L2
// plus, per consumer array (key = the pose pointer): the latest end stamp of the producer's workgroups and the earliest start stamp of the
// consumer's (s_memrealtime, one clock for all XCDs): consumer start < producer end = the launch order itself was violated.
#pragma once
#ifdef GLAMR_RACE_PROBE
#include <hip/hip_runtime.h>

namespace glamr {
namespace probe {

constexpr int TABLE = 16, MAX_REC = 4096;
struct Rec { unsigned kind, block, tid, hw_id, xcc_id, aux; unsigned long long t; float v0, v1, v2, v3; };
struct Slot { unsigned long long key, prod_end_max, cons_start_min_inv, prod_start_min_inv, cons_end_max; };
// level (set by the host): 1 producer / consumer stamps, 2 kind-2 check with the cached re-load, 4 kind-1 check, 8 kind-3 check, 16 kind-2's coherent re-load
struct Buf { unsigned long long n_rec, level; Slot slot[TABLE]; Rec rec[MAX_REC]; };

__device__ Buf* g_buf = nullptr;      // one per translation unit (set by glamr_debug_race_probe in each)

__device__ __forceinline__ unsigned level() { Buf* b = g_buf; return b ? (unsigned)b->level : 0u; }
__device__ __forceinline__ Slot* slot_of(const void* key_) {
  Buf* b = g_buf;
  if (!b || !(b->level & 1)) return nullptr;
  const unsigned long long key = reinterpret_cast<unsigned long long>(key_);
  for (int i = 0; i < TABLE; ++i) {
    const unsigned long long old = atomicCAS(&b->slot[i].key, 0ull, key);
    if (old == 0ull || old == key) return &b->slot[i];
  }
  return nullptr;
}
__device__ __forceinline__ void producer_start(const void* key) { if (Slot* s = slot_of(key)) atomicMax(&s->prod_start_min_inv, ~wall_clock64()); }
__device__ __forceinline__ void producer_end(const void* key) { if (Slot* s = slot_of(key)) atomicMax(&s->prod_end_max, wall_clock64()); }
__device__ __forceinline__ void consumer_start(const void* key) { if (Slot* s = slot_of(key)) atomicMax(&s->cons_start_min_inv, ~wall_clock64()); }
__device__ __forceinline__ void consumer_end(const void* key) { if (Slot* s = slot_of(key)) atomicMax(&s->cons_end_max, wall_clock64()); }

__device__ __forceinline__ void record(unsigned kind, unsigned aux, float v0, float v1, float v2, float v3) {
  Buf* b = g_buf;
  if (!b) return;
  const unsigned long long i = atomicAdd(&b->n_rec, 1ull);
  if (i >= (unsigned long long)MAX_REC) return;
  Rec& r = b->rec[i];
  r.kind = kind; r.block = blockIdx.x; r.tid = threadIdx.x; r.aux = aux;
  r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID: wave, SIMD, CU, SH, SE ...
  r.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
  r.t = wall_clock64();
  r.v0 = v0; r.v1 = v1; r.v2 = v2; r.v3 = v3;
}

// a load the compiler cannot merge with an earlier one, served by the caches like any other load
__device__ __forceinline__ float cached_load(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// a system-scope load: sc0 sc1, re-fetched coherently
__device__ __forceinline__ float coherent_load(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace probe
}  // namespace glamr

#define GLAMR_RACE_PROBE_EXPORT(fn)                                                                                  \
  extern "C" int fn(void* buf) {                                                                                     \
    glamr::probe::Buf* p = static_cast<glamr::probe::Buf*>(buf);                                                      \
    return hipMemcpyToSymbol(HIP_SYMBOL(glamr::probe::g_buf), &p, sizeof(p)) == hipSuccess ? 0 : 1;                  \
  }
#endif
