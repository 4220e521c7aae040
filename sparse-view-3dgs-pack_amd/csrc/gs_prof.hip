// gs_prof.hip - implementation of the optional per-stage HIP-event timers (see gs_prof.h).
#include <mutex>
#include <vector>

#include "gs_common.h"
#include "gs_prof.h"

int g_gs_prof_on = 0;
int g_gs_prof_only = -1;  // >= 0: record this stage only (an event pair costs ~10 us of drained pipeline per stage)

namespace {
struct Rec {
  int stage;
  hipEvent_t a, b;
};
std::mutex g_mu;
std::vector<Rec> g_open;           // begun, not yet ended (per stage at most one per thread in practice)
std::vector<Rec> g_done;
std::vector<hipEvent_t> g_pool;
double g_ms[ST_COUNT];
long long g_cnt[ST_COUNT];

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void drain_locked() {
  for (auto& r : g_done) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      g_ms[r.stage] += ms;
      g_cnt[r.stage] += 1;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_done.clear();
}
}  // namespace

void gs_prof_begin(int stage, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r{stage, get_event(), get_event()};
  (void)hipEventRecord(r.a, s);
  g_open.push_back(r);
}
void gs_prof_end(int stage, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = g_open.size(); i-- > 0;)
    if (g_open[i].stage == stage) {
      Rec r = g_open[i];
      g_open.erase(g_open.begin() + i);
      (void)hipEventRecord(r.b, s);
      g_done.push_back(r);
      if (g_done.size() > 4096) drain_locked();
      return;
    }
}

static const char* kNames[ST_COUNT] = {"preprocess_fwd", "scan", "duplicate", "sort_depth", "sort", "tile_ranges", "render_fwd",
                                       "bwd_memset", "render_bwd", "preprocess_bwd", "knn", "l1", "dwt2_l1_fwd",
                                       "dwt2_l1_bwd", "ssim_fwd", "ssim_bwd", "patch_dwt", "elf_map", "dwt_haar", "adam", "model_ops", "preprocess_bwd_step",
                                       "tile_order", "step_uninstanced", "chain"};

extern "C" {
int gs_profile_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_gs_prof_on = on ? 1 : 0;
  return GS_OK;
}
int gs_profile_only(int32_t stage) {
  if (stage >= ST_COUNT) return GS_E_SHAPE;
  std::lock_guard<std::mutex> lk(g_mu);
  g_gs_prof_only = stage < 0 ? -1 : stage;
  return GS_OK;
}
int gs_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  drain_locked();
  for (int i = 0; i < ST_COUNT; i++) {
    g_ms[i] = 0;
    g_cnt[i] = 0;
  }
  return GS_OK;
}
int gs_profile_stage_count(void) { return ST_COUNT; }
const char* gs_profile_stage_name(int32_t i) { return (i >= 0 && i < ST_COUNT) ? kNames[i] : ""; }
/* Waits for all recorded events, then copies accumulated milliseconds / launch counts per stage. */
int gs_profile_read(double* ms, int64_t* counts, int32_t n) {
  if (!ms || !counts) return GS_E_NULL;
  std::lock_guard<std::mutex> lk(g_mu);
  drain_locked();
  for (int i = 0; i < n && i < ST_COUNT; i++) {
    ms[i] = g_ms[i];
    counts[i] = g_cnt[i];
  }
  return GS_OK;
}
}
