// gs_prof.h - optional per-stage timing with HIP events recorded on the caller's stream.
// Off by default (zero cost: one relaxed load per stage).  bench.py switches it on for the timed region
// to obtain the average launch duration of each kernel on the stream it is launched on.
#pragma once
#include <hip/hip_runtime.h>

enum GsStage {
  ST_PREPROCESS_FWD = 0,
  ST_SCAN,
  ST_DUPLICATE,
  ST_SORT_DEPTH,
  ST_SORT,
  ST_RANGES,
  ST_RENDER_FWD,
  ST_BWD_MEMSET,
  ST_RENDER_BWD,
  ST_PREPROCESS_BWD,
  ST_KNN,
  ST_L1,
  ST_DWT2_FWD,
  ST_DWT2_BWD,
  ST_SSIM_FWD,
  ST_SSIM_BWD,
  ST_PATCH,
  ST_ELF,
  ST_DWT1,
  ST_ADAM,
  ST_MODEL,
  ST_BWD_STEP,
  ST_TILE_ORDER,
  ST_STEP_UNINST,
  ST_CHAIN,
  ST_COUNT
};

void gs_prof_begin(int stage, hipStream_t s);
void gs_prof_end(int stage, hipStream_t s);
extern int g_gs_prof_on;
extern int g_gs_prof_only;

struct GsProfScope {
  int stage;
  hipStream_t s;
  bool on;
  GsProfScope(int st, hipStream_t str)
      : stage(st), s(str), on(g_gs_prof_on != 0 && (g_gs_prof_only < 0 || g_gs_prof_only == st)) {
    // every kernel group of the library is bracketed by one of these scopes: drop any stale error another
    // runtime user of this thread left behind, so that the launch check that follows reports OUR launches only
    (void)hipGetLastError();
    if (on) gs_prof_begin(stage, s);
  }
  ~GsProfScope() {
    if (on) gs_prof_end(stage, s);
  }
};
#define GS_PROF(stage, stream) GsProfScope _gs_prof_scope_##stage(stage, stream)
