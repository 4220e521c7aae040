/*
 * TEST INFRASTRUCTURE — CPU oracle for distCUDA2 (simple-knn).  Never on the product path.
 *
 * Restates /root/reference/fs3dgs_benchmark/gaussian-splatting/submodules/simple-knn/simple_knn.cu
 * step by step (fp32, -ffp-contract=off):
 *   knn()            :186-222   AABB reduce seeded with (0,0,0), Morton codes, stable sort,
 *                               per-1024 box AABBs, per-point pruned search
 *   prepMorton       :46-53     coord2Morton :55-62    boxMinMax :79-118
 *   distBoxPoint     :120-130   updateKBest  :132-146  boxMeanDist :148-184
 * Result semantics: out[i] = (d1+d2+d3)/3 with d1<=d2<=d3 the three smallest squared distances
 * from point i to points with a DIFFERENT INDEX (coincident points count with distance 0).
 * Parity pinning: no reference test exists; pinned by a brute-force float32
 * numpy k-NN in tests/test_oracle_knn.py and the unit-lattice known answer.
 */
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "../include/gsplat.h"
#include "gs_oracle_math.h"

using namespace gso;

#define BOX_SIZE 1024

static inline uint32_t prepMorton(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
static inline uint32_t coord2Morton(V3 c, V3 mn, V3 mx) {
  uint32_t x = prepMorton(f2u_sat(((c.x - mn.x) / (mx.x - mn.x)) * ((1 << 10) - 1)));
  uint32_t y = prepMorton(f2u_sat(((c.y - mn.y) / (mx.y - mn.y)) * ((1 << 10) - 1)));
  uint32_t z = prepMorton(f2u_sat(((c.z - mn.z) / (mx.z - mn.z)) * ((1 << 10) - 1)));
  return x | (y << 1) | (z << 2);
}
struct MinMax {
  V3 minn, maxx;
};
static inline float distBoxPoint(const MinMax& box, V3 p) {
  V3 diff = {0, 0, 0};
  if (p.x < box.minn.x || p.x > box.maxx.x) diff.x = fminf(fabsf(p.x - box.minn.x), fabsf(p.x - box.maxx.x));
  if (p.y < box.minn.y || p.y > box.maxx.y) diff.y = fminf(fabsf(p.y - box.minn.y), fabsf(p.y - box.maxx.y));
  if (p.z < box.minn.z || p.z > box.maxx.z) diff.z = fminf(fabsf(p.z - box.minn.z), fabsf(p.z - box.maxx.z));
  return diff.x * diff.x + diff.y * diff.y + diff.z * diff.z;
}
static inline void updateKBest3(V3 ref, V3 point, float* knn) {
  V3 d = {point.x - ref.x, point.y - ref.y, point.z - ref.z};
  float dist = d.x * d.x + d.y * d.y + d.z * d.z;
  for (int j = 0; j < 3; j++) {
    if (knn[j] > dist) {
      float t = knn[j];
      knn[j] = dist;
      dist = t;
    }
  }
}

/* FSGS/submodules/simple-knn/simple_knn.cu:132-147: the index is carried through the same insertion */
static inline void updateKBest3Idx(V3 ref, V3 point, float* knn, int32_t* ind, int32_t pid) {
  V3 d = {point.x - ref.x, point.y - ref.y, point.z - ref.z};
  float dist = d.x * d.x + d.y * d.y + d.z * d.z;
  for (int j = 0; j < 3; j++) {
    if (knn[j] > dist) {
      float t = knn[j];
      knn[j] = dist;
      dist = t;
      int32_t ti = ind[j];
      ind[j] = pid;
      pid = ti;
    }
  }
}

static int32_t* g_nearest_out = nullptr; /* set by gso_knn_mean_dist2_idx around the shared implementation */

extern "C" {

size_t gso_knn_tmp_bytes(int32_t) { return 128; }

/* Also exposes the Morton order for tests (may be NULL). */
int gso_knn_mean_dist2_ex(const float* xyz, int32_t P, float* out, uint32_t* morton_sorted_idx) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!xyz || !out) return GS_E_NULL;
  const V3* pts = (const V3*)xyz;
  /* cub::DeviceReduce with init {0,0,0}: simple_knn.cu:192-201 */
  V3 mn = {0, 0, 0}, mx = {0, 0, 0};
  for (int i = 0; i < P; i++) {
    mn = {fminf(mn.x, pts[i].x), fminf(mn.y, pts[i].y), fminf(mn.z, pts[i].z)};
    mx = {fmaxf(mx.x, pts[i].x), fmaxf(mx.y, pts[i].y), fmaxf(mx.z, pts[i].z)};
  }
  std::vector<uint32_t> morton(P), idx(P);
  for (int i = 0; i < P; i++) morton[i] = coord2Morton(pts[i], mn, mx);
  std::iota(idx.begin(), idx.end(), 0u);
  /* cub SortPairs = stable */
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return morton[a] < morton[b]; });
  if (morton_sorted_idx) memcpy(morton_sorted_idx, idx.data(), 4 * (size_t)P);
  const uint32_t nb = (P + BOX_SIZE - 1) / BOX_SIZE;
  std::vector<MinMax> boxes(nb);
  for (uint32_t b = 0; b < nb; b++) {
    MinMax me = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    for (uint32_t i = b * BOX_SIZE; i < std::min<uint32_t>(P, (b + 1) * BOX_SIZE); i++) {
      V3 p = pts[idx[i]];
      me.minn = {fminf(me.minn.x, p.x), fminf(me.minn.y, p.y), fminf(me.minn.z, p.z)};
      me.maxx = {fmaxf(me.maxx.x, p.x), fmaxf(me.maxx.y, p.y), fmaxf(me.maxx.z, p.z)};
    }
    boxes[b] = me;
  }
#pragma omp parallel for schedule(dynamic, 256)
  for (int i = 0; i < P; i++) {
    V3 point = pts[idx[i]];
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    int32_t bi[3] = {0, 0, 0}; /* FSGS fork: not cleared between the two passes (simple_knn.cu:158-171) */
    int32_t* nearest = g_nearest_out;
    for (int j = std::max(0, i - 3); j <= std::min(P - 1, i + 3); j++) {
      if (j == i) continue;
      if (nearest) updateKBest3Idx(point, pts[idx[j]], best, bi, (int32_t)idx[j]);
      else updateKBest3(point, pts[idx[j]], best);
    }
    float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    for (uint32_t b = 0; b < nb; b++) {
      float dist = distBoxPoint(boxes[b], point);
      if (dist > reject || dist > best[2]) continue;
      for (int j = b * BOX_SIZE; j < std::min<int>(P, (b + 1) * BOX_SIZE); j++) {
        if (j == i) continue;
        if (nearest) updateKBest3Idx(point, pts[idx[j]], best, bi, (int32_t)idx[j]);
        else updateKBest3(point, pts[idx[j]], best);
      }
    }
    out[idx[i]] = (best[0] + best[1] + best[2]) / 3.0f;
    if (nearest)
      for (int j = 0; j < 3; j++) nearest[3 * (size_t)idx[i] + j] = bi[j];
  }
  return GS_OK;
}

int gso_knn_mean_dist2(const float* xyz, int32_t P, float* out, void*, size_t, void*) {
  return gso_knn_mean_dist2_ex(xyz, P, out, nullptr);
}
int gso_knn_mean_dist2_idx(const float* xyz, int32_t P, float* out, int32_t* nearest, void*, size_t, void*) {
  g_nearest_out = nearest; /* the checker is driven from one thread */
  int rc = gso_knn_mean_dist2_ex(xyz, P, out, nullptr);
  g_nearest_out = nullptr;
  return rc;
}
}
