// gs_tilecull.h - exact-safe tile culling of the instance lists (GsView.tile_cull = 1).
//
// The reference bins a Gaussian into every tile of the bounding SQUARE of 3 sigma_max (getRect,
// auxiliary.h:46-57; duplicateWithKeys, rasterizer_impl.cu:70-111) and lets the blend kernels reject, per
// pixel, everything with alpha < 1/255 (forward.cu:352-356).  alpha = min(0.99, o exp(-q/2)) >= 1/255 needs
//     q(u, v) = A u^2 + 2 B u v + C v^2 <= t,   t = 2 ln(255 o)          (u, v) = pixel - mean,
// an ellipse that typically covers ~40 % of the square's tiles (less for low opacity / anisotropic splats).
// Tiles whose pixels all lie outside that ellipse cannot change any output of the rasterizer - colour, depth,
// final_T and every gradient are sums over pairs that pass the alpha test - so they are not emitted at all:
// R shrinks ~2.6x at BASELINE C3 and with it the sort, the tile lists and both blend kernels.
//
// Per tile ROW the ellipse covers one contiguous span of tile columns, computed in closed form: the row's
// pixel band v in [v0, v1]; u_hi(v) = (-B v + sqrt(A t - det v^2)) / A is concave, so its maximum over the band
// is at v clamped to the band from the ellipse's right-most point (u, v) = (ex, -B/C ex); same for u_lo.
// The preprocess kernel counts the spans (tiles_touched), the duplicate kernel re-evaluates the SAME function
// to place the instances; both translation units are built without FP contraction so the two agree (and the
// duplicate kernel stays memory-safe even if they did not, see duplicate_kernel).
// Conservative by construction: t is inflated by 0.1 % + 4e-6 * cond(Q) * t + 1e-3 (cond = A C / det bounds the
// cancellation error of the fp32 power evaluation in the blend kernels), spans get 0.05 px + 1e-5 relative
// slack, and anything degenerate (det <= 0, cond > 1e6, non-finite) falls back to the full rectangle.
// tests/test_gpu_tilecull.py checks the property the argument rests on: every (tile, Gaussian) pair of the
// reference list that is missing here has alpha < 1/255 on all 256 pixels, and the kept pairs are in reference order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct TileCull {
  float x, y, A, B, C, t, det, invA, ex, ey, vr;
  int mode;  // 0 = whole rectangle, 1 = nothing, 2 = ellipse spans
};

__device__ inline TileCull tilecull_setup(int enabled, float x, float y, float A, float B, float C, float o) {
#pragma clang fp contract(off)
  TileCull c;
  c.x = x; c.y = y; c.A = A; c.B = B; c.C = C;
  c.t = c.det = c.invA = c.ex = c.ey = c.vr = 0.f;
  c.mode = 0;
  if (!enabled) return c;
  const float oo = o * 1.0001f;
  if (oo < 1.0f / 255.0f) {  // alpha <= opacity < 1/255 everywhere (power <= 0 is enforced by the blend kernels)
    c.mode = 1;
    return c;
  }
  const float det = A * C - B * B;
  if (!(det > 0.f) || !(A > 0.f) || !(C > 0.f) || !(oo >= 1.0f / 255.0f)) return c;
  const float cond = (A * C) / det;
  if (!(cond < 1.0e6f)) return c;
  float t = 2.0f * logf(255.0f * oo);
  t = t * (1.001f + 4.0e-6f * cond) + 1.0e-3f;
  const float ex = sqrtf(t * C / det), ey = sqrtf(t * A / det);
  if (!(t < 1.0e4f) || !(ex < 1.0e6f) || !(ey < 1.0e6f)) return c;
  c.t = t; c.det = det; c.invA = 1.0f / A; c.ex = ex; c.ey = ey;
  c.vr = -(B / C) * ex;
  c.mode = 2;
  return c;
}

// Tile columns [tx0, tx0 + n) of tile row `ty` that the ellipse can touch, clipped to the reference rectangle
// [rminx, rmaxx).
__device__ inline uint32_t tilecull_row_span(const TileCull& c, uint32_t ty, uint32_t rminx, uint32_t rmaxx,
                                             uint32_t& tx0) {
#pragma clang fp contract(off)
  tx0 = rminx;
  if (c.mode == 0) return rmaxx - rminx;
  if (c.mode == 1) return 0;
  const float v0 = (float)(ty * 16u) - c.y - 0.05f, v1 = v0 + 15.1f;
  if (v0 > c.ey || v1 < -c.ey) return 0;
  const float v0c = fmaxf(v0, -c.ey), v1c = fminf(v1, c.ey);
  const float vh = fminf(fmaxf(c.vr, v0c), v1c), vl = fminf(fmaxf(-c.vr, v0c), v1c);
  const float sh = sqrtf(fmaxf(c.A * c.t - c.det * vh * vh, 0.f));
  const float sl = sqrtf(fmaxf(c.A * c.t - c.det * vl * vl, 0.f));
  const float uh = (-c.B * vh + sh) * c.invA;
  const float ul = (-c.B * vl - sl) * c.invA;
  const float slack = 0.05f + 1.0e-5f * (fabsf(uh) + fabsf(ul) + fabsf(c.x));
  const float xlo = fminf(fmaxf(c.x + ul - slack, -1.0e6f), 1.0e6f);
  const float xhi = fminf(fmaxf(c.x + uh + slack, -1.0e6f), 1.0e6f);
  // tile tx holds pixel columns [16 tx, 16 tx + 15]
  const int a = max((int)ceilf((xlo - 15.0f) * 0.0625f), (int)rminx);
  const int b = min((int)floorf(xhi * 0.0625f) + 1, (int)rmaxx);
  if (b <= a) return 0;
  tx0 = (uint32_t)a;
  return (uint32_t)(b - a);
}

// ---------------------------------------------------------------------------------------------------------------------
// Depth-limited emission (GsScratch.tile_depth_limit).  The blend of a tile stops at the list entry where its last
// pixel saturates (T < 1e-4); everything behind that depth is sorted, stored and never visited - ~80 % of the
// instances at the bench workload.  A caller that renders the same camera again (sparse-view training does, every few
// iterations) hands back, per tile, a depth bound derived from where the previous blend of the tile and of its eight
// neighbours stopped (gs_export_tile_stop_depth; +inf where some pixel of them never saturated).  A (tile, Gaussian)
// pair whose depth exceeds that bound by more than the margin below is not emitted: a row span is cut down to
// [first tile not beyond its bound, last tile not beyond its bound], so spans stay contiguous (tiles in the middle of a
// span are kept even if they fail the test: a harmless superset).
// EXACTNESS is checked by the forward blend itself, not assumed: if a tile with a finite bound has not saturated all of
// its pixels by the end of its list, or saturates them at an entry deeper than the bound, the launch sets
// GeomHeader.trunc_failed and the caller repeats the forward without bounds.  When no tile does: every pair within the
// bound is present, lists are depth-sorted, so each tile's list up to its stopping entry is the reference's and images,
// n_contrib, final_T and all gradients are those of the uncut lists.
// Why the 3 x 3 neighbourhood: the tiles whose last pixel flips between "saturates" and "never does" from one visit to
// the next sit next to tiles that never saturate (silhouettes); with the dilated bound they get none.  Bench workload
// (tests/tools/depth_limit_probe.py, 51 re-visits): 2 views fail with the tile's own stop depth - whatever the margin -
// none with the dilated one, for 18 % instead of 16 % of the instances kept.
// The preprocess kernel (count) and the duplicate kernel (emit) compute the same cut from the same comparisons.
// ---------------------------------------------------------------------------------------------------------------------
#define GS_DEPTH_LIMIT_REL 1.05f
#define GS_DEPTH_LIMIT_ABS 0.02f
// the one definition of "beyond the limit" (count, emit and the blend's check must agree to the bit: no contraction)
__device__ inline bool depth_beyond_limit(float depth, float limit) {
#pragma clang fp contract(off)
  const float scaled = limit * GS_DEPTH_LIMIT_REL;
  const float bound = scaled + GS_DEPTH_LIMIT_ABS;
  return depth > bound;  // limit = +inf: never
}
// Limit buffer (gs_export_tile_stop_depth): T per-tile bounds, then grid_y * ceil(grid_x / 4) SEGMENT bounds - the largest
// bound of each aligned run of four tiles of a row.
__host__ __device__ inline uint32_t depth_limit_segs_x(uint32_t grid_x) { return (grid_x + 3u) / 4u; }
__host__ __device__ inline size_t depth_limit_floats(uint32_t grid_x, uint32_t grid_y) {
  return (size_t)grid_x * grid_y + (size_t)depth_limit_segs_x(grid_x) * grid_y;
}
typedef const __attribute__((address_space(3))) float* GsLdsFloatPtr;  // an LDS pointer the compiler can see is one

// Exact rule: span [tx0, tx0 + n) of tile row ty -> [first tile not beyond its bound, last tile not beyond]; four
// independent loads at a time (selects, not branches, so that they are in flight together).  n is at most 4 wherever this
// is called (Gaussians inside a 4 x 4 tile box).
template <typename Ptr>
__device__ __forceinline__ uint32_t tilecull_trim_span(Ptr limit, uint32_t grid_x, uint32_t ty, float depth, uint32_t& tx0,
                                                       uint32_t n) {
  if (!limit || n == 0) return n;
  const Ptr row = limit + (ty * grid_x + tx0);
  uint32_t first = n, last = 0;
  for (uint32_t b = 0; b < n; b += 4) {
    const uint32_t i1 = min(b + 1, n - 1), i2 = min(b + 2, n - 1), i3 = min(b + 3, n - 1);
    const float v0 = row[b], v1 = row[i1], v2 = row[i2], v3 = row[i3];
    const bool k0 = !depth_beyond_limit(depth, v0), k1 = !depth_beyond_limit(depth, v1);
    const bool k2 = !depth_beyond_limit(depth, v2), k3 = !depth_beyond_limit(depth, v3);
    first = min(first, k0 ? b : (k1 ? i1 : (k2 ? i2 : (k3 ? i3 : n))));
    last = max(last, k3 ? i3 + 1 : (k2 ? i2 + 1 : (k1 ? i1 + 1 : (k0 ? b + 1 : 0u))));
  }
  if (last <= first) return 0;
  tx0 += first;
  return last - first;
}
// Segment rule, for Gaussians that reach beyond a 4 x 4 tile box (a quarter of them at the bench workload, holding two
// thirds of the instances): the span is cut to [first segment not beyond its bound, last such segment] - a superset of
// the exact cut (a tile that is not beyond lies in a segment that is not), at a quarter of the look-ups.  Walking the
// tiles of those spans one by one made every wave wait for its largest Gaussian: +0.08 ms at C3, as much as the shorter
// sort saves - whether the bounds came from L2 or from LDS.
template <typename Ptr>
__device__ __forceinline__ uint32_t tilecull_trim_span_segments(Ptr seg_limit, uint32_t grid_x, uint32_t ty, float depth,
                                                                uint32_t& tx0, uint32_t n) {
  if (n == 0) return 0;
  const uint32_t s0 = tx0 >> 2, ns = ((tx0 + n - 1) >> 2) - s0 + 1;
  const Ptr row = seg_limit + (ty * depth_limit_segs_x(grid_x) + s0);
  uint32_t first = ns, last = 0;
  for (uint32_t b = 0; b < ns; b += 4) {
    const uint32_t i1 = min(b + 1, ns - 1), i2 = min(b + 2, ns - 1), i3 = min(b + 3, ns - 1);
    const float v0 = row[b], v1 = row[i1], v2 = row[i2], v3 = row[i3];
    const bool k0 = !depth_beyond_limit(depth, v0), k1 = !depth_beyond_limit(depth, v1);
    const bool k2 = !depth_beyond_limit(depth, v2), k3 = !depth_beyond_limit(depth, v3);
    first = min(first, k0 ? b : (k1 ? i1 : (k2 ? i2 : (k3 ? i3 : ns))));
    last = max(last, k3 ? i3 + 1 : (k2 ? i2 + 1 : (k1 ? i1 + 1 : (k0 ? b + 1 : 0u))));
  }
  if (last <= first) return 0;
  const uint32_t lo = max(tx0, (s0 + first) << 2), hi = min(tx0 + n, (s0 + last) << 2);
  tx0 = lo;
  return hi - lo;
}
// The same cut for a Gaussian whose spans fit a 4 x 4 tile box [bx0, bx1) x [by0, by1): all (at most 16) bounds are
// fetched at once into a bit mask (bit 4 r + c set = tile (by0 + r, bx0 + c) is beyond), the rows then need no loads.
template <typename Ptr>
__device__ __forceinline__ uint32_t depth_limit_box_mask(Ptr limit, uint32_t grid_x, uint32_t bx0, uint32_t by0, uint32_t bx1,
                                                         uint32_t by1, float depth) {
  // all loads first, then selects (no branches in between): sixteen loads in flight, one L2 round trip
  float b[16];
#pragma unroll
  for (uint32_t r = 0; r < 4; r++) {
    const Ptr row = limit + min(by0 + r, by1 - 1) * grid_x;  // clamped: bits past the box are never read
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) b[4 * r + c] = row[min(bx0 + c, bx1 - 1)];
  }
  uint32_t beyond = 0;
#pragma unroll
  for (uint32_t i = 0; i < 16; i++) beyond |= depth_beyond_limit(depth, b[i]) ? (1u << i) : 0u;
  return beyond;
}
__device__ inline uint32_t tilecull_trim_span_mask(uint32_t beyond, uint32_t bx0, uint32_t by0, uint32_t ty, uint32_t& tx0,
                                                   uint32_t n) {
  if (n == 0) return 0;
  const uint32_t c0 = tx0 - bx0;
  const uint32_t keep = (~beyond >> (4 * (ty - by0))) & (((1u << n) - 1u) << c0) & 0xFu;
  if (!keep) return 0;
  const uint32_t first = (uint32_t)__builtin_ctz(keep), last = 31u - (uint32_t)__builtin_clz(keep);
  tx0 = bx0 + first;
  return last - first + 1;
}
