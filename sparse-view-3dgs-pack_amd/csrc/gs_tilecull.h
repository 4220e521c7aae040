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
