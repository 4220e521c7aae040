// gs_blend.h - the per-(pixel, Gaussian) alpha evaluation shared by the forward and backward blend kernels.
//
// Both kernels must take IDENTICAL skip decisions (alpha >= 1/255, power <= 0) for the same pair, so the
// expression is spelled with explicit fmaf in one place.  The conic is staged pre-multiplied into the log2
// domain (qa = -0.5 log2e cxx, qb = -log2e cxy, qc = -0.5 log2e cyy): power2 = qa dx^2 + qb dx dy + qc dy^2 feeds
// v_exp_f32 directly (5 VALU ops + 1 transcendental instead of 8 + 1); reference: forward.cu:343-356,
// backward.cu:549-560 (power = -0.5 (A dx^2 + C dy^2) - B dx dy, alpha = min(0.99, o exp(power))).
#pragma once
#include <hip/hip_runtime.h>

#define GS_LOG2E 1.4426950408889634f

__device__ __forceinline__ float4 blend_stage_conic(float4 conic_o) {
  return make_float4(-0.5f * GS_LOG2E * conic_o.x, -GS_LOG2E * conic_o.y, -0.5f * GS_LOG2E * conic_o.z, conic_o.w);
}
// returns power in the log2 domain (same sign as the reference's `power`)
__device__ __forceinline__ float blend_power2(float4 q, float dx, float dy) {
  const float t = fmaf(q.y, dy, q.x * dx);
  return fmaf(q.z * dy, dy, t * dx);
}
__device__ __forceinline__ float blend_exp2(float p2) { return __builtin_amdgcn_exp2f(p2); }
