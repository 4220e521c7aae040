// gs_model.hip - the per-Gaussian elementwise work of the train step outside the rasterizer:
//   * parameter activations  scaling = exp(_scaling), rotation = normalize(_rotation), opacity = sigmoid(_opacity)
//     (LGDWT-GS/scene/gaussian_model.py:40-60,102-117) and their backward, in ONE kernel per direction over the
//     flat parameter / gradient buffers - the reference runs ~12 small torch kernels for them per iteration;
//   * densification statistics  max_radii2D[vis] = max(., radii[vis]), xyz_gradient_accum[vis] += |dL_dmean2D.xy|,
//     denom[vis] += 1  (train.py:266-268, gaussian_model.py:471-473) in one kernel.
// Arithmetic restates what torch does for these ops (exp, F.normalize with eps 1e-12, sigmoid and their autograd
// formulas); pinned against torch autograd in tests/test_model_ops.py.
#include <math.h>

#include "gs_common.h"
#include "gs_prof.h"

__global__ void __launch_bounds__(GS_BLOCK) act_fwd_kernel(const float* __restrict__ scaling, const float* __restrict__ rotation,
                                                           const float* __restrict__ opacity, int P, float* __restrict__ o_scales,
                                                           float* __restrict__ o_rot, float* __restrict__ o_opac) {
  const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (i >= P) return;
  o_scales[3 * i] = expf(scaling[3 * i]);
  o_scales[3 * i + 1] = expf(scaling[3 * i + 1]);
  o_scales[3 * i + 2] = expf(scaling[3 * i + 2]);
  // (scalar accesses: after a densification the rotation segment of the flat buffer starts at 220 P bytes, which is
  // 16-byte aligned only when P is a multiple of 4)
  const float4 q = make_float4(rotation[4 * i], rotation[4 * i + 1], rotation[4 * i + 2], rotation[4 * i + 3]);
  // F.normalize: v / max(||v||_2, eps), eps = 1e-12
  const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  o_rot[4 * i] = q.x / n;
  o_rot[4 * i + 1] = q.y / n;
  o_rot[4 * i + 2] = q.z / n;
  o_rot[4 * i + 3] = q.w / n;
  o_opac[i] = 1.0f / (1.0f + expf(-opacity[i]));
}

__global__ void __launch_bounds__(GS_BLOCK) act_bwd_kernel(const float* __restrict__ scaling, const float* __restrict__ rotation,
                                                           const float* __restrict__ opacity, int P,
                                                           const float* __restrict__ g_scales, const float* __restrict__ g_rot,
                                                           const float* __restrict__ g_opac, float* __restrict__ d_scaling,
                                                           float* __restrict__ d_rotation, float* __restrict__ d_opacity) {
  const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (i >= P) return;
  // exp backward: grad * result
  d_scaling[3 * i] = g_scales[3 * i] * expf(scaling[3 * i]);
  d_scaling[3 * i + 1] = g_scales[3 * i + 1] * expf(scaling[3 * i + 1]);
  d_scaling[3 * i + 2] = g_scales[3 * i + 2] * expf(scaling[3 * i + 2]);
  // v = q / n, n = max(||q||, eps):  dq = (g - v (v . g)) / n   (n clamped: dq = g / eps)
  const float4 q = make_float4(rotation[4 * i], rotation[4 * i + 1], rotation[4 * i + 2], rotation[4 * i + 3]);
  const float4 g = make_float4(g_rot[4 * i], g_rot[4 * i + 1], g_rot[4 * i + 2], g_rot[4 * i + 3]);
  const float norm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  float4 d;
  if (norm > 1e-12f) {
    const float inv = 1.0f / norm;
    const float vx = q.x * inv, vy = q.y * inv, vz = q.z * inv, vw = q.w * inv;
    const float dot = vx * g.x + vy * g.y + vz * g.z + vw * g.w;
    d = make_float4((g.x - vx * dot) * inv, (g.y - vy * dot) * inv, (g.z - vz * dot) * inv, (g.w - vw * dot) * inv);
  } else {
    d = make_float4(g.x / 1e-12f, g.y / 1e-12f, g.z / 1e-12f, g.w / 1e-12f);
  }
  d_rotation[4 * i] = d.x;
  d_rotation[4 * i + 1] = d.y;
  d_rotation[4 * i + 2] = d.z;
  d_rotation[4 * i + 3] = d.w;
  // sigmoid backward: grad * s * (1 - s)
  const float s = 1.0f / (1.0f + expf(-opacity[i]));
  d_opacity[i] = g_opac[i] * (1.0f - s) * s;
}

__global__ void __launch_bounds__(GS_BLOCK) densify_stats_kernel(const int32_t* __restrict__ radii,
                                                                 const float* __restrict__ dL_dmeans2D, int P,
                                                                 float* __restrict__ max_radii2D, float* __restrict__ accum,
                                                                 float* __restrict__ denom) {
  const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (i >= P) return;
  const int r = radii[i];
  if (r > 0) {
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    const float gx = dL_dmeans2D[3 * i], gy = dL_dmeans2D[3 * i + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
  }
}

extern "C" {

int gs_activations_fwd(const float* scaling, const float* rotation, const float* opacity, int32_t P, float* scales_out,
                       float* rotations_out, float* opacities_out, void* stream) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!scaling || !rotation || !opacity || !scales_out || !rotations_out || !opacities_out) return GS_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_MODEL, s);
  hipLaunchKernelGGL(act_fwd_kernel, dim3((P + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, scaling, rotation, opacity, P,
                     scales_out, rotations_out, opacities_out);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}

int gs_activations_bwd(const float* scaling, const float* rotation, const float* opacity, int32_t P, const float* dL_dscales,
                       const float* dL_drotations, const float* dL_dopacities, float* d_scaling, float* d_rotation,
                       float* d_opacity, void* stream) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!scaling || !rotation || !opacity || !dL_dscales || !dL_drotations || !dL_dopacities || !d_scaling || !d_rotation ||
      !d_opacity)
    return GS_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_MODEL, s);
  hipLaunchKernelGGL(act_bwd_kernel, dim3((P + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, scaling, rotation, opacity, P,
                     dL_dscales, dL_drotations, dL_dopacities, d_scaling, d_rotation, d_opacity);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}

int gs_densify_stats(const int32_t* radii, const float* dL_dmeans2D, int32_t P, float* max_radii2D, float* xyz_gradient_accum,
                     float* denom, void* stream) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!radii || !dL_dmeans2D || !max_radii2D || !xyz_gradient_accum || !denom) return GS_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_MODEL, s);
  hipLaunchKernelGGL(densify_stats_kernel, dim3((P + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, radii, dL_dmeans2D, P,
                     max_radii2D, xyz_gradient_accum, denom);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
}
