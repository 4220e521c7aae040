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

// ------------------------------------------------------------------------------------------------------------------
// The visibility-sparse gradient exchange (include/gsplat.h): rows of the union in and out of the packed buffer.
// One thread per (row, float of the row's 59 / 60): reads and writes runs of a field's width - a gather of short rows.
// ------------------------------------------------------------------------------------------------------------------
#define GS_PACK_MAX_FIELDS 8
struct PackFields {
  int n;
  int width[GS_PACK_MAX_FIELDS];
  long long flat_off[GS_PACK_MAX_FIELDS];    // start of the field's [P, width] block in the flat buffer
  long long packed_off[GS_PACK_MAX_FIELDS];  // start of its [K, width] block in the packed buffer
  int col_off[GS_PACK_MAX_FIELDS + 1];       // first column of the field among a row's total floats
};
template <bool UNPACK>
__global__ void __launch_bounds__(GS_BLOCK) rows_pack_kernel(float* __restrict__ flat, int P, PackFields f, const uint8_t* __restrict__ mask,
                                                             const int32_t* __restrict__ pos, int K, float* __restrict__ packed) {
  const int W = f.col_off[f.n];
  const long long total = (long long)P * W;
  for (long long t = (long long)blockIdx.x * GS_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * GS_BLOCK) {
    const int i = (int)(t / W), col = (int)(t - (long long)i * W);
    if (!mask[i]) continue;
    const int k = pos[i];
    if (k < 0 || k >= K) continue;
    int fi = 0;
#pragma unroll
    for (int q = 1; q < GS_PACK_MAX_FIELDS; q++) fi = (q < f.n && col >= f.col_off[q]) ? q : fi;
    const int j = col - f.col_off[fi];
    float* a = flat + f.flat_off[fi] + (long long)i * f.width[fi] + j;
    float* b = packed + f.packed_off[fi] + (long long)k * f.width[fi] + j;
    if (UNPACK) *a = *b; else *b = *a;
  }
}
static int rows_pack_impl(float* flat, int32_t P, int32_t nfields, const int32_t* widths, const uint8_t* mask, const int32_t* pos,
                          int32_t K, float* packed, bool unpack, void* stream) {
  if (P < 0 || K < 0 || nfields < 0 || nfields > GS_PACK_MAX_FIELDS) return GS_E_SHAPE;
  if (P == 0 || K == 0 || nfields == 0) return GS_OK;
  if (!flat || !widths || !mask || !pos || !packed) return GS_E_NULL;
  PackFields f;
  f.n = nfields;
  long long fo = 0, po = 0;
  int co = 0;
  for (int q = 0; q < GS_PACK_MAX_FIELDS; q++) {
    const int w = q < nfields ? widths[q] : 0;
    if (w < 0) return GS_E_SHAPE;
    f.width[q] = w;
    f.flat_off[q] = fo;
    f.packed_off[q] = po;
    f.col_off[q] = co;
    fo += (long long)P * w;
    po += (long long)K * w;
    co += w;
  }
  f.col_off[GS_PACK_MAX_FIELDS] = co;
  for (int q = nfields; q < GS_PACK_MAX_FIELDS; q++) f.col_off[q] = co;
  f.col_off[nfields] = co;
  if (co == 0) return GS_OK;
  hipStream_t s = (hipStream_t)stream;
  long long blocks = ((long long)P * co + GS_BLOCK - 1) / GS_BLOCK;
  blocks = blocks > 65536 ? 65536 : blocks;
  GS_PROF(ST_MODEL, s);
  if (unpack)
    hipLaunchKernelGGL(rows_pack_kernel<true>, dim3((unsigned)blocks), dim3(GS_BLOCK), 0, s, flat, P, f, mask, pos, K, packed);
  else
    hipLaunchKernelGGL(rows_pack_kernel<false>, dim3((unsigned)blocks), dim3(GS_BLOCK), 0, s, flat, P, f, mask, pos, K, packed);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
extern "C" int gs_rows_pack(const float* flat, int32_t P, int32_t nfields, const int32_t* widths, const uint8_t* mask,
                            const int32_t* pos, int32_t K, float* packed, void* stream) {
  return rows_pack_impl(const_cast<float*>(flat), P, nfields, widths, mask, pos, K, packed, false, stream);
}
extern "C" int gs_rows_unpack(float* flat, int32_t P, int32_t nfields, const int32_t* widths, const uint8_t* mask,
                              const int32_t* pos, int32_t K, const float* packed, void* stream) {
  return rows_pack_impl(flat, P, nfields, widths, mask, pos, K, const_cast<float*>(packed), true, stream);
}
__global__ void __launch_bounds__(GS_BLOCK) row_mask_kernel(const uint32_t* __restrict__ tiles_touched, int P, uint8_t* __restrict__ mask) {
  const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (i < P) mask[i] = tiles_touched[i] != 0u ? 1 : 0;
}
extern "C" int gs_export_row_mask(const GsScratch* sc, int32_t P, uint8_t* mask, void* stream) {
  if (!sc || !sc->geom || !mask) return GS_E_NULL;
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (sc->geom_bytes < geom_bytes((size_t)P)) return GS_E_SCRATCH;
  GeomView gv = geom_view(sc->geom, (size_t)P);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(row_mask_kernel, dim3((P + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, gv.tiles_touched, P, mask);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
