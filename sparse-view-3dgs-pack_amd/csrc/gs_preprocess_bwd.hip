// gs_preprocess_bwd.hip - backward per-Gaussian stage (HBM-bound streaming kernel).
// Compiled with -ffp-contract=off: see gs_math.h.
//
// Does the work of computeCov2DCUDA (backward.cu:147-326) and preprocessCUDA<3> backward (backward.cu:398-449,
// with the SH backward :23-142 and the cov3D->scale/quaternion backward :330-393) in ONE pass:
// the reference launches two kernels that both re-read means / radii and round-trip dL_dmeans and
// dL_dcov3D through HBM; here dL/dSigma stays in registers between the two halves.  The mean gradient is the sum
// (cov2D part) + (projection part) + (SH view-direction part).
// The kernel writes EVERY row of every output (zeros for culled Gaussians), which replaces the
// reference's 304 B/Gaussian of cudaMemset (rasterize_points.cu:163-172).
// Per visible Gaussian: reads 64 B gradient row + 64 B splat + 24 B cov3D + 12+12+16+4 inputs
// + 192 B SH; writes 12+12+192+12+4+12+16+24 B.
#include "gs_common.h"
#include "gs_math.h"
#include "gs_backward_math.h"

__global__ void __launch_bounds__(GS_BLOCK, 4) preprocess_bwd_kernel(PreprocessBwdArgs a) {
  // dL_dsh rows (192 B per Gaussian) leave through LDS so that every store instruction of a wave covers 1 KiB of
  // consecutive addresses; written per lane (12 x 16 B at a 192-B stride) the same bytes cost 2.3x the HBM write
  // traffic (rocprofv3 WRITE_SIZE, profiles/).  Row stride 19 words (odd): conflict-free per-lane writes.
  __shared__ float s_sh[GS_BLOCK * SH_LDS_ROW];
  const int idx_raw = blockIdx.x * GS_BLOCK + threadIdx.x;
  const bool in_range = idx_raw < a.P;
  const int idx = in_range ? idx_raw : a.P - 1;  // out-of-range lanes shadow the last Gaussian and store nothing
  const bool visible = a.radii[idx] > 0;

  GeomBack gb = {};  // zeros: what a culled Gaussian stores
  // dL_dsh sink: zero row first (culled Gaussians and coefficients above the active degree stay 0)
  const bool sh_lds = a.shs && a.out.dL_dsh && a.M == 16;
  const bool sh_global = a.shs && a.out.dL_dsh && !sh_lds && in_range;  // generic M: straight to the global row
  ShSink dsh{sh_global ? a.out.dL_dsh + (size_t)idx * a.M * 3 : s_sh + threadIdx.x * SH_LDS_ROW, !sh_global};
  {
    const int nfl_row = sh_global ? a.M * 3 : SH_LDS_ROW;
    for (int k = 0; k < nfl_row; k++) dsh.p[k] = 0.f;
  }

  if (visible) geometry_backward(a, idx, gb);

  // geometry outputs leave first (their registers are free again before the 48 SH coefficients arrive); every row is
  // written, zeros when culled
  const GsGrads& o = a.out;
  if (in_range) {
    if (o.dL_dmeans2D) {
      o.dL_dmeans2D[3 * idx] = gb.dmean2D_x;
      o.dL_dmeans2D[3 * idx + 1] = gb.dmean2D_y;
      o.dL_dmeans2D[3 * idx + 2] = 0.f;
    }
    if (o.dL_dcolors) {
      o.dL_dcolors[3 * idx] = gb.dcolor.x;
      o.dL_dcolors[3 * idx + 1] = gb.dcolor.y;
      o.dL_dcolors[3 * idx + 2] = gb.dcolor.z;
    }
    if (o.dL_dopacity) o.dL_dopacity[idx] = gb.dop;
    if (o.dL_dextra) o.dL_dextra[idx] = gb.dextra;
    if (o.dL_dcov3D) {
      float2* cd = reinterpret_cast<float2*>(o.dL_dcov3D + (size_t)idx * 6);
      cd[0] = make_float2(gb.dcov[0], gb.dcov[1]);
      cd[1] = make_float2(gb.dcov[2], gb.dcov[3]);
      cd[2] = make_float2(gb.dcov[4], gb.dcov[5]);
    }
    if (a.scales) {
      if (o.dL_dscales) {
        o.dL_dscales[3 * idx] = gb.dscale.x;
        o.dL_dscales[3 * idx + 1] = gb.dscale.y;
        o.dL_dscales[3 * idx + 2] = gb.dscale.z;
      }
      if (o.dL_drotations) reinterpret_cast<float4*>(o.dL_drotations)[idx] = make_float4(gb.dq[0], gb.dq[1], gb.dq[2], gb.dq[3]);
    }
  }

  if (visible && a.shs) gb.dmean = gb.dmean + sh_backward_row(a, idx, gb.dcolor, dsh);
  if (in_range && o.dL_dmeans3D) {
    o.dL_dmeans3D[3 * idx] = gb.dmean.x;
    o.dL_dmeans3D[3 * idx + 1] = gb.dmean.y;
    o.dL_dmeans3D[3 * idx + 2] = gb.dmean.z;
  }
  if (sh_lds) {
    __syncthreads();
    const int first = blockIdx.x * GS_BLOCK;
    const int nfl = min(GS_BLOCK, a.P - first) * 48;  // floats this workgroup owns, a multiple of 4
    float* dst = o.dL_dsh + (size_t)first * 48;
    for (int j = 4 * threadIdx.x; j < nfl; j += 4 * GS_BLOCK) {
      const int r = j / 48, c = j - r * 48;  // c is a multiple of 4: the four floats are in one row
      const float* src = s_sh + r * SH_LDS_ROW;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int k = (c + e) / 3, ch = (c + e) - 3 * k;
        v[e] = src[k] * src[16 + ch];  // the very product sh_backward's generic path forms: basis_k * dL_dRGB[ch]
      }
      *reinterpret_cast<float4*>(dst + j) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

int launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((a.P + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// gs_backward_step: the same per-Gaussian stage with the tail of a single-GPU train step fused in.  Every gradient of
// a Gaussian is final when its thread has run the chain rule, so the thread goes on - activation backward, view
// statistics, Adam - while the values are still in registers (dL_dsh: in the workgroup's LDS rows).  Element
// arithmetic = act_bwd_kernel / densify_stats_kernel (gs_model.hip) and adam_kernel (gs_adam.hip), same order of
// operations, so the fused step and the three-kernel step agree bit for bit given the same blend sums.
// ------------------------------------------------------------------------------------------------------------------
GS_DEV void adam_update(float& p, float g, float& m, float& v, float lr_bc1, float inv_sqrt_bc2, float b1, float b2, float eps) {
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p = p - lr_bc1 * (m / denom);
}
// rows of `n` consecutive floats per Gaussian (any 4-byte alignment: after a densification a row's segment of the flat
// buffer need not start on 16 bytes)
template <int n>
GS_DEV void adam_row(float* p, float* m, float* v, int idx, const float* g, float lr_bc1, float inv_sqrt_bc2, float b1,
                     float b2, float eps) {
#pragma unroll
  for (int k = 0; k < n; k++) {
    float pe = p[n * (size_t)idx + k], me = m[n * (size_t)idx + k], ve = v[n * (size_t)idx + k];
    adam_update(pe, g[k], me, ve, lr_bc1, inv_sqrt_bc2, b1, b2, eps);
    p[n * (size_t)idx + k] = pe;
    m[n * (size_t)idx + k] = me;
    v[n * (size_t)idx + k] = ve;
  }
}

__global__ void __launch_bounds__(GS_BLOCK, 4) preprocess_bwd_step_kernel(PreprocessBwdArgs a, StepArgs sa) {
  __shared__ float s_sh[GS_BLOCK * SH_LDS_ROW];
  const GsStepState& st = sa.st;
  // the forward ran out of binning capacity (possible only when the caller did not re-run it: a replayed graph): the
  // image was not rendered, so nothing may be updated - the host sees the flag and repeats the step eagerly
  if (*sa.overflow) return;
  // step-dependent constants from device memory when the launch is replayed from a captured graph
  if (st.coef_dev) {
#pragma unroll
    for (int k = 0; k < 6; k++) sa.lr_bc1[k] = st.coef_dev[k];
#pragma unroll
    for (int k = 0; k < 5; k++) sa.inv_sqrt_bc2[k] = st.coef_dev[6 + k];
  }
  const int idx_raw = blockIdx.x * GS_BLOCK + threadIdx.x;
  const bool in_range = idx_raw < a.P;
  const int idx = in_range ? idx_raw : a.P - 1;
  const int radius = a.radii[idx];
  const bool visible = radius > 0;
  const float b1 = st.beta1, b2 = st.beta2, eps = st.eps;

  GeomBack gb = {};
  ShSink dsh{s_sh + threadIdx.x * SH_LDS_ROW, true};
#pragma unroll
  for (int k = 0; k < SH_LDS_ROW; k++) dsh.p[k] = 0.f;
  if (visible) geometry_backward(a, idx, gb);

  if (in_range) {
    // ---- view statistics (train.py:266-268, gaussian_model.py:471-473)
    if (visible && st.max_radii2D) {
      st.max_radii2D[idx] = fmaxf(st.max_radii2D[idx], (float)radius);
      st.xyz_gradient_accum[idx] += sqrtf(gb.dmean2D_x * gb.dmean2D_x + gb.dmean2D_y * gb.dmean2D_y);
      st.denom[idx] += 1.0f;
    }
    // ---- opacity = sigmoid(raw): grad * s * (1 - s)
    if (st.step[2] > 0) {
      const float sg = 1.0f / (1.0f + expf(-st.opacity[idx]));
      const float g = gb.dop * (1.0f - sg) * sg;
      adam_row<1>(st.opacity, st.m[2], st.v[2], idx, &g, sa.lr_bc1[3], sa.inv_sqrt_bc2[2], b1, b2, eps);
    }
    // ---- scaling = exp(raw): grad * result
    if (st.step[3] > 0) {
      float g[3];
      g[0] = gb.dscale.x * expf(st.scaling[3 * (size_t)idx]);
      g[1] = gb.dscale.y * expf(st.scaling[3 * (size_t)idx + 1]);
      g[2] = gb.dscale.z * expf(st.scaling[3 * (size_t)idx + 2]);
      adam_row<3>(st.scaling, st.m[3], st.v[3], idx, g, sa.lr_bc1[4], sa.inv_sqrt_bc2[3], b1, b2, eps);
    }
    // ---- rotation = q / max(|q|, 1e-12): dq = (g - v (v . g)) / |q|
    if (st.step[4] > 0) {
      const float* qr = st.rotation + 4 * (size_t)idx;
      const float4 q = make_float4(qr[0], qr[1], qr[2], qr[3]);
      const float norm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
      float g[4];
      if (norm > 1e-12f) {
        const float inv = 1.0f / norm;
        const float vx = q.x * inv, vy = q.y * inv, vz = q.z * inv, vw = q.w * inv;
        const float dot = vx * gb.dq[0] + vy * gb.dq[1] + vz * gb.dq[2] + vw * gb.dq[3];
        g[0] = (gb.dq[0] - vx * dot) * inv;
        g[1] = (gb.dq[1] - vy * dot) * inv;
        g[2] = (gb.dq[2] - vz * dot) * inv;
        g[3] = (gb.dq[3] - vw * dot) * inv;
      } else {
        g[0] = gb.dq[0] / 1e-12f; g[1] = gb.dq[1] / 1e-12f; g[2] = gb.dq[2] / 1e-12f; g[3] = gb.dq[3] / 1e-12f;
      }
      adam_row<4>(st.rotation, st.m[4], st.v[4], idx, g, sa.lr_bc1[5], sa.inv_sqrt_bc2[4], b1, b2, eps);
    }
  }

  // ---- SH half: basis values and colour gradient to the LDS row; the view-direction part completes dL_dmean
  if (visible) gb.dmean = gb.dmean + sh_backward_row(a, idx, gb.dcolor, dsh);
  if (in_range && st.step[0] > 0) {
    const float g[3] = {gb.dmean.x, gb.dmean.y, gb.dmean.z};
    adam_row<3>(st.xyz, st.m[0], st.v[0], idx, g, sa.lr_bc1[0], sa.inv_sqrt_bc2[0], b1, b2, eps);
  }
  __syncthreads();  // every thread of the workgroup has read its SH coefficients and positions: rows may now change
  if (st.step[1] > 0) {
    const int first = blockIdx.x * GS_BLOCK;
    const int nfl = min(GS_BLOCK, a.P - first) * 48;
    float* pf = st.features + (size_t)first * 48;
    float* mf = st.m[1] + (size_t)first * 48;
    float* vf = st.v[1] + (size_t)first * 48;
    const float lr_dc = sa.lr_bc1[1], lr_rest = sa.lr_bc1[2], isb = sa.inv_sqrt_bc2[1];
    const bool vec = ((((uintptr_t)pf) | ((uintptr_t)mf) | ((uintptr_t)vf)) & 15) == 0;
    for (int j = 4 * threadIdx.x; j < nfl; j += 4 * GS_BLOCK) {
      const int r = j / 48, c = j - r * 48;
      const float* src = s_sh + r * SH_LDS_ROW;
      float pe[4], me[4], ve[4];
      if (vec) {
        const float4 p4 = *reinterpret_cast<const float4*>(pf + j), m4 = *reinterpret_cast<const float4*>(mf + j),
                     v4 = *reinterpret_cast<const float4*>(vf + j);
        pe[0] = p4.x; pe[1] = p4.y; pe[2] = p4.z; pe[3] = p4.w;
        me[0] = m4.x; me[1] = m4.y; me[2] = m4.z; me[3] = m4.w;
        ve[0] = v4.x; ve[1] = v4.y; ve[2] = v4.z; ve[3] = v4.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) { pe[e] = pf[j + e]; me[e] = mf[j + e]; ve[e] = vf[j + e]; }
      }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int k = (c + e) / 3, ch = (c + e) - 3 * k;
        const float g = src[k] * src[16 + ch];  // dL_dsh[r][k][ch] = basis_k * dL_dRGB[ch]
        adam_update(pe[e], g, me[e], ve[e], (c + e) < 3 ? lr_dc : lr_rest, isb, b1, b2, eps);
      }
      if (vec) {
        *reinterpret_cast<float4*>(pf + j) = make_float4(pe[0], pe[1], pe[2], pe[3]);
        *reinterpret_cast<float4*>(mf + j) = make_float4(me[0], me[1], me[2], me[3]);
        *reinterpret_cast<float4*>(vf + j) = make_float4(ve[0], ve[1], ve[2], ve[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) { pf[j + e] = pe[e]; mf[j + e] = me[e]; vf[j + e] = ve[e]; }
      }
    }
  }
}

int launch_preprocess_bwd_step(const PreprocessBwdArgs& a, const StepArgs& sa, hipStream_t s) {
  hipLaunchKernelGGL(preprocess_bwd_step_kernel, dim3((a.P + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, a, sa);
  return 0;
}
