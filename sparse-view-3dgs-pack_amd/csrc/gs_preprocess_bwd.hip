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
