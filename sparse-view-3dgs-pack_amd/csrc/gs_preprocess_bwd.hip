// gs_preprocess_bwd.hip - backward per-Gaussian stage (HBM-bound streaming kernel).
// Compiled with -ffp-contract=off: see gs_math.h.
//
// Fuses computeCov2DCUDA (backward.cu:147-326) and preprocessCUDA<3> backward (backward.cu:398-449,
// with the SH backward :23-142 and the cov3D->scale/quaternion backward :330-393) into ONE pass:
// the reference launches two kernels that both re-read means / radii and round-trip dL_dmeans and
// dL_dcov3D through HBM.  The mean gradient is formed in the reference's order
// ((cov2D part) + (projection part)) + (SH view-direction part).
// The kernel writes EVERY row of every output (zeros for culled Gaussians), which replaces the
// reference's 304 B/Gaussian of cudaMemset (rasterize_points.cu:163-172).
// Per visible Gaussian: reads 64 B gradient row + 64 B splat + 24 B cov3D + 12+12+16+4 inputs
// + 192 B SH; writes 12+12+192+12+4+12+16+24 B.
#include "gs_common.h"
#include "gs_math.h"

GS_DEV float sq(float x) { return x * x; }

struct ShRegsB {
  float f[48];
  __device__ __forceinline__ V3 operator()(int k) const { return {f[3 * k], f[3 * k + 1], f[3 * k + 2]}; }
};

// Where sh_backward puts row k of dL_dsh = basis_k(dir) * dL_dRGB.  M == 16: only the sixteen basis values and the
// (clamp-masked) colour gradient go to an LDS row of 19 words, and the coalesced store phase forms the 48 products
// (the full 48-word rows took 49 KB of LDS and held the kernel at 3 workgroups per CU).  Generic M: the products go
// straight to the global row.  (A register array here ends up in scratch memory - the generic-M path indexes it
// dynamically - which showed up as 384 B per Gaussian of extra HBM write traffic in the WRITE_SIZE counter.)
#define SH_LDS_ROW 19
struct ShSink {
  float* p;
  bool basis_only;
  __device__ __forceinline__ void rgb(V3 g) const {
    if (basis_only) {
      p[16] = g.x;
      p[17] = g.y;
      p[18] = g.z;
    }
  }
  __device__ __forceinline__ void set(int k, float basis, V3 g) const {
    if (basis_only) {
      p[k] = basis;
    } else {
      p[3 * k] = basis * g.x;
      p[3 * k + 1] = basis * g.y;
      p[3 * k + 2] = basis * g.z;
    }
  }
};

// backward.cu:23-142.  Writes the dL_dsh rows it computes through `dsh` and returns dL_dmean.
template <typename SH>
GS_DEV V3 sh_backward(int deg, V3 pos, V3 campos, const SH& sh, uint32_t clamped, V3 dL_dRGB, const ShSink& dsh) {
  V3 dir_orig = pos - campos;
  V3 dir = dir_orig / length3(dir_orig);
  dL_dRGB.x *= (clamped & 1u) ? 0 : 1;
  dL_dRGB.y *= (clamped & 2u) ? 0 : 1;
  dL_dRGB.z *= (clamped & 4u) ? 0 : 1;
  dsh.rgb(dL_dRGB);
  V3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
  float x = dir.x, y = dir.y, z = dir.z;
  float dRGBdsh0 = SH_C0;
  dsh.set(0, dRGBdsh0, dL_dRGB);
  if (deg > 0) {
    float dRGBdsh1 = -SH_C1 * y;
    float dRGBdsh2 = SH_C1 * z;
    float dRGBdsh3 = -SH_C1 * x;
    dsh.set(1, dRGBdsh1, dL_dRGB);
    dsh.set(2, dRGBdsh2, dL_dRGB);
    dsh.set(3, dRGBdsh3, dL_dRGB);
    dRGBdx = -SH_C1 * sh(3);
    dRGBdy = -SH_C1 * sh(1);
    dRGBdz = SH_C1 * sh(2);
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z;
      float xy = x * y, yz = y * z, xz = x * z;
      float dRGBdsh4 = SH_C2_0 * xy;
      float dRGBdsh5 = SH_C2_1 * yz;
      float dRGBdsh6 = SH_C2_2 * (2.f * zz - xx - yy);
      float dRGBdsh7 = SH_C2_3 * xz;
      float dRGBdsh8 = SH_C2_4 * (xx - yy);
      dsh.set(4, dRGBdsh4, dL_dRGB);
      dsh.set(5, dRGBdsh5, dL_dRGB);
      dsh.set(6, dRGBdsh6, dL_dRGB);
      dsh.set(7, dRGBdsh7, dL_dRGB);
      dsh.set(8, dRGBdsh8, dL_dRGB);
      dRGBdx = dRGBdx + (SH_C2_0 * y * sh(4) + SH_C2_2 * 2.f * -x * sh(6) + SH_C2_3 * z * sh(7) + SH_C2_4 * 2.f * x * sh(8));
      dRGBdy = dRGBdy + (SH_C2_0 * x * sh(4) + SH_C2_1 * z * sh(5) + SH_C2_2 * 2.f * -y * sh(6) + SH_C2_4 * 2.f * -y * sh(8));
      dRGBdz = dRGBdz + (SH_C2_1 * y * sh(5) + SH_C2_2 * 2.f * 2.f * z * sh(6) + SH_C2_3 * x * sh(7));
      if (deg > 2) {
        float dRGBdsh9 = SH_C3_0 * y * (3.f * xx - yy);
        float dRGBdsh10 = SH_C3_1 * xy * z;
        float dRGBdsh11 = SH_C3_2 * y * (4.f * zz - xx - yy);
        float dRGBdsh12 = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
        float dRGBdsh13 = SH_C3_4 * x * (4.f * zz - xx - yy);
        float dRGBdsh14 = SH_C3_5 * z * (xx - yy);
        float dRGBdsh15 = SH_C3_6 * x * (xx - 3.f * yy);
        dsh.set(9, dRGBdsh9, dL_dRGB);
        dsh.set(10, dRGBdsh10, dL_dRGB);
        dsh.set(11, dRGBdsh11, dL_dRGB);
        dsh.set(12, dRGBdsh12, dL_dRGB);
        dsh.set(13, dRGBdsh13, dL_dRGB);
        dsh.set(14, dRGBdsh14, dL_dRGB);
        dsh.set(15, dRGBdsh15, dL_dRGB);
        dRGBdx = dRGBdx + (SH_C3_0 * sh(9) * 3.f * 2.f * xy + SH_C3_1 * sh(10) * yz + SH_C3_2 * sh(11) * -2.f * xy +
                           SH_C3_3 * sh(12) * -3.f * 2.f * xz + SH_C3_4 * sh(13) * (-3.f * xx + 4.f * zz - yy) +
                           SH_C3_5 * sh(14) * 2.f * xz + SH_C3_6 * sh(15) * 3.f * (xx - yy));
        dRGBdy = dRGBdy + (SH_C3_0 * sh(9) * 3.f * (xx - yy) + SH_C3_1 * sh(10) * xz +
                           SH_C3_2 * sh(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3_3 * sh(12) * -3.f * 2.f * yz +
                           SH_C3_4 * sh(13) * -2.f * xy + SH_C3_5 * sh(14) * -2.f * yz + SH_C3_6 * sh(15) * -3.f * 2.f * xy);
        dRGBdz = dRGBdz + (SH_C3_1 * sh(10) * xy + SH_C3_2 * sh(11) * 4.f * 2.f * yz +
                           SH_C3_3 * sh(12) * 3.f * (2.f * zz - xx - yy) + SH_C3_4 * sh(13) * 4.f * 2.f * xz +
                           SH_C3_5 * sh(14) * (xx - yy));
      }
    }
  }
  V3 dL_ddir = {dot3(dRGBdx, dL_dRGB), dot3(dRGBdy, dL_dRGB), dot3(dRGBdz, dL_dRGB)};
  return dnormvdv3(dir_orig, dL_ddir);
}

__global__ void __launch_bounds__(GS_BLOCK, 4) preprocess_bwd_kernel(PreprocessBwdArgs a) {
  // dL_dsh rows (192 B per Gaussian) leave through LDS so that every store instruction of a wave covers 1 KiB of
  // consecutive addresses; written per lane (12 x 16 B at a 192-B stride) the same bytes cost 2.3x the HBM write
  // traffic (rocprofv3 WRITE_SIZE, profiles/).  Row stride 19 words (odd): conflict-free per-lane writes.
  __shared__ float s_sh[GS_BLOCK * SH_LDS_ROW];
  const int idx_raw = blockIdx.x * GS_BLOCK + threadIdx.x;
  const bool in_range = idx_raw < a.P;
  const int idx = in_range ? idx_raw : a.P - 1;  // out-of-range lanes shadow the last Gaussian and store nothing
  const bool visible = a.radii[idx] > 0;

  V3 dL_dmean = {0, 0, 0};
  float dL_dmean2D_x = 0.f, dL_dmean2D_y = 0.f;
  float dL_dcov[6] = {0, 0, 0, 0, 0, 0};
  float dL_dop = 0.f, dL_dextra = 0.f;
  V3 dL_dcolor = {0, 0, 0};
  V3 dL_dscale = {0, 0, 0};
  float dL_dq[4] = {0, 0, 0, 0};
  // dL_dsh sink: zero row first (culled Gaussians and coefficients above the active degree stay 0)
  const bool sh_lds = a.shs && a.out.dL_dsh && a.M == 16;
  const bool sh_global = a.shs && a.out.dL_dsh && !sh_lds && in_range;  // generic M: straight to the global row
  ShSink dsh{sh_global ? a.out.dL_dsh + (size_t)idx * a.M * 3 : s_sh + threadIdx.x * SH_LDS_ROW, !sh_global};
  {
    const int nfl_row = sh_global ? a.M * 3 : SH_LDS_ROW;
    for (int k = 0; k < nfl_row; k++) dsh.p[k] = 0.f;
  }

  if (visible) {
    const float4* gr = reinterpret_cast<const float4*>(a.grad_rows + (size_t)idx * GR_STRIDE);
    const float4 g0 = gr[0], g1 = gr[1], g2 = gr[2];
    dL_dmean2D_x = g0.x;
    dL_dmean2D_y = g0.y;
    V3 dL_dconic = {g0.z, g0.w, g1.x};
    dL_dop = g1.y;
    dL_dcolor = {g1.z, g1.w, g2.x};
    const float dL_dinvdepth = g2.y;
    dL_dextra = g2.z;

    // ------------------------------------------------------------------ computeCov2DCUDA
    const float* cov3D = a.cov3D + 6 * (size_t)idx;
    V3 mean = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
    float c3[6];
#pragma unroll
    for (int k = 0; k < 6; k++) c3[k] = cov3D[k];
    Cov2DInter c;
    cov2d_common(mean, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, c3, a.viewmatrix, c);
    const V3 t = c.t;
    const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0 : 1;
    const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0 : 1;
    const M3& T = c.T;
    const M3& Vrk = c.Vrk;
    const float* vm = a.viewmatrix;
    M3 Wm = mat3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    M3 cov2D = mul3(mul3(transpose3(T), transpose3(Vrk)), T);
    float c_xx = cov2D.c[0][0];
    float c_xy = cov2D.c[0][1];
    float c_yy = cov2D.c[1][1];
    const float h_var = 0.3f;
    float d_inside_root = 0.f;
    if (a.antialiasing) {
      const float det_cov = c_xx * c_yy - c_xy * c_xy;
      c_xx += h_var;
      c_yy += h_var;
      const float det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
      const float h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
      const float dL_dopacity_v = dL_dop;
      const float d_h_convolution_scaling = dL_dopacity_v * a.opacities[idx];
      dL_dop = dL_dopacity_v * h_convolution_scaling;
      d_inside_root = (det_cov / det_cov_plus_h_cov) <= 0.000025f ? 0.f : d_h_convolution_scaling / (2 * h_convolution_scaling);
    } else {
      c_xx += h_var;
      c_yy += h_var;
    }
    float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
    if (a.antialiasing) {
      const float x = c_xx, y = c_yy, z = c_xy, w = h_var;
      const float denom_f = d_inside_root / sq(w * w + w * (x + y) + x * y - z * z);
      const float dL_dx = w * (w * y + y * y + z * z) * denom_f;
      const float dL_dy = w * (w * x + x * x + z * z) * denom_f;
      const float dL_dz = -2.f * w * z * (w + x + y) * denom_f;
      dL_dc_xx = dL_dx;
      dL_dc_yy = dL_dy;
      dL_dc_xy = dL_dz;
    }
    float denom = c_xx * c_yy - c_xy * c_xy;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    if (denom2inv != 0) {
      dL_dc_xx += denom2inv * (-c_yy * c_yy * dL_dconic.x + 2 * c_xy * c_yy * dL_dconic.y + (denom - c_xx * c_yy) * dL_dconic.z);
      dL_dc_yy += denom2inv * (-c_xx * c_xx * dL_dconic.z + 2 * c_xx * c_xy * dL_dconic.y + (denom - c_xx * c_yy) * dL_dconic.x);
      dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dL_dconic.x - (denom + 2 * c_xy * c_xy) * dL_dconic.y + c_xx * c_xy * dL_dconic.z);
      dL_dcov[0] = (T.c[0][0] * T.c[0][0] * dL_dc_xx + T.c[0][0] * T.c[1][0] * dL_dc_xy + T.c[1][0] * T.c[1][0] * dL_dc_yy);
      dL_dcov[3] = (T.c[0][1] * T.c[0][1] * dL_dc_xx + T.c[0][1] * T.c[1][1] * dL_dc_xy + T.c[1][1] * T.c[1][1] * dL_dc_yy);
      dL_dcov[5] = (T.c[0][2] * T.c[0][2] * dL_dc_xx + T.c[0][2] * T.c[1][2] * dL_dc_xy + T.c[1][2] * T.c[1][2] * dL_dc_yy);
      dL_dcov[1] = 2 * T.c[0][0] * T.c[0][1] * dL_dc_xx + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_dc_xy + 2 * T.c[1][0] * T.c[1][1] * dL_dc_yy;
      dL_dcov[2] = 2 * T.c[0][0] * T.c[0][2] * dL_dc_xx + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_dc_xy + 2 * T.c[1][0] * T.c[1][2] * dL_dc_yy;
      dL_dcov[4] = 2 * T.c[0][2] * T.c[0][1] * dL_dc_xx + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_dc_xy + 2 * T.c[1][1] * T.c[1][2] * dL_dc_yy;
    }
    float dL_dT00 = 2 * (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_dc_xx +
                    (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_dc_xy;
    float dL_dT01 = 2 * (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_dc_xx +
                    (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_dc_xy;
    float dL_dT02 = 2 * (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_dc_xx +
                    (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_dc_xy;
    float dL_dT10 = 2 * (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_dc_yy +
                    (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_dc_xy;
    float dL_dT11 = 2 * (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_dc_yy +
                    (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_dc_xy;
    float dL_dT12 = 2 * (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_dc_yy +
                    (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_dc_xy;
    float dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
    float dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
    float dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
    float dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
    float tz = 1.f / t.z;
    float tz2 = tz * tz;
    float tz3 = tz2 * tz;
    const float h_x = a.focal_x, h_y = a.focal_y;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    if (a.has_invdepth == 1) dL_dtz -= dL_dinvdepth / (t.z * t.z);
    dL_dmean = xformvec4x3T({dL_dtx, dL_dty, dL_dtz}, vm);  // "=" of backward.cu:325

    // ------------------------------------------------------------------ preprocessCUDA backward
    const float* proj = a.projmatrix;
    V3 m = mean;
    V4 m_hom = xform4x4(m, proj);
    float m_w = 1.0f / (m_hom.w + 0.0000001f);
    float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    const float gx = dL_dmean2D_x, gy = dL_dmean2D_y;
    V3 dm;
    dm.x = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
    dm.y = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
    dm.z = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
    dL_dmean = dL_dmean + dm;  // "+=" of backward.cu:440
    if (a.has_invdepth == 2) {  // FSGS generation: the slot holds dL_ddepth (view-space z), -confidence backward.cu:394-403
      const float mul3 = vm[2] * m.x + vm[6] * m.y + vm[10] * m.z + vm[14];
      V3 dm2 = {(vm[2] - vm[3] * mul3) * dL_dinvdepth, (vm[6] - vm[7] * mul3) * dL_dinvdepth,
                (vm[10] - vm[11] * mul3) * dL_dinvdepth};
      dL_dmean = dL_dmean + dm2;
    }

    if (a.shs) {
      const uint32_t clamped = a.splat[idx].clamped;
      V3 campos = {a.campos[0], a.campos[1], a.campos[2]};
      V3 dms;
      if (a.M == 16) {
        ShRegsB sh;
        const float4* src = reinterpret_cast<const float4*>(a.shs + (size_t)idx * 48);
        const int nvec = a.D == 0 ? 1 : (a.D == 1 ? 3 : (a.D == 2 ? 7 : 12));
#pragma unroll
        for (int k = 0; k < 12; k++) {
          if (k < nvec) {
            float4 v = src[k];
            sh.f[4 * k] = v.x; sh.f[4 * k + 1] = v.y; sh.f[4 * k + 2] = v.z; sh.f[4 * k + 3] = v.w;
          } else {
            sh.f[4 * k] = sh.f[4 * k + 1] = sh.f[4 * k + 2] = sh.f[4 * k + 3] = 0.f;
          }
        }
        dms = sh_backward(a.D, m, campos, sh, clamped, dL_dcolor, dsh);
      } else {
        struct ShMemB {
          const float* p;
          __device__ __forceinline__ V3 operator()(int k) const { return {p[3 * k], p[3 * k + 1], p[3 * k + 2]}; }
        } sh{a.shs + (size_t)idx * a.M * 3};
        dms = sh_backward(a.D, m, campos, sh, clamped, dL_dcolor, dsh);
      }
      dL_dmean = dL_dmean + dms;  // "+=" of backward.cu:141
    }

    if (a.scales) {
      // computeCov3D backward, backward.cu:330-393
      V3 scl = {a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]};
      const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[idx];
      V4 rot = {q4.x, q4.y, q4.z, q4.w};
      float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
      M3 R = quat_to_R(rot);
      M3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
      V3 s = a.scale_modifier * scl;
      S.c[0][0] = s.x;
      S.c[1][1] = s.y;
      S.c[2][2] = s.z;
      M3 Mm = mul3(S, R);
      const float* d = dL_dcov;
      M3 dL_dSigma = mat3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
      M3 dL_dM = mul3(scale3(2.0f, Mm), dL_dSigma);
      M3 Rt = transpose3(R);
      M3 Dm = transpose3(dL_dM);
      V3 Rt0 = {Rt.c[0][0], Rt.c[0][1], Rt.c[0][2]}, Rt1 = {Rt.c[1][0], Rt.c[1][1], Rt.c[1][2]}, Rt2 = {Rt.c[2][0], Rt.c[2][1], Rt.c[2][2]};
      V3 m0 = {Dm.c[0][0], Dm.c[0][1], Dm.c[0][2]}, m1 = {Dm.c[1][0], Dm.c[1][1], Dm.c[1][2]}, m2 = {Dm.c[2][0], Dm.c[2][1], Dm.c[2][2]};
      dL_dscale.x = dot3(Rt0, m0);
      dL_dscale.y = dot3(Rt1, m1);
      dL_dscale.z = dot3(Rt2, m2);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        Dm.c[0][k] *= s.x;
        Dm.c[1][k] *= s.y;
        Dm.c[2][k] *= s.z;
      }
      dL_dq[0] = 2 * z * (Dm.c[0][1] - Dm.c[1][0]) + 2 * y * (Dm.c[2][0] - Dm.c[0][2]) + 2 * x * (Dm.c[1][2] - Dm.c[2][1]);
      dL_dq[1] = 2 * y * (Dm.c[1][0] + Dm.c[0][1]) + 2 * z * (Dm.c[2][0] + Dm.c[0][2]) + 2 * r * (Dm.c[1][2] - Dm.c[2][1]) - 4 * x * (Dm.c[2][2] + Dm.c[1][1]);
      dL_dq[2] = 2 * x * (Dm.c[1][0] + Dm.c[0][1]) + 2 * r * (Dm.c[2][0] - Dm.c[0][2]) + 2 * z * (Dm.c[1][2] + Dm.c[2][1]) - 4 * y * (Dm.c[2][2] + Dm.c[0][0]);
      dL_dq[3] = 2 * r * (Dm.c[0][1] - Dm.c[1][0]) + 2 * x * (Dm.c[2][0] + Dm.c[0][2]) + 2 * y * (Dm.c[1][2] + Dm.c[2][1]) - 4 * z * (Dm.c[1][1] + Dm.c[0][0]);
    }
  }

  // ---- write every row (zeros when culled)
  const GsGrads& o = a.out;
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
  if (!in_range) return;
  if (o.dL_dmeans3D) {
    o.dL_dmeans3D[3 * idx] = dL_dmean.x;
    o.dL_dmeans3D[3 * idx + 1] = dL_dmean.y;
    o.dL_dmeans3D[3 * idx + 2] = dL_dmean.z;
  }
  if (o.dL_dmeans2D) {
    o.dL_dmeans2D[3 * idx] = dL_dmean2D_x;
    o.dL_dmeans2D[3 * idx + 1] = dL_dmean2D_y;
    o.dL_dmeans2D[3 * idx + 2] = 0.f;
  }
  if (o.dL_dcolors) {
    o.dL_dcolors[3 * idx] = dL_dcolor.x;
    o.dL_dcolors[3 * idx + 1] = dL_dcolor.y;
    o.dL_dcolors[3 * idx + 2] = dL_dcolor.z;
  }
  if (o.dL_dopacity) o.dL_dopacity[idx] = dL_dop;
  if (o.dL_dextra) o.dL_dextra[idx] = dL_dextra;
  if (o.dL_dcov3D) {
    float2* cd = reinterpret_cast<float2*>(o.dL_dcov3D + (size_t)idx * 6);
    cd[0] = make_float2(dL_dcov[0], dL_dcov[1]);
    cd[1] = make_float2(dL_dcov[2], dL_dcov[3]);
    cd[2] = make_float2(dL_dcov[4], dL_dcov[5]);
  }
  if (a.scales) {
    if (o.dL_dscales) {
      o.dL_dscales[3 * idx] = dL_dscale.x;
      o.dL_dscales[3 * idx + 1] = dL_dscale.y;
      o.dL_dscales[3 * idx + 2] = dL_dscale.z;
    }
    if (o.dL_drotations) reinterpret_cast<float4*>(o.dL_drotations)[idx] = make_float4(dL_dq[0], dL_dq[1], dL_dq[2], dL_dq[3]);
  }
}

int launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((a.P + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, a);
  return 0;
}
