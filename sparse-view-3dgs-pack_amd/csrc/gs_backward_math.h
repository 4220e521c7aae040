// gs_backward_math.h - the per-Gaussian chain rules of the backward pass (SH colour, cov2D / anti-aliasing factor,
// cov3D -> scale / quaternion, projection), as plain inline functions over registers.  Used by gs_preprocess_bwd.hip;
// declared __host__ __device__ so that tests/tools/backward_math_host.hip can run the very same code on the CPU
// against the oracle (tests/test_backward_math_host.py) - no GPU needed to check the algebra.
#pragma once
#include "gs_common.h"
#include "gs_math.h"

struct ShRegsB {
  float f[48];
  __host__ __device__ __forceinline__ V3 operator()(int k) const { return {f[3 * k], f[3 * k + 1], f[3 * k + 2]}; }
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
  __host__ __device__ __forceinline__ void rgb(V3 g) const {
    if (basis_only) {
      p[16] = g.x;
      p[17] = g.y;
      p[18] = g.z;
    }
  }
  __host__ __device__ __forceinline__ void set(int k, float basis, V3 g) const {
    if (basis_only) {
      p[k] = basis;
    } else {
      p[3 * k] = basis * g.x;
      p[3 * k + 1] = basis * g.y;
      p[3 * k + 2] = basis * g.z;
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// The per-Gaussian chain-rule blocks below are written from the math, not from the reference's expansion.  Notation:
//   w0, w1, w2   rows of the world->camera rotation (camera axes in world coordinates)
//   a0, a1       rows of the 2x3 screen Jacobian A = J W:  a0 = j00 w0 + j02 w2,  a1 = j11 w1 + j12 w2
//   Sigma        3D covariance (symmetric), p = Sigma a0, q = Sigma a1
//   cov2D        (a, b, c) = (a0.p, a0.q, a1.q)  (+0.3 on the diagonal)
// the reference's deliberate deviations from plain calculus are kept and marked "quirk".
//
// ROUND 4: the covariance chain runs in FLOAT64.  conic -> cov2D -> cov3D -> (scale, quaternion) (backward.cu:248-275,
// 330-393) is ill-conditioned: dL/dcov2D = -conic Gc conic cancels 3-4 digits for a needle-shaped footprint (det / trace^2
// ~ 6e-4 at the worst Gaussian of the 1 M / 1080p workload), so ANY fp32 evaluation - the reference's formula or a better
// factored one - sits 2e-4 ... 2e-3 of the tensor's largest entry away from the exact image of its own inputs, and a
// 3e-7 run-to-run jitter of the conic sums (float atomics) became 1e-3 on dL_drotations.  The blend backward now accumulates
// its per-tile totals into float64 rows (global_atomic_add_f64: sums of fp32 terms are exact in a 53-bit accumulator unless
// the terms span more than ~2^20 in magnitude, hence independent of the order they arrive in), and the ~300 flop per
// Gaussian with instances of this chain are evaluated in double from the fp32 PARAMETERS (Sigma and cov2D are recomputed,
// not read back rounded): results are the float64 image of the sums, rounded once.  The kernel is HBM-bound and the chain
// runs for a fifth of the Gaussians: no measurable cost.
// ---------------------------------------------------------------------------------------------------------------
struct Sym3 {  // symmetric 3x3, every entry counted once
  float xx, xy, xz, yy, yz, zz;
};
GS_DEV V3 sym_mul(const Sym3& S, V3 v) {
  return {S.xx * v.x + S.xy * v.y + S.xz * v.z, S.xy * v.x + S.yy * v.y + S.yz * v.z, S.xz * v.x + S.yz * v.y + S.zz * v.z};
}
GS_DEV V3 axpby(float a, V3 x, float b, V3 y) { return {a * x.x + b * y.x, a * x.y + b * y.y, a * x.z + b * y.z}; }

struct D3 {
  double x, y, z;
};
struct SymD {  // symmetric 3x3 in double, every entry counted once
  double xx, xy, xz, yy, yz, zz;
};
GS_DEV double ddot(D3 a, D3 b) {
#pragma clang fp contract(fast)
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
GS_DEV D3 dsym_mul(const SymD& S, D3 v) {
#pragma clang fp contract(fast)
  return {S.xx * v.x + S.xy * v.y + S.xz * v.z, S.xy * v.x + S.yy * v.y + S.yz * v.z, S.xz * v.x + S.yz * v.y + S.zz * v.z};
}
GS_DEV D3 daxpby(double a, D3 x, double b, D3 y) {
#pragma clang fp contract(fast)
  return {a * x.x + b * y.x, a * x.y + b * y.y, a * x.z + b * y.z};
}

// Gradient of the real-SH colour w.r.t. the SH coefficients and the view vector (replaces backward.cu:23-142).
// The three colour channels share one basis, so the direction gradient only needs the 16 scalars
// c_k = sh_k . dL_dRGB and the gradient of the scalar polynomial sum_k c_k Y_k(dir): the channel sum is taken BEFORE
// the polynomial derivative (the reference differentiates per channel and sums last, three times the arithmetic).
// Writes the dL_dsh rows it computes through `dsh` and returns the view-direction part of dL_dmean.
template <typename SH>
GS_DEV V3 sh_backward(int deg, V3 pos, V3 campos, const SH& sh, uint32_t clamped, V3 g, const ShSink& dsh) {
  const V3 view = pos - campos;
  const float inv_len = 1.0f / length3(view);
  const V3 d = inv_len * view;
  // a channel clamped to zero in the forward passes no gradient
  if (clamped & 1u) g.x = 0.f;
  if (clamped & 2u) g.y = 0.f;
  if (clamped & 4u) g.z = 0.f;
  dsh.rgb(g);
  dsh.set(0, SH_C0, g);
  V3 grad = {0.f, 0.f, 0.f};  // d/d(dir) of sum_k c_k Y_k
  if (deg > 0) {
    const float x = d.x, y = d.y, z = d.z;
    dsh.set(1, -SH_C1 * y, g);
    dsh.set(2, SH_C1 * z, g);
    dsh.set(3, -SH_C1 * x, g);
    grad = {-SH_C1 * dot3(sh(3), g), -SH_C1 * dot3(sh(1), g), SH_C1 * dot3(sh(2), g)};
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      dsh.set(4, SH_C2_0 * xy, g);
      dsh.set(5, SH_C2_1 * yz, g);
      dsh.set(6, SH_C2_2 * (2.f * zz - xx - yy), g);
      dsh.set(7, SH_C2_3 * xz, g);
      dsh.set(8, SH_C2_4 * (xx - yy), g);
      const float c4 = SH_C2_0 * dot3(sh(4), g), c5 = SH_C2_1 * dot3(sh(5), g), c6 = SH_C2_2 * dot3(sh(6), g);
      const float c7 = SH_C2_3 * dot3(sh(7), g), c8 = SH_C2_4 * dot3(sh(8), g);
      // Y4 = xy, Y5 = yz, Y6 = 2zz - xx - yy, Y7 = xz, Y8 = xx - yy (times their constants)
      grad.x += c4 * y + c7 * z + 2.f * x * (c8 - c6);
      grad.y += c4 * x + c5 * z - 2.f * y * (c8 + c6);
      grad.z += c5 * y + c7 * x + 4.f * z * c6;
      if (deg > 2) {
        const float r11 = 4.f * zz - xx - yy;
        dsh.set(9, SH_C3_0 * y * (3.f * xx - yy), g);
        dsh.set(10, SH_C3_1 * xy * z, g);
        dsh.set(11, SH_C3_2 * y * r11, g);
        dsh.set(12, SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy), g);
        dsh.set(13, SH_C3_4 * x * r11, g);
        dsh.set(14, SH_C3_5 * z * (xx - yy), g);
        dsh.set(15, SH_C3_6 * x * (xx - 3.f * yy), g);
        const float c9 = SH_C3_0 * dot3(sh(9), g), c10 = SH_C3_1 * dot3(sh(10), g), c11 = SH_C3_2 * dot3(sh(11), g);
        const float c12 = SH_C3_3 * dot3(sh(12), g), c13 = SH_C3_4 * dot3(sh(13), g), c14 = SH_C3_5 * dot3(sh(14), g);
        const float c15 = SH_C3_6 * dot3(sh(15), g);
        // Y9 = y(3xx-yy)  Y10 = xyz  Y11 = y(4zz-xx-yy)  Y12 = z(2zz-3xx-3yy)  Y13 = x(4zz-xx-yy)  Y14 = z(xx-yy)
        // Y15 = x(xx-3yy)
        grad.x += c9 * 6.f * xy + c10 * yz - c11 * 2.f * xy - c12 * 6.f * xz + c13 * (4.f * zz - 3.f * xx - yy) +
                  c14 * 2.f * xz + c15 * 3.f * (xx - yy);
        grad.y += c9 * 3.f * (xx - yy) + c10 * xz + c11 * (4.f * zz - xx - 3.f * yy) - c12 * 6.f * yz - c13 * 2.f * xy -
                  c14 * 2.f * yz - c15 * 6.f * xy;
        grad.z += c10 * xy + c11 * 8.f * yz + c12 * 3.f * (2.f * zz - xx - yy) + c13 * 8.f * xz + c14 * (xx - yy);
      }
    }
  }
  // through dir = view / |view|: the component of grad orthogonal to dir, over |view|
  const float along = dot3(d, grad);
  return inv_len * (grad - along * d);
}

// Sigma = R diag(s)^2 R^T in double from the fp32 parameters (forward.cu:114-148 forms it in fp32 and stores it rounded;
// the exact image of the gradient is taken at the parameters, so the chain recomputes it), s = scale_modifier * scale,
// R(q) the rotation matrix of the quaternion q = (r, x, y, z) taken as given (forward.cu:123: no normalisation).
struct RotD {
  D3 c0, c1, c2;  // columns of R
};
GS_DEV RotD quat_to_Rd(V4 quat) {
  const double r = quat.x, x = quat.y, y = quat.z, z = quat.w;
  RotD R;
  R.c0 = {1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y + r * z), 2.0 * (x * z - r * y)};
  R.c1 = {2.0 * (x * y - r * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z + r * x)};
  R.c2 = {2.0 * (x * z + r * y), 2.0 * (y * z - r * x), 1.0 - 2.0 * (x * x + y * y)};
  return R;
}
GS_DEV SymD sigma_from_scale_rot(D3 s, const RotD& R) {
#pragma clang fp contract(fast)
  const double s0 = s.x * s.x, s1 = s.y * s.y, s2 = s.z * s.z;
  return {s0 * R.c0.x * R.c0.x + s1 * R.c1.x * R.c1.x + s2 * R.c2.x * R.c2.x,
          s0 * R.c0.x * R.c0.y + s1 * R.c1.x * R.c1.y + s2 * R.c2.x * R.c2.y,
          s0 * R.c0.x * R.c0.z + s1 * R.c1.x * R.c1.z + s2 * R.c2.x * R.c2.z,
          s0 * R.c0.y * R.c0.y + s1 * R.c1.y * R.c1.y + s2 * R.c2.y * R.c2.y,
          s0 * R.c0.y * R.c0.z + s1 * R.c1.y * R.c1.z + s2 * R.c2.y * R.c2.z,
          s0 * R.c0.z * R.c0.z + s1 * R.c1.z * R.c1.z + s2 * R.c2.z * R.c2.z};
}

// Backward of cov2D = A Sigma A^T + 0.3 I and of the anti-aliasing opacity factor (replaces backward.cu:147-326): from
// the conic gradient (per matrix entry: the blend stage accumulates HALF the derivative w.r.t. the off-diagonal
// coefficient) to G = dL/dSigma (symmetric, per entry) and dL/dmean through the Jacobian's dependence on t.  All in double.
struct Cov2DBack {
  SymD G;
  V3 dmean;
  float dop;  // dL_dopacity after the anti-aliasing factor
};
GS_DEV Cov2DBack cov2d_backward(V3 mean, const SymD& Sg, const float* vm, float focal_x, float focal_y, float tan_fovx,
                                float tan_fovy, D3 g_conic /* xx, xy, yy */, float g_opacity, float opacity_raw,
                                bool antialiasing, double g_invdepth /* dL/d(1/t.z), 0 if unused */) {
#pragma clang fp contract(fast)  // (the file is built without contraction for the fp32 bit-exactness rules; the double chain may fuse)
  Cov2DBack o;
  const double fx = focal_x, fy = focal_y;
  // view-space mean, clamped exactly as the forward clamps it (forward.cu:81-87); a clamped coordinate gets no gradient
  const double mx = mean.x, my = mean.y, mz = mean.z;
  D3 t = {vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12], vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13],
          vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14]};
  const double limx = 1.3 * (double)tan_fovx, limy = 1.3 * (double)tan_fovy;
  const double iz = 1.0 / t.z, iz2 = iz * iz;
  const double rx = t.x * iz, ry = t.y * iz;
  const bool free_x = !(rx < -limx || rx > limx), free_y = !(ry < -limy || ry > limy);
  t.x = fmin(limx, fmax(-limx, rx)) * t.z;
  t.y = fmin(limy, fmax(-limy, ry)) * t.z;
  const double j00 = fx * iz, j11 = fy * iz, j02 = -(fx * t.x) * iz2, j12 = -(fy * t.y) * iz2;
  const D3 w0 = {vm[0], vm[4], vm[8]}, w1 = {vm[1], vm[5], vm[9]}, w2 = {vm[2], vm[6], vm[10]};
  const D3 a0 = daxpby(j00, w0, j02, w2), a1 = daxpby(j11, w1, j12, w2);
  const D3 p = dsym_mul(Sg, a0), q = dsym_mul(Sg, a1);
  double a = ddot(a0, p), b = ddot(a0, q), c = ddot(a1, q);
  const double lowpass = 0.3;
  double ga = 0.0, gb = 0.0, gc = 0.0;  // dL/da, dL/db (b as ONE scalar filling both off-diagonal entries), dL/dc
  o.dop = g_opacity;
  if (antialiasing) {
    // opacity' = opacity * sqrt(max(2.5e-5, det(cov) / det(cov + 0.3 I))), forward.cu:228-234
    const double det0 = a * c - b * b;
    a += lowpass;
    c += lowpass;
    const double det1 = a * c - b * b;
    const double ratio = det0 / det1;
    const double h = sqrt(fmax(0.000025, ratio));
    o.dop = (float)((double)g_opacity * h);
    const double g_ratio = ratio <= 0.000025 ? 0.0 : ((double)g_opacity * (double)opacity_raw) / (2.0 * h);
    // quirk (backward.cu:235-245): d(ratio)/d(a, b, c) in the closed form that is exact at the UN-shifted diagonal, but
    // evaluated at the shifted one.  With k = 0.3:  d/da = k (c^2 + k c + b^2) / D^2,  d/db = -2 k b (a + c + k) / D^2,
    // D = det + k (a + c) + k^2
    const double D = lowpass * lowpass + lowpass * (a + c) + a * c - b * b;
    const double f = g_ratio / (D * D);
    ga = lowpass * (lowpass * c + c * c + b * b) * f;
    gc = lowpass * (lowpass * a + a * a + b * b) * f;
    gb = -2.0 * lowpass * b * (lowpass + a + c) * f;
  } else {
    a += lowpass;
    c += lowpass;
  }
  // conic = inverse of [[a, b], [b, c]]:  dL/dcov2D = -conic Gc conic = -(adj Gc adj) / det^2,  adj = [[c, -b], [-b, a]];
  // 1/det^2 carries the reference's regulariser (backward.cu:252)
  const double det = a * c - b * b;
  const double k = 1.0 / (det * det + 0.0000001);
  // quirk (backward.cu:253): the reference leaves dL_dcov3D zero when ITS fp32 reciprocal underflows (det^2 = inf)
  const float detf = (float)det;
  const bool conic_part = (1.0f / (detf * detf + 0.0000001f)) != 0.f;
  if (conic_part) {
    const double x00 = g_conic.x * c - g_conic.y * b, x01 = g_conic.y * a - g_conic.x * b;  // Gc adj
    const double x10 = g_conic.y * c - g_conic.z * b, x11 = g_conic.z * a - g_conic.y * b;
    ga -= k * (c * x00 - b * x10);
    gc -= k * (a * x11 - b * x01);
    gb -= 2.0 * k * (c * x01 - b * x11);
  }
  // (the mean gradient first: p and q die here, before the six entries of G come alive - the kernel has 128 VGPRs)
  // rows of A:  dL/da0 = 2 ga p + gb q,  dL/da1 = gb p + 2 gc q;  A = J W gives the four non-zero entries of J
  {
    const D3 dA0 = daxpby(2.0 * ga, p, gb, q), dA1 = daxpby(gb, p, 2.0 * gc, q);
    const double dj00 = ddot(w0, dA0), dj02 = ddot(w2, dA0), dj11 = ddot(w1, dA1), dj12 = ddot(w2, dA1);
    // J = [[fx/z, 0, -fx x/z^2], [0, fy/z, -fy y/z^2]];  quirk (backward.cu:310-313): in d/dz the clamped x, y are treated
    // as constants although they were formed as (clamped ratio) * z
    const double iz3 = iz2 * iz;
    const double dtx = free_x ? -fx * iz2 * dj02 : 0.0;
    const double dty = free_y ? -fy * iz2 * dj12 : 0.0;
    const double dtz = -fx * iz2 * dj00 - fy * iz2 * dj11 + 2.0 * fx * t.x * iz3 * dj02 + 2.0 * fy * t.y * iz3 * dj12 -
                       g_invdepth * iz2;
    o.dmean = {(float)(dtx * w0.x + dty * w1.x + dtz * w2.x), (float)(dtx * w0.y + dty * w1.y + dtz * w2.y),
               (float)(dtx * w0.z + dty * w1.z + dtz * w2.z)};
  }
  // dL/dSigma = ga a0 a0^T + gb/2 (a0 a1^T + a1 a0^T) + gc a1 a1^T = a0 u^T + a1 v^T
  if (conic_part) {
    const double hb = 0.5 * gb;
    const D3 u = daxpby(ga, a0, hb, a1), v = daxpby(hb, a0, gc, a1);
    o.G = {a0.x * u.x + a1.x * v.x, a0.x * u.y + a1.x * v.y, a0.x * u.z + a1.x * v.z,
           a0.y * u.y + a1.y * v.y, a0.y * u.z + a1.y * v.z, a0.z * u.z + a1.z * v.z};
  } else {
    o.G = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  }
  return o;
}

// Backward of Sigma = R diag(s)^2 R^T (replaces backward.cu:330-393), in double.
// With r_j the j-th column of R:  Sigma = sum_j s_j^2 r_j r_j^T  =>  dL/ds_j = 2 s_j r_j^T G r_j  and
// D = dL/dR = 2 G R diag(s)^2; the quaternion gradient contracts D with dR/dq.
// quirk: dL/ds is returned w.r.t. s itself, without the factor scale_modifier (backward.cu:372-375) - the same thing on
// the training path (scale_modifier = 1); tests/test_oracle_golden.py pins it against the reference's python autograd.
GS_DEV void cov3d_backward(const SymD& G, D3 s, const RotD& R, V4 quat, V3& dscale, float dq[4]) {
#pragma clang fp contract(fast)
  const double r = quat.x, x = quat.y, y = quat.z, z = quat.w;
  const D3 k0 = dsym_mul(G, R.c0), k1 = dsym_mul(G, R.c1), k2 = dsym_mul(G, R.c2);
  dscale = {(float)(2.0 * s.x * ddot(R.c0, k0)), (float)(2.0 * s.y * ddot(R.c1, k1)), (float)(2.0 * s.z * ddot(R.c2, k2))};
  const double e0 = 2.0 * s.x * s.x, e1 = 2.0 * s.y * s.y, e2 = 2.0 * s.z * s.z;
  // columns of D; R00 = 1-2(yy+zz) R01 = 2(xy-rz) R02 = 2(xz+ry) | R10 = 2(xy+rz) R11 = 1-2(xx+zz) R12 = 2(yz-rx) |
  // R20 = 2(xz-ry) R21 = 2(yz+rx) R22 = 1-2(xx+yy);  D_ij = (column j).(component i)
  const double D00 = e0 * k0.x, D10 = e0 * k0.y, D20 = e0 * k0.z, D01 = e1 * k1.x, D11 = e1 * k1.y, D21 = e1 * k1.z;
  const double D02 = e2 * k2.x, D12 = e2 * k2.y, D22 = e2 * k2.z;
  dq[0] = (float)(2.0 * (z * (D10 - D01) + y * (D02 - D20) + x * (D21 - D12)));
  dq[1] = (float)(2.0 * (y * (D01 + D10) + z * (D02 + D20) + r * (D21 - D12)) - 4.0 * x * (D11 + D22));
  dq[2] = (float)(2.0 * (x * (D01 + D10) + r * (D02 - D20) + z * (D12 + D21)) - 4.0 * y * (D00 + D22));
  dq[3] = (float)(2.0 * (r * (D10 - D01) + x * (D02 + D20) + y * (D12 + D21)) - 4.0 * z * (D00 + D11));
}

// Projection part of the mean gradient (replaces backward.cu:419-440): pixel mean = ndc2Pix((PV m).xy / ((PV m).w + 1e-7));
// the blend stage has already folded d(pixel)/d(ndc) = W/2, H/2 into (gx, gy).
GS_DEV V3 projection_backward(V3 m, const float* proj, float gx, float gy) {
  const V4 h = xform4x4(m, proj);
  const float iw = 1.0f / (h.w + 0.0000001f);
  // d(ndc.x)/dm_i = (P_0i - ndc.x P_3i) / w,  ndc = h.xy / w
  const float nx = h.x * iw * iw, ny = h.y * iw * iw;
  return {(proj[0] * iw - proj[3] * nx) * gx + (proj[1] * iw - proj[3] * ny) * gy,
          (proj[4] * iw - proj[7] * nx) * gx + (proj[5] * iw - proj[7] * ny) * gy,
          (proj[8] * iw - proj[11] * nx) * gx + (proj[9] * iw - proj[11] * ny) * gy};
}


// ---------------------------------------------------------------------------------------------------------------
// One Gaussian of the backward per-Gaussian stage.  It runs in two kernels (gs_preprocess_bwd.hip):
//   chain_kernel            reads the Gaussian's float64 row of blend sums, runs the float64 covariance chain and leaves
//                           a 20-float RECORD of fp32 results (layout GC_*); ~170 VGPRs, only Gaussians with instances
//   preprocess_bwd*_kernel  the streaming kernels: projection part of the mean gradient, SH backward, and - fused step -
//                           activation backward, statistics, Adam; fp32 only, 5 waves per SIMD
// (Inlined into the streaming kernels the double chain held them at 128 VGPRs with 36-58 spilled, or at 3 waves per SIMD:
// +29 ... +62 us per C3 step; as its own launch it costs ~10 us and cleans the rows on the way.)
// ---------------------------------------------------------------------------------------------------------------
struct GeomBack {
  V3 dmean;                  // (cov2D part) + (projection part) [+ depth part of the FSGS generation]
  float dmean2D_x, dmean2D_y;
  float dcov[6];             // the reference's 6-vector: an off-diagonal entry counts for both places
  float dop, dextra;
  V3 dcolor;
  V3 dscale;
  float dq[4];
};
// chain_kernel, one Gaussian: idx must be visible (radii > 0) and - when skip_uninstanced - have instances; reads its row
// of blend-backward sums (float64, layout GR_* of gs_common.h) and fills rec[GC_STRIDE]
GS_DEV void chain_from_row(const PreprocessBwdArgs& a, int idx, const gs_row_t* row, float* rec) {
  const double2* gr = reinterpret_cast<const double2*>(row);
  const double2 g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3], g4 = gr[4], g5 = gr[5];
  // [mx my] [cxx cxy] [cyy op] [r g] [b id] [extra -]
  const V3 mean = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
  // Sigma in double: from the parameters when they are there, else the caller's precomputed covariance as given
  SymD Sg;
  RotD R = {};
  D3 s = {0.0, 0.0, 0.0};
  V4 rq = {1.f, 0.f, 0.f, 0.f};
  if (a.scales) {
    const V3 scl = load_scales(a.scales, idx, a.raw_activations);
    rq = load_rotation(a.rotations, idx, a.raw_activations);
    const double mod = a.scale_modifier;
    s = {mod * scl.x, mod * scl.y, mod * scl.z};
    R = quat_to_Rd(rq);
    Sg = sigma_from_scale_rot(s, R);
  } else {
    const float2* cv = reinterpret_cast<const float2*>(a.cov3D + 6 * (size_t)idx);
    const float2 v0 = cv[0], v1 = cv[1], v2 = cv[2];
    Sg = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y};
  }
  // slot GR_ID holds dL/d(inverse depth) (dr_aa) or dL/d(depth) (FSGS generation); only the former enters through t.z
  const Cov2DBack cb = cov2d_backward(mean, Sg, a.viewmatrix, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy,
                                      {g1.x, g1.y, g2.x}, (float)g2.y,
                                      a.antialiasing ? load_opacity(a.opacities, idx, a.raw_activations) : 0.f,
                                      a.antialiasing != 0, a.has_invdepth == 1 ? g4.y : 0.0);
  V3 dmean = cb.dmean;
  if (a.has_invdepth == 2) {
    // FSGS generation (-confidence fork, backward.cu:394-403): depth = (row 2 of the view matrix).(m, 1), divided by
    // the homogeneous row as that fork writes it
    const float* vm = a.viewmatrix;
    const float zc = vm[2] * mean.x + vm[6] * mean.y + vm[10] * mean.z + vm[14];
    const float gd = (float)g4.y;
    dmean = dmean + V3{(vm[2] - vm[3] * zc) * gd, (vm[6] - vm[7] * zc) * gd, (vm[10] - vm[11] * zc) * gd};
  }
  float c[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (a.scales) {
    V3 dscale;
    cov3d_backward(cb.G, s, R, rq, dscale, c + 3);
    c[0] = dscale.x; c[1] = dscale.y; c[2] = dscale.z;
  } else {  // precomputed covariance: its gradient takes the seven slots instead
    c[0] = (float)cb.G.xx; c[1] = (float)(2.0 * cb.G.xy); c[2] = (float)(2.0 * cb.G.xz);
    c[3] = (float)cb.G.yy; c[4] = (float)(2.0 * cb.G.yz); c[5] = (float)cb.G.zz;
  }
  float4* o = reinterpret_cast<float4*>(rec);
  o[0] = make_float4((float)g0.x, (float)g0.y, dmean.x, dmean.y);
  o[1] = make_float4(dmean.z, cb.dop, c[0], c[1]);
  o[2] = make_float4(c[2], c[3], c[4], c[5]);
  o[3] = make_float4(c[6], (float)g3.x, (float)g3.y, (float)g4.x);
  o[4] = make_float4((float)g5.x, 0.f, 0.f, 0.f);
}

// streaming kernels, one Gaussian (same precondition): from its record to everything but the SH half
GS_DEV void geometry_backward(const PreprocessBwdArgs& a, int idx, GeomBack& o) {
  const float4* rc = reinterpret_cast<const float4*>(a.grad_recs + (size_t)idx * GC_STRIDE);
  const float4 r0 = rc[0], r1 = rc[1], r2 = rc[2], r3 = rc[3];
  o.dmean2D_x = r0.x;
  o.dmean2D_y = r0.y;
  o.dop = r1.y;
  o.dcolor = {r3.y, r3.z, r3.w};
  o.dextra = a.has_extra ? a.grad_recs[(size_t)idx * GC_STRIDE + GC_EXTRA] : 0.f;
  const V3 mean = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
  o.dmean = V3{r0.z, r0.w, r1.x} + projection_backward(mean, a.projmatrix, o.dmean2D_x, o.dmean2D_y);
  o.dscale = {0.f, 0.f, 0.f};
  o.dq[0] = o.dq[1] = o.dq[2] = o.dq[3] = 0.f;
  o.dcov[0] = o.dcov[1] = o.dcov[2] = o.dcov[3] = o.dcov[4] = o.dcov[5] = 0.f;
  if (a.scales) {
    o.dscale = {r1.z, r1.w, r2.x};
    o.dq[0] = r2.y; o.dq[1] = r2.z; o.dq[2] = r2.w; o.dq[3] = r3.x;
  } else {
    o.dcov[0] = r1.z; o.dcov[1] = r1.w; o.dcov[2] = r2.x; o.dcov[3] = r2.y; o.dcov[4] = r2.z; o.dcov[5] = r2.w;
  }
}

// SH half: writes the Gaussian's dL_dsh row through `dsh`, returns the view-direction part of dL_dmean
GS_DEV V3 sh_backward_row(const PreprocessBwdArgs& a, int idx, V3 dL_dcolor, const ShSink& dsh) {
  const V3 mean = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
  const uint32_t clamped = a.splat[idx].clamped;
  const V3 campos = {a.campos[0], a.campos[1], a.campos[2]};
  if (a.M == 16 && a.shs_rest) {  // split rows, as preprocess_fwd reads them
    ShRegsB sh;
    const float* dc = a.shs + (size_t)idx * 3;
    const float* rest = a.shs_rest + (size_t)idx * 45;
    const int nfl = 3 * (a.D + 1) * (a.D + 1) - 3;
    sh.f[0] = dc[0]; sh.f[1] = dc[1]; sh.f[2] = dc[2];
#pragma unroll
    for (int k = 0; k < 45; k++) sh.f[3 + k] = k < nfl ? rest[k] : 0.f;
    return sh_backward(a.D, mean, campos, sh, clamped, dL_dcolor, dsh);
  }
  if (a.M == 16) {
    ShRegsB sh;
    const float4* src = reinterpret_cast<const float4*>(a.shs + (size_t)idx * 48);
    const int nvec = a.D == 0 ? 1 : (a.D == 1 ? 3 : (a.D == 2 ? 7 : 12));  // float4s that hold the active degree
#pragma unroll
    for (int k = 0; k < 12; k++) {
      if (k < nvec) {
        const float4 v = src[k];
        sh.f[4 * k] = v.x; sh.f[4 * k + 1] = v.y; sh.f[4 * k + 2] = v.z; sh.f[4 * k + 3] = v.w;
      } else {
        sh.f[4 * k] = sh.f[4 * k + 1] = sh.f[4 * k + 2] = sh.f[4 * k + 3] = 0.f;
      }
    }
    return sh_backward(a.D, mean, campos, sh, clamped, dL_dcolor, dsh);
  }
  struct ShMemB {
    const float* p;
    __host__ __device__ __forceinline__ V3 operator()(int k) const { return {p[3 * k], p[3 * k + 1], p[3 * k + 2]}; }
  } sh{a.shs + (size_t)idx * a.M * 3};
  if (a.shs_rest) {
    struct ShMemSplitB {
      const float *dc, *rest;
      __host__ __device__ __forceinline__ V3 operator()(int k) const {
        const float* p = k == 0 ? dc : rest + 3 * (k - 1);
        return {p[0], p[1], p[2]};
      }
    } shs{a.shs + (size_t)idx * 3, a.shs_rest + (size_t)idx * (a.M - 1) * 3};
    return sh_backward(a.D, mean, campos, shs, clamped, dL_dcolor, dsh);
  }
  return sh_backward(a.D, mean, campos, sh, clamped, dL_dcolor, dsh);
}
