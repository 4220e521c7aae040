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
// Values agree with backward.cu to fp32 rounding (the parity tests compare against the oracle's transcription of it);
// the reference's deliberate deviations from plain calculus are kept and marked "quirk".
// ---------------------------------------------------------------------------------------------------------------
struct Sym3 {  // symmetric 3x3, every entry counted once
  float xx, xy, xz, yy, yz, zz;
};
GS_DEV V3 sym_mul(const Sym3& S, V3 v) {
  return {S.xx * v.x + S.xy * v.y + S.xz * v.z, S.xy * v.x + S.yy * v.y + S.yz * v.z, S.xz * v.x + S.yz * v.y + S.zz * v.z};
}
GS_DEV V3 axpby(float a, V3 x, float b, V3 y) { return {a * x.x + b * y.x, a * x.y + b * y.y, a * x.z + b * y.z}; }

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

// Backward of cov2D = A Sigma A^T + 0.3 I and of the anti-aliasing opacity factor (replaces backward.cu:147-326): from
// the conic gradient (per matrix entry: the blend stage accumulates HALF the derivative w.r.t. the off-diagonal
// coefficient) to G = dL/dSigma (symmetric, per entry) and dL/dmean through the Jacobian's dependence on t.
struct Cov2DBack {
  Sym3 G;
  V3 dmean;
  float dop;  // dL_dopacity after the anti-aliasing factor
};
GS_DEV Cov2DBack cov2d_backward(V3 mean, const float* c3, const float* vm, float fx, float fy, float tan_fovx,
                                float tan_fovy, V3 g_conic /* xx, xy, yy */, float g_opacity, float opacity_raw,
                                bool antialiasing, float g_invdepth /* dL/d(1/t.z), 0 if unused */) {
  Cov2DBack o;
  // view-space mean, clamped exactly as the forward clamps it (forward.cu:81-87); a clamped coordinate gets no gradient
  V3 t = xform4x3(mean, vm);
  const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
  const float rx = t.x / t.z, ry = t.y / t.z;
  const bool free_x = !(rx < -limx || rx > limx), free_y = !(ry < -limy || ry > limy);
  t.x = fminf(limx, fmaxf(-limx, rx)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, ry)) * t.z;
  const float iz = 1.0f / t.z, iz2 = iz * iz;
  const float j00 = fx * iz, j11 = fy * iz, j02 = -(fx * t.x) * iz2, j12 = -(fy * t.y) * iz2;
  const V3 w0 = {vm[0], vm[4], vm[8]}, w1 = {vm[1], vm[5], vm[9]}, w2 = {vm[2], vm[6], vm[10]};
  const V3 a0 = axpby(j00, w0, j02, w2), a1 = axpby(j11, w1, j12, w2);
  const Sym3 Sg = {c3[0], c3[1], c3[2], c3[3], c3[4], c3[5]};
  const V3 p = sym_mul(Sg, a0), q = sym_mul(Sg, a1);
  float a = dot3(a0, p), b = dot3(a0, q), c = dot3(a1, q);
  const float lowpass = 0.3f;
  float ga = 0.f, gb = 0.f, gc = 0.f;  // dL/da, dL/db (b as ONE scalar filling both off-diagonal entries), dL/dc
  o.dop = g_opacity;
  if (antialiasing) {
    // opacity' = opacity * sqrt(max(2.5e-5, det(cov) / det(cov + 0.3 I))), forward.cu:228-234
    const float det0 = a * c - b * b;
    a += lowpass;
    c += lowpass;
    const float det1 = a * c - b * b;
    const float ratio = det0 / det1;
    const float h = sqrtf(fmaxf(0.000025f, ratio));
    o.dop = g_opacity * h;
    const float g_ratio = ratio <= 0.000025f ? 0.f : (g_opacity * opacity_raw) / (2.f * h);
    // quirk (backward.cu:235-245): d(ratio)/d(a, b, c) in the closed form that is exact at the UN-shifted diagonal, but
    // evaluated at the shifted one.  With k = 0.3:  d/da = k (c^2 + k c + b^2) / D^2,  d/db = -2 k b (a + c + k) / D^2,
    // D = det + k (a + c) + k^2
    const float D = lowpass * lowpass + lowpass * (a + c) + a * c - b * b;
    const float f = g_ratio / (D * D);
    ga = lowpass * (lowpass * c + c * c + b * b) * f;
    gc = lowpass * (lowpass * a + a * a + b * b) * f;
    gb = -2.f * lowpass * b * (lowpass + a + c) * f;
  } else {
    a += lowpass;
    c += lowpass;
  }
  // conic = inverse of [[a, b], [b, c]]:  dL/dcov2D = -conic Gc conic = -(adj Gc adj) / det^2,  adj = [[c, -b], [-b, a]];
  // 1/det^2 carries the reference's regulariser (backward.cu:252)
  const float det = a * c - b * b;
  const float k = 1.0f / (det * det + 0.0000001f);
  const bool conic_part = k != 0.f;
  if (conic_part) {
    const float x00 = g_conic.x * c - g_conic.y * b, x01 = g_conic.y * a - g_conic.x * b;  // Gc adj
    const float x10 = g_conic.y * c - g_conic.z * b, x11 = g_conic.z * a - g_conic.y * b;
    ga -= k * (c * x00 - b * x10);
    gc -= k * (a * x11 - b * x01);
    gb -= 2.f * k * (c * x01 - b * x11);
  }
  // dL/dSigma = ga a0 a0^T + gb/2 (a0 a1^T + a1 a0^T) + gc a1 a1^T = a0 u^T + a1 v^T
  const float hb = 0.5f * gb;
  const V3 u = axpby(ga, a0, hb, a1), v = axpby(hb, a0, gc, a1);
  if (conic_part) {  // (the reference leaves dL_dcov3D zero when the regulariser's reciprocal underflows, backward.cu:253)
    o.G = {a0.x * u.x + a1.x * v.x, a0.x * u.y + a1.x * v.y, a0.x * u.z + a1.x * v.z,
           a0.y * u.y + a1.y * v.y, a0.y * u.z + a1.y * v.z, a0.z * u.z + a1.z * v.z};
  } else {
    o.G = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  }
  // rows of A:  dL/da0 = 2 ga p + gb q,  dL/da1 = gb p + 2 gc q;  A = J W gives the four non-zero entries of J
  const V3 dA0 = axpby(2.f * ga, p, gb, q), dA1 = axpby(gb, p, 2.f * gc, q);
  const float dj00 = dot3(w0, dA0), dj02 = dot3(w2, dA0), dj11 = dot3(w1, dA1), dj12 = dot3(w2, dA1);
  // J = [[fx/z, 0, -fx x/z^2], [0, fy/z, -fy y/z^2]];  quirk (backward.cu:310-313): in d/dz the clamped x, y are treated
  // as constants although they were formed as (clamped ratio) * z
  const float iz3 = iz2 * iz;
  const float dtx = free_x ? -fx * iz2 * dj02 : 0.f;
  const float dty = free_y ? -fy * iz2 * dj12 : 0.f;
  const float dtz = -fx * iz2 * dj00 - fy * iz2 * dj11 + 2.f * fx * t.x * iz3 * dj02 + 2.f * fy * t.y * iz3 * dj12 -
                    g_invdepth * iz2;
  o.dmean = {dtx * w0.x + dty * w1.x + dtz * w2.x, dtx * w0.y + dty * w1.y + dtz * w2.y, dtx * w0.z + dty * w1.z + dtz * w2.z};
  return o;
}

// Backward of Sigma = R diag(s)^2 R^T (replaces backward.cu:330-393), s = scale_modifier * scale, R(q) the rotation of
// the quaternion q = (r, x, y, z) taken as given (no normalisation Jacobian: the caller normalises, forward.cu:123).
// With r_j the j-th column of R:  Sigma = sum_j s_j^2 r_j r_j^T  =>  dL/ds_j = 2 s_j r_j^T G r_j  and
// D = dL/dR = 2 G R diag(s)^2; the quaternion gradient contracts D with dR/dq.
// quirk: dL/ds is returned w.r.t. s itself, without the factor scale_modifier (backward.cu:372-375) - the same thing on
// the training path (scale_modifier = 1); tests/test_oracle_golden.py pins it against the reference's python autograd.
GS_DEV void cov3d_backward(const Sym3& G, V3 scale, float scale_modifier, V4 quat, V3& dscale, float dq[4]) {
  const float r = quat.x, x = quat.y, y = quat.z, z = quat.w;
  const V3 s = scale_modifier * scale;
  const V3 c0 = {1.f - 2.f * (y * y + z * z), 2.f * (x * y + r * z), 2.f * (x * z - r * y)};  // columns of R
  const V3 c1 = {2.f * (x * y - r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + r * x)};
  const V3 c2 = {2.f * (x * z + r * y), 2.f * (y * z - r * x), 1.f - 2.f * (x * x + y * y)};
  const V3 k0 = sym_mul(G, c0), k1 = sym_mul(G, c1), k2 = sym_mul(G, c2);
  dscale = {2.f * s.x * dot3(c0, k0), 2.f * s.y * dot3(c1, k1), 2.f * s.z * dot3(c2, k2)};
  const V3 d0 = (2.f * s.x * s.x) * k0, d1 = (2.f * s.y * s.y) * k1, d2 = (2.f * s.z * s.z) * k2;  // columns of D
  // R00 = 1-2(yy+zz) R01 = 2(xy-rz) R02 = 2(xz+ry) | R10 = 2(xy+rz) R11 = 1-2(xx+zz) R12 = 2(yz-rx) |
  // R20 = 2(xz-ry) R21 = 2(yz+rx) R22 = 1-2(xx+yy);  D_ij = (column j).(component i)
  const float D00 = d0.x, D10 = d0.y, D20 = d0.z, D01 = d1.x, D11 = d1.y, D21 = d1.z, D02 = d2.x, D12 = d2.y, D22 = d2.z;
  dq[0] = 2.f * (z * (D10 - D01) + y * (D02 - D20) + x * (D21 - D12));
  dq[1] = 2.f * (y * (D01 + D10) + z * (D02 + D20) + r * (D21 - D12)) - 4.f * x * (D11 + D22);
  dq[2] = 2.f * (x * (D01 + D10) + r * (D02 - D20) + z * (D12 + D21)) - 4.f * y * (D00 + D22);
  dq[3] = 2.f * (r * (D10 - D01) + x * (D02 + D20) + y * (D12 + D21)) - 4.f * z * (D00 + D11);
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
// One Gaussian of the backward per-Gaussian stage, in the two halves the kernel runs (the geometry outputs are stored
// before the 48 SH coefficients are loaded, which keeps the kernel at 86 VGPRs / 5 waves per SIMD).
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
// idx must be a visible Gaussian (radii > 0); reads its 16-float row of blend-backward sums (layout: GR_* of gs_common.h)
GS_DEV void geometry_backward(const PreprocessBwdArgs& a, int idx, GeomBack& o) {
  const float4* gr = reinterpret_cast<const float4*>(a.grad_rows + (size_t)idx * GR_STRIDE);
  const float4 g0 = gr[0], g1 = gr[1], g2 = gr[2];
  o.dmean2D_x = g0.x;
  o.dmean2D_y = g0.y;
  o.dcolor = {g1.z, g1.w, g2.x};
  o.dextra = g2.z;
  const V3 mean = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
  float c3[6];
  {
    const float2* cv = reinterpret_cast<const float2*>(a.cov3D + 6 * (size_t)idx);
    const float2 v0 = cv[0], v1 = cv[1], v2 = cv[2];
    c3[0] = v0.x; c3[1] = v0.y; c3[2] = v1.x; c3[3] = v1.y; c3[4] = v2.x; c3[5] = v2.y;
  }
  // slot GR_ID holds dL/d(inverse depth) (dr_aa) or dL/d(depth) (FSGS generation); only the former enters through t.z
  const Cov2DBack cb = cov2d_backward(mean, c3, a.viewmatrix, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy,
                                      {g0.z, g0.w, g1.x}, g1.y, a.antialiasing ? load_opacity(a.opacities, idx, a.raw_activations) : 0.f,
                                      a.antialiasing != 0, a.has_invdepth == 1 ? g2.y : 0.f);
  o.dop = cb.dop;
  o.dcov[0] = cb.G.xx; o.dcov[1] = 2.f * cb.G.xy; o.dcov[2] = 2.f * cb.G.xz;
  o.dcov[3] = cb.G.yy; o.dcov[4] = 2.f * cb.G.yz; o.dcov[5] = cb.G.zz;
  o.dmean = cb.dmean + projection_backward(mean, a.projmatrix, o.dmean2D_x, o.dmean2D_y);
  if (a.has_invdepth == 2) {
    // FSGS generation (-confidence fork, backward.cu:394-403): depth = (row 2 of the view matrix).(m, 1), divided by
    // the homogeneous row as that fork writes it
    const float* vm = a.viewmatrix;
    const float zc = vm[2] * mean.x + vm[6] * mean.y + vm[10] * mean.z + vm[14];
    const float gd = g2.y;
    o.dmean = o.dmean + V3{(vm[2] - vm[3] * zc) * gd, (vm[6] - vm[7] * zc) * gd, (vm[10] - vm[11] * zc) * gd};
  }
  o.dscale = {0.f, 0.f, 0.f};
  o.dq[0] = o.dq[1] = o.dq[2] = o.dq[3] = 0.f;
  if (a.scales) {
    const V3 scl = load_scales(a.scales, idx, a.raw_activations);
    const V4 rq = load_rotation(a.rotations, idx, a.raw_activations);
    cov3d_backward(cb.G, scl, a.scale_modifier, rq, o.dscale, o.dq);
  }
}

// SH half: writes the Gaussian's dL_dsh row through `dsh`, returns the view-direction part of dL_dmean
GS_DEV V3 sh_backward_row(const PreprocessBwdArgs& a, int idx, V3 dL_dcolor, const ShSink& dsh) {
  const V3 mean = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
  const uint32_t clamped = a.splat[idx].clamped;
  const V3 campos = {a.campos[0], a.campos[1], a.campos[2]};
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
  return sh_backward(a.D, mean, campos, sh, clamped, dL_dcolor, dsh);
}
