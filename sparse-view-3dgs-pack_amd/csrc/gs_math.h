// gs_math.h - small fp32 device helpers for the per-Gaussian kernels.
//
// The per-Gaussian kernels (preprocess fwd/bwd, kNN) are compiled with -ffp-contract=off and spell
// every sum out in a fixed order, so their integer outputs (radius, tile rectangle, depth key bits,
// clamp flags) are a pure function of the inputs that the CPU oracle reproduces bit for bit.
// 3x3 products use the column-major accumulation order of the reference's matrix library
// (Result[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GS_DEV static __host__ __device__ __forceinline__

struct V3 {
  float x, y, z;
};
struct V4 {
  float x, y, z, w;
};
GS_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
GS_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
GS_DEV V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
GS_DEV V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
GS_DEV V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
GS_DEV float dot3(V3 a, V3 b) {
  float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
  return tx + ty + tz;
}
GS_DEV float length3(V3 a) { return sqrtf(dot3(a, a)); }

struct M3 {
  float c[3][3];  // c[col][row]
};
GS_DEV M3 mat3_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
  M3 m;
  m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
  m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
  m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2;
  return m;
}
GS_DEV M3 mul3(const M3& A, const M3& B) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.c[c][r] = A.c[0][r] * B.c[c][0] + A.c[1][r] * B.c[c][1] + A.c[2][r] * B.c[c][2];
  return R;
}
GS_DEV M3 transpose3(const M3& A) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.c[c][r] = A.c[r][c];
  return R;
}
GS_DEV M3 scale3(float s, const M3& A) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.c[c][r] = A.c[c][r] * s;
  return R;
}

// float -> int with truncation, saturation and NaN -> 0 (what v_cvt_i32_f32 does; spelled out so the
// rule is the same on every path)
GS_DEV int f2i_sat(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}
GS_DEV uint32_t f2u_sat(float f) {
  if (f != f) return 0u;
  if (f >= 4294967296.0f) return 0xFFFFFFFFu;
  if (f <= 0.0f) return 0u;
  return (uint32_t)f;
}

// auxiliary.h:70-109
GS_DEV V3 xform4x3(V3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
GS_DEV V4 xform4x4(V3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
GS_DEV V3 xformvec4x3T(V3 p, const float* m) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
          m[8] * p.x + m[9] * p.y + m[10] * p.z};
}
// auxiliary.h:119-129
GS_DEV V3 dnormvdv3(V3 v, V3 dv) {
  float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  V3 r;
  r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return r;
}
// auxiliary.h:40-43: evaluated in double, rounded once
GS_DEV float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:45-55
GS_DEV void get_rect(float px, float py, int max_radius, uint32_t gx, uint32_t gy, uint32_t& minx, uint32_t& miny,
                     uint32_t& maxx, uint32_t& maxy) {
  minx = min(gx, (uint32_t)max(0, f2i_sat((px - max_radius) / TILE_X)));
  miny = min(gy, (uint32_t)max(0, f2i_sat((py - max_radius) / TILE_Y)));
  maxx = min(gx, (uint32_t)max(0, f2i_sat((px + max_radius + TILE_X - 1) / TILE_X)));
  maxy = min(gy, (uint32_t)max(0, f2i_sat((py + max_radius + TILE_Y - 1) / TILE_Y)));
}

GS_DEV M3 quat_to_R(V4 q) {
  float r = q.x, x = q.y, y = q.z, z = q.w;
  return mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y), 2.f * (x * y + r * z),
                   1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x), 2.f * (x * z - r * y), 2.f * (y * z + r * x),
                   1.f - 2.f * (x * x + y * y));
}

struct Cov2DInter {
  V3 t;
  M3 T, Vrk;
  float txtz, tytz, limx, limy;
};
// forward.cu:74-109 / backward.cu:162-201
GS_DEV void cov2d_common(V3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, const float* cov3D,
                         const float* vm, Cov2DInter& o) {
  V3 t = xform4x3(mean, vm);
  const float limx = 1.3f * tan_fovx;
  const float limy = 1.3f * tan_fovy;
  const float txtz = t.x / t.z;
  const float tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  M3 J = mat3_cols(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z,
                   -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
  M3 W = mat3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
  o.T = mul3(W, J);
  o.Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  o.t = t;
  o.txtz = txtz;
  o.tytz = tytz;
  o.limx = limx;
  o.limy = limy;
}

// GsGaussians.raw_activations: the model's activations applied where the parameters are read, in the arithmetic of
// act_fwd_kernel (gs_model.hip) - exp, F.normalize (v / max(|v|, 1e-12)), sigmoid - so that both forms give the same bits.
// (raw rows live in the flat parameter buffer, where the rotation segment is 16-byte aligned only for P % 4 == 0: scalar loads)
GS_DEV V3 activate_scales(V3 s, int raw) {
  if (!raw) return s;
  return {expf(s.x), expf(s.y), expf(s.z)};
}
GS_DEV V4 activate_rotation(V4 q, int raw) {
  if (!raw) return q;
  const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  return {q.x / n, q.y / n, q.z / n, q.w / n};
}
GS_DEV V3 load_scales(const float* p, int idx, int raw) {
  const V3 s = {p[3 * (size_t)idx], p[3 * (size_t)idx + 1], p[3 * (size_t)idx + 2]};
  return activate_scales(s, raw);
}
GS_DEV V4 load_rotation(const float* p, int idx, int raw) {
  if (!raw) {
    const float4 q4 = reinterpret_cast<const float4*>(p)[idx];
    return {q4.x, q4.y, q4.z, q4.w};
  }
  const float* q = p + 4 * (size_t)idx;
  return activate_rotation({q[0], q[1], q[2], q[3]}, raw);
}
// the 4th blended channel of the multispectral model: sigmoid(raw albedo) * clamp(gain, 0.1, 10)
// (mult-dwtgs/gaussian_renderer/__init__.py:166-169; torch.clamp keeps the gradient on [min, max])
GS_DEV float clamp_gain(float g) { return fminf(10.0f, fmaxf(0.1f, g)); }
GS_DEV float load_extra(const float* p, const float* gain, int idx, int raw) {
  const float x = p[idx];
  if (!raw || !gain) return x;
  return (1.0f / (1.0f + expf(-x))) * clamp_gain(*gain);
}
GS_DEV float activate_opacity(float o, int raw) { return raw ? 1.0f / (1.0f + expf(-o)) : o; }
GS_DEV float load_opacity(const float* p, int idx, int raw) { return activate_opacity(p[idx], raw); }

// auxiliary.h:21-38
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 -1.0925484305920792f
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 -1.0925484305920792f
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 -0.5900435899266435f
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 -0.4570457994644658f
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 -0.4570457994644658f
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 -0.5900435899266435f
