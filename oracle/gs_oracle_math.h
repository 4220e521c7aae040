/*
 * TEST INFRASTRUCTURE — CPU oracle, never shipped, never on the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Small fp32 helpers restating the few pieces of the header-only glm dependency that the
 * reference rasterizer uses (vendored at
 * /root/reference/fs3dgs_benchmark/gaussian-splatting/submodules/diff-gaussian-rasterization/third_party/glm),
 * with the SAME summation order so that results are bit-identical when compiled with
 * -ffp-contract=off:
 *   mat3 * mat3   glm/detail/type_mat3x3.inl:486-519   Result[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2]
 *   transpose     glm/detail/func_matrix.inl (plain element swap)
 *   dot(vec3)     glm/detail/func_geometric.inl        a.x*b.x + a.y*b.y + a.z*b.z
 *   length(vec3)  sqrt(dot(v,v))
 * Storage is column-major like glm: m.c[col][row].
 */
#ifndef GS_ORACLE_MATH_H
#define GS_ORACLE_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>

namespace gso {

struct V3 {
  float x, y, z;
};
struct V4 {
  float x, y, z, w;
};

static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; } /* glm: v * scalar per component */
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline float dot(V3 a, V3 b) {
  float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
  return tx + ty + tz;
}
static inline float length(V3 a) { return sqrtf(dot(a, a)); }

struct M3 {
  float c[3][3]; /* c[col][row] */
};

/* glm::mat3(a0,a1,a2, b0,b1,b2, c0,c1,c2): arguments fill COLUMNS. */
static inline M3 mat3_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0,
                           float c1, float c2) {
  M3 m;
  m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
  m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
  m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2;
  return m;
}

static inline M3 mul(const M3& A, const M3& B) {
  M3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++)
      R.c[c][r] = A.c[0][r] * B.c[c][0] + A.c[1][r] * B.c[c][1] + A.c[2][r] * B.c[c][2];
  return R;
}

static inline M3 transpose(const M3& A) {
  M3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R.c[c][r] = A.c[r][c];
  return R;
}

static inline M3 scale(float s, const M3& A) {
  M3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R.c[c][r] = A.c[c][r] * s;
  return R;
}

/* CUDA float->int conversion semantics (cvt.rzi.s32.f32): truncate, saturate, NaN -> 0.
 * A plain C cast is undefined out of range; the reference relies on the CUDA behaviour in
 * getRect (auxiliary.h:45-55) and the Morton quantisation (simple_knn.cu:55-62). */
static inline int32_t f2i_sat(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return INT32_MAX;
  if (f <= -2147483648.0f) return INT32_MIN;
  return (int32_t)f;
}
static inline uint32_t f2u_sat(float f) {
  if (f != f) return 0;
  if (f >= 4294967296.0f) return UINT32_MAX;
  if (f <= 0.0f) return 0;
  return (uint32_t)f;
}
static inline uint32_t fbits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

}  // namespace gso
#endif
