/*
 * TEST INFRASTRUCTURE — CPU oracle for the rasterizer hot path.  Never shipped, never on the
 * product path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.  The product library (libgsplat_hip.so) does not link, include or call this.
 *
 * What it is: a plain C++ restatement (fp32, built with -ffp-contract=off so no FMA is formed)
 * of the reference CUDA rasterizer's ALGORITHM, function by function, with the reference's
 * evaluation order, so that integer outputs (radii, tile counts, sort order, ranges) are the
 * reference's and float outputs are one valid outcome of the reference's arithmetic.
 * All reference paths are under
 *   /root/reference/fs3dgs_benchmark/gaussian-splatting/submodules/diff-gaussian-rasterization/
 *
 * Parity pinning: the reference has no test, golden vector or CPU path for the rasterizer
 *  and its CUDA cannot be built here (no nvcc, no NVIDIA device).  This file
 * is pinned by (1) golden vectors generated from the importable reference Python
 * (utils/sh_utils.eval_sh + autograd for SH colour fwd/bwd, utils/graphics_utils for the
 * camera matrices and geom_transform_points, utils/general_utils + scene/gaussian_model
 * build_covariance_from_scaling_rotation with autograd for computeCov3D forward and the
 * cov3D -> scale / quaternion backward, the renderer's python-covariance path:
 * tests/golden/make_golden.py, tests/golden/geometry.npz), (2) an independent float64 dense autograd
 * formulation in tests/dense_reference.py, (3) analytic known-answer tests.
 *
 * Deliberate, documented deviations from a literal transcription:
 *  - per-Gaussian gradient sums that the reference forms with float atomicAdd in undefined
 *    order (backward.cu:593-635) are accumulated here in double and rounded once: the
 *    order-independent centre of all valid reference outcomes.
 *  - float->int casts use the CUDA saturating semantics explicitly (gs_oracle_math.h).
 *  - OpenMP parallelism over Gaussians / tiles.  Forward results do not depend on the thread count.  The backward adds the
 *    tiles' contributions to a Gaussian's double accumulator in the order the threads finish (omp atomic, dynamic schedule):
 *    the double sums can differ in their last bits between runs, the fp32 gradients they are rounded to are the same except
 *    where a sum cancels to ~1e-11 of its terms (a value of that size on one run, exactly 0 on another: tests/test_model_ops.py
 *    allows for it; seen with 128 threads).
 */
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/gsplat.h"
#include "gs_oracle_math.h"

using namespace gso;

#define BLOCK_X GS_TILE_X
#define BLOCK_Y GS_TILE_Y
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NCH GS_NUM_CHANNELS

/* auxiliary.h:21-38 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                              -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                              0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                              -0.5900435899266435f};

/* ------------------------------------------------------------------------------------------
 * scratch layout (oracle-private; the HIP library has its own)
 * ---------------------------------------------------------------------------------------- */
struct GeomHdr {
  int64_t num_rendered;
  int32_t overflow;
  int32_t P;
};
static inline size_t al(size_t x) { return (x + 127) & ~(size_t)127; }

struct Geom {
  GeomHdr* hdr;
  float* depths;         /* [P] */
  uint8_t* clamped;      /* [3P] */
  float* means2D;        /* [2P] */
  float* cov3D;          /* [6P] */
  float* conic_opacity;  /* [4P] */
  float* rgb;            /* [3P] */
  uint32_t* tiles_touched; /* [P] */
  uint32_t* point_offsets; /* [P] */
  int32_t* radii;          /* [P] internal_radii (rasterizer_impl.cu:160) */
};
static size_t geom_bytes(size_t P) {
  return al(sizeof(GeomHdr)) + al(4 * P) + al(3 * P) + al(8 * P) + al(24 * P) + al(16 * P) +
         al(12 * P) + al(4 * P) + al(4 * P) + al(4 * P);
}
static Geom geom_from(void* buf, size_t P) {
  char* p = (char*)buf;
  Geom g;
  g.hdr = (GeomHdr*)p; p += al(sizeof(GeomHdr));
  g.depths = (float*)p; p += al(4 * P);
  g.clamped = (uint8_t*)p; p += al(3 * P);
  g.means2D = (float*)p; p += al(8 * P);
  g.cov3D = (float*)p; p += al(24 * P);
  g.conic_opacity = (float*)p; p += al(16 * P);
  g.rgb = (float*)p; p += al(12 * P);
  g.tiles_touched = (uint32_t*)p; p += al(4 * P);
  g.point_offsets = (uint32_t*)p; p += al(4 * P);
  g.radii = (int32_t*)p; p += al(4 * P);
  return g;
}
struct Img {
  float* accum_alpha;  /* final_T [N] */
  uint32_t* n_contrib; /* [N] */
  uint32_t* ranges;    /* [T][2] */
};
static size_t img_bytes(size_t N, size_t T) { return al(4 * N) + al(4 * N) + al(8 * T); }
static Img img_from(void* buf, size_t N, size_t T) {
  char* p = (char*)buf;
  Img im;
  im.accum_alpha = (float*)p; p += al(4 * N);
  im.n_contrib = (uint32_t*)p; p += al(4 * N);
  im.ranges = (uint32_t*)p; p += al(8 * T);
  (void)T;
  return im;
}
struct Binning {
  uint32_t* point_list;
  uint32_t* point_list_unsorted;
  uint64_t* keys;
  uint64_t* keys_unsorted;
};
static size_t binning_bytes(size_t R) { return al(4 * R) + al(4 * R) + al(8 * R) + al(8 * R); }
static Binning binning_from(void* buf, size_t R) {
  char* p = (char*)buf;
  Binning b;
  b.point_list = (uint32_t*)p; p += al(4 * R);
  b.point_list_unsorted = (uint32_t*)p; p += al(4 * R);
  b.keys = (uint64_t*)p; p += al(8 * R);
  b.keys_unsorted = (uint64_t*)p; p += al(8 * R);
  return b;
}

/* ------------------------------------------------------------------------------------------
 * auxiliary.h helpers
 * ---------------------------------------------------------------------------------------- */
/* auxiliary.h:40-43 — evaluated in DOUBLE (the 1.0 / 0.5 literals promote), rounded once. */
static inline float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:45-55 */
static inline void getRect(float px, float py, int max_radius, uint32_t gx, uint32_t gy,
                           uint32_t rect_min[2], uint32_t rect_max[2]) {
  rect_min[0] = std::min(gx, (uint32_t)std::max(0, f2i_sat((px - max_radius) / BLOCK_X)));
  rect_min[1] = std::min(gy, (uint32_t)std::max(0, f2i_sat((py - max_radius) / BLOCK_Y)));
  rect_max[0] = std::min(gx, (uint32_t)std::max(0, f2i_sat((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
  rect_max[1] = std::min(gy, (uint32_t)std::max(0, f2i_sat((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}
/* auxiliary.h:70-78 */
static inline V3 transformPoint4x3(V3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
/* auxiliary.h:80-89 */
static inline V4 transformPoint4x4(V3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
/* auxiliary.h:101-109 */
static inline V3 transformVec4x3Transpose(V3 p, const float* m) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
          m[8] * p.x + m[9] * p.y + m[10] * p.z};
}
/* auxiliary.h:119-129 */
static inline V3 dnormvdv(V3 v, V3 dv) {
  float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  V3 r;
  r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return r;
}

/* rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4;
  uint32_t step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb)
      msb += step;
    else
      msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

/* ------------------------------------------------------------------------------------------
 * forward.cu device functions
 * ---------------------------------------------------------------------------------------- */
/* forward.cu:20-71 */
static V3 computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, V3 campos,
                             const float* shs, uint8_t* clamped) {
  V3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
  V3 dir = pos - campos;
  dir = dir / length(dir);
  const V3* sh = ((const V3*)shs) + (size_t)idx * max_coeffs;
  V3 result = SH_C0 * sh[0];
  if (deg > 0) {
    float x = dir.x, y = dir.y, z = dir.z;
    result = result - SH_C1 * y * sh[1] + SH_C1 * z * sh[2] - SH_C1 * x * sh[3];
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z;
      float xy = x * y, yz = y * z, xz = x * z;
      result = result + SH_C2[0] * xy * sh[4] + SH_C2[1] * yz * sh[5] +
               SH_C2[2] * (2.0f * zz - xx - yy) * sh[6] + SH_C2[3] * xz * sh[7] +
               SH_C2[4] * (xx - yy) * sh[8];
      if (deg > 2) {
        result = result + SH_C3[0] * y * (3.0f * xx - yy) * sh[9] + SH_C3[1] * xy * z * sh[10] +
                 SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11] +
                 SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                 SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13] + SH_C3[5] * z * (xx - yy) * sh[14] +
                 SH_C3[6] * x * (xx - 3.0f * yy) * sh[15];
      }
    }
  }
  result.x += 0.5f;
  result.y += 0.5f;
  result.z += 0.5f;
  clamped[3 * idx + 0] = (result.x < 0);
  clamped[3 * idx + 1] = (result.y < 0);
  clamped[3 * idx + 2] = (result.z < 0);
  return {fmaxf(result.x, 0.0f), fmaxf(result.y, 0.0f), fmaxf(result.z, 0.0f)};
}

struct Cov2DInter {
  V3 t;       /* clamped view-space mean */
  M3 T;       /* W * J */
  M3 Vrk;
  float txtz, tytz, limx, limy;
};

/* forward.cu:74-109 (also the recomputation at backward.cu:162-201) */
static inline void cov2d_common(V3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                                const float* cov3D, const float* viewmatrix, Cov2DInter& o) {
  V3 t = transformPoint4x3(mean, viewmatrix);
  const float limx = 1.3f * tan_fovx;
  const float limy = 1.3f * tan_fovy;
  const float txtz = t.x / t.z;
  const float tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  M3 J = mat3_cols(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z,
                   -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
  M3 W = mat3_cols(viewmatrix[0], viewmatrix[4], viewmatrix[8], viewmatrix[1], viewmatrix[5],
                   viewmatrix[9], viewmatrix[2], viewmatrix[6], viewmatrix[10]);
  o.T = mul(W, J);
  o.Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4],
                    cov3D[5]);
  o.t = t;
  o.txtz = txtz;
  o.tytz = tytz;
  o.limx = limx;
  o.limy = limy;
}
static inline V3 computeCov2D(V3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                              const float* cov3D, const float* viewmatrix) {
  Cov2DInter c;
  cov2d_common(mean, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, c);
  M3 cov = mul(mul(transpose(c.T), transpose(c.Vrk)), c.T);
  return {cov.c[0][0], cov.c[0][1], cov.c[1][1]};
}

static inline M3 quat_to_R(V4 q) {
  float r = q.x, x = q.y, y = q.z, z = q.w;
  return mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

/* forward.cu:114-148 */
static void computeCov3D(V3 scale, float mod, V4 rot, float* cov3D) {
  M3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
  S.c[0][0] = mod * scale.x;
  S.c[1][1] = mod * scale.y;
  S.c[2][2] = mod * scale.z;
  M3 R = quat_to_R(rot);
  M3 M = mul(S, R);
  M3 Sigma = mul(transpose(M), M);
  cov3D[0] = Sigma.c[0][0];
  cov3D[1] = Sigma.c[0][1];
  cov3D[2] = Sigma.c[0][2];
  cov3D[3] = Sigma.c[1][1];
  cov3D[4] = Sigma.c[1][2];
  cov3D[5] = Sigma.c[2][2];
}

/* forward.cu:151-269 */
static void preprocess_one(int idx, int D, int M, const GsView* v, const GsGaussians* g,
                           const float bgcam[3], float focal_x, float focal_y, int* radii, Geom& gs,
                           uint32_t gx, uint32_t gy) {
  const int W = v->image_width, H = v->image_height;
  radii[idx] = 0;
  gs.tiles_touched[idx] = 0;
  V3 p_orig = {g->means3D[3 * idx], g->means3D[3 * idx + 1], g->means3D[3 * idx + 2]};
  /* in_frustum, auxiliary.h:151-176 */
  V3 p_view = transformPoint4x3(p_orig, v->viewmatrix);
  if (p_view.z <= 0.2f) return;
  V4 p_hom = transformPoint4x4(p_orig, v->projmatrix);
  float p_w = 1.0f / (p_hom.w + 0.0000001f);
  V3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};

  const float* cov3D;
  if (g->cov3D_precomp != nullptr) {
    cov3D = g->cov3D_precomp + (size_t)idx * 6;
  } else {
    V3 sc = {g->scales[3 * idx], g->scales[3 * idx + 1], g->scales[3 * idx + 2]};
    V4 rq = {g->rotations[4 * idx], g->rotations[4 * idx + 1], g->rotations[4 * idx + 2],
             g->rotations[4 * idx + 3]};
    computeCov3D(sc, v->scale_modifier, rq, gs.cov3D + (size_t)idx * 6);
    cov3D = gs.cov3D + (size_t)idx * 6;
  }
  V3 cov = computeCov2D(p_orig, focal_x, focal_y, v->tanfovx, v->tanfovy, cov3D, v->viewmatrix);

  const float h_var = 0.3f;
  const float det_cov = cov.x * cov.z - cov.y * cov.y;
  cov.x += h_var;
  cov.z += h_var;
  const float det_cov_plus_h_cov = cov.x * cov.z - cov.y * cov.y;
  float h_convolution_scaling = 1.0f;
  if (v->antialiasing) h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
  const float det = det_cov_plus_h_cov;
  if (det == 0.0f) return;
  float det_inv = 1.f / det;
  V3 conic = {cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv};
  float mid = 0.5f * (cov.x + cov.z);
  float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
  float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
  float pix_x = ndc2Pix(p_proj.x, W), pix_y = ndc2Pix(p_proj.y, H);
  uint32_t rect_min[2], rect_max[2];
  getRect(pix_x, pix_y, f2i_sat(my_radius), gx, gy, rect_min, rect_max);
  if ((rect_max[0] - rect_min[0]) * (rect_max[1] - rect_min[1]) == 0) return;

  if (g->colors_precomp == nullptr) {
    V3 campos = {bgcam[0], bgcam[1], bgcam[2]};
    V3 res = computeColorFromSH(idx, D, M, g->means3D, campos, g->shs, gs.clamped);
    gs.rgb[idx * NCH + 0] = res.x;
    gs.rgb[idx * NCH + 1] = res.y;
    gs.rgb[idx * NCH + 2] = res.z;
  }
  gs.depths[idx] = p_view.z;
  radii[idx] = f2i_sat(my_radius);
  gs.means2D[2 * idx] = pix_x;
  gs.means2D[2 * idx + 1] = pix_y;
  float opacity = g->opacities[idx];
  gs.conic_opacity[4 * idx + 0] = conic.x;
  gs.conic_opacity[4 * idx + 1] = conic.y;
  gs.conic_opacity[4 * idx + 2] = conic.z;
  gs.conic_opacity[4 * idx + 3] = opacity * h_convolution_scaling;
  gs.tiles_touched[idx] = (rect_max[1] - rect_min[1]) * (rect_max[0] - rect_min[0]);
}

static int check_args(const GsView* v, const GsGaussians* g) {
  if (!v || !g) return GS_E_NULL;
  if (g->P < 0 || v->image_width <= 0 || v->image_height <= 0) return GS_E_SHAPE;
  if (g->P == 0) return GS_OK;
  if (!g->means3D || !g->opacities || !v->viewmatrix || !v->projmatrix || !v->bg) return GS_E_NULL;
  if ((g->shs == nullptr) == (g->colors_precomp == nullptr)) return GS_E_SHAPE;
  bool has_sr = g->scales != nullptr && g->rotations != nullptr;
  bool any_sr = g->scales != nullptr || g->rotations != nullptr;
  if ((!has_sr && g->cov3D_precomp == nullptr) || (any_sr && g->cov3D_precomp != nullptr)) return GS_E_SHAPE;
  if (g->shs && (g->M < (v->sh_degree + 1) * (v->sh_degree + 1) || !v->campos)) return GS_E_SHAPE;
  /* the split SH rows are a memory layout of the product (same numbers from two arrays): the oracle takes the [P,M,3] rows and
   * tests/test_gpu_render_raw.py compares the product's two layouts bit for bit */
  if (g->shs_rest) return GS_E_UNSUPPORTED;
  return GS_OK;
}

extern "C" {

int gso_abi_version(void) { return GS_ABI_VERSION; }
size_t gso_struct_bytes(int32_t which) {
  const size_t n[7] = {sizeof(GsView), sizeof(GsGaussians), sizeof(GsScratch), sizeof(GsGrads), sizeof(GsStepState),
                       sizeof(GsLgdwtParams), sizeof(GsAdamSeg)};
  return (which >= 0 && which < 7) ? n[which] : 0;
}
const char* gso_build_info(void) { return "gs_oracle: CPU restatement, fp32, -ffp-contract=off"; }

int gso_scratch_bytes(int32_t P, int32_t W, int32_t H, int64_t R, size_t out[3], size_t* bwd_ws) {
  if (!out) return GS_E_NULL;
  if (P < 0 || W <= 0 || H <= 0 || R < 0) return GS_E_SHAPE;
  size_t T = (size_t)((W + BLOCK_X - 1) / BLOCK_X) * ((H + BLOCK_Y - 1) / BLOCK_Y);
  out[0] = geom_bytes((size_t)P);
  out[1] = img_bytes((size_t)W * H, T);
  out[2] = binning_bytes((size_t)std::max<int64_t>(R, 1));
  if (bwd_ws) *bwd_ws = 256 + (size_t)P * 16 * sizeof(double); /* receives the blend-backward sums (parity probe) */
  return GS_OK;
}

int gso_forward_geometry(const GsView* v, const GsGaussians* g, GsScratch* s, int32_t* radii,
                         int32_t* num_rendered_host, void* /*stream*/) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!s || !s->geom) return GS_E_NULL;
  const int P = g->P;
  if (s->geom_bytes < geom_bytes(P)) return GS_E_SCRATCH;
  Geom gs = geom_from(s->geom, P);
  gs.hdr->P = P;
  gs.hdr->overflow = 0;
  gs.hdr->num_rendered = 0;
  if (P == 0) {
    if (num_rendered_host) *num_rendered_host = 0;
    return GS_OK;
  }
  if (!radii) return GS_E_NULL;
  /* rasterizer_impl.cu:224-225 */
  const float focal_y = v->image_height / (2.0f * v->tanfovy);
  const float focal_x = v->image_width / (2.0f * v->tanfovx);
  const uint32_t gx = (v->image_width + BLOCK_X - 1) / BLOCK_X, gy = (v->image_height + BLOCK_Y - 1) / BLOCK_Y;
  float campos[3] = {0, 0, 0};
  if (v->campos) memcpy(campos, v->campos, 12);
  memset(gs.clamped, 0, 3 * (size_t)P);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++)
    preprocess_one(i, v->sh_degree, g->M, v, g, campos, focal_x, focal_y, radii, gs, gx, gy);
  /* cub::DeviceScan::InclusiveSum, rasterizer_impl.cu:280 */
  uint32_t acc = 0;
  for (int i = 0; i < P; i++) {
    acc += gs.tiles_touched[i];
    gs.point_offsets[i] = acc;
  }
  gs.hdr->num_rendered = acc;
  memcpy(gs.radii, radii, 4 * (size_t)P);
  if (num_rendered_host) *num_rendered_host = (int32_t)acc;
  return GS_OK;
}

/* stable LSD radix sort on key bits [0, end_bit) — the semantics of
 * cub::DeviceRadixSort::SortPairs(..., 0, 32+bit), rasterizer_impl.cu:306-311 */
static void radix_sort_pairs(uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, size_t n,
                             int end_bit) {
  uint64_t* ka = kin;
  uint64_t* kb = kout;
  uint32_t* va = vin;
  uint32_t* vb = vout;
  for (int shift = 0; shift < end_bit; shift += 8) {
    int bits = std::min(8, end_bit - shift);
    uint32_t mask = (1u << bits) - 1;
    size_t cnt[257] = {0};
    for (size_t i = 0; i < n; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
    for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
    for (size_t i = 0; i < n; i++) {
      size_t d = (ka[i] >> shift) & mask;
      kb[cnt[d]] = ka[i];
      vb[cnt[d]++] = va[i];
    }
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  if (ka != kout) {
    memcpy(kout, ka, n * 8);
    memcpy(vout, va, n * 4);
  }
}

/* forward.cu:274-397, one tile */
static int g_fsgs_exact_T = getenv("GSO_FSGS_T") != nullptr;

static void render_tile_fwd(uint32_t tx, uint32_t ty, uint32_t hblocks, int W, int H, const Img& im,
                            const uint32_t* point_list, const Geom& gs, const float* features,
                            const float* bg, float* out_color, float* invdepth, float* fs_depth = nullptr,
                            float* fs_alpha = nullptr) {
  /* fs_depth / fs_alpha != NULL: the older rasterizer generation of FSGS / DNGaussian
   * (FSGS/submodules/diff-gaussian-rasterization-confidence/cuda_rasterizer/forward.cu:262-380): per pixel
   * depth = sum depth_i alpha_i T_i and alpha = sum alpha_i T_i; its backward reads T_final back as 1 - alpha
   * (backward.cu:461), so that is what the image state keeps. */
  const uint32_t r0 = im.ranges[2 * (ty * hblocks + tx)], r1 = im.ranges[2 * (ty * hblocks + tx) + 1];
  for (uint32_t ly = 0; ly < BLOCK_Y; ly++)
    for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
      uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
      if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
      uint32_t pix_id = W * py + px;
      float pixfx = (float)px, pixfy = (float)py;
      float T = 1.0f;
      uint32_t contributor = 0, last_contributor = 0;
      float C[NCH] = {0};
      float expected_invdepth = 0.0f;
      float weight = 0.0f, Dsum = 0.0f;
      for (uint32_t k = r0; k < r1; k++) {
        contributor++;
        uint32_t id = point_list[k];
        float dx = gs.means2D[2 * id] - pixfx, dy = gs.means2D[2 * id + 1] - pixfy;
        const float* co = gs.conic_opacity + 4 * (size_t)id;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        float alpha = fminf(0.99f, co[3] * expf(power));
        if (alpha < 1.0f / 255.0f) continue;
        float test_T = T * (1 - alpha);
        if (test_T < 0.0001f) break; /* done = true */
        for (int ch = 0; ch < NCH; ch++) C[ch] += features[id * NCH + ch] * alpha * T;
        if (invdepth) expected_invdepth += (1 / gs.depths[id]) * alpha * T;
        if (fs_alpha) {
          weight += alpha * T;
          Dsum += gs.depths[id] * alpha * T;
        }
        T = test_T;
        last_contributor = contributor;
      }
      /* test probe (GSO_FSGS_T=1 or gso_set_fsgs_exact_T(1)): keep the exact product instead of the reference's
       * 1 - alpha read-back, which loses up to ~1e-3 relative accuracy of T on saturated pixels
       * (tests/test_fsgs_cpu.py attributes the difference; the HIP path keeps the product) */
      im.accum_alpha[pix_id] = (fs_alpha && !g_fsgs_exact_T) ? 1 - weight : T;
      im.n_contrib[pix_id] = last_contributor;
      if (fs_alpha) {
        fs_alpha[pix_id] = weight;
        fs_depth[pix_id] = Dsum;
      }
      for (int ch = 0; ch < NCH; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg[ch];
      if (invdepth) invdepth[pix_id] = expected_invdepth;
    }
}

static int forward_render_impl(const GsView* v, const GsGaussians* g, GsScratch* s, float* out_color,
                               float* out_invdepth, float* fs_depth, float* fs_alpha);

int gso_forward_render(const GsView* v, const GsGaussians* g, GsScratch* s, float* out_color,
                       float* out_invdepth, void* /*stream*/) {
  return forward_render_impl(v, g, s, out_color, out_invdepth, nullptr, nullptr);
}

/* dgr_fsgs `rasterize_gaussians` (rasterize_points.cu of the -confidence fork): colour, depth, alpha. */
int gso_forward_render_fsgs(const GsView* v, const GsGaussians* g, GsScratch* s, float* out_color,
                            float* out_depth, float* out_alpha, void* /*stream*/) {
  if (!out_depth || !out_alpha) return GS_E_NULL;
  if (v && v->antialiasing) return GS_E_UNSUPPORTED; /* that generation has no anti-aliasing */
  return forward_render_impl(v, g, s, out_color, nullptr, out_depth, out_alpha);
}

static int forward_render_impl(const GsView* v, const GsGaussians* g, GsScratch* s, float* out_color,
                               float* out_invdepth, float* fs_depth, float* fs_alpha) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!s || !s->geom || !s->img || !out_color) return GS_E_NULL;
  const int P = g->P, W = v->image_width, H = v->image_height;
  const uint32_t gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  if (s->img_bytes < img_bytes(N, T)) return GS_E_SCRATCH;
  if (P == 0) { /* rasterize_points.cu:88: outputs stay zero */
    memset(out_color, 0, sizeof(float) * NCH * N);
    if (out_invdepth) memset(out_invdepth, 0, sizeof(float) * N);
    if (fs_alpha) {
      memset(fs_alpha, 0, sizeof(float) * N);
      memset(fs_depth, 0, sizeof(float) * N);
    }
    return GS_OK;
  }
  Geom gs = geom_from(s->geom, P);
  Img im = img_from(s->img, N, T);
  const int64_t R = gs.hdr->num_rendered;
  if (R > s->binning_capacity || (R > 0 && (!s->binning || s->binning_bytes < binning_bytes(R)))) {
    gs.hdr->overflow = 1;
    return GS_E_OVERFLOW;
  }
  gs.hdr->overflow = 0;
  Binning b = binning_from(s->binning, (size_t)std::max<int64_t>(s->binning_capacity, 1));

  /* duplicateWithKeys, rasterizer_impl.cu:70-111 */
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    if (gs.radii[idx] > 0) {
      uint32_t off = (idx == 0) ? 0 : gs.point_offsets[idx - 1];
      uint32_t rect_min[2], rect_max[2];
      getRect(gs.means2D[2 * idx], gs.means2D[2 * idx + 1], gs.radii[idx], gx, gy, rect_min, rect_max);
      for (uint32_t y = rect_min[1]; y < rect_max[1]; y++)
        for (uint32_t x = rect_min[0]; x < rect_max[0]; x++) {
          uint64_t key = (uint64_t)y * gx + x;
          key <<= 32;
          key |= fbits(gs.depths[idx]);
          b.keys_unsorted[off] = key;
          b.point_list_unsorted[off] = (uint32_t)idx;
          off++;
        }
    }
  }
  /* rasterizer_impl.cu:303-311 */
  int bit = (int)getHigherMsb(gx * gy);
  radix_sort_pairs(b.keys_unsorted, b.keys, b.point_list_unsorted, b.point_list, (size_t)R, 32 + bit);
  /* NB: radix_sort_pairs may have used keys_unsorted as ping-pong space (as cub does). */

  /* cudaMemset ranges + identifyTileRanges, rasterizer_impl.cu:313-321,116-138 */
  memset(im.ranges, 0, 8 * T);
  for (int64_t idx = 0; idx < R; idx++) {
    uint32_t currtile = (uint32_t)(b.keys[idx] >> 32);
    if (idx == 0)
      im.ranges[2 * currtile] = 0;
    else {
      uint32_t prevtile = (uint32_t)(b.keys[idx - 1] >> 32);
      if (currtile != prevtile) {
        im.ranges[2 * prevtile + 1] = (uint32_t)idx;
        im.ranges[2 * currtile] = (uint32_t)idx;
      }
    }
    if (idx == R - 1) im.ranges[2 * currtile + 1] = (uint32_t)R;
  }

  const float* feature_ptr = g->colors_precomp != nullptr ? g->colors_precomp : gs.rgb;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (uint32_t ty = 0; ty < gy; ty++)
    for (uint32_t tx = 0; tx < gx; tx++)
      render_tile_fwd(tx, ty, gx, W, H, im, b.point_list, gs, feature_ptr, v->bg, out_color, out_invdepth, fs_depth,
                      fs_alpha);
  return GS_OK;
}

/* ------------------------------------------------------------------------------------------
 * backward
 * ---------------------------------------------------------------------------------------- */
enum { A_MX = 0, A_MY, A_CXX, A_CXY, A_CYY, A_OP, A_CR, A_CG, A_CB, A_ID, A_N };

/* backward.cu:452-638, one tile; per-instance partial sums go to loc[(k - r0) * A_N + ...] */
static void render_tile_bwd(uint32_t tx, uint32_t ty, uint32_t hblocks, int W, int H, const Img& im,
                            const uint32_t* point_list, const Geom& gs, const float* colors,
                            const float* bg, const float* dL_dpixels, const float* dL_invdepths,
                            std::vector<double>& loc, std::vector<uint8_t>& touched,
                            const float* fs_dL_ddepth = nullptr, const float* fs_dL_dalpha = nullptr) {
  /* fs_*: FSGS generation (-confidence fork, backward.cu:414-600): depth and alpha image gradients; A_ID then
   * accumulates dL_ddepth per Gaussian (backward.cu:563). */
  const uint32_t r0 = im.ranges[2 * (ty * hblocks + tx)], r1 = im.ranges[2 * (ty * hblocks + tx) + 1];
  const uint32_t n = r1 - r0;
  loc.assign((size_t)n * A_N, 0.0);
  touched.assign(n, 0);
  const float ddelx_dx = (float)(0.5 * W);
  const float ddely_dy = (float)(0.5 * H);
  for (uint32_t ly = 0; ly < BLOCK_Y; ly++)
    for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
      uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
      if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
      const uint32_t pix_id = W * py + px;
      const float pixfx = (float)px, pixfy = (float)py;
      const float T_final = im.accum_alpha[pix_id];
      float T = T_final;
      uint32_t contributor = n;
      const uint32_t last_contributor = im.n_contrib[pix_id];
      float accum_rec[NCH] = {0};
      float dL_dpixel[NCH];
      float dL_invdepth = 0;
      float accum_invdepth_rec = 0;
      for (int i = 0; i < NCH; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
      if (dL_invdepths) dL_invdepth = dL_invdepths[pix_id];
      float fs_gd = 0, fs_ga = 0, accum_depth_rec = 0, accum_alpha_rec = 0, last_depth = 0;
      if (fs_dL_dalpha) {
        fs_gd = fs_dL_ddepth[pix_id];
        fs_ga = fs_dL_dalpha[pix_id];
      }
      float last_alpha = 0;
      float last_color[NCH] = {0};
      float last_invdepth = 0;
      for (uint32_t k = r1; k-- > r0;) {
        contributor--;
        if (contributor >= last_contributor) continue;
        const uint32_t id = point_list[k];
        const float dx = gs.means2D[2 * id] - pixfx, dy = gs.means2D[2 * id + 1] - pixfy;
        const float* co = gs.conic_opacity + 4 * (size_t)id;
        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        const float G = expf(power);
        const float alpha = fminf(0.99f, co[3] * G);
        if (alpha < 1.0f / 255.0f) continue;
        T = T / (1.f - alpha);
        const float dchannel_dcolor = alpha * T;
        double* acc = &loc[(size_t)(k - r0) * A_N];
        touched[k - r0] = 1;
        float dL_dalpha = 0.0f;
        for (int ch = 0; ch < NCH; ch++) {
          const float c = colors[id * NCH + ch];
          accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
          last_color[ch] = c;
          const float dL_dchannel = dL_dpixel[ch];
          dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
          acc[A_CR + ch] += (double)(dchannel_dcolor * dL_dchannel);
        }
        if (dL_invdepths) {
          const float invd = 1.f / gs.depths[id];
          accum_invdepth_rec = last_alpha * last_invdepth + (1.f - last_alpha) * accum_invdepth_rec;
          last_invdepth = invd;
          dL_dalpha += (invd - accum_invdepth_rec) * dL_invdepth;
          acc[A_ID] += (double)(dchannel_dcolor * dL_invdepth);
        }
        if (fs_dL_dalpha) {
          const float c_d = gs.depths[id];
          accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
          last_depth = c_d;
          dL_dalpha += (c_d - accum_depth_rec) * fs_gd;
          acc[A_ID] += (double)(dchannel_dcolor * fs_gd);
          accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
          dL_dalpha += (1 - accum_alpha_rec) * fs_ga;
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        float bg_dot_dpixel = 0;
        for (int i = 0; i < NCH; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
        const float dL_dG = co[3] * dL_dalpha;
        const float gdx = G * dx;
        const float gdy = G * dy;
        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
        const float dG_ddely = -gdy * co[2] - gdx * co[1];
        acc[A_MX] += (double)(dL_dG * dG_ddelx * ddelx_dx);
        acc[A_MY] += (double)(dL_dG * dG_ddely * ddely_dy);
        acc[A_CXX] += (double)(-0.5f * gdx * dx * dL_dG);
        acc[A_CXY] += (double)(-0.5f * gdx * dy * dL_dG);
        acc[A_CYY] += (double)(-0.5f * gdy * dy * dL_dG);
        acc[A_OP] += (double)(G * dL_dalpha);
      }
    }
}

static inline float sq(float x) { return x * x; }

/* backward.cu:147-326 */
static void computeCov2D_bwd(int idx, const GsView* v, const float* means, const int* radii,
                             const float* cov3Ds, float h_x, float h_y, const float* opacities,
                             const float* dL_dconics /*[P][3]*/, float* dL_dopacity,
                             const float* dL_dinvdepth, float* dL_dmeans, float* dL_dcov) {
  if (!(radii[idx] > 0)) return;
  const float* cov3D = cov3Ds + 6 * (size_t)idx;
  V3 mean = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
  V3 dL_dconic = {dL_dconics[3 * idx], dL_dconics[3 * idx + 1], dL_dconics[3 * idx + 2]};
  Cov2DInter c;
  cov2d_common(mean, h_x, h_y, v->tanfovx, v->tanfovy, cov3D, v->viewmatrix, c);
  const V3 t = c.t;
  const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0 : 1;
  const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0 : 1;
  const M3& T = c.T;
  const M3& Vrk = c.Vrk;
  M3 W = mat3_cols(v->viewmatrix[0], v->viewmatrix[4], v->viewmatrix[8], v->viewmatrix[1],
                   v->viewmatrix[5], v->viewmatrix[9], v->viewmatrix[2], v->viewmatrix[6],
                   v->viewmatrix[10]);
  M3 cov2D = mul(mul(transpose(T), transpose(Vrk)), T);
  float c_xx = cov2D.c[0][0];
  float c_xy = cov2D.c[0][1];
  float c_yy = cov2D.c[1][1];
  const float h_var = 0.3f;
  float d_inside_root = 0.f;
  if (v->antialiasing) {
    const float det_cov = c_xx * c_yy - c_xy * c_xy;
    c_xx += h_var;
    c_yy += h_var;
    const float det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
    const float h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
    const float dL_dopacity_v = dL_dopacity[idx];
    const float d_h_convolution_scaling = dL_dopacity_v * opacities[idx];
    dL_dopacity[idx] = dL_dopacity_v * h_convolution_scaling;
    d_inside_root = (det_cov / det_cov_plus_h_cov) <= 0.000025f ? 0.f : d_h_convolution_scaling / (2 * h_convolution_scaling);
  } else {
    c_xx += h_var;
    c_yy += h_var;
  }
  float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
  if (v->antialiasing) {
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
    dL_dcov[6 * idx + 0] = (T.c[0][0] * T.c[0][0] * dL_dc_xx + T.c[0][0] * T.c[1][0] * dL_dc_xy + T.c[1][0] * T.c[1][0] * dL_dc_yy);
    dL_dcov[6 * idx + 3] = (T.c[0][1] * T.c[0][1] * dL_dc_xx + T.c[0][1] * T.c[1][1] * dL_dc_xy + T.c[1][1] * T.c[1][1] * dL_dc_yy);
    dL_dcov[6 * idx + 5] = (T.c[0][2] * T.c[0][2] * dL_dc_xx + T.c[0][2] * T.c[1][2] * dL_dc_xy + T.c[1][2] * T.c[1][2] * dL_dc_yy);
    dL_dcov[6 * idx + 1] = 2 * T.c[0][0] * T.c[0][1] * dL_dc_xx + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_dc_xy + 2 * T.c[1][0] * T.c[1][1] * dL_dc_yy;
    dL_dcov[6 * idx + 2] = 2 * T.c[0][0] * T.c[0][2] * dL_dc_xx + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_dc_xy + 2 * T.c[1][0] * T.c[1][2] * dL_dc_yy;
    dL_dcov[6 * idx + 4] = 2 * T.c[0][2] * T.c[0][1] * dL_dc_xx + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_dc_xy + 2 * T.c[1][1] * T.c[1][2] * dL_dc_yy;
  } else {
    for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
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
  float dL_dJ00 = W.c[0][0] * dL_dT00 + W.c[0][1] * dL_dT01 + W.c[0][2] * dL_dT02;
  float dL_dJ02 = W.c[2][0] * dL_dT00 + W.c[2][1] * dL_dT01 + W.c[2][2] * dL_dT02;
  float dL_dJ11 = W.c[1][0] * dL_dT10 + W.c[1][1] * dL_dT11 + W.c[1][2] * dL_dT12;
  float dL_dJ12 = W.c[2][0] * dL_dT10 + W.c[2][1] * dL_dT11 + W.c[2][2] * dL_dT12;
  float tz = 1.f / t.z;
  float tz2 = tz * tz;
  float tz3 = tz2 * tz;
  float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
  float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
  float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
  if (dL_dinvdepth) dL_dtz -= dL_dinvdepth[idx] / (t.z * t.z);
  V3 dL_dmean = transformVec4x3Transpose({dL_dtx, dL_dty, dL_dtz}, v->viewmatrix);
  dL_dmeans[3 * idx + 0] = dL_dmean.x;
  dL_dmeans[3 * idx + 1] = dL_dmean.y;
  dL_dmeans[3 * idx + 2] = dL_dmean.z;
}

/* backward.cu:23-142 */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const float* means, V3 campos,
                                   const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                                   float* dL_dmeans, float* dL_dshs) {
  V3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
  V3 dir_orig = pos - campos;
  V3 dir = dir_orig / length(dir_orig);
  const V3* sh = ((const V3*)shs) + (size_t)idx * max_coeffs;
  V3 dL_dRGB = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
  dL_dRGB.x *= clamped[3 * idx + 0] ? 0 : 1;
  dL_dRGB.y *= clamped[3 * idx + 1] ? 0 : 1;
  dL_dRGB.z *= clamped[3 * idx + 2] ? 0 : 1;
  V3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
  float x = dir.x, y = dir.y, z = dir.z;
  V3* dL_dsh = ((V3*)dL_dshs) + (size_t)idx * max_coeffs;
  float dRGBdsh0 = SH_C0;
  dL_dsh[0] = dRGBdsh0 * dL_dRGB;
  if (deg > 0) {
    float dRGBdsh1 = -SH_C1 * y;
    float dRGBdsh2 = SH_C1 * z;
    float dRGBdsh3 = -SH_C1 * x;
    dL_dsh[1] = dRGBdsh1 * dL_dRGB;
    dL_dsh[2] = dRGBdsh2 * dL_dRGB;
    dL_dsh[3] = dRGBdsh3 * dL_dRGB;
    dRGBdx = -SH_C1 * sh[3];
    dRGBdy = -SH_C1 * sh[1];
    dRGBdz = SH_C1 * sh[2];
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z;
      float xy = x * y, yz = y * z, xz = x * z;
      float dRGBdsh4 = SH_C2[0] * xy;
      float dRGBdsh5 = SH_C2[1] * yz;
      float dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
      float dRGBdsh7 = SH_C2[3] * xz;
      float dRGBdsh8 = SH_C2[4] * (xx - yy);
      dL_dsh[4] = dRGBdsh4 * dL_dRGB;
      dL_dsh[5] = dRGBdsh5 * dL_dRGB;
      dL_dsh[6] = dRGBdsh6 * dL_dRGB;
      dL_dsh[7] = dRGBdsh7 * dL_dRGB;
      dL_dsh[8] = dRGBdsh8 * dL_dRGB;
      dRGBdx = dRGBdx + (SH_C2[0] * y * sh[4] + SH_C2[2] * 2.f * -x * sh[6] + SH_C2[3] * z * sh[7] + SH_C2[4] * 2.f * x * sh[8]);
      dRGBdy = dRGBdy + (SH_C2[0] * x * sh[4] + SH_C2[1] * z * sh[5] + SH_C2[2] * 2.f * -y * sh[6] + SH_C2[4] * 2.f * -y * sh[8]);
      dRGBdz = dRGBdz + (SH_C2[1] * y * sh[5] + SH_C2[2] * 2.f * 2.f * z * sh[6] + SH_C2[3] * x * sh[7]);
      if (deg > 2) {
        float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy);
        float dRGBdsh10 = SH_C3[1] * xy * z;
        float dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
        float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
        float dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
        float dRGBdsh14 = SH_C3[5] * z * (xx - yy);
        float dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
        dL_dsh[9] = dRGBdsh9 * dL_dRGB;
        dL_dsh[10] = dRGBdsh10 * dL_dRGB;
        dL_dsh[11] = dRGBdsh11 * dL_dRGB;
        dL_dsh[12] = dRGBdsh12 * dL_dRGB;
        dL_dsh[13] = dRGBdsh13 * dL_dRGB;
        dL_dsh[14] = dRGBdsh14 * dL_dRGB;
        dL_dsh[15] = dRGBdsh15 * dL_dRGB;
        dRGBdx = dRGBdx + (SH_C3[0] * sh[9] * 3.f * 2.f * xy + SH_C3[1] * sh[10] * yz + SH_C3[2] * sh[11] * -2.f * xy +
                           SH_C3[3] * sh[12] * -3.f * 2.f * xz + SH_C3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) +
                           SH_C3[5] * sh[14] * 2.f * xz + SH_C3[6] * sh[15] * 3.f * (xx - yy));
        dRGBdy = dRGBdy + (SH_C3[0] * sh[9] * 3.f * (xx - yy) + SH_C3[1] * sh[10] * xz +
                           SH_C3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * sh[12] * -3.f * 2.f * yz +
                           SH_C3[4] * sh[13] * -2.f * xy + SH_C3[5] * sh[14] * -2.f * yz +
                           SH_C3[6] * sh[15] * -3.f * 2.f * xy);
        dRGBdz = dRGBdz + (SH_C3[1] * sh[10] * xy + SH_C3[2] * sh[11] * 4.f * 2.f * yz +
                           SH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * sh[13] * 4.f * 2.f * xz +
                           SH_C3[5] * sh[14] * (xx - yy));
      }
    }
  }
  V3 dL_ddir = {dot(dRGBdx, dL_dRGB), dot(dRGBdy, dL_dRGB), dot(dRGBdz, dL_dRGB)};
  V3 dL_dmean = dnormvdv(dir_orig, dL_ddir);
  dL_dmeans[3 * idx + 0] += dL_dmean.x;
  dL_dmeans[3 * idx + 1] += dL_dmean.y;
  dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

/* backward.cu:330-393 */
static void computeCov3D_bwd(int idx, V3 scl, float mod, V4 rot, const float* dL_dcov3Ds,
                             float* dL_dscales, float* dL_drots) {
  float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
  M3 R = quat_to_R(rot);
  M3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
  V3 s = mod * scl;
  S.c[0][0] = s.x;
  S.c[1][1] = s.y;
  S.c[2][2] = s.z;
  M3 M = mul(S, R);
  const float* d = dL_dcov3Ds + 6 * (size_t)idx;
  M3 dL_dSigma = mat3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2],
                           0.5f * d[4], d[5]);
  M3 dL_dM = mul(scale(2.0f, M), dL_dSigma);
  M3 Rt = transpose(R);
  M3 dL_dMt = transpose(dL_dM);
  V3 Rt0 = {Rt.c[0][0], Rt.c[0][1], Rt.c[0][2]}, Rt1 = {Rt.c[1][0], Rt.c[1][1], Rt.c[1][2]},
     Rt2 = {Rt.c[2][0], Rt.c[2][1], Rt.c[2][2]};
  V3 m0 = {dL_dMt.c[0][0], dL_dMt.c[0][1], dL_dMt.c[0][2]}, m1 = {dL_dMt.c[1][0], dL_dMt.c[1][1], dL_dMt.c[1][2]},
     m2 = {dL_dMt.c[2][0], dL_dMt.c[2][1], dL_dMt.c[2][2]};
  dL_dscales[3 * idx + 0] = dot(Rt0, m0);
  dL_dscales[3 * idx + 1] = dot(Rt1, m1);
  dL_dscales[3 * idx + 2] = dot(Rt2, m2);
  for (int k = 0; k < 3; k++) {
    dL_dMt.c[0][k] *= s.x;
    dL_dMt.c[1][k] *= s.y;
    dL_dMt.c[2][k] *= s.z;
  }
  const M3& D = dL_dMt;
  float qx = 2 * z * (D.c[0][1] - D.c[1][0]) + 2 * y * (D.c[2][0] - D.c[0][2]) + 2 * x * (D.c[1][2] - D.c[2][1]);
  float qy = 2 * y * (D.c[1][0] + D.c[0][1]) + 2 * z * (D.c[2][0] + D.c[0][2]) + 2 * r * (D.c[1][2] - D.c[2][1]) - 4 * x * (D.c[2][2] + D.c[1][1]);
  float qz = 2 * x * (D.c[1][0] + D.c[0][1]) + 2 * r * (D.c[2][0] - D.c[0][2]) + 2 * z * (D.c[1][2] + D.c[2][1]) - 4 * y * (D.c[2][2] + D.c[0][0]);
  float qw = 2 * r * (D.c[0][1] - D.c[1][0]) + 2 * x * (D.c[2][0] + D.c[0][2]) + 2 * y * (D.c[1][2] + D.c[2][1]) - 4 * z * (D.c[1][1] + D.c[0][0]);
  dL_drots[4 * idx + 0] = qx;
  dL_drots[4 * idx + 1] = qy;
  dL_drots[4 * idx + 2] = qz;
  dL_drots[4 * idx + 3] = qw;
}

/* ------------------------------------------------------------------------------------------------------------------
 * The reference's conic -> cov2D -> cov3D -> (scale, quaternion) chain (backward.cu:162-275 and :330-393, the statements
 * of computeCov2D_bwd / computeCov3D_bwd above) evaluated in DOUBLE from the fp32 parameters and the double sums: the
 * reference algorithm in (practically) exact arithmetic.  Arbiter for dL_dscales / dL_drotations / dL_dcov3D, for which
 * the fp32 run above is 2e-4 ... 2e-3 of the tensor's largest entry away from this image on needle-shaped footprints
 * (tests/test_gpu_fullsize.py).  Switched on with gso_set_exact_chain(1); off by default - the fp32
 * transcription stays what "the oracle" means everywhere else.
 * ------------------------------------------------------------------------------------------------------------------ */
static int g_exact_chain = 0;
struct M3d {
  double c[3][3]; /* c[col][row], as M3 */
};
static inline M3d mat3d_cols(double a0, double a1, double a2, double b0, double b1, double b2, double c0, double c1, double c2) {
  M3d m;
  m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
  m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
  m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2;
  return m;
}
static inline M3d muld(const M3d& A, const M3d& B) {
  M3d R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R.c[c][r] = A.c[0][r] * B.c[c][0] + A.c[1][r] * B.c[c][1] + A.c[2][r] * B.c[c][2];
  return R;
}
static inline M3d transposed(const M3d& A) {
  M3d R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R.c[c][r] = A.c[r][c];
  return R;
}
static inline M3d quat_to_Rd(V4 q) {
  const double r = q.x, x = q.y, y = q.z, z = q.w;
  return mat3d_cols(1. - 2. * (y * y + z * z), 2. * (x * y - r * z), 2. * (x * z + r * y), 2. * (x * y + r * z),
                    1. - 2. * (x * x + z * z), 2. * (y * z - r * x), 2. * (x * z - r * y), 2. * (y * z + r * x),
                    1. - 2. * (x * x + y * y));
}
/* dL_dconic (double sums) -> dL_dcov3D[6] (double), backward.cu:162-275 */
static void exact_chain_cov2d(int idx, const GsView* v, const GsGaussians* g, const float* cov3D_given, double h_x, double h_y,
                              const double dL_dconic[3], double dL_dopacity_v, double dL_dcov[6]) {
  const float* vm = v->viewmatrix;
  const double mx = g->means3D[3 * idx], my = g->means3D[3 * idx + 1], mz = g->means3D[3 * idx + 2];
  double tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
  double ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
  const double tz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
  const double limx = 1.3 * (double)v->tanfovx, limy = 1.3 * (double)v->tanfovy;
  tx = fmin(limx, fmax(-limx, tx / tz)) * tz;
  ty = fmin(limy, fmax(-limy, ty / tz)) * tz;
  const M3d J = mat3d_cols(h_x / tz, 0.0, -(h_x * tx) / (tz * tz), 0.0, h_y / tz, -(h_y * ty) / (tz * tz), 0, 0, 0);
  const M3d W = mat3d_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
  const M3d T = muld(W, J);
  M3d Vrk;
  if (g->scales) { /* forward.cu:114-148 in double */
    const double mod = v->scale_modifier;
    M3d S = mat3d_cols(mod * g->scales[3 * idx], 0, 0, 0, mod * g->scales[3 * idx + 1], 0, 0, 0, mod * g->scales[3 * idx + 2]);
    const V4 rq = {g->rotations[4 * idx], g->rotations[4 * idx + 1], g->rotations[4 * idx + 2], g->rotations[4 * idx + 3]};
    const M3d M = muld(S, quat_to_Rd(rq));
    Vrk = muld(transposed(M), M);
  } else {
    const float* c = cov3D_given + 6 * (size_t)idx;
    Vrk = mat3d_cols(c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]);
  }
  const M3d cov2D = muld(muld(transposed(T), transposed(Vrk)), T);
  double c_xx = cov2D.c[0][0], c_xy = cov2D.c[0][1], c_yy = cov2D.c[1][1];
  const double h_var = 0.3;
  double dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
  if (v->antialiasing) {
    const double det_cov = c_xx * c_yy - c_xy * c_xy;
    c_xx += h_var;
    c_yy += h_var;
    const double det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
    const double hcs = sqrt(fmax(0.000025, det_cov / det_cov_plus_h_cov));
    const double d_hcs = dL_dopacity_v * (double)g->opacities[idx];
    const double d_inside_root = (det_cov / det_cov_plus_h_cov) <= 0.000025 ? 0. : d_hcs / (2 * hcs);
    const double x = c_xx, y = c_yy, z = c_xy, w = h_var;
    const double q = w * w + w * (x + y) + x * y - z * z;
    const double denom_f = d_inside_root / (q * q);
    dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
    dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
    dL_dc_xy = -2. * w * z * (w + x + y) * denom_f;
  } else {
    c_xx += h_var;
    c_yy += h_var;
  }
  const double denom = c_xx * c_yy - c_xy * c_xy;
  const double denom2inv = 1.0 / ((denom * denom) + 0.0000001);
  const float denomf = (float)denom; /* the reference's own `denom2inv != 0` test is an fp32 underflow test */
  if (1.0f / ((denomf * denomf) + 0.0000001f) != 0) {
    dL_dc_xx += denom2inv * (-c_yy * c_yy * dL_dconic[0] + 2 * c_xy * c_yy * dL_dconic[1] + (denom - c_xx * c_yy) * dL_dconic[2]);
    dL_dc_yy += denom2inv * (-c_xx * c_xx * dL_dconic[2] + 2 * c_xx * c_xy * dL_dconic[1] + (denom - c_xx * c_yy) * dL_dconic[0]);
    dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dL_dconic[0] - (denom + 2 * c_xy * c_xy) * dL_dconic[1] + c_xx * c_xy * dL_dconic[2]);
    dL_dcov[0] = (T.c[0][0] * T.c[0][0] * dL_dc_xx + T.c[0][0] * T.c[1][0] * dL_dc_xy + T.c[1][0] * T.c[1][0] * dL_dc_yy);
    dL_dcov[3] = (T.c[0][1] * T.c[0][1] * dL_dc_xx + T.c[0][1] * T.c[1][1] * dL_dc_xy + T.c[1][1] * T.c[1][1] * dL_dc_yy);
    dL_dcov[5] = (T.c[0][2] * T.c[0][2] * dL_dc_xx + T.c[0][2] * T.c[1][2] * dL_dc_xy + T.c[1][2] * T.c[1][2] * dL_dc_yy);
    dL_dcov[1] = 2 * T.c[0][0] * T.c[0][1] * dL_dc_xx + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_dc_xy + 2 * T.c[1][0] * T.c[1][1] * dL_dc_yy;
    dL_dcov[2] = 2 * T.c[0][0] * T.c[0][2] * dL_dc_xx + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_dc_xy + 2 * T.c[1][0] * T.c[1][2] * dL_dc_yy;
    dL_dcov[4] = 2 * T.c[0][2] * T.c[0][1] * dL_dc_xx + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_dc_xy + 2 * T.c[1][1] * T.c[1][2] * dL_dc_yy;
  } else {
    for (int i = 0; i < 6; i++) dL_dcov[i] = 0;
  }
}
/* backward.cu:330-393 in double */
static void exact_chain_cov3d(int idx, const GsView* v, const GsGaussians* g, const double d[6], float* dL_dscales, float* dL_drots) {
  const V4 rot = {g->rotations[4 * idx], g->rotations[4 * idx + 1], g->rotations[4 * idx + 2], g->rotations[4 * idx + 3]};
  const double r = rot.x, x = rot.y, y = rot.z, z = rot.w;
  const M3d R = quat_to_Rd(rot);
  const double mod = v->scale_modifier;
  const double sx = mod * g->scales[3 * idx], sy = mod * g->scales[3 * idx + 1], sz = mod * g->scales[3 * idx + 2];
  const M3d S = mat3d_cols(sx, 0, 0, 0, sy, 0, 0, 0, sz);
  M3d M = muld(S, R);
  const M3d dL_dSigma = mat3d_cols(d[0], 0.5 * d[1], 0.5 * d[2], 0.5 * d[1], d[3], 0.5 * d[4], 0.5 * d[2], 0.5 * d[4], d[5]);
  for (int c = 0; c < 3; c++)
    for (int k = 0; k < 3; k++) M.c[c][k] *= 2.0;
  const M3d dL_dM = muld(M, dL_dSigma);
  const M3d Rt = transposed(R);
  M3d D = transposed(dL_dM);
  for (int j = 0; j < 3; j++)
    dL_dscales[3 * idx + j] = (float)(Rt.c[j][0] * D.c[j][0] + Rt.c[j][1] * D.c[j][1] + Rt.c[j][2] * D.c[j][2]);
  for (int k = 0; k < 3; k++) {
    D.c[0][k] *= sx;
    D.c[1][k] *= sy;
    D.c[2][k] *= sz;
  }
  dL_drots[4 * idx + 0] = (float)(2 * z * (D.c[0][1] - D.c[1][0]) + 2 * y * (D.c[2][0] - D.c[0][2]) + 2 * x * (D.c[1][2] - D.c[2][1]));
  dL_drots[4 * idx + 1] = (float)(2 * y * (D.c[1][0] + D.c[0][1]) + 2 * z * (D.c[2][0] + D.c[0][2]) + 2 * r * (D.c[1][2] - D.c[2][1]) - 4 * x * (D.c[2][2] + D.c[1][1]));
  dL_drots[4 * idx + 2] = (float)(2 * x * (D.c[1][0] + D.c[0][1]) + 2 * r * (D.c[2][0] - D.c[0][2]) + 2 * z * (D.c[1][2] + D.c[2][1]) - 4 * y * (D.c[2][2] + D.c[0][0]));
  dL_drots[4 * idx + 3] = (float)(2 * r * (D.c[0][1] - D.c[1][0]) + 2 * x * (D.c[2][0] + D.c[0][2]) + 2 * y * (D.c[1][2] + D.c[2][1]) - 4 * z * (D.c[1][1] + D.c[0][0]));
}

/* backward.cu:398-449 */
static void preprocess_bwd_one(int idx, int D, int M, const GsView* v, const GsGaussians* g,
                               const int* radii, const uint8_t* clamped, const float* dL_dmean2D /*[P][3]*/,
                               float* dL_dmeans, float* dL_dcolor, const float* dL_dcov3D, float* dL_dsh,
                               float* dL_dscale, float* dL_drot, const float* fs_dL_ddepth = nullptr) {
  if (!(radii[idx] > 0)) return;
  const float* proj = v->projmatrix;
  V3 m = {g->means3D[3 * idx], g->means3D[3 * idx + 1], g->means3D[3 * idx + 2]};
  V4 m_hom = transformPoint4x4(m, proj);
  float m_w = 1.0f / (m_hom.w + 0.0000001f);
  float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
  float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
  const float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
  V3 dL_dmean;
  dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
  dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
  dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
  dL_dmeans[3 * idx + 0] += dL_dmean.x;
  dL_dmeans[3 * idx + 1] += dL_dmean.y;
  dL_dmeans[3 * idx + 2] += dL_dmean.z;
  if (fs_dL_ddepth) { /* -confidence fork, backward.cu:394-403: the depth = view-space z path */
    const float* view = v->viewmatrix;
    const float mul3 = view[2] * m.x + view[6] * m.y + view[10] * m.z + view[14];
    dL_dmeans[3 * idx + 0] += (view[2] - view[3] * mul3) * fs_dL_ddepth[idx];
    dL_dmeans[3 * idx + 1] += (view[6] - view[7] * mul3) * fs_dL_ddepth[idx];
    dL_dmeans[3 * idx + 2] += (view[10] - view[11] * mul3) * fs_dL_ddepth[idx];
  }
  if (g->shs) {
    V3 campos = {v->campos[0], v->campos[1], v->campos[2]};
    computeColorFromSH_bwd(idx, D, M, g->means3D, campos, g->shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
  }
  if (g->scales) {
    V3 sc = {g->scales[3 * idx], g->scales[3 * idx + 1], g->scales[3 * idx + 2]};
    V4 rq = {g->rotations[4 * idx], g->rotations[4 * idx + 1], g->rotations[4 * idx + 2], g->rotations[4 * idx + 3]};
    computeCov3D_bwd(idx, sc, v->scale_modifier, rq, dL_dcov3D, dL_dscale, dL_drot);
  }
}

/* Stage 2 of the backward: computeCov2DCUDA + preprocessCUDA backward (backward.cu:147-449) from the per-Gaussian sums
 * of the blend backward, rows [P][16] = mean2D.x, mean2D.y, conic.xx, conic.xy, conic.yy, opacity, r, g, b, depth slot.
 * depth_mode: 0 none, 1 inverse depth (dr_aa), 2 depth (FSGS generation). */
static void stage2_from_rows(const GsView* v, const GsGaussians* g, const int32_t* radii, const Geom& gs,
                             const double* rows, int depth_mode, const GsGrads* out) {
  const int P = g->P, W = v->image_width, H = v->image_height, M = g->M;
  const float focal_y = H / (2.0f * v->tanfovy);
  const float focal_x = W / (2.0f * v->tanfovx);
  std::vector<float> dL_dmean2D((size_t)P * 3, 0.f), dL_dconic((size_t)P * 3, 0.f), dL_dopacity(P, 0.f),
      dL_dcolors((size_t)P * 3, 0.f), dL_dinvd(P, 0.f), dL_dmeans3D((size_t)P * 3, 0.f), dL_dcov3D((size_t)P * 6, 0.f),
      dL_dsh((size_t)P * std::max(M, 1) * 3, 0.f), dL_dscales((size_t)P * 3, 0.f), dL_drot((size_t)P * 4, 0.f);
  for (int i = 0; i < P; i++) {
    /* the reference's gradient tensors are fp32 (rasterize_points.cu:163-178): each sum is rounded ONCE here */
    const double* a = rows + (size_t)i * 16;
    dL_dmean2D[3 * i] = (float)a[A_MX];
    dL_dmean2D[3 * i + 1] = (float)a[A_MY];
    dL_dconic[3 * i] = (float)a[A_CXX];
    dL_dconic[3 * i + 1] = (float)a[A_CXY];
    dL_dconic[3 * i + 2] = (float)a[A_CYY];
    dL_dopacity[i] = (float)a[A_OP];
    dL_dcolors[3 * i] = (float)a[A_CR];
    dL_dcolors[3 * i + 1] = (float)a[A_CG];
    dL_dcolors[3 * i + 2] = (float)a[A_CB];
    dL_dinvd[i] = (float)a[A_ID];
  }
  const float* cov3D_ptr = g->cov3D_precomp ? g->cov3D_precomp : gs.cov3D;
  const float* dinvd_ptr = depth_mode == 1 ? dL_dinvd.data() : nullptr;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++)
    computeCov2D_bwd(i, v, g->means3D, radii, cov3D_ptr, focal_x, focal_y, g->opacities, dL_dconic.data(),
                     dL_dopacity.data(), dinvd_ptr, dL_dmeans3D.data(), dL_dcov3D.data());
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++)
    preprocess_bwd_one(i, v->sh_degree, M, v, g, radii, gs.clamped, dL_dmean2D.data(), dL_dmeans3D.data(),
                       dL_dcolors.data(), dL_dcov3D.data(), dL_dsh.data(), dL_dscales.data(), dL_drot.data(),
                       depth_mode == 2 ? dL_dinvd.data() : nullptr);
  if (g_exact_chain) { /* dL_dcov3D / dL_dscales / dL_drotations: the same statements in double (see exact_chain_*) */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
      if (!(radii[i] > 0)) continue;
      const double* a = rows + (size_t)i * 16;
      const double dconic[3] = {a[A_CXX], a[A_CXY], a[A_CYY]};
      double dcov[6];
      exact_chain_cov2d(i, v, g, cov3D_ptr, (double)focal_x, (double)focal_y, dconic, a[A_OP], dcov);
      for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = (float)dcov[k];
      if (g->scales) exact_chain_cov3d(i, v, g, dcov, dL_dscales.data(), dL_drot.data());
    }
  }
  if (out->dL_dmeans3D) memcpy(out->dL_dmeans3D, dL_dmeans3D.data(), 12 * (size_t)P);
  if (out->dL_dmeans2D) memcpy(out->dL_dmeans2D, dL_dmean2D.data(), 12 * (size_t)P);
  if (out->dL_dsh && M > 0) memcpy(out->dL_dsh, dL_dsh.data(), 12 * (size_t)P * M);
  if (out->dL_dcolors) memcpy(out->dL_dcolors, dL_dcolors.data(), 12 * (size_t)P);
  if (out->dL_dopacity) memcpy(out->dL_dopacity, dL_dopacity.data(), 4 * (size_t)P);
  if (out->dL_dscales) memcpy(out->dL_dscales, dL_dscales.data(), 12 * (size_t)P);
  if (out->dL_drotations) memcpy(out->dL_drotations, dL_drot.data(), 16 * (size_t)P);
  if (out->dL_dcov3D) memcpy(out->dL_dcov3D, dL_dcov3D.data(), 24 * (size_t)P);
}

int gso_backward_from_rows(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* s,
                           const double* rows, int32_t depth_mode, const GsGrads* out, void* /*workspace*/,
                           size_t /*workspace_bytes*/, void* /*stream*/) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!s || !out) return GS_E_NULL;
  if (depth_mode < 0 || depth_mode > 2) return GS_E_SHAPE;
  if (g->P == 0) return GS_OK;
  if (!radii || !s->geom || !rows) return GS_E_NULL;
  Geom gs = geom_from(s->geom, g->P);
  stage2_from_rows(v, g, radii, gs, rows, depth_mode, out);
  return GS_OK;
}

static int backward_impl(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* s,
                         int64_t num_rendered, const float* dL_dcolor_img, const float* dL_dinvdepth_img,
                         const float* fs_dL_ddepth, const float* fs_dL_dalpha, const GsGrads* out,
                         void* ws = nullptr, size_t ws_bytes = 0);

int gso_backward(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* s,
                 int64_t num_rendered, const float* dL_dcolor_img, const float* dL_dinvdepth_img,
                 const GsGrads* out, void* ws, size_t ws_bytes, void* /*stream*/) {
  return backward_impl(v, g, radii, s, num_rendered, dL_dcolor_img, dL_dinvdepth_img, nullptr, nullptr, out, ws, ws_bytes);
}

/* dgr_fsgs `rasterize_gaussians_backward`: image gradients of colour, depth and alpha. */
int gso_backward_fsgs(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* s,
                      int64_t num_rendered, const float* dL_dcolor_img, const float* dL_ddepth_img,
                      const float* dL_dalpha_img, const GsGrads* out, void* /*ws*/, size_t /*ws_bytes*/,
                      void* /*stream*/) {
  if (!dL_ddepth_img || !dL_dalpha_img) return GS_E_NULL;
  if (v && v->antialiasing) return GS_E_UNSUPPORTED;
  return backward_impl(v, g, radii, s, num_rendered, dL_dcolor_img, nullptr, dL_ddepth_img, dL_dalpha_img, out);
}

static int backward_impl(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* s,
                         int64_t num_rendered, const float* dL_dcolor_img, const float* dL_dinvdepth_img,
                         const float* fs_dL_ddepth, const float* fs_dL_dalpha, const GsGrads* out, void* ws,
                         size_t ws_bytes) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!s || !out || !dL_dcolor_img) return GS_E_NULL;
  const int P = g->P, W = v->image_width, H = v->image_height;
  if (P == 0) return GS_OK;
  if (!radii || !s->geom || !s->img) return GS_E_NULL;
  const uint32_t gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  Geom gs = geom_from(s->geom, P);
  Img im = img_from(s->img, N, T);
  if (gs.hdr->num_rendered != num_rendered) return GS_E_SHAPE;
  Binning b = binning_from(s->binning, (size_t)std::max<int64_t>(s->binning_capacity, 1));
  const float* color_ptr = g->colors_precomp ? g->colors_precomp : gs.rgb;

  /* zero-initialised gradient tensors, rasterize_points.cu:163-178 */
  std::vector<double> acc((size_t)P * A_N, 0.0);
  if (num_rendered > 0) {
#pragma omp parallel
    {
      std::vector<double> loc;
      std::vector<uint8_t> touched;
#pragma omp for schedule(dynamic, 1) collapse(2)
      for (uint32_t ty = 0; ty < gy; ty++)
        for (uint32_t tx = 0; tx < gx; tx++) {
          render_tile_bwd(tx, ty, gx, W, H, im, b.point_list, gs, color_ptr, v->bg, dL_dcolor_img,
                          dL_dinvdepth_img, loc, touched, fs_dL_ddepth, fs_dL_dalpha);
          const uint32_t r0 = im.ranges[2 * (ty * gx + tx)];
          for (size_t k = 0; k < touched.size(); k++)
            if (touched[k]) {
              const uint32_t id = b.point_list[r0 + k];
              for (int a = 0; a < A_N; a++) {
                double val = loc[k * A_N + a];
#pragma omp atomic
                acc[(size_t)id * A_N + a] += val;
              }
            }
        }
    }
  }
  /* the per-Gaussian sums (double accumulators; stage 2 rounds each once to the reference's fp32 gradient tensors) in
   * the product's row layout: 16 float64 slots per Gaussian */
  std::vector<double> rows_own;
  double* rows = nullptr;
  if (ws && ws_bytes >= (size_t)P * 16 * sizeof(double)) {
    rows = (double*)ws; /* parity probe: a test can compare this intermediate with the product's workspace */
  } else {
    rows_own.resize((size_t)P * 16);
    rows = rows_own.data();
  }
  for (int i = 0; i < P; i++) {
    const double* a = &acc[(size_t)i * A_N];
    for (int k = 0; k < 16; k++) rows[(size_t)i * 16 + k] = k < A_N ? a[k] : 0.0;
  }
  stage2_from_rows(v, g, radii, gs, rows, fs_dL_dalpha ? 2 : (dL_dinvdepth_img ? 1 : 0), out);
  return GS_OK;
}

/* rasterizer_impl.cu:54-66,141-153 */
int gso_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* /*proj*/,
                     uint8_t* present, void* /*stream*/) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!means3D || !viewmatrix || !present) return GS_E_NULL;
  for (int i = 0; i < P; i++) {
    V3 p = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    V3 pv = transformPoint4x3(p, viewmatrix);
    present[i] = !(pv.z <= 0.2f);
  }
  return GS_OK;
}

int gso_export_geom(const GsScratch* s, int32_t P, float* depths, float* means2D, float* cov3D,
                    float* conic_opacity, float* rgb, uint8_t* clamped, uint32_t* tiles_touched,
                    uint32_t* point_offsets, void* /*stream*/) {
  if (!s || !s->geom) return GS_E_NULL;
  Geom gs = geom_from(s->geom, P);
  size_t p = (size_t)P;
  /* rows of culled Gaussians were never written by the reference either; export zeros there so
   * that comparisons are well defined */
  for (size_t i = 0; i < p; i++) {
    bool vis = gs.radii[i] > 0;
    if (depths) depths[i] = vis ? gs.depths[i] : 0.f;
    for (int k = 0; k < 2; k++) if (means2D) means2D[2 * i + k] = vis ? gs.means2D[2 * i + k] : 0.f;
    for (int k = 0; k < 4; k++) if (conic_opacity) conic_opacity[4 * i + k] = vis ? gs.conic_opacity[4 * i + k] : 0.f;
    for (int k = 0; k < 3; k++) if (rgb) rgb[3 * i + k] = vis ? gs.rgb[3 * i + k] : 0.f;
    for (int k = 0; k < 3; k++) if (clamped) clamped[3 * i + k] = vis ? gs.clamped[3 * i + k] : 0;
    for (int k = 0; k < 6; k++) if (cov3D) cov3D[6 * i + k] = vis ? gs.cov3D[6 * i + k] : 0.f;
  }
  if (tiles_touched) memcpy(tiles_touched, gs.tiles_touched, 4 * p);
  if (point_offsets) memcpy(point_offsets, gs.point_offsets, 4 * p);
  return GS_OK;
}
int gso_export_binning(const GsScratch* s, int64_t R, uint64_t* keys_sorted, uint32_t* point_list, void*) {
  if (!s) return GS_E_NULL;
  if (R == 0) return GS_OK;
  if (!s->binning) return GS_E_NULL;
  Binning b = binning_from(s->binning, (size_t)std::max<int64_t>(s->binning_capacity, 1));
  if (keys_sorted) memcpy(keys_sorted, b.keys, 8 * (size_t)R);
  if (point_list) memcpy(point_list, b.point_list, 4 * (size_t)R);
  return GS_OK;
}
int gso_export_img(const GsScratch* s, int32_t W, int32_t H, float* final_T, uint32_t* n_contrib,
                   uint32_t* ranges, void*) {
  if (!s || !s->img) return GS_E_NULL;
  const uint32_t gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  Img im = img_from(s->img, N, T);
  if (final_T) memcpy(final_T, im.accum_alpha, 4 * N);
  if (n_contrib) memcpy(n_contrib, im.n_contrib, 4 * N);
  if (ranges) memcpy(ranges, im.ranges, 8 * T);
  return GS_OK;
}

/* ---- test-only probes of single device functions (golden-vector pinning) ---- */
int gso_test_sh_fwd(int32_t P, int32_t deg, int32_t M, const float* means, const float* campos, const float* shs,
                    float* rgb, uint8_t* clamped) {
  V3 cp = {campos[0], campos[1], campos[2]};
  for (int i = 0; i < P; i++) {
    V3 r = computeColorFromSH(i, deg, M, means, cp, shs, clamped);
    rgb[3 * i] = r.x; rgb[3 * i + 1] = r.y; rgb[3 * i + 2] = r.z;
  }
  return GS_OK;
}
int gso_test_sh_bwd(int32_t P, int32_t deg, int32_t M, const float* means, const float* campos, const float* shs,
                    const uint8_t* clamped, const float* dL_dcolor, float* dL_dmeans /* += */, float* dL_dsh) {
  V3 cp = {campos[0], campos[1], campos[2]};
  for (int i = 0; i < P; i++) computeColorFromSH_bwd(i, deg, M, means, cp, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
  return GS_OK;
}

/* computeCov3D forward / backward (forward.cu:114-148, backward.cu:330-393) and the homogeneous projection of
 * preprocessCUDA (forward.cu:193-195) on their own: pinned by tests/golden/geometry.npz, which the reference's python
 * covariance path (utils/general_utils.py:64-110, scene/gaussian_model.py:33-37) and geom_transform_points
 * (utils/graphics_utils.py:22-29) produced. */
int gso_test_cov3d_fwd(int32_t P, const float* scales, float mod, const float* rots, float* cov3D) {
  for (int i = 0; i < P; i++) {
    V3 sc = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    V4 rq = {rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]};
    computeCov3D(sc, mod, rq, cov3D + 6 * (size_t)i);
  }
  return GS_OK;
}
int gso_test_cov3d_bwd(int32_t P, const float* scales, float mod, const float* rots, const float* dL_dcov3D,
                       float* dL_dscales, float* dL_drots) {
  for (int i = 0; i < P; i++) {
    V3 sc = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    V4 rq = {rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]};
    computeCov3D_bwd(i, sc, mod, rq, dL_dcov3D, dL_dscales, dL_drots);
  }
  return GS_OK;
}
int gso_test_project(int32_t P, const float* means, const float* projmatrix, float* p_proj) {
  for (int i = 0; i < P; i++) {
    V3 p = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    V4 h = transformPoint4x4(p, projmatrix);
    float p_w = 1.0f / (h.w + 0.0000001f);
    p_proj[3 * i] = h.x * p_w; p_proj[3 * i + 1] = h.y * p_w; p_proj[3 * i + 2] = h.z * p_w;
  }
  return GS_OK;
}

/* thread control for the cpu_baseline leg of bench.py ("cores" = threads actually used) */
int gso_set_fsgs_exact_T(int32_t on) {
  g_fsgs_exact_T = on != 0;
  return GS_OK;
}

/* 1: stage 2 evaluates the conic -> covariance -> (scale, quaternion) chain in double (exact_chain_*): the arbiter for
 * dL_dscales / dL_drotations / dL_dcov3D.  Returns the previous setting. */
int gso_set_exact_chain(int32_t on) {
  const int old = g_exact_chain;
  g_exact_chain = on != 0;
  return old;
}

int gso_set_num_threads(int32_t n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

} /* extern "C" */
