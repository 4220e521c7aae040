// gs_preprocess.hip - forward per-Gaussian stage (HBM-bound streaming kernel).
// Compiled with -ffp-contract=off: see gs_math.h.
//
// Replaces preprocessCUDA<3> (forward.cu:151-269) + checkFrustum (rasterizer_impl.cu:54-66) and
// produces the per-block partial sums for the tiles_touched prefix sum (cub InclusiveSum,
// rasterizer_impl.cu:280).  One thread per Gaussian, 256 per workgroup; each Gaussian costs
// 236 B of reads (12 xyz + 12 scale + 16 quat + 4 opacity + 192 SH) and 96 B of writes
// (64-B splat record, 24 B cov3D, 4 B tiles, 4 B radii).
#include "gs_common.h"
#include "gs_tilecull.h"
#include "gs_math.h"

// forward.cu:20-71.  sh points at this Gaussian's coefficients as 3*M floats.
template <typename SH>
GS_DEV V3 color_from_sh(int deg, V3 pos, V3 campos, const SH& sh, uint32_t& clamped) {
  V3 dir = pos - campos;
  dir = dir / length3(dir);
  V3 result = SH_C0 * sh(0);
  if (deg > 0) {
    float x = dir.x, y = dir.y, z = dir.z;
    result = result - SH_C1 * y * sh(1) + SH_C1 * z * sh(2) - SH_C1 * x * sh(3);
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z;
      float xy = x * y, yz = y * z, xz = x * z;
      result = result + SH_C2_0 * xy * sh(4) + SH_C2_1 * yz * sh(5) + SH_C2_2 * (2.0f * zz - xx - yy) * sh(6) +
               SH_C2_3 * xz * sh(7) + SH_C2_4 * (xx - yy) * sh(8);
      if (deg > 2) {
        result = result + SH_C3_0 * y * (3.0f * xx - yy) * sh(9) + SH_C3_1 * xy * z * sh(10) +
                 SH_C3_2 * y * (4.0f * zz - xx - yy) * sh(11) + SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh(12) +
                 SH_C3_4 * x * (4.0f * zz - xx - yy) * sh(13) + SH_C3_5 * z * (xx - yy) * sh(14) +
                 SH_C3_6 * x * (xx - 3.0f * yy) * sh(15);
      }
    }
  }
  result.x += 0.5f;
  result.y += 0.5f;
  result.z += 0.5f;
  clamped = (result.x < 0 ? 1u : 0u) | (result.y < 0 ? 2u : 0u) | (result.z < 0 ? 4u : 0u);
  return {fmaxf(result.x, 0.0f), fmaxf(result.y, 0.0f), fmaxf(result.z, 0.0f)};
}

struct ShRegs {  // coefficients held in registers, loaded as float4 (M == 16 layout, 192 B per Gaussian)
  float f[48];
  __device__ __forceinline__ V3 operator()(int k) const { return {f[3 * k], f[3 * k + 1], f[3 * k + 2]}; }
};
struct ShMem {  // generic M: scalar loads
  const float* p;
  __device__ __forceinline__ V3 operator()(int k) const { return {p[3 * k], p[3 * k + 1], p[3 * k + 2]}; }
};
struct ShMemSplit {  // generic M, split rows (GsGaussians.shs_rest)
  const float *dc, *rest;
  __device__ __forceinline__ V3 operator()(int k) const {
    const float* p = k == 0 ? dc : rest + 3 * (k - 1);
    return {p[0], p[1], p[2]};
  }
};

// Depth limits (gs_tilecull.h): the look-ups of a Gaussian have nothing to do with its neighbours' in the wave - fully
// divergent loads, served one lane at a time by the vector memory path.  The tables are small, so each workgroup first
// copies them into LDS (coalesced, out of L2) and the look-ups become ds_reads: always the segment table (1 B per tile),
// the per-tile table (4 B per tile, 32 KB at 1080p) only if GS_LIMIT_TILES_IN_LDS - measured no better (C3:
// preprocess 0.156 vs 0.150 ms; 40 KB of LDS cost a resident workgroup).  Dynamic LDS: nothing is allocated when there
// are no limits.
#ifndef GS_LIMIT_TILES_IN_LDS
#define GS_LIMIT_TILES_IN_LDS 0
#endif
#define GS_LIMIT_LDS_MAX_FLOATS (GS_LIMIT_TILES_IN_LDS ? 10240 : 4096)
extern __shared__ float s_limit[];
static inline size_t preprocess_limit_lds_floats(const PreprocessArgs& a) {
  if (!a.tile_depth_limit) return 0;
  const size_t T = (size_t)a.grid_x * a.grid_y, all = depth_limit_floats((uint32_t)a.grid_x, (uint32_t)a.grid_y);
  const size_t want = GS_LIMIT_TILES_IN_LDS ? all : all - T;
  return want <= GS_LIMIT_LDS_MAX_FLOATS ? want : 0;
}

__global__ void __launch_bounds__(GS_BLOCK) preprocess_fwd_kernel(PreprocessArgs a, GeomView g, int lds_floats) {
  const int idx = blockIdx.x * GS_BLOCK + threadIdx.x;
  uint32_t tiles = 0, entries = TB_ENTRIES_UNKNOWN;
  const int T = a.grid_x * a.grid_y;
  // region binning: up to four bucket counters waiting to be bumped (selects, not an indexed array: that would live in
  // scratch memory), the slots they returned, and this Gaussian's bucket entry.
  // (Round 4 tried the atomics per WAVE - lanes that bump the same counter form a group, its first lane adds the group's
  //  size, members take base + rank through a shuffle -: correct, and slower with rows in random order (0.114 -> 0.134 ms)
  //  AND in spatial order (0.120 -> 0.128 ms, where a wave shares a handful of regions): the scalar grouping loop and the
  //  shuffles cost more than the returning atomics they save.)
  int p0 = 0, p1 = 0, p2 = 0, p3 = 0, n_pend = 0;
  uint32_t s0 = 0xFFFFFFFFu, s1 = 0xFFFFFFFFu, s2 = 0xFFFFFFFFu, s3 = 0xFFFFFFFFu, dbits = 0;
  auto bump = [&]() {
    if (n_pend > 0) s0 = atomicAdd(&a.region_count[(size_t)p0 * RG_COUNT_STRIDE], 1u);
    if (n_pend > 1) s1 = atomicAdd(&a.region_count[(size_t)p1 * RG_COUNT_STRIDE], 1u);
    if (n_pend > 2) s2 = atomicAdd(&a.region_count[(size_t)p2 * RG_COUNT_STRIDE], 1u);
    if (n_pend > 3) s3 = atomicAdd(&a.region_count[(size_t)p3 * RG_COUNT_STRIDE], 1u);
    n_pend = 0;
  };
  auto collect = [&]() {
    const uint2 entry = make_uint2(dbits, (uint32_t)idx);
    if (s0 < a.region_cap) a.region_bucket[(size_t)p0 * a.region_cap + s0] = entry;
    if (s1 < a.region_cap) a.region_bucket[(size_t)p1 * a.region_cap + s1] = entry;
    if (s2 < a.region_cap) a.region_bucket[(size_t)p2 * a.region_cap + s2] = entry;
    if (s3 < a.region_cap) a.region_bucket[(size_t)p3 * a.region_cap + s3] = entry;
    s0 = s1 = s2 = s3 = 0xFFFFFFFFu;
  };
  const int lidx = min(idx, a.P - 1);  // (the last workgroup's spare lanes load the last Gaussian's rows and drop them)
  // Every small row of this Gaussian is requested at once, ahead of the frustum test and of the limit table's copy:
  // five dependent round trips per wave (table -> mean -> scale, rotation -> opacity -> SH row) become two, for 32 B
  // more per Gaussian behind the camera.
  // The camera (two 4 x 4 matrices, the camera centre): the same 35 floats for every lane.  Read through constant-address-
  // space pointers they become scalar loads into SGPRs, issued here with everything else - as plain global pointers the
  // compiler reads them with vector loads where they are first used: two more dependent round trips per wave.
  typedef const __attribute__((address_space(4))) float* GsConstFloatPtr;
  float vm[16], pm[16], cam[3];
  {
    const GsConstFloatPtr vmp = (GsConstFloatPtr)(uintptr_t)a.viewmatrix, pmp = (GsConstFloatPtr)(uintptr_t)a.projmatrix;
    const GsConstFloatPtr cp = (GsConstFloatPtr)(uintptr_t)a.campos;
#pragma unroll
    for (int k = 0; k < 16; k++) { vm[k] = vmp[k]; pm[k] = pmp[k]; }
    cam[0] = cp[0]; cam[1] = cp[1]; cam[2] = cp[2];
  }
  const bool own_cov = a.cov3D_precomp == nullptr;
  // (no branch around these loads - the merge of its two arms would wait for them -: with a precomputed covariance the two
  // pointers aim at that row instead and the values are dropped)
  const float* sc_p = own_cov ? a.scales + 3 * (size_t)lidx : a.cov3D_precomp + 6 * (size_t)lidx;
  const float* rq_p = own_cov ? a.rotations + 4 * (size_t)lidx : a.cov3D_precomp + 6 * (size_t)lidx;
  float op_in = a.opacities[lidx];
  V3 p_orig = {a.means3D[3 * (size_t)lidx], a.means3D[3 * (size_t)lidx + 1], a.means3D[3 * (size_t)lidx + 2]};
  V3 sc_in = {sc_p[0], sc_p[1], sc_p[2]};
  V4 rq_in = {rq_p[0], rq_p[1], rq_p[2], rq_p[3]};
  if (lds_floats) {  // the tail of the limit buffer (segments), or all of it
    const float* src = a.tile_depth_limit + (GS_LIMIT_TILES_IN_LDS ? 0 : T);
    for (int i = threadIdx.x; i < lds_floats; i += GS_BLOCK) s_limit[i] = src[i];
    __syncthreads();
  }
  // (keeps the compiler from sinking these loads to their uses, behind the frustum branch)
  asm volatile("" : "+v"(op_in), "+v"(sc_in.x), "+v"(sc_in.y), "+v"(sc_in.z), "+v"(rq_in.x), "+v"(rq_in.y), "+v"(rq_in.z),
               "+v"(rq_in.w));
  const GsLdsFloatPtr lds_seg = (GsLdsFloatPtr)s_limit + (GS_LIMIT_TILES_IN_LDS ? T : 0);
  if (idx < a.P) {
    Splat sp;
    sp.x = sp.y = sp.depth = sp.invdepth = 0.f;
    sp.cxx = sp.cxy = sp.cyy = sp.opacity = 0.f;
    sp.r = sp.g = sp.b = 0.f;
    sp.extra = 0.f;
    int radius_out = 0;
    sp.rect_min = sp.rect_max = sp.tiles = sp.clamped = 0;
    float cov3D[6] = {0, 0, 0, 0, 0, 0};
    bool write_cov = false;
    do {
      // in_frustum, auxiliary.h:151-176
      V3 p_view = xform4x3(p_orig, vm);
      if (p_view.z <= 0.2f) break;
      V4 p_hom = xform4x4(p_orig, pm);
      float p_w = 1.0f / (p_hom.w + 0.0000001f);
      V3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};

      if (a.cov3D_precomp != nullptr) {
#pragma unroll
        for (int k = 0; k < 6; k++) cov3D[k] = a.cov3D_precomp[(size_t)idx * 6 + k];
      } else {
        // computeCov3D, forward.cu:114-148
        const V3 sc = activate_scales(sc_in, a.raw_activations);
        const V4 rq = activate_rotation(rq_in, a.raw_activations);
        M3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
        S.c[0][0] = a.scale_modifier * sc.x;
        S.c[1][1] = a.scale_modifier * sc.y;
        S.c[2][2] = a.scale_modifier * sc.z;
        M3 R = quat_to_R(rq);
        M3 Mm = mul3(S, R);
        M3 Sigma = mul3(transpose3(Mm), Mm);
        cov3D[0] = Sigma.c[0][0];
        cov3D[1] = Sigma.c[0][1];
        cov3D[2] = Sigma.c[0][2];
        cov3D[3] = Sigma.c[1][1];
        cov3D[4] = Sigma.c[1][2];
        cov3D[5] = Sigma.c[2][2];
        write_cov = true;
      }
      // computeCov2D, forward.cu:74-109
      Cov2DInter ci;
      cov2d_common(p_orig, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov3D, vm, ci);
      M3 cv = mul3(mul3(transpose3(ci.T), transpose3(ci.Vrk)), ci.T);
      V3 cov = {cv.c[0][0], cv.c[0][1], cv.c[1][1]};

      const float h_var = 0.3f;
      const float det_cov = cov.x * cov.z - cov.y * cov.y;
      cov.x += h_var;
      cov.z += h_var;
      const float det_cov_plus_h_cov = cov.x * cov.z - cov.y * cov.y;
      float h_convolution_scaling = 1.0f;
      if (a.antialiasing) h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
      const float det = det_cov_plus_h_cov;
      if (det == 0.0f) break;
      float det_inv = 1.f / det;
      V3 conic = {cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv};
      float mid = 0.5f * (cov.x + cov.z);
      float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
      float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
      float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
      float pix_x = ndc2pix(p_proj.x, a.W), pix_y = ndc2pix(p_proj.y, a.H);
      uint32_t minx, miny, maxx, maxy;
      const int radius_i = f2i_sat(my_radius);
      get_rect(pix_x, pix_y, radius_i, a.grid_x, a.grid_y, minx, miny, maxx, maxy);
      if ((maxx - minx) * (maxy - miny) == 0) break;

      sp.depth = p_view.z;
      sp.invdepth = 1 / p_view.z;
      radius_out = radius_i;
      sp.extra = a.extra_channel ? load_extra(a.extra_channel, a.extra_gain, idx, a.raw_activations) : 0.f;
      sp.x = pix_x;
      sp.y = pix_y;
      sp.cxx = conic.x; sp.cxy = conic.y; sp.cyy = conic.z;
      sp.opacity = activate_opacity(op_in, a.raw_activations) * h_convolution_scaling;
      sp.rect_min = minx | (miny << 16);
      sp.rect_max = maxx | (maxy << 16);
      if (a.tile_cull == 2) {
        // Region binning (gs_regionbin.hip): into the bucket of every 4 x 4-tile region that the bounding box of the
        // alpha >= 1/255 ellipse reaches (inside the reference rectangle) and, with depth limits, whose largest tile bound the
        // Gaussian is not beyond.  `tiles` counts the regions: non-zero = "may have instances" for everything downstream.
        const TileCull tc = tilecull_setup(1, sp.x, sp.y, sp.cxx, sp.cxy, sp.cyy, sp.opacity);
        if (tc.mode != 1) {
          int tx0 = (int)minx, tx1 = (int)maxx, ty0 = (int)miny, ty1 = (int)maxy;  // tile rectangle [t0, t1)
          if (tc.mode == 2) {  // one pixel of slack around the ellipse's extents (tilecull_row_span: 0.05 px + 1e-5 relative)
            const float big = 1.0e6f;
            const float xl = fminf(fmaxf(sp.x - tc.ex - 1.0f, -big), big), xh = fminf(fmaxf(sp.x + tc.ex + 1.0f, -big), big);
            const float yl = fminf(fmaxf(sp.y - tc.ey - 1.0f, -big), big), yh = fminf(fmaxf(sp.y + tc.ey + 1.0f, -big), big);
            tx0 = max(tx0, (int)floorf(xl * 0.0625f));
            tx1 = min(tx1, (int)floorf(xh * 0.0625f) + 1);
            ty0 = max(ty0, (int)floorf(yl * 0.0625f));
            ty1 = min(ty1, (int)floorf(yh * 0.0625f) + 1);
          }
          if (tx1 > tx0 && ty1 > ty0) {
            const int segs_x = (int)depth_limit_segs_x((uint32_t)a.grid_x);
            // A returning device-scope atomic is the slowest thing this kernel does (about 5 us under load at C3; measured: a
            // second set of them costs 16 us of the kernel's 120).  So: never one after the other - the bucket counters of a
            // Gaussian are bumped four at a time, back to back (a footprint within 2 x 2 regions, nearly all of them, has one
            // such batch; taking the regions beyond the fourth one by one cost 29 us, the wave waiting for its largest
            // Gaussian) - and the last batch is issued behind the loads of the SH row and collected after the colour is
            // computed, so that its round trip hides under that of the coefficients (`bump` / `collect` below).
            dbits = __float_as_uint(sp.depth);
            for (int ry = ty0 >> 2; ry <= (ty1 - 1) >> 2; ry++) {
              for (int rx = tx0 >> 2; rx <= (tx1 - 1) >> 2; rx++) {
                if (a.tile_depth_limit) {
                  // the region's bound = the largest of its (up to) four row-segment bounds: a pair within its TILE's
                  // bound lies in a region it is not beyond
                  float bound = -__builtin_inff();
#pragma unroll
                  for (int k = 0; k < RG_TILES; k++) {
                    const int ty = min(ry * RG_TILES + k, a.grid_y - 1);
                    const float v = lds_floats ? lds_seg[ty * segs_x + rx] : a.tile_depth_limit[(size_t)T + ty * segs_x + rx];
                    bound = fmaxf(bound, v);
                  }
                  if (depth_beyond_limit(sp.depth, bound)) continue;
                }
                const int r = ry * a.rg_x + rx;
                tiles++;
                if (n_pend == 4) { bump(); collect(); }
                p0 = n_pend == 0 ? r : p0;
                p1 = n_pend == 1 ? r : p1;
                p2 = n_pend == 2 ? r : p2;
                p3 = n_pend == 3 ? r : p3;
                n_pend++;
              }
            }
          }
        }
      } else if (a.tile_cull) {
        const TileCull tc = tilecull_setup(1, sp.x, sp.y, sp.cxx, sp.cxy, sp.cyy, sp.opacity);
        if (tc.mode == 0) {
          tiles = (maxy - miny) * (maxx - minx);
          if (!a.tile_depth_limit) entries = tb_rect_entries(minx, miny, maxx, maxy);
        } else if (tc.mode == 2) {
          // the tiles the ellipse really reaches (far fewer than the bounding square's), their bounding box and - with
          // depth limits - how many of them survive the segment rule (gs_tilecull.h)
          uint32_t bx0 = 0xFFFFFFFFu, bx1 = 0, by0 = 0xFFFFFFFFu, by1 = 0, tiles_seg = 0;
          // (+ the region entries of the two-level binning, gs_tilebin.hip: per region row the hull of its four tile rows'
          //  spans, or - should the hull hold more regions than the Gaussian has tiles - per tile row; without depth limits only:
          //  with them the spans are trimmed further down and the binning counts for itself)
          uint32_t hull = 0, rowwise = 0, h_lo = 0xFFFFFFFFu, h_hi = 0;
          for (uint32_t ty = miny; ty < maxy; ty++) {
            uint32_t tx0;
            const uint32_t n = tilecull_row_span(tc, ty, minx, maxx, tx0);
            tiles += n;
            if (n) {
              h_lo = min(h_lo, tx0); h_hi = max(h_hi, tx0 + n - 1u);
              rowwise += (tx0 + n - 1u) / RG_TILES - tx0 / RG_TILES + 1u;
            }
            if ((ty % RG_TILES) == RG_TILES - 1u || ty + 1u == maxy) {  // the region row ends here
              if (h_lo <= h_hi) hull += h_hi / RG_TILES - h_lo / RG_TILES + 1u;
              h_lo = 0xFFFFFFFFu; h_hi = 0;
            }
            if (n && a.tile_depth_limit) {
              bx0 = min(bx0, tx0); bx1 = max(bx1, tx0 + n);
              by0 = min(by0, ty); by1 = ty + 1;
              if (lds_floats) {
                uint32_t t0 = tx0;
                tiles_seg += tilecull_trim_span_segments(lds_seg, (uint32_t)a.grid_x, ty, sp.depth, t0, n);
              }
            }
          }
          if (!a.tile_depth_limit) entries = hull > tiles ? (rowwise | 0x8000u) : hull;
          if (a.tile_depth_limit && tiles) {
            // The verdict travels to the duplicate kernel in the record (bits 8-9 of `clamped`): 0 nothing cut, 1 all,
            // 2 cut by the exact rule (Gaussians inside a 4 x 4 tile box: their <= 16 bounds fetched at once, rows then
            // need no loads), 3 cut by the segment rule (larger Gaussians; only with the LDS table).
            const uint32_t full = tiles;
            uint32_t how = 0;
            if (bx1 - bx0 <= 4u && by1 - by0 <= 4u) {
              uint32_t beyond;
              if (GS_LIMIT_TILES_IN_LDS && lds_floats)
                beyond = depth_limit_box_mask((GsLdsFloatPtr)s_limit, (uint32_t)a.grid_x, bx0, by0, bx1, by1, sp.depth);
              else
                beyond = depth_limit_box_mask(a.tile_depth_limit, (uint32_t)a.grid_x, bx0, by0, bx1, by1, sp.depth);
              if (beyond) {
                tiles = 0;
                for (uint32_t ty = by0; ty < by1; ty++) {
                  uint32_t tx0;
                  const uint32_t n = tilecull_row_span(tc, ty, minx, maxx, tx0);
                  tiles += tilecull_trim_span_mask(beyond, bx0, by0, ty, tx0, n);
                }
                how = 2;
              }
            } else if (lds_floats) {
              tiles = tiles_seg;
              how = 3;
            }
            sp.clamped |= (tiles == 0 ? 1u : (tiles == full ? 0u : how)) << 8;
          }
        }
      } else {
        tiles = (maxy - miny) * (maxx - minx);
        entries = tb_rect_entries(minx, miny, maxx, maxy);
      }
      // bits 16-31 of `clamped`: region entries of this Gaussian (bit 15 of them: row-wise enumeration), TB_ENTRIES_UNKNOWN when
      // the binning has to count for itself
      sp.clamped |= (tiles ? entries : 0u) << 16;
      // the colour comes last: a Gaussian whose every pair the depth limits removed is blended nowhere, so its 192 B of SH
      // coefficients are neither read nor evaluated (its gradient is zero as well, see preprocess_bwd)
      if (!(a.tile_depth_limit && tiles == 0)) {
        if (a.colors_precomp == nullptr) {
          V3 campos = {cam[0], cam[1], cam[2]};
          V3 rgb;
          uint32_t cl = 0;
          if (a.M == 16 && a.shs_rest) {
            // the model's split rows (GsGaussians.shs_rest): 12 B of _features_dc + the active part of the 180 B row of
            // _features_rest, dword loads (a 180 B row starts on a 16 B boundary for every fourth Gaussian only)
            ShRegs sh;
            const float* dc = a.shs + (size_t)idx * 3;
            const float* rest = a.shs_rest + (size_t)idx * 45;
            const int nfl = 3 * (a.D + 1) * (a.D + 1) - 3;
            sh.f[0] = dc[0]; sh.f[1] = dc[1]; sh.f[2] = dc[2];
  #pragma unroll
            for (int k = 0; k < 45; k++) sh.f[3 + k] = k < nfl ? rest[k] : 0.f;
            bump();
            rgb = color_from_sh(a.D, p_orig, campos, sh, cl);
          } else if (a.M == 16) {
            ShRegs sh;
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
            bump();
            rgb = color_from_sh(a.D, p_orig, campos, sh, cl);
          } else if (a.shs_rest) {
            bump();
            ShMemSplit sh{a.shs + (size_t)idx * 3, a.shs_rest + (size_t)idx * (a.M - 1) * 3};
            rgb = color_from_sh(a.D, p_orig, campos, sh, cl);
          } else {
            bump();
            ShMem sh{a.shs + (size_t)idx * a.M * 3};
            rgb = color_from_sh(a.D, p_orig, campos, sh, cl);
          }
          sp.r = rgb.x; sp.g = rgb.y; sp.b = rgb.z;
          sp.clamped |= cl;  // (bits 8-9 already hold the depth-limit verdict)
        } else {
          bump();
          sp.r = a.colors_precomp[3 * idx];
          sp.g = a.colors_precomp[3 * idx + 1];
          sp.b = a.colors_precomp[3 * idx + 2];
        }
      }
      sp.tiles = tiles;
    } while (false);

    // a Gaussian the depth limits removed entirely leaves no record and no covariance behind: nothing downstream reads
    // them (it is not in the depth order, no list names it, and the backward skips it on tiles_touched == 0)
    const bool write_record = !(a.tile_depth_limit && tiles == 0 && radius_out > 0);
    if (write_record) {
      float4* dst = reinterpret_cast<float4*>(&g.splat[idx]);
      const float4* srcv = reinterpret_cast<const float4*>(&sp);
      dst[0] = srcv[0]; dst[1] = srcv[1]; dst[2] = srcv[2]; dst[3] = srcv[3];
    }
    if (a.cov3D_precomp == nullptr && write_record) {
      float2* cd = reinterpret_cast<float2*>(g.cov3D + (size_t)idx * 6);
      if (!write_cov) { cov3D[0] = cov3D[1] = cov3D[2] = cov3D[3] = cov3D[4] = cov3D[5] = 0.f; }
      cd[0] = make_float2(cov3D[0], cov3D[1]);
      cd[1] = make_float2(cov3D[2], cov3D[3]);
      cd[2] = make_float2(cov3D[4], cov3D[5]);
    }
    g.tiles_touched[idx] = tiles;
    collect();  // (the last batch of bucket slots: nothing pending outside region mode)
    // key of the per-Gaussian depth sort (gs_binning.hip): culled Gaussians sort behind everything
    if (a.tile_cull != 2) g.depth_keys[idx] = tiles ? __float_as_uint(sp.depth) : 0xFFFFFFFFu;
    a.radii[idx] = radius_out;
  }
  // per-workgroup partial sum of tiles_touched for the prefix sum
  __shared__ uint32_t red[GS_BLOCK / 64];
  uint32_t v = tiles;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) g.block_sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Exclusive scan of the per-workgroup sums, in place, one workgroup (nb <= 2^22); the grand total
// (num_rendered) goes to the header and to block_sums[nb].
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(GeomView g, int nb, uint32_t P) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    int i = base + tid;
    uint32_t v = (i < nb) ? g.block_sums[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    uint32_t carry = carry_s;
    if (i < nb) g.block_sums[i] = carry + woff + inc - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) {
    g.block_sums[nb] = carry_s;
    g.hdr->num_rendered = carry_s;
    g.hdr->overflow = 0;
    g.hdr->P = P;
    g.hdr->region_mode = 0;
    g.hdr->pad[HDR_SIDE_CURSOR] = 0;
  }
}

// rasterizer_impl.cu:54-66
__global__ void __launch_bounds__(GS_BLOCK) mark_visible_kernel(int P, const float* means3D, const float* vm,
                                                                uint8_t* present) {
  const int idx = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (idx >= P) return;
  V3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
  V3 pv = xform4x3(p, vm);
  present[idx] = !(pv.z <= 0.2f);
}

int launch_preprocess_fwd(const PreprocessArgs& a, const GeomView& g, hipStream_t s) {
  const int nb = (a.P + GS_BLOCK - 1) / GS_BLOCK;
  const size_t lds = preprocess_limit_lds_floats(a);
  hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(nb), dim3(GS_BLOCK), 4 * lds, s, a, g, (int)lds);
  return 0;
}
int launch_scan_block_sums(const GeomView& g, int P, hipStream_t s) {
  const int nb = (P + GS_BLOCK - 1) / GS_BLOCK;
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, g, nb, (uint32_t)P);
  return 0;
}
int launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s) {
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, P, means3D,
                     viewmatrix, present);
  return 0;
}
