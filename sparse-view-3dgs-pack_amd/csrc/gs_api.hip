// gs_api.hip - extern "C" entry points of libgsplat_hip.so for the rasterizer (include/gsplat.h).
// Argument checking, scratch carving and launch orchestration only; kernels live in the other files.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "gs_common.h"
#include "gs_tilecull.h"
#include "gs_prof.h"

static int check_args(const GsView* v, const GsGaussians* g) {
  if (!v || !g) return GS_E_NULL;
  if (g->P < 0 || v->image_width <= 0 || v->image_height <= 0) return GS_E_SHAPE;
  if (g->P == 0) return GS_OK;
  if (!g->means3D || !g->opacities || !v->viewmatrix || !v->projmatrix || !v->bg) return GS_E_NULL;
  if ((g->shs == nullptr) == (g->colors_precomp == nullptr)) return GS_E_SHAPE;
  const bool has_sr = g->scales != nullptr && g->rotations != nullptr;
  const bool any_sr = g->scales != nullptr || g->rotations != nullptr;
  if ((!has_sr && g->cov3D_precomp == nullptr) || (any_sr && g->cov3D_precomp != nullptr)) return GS_E_SHAPE;
  if (g->shs && (g->M < (v->sh_degree + 1) * (v->sh_degree + 1) || !v->campos)) return GS_E_SHAPE;
  if (v->sh_degree < 0 || v->sh_degree > 3) return GS_E_SHAPE;
  if (g->shs_rest && (!g->shs || g->M < 2)) return GS_E_SHAPE;  // split rows: DC in shs, coefficients 1..M-1 in shs_rest
  if (v->image_width > 65535 * TILE_X || v->image_height > 65535 * TILE_Y) return GS_E_UNSUPPORTED;
  return GS_OK;
}

static inline void tile_grid(const GsView* v, int& gx, int& gy) {
  gx = (v->image_width + TILE_X - 1) / TILE_X;
  gy = (v->image_height + TILE_Y - 1) / TILE_Y;
}

// where region binning (GsView.tile_cull = 2, gs_regionbin.hip) keeps its state inside the caller's scratch buffers
struct RegionLayout {
  uint32_t* count;     // [rg_x * rg_y]  (image buffer)
  uint2* bucket;       // [rg_x * rg_y][cap]  (the two key halves of the binning buffer: 8 B x capacity)
  uint32_t* point_list;  // (Gaussian ids[0] of the binning buffer: where the backward looks for it)
  uint32_t cap;
  int rg_x, rg_y;
};
static int region_layout(const GsView* v, const GsScratch* sc, RegionLayout& rl) {
  int gx, gy;
  tile_grid(v, gx, gy);
  const size_t T = (size_t)gx * gy, N = (size_t)v->image_width * v->image_height;
  if (!sc->img || !sc->binning) return GS_E_NULL;
  if (sc->img_bytes < img_bytes(N, T)) return GS_E_SCRATCH;
  const int64_t cap = sc->binning_capacity;
  if (cap <= 0 || cap > 0xFFFFFFFFll) return GS_E_SHAPE;
  if (sc->binning_bytes < bin_bytes((size_t)cap, T)) return GS_E_SCRATCH;
  rl.rg_x = (gx + RG_TILES - 1) / RG_TILES;
  rl.rg_y = (gy + RG_TILES - 1) / RG_TILES;
  rl.cap = region_capacity(cap, rl.rg_x * rl.rg_y);
  if (rl.cap == 0) return GS_E_SHAPE;  // fewer instances of capacity than regions
  ImgView iv = img_view(sc->img, N, T);
  SortBufs bv = sort_view(sc->binning, (size_t)cap);
  rl.count = iv.region_count;
  rl.bucket = reinterpret_cast<uint2*>(bv.keys[0]);  // keys[0] and keys[1] are adjacent: 2 x align(4 cap) >= 8 regions cap_r
  rl.point_list = bv.vals[0];
  return GS_OK;
}

extern "C" {

int gs_abi_version(void) { return GS_ABI_VERSION; }

size_t gs_struct_bytes(int32_t which) {
  switch (which) {
    case 0: return sizeof(GsView);
    case 1: return sizeof(GsGaussians);
    case 2: return sizeof(GsScratch);
    case 3: return sizeof(GsGrads);
    case 4: return sizeof(GsStepState);
    case 5: return sizeof(GsLgdwtParams);
    case 6: return sizeof(GsAdamSeg);
    default: return 0;
  }
}

const char* gs_build_info(void) {
  return "libgsplat_hip gfx950 | hipcc " __VERSION__
         " | preprocess/knn: -ffp-contract=off, blend: fast contract | radix 8-bit LSD | tile 16x16, wave64 8x8 quadrants";
}

int gs_scratch_bytes(int32_t P, int32_t W, int32_t H, int64_t R_capacity, size_t out[3], size_t* bwd_ws) {
  if (!out) return GS_E_NULL;
  if (P < 0 || W <= 0 || H <= 0 || R_capacity < 0) return GS_E_SHAPE;
  const size_t T = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
  out[0] = geom_bytes((size_t)P);
  out[1] = img_bytes((size_t)W * H, T);
  out[2] = bin_bytes((size_t)R_capacity, T);
  if (bwd_ws) *bwd_ws = bwd_workspace_bytes((size_t)P);
  return GS_OK;
}

int gs_forward_geometry(const GsView* v, const GsGaussians* g, GsScratch* sc, int32_t* radii,
                        int32_t* num_rendered_host, void* stream) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!sc || !sc->geom) return GS_E_NULL;
  sc->binned = 0;  // a new geometry state invalidates whatever lists these buffers held (gs_forward_bin sets it again)
  hipStream_t s = (hipStream_t)stream;
  const int P = g->P;
  if (sc->geom_bytes < geom_bytes((size_t)P)) return GS_E_SCRATCH;
  GeomView gv = geom_view(sc->geom, (size_t)P);
  if (P == 0) {
    GS_HIP_CHECK(hipMemsetAsync(gv.hdr, 0, sizeof(GeomHeader), s));
    if (num_rendered_host) GS_HIP_CHECK(hipMemcpyAsync(num_rendered_host, &gv.hdr->num_rendered, 4, hipMemcpyDeviceToHost, s));
    return GS_OK;
  }
  if (!radii) return GS_E_NULL;
  PreprocessArgs a;
  a.P = P;
  a.D = v->sh_degree;
  a.M = g->M;
  a.means3D = g->means3D;
  a.scales = g->scales;
  a.scale_modifier = v->scale_modifier;
  a.rotations = g->rotations;
  a.opacities = g->opacities;
  a.raw_activations = g->raw_activations;
  a.shs = g->shs;
  a.shs_rest = g->shs_rest;
  a.cov3D_precomp = g->cov3D_precomp;
  a.colors_precomp = g->colors_precomp;
  a.viewmatrix = v->viewmatrix;
  a.projmatrix = v->projmatrix;
  a.campos = v->campos;
  a.W = v->image_width;
  a.H = v->image_height;
  // rasterizer_impl.cu:224-225
  a.focal_y = v->image_height / (2.0f * v->tanfovy);
  a.focal_x = v->image_width / (2.0f * v->tanfovx);
  a.tan_fovx = v->tanfovx;
  a.tan_fovy = v->tanfovy;
  a.radii = radii;
  tile_grid(v, a.grid_x, a.grid_y);
  a.antialiasing = v->antialiasing;
  a.extra_channel = g->extra_channel;
  a.extra_gain = g->extra_gain;
  a.tile_cull = v->tile_cull;
  a.tile_depth_limit = v->tile_cull ? sc->tile_depth_limit : nullptr;
  a.region_count = nullptr;
  a.region_bucket = nullptr;
  a.region_cap = 0;
  a.rg_x = a.rg_y = 0;
  if (v->tile_cull == 2) {
    // region binning: the buckets live in the binning buffer, the counters in the image buffer - both must be here already
    RegionLayout rl;
    rc = region_layout(v, sc, rl);
    if (rc) return rc;
    a.region_count = rl.count;
    a.region_bucket = rl.bucket;
    a.region_cap = rl.cap;
    a.rg_x = rl.rg_x;
    a.rg_y = rl.rg_y;
    launch_region_prepare(gv, rl.count, rl.rg_x * rl.rg_y, (uint32_t)P, s);
    GS_LAUNCH_CHECK(s, v->debug);
  }
  {
    GS_PROF(ST_PREPROCESS_FWD, s);
    launch_preprocess_fwd(a, gv, s);
  }
  GS_LAUNCH_CHECK(s, v->debug);
  if (v->tile_cull != 2) {  // (region binning needs no prefix sum: num_rendered is known when the lists are)
    GS_PROF(ST_SCAN, s);
    launch_scan_block_sums(gv, P, s);
  }
  GS_LAUNCH_CHECK(s, v->debug);
  if (num_rendered_host)
    GS_HIP_CHECK(hipMemcpyAsync(num_rendered_host, &gv.hdr->num_rendered, 4, hipMemcpyDeviceToHost, s));
  return GS_OK;
}

int gs_forward_render(const GsView* v, const GsGaussians* g, GsScratch* sc, float* out_color, float* out_invdepth,
                      void* stream) {
  return gs_forward_render_x(v, g, sc, out_color, out_invdepth, nullptr, stream);
}

// The binning stage: instance lists (point_list = Gaussian ids[0] of the binning buffer) and ranges[] of the view whose
// geometry phase has run on these buffers.
static int forward_bin_stage(const GsView* v, const GsScratch* sc, const GeomView& gv, const ImgView& iv, const SortBufs& bv, int P,
                             int gx, int gy, hipStream_t s) {
  int rc;
  const int64_t cap = sc->binning_capacity;
  const size_t T = (size_t)gx * gy;
  if (v->tile_cull == 2) {
    RegionLayout rl;
    rc = region_layout(v, sc, rl);
    if (rc) return rc;
    GS_PROF(ST_SORT, s);
    rc = launch_region_bin(gv, rl.count, rl.bucket, rl.cap, rl.rg_x, rl.rg_y, gx, gy, sc->tile_depth_limit, iv.ranges, rl.point_list,
                           cap, s);
    if (rc) return rc;
    GS_LAUNCH_CHECK(s, v->debug);
    return GS_OK;
  }
  launch_bin_prepare(gv, cap, iv.ranges, (int)T, s);
  GS_LAUNCH_CHECK(s, v->debug);
  if (cap > 0) {
    {  // 1. depth order of the P Gaussians: 4 passes; pass 0 reads the keys preprocess wrote (kept intact, so the
       //    phase can be re-run), then half 1 -> 0 -> 1 -> 0: the order ends in gsort.vals[0]
      GS_PROF(ST_SORT_DEPTH, s);
      rc = launch_radix_sort(gv.gsort, &gv.hdr->P, P, 32, 0, gv.depth_keys, s, v->debug, &gv.hdr->n_ordered);
      if (rc) return rc;
    }
    // 2. region entries in depth order, stable partition by region id, expansion into the tile lists (gs_tilebin.hip):
    //    point_list ends in ids[0] of the binning buffer, where the blend kernels and the exports look for it
    const TileBinView tb = tilebin_view((char*)sc->binning + sort_bytes((size_t)cap), (size_t)cap, T);
    rc = launch_tile_binning(gv, bv, tb, P, cap, gx, gy, v->tile_cull, sc->tile_depth_limit, iv.ranges, (v->debug & 2) ? 1 : 0, s,
                             v->debug & 1);
    if (rc) return rc;
  }
  GS_LAUNCH_CHECK(s, v->debug);
  return GS_OK;
}

static int forward_render_impl(const GsView* v, const GsGaussians* g, GsScratch* sc, float* out_color, float* out_invdepth,
                               float* out_extra, int fsgs, void* stream);

int gs_forward_bin(const GsView* v, const GsGaussians* g, GsScratch* sc, uint32_t* status_host, void* stream) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!sc || !sc->geom || !sc->img) return GS_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  const int P = g->P, W = v->image_width, H = v->image_height;
  int gx, gy;
  tile_grid(v, gx, gy);
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  if (sc->img_bytes < img_bytes(N, T)) return GS_E_SCRATCH;
  if (P > 0) {
    if (sc->geom_bytes < geom_bytes((size_t)P)) return GS_E_SCRATCH;
    const int64_t cap = sc->binning_capacity;
    if (cap < 0) return GS_E_SHAPE;
    if (cap > 0 && (!sc->binning || sc->binning_bytes < bin_bytes((size_t)cap, T))) return GS_E_SCRATCH;
    if (cap > 0xFFFFFFFFll) return GS_E_UNSUPPORTED;
    GeomView gv = geom_view(sc->geom, (size_t)P);
    ImgView iv = img_view(sc->img, N, T);
    SortBufs bv = sort_view(sc->binning, (size_t)cap);
    rc = forward_bin_stage(v, sc, gv, iv, bv, P, gx, gy, s);
    if (rc) return rc;
  }
  sc->binned = 1;
  if (status_host) GS_HIP_CHECK(hipMemcpyAsync(status_host, sc->geom, 16, hipMemcpyDeviceToHost, s));
  return GS_OK;
}

int gs_forward_render_x(const GsView* v, const GsGaussians* g, GsScratch* sc, float* out_color, float* out_invdepth,
                        float* out_extra, void* stream) {
  if (out_extra && g && !g->extra_channel) return GS_E_NULL;
  return forward_render_impl(v, g, sc, out_color, out_invdepth, out_extra, 0, stream);
}

int gs_forward_render_fsgs(const GsView* v, const GsGaussians* g, GsScratch* sc, float* out_color, float* out_depth,
                           float* out_alpha, void* stream) {
  if (!out_depth || !out_alpha) return GS_E_NULL;
  if (v && v->antialiasing) return GS_E_UNSUPPORTED;  // that rasterizer generation has no anti-aliasing
  return forward_render_impl(v, g, sc, out_color, out_depth, out_alpha, 1, stream);
}

static int forward_render_impl(const GsView* v, const GsGaussians* g, GsScratch* sc, float* out_color, float* out_invdepth,
                               float* out_extra, int fsgs, void* stream) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!sc || !sc->geom || !sc->img || !out_color) return GS_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  const int P = g->P, W = v->image_width, H = v->image_height;
  int gx, gy;
  tile_grid(v, gx, gy);
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  if (sc->img_bytes < img_bytes(N, T)) return GS_E_SCRATCH;
  if (P == 0) {  // rasterize_points.cu:88 - outputs stay zero
    GS_HIP_CHECK(hipMemsetAsync(out_color, 0, sizeof(float) * GS_NUM_CHANNELS * N, s));
    if (out_invdepth) GS_HIP_CHECK(hipMemsetAsync(out_invdepth, 0, sizeof(float) * N, s));
    if (out_extra) GS_HIP_CHECK(hipMemsetAsync(out_extra, 0, sizeof(float) * N, s));
    return GS_OK;
  }
  if (sc->geom_bytes < geom_bytes((size_t)P)) return GS_E_SCRATCH;
  const int64_t cap = sc->binning_capacity;
  if (cap < 0) return GS_E_SHAPE;
  if (cap > 0 && (!sc->binning || sc->binning_bytes < bin_bytes((size_t)cap, T))) return GS_E_SCRATCH;
  if (cap > 0xFFFFFFFFll) return GS_E_UNSUPPORTED;
  GeomView gv = geom_view(sc->geom, (size_t)P);
  ImgView iv = img_view(sc->img, N, T);
  SortBufs bv = sort_view(sc->binning, (size_t)cap);

  if (!sc->binned) {
    rc = forward_bin_stage(v, sc, gv, iv, bv, P, gx, gy, s);
    if (rc) return rc;
  }
  {
    GS_PROF(ST_RENDER_FWD, s);
    launch_render_fwd_wave(iv.ranges, bv.vals[0], W, H, gx, gy, gv.splat, v->bg, iv.final_T, iv.n_contrib, iv.tile_work,
                           sc->tile_order_hint, sc->tile_depth_limit, iv.tile_stop_depth, &gv.hdr->trunc_failed, out_color,
                           out_invdepth, out_extra, fsgs, v->tile_cull ? 0 : 1, s);
  }
  if (!sc->defer_tile_order) {
    GS_PROF(ST_TILE_ORDER, s);
    launch_tile_order(iv.tile_work, iv.tile_order, (int)T, sc->tile_order_out, iv.tile_stop_depth, sc->tile_depth_limit_out, gx, gy,
                      gv.hdr, s, sc->tile_depth_limit_slack, sc->status_host, sc->step_tag);
  }
  GS_LAUNCH_CHECK(s, v->debug);
  return GS_OK;
}

int gs_backward(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* sc,
                int64_t num_rendered, const float* dL_dcolor, const float* dL_dinvdepth, const GsGrads* grads,
                void* workspace, size_t workspace_bytes, void* stream) {
  return gs_backward_x(v, g, radii, sc, num_rendered, dL_dcolor, dL_dinvdepth, nullptr, grads, workspace, workspace_bytes,
                       stream);
}

static int backward_impl(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* sc,
                         int64_t num_rendered, const float* dL_dcolor, const float* dL_dinvdepth, const float* dL_dextra,
                         int fsgs, const GsGrads* grads, void* workspace, size_t workspace_bytes, void* stream,
                         const GsStepState* step = nullptr);

int gs_backward_x(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* sc,
                  int64_t num_rendered, const float* dL_dcolor, const float* dL_dinvdepth, const float* dL_dextra,
                  const GsGrads* grads, void* workspace, size_t workspace_bytes, void* stream) {
  if (dL_dextra && g && grads && (!g->extra_channel || !grads->dL_dextra)) return GS_E_NULL;
  return backward_impl(v, g, radii, sc, num_rendered, dL_dcolor, dL_dinvdepth, dL_dextra, 0, grads, workspace,
                       workspace_bytes, stream);
}

int gs_backward_fsgs(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* sc,
                     int64_t num_rendered, const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                     const GsGrads* grads, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dL_ddepth || !dL_dalpha) return GS_E_NULL;
  if (v && v->antialiasing) return GS_E_UNSUPPORTED;
  return backward_impl(v, g, radii, sc, num_rendered, dL_dcolor, dL_ddepth, dL_dalpha, 1, grads, workspace,
                       workspace_bytes, stream);
}

static PreprocessBwdArgs preprocess_bwd_args(const GsView* v, const GsGaussians* g, const int32_t* radii, const GeomView& gv,
                                             int depth_mode, const gs_row_t* rows, float* recs, const GsGrads* grads) {
  PreprocessBwdArgs a;
  a.P = g->P;
  a.D = v->sh_degree;
  a.M = g->M;
  a.means3D = g->means3D;
  a.radii = radii;
  a.shs = g->shs;
  a.shs_rest = g->shs_rest;
  a.scales = g->scales;
  a.rotations = g->rotations;
  a.opacities = g->opacities;
  a.raw_activations = g->raw_activations;
  a.skip_uninstanced = 0;
  a.tiles_touched = gv.tiles_touched;
  a.colors_precomp = g->colors_precomp;
  a.scale_modifier = v->scale_modifier;
  a.cov3D = g->cov3D_precomp ? g->cov3D_precomp : gv.cov3D;
  a.viewmatrix = v->viewmatrix;
  a.projmatrix = v->projmatrix;
  a.campos = v->campos;
  a.focal_y = v->image_height / (2.0f * v->tanfovy);
  a.focal_x = v->image_width / (2.0f * v->tanfovx);
  a.tan_fovx = v->tanfovx;
  a.tan_fovy = v->tanfovy;
  a.antialiasing = v->antialiasing;
  a.has_invdepth = depth_mode;
  a.has_extra = g->extra_channel != nullptr;
  a.extra_raw = a.extra_gain = nullptr;
  a.gain_partials = nullptr;
  a.grad_rows = rows;
  a.grad_recs = recs;
  a.clean_rows = 0;
  a.splat = gv.splat;
  a.out = *grads;
  return a;
}

// host half of gs_backward_step: argument rules and the bias corrections (as gs_adam_step computes them)
static int step_args(const GsGaussians* g, const GsStepState* st, StepArgs& sa) {
  if (!st->xyz || !st->features || !st->opacity || !st->scaling || !st->rotation) return GS_E_NULL;
  const bool grads_out = st->grad_out[0] != nullptr;
  for (int k = 0; k < 5; k++) {
    if (grads_out ? !st->grad_out[k] : (!st->m[k] || !st->v[k])) return GS_E_NULL;
  }
  const bool any_stat = st->max_radii2D || st->xyz_gradient_accum || st->denom;
  if (any_stat && !(st->max_radii2D && st->xyz_gradient_accum && st->denom)) return GS_E_NULL;
  if (g->means3D != st->xyz || g->shs != st->features) return GS_E_SHAPE;  // no activation between them
  // split SH rows: only the gradients-out form serves them (one gradient tensor per model tensor)
  if ((g->shs_rest != nullptr) != (st->grad_out_rest != nullptr)) return GS_E_SHAPE;
  if (g->shs_rest && !grads_out) return GS_E_UNSUPPORTED;
  if (g->M != 16 || g->colors_precomp || g->cov3D_precomp || !g->scales || !g->rotations) return GS_E_UNSUPPORTED;
  // the 4th channel: its raw row and the gain are this step's parameters, or there is no 4th channel at all
  if ((g->extra_channel != nullptr) != (st->extra != nullptr)) return GS_E_UNSUPPORTED;
  if (st->extra) {
    if (g->extra_channel != st->extra || g->extra_gain != st->gain || !st->gain || !g->raw_activations) return GS_E_SHAPE;
    if (grads_out ? (!st->grad_out_extra || !st->grad_out_gain) : (!st->extra_m || !st->extra_v || !st->gain_m || !st->gain_v))
      return GS_E_NULL;
    if (st->step_extra < 0 || st->step_gain < 0) return GS_E_SHAPE;
  }
  sa.st = *st;
  sa.phase = 0;
  sa.phase1_workgroups = 0;
  sa.gain_partials = nullptr;
  {
    const int tx = st->step_extra > 0 ? st->step_extra : 1, tg = st->step_gain > 0 ? st->step_gain : 1;
    sa.x_inv_sqrt_bc2[0] = (float)(1.0 / sqrt(1.0 - pow((double)st->beta2, (double)tx)));
    sa.x_inv_sqrt_bc2[1] = (float)(1.0 / sqrt(1.0 - pow((double)st->beta2, (double)tg)));
    sa.x_lr_bc1[0] = st->lr_extra * (float)(1.0 / (1.0 - pow((double)st->beta1, (double)tx)));
    sa.x_lr_bc1[1] = st->lr_gain * (float)(1.0 / (1.0 - pow((double)st->beta1, (double)tg)));
  }
  static const int row_of_lr[6] = {0, 1, 1, 2, 3, 4};
  for (int k = 0; k < 5; k++) {
    if (st->step[k] < 0) return GS_E_SHAPE;
    const int t = st->step[k] > 0 ? st->step[k] : 1;
    sa.inv_sqrt_bc2[k] = (float)(1.0 / sqrt(1.0 - pow((double)st->beta2, (double)t)));
  }
  for (int c = 0; c < 6; c++) {
    const int t = st->step[row_of_lr[c]] > 0 ? st->step[row_of_lr[c]] : 1;
    sa.lr_bc1[c] = st->lr[c] * (float)(1.0 / (1.0 - pow((double)st->beta1, (double)t)));
  }
  return GS_OK;
}

static int backward_impl(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* sc,
                         int64_t num_rendered, const float* dL_dcolor, const float* dL_dinvdepth, const float* dL_dextra,
                         int fsgs, const GsGrads* grads, void* workspace, size_t workspace_bytes, void* stream,
                         const GsStepState* step) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!sc || !grads || !dL_dcolor) return GS_E_NULL;
  if (g->raw_activations && !step) return GS_E_UNSUPPORTED;  // gradients w.r.t. activated values that were never formed
  if (g->shs_rest && !step) return GS_E_UNSUPPORTED;         // GsGrads.dL_dsh is one [P,M,3] array
  StepArgs sa;
  if (step && g->P > 0) {
    rc = step_args(g, step, sa);
    if (rc) return rc;
  }
  const int P = g->P, W = v->image_width, H = v->image_height;
  if (P == 0) return GS_OK;
  if (!radii || !sc->geom || !sc->img || !workspace) return GS_E_NULL;
  if (workspace_bytes < bwd_workspace_bytes((size_t)P)) return GS_E_SCRATCH;
  if (num_rendered < 0 || num_rendered > sc->binning_capacity) return GS_E_SHAPE;
  if (num_rendered > 0 && !sc->binning) return GS_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  int gx, gy;
  tile_grid(v, gx, gy);
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  GeomView gv = geom_view(sc->geom, (size_t)P);
  ImgView iv = img_view(sc->img, N, T);
  SortBufs bv = sort_view(sc->binning, (size_t)sc->binning_capacity);
  gs_row_t* rows = (gs_row_t*)workspace;
  float* recs = reinterpret_cast<float*>((char*)workspace + gs_align((size_t)P * GR_ROW_BYTES));
  const bool given_rows = step && step->rows_override;
  if (step && (step->rows_clean < 0 || step->rows_clean > 2)) return GS_E_SHAPE;
  const int rows_clean = (step && !given_rows) ? step->rows_clean : 0;
  if (given_rows) {
    rows = const_cast<gs_row_t*>(step->rows_override);
  } else if (rows_clean != 2) {  // (2: the previous step's per-Gaussian kernel left every row zero)
    GS_PROF(ST_BWD_MEMSET, s);
    launch_zero_rows(rows, (size_t)P, (v->tile_cull || (step && step->phase == 2)) ? gv.tiles_touched : nullptr, s);
  }
  if (num_rendered > 0 && !given_rows) {
    {
      GS_PROF(ST_RENDER_BWD, s);
      launch_render_bwd_wave(iv.ranges, bv.vals[0], W, H, gx, gy, gv.splat, v->bg, iv.final_T, iv.n_contrib, iv.tile_work,
                             iv.tile_order, dL_dcolor, dL_dinvdepth, dL_dextra, rows, fsgs, s);
    }
    GS_LAUNCH_CHECK(s, v->debug);
  }
  PreprocessBwdArgs a = preprocess_bwd_args(v, g, radii, gv, fsgs ? 2 : (dL_dinvdepth != nullptr ? 1 : 0), rows, recs, grads);
  // (with the reference's lists every visible Gaussian has instances and tiles_touched == 0 means "not visible": skipping on it
  //  is the same thing there; the two-phase step splits the Gaussians on it in either list mode)
  a.skip_uninstanced = (v->tile_cull || (step && step->phase == 2)) ? 1 : 0;
  a.clean_rows = rows_clean != 0;
  if (step && step->extra) {  // per-wave partial sums of dL/dgain: behind the records
    float* gp = reinterpret_cast<float*>((char*)workspace + gs_align((size_t)P * GR_ROW_BYTES) + gs_align((size_t)P * GC_REC_BYTES));
    a.extra_raw = step->extra;
    a.extra_gain = step->gain;
    a.gain_partials = gp;
    sa.gain_partials = gp;
  }
  {  // float64 sums -> fp32 records (the covariance chain in double); cleans the rows it read when asked to
    GS_PROF(ST_CHAIN, s);
    launch_chain(a, step ? gv.hdr : nullptr, s);
  }
  if (step) {
    GS_PROF(ST_BWD_STEP, s);
    sa.hdr = gv.hdr;

    if (step->phase == 2) {  // the Gaussians without instances are stepped by gs_step_uninstanced (maybe still running)
      if (step->grad_out[0]) return GS_E_UNSUPPORTED;
      sa.phase = 2;
      if (step->phase1_done) {
        const hipError_t e = hipStreamWaitEvent(s, (hipEvent_t)step->phase1_done, 0);
        if (e != hipSuccess) return (int)e;  // (positive: a HIP error code, as after a failed launch)
      }
    } else if (step->phase != 0) {
      return GS_E_SHAPE;
    }
    launch_preprocess_bwd_step(a, sa, s);
  } else {
    GS_PROF(ST_PREPROCESS_BWD, s);
    launch_preprocess_bwd(a, s);
  }
  GS_LAUNCH_CHECK(s, v->debug);
  return GS_OK;
}

int gs_backward_step(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* sc, int64_t num_rendered,
                     const float* dL_dcolor, const float* dL_dinvdepth, const GsStepState* st, void* workspace,
                     size_t workspace_bytes, void* stream) {
  if (!st) return GS_E_NULL;
  const GsGrads none = {};
  return backward_impl(v, g, radii, sc, num_rendered, dL_dcolor, dL_dinvdepth, nullptr, 0, &none, workspace, workspace_bytes,
                       stream, st);
}

int gs_backward_step_x(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* sc, int64_t num_rendered,
                       const float* dL_dcolor, const float* dL_dinvdepth, const float* dL_dextra_img, const GsStepState* st,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!st) return GS_E_NULL;
  if (g && ((dL_dextra_img != nullptr) != (g->extra_channel != nullptr))) return GS_E_NULL;
  const GsGrads none = {};
  return backward_impl(v, g, radii, sc, num_rendered, dL_dcolor, dL_dinvdepth, dL_dextra_img, 0, &none, workspace, workspace_bytes,
                       stream, st);
}

int gs_step_uninstanced(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* sc, const GsStepState* st,
                        void* stream) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!sc || !st) return GS_E_NULL;
  if (st->grad_out[0]) return GS_E_UNSUPPORTED;
  const int P = g->P;
  if (P == 0) return GS_OK;
  StepArgs sa;
  rc = step_args(g, st, sa);
  if (rc) return rc;
  if (!radii || !sc->geom) return GS_E_NULL;
  if (sc->geom_bytes < geom_bytes((size_t)P)) return GS_E_SCRATCH;
  hipStream_t s = (hipStream_t)stream;
  GeomView gv = geom_view(sc->geom, (size_t)P);
  const GsGrads none = {};
  PreprocessBwdArgs a = preprocess_bwd_args(v, g, radii, gv, 0, nullptr, nullptr, &none);
  a.skip_uninstanced = 1;
  sa.hdr = gv.hdr;
  sa.phase = 1;
  {
    static const int wgs_env = [] { const char* e = getenv("GS_PHASE1_WORKGROUPS"); return e ? atoi(e) : 0; }();
    sa.phase1_workgroups = wgs_env;
  }
  {
    GS_PROF(ST_STEP_UNINST, s);
    launch_preprocess_bwd_step(a, sa, s);
  }
  GS_LAUNCH_CHECK(s, v->debug);
  return GS_OK;
}

int gs_backward_from_rows(const GsView* v, const GsGaussians* g, const int32_t* radii, const GsScratch* sc, const double* rows,
                          int32_t depth_mode, const GsGrads* grads, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_args(v, g);
  if (rc) return rc;
  if (!sc || !grads) return GS_E_NULL;
  if (g->raw_activations || g->shs_rest) return GS_E_UNSUPPORTED;
  if (depth_mode < 0 || depth_mode > 2) return GS_E_SHAPE;
  const int P = g->P;
  if (P == 0) return GS_OK;
  if (!radii || !sc->geom || !rows || !workspace) return GS_E_NULL;
  if (sc->geom_bytes < geom_bytes((size_t)P)) return GS_E_SCRATCH;
  if (workspace_bytes < bwd_workspace_bytes((size_t)P)) return GS_E_SCRATCH;
  hipStream_t s = (hipStream_t)stream;
  GeomView gv = geom_view(sc->geom, (size_t)P);
  float* recs = reinterpret_cast<float*>((char*)workspace + gs_align((size_t)P * GR_ROW_BYTES));  // (the rows part stays unused)
  PreprocessBwdArgs a = preprocess_bwd_args(v, g, radii, gv, depth_mode, rows, recs, grads);
  (void)hipGetLastError();
  launch_chain(a, nullptr, s);
  launch_preprocess_bwd(a, s);
  GS_LAUNCH_CHECK(s, v->debug);
  return GS_OK;
}

int gs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* /*projmatrix*/,
                    uint8_t* present, void* stream) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!means3D || !viewmatrix || !present) return GS_E_NULL;
  (void)hipGetLastError();
  launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
  GS_LAUNCH_CHECK((hipStream_t)stream, 0);
  return GS_OK;
}

// ------------------------------------------------------------------------------------------------
// parity exports
// ------------------------------------------------------------------------------------------------
__global__ void export_geom_kernel(GeomView g, int P, float* depths, float* means2D, float* cov3D, float* conic_opacity,
                                   float* rgb, uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_offsets) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const Splat sp = g.splat[i];
  const bool vis = sp.rect_max != 0;  // a visible Gaussian has a non-empty tile rectangle (maxx >= 1)
  if (depths) depths[i] = vis ? sp.depth : 0.f;
  if (means2D) {
    means2D[2 * i] = vis ? sp.x : 0.f;
    means2D[2 * i + 1] = vis ? sp.y : 0.f;
  }
  if (conic_opacity) {
    conic_opacity[4 * i] = vis ? sp.cxx : 0.f;
    conic_opacity[4 * i + 1] = vis ? sp.cxy : 0.f;
    conic_opacity[4 * i + 2] = vis ? sp.cyy : 0.f;
    conic_opacity[4 * i + 3] = vis ? sp.opacity : 0.f;
  }
  if (rgb) {
    rgb[3 * i] = vis ? sp.r : 0.f;
    rgb[3 * i + 1] = vis ? sp.g : 0.f;
    rgb[3 * i + 2] = vis ? sp.b : 0.f;
  }
  if (clamped)
    for (int c = 0; c < 3; c++) clamped[3 * i + c] = vis ? ((sp.clamped >> c) & 1u) : 0;
  if (cov3D)
    for (int k = 0; k < 6; k++) cov3D[6 * i + k] = vis ? g.cov3D[6 * (size_t)i + k] : 0.f;
  if (tiles_touched) tiles_touched[i] = g.tiles_touched[i];
}
// inclusive scan of tiles_touched in index order (what the reference keeps as point_offsets): the product
// path never materialises it (instances are emitted in depth order), so the export recomputes it
__global__ void export_offsets_kernel(GeomView g, int P, uint32_t* point_offsets) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = (P + GS_BLOCK - 1) / GS_BLOCK;
  if (b >= nb) return;
  uint32_t run = g.block_sums[b];
  for (int i = b * GS_BLOCK; i < min(P, (b + 1) * GS_BLOCK); i++) {
    run += g.tiles_touched[i];
    point_offsets[i] = run;
  }
}
int gs_export_geom(const GsScratch* sc, int32_t P, float* depths, float* means2D, float* cov3D, float* conic_opacity,
                   float* rgb, uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_offsets, void* stream) {
  if (!sc || !sc->geom) return GS_E_NULL;
  if (P <= 0) return GS_OK;
  GeomView gv = geom_view(sc->geom, (size_t)P);
  (void)hipGetLastError();
  hipLaunchKernelGGL(export_geom_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, gv, P, depths,
                     means2D, cov3D, conic_opacity, rgb, clamped, tiles_touched, point_offsets);
  if (point_offsets) {
    const int nb = (P + GS_BLOCK - 1) / GS_BLOCK;
    hipLaunchKernelGGL(export_offsets_kernel, dim3((nb + 63) / 64), dim3(64), 0, (hipStream_t)stream, gv, P, point_offsets);
  }
  GS_LAUNCH_CHECK((hipStream_t)stream, 0);
  return GS_OK;
}

int gs_export_img(const GsScratch* sc, int32_t W, int32_t H, float* final_T, uint32_t* n_contrib, uint32_t* ranges,
                  void* stream) {
  if (!sc || !sc->img) return GS_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  ImgView iv = img_view(sc->img, N, T);
  if (final_T) GS_HIP_CHECK(hipMemcpyAsync(final_T, iv.final_T, 4 * N, hipMemcpyDeviceToDevice, s));
  if (n_contrib) GS_HIP_CHECK(hipMemcpyAsync(n_contrib, iv.n_contrib, 4 * N, hipMemcpyDeviceToDevice, s));
  if (ranges) GS_HIP_CHECK(hipMemcpyAsync(ranges, iv.ranges, 8 * T, hipMemcpyDeviceToDevice, s));
  return GS_OK;
}

int gs_export_tile_order(const GsScratch* sc, int32_t W, int32_t H, uint32_t* out, void* stream) {
  if (!sc || !sc->img || !out) return GS_E_NULL;
  if (W <= 0 || H <= 0) return GS_E_SHAPE;
  const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  if (sc->img_bytes < img_bytes(N, T)) return GS_E_SCRATCH;
  ImgView iv = img_view(sc->img, N, T);
  GS_HIP_CHECK(hipMemcpyAsync(out, iv.tile_order, 4 * (((T + 7) / 8) * 8), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return GS_OK;
}

int gs_export_tile_stop_depth(const GsScratch* sc, int32_t W, int32_t H, float* out, void* stream) {
  if (!sc || !sc->img || !out) return GS_E_NULL;
  if (W <= 0 || H <= 0) return GS_E_SHAPE;
  const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  if (sc->img_bytes < img_bytes(N, T)) return GS_E_SCRATCH;
  ImgView iv = img_view(sc->img, N, T);
  return launch_export_stop_depth(iv.tile_stop_depth, out, gx, gy, (hipStream_t)stream);
}

size_t gs_tile_depth_limit_floats(int32_t W, int32_t H) {
  if (W <= 0 || H <= 0) return 0;
  return depth_limit_floats((uint32_t)((W + TILE_X - 1) / TILE_X), (uint32_t)((H + TILE_Y - 1) / TILE_Y));
}

// the tag of the replay plus a check word over (tag, status words): a host that POLLS the pinned copy of this block for the
// tag cannot know in which order the bytes of the device-to-host copy land - it accepts the block only when the check word
// matches what it reads (gs_forward_status)
__global__ void status_tag_kernel(GeomHeader* hdr, const uint32_t* __restrict__ tag) {
  const uint32_t t = *tag;
  hdr->step_tag = t;
  hdr->pad[0] = gs_status_check(t, hdr->num_rendered, hdr->overflow, hdr->trunc_failed);
}

int gs_forward_tile_order(const GsView* v, const GsScratch* sc, void* stream) {
  if (!v || !sc || !sc->geom || !sc->img) return GS_E_NULL;
  const int W = v->image_width, H = v->image_height;
  if (W <= 0 || H <= 0) return GS_E_SHAPE;
  int gx, gy;
  tile_grid(v, gx, gy);
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  if (sc->img_bytes < img_bytes(N, T) || sc->geom_bytes < sizeof(GeomHeader)) return GS_E_SCRATCH;
  hipStream_t s = (hipStream_t)stream;
  ImgView iv = img_view(sc->img, N, T);
  GS_PROF(ST_TILE_ORDER, s);
  launch_tile_order(iv.tile_work, iv.tile_order, (int)T, sc->tile_order_out, iv.tile_stop_depth, sc->tile_depth_limit_out, gx, gy,
                    (const GeomHeader*)sc->geom, s, sc->tile_depth_limit_slack, sc->status_host, sc->step_tag);
  GS_LAUNCH_CHECK(s, v->debug);
  return GS_OK;
}

int gs_forward_status(const GsScratch* sc, uint32_t* out, void* stream) {
  if (!sc || !sc->geom || !out) return GS_E_NULL;
  if (sc->geom_bytes < sizeof(GeomHeader)) return GS_E_SCRATCH;
  hipStream_t s = (hipStream_t)stream;
  size_t bytes = 16;
  if (sc->step_tag) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(status_tag_kernel, dim3(1), dim3(1), 0, s, (GeomHeader*)sc->geom, sc->step_tag);
    GS_LAUNCH_CHECK(s, 0);
    bytes = 48;
  }
  GS_HIP_CHECK(hipMemcpyAsync(out, sc->geom, bytes, hipMemcpyDeviceToHost, s));
  return GS_OK;
}

int gs_export_binning(const GsScratch* sc, int64_t R, uint64_t* keys_sorted, uint32_t* point_list, void* stream) {
  if (!sc) return GS_E_NULL;
  if (R <= 0) return GS_OK;
  if (!sc->binning) return GS_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (!sc->geom) return GS_E_NULL;
  SortBufs bv = sort_view(sc->binning, (size_t)sc->binning_capacity);
  // no tile-key array exists in any list mode (region binning, two-level binning): the reference's 64-bit key of every list
  // entry is rebuilt from ranges[] (tile) and the Gaussian's depth by gs_export_binning_region, which knows the image size
  if (keys_sorted) return GS_E_UNSUPPORTED;
  if (point_list) GS_HIP_CHECK(hipMemcpyAsync(point_list, bv.vals[0], 4 * (size_t)R, hipMemcpyDeviceToDevice, s));
  return GS_OK;
}

int gs_debug_blend_stats(const GsScratch* sc, int32_t P, int32_t W, int32_t H, uint64_t* out /* device, 8 words, zeroed by the caller */,
                         void* stream) {
  if (!sc || !sc->geom || !sc->img || !sc->binning || !out) return GS_E_NULL;
  if (P <= 0 || W <= 0 || H <= 0) return GS_E_SHAPE;
  const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  GeomView gv = geom_view(sc->geom, (size_t)P);
  ImgView iv = img_view(sc->img, N, T);
  SortBufs bv = sort_view(sc->binning, (size_t)sc->binning_capacity);
  (void)hipGetLastError();
  launch_blend_stats(iv.ranges, bv.vals[0], W, H, gx, gy, gv.splat, iv.n_contrib, iv.tile_work, (unsigned long long*)out,
                     (hipStream_t)stream);
  GS_LAUNCH_CHECK((hipStream_t)stream, 0);
  return GS_OK;
}

/* gs_export_binning for lists built by region binning (GsView.tile_cull = 2): the key of every list entry is rebuilt from
 * ranges[] (tile) and the Gaussian's depth; point_list as it lies in the buffer (tile lists in no particular order). */
int gs_export_binning_region(const GsScratch* sc, int32_t W, int32_t H, int64_t R, uint64_t* keys_sorted, uint32_t* point_list,
                             void* stream) {
  if (!sc || !sc->geom || !sc->img) return GS_E_NULL;
  if (R <= 0) return GS_OK;
  if (!sc->binning) return GS_E_NULL;
  if (W <= 0 || H <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
  const size_t T = (size_t)gx * gy, N = (size_t)W * H;
  if (sc->img_bytes < img_bytes(N, T)) return GS_E_SCRATCH;
  ImgView iv = img_view(sc->img, N, T);
  SortBufs bv = sort_view(sc->binning, (size_t)sc->binning_capacity);
  const Splat* splat = (const Splat*)((const char*)sc->geom + sizeof(GeomHeader));
  (void)hipGetLastError();
  if (keys_sorted) {
    GS_HIP_CHECK(hipMemsetAsync(keys_sorted, 0xFF, 8 * (size_t)R, s));  // (entries no tile owns would show as all-ones)
    launch_export_keys_region(iv.ranges, bv.vals[0], splat, (int)T, keys_sorted, s);
    GS_LAUNCH_CHECK(s, 0);
  }
  if (point_list) GS_HIP_CHECK(hipMemcpyAsync(point_list, bv.vals[0], 4 * (size_t)R, hipMemcpyDeviceToDevice, s));
  return GS_OK;
}

}  // extern "C"
