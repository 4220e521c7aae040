/*
 * gsplat.h — C ABI of libgsplat_hip.so, the MI355X (gfx950) differentiable 3D-Gaussian
 * rasterizer + distCUDA2 kNN + Haar-DWT / patch-ELF / SSIM loss kernels.
 *
 * This is the drop-in boundary for the hot path of sparse-view-3dgs-pack (LGDWT-GS).
 * Every entry point takes plain pointers + sizes + a hipStream_t passed as void*; no torch
 * types cross it.  All pointers are DEVICE pointers unless a parameter says "host".
 * Every function returns 0 on success, <0 for an invalid argument (GS_E_*), >0 = hipError_t.
 * Nothing throws across this ABI and nothing here allocates, frees or synchronises the
 * device except where stated (gs_forward_geometry's optional host read-back).
 *
 * Reference interfaces replaced (paths under
 * /root/reference/fs3dgs_benchmark/gaussian-splatting/submodules/):
 *   gs_forward_geometry + gs_forward_render
 *        = CudaRasterizer::Rasterizer::forward   diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:198-341
 *          as bound by RasterizeGaussiansCUDA     diff-gaussian-rasterization/rasterize_points.cu:35-124
 *          and pybind `rasterize_gaussians`       diff-gaussian-rasterization/ext.cpp:15-19
 *   gs_backward
 *        = CudaRasterizer::Rasterizer::backward   cuda_rasterizer/rasterizer_impl.cu:345-450
 *          as bound by RasterizeGaussiansBackwardCUDA  rasterize_points.cu:126-223
 *   gs_mark_visible
 *        = Rasterizer::markVisible                rasterizer_impl.cu:141-153, rasterize_points.cu:225-244
 *   gs_knn_mean_dist2
 *        = SimpleKNN::knn / distCUDA2             simple-knn/simple_knn.cu:186-222, simple-knn/spatial.cu:15-26
 *   gs_ssim_fwd / gs_ssim_bwd
 *        = fusedssim / fusedssim_backward         fused-ssim/ssim.cu:187-366
 *   gs_dwt_* / gs_elf_* / gs_patch_*
 *        = get_dwt_subbands / compute_elf_map / compute_patch_dwt_loss
 *          /root/reference/fs3dgs_benchmark/LGDWT-GS/utils/loss_utils.py:106-153,336-442
 *          (arithmetic of the un-vendored pytorch_wavelets DWTForward(J=1,'db1','symmetric'))
 */
#ifndef GSPLAT_H_INCLUDED
#define GSPLAT_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 7  /* round 5: gs_depth_l1, GsView.debug bit 1, binning buffer = sort arrays + region / chunk / tile tables;
                             GsGaussians.shs_rest / GsStepState.grad_out_rest (the model's split SH rows) */

/* error codes (negative = caller error) */
#define GS_OK 0
#define GS_E_NULL (-1)      /* required pointer is NULL */
#define GS_E_SHAPE (-2)     /* inconsistent sizes / exclusive-argument rule violated */
#define GS_E_SCRATCH (-3)   /* scratch buffer too small */
#define GS_E_OVERFLOW (-4)  /* num_rendered exceeded binning capacity (re-run gs_forward_render) */
#define GS_E_UNSUPPORTED (-5)

#define GS_TILE_X 16 /* cuda_rasterizer/config.h:16 */
#define GS_TILE_Y 16 /* cuda_rasterizer/config.h:17 */
#define GS_NUM_CHANNELS 3 /* cuda_rasterizer/config.h:15 */

/* Per-call camera / raster configuration.
 * Mirrors GaussianRasterizationSettings (dgr_3dgs/__init__.py:143-156).  Matrices are the
 * transposed (row-vector) forms the reference's Python hands over: viewmatrix = W2C^T,
 * projmatrix = (P*W2C)^T, each 16 contiguous floats. */
typedef struct GsView {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  float scale_modifier;
  int32_t sh_degree;   /* active degree D, 0..3 */
  int32_t prefiltered; /* bool */
  int32_t antialiasing; /* bool */
  int32_t debug;       /* non-zero: synchronise + check after every launch (auxiliary.h:178-185).  Bit 1 (value 2 or 3) is a
                          self-test of the list construction of tile_cull = 0 / 1 (csrc/gs_tilebin.hip): every third Gaussian
                          takes the row-wise entry enumeration that is normally reserved for Gaussians whose span hull holds
                          more regions than tiles; the lists must not change */
  int32_t tile_cull;   /* 0: instance lists = the reference's bounding-square rule (rasterizer_impl.cu:70-111), bit-identical
                          point_list / ranges / num_rendered; 1: additionally drop (tile, Gaussian) pairs on which alpha <
                          1/255 for every pixel (csrc/gs_tilecull.h) - same images and gradients, ~2.6x fewer instances;
                          2: the lists of 1 - every tile's list holds the same Gaussians in the same (depth, index) order -
                          built by REGION BINNING (csrc/gs_regionbin.hip: two launches instead of twenty-three; the lists of
                          different tiles lie in point_list in no particular order, ranges[] says where).  With 2 the
                          binning buffer must be given to gs_forward_geometry already (scratch->binning / binning_capacity) */
  const float* bg;         /* [3] */
  const float* viewmatrix; /* [16] */
  const float* projmatrix; /* [16] */
  const float* campos;     /* [3] */
} GsView;

/* Per-Gaussian inputs.  Exactly one of {shs, colors_precomp} and exactly one of
 * {scales+rotations, cov3D_precomp} is non-NULL (dgr_3dgs/__init__.py:178-182).
 * "Absent" is an explicit NULL (the reference passed data_ptr() of an empty tensor). */
typedef struct GsGaussians {
  int32_t P; /* number of Gaussians */
  int32_t M; /* SH coefficients stored per Gaussian (16 for degree 3); 0 when shs == NULL */
  const float* means3D;        /* [P,3] */
  const float* shs;            /* [P,M,3] or NULL */
  const float* colors_precomp; /* [P,3] or NULL */
  const float* opacities;      /* [P]   (the [P,1] tensor) */
  const float* scales;         /* [P,3] or NULL */
  const float* rotations;      /* [P,4] or NULL (r,x,y,z; NOT renormalised, forward.cu:123) */
  const float* cov3D_precomp;  /* [P,6] or NULL */
  const float* extra_channel;  /* [P] or NULL: a 4th per-Gaussian value blended with the same weights as the colour
                                  (gs_forward_render_x); replaces the reference's second rasterizer pass for the NIR
                                  albedo, mult-dwtgs/gaussian_renderer/__init__.py:151-258 */
  int32_t raw_activations;     /* 0: opacities / scales / rotations are the activated values, as the reference's rasterizer
                                  takes them.  1: they are the model's RAW rows (gaussian_model.py:60-78) and the kernels
                                  apply sigmoid / exp / F.normalize on the fly, with gs_activations_fwd's arithmetic bit for
                                  bit - so the activated copies are never written nor read.  Honoured by
                                  gs_forward_geometry and gs_backward_step only (the plain backward returns gradients
                                  with respect to the activated values: GS_E_UNSUPPORTED with this flag) */
  const float* extra_gain;     /* with raw_activations and extra_channel: device scalar; the blended 4th channel is then
                                  sigmoid(extra_channel[i]) * clamp(*extra_gain, 0.1, 10) - the multispectral model's raw NIR
                                  albedo and global gain (mult-dwtgs/gaussian_renderer/__init__.py:166-169), activated in
                                  the kernel like the other raw rows.  NULL: extra_channel holds the values to blend */
  const float* shs_rest;       /* NULL: `shs` holds all M coefficients of a Gaussian in one row.  Otherwise the SPLIT layout of the
                                  reference's GaussianModel (scene/gaussian_model.py:45-46,127-130): `shs` = _features_dc [P,1,3],
                                  `shs_rest` = _features_rest [P,M-1,3] - the kernels read the two rows where the model keeps
                                  them, so no torch.cat copy of 192 B per Gaussian runs per view (gaussian_model.py:get_features).
                                  Honoured by gs_forward_geometry and by gs_backward_step in its gradients-out form
                                  (GsStepState.grad_out_rest); the other entries return GS_E_UNSUPPORTED */
} GsGaussians;

/* Caller-owned scratch.  geom and img sizes depend on (P, W, H); binning on the capacity in
 * instances (Gaussian x tile pairs).  The three buffers are the private fwd<->bwd contract
 * (the reference's geomBuffer / binningBuffer / imgBuffer byte tensors): backward must get
 * back exactly the bytes forward wrote. */
typedef struct GsScratch {
  void* geom;
  size_t geom_bytes;
  void* img;
  size_t img_bytes;
  void* binning;
  size_t binning_bytes;
  int64_t binning_capacity; /* instances the binning buffer was sized for */
  const uint32_t* tile_order_hint; /* optional (NULL: image order): launch order of the forward blend's tiles, as
                                      gs_export_tile_order returned it for an earlier view of the SAME image size -
                                      typically the previous visit of this camera, or just the previous view.  Pure
                                      scheduling (longest tile first, see gs_export_tile_order): outputs do not depend on it.
                                      Must be an unmodified export - every tile exactly once. */
  const float* tile_depth_limit;   /* optional (NULL: none), only read when GsView.tile_cull is set; gs_tile_depth_limit_floats(W, H) floats: per tile, the view
                                      depth beyond which (plus a 5 % + 0.02 margin) no (tile, Gaussian) pair is emitted -
                                      gs_export_tile_stop_depth of an earlier forward of the SAME camera.  The forward checks
                                      that the cut lists were long enough (gs_forward_status.trunc_failed == 0); when they
                                      were, every output and gradient equals that of the uncut lists.  When the flag is set
                                      the outputs are NOT valid: repeat gs_forward_geometry + gs_forward_render without
                                      limits.  The same pointer and contents must be passed to both forward calls. */
  uint32_t* tile_order_out;        /* optional: gs_forward_render also writes what gs_export_tile_order would return here */
  float* tile_depth_limit_out;     /* optional: ... and what gs_export_tile_stop_depth would return here (one launch serves
                                      both; the buffers may be the ones passed as tile_order_hint / tile_depth_limit: they
                                      are read before they are written; a forward that overflowed its binning capacity
                                      writes neither, so that it can be repeated with the same hint and bounds) */
  int32_t binned;                  /* 1: gs_forward_bin has already built the instance lists of this view in these buffers;
                                      gs_forward_render* then only blends.  Written by the library: gs_forward_geometry
                                      clears it (a new geometry state invalidates any lists), gs_forward_bin sets it - a
                                      caller that re-uses one GsScratch across views need not touch it */
  int32_t defer_tile_order;        /* 1: gs_forward_render* stops after the blend; the caller runs gs_forward_tile_order (the forward's
                                      last kernel: launch order of the backward, the camera's next hints / depth bounds, status_host)
                                      itself - on another stream, beside whatever follows the forward, ordered before gs_backward* */
  const uint32_t* step_tag;        /* optional, device: gs_forward_status then copies TWELVE words out - the four status words,
                                      four reserved ones, *step_tag, three reserved - so that a caller that replays a
                                      captured graph of the step can tell, by polling its pinned block for the tag it
                                      uploaded before the replay, that the status of THAT replay has arrived (no stream
                                      synchronisation, no event inside the graph) */
  const float* tile_depth_limit_slack; /* optional, device, one float >= 1 (NULL: 1): the bounds written to tile_depth_limit_out
                                      are multiplied by it.  A caller whose model moves fast between two visits of a camera (early
                                      training) raises it for the cameras whose limits fail and lowers it again when they hold:
                                      longer lists, fewer repeated views.  Device memory so that a replayed graph sees changes. */
  uint32_t* status_host;           /* optional: 16 words of PINNED, device-mapped host memory (hipHostMalloc / torch pin_memory).
                                      gs_forward_render* then delivers what gs_forward_status would - words 0-3 = num_rendered,
                                      overflow, trunc_failed, largest region count; with step_tag also word 8 = *step_tag and word 9 =
                                      GS_STATUS_CHECK - from its LAST kernel, with system-scope stores: no copy command, no extra
                                      launch on the stream.  Complete when the call's work has completed; a host that polls for
                                      the tag accepts the block when the check word matches, as with gs_forward_status. */
} GsScratch;

/* Gradient outputs of gs_backward (rasterize_points.cu:163-178).  All are written in full by
 * the call (rows of culled Gaussians become 0) - the caller need not zero them.  Any pointer
 * may be NULL when that gradient is not wanted, except that dL_dsh / dL_dcolors follow the
 * colour mode and dL_dscales+dL_drotations / dL_dcov3D follow the covariance mode. */
typedef struct GsGrads {
  float* dL_dmeans3D;   /* [P,3] */
  float* dL_dmeans2D;   /* [P,3]  (z = 0) */
  float* dL_dsh;        /* [P,M,3] */
  float* dL_dcolors;    /* [P,3] */
  float* dL_dopacity;   /* [P] */
  float* dL_dscales;    /* [P,3] */
  float* dL_drotations; /* [P,4] */
  float* dL_dcov3D;     /* [P,6] */
  float* dL_dextra;     /* [P]  gradient of GsGaussians.extra_channel (gs_backward_x only), may be NULL */
} GsGrads;

int gs_abi_version(void);
/* Human readable build string: arch, compiler, kernel variants. */
const char* gs_build_info(void);
/* sizeof() of a public struct as this library was compiled: 0 GsView, 1 GsGaussians, 2 GsScratch, 3 GsGrads, 4 GsStepState,
 * 5 GsLgdwtParams, 6 GsAdamSeg; 0 for any other index.  A binding written in another language (the ctypes mirror of
 * gsplat_amd/capi.py, a cgo / JNI struct) checks its own layout against it at load time. */
size_t gs_struct_bytes(int32_t which);

/* out[0..2] = bytes needed for geom, img, binning given capacity R_capacity instances.
 * workspace_bytes (may be NULL) = bytes gs_backward needs as its workspace. */
int gs_scratch_bytes(int32_t P, int32_t W, int32_t H, int64_t R_capacity, size_t out[3],
                     size_t* backward_workspace_bytes);

/* Phase 1 of forward: per-Gaussian preprocess (cull, project, cov3D, EWA cov2D, conic,
 * radius, tile rect, SH->RGB) and the prefix sum of tiles_touched.  Writes radii[P].
 * num_rendered is left in the geom header on the device; if num_rendered_host != NULL it is
 * also copied there with hipMemcpyAsync on `stream` (pass pinned memory, then wait on the
 * stream/an event before reading it).  Does not block the host. */
int gs_forward_geometry(const GsView* view, const GsGaussians* g, GsScratch* scratch,
                        int32_t* radii, int32_t* num_rendered_host, void* stream);

/* Phase 2 of forward: duplicate (tile|depth) keys, stable radix sort on the low
 * 32+ceil_log2(T) key bits, tile ranges, front-to-back alpha blend.  Reads num_rendered from
 * the geom header on the device.  If it exceeds scratch->binning_capacity nothing is blended,
 * an overflow flag is left in the geom header and the image outputs are undefined: the caller
 * (who learns num_rendered from phase 1) re-runs this phase with a larger binning buffer.
 * out_color [3,H,W], out_invdepth [H,W] (may be NULL). */
int gs_forward_render(const GsView* view, const GsGaussians* g, GsScratch* scratch,
                      float* out_color, float* out_invdepth, void* stream);

/* The binning stage of gs_forward_render on its own (instance lists + tile ranges), followed by an asynchronous copy of
 * the four status words of gs_forward_status to status_host (pinned; may be NULL).  A caller that wants to know whether
 * the lists fitted the binning capacity without waiting for the blend calls this, records an event, calls
 * gs_forward_render* with scratch->binned = 1 and waits for the event while the blend runs. */
int gs_forward_bin(const GsView* view, const GsGaussians* g, GsScratch* scratch, uint32_t* status_host, void* stream);

/* gs_forward_render with a 4th blended channel: out_extra[H,W] = sum_i extra_i alpha_i T_i + T_final * bg[0], i.e.
 * channel 0 of a second rasterizer pass with colors_precomp = extra.repeat(1,3) - what the reference's render_nir
 * keeps (mult-dwtgs/gaussian_renderer/__init__.py:190-256) - sharing preprocess, binning and the alpha evaluation
 * with the colour pass.  out_extra == NULL is gs_forward_render. */
int gs_forward_render_x(const GsView* view, const GsGaussians* g, GsScratch* scratch,
                        float* out_color, float* out_invdepth, float* out_extra, void* stream);

/* Backward of the whole rasterizer.  num_rendered is the value forward produced.
 * dL_dinvdepth may be NULL (then no inverse-depth gradient path runs).
 * workspace: >= backward_workspace_bytes from gs_scratch_bytes (per Gaussian one 128-byte row of float64 blend sums -
 * stage 1 - and one 80-byte record of stage 2's float64 covariance chain).
 * Accuracy (fp32 outputs): every gradient tensor within 1e-4 of its largest entry of the reference algorithm evaluated
 * in exact arithmetic on the same blend sums.  For dL_dscales / dL_drotations that is a stronger statement than "1e-4 of
 * the reference's fp32 run": the chain conic -> cov2D -> cov3D -> (scale, quaternion) (backward.cu:248-275, 330-393)
 * cancels 3-4 digits for needle-shaped footprints, and the reference's own fp32 formula is 1.0e-3 / 2.0e-3 of the
 * tensor's max away from the float64 image of its inputs at 1 M Gaussians / 1080p.  Here the blend sums are accumulated
 * in float64 rows and that chain is evaluated in float64 (csrc/gs_backward_math.h), so these two tensors are the exact
 * image rounded once and repeat from run to run; tests/test_gpu_fullsize.py asserts it, DESIGN.md section 2 has the analysis. */
int gs_backward(const GsView* view, const GsGaussians* g, const int32_t* radii,
                const GsScratch* scratch, int64_t num_rendered, const float* dL_dcolor,
                const float* dL_dinvdepth, const GsGrads* grads, void* workspace,
                size_t workspace_bytes, void* stream);

/* gs_backward with the image gradient of the 4th channel (dL_dextra_img [H,W], NULL = gs_backward); geometry
 * gradients are those of the sum of both passes, grads->dL_dextra[P] receives the channel's own gradient. */
int gs_backward_x(const GsView* view, const GsGaussians* g, const int32_t* radii,
                  const GsScratch* scratch, int64_t num_rendered, const float* dL_dcolor,
                  const float* dL_dinvdepth, const float* dL_dextra_img, const GsGrads* grads,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The older rasterizer generation used by FSGS / DNGaussian (SURVEY 8f-4;
 * FSGS/submodules/diff-gaussian-rasterization-confidence: `rasterize_gaussians` returns colour, depth, alpha -
 * rasterize_points.cu, forward.cu:262-380; backward takes their three image gradients, backward.cu:414-600).
 * Same geometry phase (gs_forward_geometry with antialiasing = 0: the 0.3 low-pass is unconditional there).
 * out_depth[H,W] = sum depth_i alpha_i T_i (view-space z), out_alpha[H,W] = sum alpha_i T_i; no inverse depth. */
int gs_forward_render_fsgs(const GsView* view, const GsGaussians* g, GsScratch* scratch,
                           float* out_color, float* out_depth, float* out_alpha, void* stream);
int gs_backward_fsgs(const GsView* view, const GsGaussians* g, const int32_t* radii,
                     const GsScratch* scratch, int64_t num_rendered, const float* dL_dcolor,
                     const float* dL_ddepth, const float* dL_dalpha, const GsGrads* grads,
                     void* workspace, size_t workspace_bytes, void* stream);

/* present[i] = (view-space z of means3D[i]) > 0.2   (rasterizer_impl.cu:54-66) */
int gs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix,
                    const float* projmatrix, uint8_t* present, void* stream);

/* ---- debug / parity exports: copy internal state into plain arrays (any pointer may be
 * NULL).  Used by the parity tests to compare against the oracle; not on the hot path. ---- */
int gs_export_geom(const GsScratch* scratch, int32_t P, float* depths, float* means2D /*[P,2]*/,
                   float* cov3D /*[P,6]*/, float* conic_opacity /*[P,4]*/, float* rgb /*[P,3]*/,
                   uint8_t* clamped /*[P,3]*/, uint32_t* tiles_touched, uint32_t* point_offsets,
                   void* stream);
int gs_export_binning(const GsScratch* scratch, int64_t num_rendered, uint64_t* keys_sorted,
                      uint32_t* point_list, void* stream);
int gs_export_img(const GsScratch* scratch, int32_t W, int32_t H, float* final_T,
                  uint32_t* n_contrib, uint32_t* ranges /*[T,2]*/, void* stream);

/* The tail of a single-GPU train step fused into the backward (LGDWT-GS/train.py:262-288,
 * scene/gaussian_model.py:40-60,183-193,471-473): gs_backward, then IN THE SAME per-Gaussian kernel the backward of the
 * parameter activations (exp / F.normalize / sigmoid), the densification statistics of the view and the Adam update of
 * the Gaussian's 59 parameters - the 236 B/Gaussian of gradients are never written to memory nor read back, and three
 * launches (activation backward, statistics, Adam) disappear.  Valid when every gradient is final after this one view
 * (one camera per optimizer step on one GPU - the reference's own loop); the data-parallel step keeps gs_backward +
 * all-reduce + gs_adam_step.  Arithmetic per element is that of gs_activations_bwd, gs_densify_stats and gs_adam_step.
 *
 * The raw parameter rows live in the caller's buffers; g->means3D must BE st->xyz and g->shs must BE st->features (no
 * activation in between), g->scales / rotations / opacities are the activated tensors handed to the forward;
 * g->M == 16, no precomputed colours / covariances. */
typedef struct GsStepState {
  float* xyz;       /* [P,3]   raw parameters, updated in place */
  float* features;  /* [P,16,3] */
  float* opacity;   /* [P]   pre-sigmoid */
  float* scaling;   /* [P,3] pre-exp */
  float* rotation;  /* [P,4] un-normalised */
  float* m[5];      /* Adam exp_avg of the five rows above, same shapes */
  float* v[5];      /* Adam exp_avg_sq */
  float lr[6];      /* xyz, features DC (k = 0), features rest (k = 1..15), opacity, scaling, rotation */
  int32_t step[5];  /* the row's 1-based Adam step count (bias correction); 0 = row skipped: parameter and moments untouched */
  float beta1, beta2, eps;
  float* max_radii2D;        /* [P] densification statistics of train.py:266-268 (all three or none) */
  float* xyz_gradient_accum; /* [P] */
  float* denom;              /* [P] */
  const float* coef_dev;      /* optional, device, 11 floats (15 with st->extra: + the same two constants of the sixth row and
                                of the gain): lr[c] / (1 - beta1^t) for the six learning-rate classes, then
                                1 / sqrt(1 - beta2^t) for the five rows.  NULL: computed from lr[] / step[] on the host and
                                passed as kernel arguments.  A caller that REPLAYS a captured graph of the step must use
                                this form (kernel arguments are frozen at capture) and refresh the buffer before each
                                replay; step[] then only says which rows are skipped. */
  const double* rows_override; /* parity probe, normally NULL: [P,16] float64 blend sums to use INSTEAD of running stage 1 (layout:
                                gs_backward_from_rows) - lets a test hand the fused tail and the three-kernel tail the
                                very same sums and compare them bit for bit */
  /* ---- data-parallel form (N > 1): gradients OUT instead of the Adam step ----
   * grad_out[0] != NULL: the gradients with respect to the RAW rows (xyz [P,3], features [P,16,3], opacity [P], scaling [P,3],
   * rotation [P,4]: the five pointers, typically into one flat buffer that is then all-reduced) are WRITTEN there and no
   * parameter or moment is touched; m[] / v[] / lr[] / step[] are ignored.  The statistics are then this view's
   * INCREMENTS, assigned: xyz_gradient_accum[i] = |dL/dmean2D_i| (0 when culled), denom[i] = visible ? 1 : 0 (summed over
   * ranks by the caller); max_radii2D is updated in place as always.  fail_flag (device, may be NULL): set to 1.0f when
   * the forward had flagged overflow or trunc_failed - the call then writes ZERO statistic increments and no gradients -
   * else to 0.0f; reduced (sum) with the gradients it tells every rank whether any rank's view was invalid
   * (gs_adam_step_gated takes it as its gate). */
  float* grad_out[5];
  float* fail_flag;
  /* ---- two-phase step (single GPU; any list mode) ----
   * 0: gs_backward_step handles every Gaussian.  2: only the Gaussians that emitted instances in this view - the caller has
   * already run gs_step_uninstanced on the same state for the others. */
  int32_t phase;
  /* The workspace rows between steps (single GPU, rows_override == NULL).  0: gs_backward_step clears the rows of the
   * Gaussians with instances before the blend accumulates into them and leaves the sums behind (a probe may read them).
   * 1: as 0, and the per-Gaussian kernel zeroes every row it has consumed - also when the step is a no-op on the device -
   * so the workspace is all zero again when the call has run.  2: the caller guarantees exactly that state on entry (same
   * workspace, last used by a call with rows_clean != 0, or zero-filled): no clear launch at all. */
  int32_t rows_clean;
  void* phase1_done; /* with phase = 2: a hipEvent_t recorded behind gs_step_uninstanced on its stream, or NULL.  gs_backward_step
                        makes its stream wait for it AFTER the backward blend has been launched - the two then run side by
                        side - and before the per-Gaussian kernel.  NULL: the caller has ordered the two calls itself. */
  /* ---- 4th blended channel (gs_backward_step_x; multispectral train step, mult-dwtgs/train_nir.py): g->extra_channel must
   * BE st->extra (the raw per-Gaussian parameter, e.g. the NIR albedo before its sigmoid) and g->extra_gain st->gain.  The
   * raw row is stepped like the five rows above (sixth row: lr_extra, step_extra, coef_dev[11..12]); the global gain - one
   * scalar, dL/dgain = sum_i dL/dextra_i sigmoid(raw_i) inside the clamp, 0 outside - by the first workgroup of the
   * per-Gaussian kernel, from one partial sum per 256 Gaussians that the float64 chain kernel left behind the records in
   * the workspace, added in index order (deterministic; lr_gain, step_gain, coef_dev[13..14]).  Data-parallel form:
   * grad_out_extra [P] / grad_out_gain [1] receive the gradients instead. */
  float* extra;    /* [P] or NULL (then none of the fields below is read) */
  float* extra_m;
  float* extra_v;
  float* gain;     /* device scalar */
  float* gain_m;
  float* gain_v;
  float lr_extra, lr_gain;
  int32_t step_extra, step_gain;
  float* grad_out_extra;
  float* grad_out_gain;
  /* data-parallel form, optional: [P] bytes, 1 where this view's gradients of the Gaussian may be non-zero (it emitted
   * instances), 0 where every one of its gradient floats was written as zero.  The union of the ranks' masks is what a
   * sparse exchange has to move (gsplat_amd.trainer: pack the union's rows, all-reduce, scatter back). */
  uint8_t* grad_mask;
  /* Adam-inside-the-backward form, optional: one byte per block of GS_STEP_BLOCK (256) consecutive Gaussians, 1 = "BOTH
   * Adam moments of EVERY row of every Gaussian of the block are +0" (the caller's claim; gsplat_amd.trainer derives it
   * from the moments themselves).  The zero-gradient update of such a block changes no bit of parameters or moments
   * (m' = v' = +0, p' = p - lr * (0 / eps) = p; needs eps > 0), so the kernels skip its parameter and moment traffic;
   * a block in which a Gaussian receives a gradient gets its byte cleared by the kernel that writes its moments.
   * View statistics are kept for every Gaussian as without it.  With the model's rows in spatial order (neighbours in
   * memory = neighbours in space) the Gaussians no camera reaches fill whole blocks. */
  uint8_t* dormant;
  /* "sparse_adam" (train.py:282-284): 1 = a Gaussian with radii <= 0 in this view is not stepped at all - its parameters and
   * both moments keep their bits (with the default optimizer it takes a zero-gradient step: the moments decay, the
   * parameter moves by the old momentum).  Visible Gaussians without instances (culled spans, depth limits) still take
   * their zero-gradient step.  Blocks of 256 rows without a visible Gaussian cost no parameter or moment traffic. */
  int32_t sparse;
  /* Gradients-out form with the split SH layout (GsGaussians.shs_rest != NULL): grad_out[1] receives the gradient of
   * _features_dc [P,1,3] and grad_out_rest that of _features_rest [P,M-1,3] - each a contiguous tensor as autograd's
   * AccumulateGrad wants it (a strided slice of one [P,16,3] buffer is copied again).  NULL with the one-row layout. */
  float* grad_out_rest;
} GsStepState;
#define GS_STEP_BLOCK 256
int gs_backward_step(const GsView* view, const GsGaussians* g, const int32_t* radii,
                     const GsScratch* scratch, int64_t num_rendered, const float* dL_dcolor,
                     const float* dL_dinvdepth, const GsStepState* st, void* workspace,
                     size_t workspace_bytes, void* stream);
/* gs_backward_step behind gs_forward_render_x: dL_dextra_img [H,W] is the image gradient of the 4th channel; st->extra et
 * al. describe its parameters (see GsStepState). */
int gs_backward_step_x(const GsView* view, const GsGaussians* g, const int32_t* radii,
                       const GsScratch* scratch, int64_t num_rendered, const float* dL_dcolor,
                       const float* dL_dinvdepth, const float* dL_dextra_img, const GsStepState* st,
                       void* workspace, size_t workspace_bytes, void* stream);

/* The part of gs_backward_step that does not wait for the loss: a Gaussian that emitted no instance in this view
 * (tiles_touched == 0: off screen, culled, or cut by the depth limits - four fifths of them on depth-limited lists) has
 * zero gradients whatever the image looks like, so its view statistics and its Adam step (zero gradient: the moments decay,
 * the parameter moves by the old momentum - torch.optim.Adam does the same) need nothing but the forward's geometry stage.
 * Call it on a second stream, ordered after gs_forward_render, right before gs_backward_step with st->phase = 2 on the
 * first: it then runs CONCURRENTLY with stage 1 of the backward, the blend - a pure HBM stream next to a kernel that is
 * bound by vector issue and leaves HBM idle (next to the criterion's kernels it gains nothing: they slow down by what it
 * takes, measured).  gs_backward_step steps the Gaussians with instances; its per-Gaussian kernel must be ordered after
 * this call (st->phase1_done; the two split the float4s of the parameter rows between them).  Like gs_backward_step a no-op on the device when the forward had flagged
 * overflow or trunc_failed - so it must not start before the forward BLEND has finished.  Same element arithmetic: the
 * two calls together leave the bits gs_backward_step (phase 0) leaves.  With tile_cull = 0 every visible Gaussian has
 * instances, and the Gaussians without are the ones outside the frustum (59 % of bench.py's scene: with the rows in spatial
 * order most of their blocks are dormant, see GsStepState.dormant).  GS_E_UNSUPPORTED in the data-parallel form.
 * Although `scratch` is const, the call WRITES one word of the geometry header it points at: the block cursor its
 * workgroups draw their work from (zeroed by gs_forward_geometry, re-armed by the call's last workgroup).  Every call is
 * therefore one full zero-gradient step of the uninstanced Gaussians - calling it twice behind one forward steps them
 * twice; two calls on the SAME scratch must not overlap in time (they would share the cursor). */
int gs_step_uninstanced(const GsView* view, const GsGaussians* g, const int32_t* radii,
                        const GsScratch* scratch, const GsStepState* st, void* stream);

/* Stage 2 of gs_backward on its own (parity export): the per-Gaussian chain rule from given sums of the blend
 * backward.  rows [P,16] FLOAT64 (device): mean2D.x, mean2D.y, conic.xx, conic.xy, conic.yy, opacity, r, g, b, depth slot,
 * 4th channel, 5 pad - what stage 1 leaves in the workspace (backward.cu:593-635 accumulates the same ten sums with
 * float atomics; here a tile's fp32 totals are added into float64 slots, so the sums do not depend on the order the
 * tiles arrive in).  depth_mode: 0 none, 1 the depth slot is dL/d(inverse depth) (dr_aa), 2 dL/d(depth) (FSGS generation).
 * Lets a test feed both implementations the SAME sums and so separate the (ill-conditioned)
 * conic -> scale / rotation chain from the accumulation that precedes it.  workspace: as gs_backward (stage 2 keeps its
 * per-Gaussian records there). */
int gs_backward_from_rows(const GsView* view, const GsGaussians* g, const int32_t* radii,
                          const GsScratch* scratch, const double* rows, int32_t depth_mode,
                          const GsGrads* grads, void* workspace, size_t workspace_bytes, void* stream);

/* The order in which the backward blend of this view takes its tiles: 8 * ceil(T/8) entries, entry b = the tile of
 * workgroup b (0xFFFFFFFF: none), per XCD band of the image by decreasing number of list entries the tile visits.  One
 * wave owns a tile and runs at the pace of its dependent instruction chain, so handing out the longest tiles first
 * shortens the kernel's tail (csrc/gs_render_fwd_wave.hip).  The forward blend's cost per tile is nearly the same
 * quantity, unknown before it has run - but a good predictor is the previous view: pass the export back in as
 * GsScratch.tile_order_hint.  out: device, 8 * ceil(T/8) uint32. */
int gs_export_tile_order(const GsScratch* scratch, int32_t W, int32_t H, uint32_t* out, void* stream);

/* Depth-limited emission.  Most (tile, Gaussian) pairs of a trained scene lie behind the depth at which the tile's
 * pixels saturate (forward.cu:326-328 stops there); they are keyed, sorted and stored but never blended.
 * gs_export_tile_stop_depth returns, per tile, the view depth of the list entry at which the last forward on this
 * scratch stopped (+inf: some pixel of the tile never saturated - the whole list matters).  Handed back as
 * GsScratch.tile_depth_limit on the next forward of the same camera, pairs deeper than that (with margin) are not emitted.
 * out: device, gs_tile_depth_limit_floats(W, H) floats: per tile the largest stop depth of its 3 x 3 neighbourhood, then
 * the largest of those per aligned run of four tiles of a row (why: csrc/gs_tilecull.h).  "No limit" = all +inf. */
size_t gs_tile_depth_limit_floats(int32_t W, int32_t H);
int gs_export_tile_stop_depth(const GsScratch* scratch, int32_t W, int32_t H, float* out, void* stream);
/* Asynchronous read-back of the last forward's counters into host memory (pinned for a true async copy):
 * out[0] = instances emitted (num_rendered), out[1] = overflow (binning capacity exceeded: nothing was blended),
 * out[2] = trunc_failed (a depth-limited tile ran out of list entries: outputs invalid, see GsScratch.tile_depth_limit),
 * out[3] = 0, or with GsView.tile_cull = 2 the largest number of Gaussians any 4 x 4-tile region received (the region
 * buckets hold binning_capacity / regions entries each, at most 16 384: a larger value set `overflow` too).  Valid after
 * `stream` has been synchronised.
 * With GsScratch.step_tag set the copy is 12 words: out[8] = *step_tag as the device held it when this call ran, out[9] =
 * GS_STATUS_CHECK(out[8], out[0], out[1], out[2]).  A host that polls out[8] for its tag instead of synchronising must not
 * trust out[0..2] until out[9] matches them: nothing orders the bytes of a device-to-host copy for a concurrent reader. */
#define GS_STATUS_CHECK(tag, num_rendered, overflow, trunc_failed) \
  ((((uint32_t)(tag)) * 2654435761u) ^ (((uint32_t)(num_rendered)) * 40503u) ^ (((uint32_t)(overflow)) << 30) ^ \
   (((uint32_t)(trunc_failed)) << 31) ^ 0x5bd1e995u)
int gs_forward_status(const GsScratch* scratch, uint32_t* out /*[4] or [12] host*/, void* stream);
/* The last kernel of gs_forward_render* on its own (GsScratch.defer_tile_order = 1): same arguments, any stream that is ordered
 * after the render's. */
int gs_forward_tile_order(const GsView* view, const GsScratch* scratch, void* stream);
/* Developer statistics of the last forward on this scratch: what the backward blend's loop meets.  out (device, 8 x u64,
 * zeroed by the caller) += [entries visited, entries with a valid pixel, (entry, quadrant) pairs with a valid pixel,
 * valid (entry, pixel) pairs, tiles with work, list entries of those tiles, 0, 0].  tests/tools/blend_stats.py */
int gs_debug_blend_stats(const GsScratch* scratch, int32_t P, int32_t W, int32_t H, uint64_t* out, void* stream);
/* keys_sorted for the product's lists (the product keeps no tile-key array in any list mode: gs_export_binning returns
 * GS_E_UNSUPPORTED when asked for keys): keys_sorted[i] = (tile << 32 | depth bits) of list entry i, rebuilt from ranges[] and
 * the Gaussians' depths.  tile_cull = 0 / 1: point_list is the reference's, tile after tile.  tile_cull = 2 (region
 * binning): the lists of different tiles lie in point_list in no particular order, inside a tile the order is the reference's. */
int gs_export_binning_region(const GsScratch* scratch, int32_t W, int32_t H, int64_t num_rendered, uint64_t* keys_sorted,
                             uint32_t* point_list, void* stream);

/* ---- simple-knn ---- */
/* out[i] = mean of the 3 smallest squared distances from point i to the other points.
 * tmp: >= gs_knn_tmp_bytes(P) bytes of device scratch. */
size_t gs_knn_tmp_bytes(int32_t P);
int gs_knn_mean_dist2(const float* xyz /*[P,3]*/, int32_t P, float* out /*[P]*/, void* tmp,
                      size_t tmp_bytes, void* stream);
/* FSGS's fork of simple-knn (FSGS/submodules/simple-knn/simple_knn.cu:132-189, spatial.cu) also returns the three
 * neighbours themselves: nearest[3 i + j] = index of the j-th nearest point of point i (nearest first; ties and
 * the traversal order are the reference's).  nearest == NULL is gs_knn_mean_dist2. */
int gs_knn_mean_dist2_idx(const float* xyz /*[P,3]*/, int32_t P, float* out /*[P]*/,
                          int32_t* nearest /*[P,3]*/, void* tmp, size_t tmp_bytes, void* stream);

/* ---- losses (images are [C,H,W] or [N,C,H,W] contiguous fp32) ---- */

/* sums[0] = sum |a-b| over n elements (atomically added: zero it first).
 * grad (may be NULL) = coef * sign(a-b). */
int gs_l1_fwd(const float* a, const float* b, int64_t n, float* sum, void* stream);
int gs_l1_bwd(const float* a, const float* b, int64_t n, float coef, float* grad_a,
              int32_t accumulate, void* stream);

/* One level of the Haar analysis ('db1', mode 'symmetric': odd sizes repeat the last sample).
 * x [NC,H,W] -> ll, lh, hl, hh each [NC, ceil(H/2), ceil(W/2)].  lh = low along W, high along H. */
int gs_dwt_haar_fwd(const float* x, int32_t NC, int32_t H, int32_t W, float* ll, float* lh,
                    float* hl, float* hh, void* stream);
/* adjoint: dx [NC,H,W] (=, not +=) from the four band gradients (any may be NULL = zero). */
int gs_dwt_haar_bwd(const float* dll, const float* dlh, const float* dhl, const float* dhh,
                    int32_t NC, int32_t H, int32_t W, float* dx, void* stream);

/* Fused global 2-level DWT L1 loss (train.py:132-164).
 * band_sums[8] += sum |band(pred) - band(gt)| for LL1,LH1,HL1,HH1,LL2,LH2,HL2,HH2 (zero first).
 * counts: level-1 bands have C*ceil(H/2)*ceil(W/2) elements, level-2 C*ceil(h1/2)*ceil(w1/2). */
int gs_dwt2_l1_fwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W,
                   float* band_sums, void* stream);
/* grad_pred (+= if accumulate) = sum_b coef[b] * DWT_b^T sign(band_b(pred-gt)); coef is a
 * DEVICE array of 8 floats (w_b * upstream / N_b). */
int gs_dwt2_l1_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W,
                   const float* coef_dev, float* grad_pred, int32_t accumulate, void* stream);

/* ELF map (loss_utils.py:336-366): elf_low = LL/(LL+LH+HL+HH+1e-8) with per-pixel channel-L1
 * of the level-1 bands, then bilinear x2 upsample (align_corners=False) to [H,W]. */
int gs_elf_map(const float* img, int32_t C, int32_t H, int32_t W, float* elf_low /*[h1,w1] tmp*/,
               float* elf /*[H,W]*/, void* stream);
/* means[L] = mean of elf over each non-overlapping patch (row-major patch order, remainder
 * dropped: F.unfold semantics, loss_utils.py:391-400). */
int gs_patch_means(const float* elf, int32_t H, int32_t W, int32_t patch, float* means,
                   void* stream);
/* Patch DWT loss over the selected patches (mask[L] != 0): sums[3] += sum |band(pred-gt)| for
 * LH1, HL1, HH1 restricted to selected patches (zero first). */
int gs_patch_dwt_fwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W,
                     int32_t patch, const uint8_t* mask, float* sums, void* stream);
int gs_patch_dwt_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W,
                     int32_t patch, const uint8_t* mask, const float* coef_dev /*[3]*/,
                     float* grad_pred, int32_t accumulate, void* stream);

/* Fused SSIM, 11-tap sigma=1.5 separable window, zero padding ("same").
 * img1,img2 [B,C,H,W]; ssim_map and the three partial-derivative maps (may be NULL when no
 * backward is wanted) have the same shape. */
int gs_ssim_fwd(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W,
                float C1, float C2, float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq,
                float* dm_dsigma12, void* stream);
int gs_ssim_bwd(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W,
                float C1, float C2, const float* dL_dmap, const float* dm_dmu1,
                const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1,
                void* stream);

/* ---- fused form of the LGDWT-GS criterion (LGDWT-GS/train.py:128-202): the per-term kernels above with
 * device-resident coefficients, one shared image-gradient buffer, SSIM's mean and the clamp(0,1) backward folded
 * in, and the loss composition itself (running-mean DWT scale included) done by one tiny kernel instead of
 * host arithmetic on .item() values. ---- */
int gs_l1_bwd_dev(const float* a, const float* b, int64_t n, const float* coef_dev /*[1]*/, float* grad_a,
                  int32_t accumulate, void* stream);
/* gs_l1_fwd + gs_dwt2_l1_fwd from ONE read of the two images: l1_sum[0] += sum |pred - gt|, band_sums[8] as
 * gs_dwt2_l1_fwd; and their two backward kernels as one write of the gradient image:
 * grad_pred (+= if accumulate) = l1_coef_dev[0] * sign(pred - gt) + the gs_dwt2_l1_bwd term. */
int gs_l1_dwt2_fwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, float* l1_sum,
                   float* band_sums, void* stream);
/* gs_l1_dwt2_fwd on clamp(raw, 0, 1), which is also written to clamped_out [C,H,W] (the clamp of
 * gaussian_renderer/__init__.py:119 without a pass of its own).  H and W multiples of 4 and 16-byte aligned planes, else
 * GS_E_UNSUPPORTED (clamp first and call gs_l1_dwt2_fwd). */
int gs_l1_dwt2_fwd_clamp(const float* raw, const float* gt, int32_t C, int32_t H, int32_t W, float* l1_sum,
                         float* band_sums, float* clamped_out, void* stream);
/* gs_l1_dwt2_fwd_clamp and gs_patch_dwt_fwd in one pass, gs_l1_dwt2_bwd and gs_patch_dwt_bwd in one: the patch term's sums
 * are the level-1 LH / HL / HH differences of the same 2 x 2 blocks, restricted to the selected patches (mask [H/ps * W/ps],
 * loss_utils.py:395-442), and its gradient rides on the same band signs.  patch_sums[3] +=; patch_coef_dev[3] as coef_dev of
 * gs_patch_dwt_bwd.  H, W and ps multiples of 4 and 16-byte aligned planes, else GS_E_UNSUPPORTED (use the separate calls). */
int gs_l1_dwt2_patch_fwd_clamp(const float* raw, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps,
                               const uint8_t* mask, float* l1_sum, float* band_sums, float* patch_sums,
                               float* clamped_out, void* stream);
int gs_l1_dwt2_patch_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps,
                         const uint8_t* mask, const float* l1_coef_dev /*[1]*/, const float* coef_dev /*[8]*/,
                         const float* patch_coef_dev /*[3]*/, float* grad_pred, int32_t accumulate, void* stream);
int gs_l1_dwt2_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W,
                   const float* l1_coef_dev /*[1]*/, const float* coef_dev /*[8]*/, float* grad_pred,
                   int32_t accumulate, void* stream);
/* Order-independent sums (round 4).  The *_fwd kernels above add each workgroup's sums to their targets with float atomics:
 * the totals then depend, in the last bits, on the order in which the workgroups finish.  These forms store the
 * workgroups' sums instead - partials[12 w + k] for workgroup w of gs_l1_dwt2_patch_fwd_clamp_p (k: 0..7 the band sums,
 * 8 the L1 sum, 9..11 the patch sums; gs_dwt_partials_count(C, H, W) rows; ps = 0 / mask = NULL: no patch term),
 * partials[w] for gs_l1_fwd_p (gs_l1_partials_count(n) entries) - and gs_lgdwt_combine_pp adds them to sums[] in index
 * order before it composes the loss: two runs of a train step give the same bits (tests/test_gpu_fused_step.py).
 * H, W (and ps) multiples of 4, 16-byte aligned planes, else GS_E_UNSUPPORTED (use the atomic forms). */
int64_t gs_dwt_partials_count(int32_t C, int32_t H, int32_t W);
int gs_l1_dwt2_patch_fwd_clamp_p(const float* raw, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps,
                                 const uint8_t* mask, float* partials, float* clamped_out, void* stream);
int64_t gs_l1_partials_count(int64_t n);
int gs_l1_fwd_p(const float* a, const float* b, int64_t n, float* partials, void* stream);
/* Depth regularisation of the step (LGDWT-GS/train.py:204-216): Ll1depth_pure = mean |(invDepth - mono_invdepth) * depth_mask|
 * over the n pixels of the rasterizer's inverse-depth output.  One pass gives both halves (either pointer may be NULL, not
 * both): partials[w] = workgroup w's sum of |(d - m) k| (gs_depth_l1_partials_count(n) entries; the caller adds them in
 * index order and divides by n), and grad[i] = coef * (coef_dev ? coef_dev[0] : 1) * sign((d - m) k) * k - the gradient of
 * weight * mean with coef = weight / n (sign(0) = 0 as torch.abs has it).  mask NULL: all ones.  grad is what
 * gs_backward* takes as dL_dinvdepth. */
int64_t gs_depth_l1_partials_count(int64_t n);
int gs_depth_l1(const float* invdepth, const float* mono_invdepth, const float* mask, int64_t n, float* partials, float coef,
                const float* coef_dev, float* grad, void* stream);
/* like gs_ssim_fwd but returns sum(ssim_map) (+=, zero it first) instead of the map.  One atomic per workgroup on
 * sum_out: prefer gs_ssim_fwd_partials below for large images (the atomics serialise: 98 vs 59 us at 1080p). */
int gs_ssim_fwd_sum(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float C1,
                    float C2, float* sum_out, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                    void* stream);
/* gs_ssim_fwd_sum without the atomic: workgroup w of the launch stores its share of sum(ssim_map) to partials[w]
 * (gs_ssim_partials_count() of them, all written); gs_lgdwt_combine_p adds them to sums[1] in a fixed order.
 * (One device-scope atomic per workgroup on ONE address is serialised at ~14 ns each on MI355X: 86 us at 1080p.) */
int64_t gs_ssim_partials_count(int32_t B, int32_t C, int32_t H, int32_t W);
int gs_ssim_fwd_partials(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float C1,
                         float C2, float* partials, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                         void* stream);
/* like gs_ssim_bwd with dL_dmap == coef_dev[0] everywhere; optionally adds to dL_dimg1 and then zeroes the
 * result where clamp_src (the un-clamped render) lies outside [0,1].  With clamp_src given, img1 must be
 * clamp(clamp_src, 0, 1) - the kernel then forms it from clamp_src instead of reading img1. */
int gs_ssim_bwd_uniform(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W,
                        const float* coef_dev, const float* dm_dmu1, const float* dm_dsigma1_sq,
                        const float* dm_dsigma12, float* dL_dimg1, int32_t accumulate, const float* clamp_src,
                        void* stream);
typedef struct GsLgdwtParams {
  float lambda_dssim;        /* 0.2 */
  float n_pix;               /* C*H*W */
  float n_band1, n_band2;    /* elements per level-1 / level-2 band */
  float dwt_w[8];            /* LL1 LH1 HL1 HH1 LL2 LH2 HL2 HH2 */
  float patch_w[3];          /* w_lh, w_hl, (w_lh+w_hl)/2 */
  float patch_weight;        /* beta = 0.1 */
  float patch_elems_per_sel; /* C * (patch/2)^2 */
  int32_t dwt_enable, patch_enable;
  int32_t reset_sums;        /* 1: the call zeroes sums[0..12] after it has read them (the accumulators of the next view:
                                a caller that keeps ONE sums buffer per camera, word 13 preset, then needs no fill kernel) */
  int32_t custom_base;       /* 1: base = w_l1 * L1 + w_ssim * (1 - SSIM) instead of (1 - lambda) L1 + lambda (1 - SSIM): the NIR
                                term of mult-dwtgs/train_nir.py:96-104 (nir_weight * (L1 + 0.2 (1 - SSIM)),
                                mult-dwtgs/utils/loss_utils.py:93-144) through the same kernels */
  float w_l1, w_ssim;
} GsLgdwtParams;
/* sums[16]: 0 l1 | 1 ssim | 2..9 bands | 10..12 patch | 13 selected patches.  running_mean: device scalar,
 * updated in place (train.py:193-195).  out[24]: 0 loss 1 base 2 dwt 3 patch 4 dwt_scale 5 l1 6 ssim 7 the running mean
 * before the call;
 * 8 c_l1, 9 c_ssim, 10..17 c_band, 18..20 c_patch = dLoss/d(term sum). */
int gs_lgdwt_combine(const float* sums, float* running_mean, const GsLgdwtParams* params /*host*/, float* out,
                     void* stream);
/* the same with the SSIM sum given as sums[1] + sum(ssim_partials[0..n_partials)) */
int gs_lgdwt_combine_p(const float* sums, const float* ssim_partials, int64_t n_partials, float* running_mean,
                       const GsLgdwtParams* params /*host*/, float* out, void* stream);
/* ... and with the DWT / L1 / patch sums given as sums[] + the per-workgroup partials of gs_l1_dwt2_patch_fwd_clamp_p
 * (n_dwt rows of 12) and of gs_l1_fwd_p (n_l1 entries), added in index order */
int gs_lgdwt_combine_pp(const float* sums, const float* ssim_partials, int64_t n_partials, const float* dwt_partials,
                        int64_t n_dwt, const float* l1_partials, int64_t n_l1, float* running_mean,
                        const GsLgdwtParams* params /*host*/, float* out, void* stream);

/* ---- optimiser (caller side of the path): fused Adam over ONE flat fp32 parameter buffer.
 * Replaces torch.optim.Adam(lr=0, eps=1e-15) with per-group learning rates,
 * LGDWT-GS/scene/gaussian_model.py:183-193 + train.py:279-288.  Segment k covers elements [begin, end);
 * element i uses lr_a, or lr_b when period > 0 and (i - begin) % period >= split (interleaved SH DC / rest).
 * step: the segment's own 1-based step count for the bias correction, 0 = the call's `step` (torch keeps the
 * counter per parameter: a group whose tensor was just replaced - reset_opacity, gaussian_model.py:258-261 - has
 * no gradient that iteration, is skipped by optimizer.step() and falls one step behind the others).
 * Elements covered by no segment are left untouched (parameter AND moments). */
typedef struct GsAdamSeg {
  int64_t begin, end;
  float lr_a, lr_b;
  int32_t period, split;
  int32_t step;
  int32_t row_width; /* gs_adam_step_masked: floats per Gaussian in this segment (element i belongs to row (i - begin) / row_width);
                        0: the segment is not masked */
} GsAdamSeg;
/* segs is a HOST array (<= 8 entries, copied into the launch); step = 1-based iteration for the bias correction. */
int gs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                 const GsAdamSeg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                 void* stream);
/* the same, but a no-op on the device when *gate != 0 (gate: device float, may be NULL = gs_adam_step): the data-parallel
 * step passes the all-reduced GsStepState.fail_flag - if any rank rendered from lists that proved too short, no replica
 * steps, and every rank repeats the step once its host has seen the flag */
int gs_adam_step_gated(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const GsAdamSeg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                       const float* gate, void* stream);
/* "sparse_adam" (LGDWT-GS/train.py:282-284: `visible = radii > 0; optimizer.step(visible, N)`; the optimizer class itself,
 * SparseGaussianAdam, lives in a branch of the rasterizer that the reference does not vendor): the same update, but only for
 * the rows with row_mask[row] > 0 (device, float [P]; segments with row_width > 0) - the parameters AND both moments of
 * every other row keep their bits.  gate as gs_adam_step_gated (may be NULL). */
int gs_adam_step_masked(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const GsAdamSeg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                        const float* gate, const float* row_mask, void* stream);

/* ---- per-Gaussian elementwise work of the train step outside the rasterizer ----
 * Parameter activations of GaussianModel (LGDWT-GS/scene/gaussian_model.py:40-60,102-117): scales = exp(scaling),
 * rotations = F.normalize(rotation) (eps 1e-12), opacities = sigmoid(opacity); all [P,k] contiguous fp32 (any
 * 4-byte alignment).  One kernel instead of ~5 torch kernels. */
int gs_activations_fwd(const float* scaling /*[P,3]*/, const float* rotation /*[P,4]*/,
                       const float* opacity /*[P]*/, int32_t P, float* scales_out, float* rotations_out,
                       float* opacities_out, void* stream);
/* Their backward (the autograd formulas of exp / normalize / sigmoid), written (=) into d_*; one kernel. */
int gs_activations_bwd(const float* scaling, const float* rotation, const float* opacity, int32_t P,
                       const float* dL_dscales, const float* dL_drotations, const float* dL_dopacities,
                       float* d_scaling, float* d_rotation, float* d_opacity, void* stream);
/* Densification statistics of one view (train.py:266-268, gaussian_model.py:471-473), for Gaussians with
 * radii > 0: max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += |dL_dmeans2D.xy|; denom += 1. */
int gs_densify_stats(const int32_t* radii, const float* dL_dmeans2D /*[P,3]*/, int32_t P,
                     float* max_radii2D /*[P]*/, float* xyz_gradient_accum /*[P]*/, float* denom /*[P]*/,
                     void* stream);

/* ---- the visibility-sparse gradient exchange of the data-parallel step (N > 1; no reference counterpart: the reference trains
 * on one GPU).  Only Gaussians that emitted instances in SOME rank's view have a non-zero gradient row, so the ranks exchange
 * the rows of the union only.  The flat gradient buffer is field-major: field f is a [P, widths[f]] row-major block, the blocks
 * back to back.  pos[i] = (number of j <= i with mask[j] != 0) - 1: the inclusive prefix of the union mask, minus one.
 * gs_rows_pack: packed = the K union rows of every field, field after field ([K, widths[f]] blocks): one launch instead of an
 * index_select per field.  gs_rows_unpack: the inverse, into the rows of the union (the others are not touched). */
int gs_rows_pack(const float* flat, int32_t P, int32_t nfields, const int32_t* widths /*host*/, const uint8_t* mask,
                 const int32_t* pos, int32_t K, float* packed, void* stream);
int gs_rows_unpack(float* flat, int32_t P, int32_t nfields, const int32_t* widths /*host*/, const uint8_t* mask,
                   const int32_t* pos, int32_t K, const float* packed, void* stream);
/* mask[i] = 1 when Gaussian i emitted instances in the forward whose geometry state `scratch` holds (what gs_backward_step's
 * data-parallel form writes into GsStepState.grad_mask later - available as soon as the forward's geometry phase has run,
 * so the union of the ranks' masks can be exchanged while the loss and the backward are still running). */
int gs_export_row_mask(const GsScratch* scratch, int32_t P, uint8_t* mask, void* stream);

/* ---- optional per-stage timing (HIP events recorded on the caller's stream around every kernel
 * group).  Off by default.  bench.py enables it over the timed region to get each kernel's average
 * launch duration on the stream it is launched on.  (No reference counterpart: the reference only
 * brackets whole iterations with a torch.cuda.Event pair, LGDWT-GS/train.py:65-66,97,220.) ---- */
int gs_profile_enable(int32_t on);
/* Restrict the timers to one stage (index as in gs_profile_stage_name; < 0 = all stages again).  Every recorded
 * event pair drains the pipeline for ~10 us, so a throughput measurement keeps only the kernel it reports on. */
int gs_profile_only(int32_t stage);
int gs_profile_reset(void);
int gs_profile_stage_count(void);
const char* gs_profile_stage_name(int32_t stage);
/* Waits for every recorded event, then ms[i] / counts[i] = accumulated milliseconds / launches of stage i. */
int gs_profile_read(double* ms, int64_t* counts, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_H_INCLUDED */
