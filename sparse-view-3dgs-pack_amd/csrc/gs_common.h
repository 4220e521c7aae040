// gs_common.h - shared host/device definitions of libgsplat_hip (gfx950 only).
//
// HBM layout of the private fwd<->bwd state (the three byte buffers of include/gsplat.h):
//
//  geom   : GeomHeader (256 B)
//           splat[P]        64-byte record per Gaussian (one cache line per gather in the blend kernels)
//           cov3D[P][6]     fp32, only read again by the backward per-Gaussian kernel
//           tiles_touched[P] u32
//           block_sums / sorted_sums [ceil(P/256)+1] u32   workgroup sums of tiles_touched (index / depth order)
//           gsort            radix-sort buffers of the per-Gaussian depth sort (16 B x P + histograms)
//  img    : final_T[N] f32 | n_contrib[N] u32 | ranges[T] uint2 | tile_work[T] u32 | tile_order[T] u32 | region_count[<= T] u32
//  binning: keys[2][cap] u32 | Gaussian ids[2][cap] u32 | radix histograms   (16 B per instance of capacity) | region / chunk /
//           tile tables (gs_tilebin.hip).  tile_cull = 0 / 1: the arrays hold the region ENTRIES (key = region id | tile mask << 16)
//           while they are partitioned, then ids[0] receives point_list
//           region binning (GsView.tile_cull = 2): the key halves hold the region buckets [regions][cap / regions] of
//           (depth bits, Gaussian index), Gaussian ids[0] is point_list
//
// All sub-arrays start on 256-byte boundaries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsplat.h"

#define GS_BLOCK 256
#define TILE_X GS_TILE_X
#define TILE_Y GS_TILE_Y

// ---------------------------------------------------------------------------------------------
struct GeomHeader {
  uint32_t num_rendered;  // R = sum tiles_touched
  uint32_t overflow;      // set by the duplicate kernel when R > binning capacity
  uint32_t trunc_failed;  // depth-limited emission (gs_tilecull.h): a bounded tile ran out of list entries before saturating
  uint32_t zero;          // 0; region binning: the largest region_count (the first 16 bytes are what gs_forward_status copies out)
  uint32_t P;
  uint32_t sort_n;        // instances the binning stage really processes: overflow ? 0 : num_rendered
  uint32_t n_ordered;     // Gaussians in the depth order = those that emit instances (the sort's first pass drops the rest)
  uint32_t region_mode;   // 1: the lists of this view were built by region binning (gs_regionbin.hip), 0: by the LSD path
  uint32_t step_tag;      // word 8: GsScratch.step_tag as gs_forward_status found it
  uint32_t pad[55];       // pad[0] = word 9: GS_STATUS_CHECK over (step_tag, num_rendered, overflow, trunc_failed)
                          // pad[HDR_SIDE_CURSOR]: next Gaussian block of step_uninstanced_kernel (zeroed with the header)
};
#define HDR_SIDE_CURSOR 1
static_assert(sizeof(GeomHeader) == 256, "header is one 256-B block");

// 64-byte per-Gaussian record written by the forward preprocess kernel.
struct __attribute__((aligned(64))) Splat {
  float x, y;        // pixel-space mean (ndc2Pix)
  float depth;       // view-space z
  float invdepth;    // 1 / depth (what the blend accumulates, forward.cu:375)
  float cxx, cxy, cyy;  // conic
  float opacity;     // opacity * antialiasing scaling
  float r, g, b;     // SH colour (+0.5, clamped) or colors_precomp
  float extra;       // optional 4th blended channel (GsGaussians.extra_channel, e.g. NIR albedo), else 0
  uint32_t rect_min; // x | y << 16
  uint32_t rect_max;
  uint32_t tiles;    // tiles_touched
  uint32_t clamped;  // bit c set = channel c was clamped to 0
};
static_assert(sizeof(Splat) == 64, "Splat is one 64-B line");

static inline __host__ __device__ uint32_t gs_status_check(uint32_t tag, uint32_t nr, uint32_t ov, uint32_t tf) {
  return GS_STATUS_CHECK(tag, nr, ov, tf);
}
static inline __host__ __device__ size_t gs_align(size_t x) { return (x + 255) & ~(size_t)255; }

// radix sort geometry: 8-bit digits, 256 threads x 16 keys per workgroup
#define RS_BITS 8
#define RS_RADIX 256
#define RS_RADIX_MAX 512  // the one-pass partition of up to 512 regions uses 9-bit digits (tables are sized for it)
#define RS_ITEMS 16
#define RS_TILE (GS_BLOCK * RS_ITEMS)

// ping-pong buffers + histogram workspace of one (u32 key, u32 value) radix sort of up to `cap` pairs
struct SortBufs {
  uint32_t* keys[2];
  uint32_t* vals[2];
  uint32_t* hist;      // [RS_RADIX_MAX][nblk] digit-major workgroup histograms
  uint32_t* scan_tmp;  // [RS_RADIX_MAX] digit totals of the current pass (rs_rowscan_kernel)
};
static inline __host__ __device__ size_t sort_bytes(size_t cap) {
  if (cap == 0) cap = 1;
  size_t nblk = (cap + RS_TILE - 1) / RS_TILE;
  size_t hist_n = nblk * RS_RADIX_MAX;
  return 4 * gs_align(4 * cap) + gs_align(4 * hist_n) + gs_align(4 * RS_RADIX_MAX);
}
static inline __host__ __device__ SortBufs sort_view(void* buf, size_t cap) {
  if (cap == 0) cap = 1;
  char* p = (char*)buf;
  SortBufs b;
  size_t nblk = (cap + RS_TILE - 1) / RS_TILE;
  size_t hist_n = nblk * RS_RADIX_MAX;
  b.keys[0] = (uint32_t*)p; p += gs_align(4 * cap);
  b.keys[1] = (uint32_t*)p; p += gs_align(4 * cap);
  b.vals[0] = (uint32_t*)p; p += gs_align(4 * cap);
  b.vals[1] = (uint32_t*)p; p += gs_align(4 * cap);
  b.hist = (uint32_t*)p; p += gs_align(4 * hist_n);
  b.scan_tmp = (uint32_t*)p; p += gs_align(4 * RS_RADIX_MAX);
  return b;
}

struct GeomView {
  GeomHeader* hdr;
  Splat* splat;
  float* cov3D;
  uint32_t* tiles_touched;
  uint32_t* block_sums;   // [nb+1] per-workgroup sums of tiles_touched (index order) -> exclusive offsets
  uint32_t* sorted_sums;  // [nb+1] the same in depth order (instance emission)
  uint32_t* depth_keys;   // [P] depth bits (0xFFFFFFFF when culled): read-only input of the depth sort
  SortBufs gsort;         // ping-pong buffers of the depth sort of the P Gaussians
};
static inline __host__ __device__ size_t geom_bytes(size_t P) {
  size_t nb = (P + GS_BLOCK - 1) / GS_BLOCK;
  return sizeof(GeomHeader) + gs_align(64 * P) + gs_align(24 * P) + 2 * gs_align(4 * P) + 2 * gs_align(4 * (nb + 1)) +
         sort_bytes(P);
}
static inline __host__ __device__ GeomView geom_view(void* buf, size_t P) {
  char* p = (char*)buf;
  size_t nb = (P + GS_BLOCK - 1) / GS_BLOCK;
  GeomView g;
  g.hdr = (GeomHeader*)p; p += sizeof(GeomHeader);
  g.splat = (Splat*)p; p += gs_align(64 * P);
  g.cov3D = (float*)p; p += gs_align(24 * P);
  g.tiles_touched = (uint32_t*)p; p += gs_align(4 * P);
  g.block_sums = (uint32_t*)p; p += gs_align(4 * (nb + 1));
  g.sorted_sums = (uint32_t*)p; p += gs_align(4 * (nb + 1));
  g.depth_keys = (uint32_t*)p; p += gs_align(4 * P);
  g.gsort = sort_view(p, P);
  return g;
}

#define RG_COUNT_STRIDE 32  // u32 words between two region counters (one 128-B line per counter)
struct ImgView {
  float* final_T;
  uint32_t* n_contrib;
  uint2* ranges;
  uint32_t* tile_work;   // [T] list entries the backward blend will visit in this tile = max last contributor (forward)
  uint32_t* tile_order;  // [8 ceil(T/8)] launch order of the backward blend: per XCD band, by decreasing tile_work
  float* tile_stop_depth;  // [T] view depth of the list entry at which the tile's last pixel saturated (+inf: never)
  uint32_t* region_count;  // [regions <= T][RG_COUNT_STRIDE] region binning: Gaussians each 4 x 4-tile region received
                           // (gs_regionbin.hip; word 0 of a 128-B line each - sixteen hot counters on one line serialise)
};
static inline __host__ __device__ size_t img_bytes(size_t N, size_t T) {
  return gs_align(4 * N) + gs_align(4 * N) + gs_align(8 * T) + gs_align(4 * T) + gs_align(4 * (T + 8)) + gs_align(4 * T) +
         gs_align(4 * RG_COUNT_STRIDE * T);
}
static inline __host__ __device__ ImgView img_view(void* buf, size_t N, size_t T) {
  char* p = (char*)buf;
  ImgView v;
  v.final_T = (float*)p; p += gs_align(4 * N);
  v.n_contrib = (uint32_t*)p; p += gs_align(4 * N);
  v.ranges = (uint2*)p; p += gs_align(8 * T);
  v.tile_work = (uint32_t*)p; p += gs_align(4 * T);
  v.tile_order = (uint32_t*)p; p += gs_align(4 * (T + 8));
  v.tile_stop_depth = (float*)p; p += gs_align(4 * T);
  v.region_count = (uint32_t*)p;
  return v;
}

// Region entries of a Gaussian (two-level binning, gs_tilebin.hip), counted by the preprocess kernel where that is cheap and left in
// bits 16-31 of Splat.clamped: 15 bits of count, bit 15 = "row-wise enumeration" (gs_tilebin.hip); all ones = not counted
#define TB_ENTRIES_UNKNOWN 0xFFFFu
#define TB_ENTRIES_MAX 0x7FFEu
static inline __host__ __device__ uint32_t tb_rect_entries(uint32_t minx, uint32_t miny, uint32_t maxx, uint32_t maxy) {
  // the reference's rectangle [min, max) in tiles -> 4 x 4-tile regions it reaches (every one of them holds a tile of it)
  const uint32_t n = ((maxx - 1u) / 4u - minx / 4u + 1u) * ((maxy - 1u) / 4u - miny / 4u + 1u);
  return n > TB_ENTRIES_MAX ? TB_ENTRIES_UNKNOWN : n;
}
// ---- two-level binning of tile_cull = 0 / 1 (gs_tilebin.hip): what lies behind the SortBufs in the binning buffer ----
#define TB_CHUNK 256   // entries of one region a WAVE counts / writes at a time (four rounds of 64)
struct TileBinView {
  uint2* region_ranges;    // [regions] range of each region's entries in the partitioned entry arrays
  uint32_t* chunk_first;   // [regions + 1] exclusive prefix of the regions' chunk counts
  uint4* chunk_desc;       // [max_chunks] (first entry, end of the chunk's entries, region, -): one load tells a workgroup its work
  uint32_t* chunk_counts;  // [max_chunks][16] entries of each of the region's tiles in the chunk -> exclusive prefix over the region's chunks
  uint32_t* tile_start;    // [T + 1] start of each tile's list in point_list (tile order)
  uint32_t max_chunks;
};
// sum over regions of ceil(entries / TB_CHUNK) <= entries / TB_CHUNK + regions; entries <= capacity, regions <= T
static inline __host__ __device__ size_t tilebin_max_chunks(size_t cap, size_t T) { return cap / TB_CHUNK + T + 1; }
static inline __host__ __device__ size_t tilebin_bytes(size_t cap, size_t T) {
  const size_t mc = tilebin_max_chunks(cap, T);
  return gs_align(8 * T) + gs_align(4 * (T + 1)) + gs_align(16 * mc) + gs_align(64 * mc) + gs_align(4 * (T + 1));
}
static inline __host__ __device__ TileBinView tilebin_view(void* buf, size_t cap, size_t T) {
  char* p = (char*)buf;
  TileBinView v;
  const size_t mc = tilebin_max_chunks(cap, T);
  v.region_ranges = (uint2*)p; p += gs_align(8 * T);
  v.chunk_first = (uint32_t*)p; p += gs_align(4 * (T + 1));
  v.chunk_desc = (uint4*)p; p += gs_align(16 * mc);
  v.chunk_counts = (uint32_t*)p; p += gs_align(64 * mc);
  v.tile_start = (uint32_t*)p;
  v.max_chunks = (uint32_t)mc;
  return v;
}
// binning buffer = one SortBufs sized for the instances (region entries: key = region id | tile mask << 16, value = Gaussian
// index; ids[0] ends up as point_list) + the region / chunk / tile tables
static inline __host__ __device__ size_t bin_bytes(size_t cap, size_t T) { return sort_bytes(cap) + tilebin_bytes(cap, T); }

// per-Gaussian gradient row accumulated by the backward blend: FLOAT64 slots (round 4), one 128-B line per Gaussian.
// A tile's totals are fp32 (fixed lane order, fixed reduction tree: deterministic); what is run-dependent is the order in
// which the tiles' totals arrive at a Gaussian's row.  In a 53-bit accumulator a sum of 24-bit terms is exact - hence the
// same whatever the order - unless the terms span more than ~2^20 in magnitude, and then differs in the 16th digit; fp32
// rows differed in the 7th, which the conic -> covariance chain amplifies a thousandfold (gs_backward_math.h).
enum { GR_MX = 0, GR_MY, GR_CXX, GR_CXY, GR_CYY, GR_OP, GR_CR, GR_CG, GR_CB, GR_ID, GR_EXTRA, GR_N, GR_STRIDE = 16 };
typedef double gs_row_t;
#define GR_ROW_BYTES (GR_STRIDE * sizeof(gs_row_t))
// per-Gaussian RECORD the chain kernel leaves for the streaming kernels (gs_backward_math.h): fp32, five float4.
// mean2D gradient | covariance part of the mean gradient | opacity gradient (after the anti-aliasing factor) | seven slots:
// dL_dscale[3] + dL_dquaternion[4], or dL_dcov3D[6] when the covariance was given | colour gradient | 4th channel
enum { GC_MX = 0, GC_MY, GC_DMX, GC_DMY, GC_DMZ, GC_DOP, GC_C0, GC_R = GC_C0 + 7, GC_G, GC_B, GC_EXTRA, GC_STRIDE = 20 };
#define GC_REC_BYTES (GC_STRIDE * sizeof(float))
// backward workspace = rows [P][GR_STRIDE] float64, then records [P][GC_STRIDE] fp32
// ... then the per-wave partial sums of dL/dgain (fused multispectral step)
static inline __host__ __device__ size_t bwd_workspace_bytes(size_t P) {
  return gs_align(P * GR_ROW_BYTES) + gs_align(P * GC_REC_BYTES) + gs_align(4 * ((P + GS_BLOCK - 1) / GS_BLOCK + 4));
}

// rasterizer_impl.cu:35-50 (host)
static inline uint32_t gs_higher_msb(uint32_t n) {
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

#define GS_HIP_CHECK(expr)                 \
  do {                                     \
    hipError_t _e = (expr);                \
    if (_e != hipSuccess) return (int)_e;  \
  } while (0)

// after a launch: surface launch errors; in debug mode also synchronise (auxiliary.h:178-185)
static inline int gs_after_launch(hipStream_t s, int debug) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (debug) {
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
#define GS_LAUNCH_CHECK(stream, debug)             \
  do {                                             \
    int _rc = gs_after_launch((stream), (debug));  \
    if (_rc) return _rc;                           \
  } while (0)

// ---- launchers implemented in the kernel files (host functions, C++ linkage) ----
struct PreprocessArgs {
  int P, D, M;
  const float* means3D;
  const float* scales;
  float scale_modifier;
  const float* rotations;
  const float* opacities;
  const float* shs;
  const float* shs_rest;  // GsGaussians.shs_rest: split rows (shs = [P,1,3], shs_rest = [P,M-1,3]) or NULL
  const float* cov3D_precomp;
  const float* colors_precomp;
  const float* viewmatrix;
  const float* projmatrix;
  const float* campos;
  int W, H;
  float focal_x, focal_y, tan_fovx, tan_fovy;
  int* radii;
  int grid_x, grid_y;
  int antialiasing;
  const float* extra_channel;  // [P] or NULL
  const float* extra_gain;     // GsGaussians.extra_gain
  int tile_cull;  // GsView.tile_cull: 1 = emit only tiles the alpha >= 1/255 ellipse can reach (gs_tilecull.h); 2 = region binning
  const float* tile_depth_limit;  // [T] or NULL: depth-limited emission (gs_tilecull.h); only with tile_cull
  int raw_activations;            // GsGaussians.raw_activations
  // region binning (tile_cull == 2, gs_regionbin.hip): every Gaussian is dropped into the bucket of each 4 x 4-tile region
  // its alpha >= 1/255 ellipse may reach
  uint32_t* region_count;         // [rg_x * rg_y][RG_COUNT_STRIDE], zeroed by launch_region_prepare
  uint2* region_bucket;           // [rg_x * rg_y][region_cap]: (depth bits, Gaussian index)
  uint32_t region_cap;
  int rg_x, rg_y;
};
int launch_preprocess_fwd(const PreprocessArgs& a, const GeomView& g, hipStream_t s);
int launch_scan_block_sums(const GeomView& g, int P, hipStream_t s);
int launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);

int launch_bin_prepare(const GeomView& g, int64_t capacity, uint2* ranges, int T, hipStream_t s);
// first_keys != NULL: the first pass reads its keys from there (left untouched) and takes value = index
int launch_radix_sort(const SortBufs& b, const uint32_t* n_dev, int64_t n_host_bound, int end_bit, int start_buf,
                      const uint32_t* first_keys, hipStream_t s, int debug, uint32_t* n_kept = nullptr, int digit_bits = 8);
int launch_scan_sums(uint32_t* sums, int nb, hipStream_t s);  // in-place exclusive scan of sums[nb], total -> sums[nb]
// the lists of tile_cull = 0 / 1 from the depth order (g.gsort.vals[0]): gs_tilebin.hip
int launch_tile_binning(const GeomView& g, const SortBufs& bv, const TileBinView& tb, int P, int64_t capacity, int grid_x, int grid_y,
                        int tile_cull, const float* tile_depth_limit, uint2* ranges, int force_rowwise, hipStream_t s, int debug);

// ---- region binning (gs_regionbin.hip) ----
#define RG_TILES 4            // a region is RG_TILES x RG_TILES tiles (64 x 64 pixels)
#define RG_MAX_ENTRIES 16384  // Gaussians one region's workgroup can sort in LDS (128 KB of 64-bit keys)
static inline __host__ __device__ uint32_t region_capacity(int64_t capacity, int regions) {
  const int64_t c = regions > 0 ? capacity / regions : 0;
  return (uint32_t)(c > RG_MAX_ENTRIES ? RG_MAX_ENTRIES : (c < 0 ? 0 : c));
}
int launch_region_prepare(const GeomView& g, uint32_t* region_count, int regions, uint32_t P, hipStream_t s);
int launch_region_bin(const GeomView& g, const uint32_t* region_count, const uint2* region_bucket, uint32_t region_cap, int rg_x,
                      int rg_y, int grid_x, int grid_y, const float* tile_depth_limit, uint2* ranges, uint32_t* point_list,
                      int64_t capacity, hipStream_t s);
int launch_export_keys_region(const uint2* ranges, const uint32_t* point_list, const Splat* splat, int T, uint64_t* keys_sorted,
                              hipStream_t s);


int launch_render_fwd_wave(const uint2* ranges, const uint32_t* point_list, int W, int H, int grid_x, int grid_y,
                           const Splat* splat, const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_work,
                           const uint32_t* order_hint, const float* depth_limit, float* stop_depth, uint32_t* trunc_failed,
                           float* out_color, float* out_invdepth, float* out_extra, int fsgs, int cull, hipStream_t s);
// tile_order[] = per XCD band of the image, the tiles by decreasing tile_work[] (order inside a bucket of equal work is free)
int launch_export_stop_depth(const float* stop_depth, float* out, int grid_x, int grid_y, hipStream_t s);
int launch_tile_order(const uint32_t* tile_work, uint32_t* tile_order, int T, uint32_t* order_out, const float* stop_depth,
                      float* limit_out, int grid_x, int grid_y, const GeomHeader* hdr, hipStream_t s,
                      const float* slack_dev = nullptr, uint32_t* status_host = nullptr, const uint32_t* step_tag = nullptr);
int launch_zero_rows(gs_row_t* rows, size_t P, const uint32_t* tiles_touched, hipStream_t s);
int launch_render_bwd_wave(const uint2* ranges, const uint32_t* point_list, int W, int H, int grid_x, int grid_y,
                           const Splat* splat, const float* bg, const float* final_T, const uint32_t* n_contrib,
                           const uint32_t* tile_work, const uint32_t* tile_order, const float* dL_dpix,
                           const float* dL_dinvdepth, const float* dL_dextra, gs_row_t* grad_rows, int fsgs, hipStream_t s);

int launch_blend_stats(const uint2* ranges, const uint32_t* point_list, int W, int H, int grid_x, int grid_y, const Splat* splat,
                       const uint32_t* n_contrib, const uint32_t* tile_work, unsigned long long* out, hipStream_t s);

struct PreprocessBwdArgs {
  int P, D, M;
  const float* means3D;
  const int* radii;
  const float* shs;
  const float* shs_rest;  // GsGaussians.shs_rest: split rows (shs = [P,1,3], shs_rest = [P,M-1,3]) or NULL
  const float* scales;
  const float* rotations;
  const float* opacities;
  const float* colors_precomp;
  float scale_modifier;
  const float* cov3D;  // precomputed or geom.cov3D
  const float* viewmatrix;
  const float* projmatrix;
  const float* campos;
  float focal_x, focal_y, tan_fovx, tan_fovy;
  int antialiasing;
  int has_invdepth;  // 0: none, 1: inverse-depth image gradient (dr_aa), 2: depth image gradient (FSGS generation)
  int has_extra;     // a 4th channel was blended: slot GC_EXTRA of the records carries its per-Gaussian gradient
  // fused multispectral step (GsStepState.extra): the channel is sigmoid(extra_raw[i]) * clamp(*extra_gain, 0.1, 10).  The chain
  // kernel then stores dL/dextra_i * clamp(gain) in GC_EXTRA (what the raw row's activation backward needs) and wave w's share
  // of dL/dgain = sum_i dL/dextra_i sigmoid(raw_i) [gain inside the clamp] in gain_partials[w] - so that the streaming
  // kernel never reads the gain and its first workgroup can step it from the partials, added in index order
  const float* extra_raw;
  const float* extra_gain;
  float* gain_partials;  // [ceil(P / 256)]
  int raw_activations;  // GsGaussians.raw_activations
  int skip_uninstanced;  // rows come from the blend backward of THIS forward: a Gaussian that emitted no instance (culled spans,
                         // depth limits) has all-zero sums and so all-zero gradients - its geometry / SH backward is skipped
  const gs_row_t* grad_rows;  // [P][GR_STRIDE] float64: read by the chain kernel only
  float* grad_recs;           // [P][GC_STRIDE]: written by the chain kernel, read by the streaming kernels
  int clean_rows;          // gs_backward_step (GsStepState.rows_clean): the chain kernel zeroes every row it has consumed
  const uint32_t* tiles_touched;  // [P] instances each Gaussian emitted in this forward (skip_uninstanced)
  const Splat* splat;
  GsGrads out;
};
// the float64 covariance chain of every Gaussian with sums: rows -> records (hdr != NULL: a forward that flagged overflow /
// trunc_failed leaves nothing to compute - only the rows are cleaned)
int launch_chain(const PreprocessBwdArgs& a, const GeomHeader* hdr, hipStream_t s);
int launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s);
// the same stage with the train step's tail fused in (gs_backward_step); bias corrections precomputed on the host
struct StepArgs {
  GsStepState st;
  float lr_bc1[6];        // lr / (1 - beta1^t) per learning-rate class   (st.coef_dev, when given, replaces both arrays)
  float inv_sqrt_bc2[5];  // 1 / sqrt(1 - beta2^t) per row
  float x_lr_bc1[2], x_inv_sqrt_bc2[2];  // the same for the 4th channel's raw row [0] and its global gain [1]
  const float* gain_partials;  // [workgroups] written by the chain kernel: per-wave sums of dL/dgain
  const GeomHeader* hdr;  // overflow (binning capacity exceeded, nothing blended) or trunc_failed set -> the step is a no-op
  int phase;              // 0 every Gaussian | 1 only those without instances | 2 only those with (gs_step_uninstanced)
  int phase1_workgroups;  // grid of the throttled phase-1 kernel (0 = default)
};
int launch_preprocess_bwd_step(const PreprocessBwdArgs& a, const StepArgs& sa, hipStream_t s);
