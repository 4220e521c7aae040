// gs_regionbin.hip - region binning: the instance lists of GsView.tile_cull = 2 in TWO launches.
//
// Replaces, for the culled lists, the chain  depth sort of the Gaussians (4 radix passes x 3 launches) -> instance
// emission (3 launches) -> stable partition by tile id (2 x 3 launches) -> tile ranges  (gs_binning.hip; the
// reference: duplicateWithKeys + cub::DeviceRadixSort::SortPairs + identifyTileRanges, rasterizer_impl.cu:70-138,
// 280-321).  With depth-limited lists that chain moves 2 M instances and 0.2 M depth keys through 23 launches of 3-10 us
// each: it is launch latency, not bytes (profiles/r02_step_timeline.txt: 174 us of kernel time for 47 MB of algorithmic
// traffic).  The order the blend needs is local: WITHIN a tile, by (depth bits, Gaussian index).  So:
//
//   1. preprocess_fwd (gs_preprocess.hip, tile_cull == 2) drops every visible Gaussian into the bucket of each
//      4 x 4-tile REGION (64 x 64 pixels) the bounding box of its alpha >= 1/255 ellipse reaches - and, with depth
//      limits, whose deepest tile bound it is not beyond: one returning atomic per (Gaussian, region) pair, ~2 per
//      Gaussian.  (Per-TILE cursors would need ~10 scattered atomics per Gaussian: 2 M per view at ~20 G/s.)
//   2. region_bin_kernel: ONE workgroup per region loads its bucket, sorts it in LDS by the 64-bit key
//      (depth bits << 32 | index) - bitonic, at most 16 384 entries - evaluates for every entry the SAME per-row ellipse
//      spans gs_binning.hip's duplicate_kernel emits from (tilecull_row_span) and the per-tile depth bound, which gives a
//      16-bit tile mask, counts the sixteen lists, reserves room for them in point_list with ONE atomic on a cursor
//      (the cursor's final value is num_rendered), writes ranges[] for its sixteen tiles and the lists themselves.
//
// Every tile's list holds exactly the Gaussians the LSD path emits for it (with depth limits: the exact per-tile cut, a
// subset of the LSD path's span-trimmed superset that still contains every pair within the bound) in the same order, so
// images, n_contrib and gradients are those of tile_cull = 1.  What differs is the PLACE of a tile's list inside
// point_list (regions reserve their room in completion order): tests/test_gpu_regionbin.py compares list by list.
// The reference's own lists (tile_cull = 0: point_list bit-identical to the reference's) stay on the LSD path.
#include "gs_common.h"
#include "gs_tilecull.h"

#define RB_THREADS 1024
#define RB_WAVES (RB_THREADS / 64)
#define RB_TILES (RG_TILES * RG_TILES)

__global__ void __launch_bounds__(GS_BLOCK) region_prepare_kernel(GeomHeader* hdr, uint32_t* __restrict__ region_count,
                                                                   int regions, uint32_t P) {
  const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (i < regions) region_count[(size_t)i * RG_COUNT_STRIDE] = 0u;
  if (i == 0) {
    hdr->num_rendered = 0u;  // the list cursor of region_bin_kernel
    hdr->overflow = 0u;
    hdr->trunc_failed = 0u;
    hdr->zero = 0u;          // largest region count
    hdr->P = P;
    hdr->sort_n = 0u;
    hdr->n_ordered = 0u;
    hdr->region_mode = 1u;
    hdr->pad[HDR_SIDE_CURSOR] = 0u;
  }
}

__global__ void __launch_bounds__(RB_THREADS) region_bin_kernel(GeomHeader* hdr, const Splat* __restrict__ splat,
                                                                 const uint32_t* __restrict__ region_count,
                                                                 const uint2* __restrict__ region_bucket, uint32_t region_cap,
                                                                 int rg_x, int grid_x, int grid_y,
                                                                 const float* __restrict__ depth_limit,
                                                                 uint2* __restrict__ ranges, uint32_t* __restrict__ point_list,
                                                                 uint32_t capacity, uint32_t n_pad_max) {
  extern __shared__ unsigned long long s_key[];  // [n_pad]: depth bits << 32 | index, later tile mask << 32 | index
  __shared__ uint32_t s_wcnt[RB_WAVES][RB_TILES];  // per wave: entries of each tile in the wave's slice -> exclusive prefix
  __shared__ uint32_t s_off[RB_TILES + 1];         // start of each tile's list inside the region's reservation
  __shared__ uint32_t s_base;
  __shared__ float s_lim[RB_TILES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = blockIdx.x, rx = r % rg_x, ry = r / rg_x;
  const uint32_t cnt_raw = region_count[(size_t)r * RG_COUNT_STRIDE];
  const uint32_t n = min(min(cnt_raw, region_cap), n_pad_max);
  if (tid == 0) {
    atomicMax(&hdr->zero, cnt_raw);
    if (cnt_raw > n) hdr->overflow = 1u;  // the bucket was cut short: every output of this forward is invalid
  }
  if (tid < RB_TILES) {
    const int ty = ry * RG_TILES + (tid >> 2), tx = rx * RG_TILES + (tid & 3);
    s_lim[tid] = (depth_limit && ty < grid_y && tx < grid_x) ? depth_limit[ty * grid_x + tx] : __builtin_inff();
  }
  uint32_t n_pad = 64;
  while (n_pad < n) n_pad <<= 1;
  const uint2* bucket = region_bucket + (size_t)r * region_cap;
  for (uint32_t i = tid; i < n_pad; i += RB_THREADS) {
    unsigned long long k = ~0ull;
    if (i < n) {
      const uint2 e = bucket[i];
      k = ((unsigned long long)e.x << 32) | (unsigned long long)e.y;
    }
    s_key[i] = k;
  }
  __syncthreads();
  // ---- bitonic sort, ascending (keys are unique: an index appears once per region).
  // Pair i of a step with distance j <= 64 lies inside the 128-element chunk i / 64 - and the 64 pairs of a chunk are one wave's
  // (i = tid + t * RB_THREADS: chunk = wave + 16 t).  So the steps with j <= 64 of a stage run back to back inside each wave
  // (LDS operations of a wave execute in order; the compiler is told not to move them) and only the steps with j >= 128
  // need the workgroup barrier: 14 barriers instead of 66 for 2 048 entries, 27 instead of 91 for 8 192.
  auto step = [&](uint32_t k, uint32_t j) {
    for (uint32_t i = tid; i < (n_pad >> 1); i += RB_THREADS) {
      const uint32_t lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u));
      const uint32_t hi = lo | j;
      const unsigned long long a = s_key[lo], b = s_key[hi];
      const bool up = (lo & k) == 0u;
      if ((a > b) == up) {
        s_key[lo] = b;
        s_key[hi] = a;
      }
    }
  };
  for (uint32_t k = 2; k <= n_pad; k <<= 1) {
    uint32_t j = k >> 1;
    for (; j > 64; j >>= 1) {
      step(k, j);
      __syncthreads();
    }
    for (; j > 0; j >>= 1) {   // inside the wave's own chunks
      step(k, j);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (k >= 128u || k == n_pad) __syncthreads();   // (the next stage's first steps - or the pass below - read other waves' chunks)
  }
  // ---- tile mask of every entry: bit 4 k + c = tile (4 ry + k, 4 rx + c)
  for (uint32_t e = tid; e < n; e += RB_THREADS) {
    const uint32_t id = (uint32_t)s_key[e];
    const float4* rec = reinterpret_cast<const float4*>(&splat[id]);
    const float4 ra = rec[0], rc = rec[1];
    const uint4 tail = reinterpret_cast<const uint4*>(rec)[3];  // rect_min, rect_max, -, -
    const TileCull tc = tilecull_setup(1, ra.x, ra.y, rc.x, rc.y, rc.z, rc.w);
    const uint32_t rminx = tail.x & 0xFFFFu, rminy = tail.x >> 16, rmaxx = tail.y & 0xFFFFu, rmaxy = tail.y >> 16;
    uint32_t mask = 0;
#pragma unroll
    for (uint32_t k = 0; k < RG_TILES; k++) {
      const uint32_t ty = (uint32_t)ry * RG_TILES + k;
      if (ty < rminy || ty >= rmaxy) continue;
      uint32_t tx0;
      const uint32_t nt = tilecull_row_span(tc, ty, rminx, rmaxx, tx0);
      if (nt == 0) continue;
      // columns [tx0, tx0 + nt) cut to the region's four
      const int c0 = max((int)tx0 - rx * RG_TILES, 0), c1 = min((int)(tx0 + nt) - rx * RG_TILES, RG_TILES);
      if (c1 > c0) mask |= (((1u << (c1 - c0)) - 1u) << c0) << (4u * k);
    }
    if (depth_limit) {  // the per-tile rule of gs_tilecull.h: a pair beyond its tile's bound is not emitted
#pragma unroll
      for (uint32_t t = 0; t < RB_TILES; t++)
        if (depth_beyond_limit(ra.z, s_lim[t])) mask &= ~(1u << t);
    }
    s_key[e] = ((unsigned long long)mask << 32) | (unsigned long long)id;
  }
  __syncthreads();
  // ---- counts: wave w owns the contiguous slice [w * per_wave, (w + 1) * per_wave) of the sorted entries
  const uint32_t per_wave = ((n + RB_WAVES * 64 - 1) / (RB_WAVES * 64)) * 64;
  const uint32_t w_lo = min((uint32_t)wid * per_wave, n), w_hi = min(w_lo + per_wave, n);
  uint32_t my_cnt = 0;  // lane t < 16: this wave's count of tile t
  for (uint32_t e0 = w_lo; e0 < w_hi; e0 += 64) {
    const uint32_t e = e0 + lane;
    const uint32_t m = e < w_hi ? (uint32_t)(s_key[e] >> 32) : 0u;
#pragma unroll
    for (uint32_t t = 0; t < RB_TILES; t++) {
      const uint32_t c = (uint32_t)__popcll(__ballot((m >> t) & 1u));
      if (lane == (int)t) my_cnt += c;
    }
  }
  if (lane < RB_TILES) s_wcnt[wid][lane] = my_cnt;
  __syncthreads();
  if (tid < RB_TILES) {  // exclusive prefix over the waves, per tile
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < RB_WAVES; w++) {
      const uint32_t c = s_wcnt[w][tid];
      s_wcnt[w][tid] = run;
      run += c;
    }
    s_off[tid + 1] = run;  // tile totals for now
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    s_off[0] = 0;
    for (int t = 0; t < RB_TILES; t++) {
      const uint32_t c = s_off[t + 1];
      s_off[t] = run;
      run += c;
    }
    s_off[RB_TILES] = run;
    uint32_t base = 0xFFFFFFFFu;
    if (run) {
      const uint32_t at = atomicAdd(&hdr->num_rendered, run);
      if ((unsigned long long)at + run <= (unsigned long long)capacity) base = at;
      else hdr->overflow = 1u;
    } else {
      base = 0u;
    }
    s_base = base;
  }
  __syncthreads();
  const uint32_t base = s_base;
  if (tid < RB_TILES) {
    const int ty = ry * RG_TILES + (tid >> 2), tx = rx * RG_TILES + (tid & 3);
    if (ty < grid_y && tx < grid_x) {
      const uint32_t b = base + s_off[tid], c = s_off[tid + 1] - s_off[tid];
      ranges[ty * grid_x + tx] = (base == 0xFFFFFFFFu || c == 0u) ? make_uint2(0u, 0u) : make_uint2(b, b + c);
    }
  }
  if (base == 0xFFFFFFFFu) return;  // no room in point_list: nothing is written (the caller repeats the forward)
  // ---- the lists: position = reservation + tile start + entries of the tile in earlier waves + ... earlier in this wave
  uint32_t run = 0;  // lane t < 16: entries of tile t this wave has written so far
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (uint32_t e0 = w_lo; e0 < w_hi; e0 += 64) {
    const uint32_t e = e0 + lane;
    const unsigned long long k = e < w_hi ? s_key[e] : 0ull;
    const uint32_t m = (uint32_t)(k >> 32), id = (uint32_t)k;
#pragma unroll
    for (uint32_t t = 0; t < RB_TILES; t++) {
      const unsigned long long b = __ballot((m >> t) & 1u);
      const uint32_t before = (uint32_t)__shfl((int)run, (int)t, 64);
      if ((m >> t) & 1u) point_list[base + s_off[t] + s_wcnt[wid][t] + before + (uint32_t)__popcll(b & lt_mask)] = id;
      if (lane == (int)t) run += (uint32_t)__popcll(b);
    }
  }
}

// parity export: the reference's 64-bit sorted key (tile << 32 | depth bits) of every list entry, tile by tile
__global__ void __launch_bounds__(GS_BLOCK) export_keys_region_kernel(const uint2* __restrict__ ranges,
                                                                      const uint32_t* __restrict__ point_list,
                                                                      const Splat* __restrict__ splat,
                                                                      uint64_t* __restrict__ keys_sorted) {
  const uint2 r = ranges[blockIdx.x];
  for (uint32_t i = r.x + threadIdx.x; i < r.y; i += GS_BLOCK)
    keys_sorted[i] = ((uint64_t)blockIdx.x << 32) | (uint64_t)__float_as_uint(splat[point_list[i]].depth);
}

int launch_region_prepare(const GeomView& g, uint32_t* region_count, int regions, uint32_t P, hipStream_t s) {
  hipLaunchKernelGGL(region_prepare_kernel, dim3((regions + GS_BLOCK - 1) / GS_BLOCK > 0 ? (regions + GS_BLOCK - 1) / GS_BLOCK : 1),
                     dim3(GS_BLOCK), 0, s, g.hdr, region_count, regions, P);
  return 0;
}

int launch_region_bin(const GeomView& g, const uint32_t* region_count, const uint2* region_bucket, uint32_t region_cap, int rg_x,
                      int rg_y, int grid_x, int grid_y, const float* tile_depth_limit, uint2* ranges, uint32_t* point_list,
                      int64_t capacity, hipStream_t s) {
  // LDS for the largest bucket a region can hold, as a power of two of 64-bit keys (8 KB .. 128 KB)
  uint32_t n_pad = 1024;
  while (n_pad < region_cap && n_pad < RG_MAX_ENTRIES) n_pad <<= 1;
  const size_t lds = (size_t)n_pad * sizeof(unsigned long long);
  // the attribute is per DEVICE (and sticky): remember what each device of this process was raised to
  static_assert(RG_MAX_ENTRIES * sizeof(unsigned long long) <= 160 * 1024, "a region's sort must fit gfx950's 160 KB of LDS");
  static size_t lds_allowed[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev < 0 || lds > lds_allowed[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)region_bin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) lds_allowed[dev] = lds;
  }
  const uint32_t cap32 = capacity > 0xFFFFFFFFll ? 0xFFFFFFFFu : (capacity < 0 ? 0u : (uint32_t)capacity);
  hipLaunchKernelGGL(region_bin_kernel, dim3(rg_x * rg_y), dim3(RB_THREADS), lds, s, g.hdr, g.splat, region_count, region_bucket,
                     region_cap, rg_x, grid_x, grid_y, tile_depth_limit, ranges, point_list, cap32, n_pad);
  const hipError_t le = hipGetLastError();  // (a refused launch - too much dynamic LDS - must not pass for stale lists)
  return le == hipSuccess ? 0 : (int)le;
}

int launch_export_keys_region(const uint2* ranges, const uint32_t* point_list, const Splat* splat, int T, uint64_t* keys_sorted,
                              hipStream_t s) {
  hipLaunchKernelGGL(export_keys_region_kernel, dim3(T), dim3(GS_BLOCK), 0, s, ranges, point_list, splat, keys_sorted);
  return 0;
}
