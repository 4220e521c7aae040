// gs_tilebin.hip - the instance lists of GsView.tile_cull = 0 / 1 (the reference's bounding-square lists and the culled lists
// when they are too long for region binning): two-level binning, regions first.
//
// Replaces duplicateWithKeys (rasterizer_impl.cu:70-111), cub::DeviceRadixSort::SortPairs over R 64-bit keys (:306-311) and
// identifyTileRanges (:116-138), and produces EXACTLY the reference's point_list and ranges (tile after tile; inside a tile by
// depth bits, then Gaussian index).
//
// Rounds 1-4 emitted one (tile id, Gaussian) pair per instance in depth order and ran a stable two-pass LSD partition over the
// R = 25 M pairs of BASELINE C3 (6 launches, 2 x 20 B x R moved through kernels that are bound by their own dependent chain:
// 0.51 ms with emission and ranges, 0.22 of the HBM roofline).  The order the blend needs is local - WITHIN a tile - and a
// Gaussian covers its tiles in blocks, so the unit that has to travel through a partition is not the instance but the
// (Gaussian, 4 x 4-tile REGION) pair with a 16-bit tile mask: ~6 instances each at C3.
//
//   1. depth order of the P Gaussians (gs_binning.hip: 4 LSD passes over 8 B x P, Gaussians without instances dropped);
//   2. tb_entries_kernel + scan + tb_emit_kernel: in depth order, every Gaussian emits one 8-byte ENTRY per region its tiles
//      reach: key = region id | tile mask << 16, value = Gaussian index (bit 4 k + c of the mask = tile (4 ry + k, 4 rx + c));
//   3. a stable partition of the entries by region id (rs_* of gs_binning.hip; 1-2 passes over 8 B x entries, 4-6 x fewer
//      than instances) leaves every region's entries contiguous and in (depth, index) order;
//   4. tb_regions_kernel: region ranges by binary search, regions cut into chunks of TB_CHUNK entries;
//   5. tb_tile_count_kernel (per chunk: entries of each of its 16 tiles), tb_region_scan_kernel (per region: exclusive prefix
//      over its chunks), tb_tile_scan_kernel (all tiles in tile order: ranges[] - no pass over the instances finds them);
//   6. tb_write_kernel: every chunk writes its entries' indices into the sixteen lists - 4 B x R, the only instance-sized
//      traffic of the whole stage.
//
// Capacity: every entry holds at least one tile (see tb_entries_kernel), so entries <= instances <= binning capacity and the
// entries fit the four capacity-sized arrays the caller provides anyway.
#include "gs_common.h"
#include "gs_prof.h"
#include "gs_tilecull.h"

#define TB_ROWWISE 0x80000000u   // bit 31 of a depth-order slot: this Gaussian enumerates its entries tile row by tile row

// what the enumeration needs of one Gaussian
struct TbOwner {
  uint32_t rmin, rmax;  // rect in tiles: x | y << 16
  float depth;
  int dq;               // depth-limit verdict of the preprocess kernel (gs_tilecull.h): 0 none, 2 per-tile trim, 3 segment trim
};

// tile columns [tx0, tx0 + n) of tile row ty that hold an instance of this Gaussian: the reference's rectangle row
// (tile_cull = 0), the ellipse span (1), the span trimmed by the depth limits - the very comparisons the preprocess kernel
// counted tiles_touched with (gs_preprocess.hip) and rounds 1-4 emitted instances from
__device__ __forceinline__ uint32_t tb_tile_span(const TileCull& c, const TbOwner& o, uint32_t ty, const float* __restrict__ depth_limit,
                                                 uint32_t grid_x, uint32_t grid_y, uint32_t& tx0) {
  uint32_t n = tilecull_row_span(c, ty, o.rmin & 0xFFFFu, o.rmax & 0xFFFFu, tx0);
  if (depth_limit) {
    if (o.dq == 2) n = tilecull_trim_span(depth_limit, grid_x, ty, o.depth, tx0, n);
    if (o.dq == 3) n = tilecull_trim_span_segments(depth_limit + (size_t)grid_x * grid_y, grid_x, ty, o.depth, tx0, n);
  }
  return n;
}
// region columns [c0, c0 + n) that hold an instance in one of the tile rows 4 ry .. 4 ry + 3: the hull of the rows' spans
__device__ __forceinline__ uint32_t tb_region_row_span(const TileCull& c, const TbOwner& o, uint32_t ry, const float* __restrict__ depth_limit,
                                                       uint32_t grid_x, uint32_t grid_y, uint32_t& c0) {
  const uint32_t rminy = o.rmin >> 16, rmaxy = o.rmax >> 16;
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
  for (uint32_t k = 0; k < RG_TILES; k++) {
    const uint32_t ty = ry * RG_TILES + k;
    if (ty < rminy || ty >= rmaxy) continue;
    uint32_t tx0;
    const uint32_t n = tb_tile_span(c, o, ty, depth_limit, grid_x, grid_y, tx0);
    if (n == 0) continue;
    lo = min(lo, tx0);
    hi = max(hi, tx0 + n - 1u);
  }
  if (lo > hi) return 0u;
  c0 = lo / RG_TILES;
  return hi / RG_TILES - c0 + 1u;
}

__device__ __forceinline__ void tb_load_owner(const GeomView& g, uint32_t id, int tile_cull, bool limits, TbOwner& o, TileCull& c) {
  const float4* rec = reinterpret_cast<const float4*>(&g.splat[id]);
  const uint4 tail = reinterpret_cast<const uint4*>(rec)[3];  // rect_min, rect_max, tiles, clamped | verdict << 8
  o.rmin = tail.x;
  o.rmax = tail.y;
  o.depth = 0.f;
  o.dq = 0;
  if (tile_cull) {
    const float4 ra = rec[0], rc = rec[1];
    c = tilecull_setup(1, ra.x, ra.y, rc.x, rc.y, rc.z, rc.w);
    o.depth = ra.z;
    o.dq = limits ? (int)((tail.w >> 8) & 3u) : 0;
  } else {
    c.mode = 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2a. entries per Gaussian of the depth order, summed per workgroup.  Two enumerations:
//   hull     one entry per (region row, region column in the hull of that row's four tile spans).  With culled spans a region of
//            the hull can hold NO tile (two short spans of a steep needle, a column apart): a harmless entry with an empty mask,
//            but then "every entry holds a tile" - the capacity argument - fails.  So the count is compared with tiles_touched:
//   row-wise if the hull has more entries than the Gaussian has tiles (never seen on the bench scenes; forced on every third
//            Gaussian by GsView.debug bit 1 so that the tests cover it) one entry per (TILE row, region column of its span): each
//            holds a tile of the span by construction.  The choice is left in bit 31 of the Gaussian's slot of the depth order.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GS_BLOCK) tb_entries_kernel(GeomView g, const uint32_t* __restrict__ n_ordered, uint32_t* __restrict__ order,
                                                              uint32_t grid_x, uint32_t grid_y, int tile_cull,
                                                              const float* __restrict__ depth_limit, int force_rowwise,
                                                              uint32_t* __restrict__ sums) {
  __shared__ uint32_t red[GS_BLOCK / 64];
  const uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x;
  uint32_t v = 0;
  if (i < *n_ordered && !g.hdr->overflow) {
    const uint32_t id = order[i] & ~TB_ROWWISE;
    const uint32_t tiles = g.tiles_touched[id];
    if (tiles) {
      TbOwner o;
      TileCull c;
      tb_load_owner(g, id, tile_cull, depth_limit != nullptr, o, c);
      const uint32_t rminy = o.rmin >> 16, rmaxy = o.rmax >> 16;
      uint32_t hull = 0;
      if (c.mode == 0 && o.dq == 0) {  // whole rectangle rows: closed form
        const uint32_t nrx = ((o.rmax & 0xFFFFu) - 1u) / RG_TILES - (o.rmin & 0xFFFFu) / RG_TILES + 1u;
        hull = nrx * ((rmaxy - 1u) / RG_TILES - rminy / RG_TILES + 1u);
      } else {
        for (uint32_t ry = rminy / RG_TILES; ry <= (rmaxy - 1u) / RG_TILES; ry++) {
          uint32_t c0;
          hull += tb_region_row_span(c, o, ry, depth_limit, grid_x, grid_y, c0);
        }
      }
      bool rowwise = hull > tiles || (force_rowwise && id % 3u == 0u);
      v = hull;
      if (rowwise) {
        v = 0;
        for (uint32_t ty = rminy; ty < rmaxy; ty++) {
          uint32_t tx0;
          const uint32_t n = tb_tile_span(c, o, ty, depth_limit, grid_x, grid_y, tx0);
          if (n) v += (tx0 + n - 1u) / RG_TILES - tx0 / RG_TILES + 1u;
        }
      }
      order[i] = id | (rowwise ? TB_ROWWISE : 0u);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ uint32_t tb_block_exclusive_scan256(uint32_t v, uint32_t* s_wsum, uint32_t& total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  __syncthreads();  // s_wsum may still be read from a previous call
  if (lane == 63) s_wsum[wid] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wid; w++) woff += s_wsum[w];
  total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
  return woff + inc - v;
}

// ---------------------------------------------------------------------------------------------------------------------
// 2b. entry emission in depth order.  A workgroup takes 256 consecutive Gaussians of the depth order and expands them into
// ROWS (region rows of the hull enumeration, tile rows of the row-wise one: binary search over the row-count prefix in LDS),
// evaluates each row's span of region columns once, scans the span lengths and then writes the entries with one binary search
// per entry: coalesced stores, no lane serialised on a Gaussian that covers the whole image.  The entry's tile mask is computed
// where it is written, from the owner's record in LDS.  Rows are processed TB_RC at a time to bound LDS.
// ---------------------------------------------------------------------------------------------------------------------
#define TB_RC 1024
__global__ void __launch_bounds__(GS_BLOCK) tb_emit_kernel(GeomView g, const uint32_t* __restrict__ n_ordered, uint32_t grid_x,
                                                           uint32_t grid_y, uint32_t rg_x, int tile_cull,
                                                           const float* __restrict__ depth_limit,
                                                           const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ block_base, uint32_t capacity,
                                                           uint32_t* __restrict__ ekeys, uint32_t* __restrict__ evals) {
  __shared__ uint32_t s_wsum[GS_BLOCK / 64];
  __shared__ uint32_t s_rowoff[GS_BLOCK + 1];  // exclusive prefix of rows per Gaussian
  __shared__ uint32_t s_id[GS_BLOCK];          // Gaussian index | TB_ROWWISE
  __shared__ TbOwner s_own[GS_BLOCK];
  __shared__ TileCull s_cull[GS_BLOCK];
  __shared__ uint32_t s_span_off[TB_RC + 1];   // exclusive prefix of span lengths within the row chunk
  __shared__ uint32_t s_span_key[TB_RC];       // region row << 16 | first region column of the span
  __shared__ uint32_t s_span_own[TB_RC];       // owner (index into s_id) | tile row inside the region << 16 (4: all four)
  if (g.hdr->overflow) return;
  const int tid = threadIdx.x;
  const uint32_t i = blockIdx.x * GS_BLOCK + tid;
  uint32_t rows = 0;
  s_id[tid] = 0;
  if (i < *n_ordered) {
    const uint32_t slot = order[i];
    const uint32_t id = slot & ~TB_ROWWISE;
    s_id[tid] = slot;
    if (g.tiles_touched[id]) {
      TbOwner o;
      TileCull c;
      tb_load_owner(g, id, tile_cull, depth_limit != nullptr, o, c);
      s_own[tid] = o;
      s_cull[tid] = c;
      const uint32_t rminy = o.rmin >> 16, rmaxy = o.rmax >> 16;
      rows = (slot & TB_ROWWISE) ? rmaxy - rminy : (rmaxy - 1u) / RG_TILES - rminy / RG_TILES + 1u;
    }
  }
  uint32_t total_rows;
  const uint32_t my_rowoff = tb_block_exclusive_scan256(rows, s_wsum, total_rows);
  s_rowoff[tid] = my_rowoff;
  if (tid == 0) s_rowoff[GS_BLOCK] = total_rows;
  const uint32_t base = block_base[blockIdx.x];
  const uint32_t expected = block_base[blockIdx.x + 1] - base;  // what the prefix sum reserved for this workgroup
  uint32_t written = 0;                                         // entries emitted so far (uniform)
  __syncthreads();
  for (uint32_t rbase = 0; rbase < total_rows; rbase += TB_RC) {
    const uint32_t nrow = min((uint32_t)TB_RC, total_rows - rbase);
    // each thread evaluates TB_RC / 256 consecutive rows
    uint32_t n4[TB_RC / GS_BLOCK], local = 0;
#pragma unroll
    for (int j = 0; j < TB_RC / GS_BLOCK; j++) {
      const uint32_t r = tid * (TB_RC / GS_BLOCK) + j;
      uint32_t n = 0;
      if (r < nrow) {
        const uint32_t rs = rbase + r;
        int lo = 0, hi = GS_BLOCK - 1;  // largest owner with s_rowoff[owner] <= rs
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const int mid = (lo + hi + 1) >> 1;
          if (s_rowoff[mid] <= rs) lo = mid; else hi = mid - 1;
        }
        const TbOwner o = s_own[lo];
        const uint32_t j_row = rs - s_rowoff[lo];
        uint32_t c0 = 0, ry, sub;
        if (s_id[lo] & TB_ROWWISE) {
          const uint32_t ty = (o.rmin >> 16) + j_row;
          uint32_t tx0;
          const uint32_t nt = tb_tile_span(s_cull[lo], o, ty, depth_limit, grid_x, grid_y, tx0);
          if (nt) {
            c0 = tx0 / RG_TILES;
            n = (tx0 + nt - 1u) / RG_TILES - c0 + 1u;
          }
          ry = ty / RG_TILES;
          sub = ty % RG_TILES;
        } else {
          ry = (o.rmin >> 16) / RG_TILES + j_row;
          sub = RG_TILES;
          n = tb_region_row_span(s_cull[lo], o, ry, depth_limit, grid_x, grid_y, c0);
        }
        s_span_key[r] = (ry << 16) | c0;
        s_span_own[r] = (uint32_t)lo | (sub << 16);
      }
      n4[j] = n;
      local += n;
    }
    uint32_t chunk_total;
    uint32_t off = tb_block_exclusive_scan256(local, s_wsum, chunk_total);
#pragma unroll
    for (int j = 0; j < TB_RC / GS_BLOCK; j++) {
      const uint32_t r = tid * (TB_RC / GS_BLOCK) + j;
      if (r < nrow) s_span_off[r] = off;
      off += n4[j];
    }
    if (tid == 0) s_span_off[nrow] = chunk_total;
    __syncthreads();
    // never write past what the prefix sum reserved (the two evaluations of the spans agree; this keeps the kernel memory-safe
    // even if they did not)
    const uint32_t room = expected - min(expected, written);
    const uint32_t emit = min(chunk_total, room);
    for (uint32_t k = tid; k < emit; k += GS_BLOCK) {
      int lo = 0, hi = (int)nrow - 1;  // largest row with s_span_off[row] <= k
#pragma unroll
      for (int it = 0; it < 10; it++) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_span_off[mid] <= k) lo = mid; else hi = mid - 1;
      }
      const uint32_t key = s_span_key[lo], own = s_span_own[lo] & 0xFFFFu, sub = s_span_own[lo] >> 16;
      const uint32_t ry = key >> 16, rx = (key & 0xFFFFu) + (k - s_span_off[lo]);
      const TbOwner o = s_own[own];
      const uint32_t rminy = o.rmin >> 16, rmaxy = o.rmax >> 16;
      uint32_t mask = 0;
#pragma unroll
      for (uint32_t kk = 0; kk < RG_TILES; kk++) {
        const uint32_t ty = ry * RG_TILES + kk;
        if ((sub != RG_TILES && kk != sub) || ty < rminy || ty >= rmaxy) continue;
        uint32_t tx0;
        const uint32_t nt = tb_tile_span(s_cull[own], o, ty, depth_limit, grid_x, grid_y, tx0);
        if (nt == 0) continue;
        // columns [tx0, tx0 + nt) cut to the region's four
        const int c0 = max((int)tx0 - (int)(rx * RG_TILES), 0), c1 = min((int)(tx0 + nt) - (int)(rx * RG_TILES), RG_TILES);
        if (c1 > c0) mask |= (((1u << (c1 - c0)) - 1u) << c0) << (4u * kk);
      }
      const uint32_t pos = base + written + k;
      if (pos < capacity) {
        ekeys[pos] = (ry * rg_x + rx) | (mask << 16);
        evals[pos] = s_id[own] & ~TB_ROWWISE;
      }
    }
    written += emit;
    __syncthreads();  // the chunk arrays are rewritten by the next iteration
  }
  // (unreachable when both evaluations agree) what is left of the reservation: entries with an empty mask
  for (uint32_t k = written + tid; k < expected; k += GS_BLOCK) {
    if (base + k < capacity) {
      ekeys[base + k] = 0u;
      evals[base + k] = 0u;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. regions: range of every region in the partitioned entries (two binary searches - no zero-fill + boundary pass over the
// entries), chunks of TB_CHUNK entries per region, exclusive prefix of the chunk counts, region of every chunk.  ONE workgroup.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) tb_regions_kernel(GeomHeader* hdr, const uint32_t* __restrict__ ekeys,
                                                          const uint32_t* __restrict__ n_entries_dev, uint32_t capacity, int NR,
                                                          TileBinView tb) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint32_t n = *n_entries_dev;
  if (n > capacity) {  // (cannot happen: entries <= instances <= capacity; the lists of this view would be incomplete)
    if (tid == 0) hdr->overflow = 1u;
    n = 0;
  }
  if (hdr->overflow) n = 0;
  auto lower = [&](uint32_t r) {  // first entry whose region id is >= r
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((ekeys[mid] & 0xFFFFu) < r) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < NR; base += 1024) {
    const int r = base + tid;
    uint32_t lo = 0, hi = 0;
    if (r < NR) {
      lo = lower((uint32_t)r);
      hi = lower((uint32_t)r + 1u);
      tb.region_ranges[r] = make_uint2(lo, hi);
    }
    const uint32_t v = (hi - lo + TB_CHUNK - 1u) / TB_CHUNK;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    const uint32_t first = carry_s + woff + inc - v;
    if (r < NR) {
      tb.chunk_first[r] = first;
      for (uint32_t c = 0; c < v; c++)
        if (first + c < tb.max_chunks) tb.chunk_region[first + c] = (uint32_t)r;
    }
    __syncthreads();
    if (tid == 1023) carry_s = first + v;
    __syncthreads();
  }
  if (tid == 0) tb.chunk_first[NR] = min(carry_s, tb.max_chunks);
}

// 5a. per chunk: entries of each of the region's sixteen tiles
__global__ void __launch_bounds__(GS_BLOCK) tb_tile_count_kernel(const uint32_t* __restrict__ ekeys, int NR, TileBinView tb) {
  __shared__ uint32_t s_cnt[GS_BLOCK / 64][16];
  const uint32_t c = blockIdx.x;
  if (c >= tb.chunk_first[NR]) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t r = tb.chunk_region[c];
  const uint2 rr = tb.region_ranges[r];
  const uint32_t e0 = rr.x + (c - tb.chunk_first[r]) * TB_CHUNK + (uint32_t)wid * (TB_CHUNK / (GS_BLOCK / 64));
  uint32_t my = 0;  // lane t < 16: this wave's count of tile t
#pragma unroll
  for (int j = 0; j < TB_CHUNK / GS_BLOCK; j++) {
    const uint32_t e = e0 + j * 64 + lane;
    const uint32_t m = e < rr.y ? ekeys[e] >> 16 : 0u;
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) {
      const uint32_t cnt = (uint32_t)__popcll(__ballot((m >> t) & 1u));
      if (lane == (int)t) my += cnt;
    }
  }
  if (lane < 16) s_cnt[wid][lane] = my;
  __syncthreads();
  if (tid < 16) tb.chunk_counts[(size_t)c * 16 + tid] = s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
}

// 5b. per region (one wave): exclusive prefix of every tile's counts over the region's chunks, in place; tile totals
__global__ void __launch_bounds__(64) tb_region_scan_kernel(int rg_x, int grid_x, int grid_y, TileBinView tb) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const uint32_t c0 = tb.chunk_first[r], nc = tb.chunk_first[r + 1] - c0;
  const int rx = r % rg_x, ry = r / rg_x;
  for (uint32_t t = 0; t < 16; t++) {
    uint32_t run = 0;
    for (uint32_t b = 0; b < nc; b += 64) {
      const bool in = b + lane < nc;
      uint32_t* p = tb.chunk_counts + (size_t)(c0 + b + lane) * 16 + t;
      const uint32_t v = in ? *p : 0u;
      uint32_t inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(inc, off, 64);
        if (lane >= off) inc += u;
      }
      if (in) *p = run + inc - v;
      run += (uint32_t)__shfl((int)inc, 63, 64);
    }
    const int ty = ry * RG_TILES + (int)(t >> 2), tx = rx * RG_TILES + (int)(t & 3u);
    if (lane == 0 && ty < grid_y && tx < grid_x) tb.tile_start[ty * grid_x + tx] = run;  // (the total for now)
  }
}

// 5c. all tiles in tile order: tile_start[] (exclusive prefix of the totals), ranges[] as identifyTileRanges leaves them
// (rasterizer_impl.cu:116-138: a tile no instance reaches keeps its zero-initialised (0, 0)), the instance total.  ONE workgroup.
__global__ void __launch_bounds__(1024) tb_tile_scan_kernel(GeomHeader* hdr, int T, uint32_t capacity, TileBinView tb,
                                                            uint2* __restrict__ ranges) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (hdr->overflow) return;  // (ranges stay as launch_bin_prepare zeroed them)
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < T; base += 1024) {
    const int i = base + tid;
    const uint32_t v = i < T ? tb.tile_start[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    const uint32_t start = carry_s + woff + inc - v;
    if (i < T) {
      tb.tile_start[i] = start;
      ranges[i] = v ? make_uint2(start, start + v) : make_uint2(0u, 0u);
    }
    __syncthreads();
    if (tid == 1023) carry_s = start + v;
    __syncthreads();
  }
  if (tid == 0) {
    tb.tile_start[T] = carry_s;
    hdr->sort_n = carry_s;                       // instances the lists hold (= num_rendered)
    if (carry_s > capacity) hdr->overflow = 1u;  // (cannot happen: launch_bin_prepare has compared num_rendered with the capacity)
  }
}

// 6. the lists: position = start of the tile's list + entries of the tile in the region's earlier chunks + ... in earlier waves
// of this chunk + ... earlier in this wave
__global__ void __launch_bounds__(GS_BLOCK) tb_write_kernel(const GeomHeader* __restrict__ hdr, const uint32_t* __restrict__ ekeys,
                                                            const uint32_t* __restrict__ evals, int NR, int rg_x, int grid_x,
                                                            int grid_y, uint32_t capacity, TileBinView tb,
                                                            uint32_t* __restrict__ point_list) {
  constexpr int NW = GS_BLOCK / 64, PER_WAVE = TB_CHUNK / NW, ROUNDS = PER_WAVE / 64;
  __shared__ uint32_t s_wcnt[NW][16];
  __shared__ uint32_t s_base[16];
  const uint32_t c = blockIdx.x;
  if (c >= tb.chunk_first[NR] || hdr->overflow) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t r = tb.chunk_region[c];
  const uint2 rr = tb.region_ranges[r];
  const uint32_t e0 = rr.x + (c - tb.chunk_first[r]) * TB_CHUNK + (uint32_t)wid * PER_WAVE;
  uint32_t m[ROUNDS], id[ROUNDS];
#pragma unroll
  for (int j = 0; j < ROUNDS; j++) {
    const uint32_t e = e0 + j * 64 + lane;
    const bool in = e < rr.y;
    m[j] = in ? ekeys[e] >> 16 : 0u;
    id[j] = in ? evals[e] : 0u;
  }
  uint32_t my = 0;  // lane t < 16: this wave's count of tile t
#pragma unroll
  for (int j = 0; j < ROUNDS; j++) {
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) {
      const uint32_t cnt = (uint32_t)__popcll(__ballot((m[j] >> t) & 1u));
      if (lane == (int)t) my += cnt;
    }
  }
  if (lane < 16) s_wcnt[wid][lane] = my;
  if (tid < 16) {
    const int rx = (int)(r % (uint32_t)rg_x), ry = (int)(r / (uint32_t)rg_x);
    const int ty = ry * RG_TILES + (tid >> 2), tx = rx * RG_TILES + (tid & 3);
    s_base[tid] = (ty < grid_y && tx < grid_x ? tb.tile_start[ty * grid_x + tx] : 0u) + tb.chunk_counts[(size_t)c * 16 + tid];
  }
  __syncthreads();
  uint32_t run = 0;  // lane t < 16: where this wave's next entry of tile t goes
  if (lane < 16) {
    run = s_base[lane];
    for (int w = 0; w < wid; w++) run += s_wcnt[w][lane];
  }
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int j = 0; j < ROUNDS; j++) {
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) {
      const unsigned long long b = __ballot((m[j] >> t) & 1u);
      const uint32_t at = (uint32_t)__shfl((int)run, (int)t, 64);
      if ((m[j] >> t) & 1u) {
        const uint32_t pos = at + (uint32_t)__popcll(b & lt_mask);
        if (pos < capacity) point_list[pos] = id[j];
      }
      if (lane == (int)t) run += (uint32_t)__popcll(b);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
int launch_tile_binning(const GeomView& g, const SortBufs& bv, const TileBinView& tb, int P, int64_t capacity, int grid_x, int grid_y,
                        int tile_cull, const float* tile_depth_limit, uint2* ranges, int force_rowwise, hipStream_t s, int debug) {
  const int rg_x = (grid_x + RG_TILES - 1) / RG_TILES, rg_y = (grid_y + RG_TILES - 1) / RG_TILES;
  const int NR = rg_x * rg_y, T = grid_x * grid_y;
  if (NR > 65536) return GS_E_UNSUPPORTED;  // (the region id shares a 32-bit word with the 16-bit tile mask: images beyond 16 k x 16 k)
  const uint32_t cap32 = (uint32_t)capacity;
  const int nb = (P + GS_BLOCK - 1) / GS_BLOCK;
  const float* limit = tile_cull ? tile_depth_limit : nullptr;
  uint32_t* order = g.gsort.vals[0];  // the depth order (launch_radix_sort, 4 passes: ends in half 0)
  const int bits = (int)gs_higher_msb((uint32_t)NR);
  const int passes = (bits + RS_BITS - 1) / RS_BITS;
  const int start = passes == 1 ? 0 : 1;  // the partition ends in half 1: half 0 of the ids is point_list
  int rc;
  {
    GS_PROF(ST_DUPLICATE, s);
    hipLaunchKernelGGL(tb_entries_kernel, dim3(nb), dim3(GS_BLOCK), 0, s, g, &g.hdr->n_ordered, order, (uint32_t)grid_x,
                       (uint32_t)grid_y, tile_cull, limit, force_rowwise, g.sorted_sums);
    rc = launch_scan_sums(g.sorted_sums, nb, s);
    if (rc) return rc;
    GS_LAUNCH_CHECK(s, debug);
    hipLaunchKernelGGL(tb_emit_kernel, dim3(nb), dim3(GS_BLOCK), 0, s, g, &g.hdr->n_ordered, (uint32_t)grid_x, (uint32_t)grid_y,
                       (uint32_t)rg_x, tile_cull, limit, order, g.sorted_sums, cap32, bv.keys[start], bv.vals[start]);
    GS_LAUNCH_CHECK(s, debug);
  }
  const uint32_t* n_entries = g.sorted_sums + nb;
  {
    GS_PROF(ST_SORT, s);
    rc = launch_radix_sort(bv, n_entries, capacity, bits, start, nullptr, s, debug);
    if (rc) return rc;
  }
  {
    GS_PROF(ST_RANGES, s);
    hipLaunchKernelGGL(tb_regions_kernel, dim3(1), dim3(1024), 0, s, g.hdr, bv.keys[1], n_entries, cap32, NR, tb);
    GS_LAUNCH_CHECK(s, debug);
    hipLaunchKernelGGL(tb_tile_count_kernel, dim3(tb.max_chunks), dim3(GS_BLOCK), 0, s, bv.keys[1], NR, tb);
    hipLaunchKernelGGL(tb_region_scan_kernel, dim3(NR), dim3(64), 0, s, rg_x, grid_x, grid_y, tb);
    GS_LAUNCH_CHECK(s, debug);
    hipLaunchKernelGGL(tb_tile_scan_kernel, dim3(1), dim3(1024), 0, s, g.hdr, T, cap32, tb, ranges);
    hipLaunchKernelGGL(tb_write_kernel, dim3(tb.max_chunks), dim3(GS_BLOCK), 0, s, g.hdr, bv.keys[1], bv.vals[1], NR, rg_x, grid_x,
                       grid_y, cap32, tb, bv.vals[0]);
    GS_LAUNCH_CHECK(s, debug);
  }
  return 0;
}
