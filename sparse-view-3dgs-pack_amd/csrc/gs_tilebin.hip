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
// 2a. entries per Gaussian of the depth order (cnt[i], bit 31 = row-wise), summed per workgroup.  Two enumerations:
//   hull     one entry per (region row, region column in the hull of that row's four tile spans).  With culled spans a region of
//            the hull can hold NO tile (two short spans of a steep needle, a column apart): a harmless entry with an empty mask,
//            but then "every entry holds a tile" - the capacity argument - fails.  So the count is compared with tiles_touched:
//   row-wise if the hull has more entries than the Gaussian has tiles (never seen on the bench scenes; forced on every third
//            Gaussian by GsView.debug bit 1 so that the tests cover it) one entry per (TILE row, region column of its span): each
//            holds a tile of the span by construction.
// Where the preprocess kernel could count on its way (the reference's rectangles, culled spans without depth limits) the number
// is in the record (bits 16-31 of `clamped`) and this kernel is two dependent loads per Gaussian.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GS_BLOCK) tb_entries_kernel(GeomView g, const uint32_t* __restrict__ n_ordered,
                                                              const uint32_t* __restrict__ order, uint32_t* __restrict__ cnt,
                                                              uint32_t grid_x, uint32_t grid_y, int tile_cull,
                                                              const float* __restrict__ depth_limit, int force_rowwise,
                                                              uint32_t* __restrict__ sums) {
  __shared__ uint32_t red[GS_BLOCK / 64];
  const uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x;
  uint32_t v = 0;
  if (i < *n_ordered && !g.hdr->overflow) {
    const uint32_t id = order[i];
    const uint4 tail = reinterpret_cast<const uint4*>(&g.splat[id])[3];  // rect_min, rect_max, tiles, clamped | verdict << 8 | entries << 16
    const uint32_t pre = tail.w >> 16, tiles = tail.z;
    uint32_t word = 0;
    if (tiles && pre != TB_ENTRIES_UNKNOWN && !force_rowwise) {
      v = pre & 0x7FFFu;
      word = v | ((pre & 0x8000u) ? TB_ROWWISE : 0u);
    } else if (tiles) {
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
      const bool rowwise = hull > tiles || (force_rowwise && id % 3u == 0u);
      v = hull;
      if (rowwise) {
        v = 0;
        for (uint32_t ty = rminy; ty < rmaxy; ty++) {
          uint32_t tx0;
          const uint32_t n = tb_tile_span(c, o, ty, depth_limit, grid_x, grid_y, tx0);
          if (n) v += (tx0 + n - 1u) / RG_TILES - tx0 / RG_TILES + 1u;
        }
      }
      word = v | (rowwise ? TB_ROWWISE : 0u);
    }
    cnt[i] = word;
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

// the four tile spans of one region row (tile rows 4 ry .. 4 ry + 3; n = 0 outside the rectangle) and their hull in region columns
struct TbRowSpans {
  uint32_t tx0[RG_TILES], n[RG_TILES];
};
__device__ __forceinline__ uint32_t tb_region_row(const TileCull& c, const TbOwner& o, uint32_t ry, const float* __restrict__ depth_limit,
                                                  uint32_t grid_x, uint32_t grid_y, TbRowSpans& rs, uint32_t& c0) {
  const uint32_t rminy = o.rmin >> 16, rmaxy = o.rmax >> 16;
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
  for (uint32_t k = 0; k < RG_TILES; k++) {
    const uint32_t ty = ry * RG_TILES + k;
    rs.n[k] = 0;
    rs.tx0[k] = 0;
    if (ty < rminy || ty >= rmaxy) continue;
    rs.n[k] = tb_tile_span(c, o, ty, depth_limit, grid_x, grid_y, rs.tx0[k]);
    if (rs.n[k] == 0) continue;
    lo = min(lo, rs.tx0[k]);
    hi = max(hi, rs.tx0[k] + rs.n[k] - 1u);
  }
  if (lo > hi) return 0u;
  c0 = lo / RG_TILES;
  return hi / RG_TILES - c0 + 1u;
}
// bit 4 k + col of the mask = tile (4 ry + k, 4 rx + col) holds an instance
__device__ __forceinline__ uint32_t tb_mask(const TbRowSpans& rs, uint32_t rx) {
  uint32_t mask = 0;
#pragma unroll
  for (uint32_t k = 0; k < RG_TILES; k++) {
    if (rs.n[k] == 0) continue;
    const int a0 = max((int)rs.tx0[k] - (int)(rx * RG_TILES), 0), a1 = min((int)(rs.tx0[k] + rs.n[k]) - (int)(rx * RG_TILES), RG_TILES);
    if (a1 > a0) mask |= (((1u << (a1 - a0)) - 1u) << a0) << (4u * k);
  }
  return mask;
}

// ---------------------------------------------------------------------------------------------------------------------
// 2b. entry emission in depth order.  A workgroup takes 256 consecutive Gaussians of the depth order; a thread's Gaussian has
// cnt[i] entries (a handful: 5 on average at C3) starting at the workgroup's base + the exclusive scan of the counts.  The thread
// walks its own region rows / columns and writes its entries itself - no row tables, no searches (rounds 1-4's instance
// emission and this kernel's first form expanded everything cooperatively through LDS prefix tables and two binary searches
// per entry: 67 us at C3; a Gaussian's entries are too few for that machinery to pay).  The rare large Gaussian (more than
// TB_SMALL entries: it would keep its wave waiting) is put on a list instead and expanded afterwards by the whole workgroup,
// a thread per entry.
// ---------------------------------------------------------------------------------------------------------------------
#define TB_SMALL 24
#define TB_STAGE 4096  // entries of a workgroup staged in LDS (32 KB); what lies beyond goes out directly
#define TB_BIG_ROWS 512   // rows of a large Gaussian expanded at a time (region rows, or tile rows of the row-wise enumeration)
// CULL = false: the reference's rectangles (GsView.tile_cull = 0) - no ellipse state in registers, every span is the rectangle's row
template <bool CULL>
__global__ void __launch_bounds__(GS_BLOCK) tb_emit_kernel(GeomView g, const uint32_t* __restrict__ n_ordered, uint32_t grid_x,
                                                           uint32_t grid_y, uint32_t rg_x, int tile_cull_arg,
                                                           const float* __restrict__ depth_limit_arg,
                                                           const uint32_t* __restrict__ order, const uint32_t* __restrict__ cnt,
                                                           const uint32_t* __restrict__ block_base, uint32_t capacity,
                                                           uint32_t* __restrict__ ekeys, uint32_t* __restrict__ evals) {
  __shared__ uint32_t s_wsum[GS_BLOCK / 64];
  __shared__ uint32_t s_big[GS_BLOCK];         // threads whose Gaussian is expanded by the whole workgroup
  __shared__ uint32_t s_nbig;
  __shared__ uint32_t s_rowoff[TB_BIG_ROWS + 1];
  __shared__ uint32_t s_rowc0[TB_BIG_ROWS];
  // the threads' own entries are staged here and leave the workgroup as contiguous runs: written straight from the threads'
  // loops every lane of a store instruction lands in another cache line (10 M four-byte requests at C3)
  __shared__ uint32_t s_ek[TB_STAGE], s_ev[TB_STAGE];
  if (g.hdr->overflow) return;
  const int tile_cull = CULL ? tile_cull_arg : 0;
  const float* const depth_limit = CULL ? depth_limit_arg : nullptr;
  const int tid = threadIdx.x;
  const uint32_t i = blockIdx.x * GS_BLOCK + tid;
  if (tid == 0) s_nbig = 0;
  for (uint32_t k = tid; k < TB_STAGE; k += GS_BLOCK) s_ek[k] = 0xFFFFFFFFu;   // (region 0xFFFF with a full mask: never a real entry)
  uint32_t word = 0, id = 0;
  TbOwner o = {};
  TileCull c;
  c.mode = 0;
  if (i < *n_ordered) {
    word = cnt[i];
    if (word & ~TB_ROWWISE) {
      id = order[i];
      tb_load_owner(g, id, tile_cull, depth_limit != nullptr, o, c);
    }
  }
  const uint32_t v = word & ~TB_ROWWISE;
  const bool rowwise = (word & TB_ROWWISE) != 0u;
  uint32_t total;
  const uint32_t off = tb_block_exclusive_scan256(v, s_wsum, total);
  const uint32_t base = block_base[blockIdx.x];
  const uint32_t room = block_base[blockIdx.x + 1] - base;  // what the prefix sum reserved for this workgroup (= total)
  auto put = [&](uint32_t at, uint32_t region, uint32_t mask, uint32_t gid) {
    if (at < room && base + at < capacity) {   // (never past the reservation, even if the two evaluations of the spans disagreed)
      ekeys[base + at] = region | (mask << 16);
      evals[base + at] = gid;
    }
  };
  auto stage = [&](uint32_t at, uint32_t region, uint32_t mask, uint32_t gid) {
    if (at < TB_STAGE) {
      s_ek[at] = region | (mask << 16);
      s_ev[at] = gid;
    } else {
      put(at, region, mask, gid);
    }
  };
  const uint32_t rminy = o.rmin >> 16, rmaxy = o.rmax >> 16;
  if (v > TB_SMALL) {
    s_big[atomicAdd(&s_nbig, 1u)] = (uint32_t)tid;
  } else if (v) {
    uint32_t at = off;
    const uint32_t end = off + v;
    if (!rowwise) {
      for (uint32_t ry = rminy / RG_TILES; ry <= (rmaxy - 1u) / RG_TILES && at < end; ry++) {
        TbRowSpans rs;
        uint32_t c0 = 0;
        const uint32_t n = tb_region_row(c, o, ry, depth_limit, grid_x, grid_y, rs, c0);
        for (uint32_t k = 0; k < n && at < end; k++, at++) stage(at, ry * rg_x + c0 + k, tb_mask(rs, c0 + k), id);
      }
    } else {
      for (uint32_t ty = rminy; ty < rmaxy && at < end; ty++) {
        TbRowSpans rs;
        uint32_t tx0;
        const uint32_t nt = tb_tile_span(c, o, ty, depth_limit, grid_x, grid_y, tx0);
        if (nt == 0) continue;
#pragma unroll
        for (uint32_t k = 0; k < RG_TILES; k++) {   // (a select, not rs.n[ty % 4]: a dynamically indexed array lives in scratch)
          rs.n[k] = k == ty % RG_TILES ? nt : 0u;
          rs.tx0[k] = k == ty % RG_TILES ? tx0 : 0u;
        }
        const uint32_t c0 = tx0 / RG_TILES, n = (tx0 + nt - 1u) / RG_TILES - c0 + 1u;
        for (uint32_t k = 0; k < n && at < end; k++, at++) stage(at, (ty / RG_TILES) * rg_x + c0 + k, tb_mask(rs, c0 + k), id);
      }
    }
    for (; at < end; at++) stage(at, 0u, 0u, id);   // (unreachable when count and emission agree: entries with an empty mask)
  }
  __syncthreads();
  // ---- the staged entries out (the large Gaussians' slots inside the staged range are written by the loop below: skipped here
  //      through their mask word, which stays at the 0xFFFFFFFF it is initialised with)
  {
    const uint32_t staged = min(min(total, room), (uint32_t)TB_STAGE);
    for (uint32_t k = tid; k < staged; k += GS_BLOCK) {
      const uint32_t key = s_ek[k];
      if (key != 0xFFFFFFFFu && base + k < capacity) {
        ekeys[base + k] = key;
        evals[base + k] = s_ev[k];
      }
    }
  }
  // ---- the large ones, a thread per entry
  const uint32_t nbig = s_nbig;
  if constexpr (!CULL) {
    // the reference's rectangles: every row of a Gaussian holds the same number of regions, so entry k of it is row k / nrx,
    // column k % nrx - all the large Gaussians of the workgroup are expanded in ONE pass over their concatenated entries
    // (a loop over them with row tables and barriers per Gaussian, as the culled form below has it, was most of this
    // kernel's 66-86 us at C3: a tenth of the Gaussians there reach more than TB_SMALL regions)
    __shared__ TbOwner s_bo[GS_BLOCK];
    __shared__ uint32_t s_bid[GS_BLOCK], s_boff[GS_BLOCK], s_bword[GS_BLOCK], s_bstart[GS_BLOCK + 1];
    if (v > TB_SMALL) {
      uint32_t slot = 0;
      for (uint32_t b = 0; b < nbig; b++) slot = s_big[b] == (uint32_t)tid ? b : slot;
      s_bo[slot] = o;
      s_bid[slot] = id;
      s_boff[slot] = off;
      s_bword[slot] = word;
    }
    __syncthreads();
    uint32_t tot_big;
    const uint32_t mine = (uint32_t)tid < nbig ? (s_bword[tid] & ~TB_ROWWISE) : 0u;
    const uint32_t st = tb_block_exclusive_scan256(mine, s_wsum, tot_big);
    s_bstart[tid] = st;
    if (tid == 0) s_bstart[GS_BLOCK] = tot_big;
    __syncthreads();
    for (uint32_t k = tid; k < tot_big; k += GS_BLOCK) {
      int lo = 0, hi = (int)nbig - 1;  // largest b with s_bstart[b] <= k
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_bstart[mid] <= k) lo = mid; else hi = mid - 1;
      }
      const TbOwner bo = s_bo[lo];
      const uint32_t local = k - s_bstart[lo];
      const uint32_t bx0 = bo.rmin & 0xFFFFu, bx1 = bo.rmax & 0xFFFFu, by0 = bo.rmin >> 16, by1 = bo.rmax >> 16;
      const uint32_t rx0 = bx0 / RG_TILES, nrx = (bx1 - 1u) / RG_TILES - rx0 + 1u;
      const bool brow = (s_bword[lo] & TB_ROWWISE) != 0u;
      const uint32_t row = local / nrx, rx = rx0 + local % nrx;
      TbRowSpans rs;
      uint32_t ry;
      if (brow) {   // (only the test hook sends a rectangle here: one entry per tile row)
        const uint32_t ty = by0 + row;
        ry = ty / RG_TILES;
#pragma unroll
        for (uint32_t kk = 0; kk < RG_TILES; kk++) {
          rs.n[kk] = kk == ty % RG_TILES ? bx1 - bx0 : 0u;
          rs.tx0[kk] = bx0;
        }
      } else {
        ry = by0 / RG_TILES + row;
#pragma unroll
        for (uint32_t kk = 0; kk < RG_TILES; kk++) {
          const uint32_t ty = ry * RG_TILES + kk;
          rs.n[kk] = (ty >= by0 && ty < by1) ? bx1 - bx0 : 0u;
          rs.tx0[kk] = bx0;
        }
      }
      put(s_boff[lo] + local, ry * rg_x + rx, tb_mask(rs, rx), s_bid[lo]);
    }
    return;
  }
  if constexpr (CULL) {
    // culled spans: one large Gaussian after the other through a row table (span per row -> prefix -> a thread per entry).
    // (Round 5 also built the one-table form over the concatenated rows of ALL of them, as the rectangles have it above: 99 us
    //  against 76 at C3 - every entry re-derives its row's four ellipse spans from LDS-resident state there.)
    __shared__ TbOwner s_o;
    __shared__ TileCull s_c;
    __shared__ uint32_t s_big_id, s_big_off, s_big_word;
    for (uint32_t b = 0; b < nbig; b++) {
      if ((uint32_t)tid == s_big[b]) {
        s_o = o;
        s_c = c;
        s_big_id = id;
        s_big_off = off;
        s_big_word = word;
      }
      __syncthreads();
      const TbOwner bo = s_o;
      const uint32_t bid = s_big_id, bword = s_big_word, bv = bword & ~TB_ROWWISE;
      const bool brow = (bword & TB_ROWWISE) != 0u;
      const uint32_t r_lo = brow ? (bo.rmin >> 16) : (bo.rmin >> 16) / RG_TILES;
      const uint32_t r_hi = brow ? (bo.rmax >> 16) : ((bo.rmax >> 16) - 1u) / RG_TILES + 1u;   // rows [r_lo, r_hi)
      uint32_t written = 0;
      for (uint32_t r0 = r_lo; r0 < r_hi; r0 += TB_BIG_ROWS) {
        const uint32_t nrow = min((uint32_t)TB_BIG_ROWS, r_hi - r0);
        uint32_t n2[TB_BIG_ROWS / GS_BLOCK], local = 0;   // spans of up to two rows per thread, their exclusive prefix
#pragma unroll
        for (int j = 0; j < TB_BIG_ROWS / GS_BLOCK; j++) {
          const uint32_t r = tid * (TB_BIG_ROWS / GS_BLOCK) + j;
          uint32_t n = 0, c0 = 0;
          if (r < nrow) {
            if (brow) {
              uint32_t tx0;
              const uint32_t nt = tb_tile_span(s_c, bo, r0 + r, depth_limit, grid_x, grid_y, tx0);
              if (nt) {
                c0 = tx0 / RG_TILES;
                n = (tx0 + nt - 1u) / RG_TILES - c0 + 1u;
              }
            } else {
              TbRowSpans rs;
              n = tb_region_row(s_c, bo, r0 + r, depth_limit, grid_x, grid_y, rs, c0);
            }
            s_rowc0[r] = c0;
          }
          n2[j] = n;
          local += n;
        }
        uint32_t chunk_total;
        uint32_t roff = tb_block_exclusive_scan256(local, s_wsum, chunk_total);
#pragma unroll
        for (int j = 0; j < TB_BIG_ROWS / GS_BLOCK; j++) {
          const uint32_t r = tid * (TB_BIG_ROWS / GS_BLOCK) + j;
          if (r < nrow) s_rowoff[r] = roff;
          roff += n2[j];
        }
        if (tid == 0) s_rowoff[nrow] = chunk_total;
        __syncthreads();
        for (uint32_t k = tid; k < chunk_total && written + k < bv; k += GS_BLOCK) {
          int lo = 0, hi = (int)nrow - 1;  // largest row with s_rowoff[row] <= k
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_rowoff[mid] <= k) lo = mid; else hi = mid - 1;
          }
          const uint32_t row = r0 + (uint32_t)lo, rx = s_rowc0[lo] + (k - s_rowoff[lo]);
          TbRowSpans rs;
          uint32_t ry;
          if (brow) {
            uint32_t tx0;
            const uint32_t nt = tb_tile_span(s_c, bo, row, depth_limit, grid_x, grid_y, tx0);
#pragma unroll
            for (uint32_t kk = 0; kk < RG_TILES; kk++) {
              rs.n[kk] = kk == row % RG_TILES ? nt : 0u;
              rs.tx0[kk] = kk == row % RG_TILES ? tx0 : 0u;
            }
            ry = row / RG_TILES;
          } else {
            uint32_t c0;
            (void)tb_region_row(s_c, bo, row, depth_limit, grid_x, grid_y, rs, c0);
            ry = row;
          }
          put(s_big_off + written + k, ry * rg_x + rx, tb_mask(rs, rx), bid);
        }
        written += chunk_total;
        __syncthreads();  // the row tables are rewritten by the next chunk
      }
      for (uint32_t k = min(written, bv) + tid; k < bv; k += GS_BLOCK) put(s_big_off + k, 0u, 0u, bid);  // (unreachable, as above)
      __syncthreads();    // (s_o ... are rewritten for the next large Gaussian)
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
        if (first + c < tb.max_chunks)
          tb.chunk_desc[first + c] = make_uint4(lo + c * TB_CHUNK, min(lo + (c + 1u) * TB_CHUNK, hi), (uint32_t)r, 0u);
    }
    __syncthreads();
    if (tid == 1023) carry_s = first + v;
    __syncthreads();
  }
  if (tid == 0) tb.chunk_first[NR] = min(carry_s, tb.max_chunks);
}

// 4'. the same from the partition's own digit totals when it was ONE pass (regions <= 512: digit = region id): no search at all
__global__ void __launch_bounds__(1024) tb_regions_from_counts_kernel(GeomHeader* hdr, const uint32_t* __restrict__ counts,
                                                                      const uint32_t* __restrict__ n_entries_dev, uint32_t capacity,
                                                                      int NR, TileBinView tb) {
  __shared__ uint32_t wsum[16], wsum2[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t n = *n_entries_dev;
  if (n > capacity && tid == 0) hdr->overflow = 1u;  // (cannot happen: entries <= instances <= capacity)
  const bool dead = n > capacity || hdr->overflow != 0u;
  const uint32_t cnt = (tid < NR && !dead) ? counts[tid] : 0u;   // (NR <= 512 < 1024 threads)
  const uint32_t v = (cnt + TB_CHUNK - 1u) / TB_CHUNK;
  uint32_t inc = cnt, inc2 = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64), t2 = __shfl_up(inc2, off, 64);
    if (lane >= off) {
      inc += t;
      inc2 += t2;
    }
  }
  if (lane == 63) {
    wsum[wid] = inc;
    wsum2[wid] = inc2;
  }
  __syncthreads();
  uint32_t woff = 0, woff2 = 0;
  for (int w = 0; w < wid; w++) {
    woff += wsum[w];
    woff2 += wsum2[w];
  }
  const uint32_t lo = woff + inc - cnt, first = woff2 + inc2 - v;
  if (tid < NR) {
    tb.region_ranges[tid] = make_uint2(lo, lo + cnt);
    tb.chunk_first[tid] = first;
    for (uint32_t c = 0; c < v; c++)
      if (first + c < tb.max_chunks)
        tb.chunk_desc[first + c] = make_uint4(lo + c * TB_CHUNK, min(lo + (c + 1u) * TB_CHUNK, lo + cnt), (uint32_t)tid, 0u);
  }
  if (tid == NR - 1) tb.chunk_first[NR] = min(first + v, tb.max_chunks);
}

// 5a. per chunk (one wave each; a workgroup's four waves stride over the chunks on their own - no LDS, no barrier): entries of
// each of the region's sixteen tiles
__global__ void __launch_bounds__(GS_BLOCK) tb_tile_count_kernel(const uint32_t* __restrict__ ekeys, int NR, TileBinView tb) {
  constexpr int NW = GS_BLOCK / 64;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t nchunks = tb.chunk_first[NR];
  // (the grid is sized by the CAPACITY - the host does not know how many entries there are - and most of it would be empty
  //  workgroups: a pool that strides over the chunks instead)
  for (uint32_t c = blockIdx.x * NW + wid; c < nchunks; c += gridDim.x * NW) {
    const uint4 d = tb.chunk_desc[c];
    uint32_t my = 0;  // lane t < 16: this chunk's count of tile t
#pragma unroll
    for (int j = 0; j < TB_CHUNK / 64; j++) {
      const uint32_t e = d.x + j * 64 + lane;
      const uint32_t m = e < d.y ? ekeys[e] >> 16 : 0u;
#pragma unroll
      for (uint32_t t = 0; t < 16; t++) {
        const uint32_t cnt = (uint32_t)__popcll(__ballot((m >> t) & 1u));
        if (lane == (int)t) my += cnt;
      }
    }
    if (lane < 16) tb.chunk_counts[(size_t)c * 16 + lane] = my;
  }
}

// 5b. per region (one wave): exclusive prefix of every tile's counts over the region's chunks, in place; tile totals.
// A lane takes a chunk's whole row of sixteen counts (one 64-byte load), the sixteen scans run side by side
__global__ void __launch_bounds__(64) tb_region_scan_kernel(int rg_x, int grid_x, int grid_y, TileBinView tb) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const uint32_t c0 = tb.chunk_first[r], nc = tb.chunk_first[r + 1] - c0;
  const int rx = r % rg_x, ry = r / rg_x;
  uint32_t run[16];
#pragma unroll
  for (int t = 0; t < 16; t++) run[t] = 0;
  for (uint32_t b = 0; b < nc; b += 64) {
    const bool in = b + lane < nc;
    uint4* p = reinterpret_cast<uint4*>(tb.chunk_counts + (size_t)(c0 + b + lane) * 16);
    uint32_t v[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint4 x = in ? p[q] : make_uint4(0u, 0u, 0u, 0u);
      v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
    }
    uint32_t ex[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
      uint32_t inc = v[t];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(inc, off, 64);
        if (lane >= off) inc += u;
      }
      ex[t] = run[t] + inc - v[t];
      run[t] += (uint32_t)__shfl((int)inc, 63, 64);
    }
    if (in) {
#pragma unroll
      for (int q = 0; q < 4; q++) p[q] = make_uint4(ex[4 * q], ex[4 * q + 1], ex[4 * q + 2], ex[4 * q + 3]);
    }
  }
  if (lane < 16) {
    const int ty = ry * RG_TILES + (lane >> 2), tx = rx * RG_TILES + (lane & 3);
    uint32_t mine = 0;
#pragma unroll
    for (int t = 0; t < 16; t++) mine = lane == t ? run[t] : mine;
    if (ty < grid_y && tx < grid_x) tb.tile_start[ty * grid_x + tx] = mine;  // (the total for now)
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
  for (int base = 0; base < T; base += 4096) {   // four consecutive tiles per thread: 1080p is two rounds
    const int i0 = base + 4 * tid;
    uint32_t v[4], tsum = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      v[e] = i0 + e < T ? tb.tile_start[i0 + e] : 0u;
      tsum += v[e];
    }
    uint32_t inc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    uint32_t start = carry_s + woff + inc - tsum;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (i0 + e < T) {
        tb.tile_start[i0 + e] = start;
        ranges[i0 + e] = v[e] ? make_uint2(start, start + v[e]) : make_uint2(0u, 0u);
      }
      start += v[e];
    }
    __syncthreads();
    if (tid == 1023) carry_s = start;
    __syncthreads();
  }
  if (tid == 0) {
    tb.tile_start[T] = carry_s;
    hdr->sort_n = carry_s;                       // instances the lists hold (= num_rendered)
    if (carry_s > capacity) hdr->overflow = 1u;  // (cannot happen: launch_bin_prepare has compared num_rendered with the capacity)
  }
}

// 6. the lists (one wave per chunk, as above): position = start of the tile's list + entries of the tile in the region's earlier
// chunks + ... earlier in this chunk
__global__ void __launch_bounds__(GS_BLOCK) tb_write_kernel(const GeomHeader* __restrict__ hdr, const uint32_t* __restrict__ ekeys,
                                                            const uint32_t* __restrict__ evals, int NR, int rg_x, int grid_x,
                                                            int grid_y, uint32_t capacity, TileBinView tb,
                                                            uint32_t* __restrict__ point_list) {
  constexpr int NW = GS_BLOCK / 64, ROUNDS = TB_CHUNK / 64;
  if (hdr->overflow) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t nchunks = tb.chunk_first[NR];
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (uint32_t c = blockIdx.x * NW + wid; c < nchunks; c += gridDim.x * NW) {
    const uint4 d = tb.chunk_desc[c];
    const uint32_t r = d.z;
    uint32_t m[ROUNDS], id[ROUNDS];
#pragma unroll
    for (int j = 0; j < ROUNDS; j++) {
      const uint32_t e = d.x + j * 64 + lane;
      const bool in = e < d.y;
      m[j] = in ? ekeys[e] >> 16 : 0u;
      id[j] = in ? evals[e] : 0u;
    }
    uint32_t run = 0;  // lane t < 16: where this chunk's next entry of tile t goes
    if (lane < 16) {
      const int rx = (int)(r % (uint32_t)rg_x), ry = (int)(r / (uint32_t)rg_x);
      const int ty = ry * RG_TILES + (lane >> 2), tx = rx * RG_TILES + (lane & 3);
      run = (ty < grid_y && tx < grid_x ? tb.tile_start[ty * grid_x + tx] : 0u) + tb.chunk_counts[(size_t)c * 16 + lane];
    }
#pragma unroll
    for (int j = 0; j < ROUNDS; j++) {
      const uint32_t mj = m[j], idj = id[j];
#pragma unroll
      for (uint32_t t = 0; t < 16; t++) {
        const unsigned long long b = __ballot((mj >> t) & 1u);
        const uint32_t at = (uint32_t)__shfl((int)run, (int)t, 64);
        if ((mj >> t) & 1u) {
          const uint32_t pos = at + (uint32_t)__popcll(b & lt_mask);
          if (pos < capacity) point_list[pos] = idj;
        }
        if (lane == (int)t) run += (uint32_t)__popcll(b);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
int launch_tile_binning(const GeomView& g, const SortBufs& bv, const TileBinView& tb, int P, int64_t capacity, int grid_x, int grid_y,
                        int tile_cull, const float* tile_depth_limit, uint2* ranges, int force_rowwise, hipStream_t s, int debug) {
  const int rg_x = (grid_x + RG_TILES - 1) / RG_TILES, rg_y = (grid_y + RG_TILES - 1) / RG_TILES;
  const int NR = rg_x * rg_y, T = grid_x * grid_y;
  if (NR > 65535) return GS_E_UNSUPPORTED;  // (the region id shares a 32-bit word with the 16-bit tile mask, and the all-ones word marks an empty staging slot: images beyond 16 k x 16 k)
  const uint32_t cap32 = (uint32_t)capacity;
  const int nb = (P + GS_BLOCK - 1) / GS_BLOCK;
  const float* limit = tile_cull ? tile_depth_limit : nullptr;
  uint32_t* order = g.gsort.vals[0];  // the depth order (launch_radix_sort, 4 passes: ends in half 0)
  uint32_t* cnt = g.gsort.keys[0];    // (the sorted depth keys are not read again: the entries of each Gaussian of the order)
  const int bits = (int)gs_higher_msb((uint32_t)NR);
  // up to 512 regions (1080p: 510): ONE pass with a 9-bit digit; more: 8-bit passes
  const int digit_bits = bits <= 8 ? 8 : (bits == 9 ? 9 : 8);
  const int passes = (bits + digit_bits - 1) / digit_bits;
  const int start = passes == 1 ? 0 : 1;  // the partition ends in half 1: half 0 of the ids is point_list
  int rc;
  {
    GS_PROF(ST_DUPLICATE, s);
    hipLaunchKernelGGL(tb_entries_kernel, dim3(nb), dim3(GS_BLOCK), 0, s, g, &g.hdr->n_ordered, order, cnt, (uint32_t)grid_x,
                       (uint32_t)grid_y, tile_cull, limit, force_rowwise, g.sorted_sums);
    rc = launch_scan_sums(g.sorted_sums, nb, s);
    if (rc) return rc;
    GS_LAUNCH_CHECK(s, debug);
    if (tile_cull)
      hipLaunchKernelGGL(tb_emit_kernel<true>, dim3(nb), dim3(GS_BLOCK), 0, s, g, &g.hdr->n_ordered, (uint32_t)grid_x, (uint32_t)grid_y,
                         (uint32_t)rg_x, tile_cull, limit, order, cnt, g.sorted_sums, cap32, bv.keys[start], bv.vals[start]);
    else
      hipLaunchKernelGGL(tb_emit_kernel<false>, dim3(nb), dim3(GS_BLOCK), 0, s, g, &g.hdr->n_ordered, (uint32_t)grid_x, (uint32_t)grid_y,
                         (uint32_t)rg_x, 0, limit, order, cnt, g.sorted_sums, cap32, bv.keys[start], bv.vals[start]);
    GS_LAUNCH_CHECK(s, debug);
  }
  const uint32_t* n_entries = g.sorted_sums + nb;
  {
    GS_PROF(ST_SORT, s);
    rc = launch_radix_sort(bv, n_entries, capacity, bits, start, nullptr, s, debug, nullptr, digit_bits);
    if (rc) return rc;
  }
  {
    GS_PROF(ST_RANGES, s);
    if (passes == 1)   // the pass's digit totals ARE the regions' entry counts
      hipLaunchKernelGGL(tb_regions_from_counts_kernel, dim3(1), dim3(1024), 0, s, g.hdr, bv.scan_tmp, n_entries, cap32, NR, tb);
    else
      hipLaunchKernelGGL(tb_regions_kernel, dim3(1), dim3(1024), 0, s, g.hdr, bv.keys[1], n_entries, cap32, NR, tb);
    GS_LAUNCH_CHECK(s, debug);
    const uint32_t pool = tb.max_chunks / 4u + 1u < 4096u ? tb.max_chunks / 4u + 1u : 4096u;   // (four chunks per workgroup at a time)
    hipLaunchKernelGGL(tb_tile_count_kernel, dim3(pool), dim3(GS_BLOCK), 0, s, bv.keys[1], NR, tb);
    hipLaunchKernelGGL(tb_region_scan_kernel, dim3(NR), dim3(64), 0, s, rg_x, grid_x, grid_y, tb);
    GS_LAUNCH_CHECK(s, debug);
    hipLaunchKernelGGL(tb_tile_scan_kernel, dim3(1), dim3(1024), 0, s, g.hdr, T, cap32, tb, ranges);
    hipLaunchKernelGGL(tb_write_kernel, dim3(pool), dim3(GS_BLOCK), 0, s, g.hdr, bv.keys[1], bv.vals[1], NR, rg_x, grid_x,
                       grid_y, cap32, tb, bv.vals[0]);
    GS_LAUNCH_CHECK(s, debug);
  }
  return 0;
}
