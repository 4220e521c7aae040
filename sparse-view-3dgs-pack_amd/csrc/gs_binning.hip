// gs_binning.hip - tile binning: depth order of the Gaussians, key duplication, stable tile partition,
// tile ranges.  Integer/byte work, HBM-bound.
//
// Replaces duplicateWithKeys (rasterizer_impl.cu:70-111), cub::DeviceRadixSort::SortPairs over R
// 64-bit (tile|depth) keys (rasterizer_impl.cu:306-311) and identifyTileRanges (:116-138), and produces
// EXACTLY the reference's point_list (stable order: tile, then depth bits, then Gaussian index).
//
// MI355X-first restructuring.  The reference sorts R = sum(tiles_touched) 12-byte pairs on 45 key bits
// (6 radix passes, ~24 B x R each).  Here the two key fields are separated:
//   1. the P Gaussians (not the R instances; R/P ~ 20) are stably sorted by their 32 depth bits
//      (culled ones get key 0xFFFFFFFF) - 4 passes over 8 B x P;
//   2. instances are emitted in that depth order (key = tile id only, 4 B);
//   3. a STABLE partition by tile id (ceil(log2 T)/8 = 2 passes over 8 B x R at 1080p) then leaves every
//      tile's list in (depth, index) order - the same total order as the 64-bit sort.
// The instance-level traffic drops from 6 x 24 B to 2 x 16 B per instance.
//
// duplicate: one workgroup per 256 (depth-ordered) Gaussians; ALL lanes walk the workgroup's instance
// range cooperatively (binary search of the owner in LDS): coalesced stores, no lane serialised on a
// Gaussian that covers thousands of tiles (the reference loops one thread over a Gaussian's tiles).
//
// All kernels read element counts from device memory so the host never waits for them; grids are sized
// from a host-side upper bound.
#include "gs_common.h"
#include "gs_tilecull.h"

// header of the binning phase + the zeroed tile ranges (tiles no instance reaches stay (0,0)) in one launch
__global__ void __launch_bounds__(GS_BLOCK) bin_prepare_kernel(GeomHeader* hdr, uint32_t capacity, uint2* __restrict__ ranges,
                                                               int T) {
  const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (i < T) ranges[i] = make_uint2(0u, 0u);
  if (i == 0) {
    const uint32_t R = hdr->num_rendered;
    const bool ovf = R > capacity;
    hdr->overflow = ovf ? 1u : 0u;
    hdr->sort_n = ovf ? 0u : R;
    hdr->trunc_failed = 0u;
    hdr->zero = 0u;
  }
}

// ------------------------------------------------------------------------------------------------
// stable LSD radix sort of (u32 key, u32 value) pairs, 8-bit digits.
// Per pass: (a) per-workgroup digit histograms (digit-major table), (b) exclusive scan, (c) scatter.
// A workgroup owns a 4096-key tile; wave w owns the contiguous quarter [w*1024, (w+1)*1024).
// ------------------------------------------------------------------------------------------------
// 1024 threads per 4096-key tile: a wave counts 256 keys in 4 rounds.  The kernel is latency-bound (one workgroup's
// chain: global load -> LDS atomics, serialised where lanes share a digit -> barrier -> store), so the chain is made
// short rather than the workgroup small: with 256 threads (16 rounds per wave) a pass took 18 us at 9.8 M keys.
#define RS_HIST_THREADS 1024
// drop_max (first pass of the depth sort): keys 0xFFFFFFFF - Gaussians that emit no instance: culled, or cut away entirely
// by the depth limits, 80 % of them at the bench workload - are not counted and not scattered: the pass filters while it
// sorts, and the later passes (and everything downstream of the order) handle the survivors only.
__global__ void __launch_bounds__(RS_HIST_THREADS) rs_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* n_dev,
                                                                  int shift, uint32_t* __restrict__ hist, uint32_t nblk,
                                                                  int drop_max) {
  __shared__ uint32_t h[RS_HIST_THREADS / 64][RS_RADIX];  // one private histogram per wave
  const uint32_t n = *n_dev;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = RS_HIST_THREADS / 64, PER_WAVE = RS_TILE / NW, ROUNDS = PER_WAVE / 64;
#pragma unroll
  for (int k = tid; k < NW * RS_RADIX; k += RS_HIST_THREADS) (&h[0][0])[k] = 0;
  __syncthreads();
  const uint32_t w0 = blockIdx.x * RS_TILE + wid * PER_WAVE;
  if (w0 < n) {
    uint32_t k[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
      const uint32_t i = w0 + r * 64 + lane;
      k[r] = i < n ? keys[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
      const uint32_t i = w0 + r * 64 + lane;
      if (i < n && !(drop_max && k[r] == 0xFFFFFFFFu)) atomicAdd(&h[wid][(k[r] >> shift) & 0xFFu], 1u);
    }
  }
  __syncthreads();
  if (tid < RS_RADIX) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) t += h[w][tid];
    hist[(size_t)tid * nblk + blockIdx.x] = t;
  }
}

// in-place exclusive scan of sums[nb] by ONE workgroup; total -> sums[nb]
__global__ void __launch_bounds__(1024) scan_sums_kernel(uint32_t* sums, int nb) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    int i = base + tid;
    uint32_t v = (i < nb) ? sums[i] : 0u;
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
    if (i < nb) sums[i] = carry + woff + inc - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) sums[nb] = carry_s;
}
// (b) of a radix pass in ONE launch: workgroup d scans row d of the digit-major table in place (exclusive, over
// the workgroups of the sort) and stores the row total; the scatter kernel adds the exclusive scan of the 256
// totals itself.  (The generic three-kernel scan of the whole table cost two more launches per pass.)
__global__ void __launch_bounds__(GS_BLOCK) rs_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t nblk,
                                                              uint32_t* __restrict__ totals) {
  constexpr int IT = 8;
  __shared__ uint32_t wsum[GS_BLOCK / 64];
  uint32_t* row = hist + (size_t)blockIdx.x * nblk;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nblk; base += GS_BLOCK * IT) {
    const uint32_t i0 = base + tid * IT;
    uint32_t v[IT];
    uint32_t tsum = 0;
#pragma unroll
    for (int r = 0; r < IT; r++) {
      v[r] = (i0 + r < nblk) ? row[i0 + r] : 0u;
      tsum += v[r];
    }
    uint32_t inc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t run = carry + inc - tsum;
    for (int w = 0; w < wid; w++) run += wsum[w];
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
#pragma unroll
    for (int r = 0; r < IT; r++) {
      if (i0 + r < nblk) row[i0 + r] = run;
      run += v[r];
    }
    __syncthreads();
  }
  if (tid == 0) totals[blockIdx.x] = carry;
}

// Stable scatter.  Phase 1: every wave ranks its RS_TILE / NW keys in rounds of 64 with wave-level
// match-any (8 ballots per round; no barrier, LDS traffic stays inside the wave) and builds its private
// digit histogram.  Phase 2: 256 threads turn the NW wave histograms into per-wave offsets inside the tile.
// Phase 3: keys/values are exchanged through LDS into digit order and stored as contiguous runs.
#define RS_SCATTER_THREADS 512
template <int NT>
__global__ void __launch_bounds__(NT) rs_scatter_kernel(const uint32_t* __restrict__ kin,
                                                        const uint32_t* __restrict__ vin,  // NULL: value = index
                                                        uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                        const uint32_t* n_dev, int shift,
                                                        const uint32_t* __restrict__ hist, uint32_t nblk,
                                                        const uint32_t* __restrict__ totals, int drop_max,
                                                        uint32_t* __restrict__ n_kept) {
  constexpr int NW = NT / 64;               // waves; wave w ranks the contiguous RS_TILE / NW keys [w0, w0 + ...)
  constexpr int ITEMS = RS_TILE / NT;       // keys per thread = ranking rounds per wave
  static_assert(NT >= RS_RADIX && RS_TILE % NT == 0, "one thread per digit in phase 2");
  __shared__ uint32_t s_hist[NW][RS_RADIX];  // phase 1: wave digit counts; phase 3: output bases
  const uint32_t n = *n_dev;
  const uint32_t t0 = blockIdx.x * RS_TILE;
  if (t0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#pragma unroll
  for (int k = tid; k < NW * RS_RADIX; k += NT) (&s_hist[0][0])[k] = 0;
  __syncthreads();
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint32_t w0 = t0 + wid * (RS_TILE / NW);
  uint32_t key[ITEMS], val[ITEMS], pre[ITEMS];  // pre = same-digit keys of my wave before me
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = w0 + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? kin[i] : 0xFFFFFFFFu;
    val[r] = valid ? (vin ? vin[i] : i) : 0u;
  }
  __shared__ uint32_t s_cnt;  // keys of this tile that take part (all of them unless drop_max)
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = w0 + r * 64 + lane;
    const bool valid = i < n && !(drop_max && key[r] == 0xFFFFFFFFu);
    const uint32_t d = (key[r] >> shift) & 0xFFu;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RS_BITS; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(peers & lt_mask);
    // same wave, program order: the read sees the previous rounds' updates, and every lane of the wave has
    // read before the leader's store below is issued (LDS ops of one wave execute in order)
    const uint32_t before = s_hist[wid][d];
    pre[r] = valid ? before + rank : 0xFFFFFFFFu;
    if (valid && rank == 0) s_hist[wid][d] = before + (uint32_t)__popcll(peers);
  }
  __syncthreads();
  // Phase 2: thread d < 256 owns digit d: wave prefixes, workgroup-local exclusive digit offset (scan over the 256
  // digit totals) and the distance from the local to the global position of that digit's run.
  __shared__ uint32_t s_delta[RS_RADIX];  // global start - local start
  __shared__ uint32_t s_wtot[RS_RADIX / 64];
  __shared__ uint32_t s_gtot[RS_RADIX / 64];
  const bool digit_thread = tid < RS_RADIX;
  uint32_t cw[NW];
  uint32_t tot = 0, gt = 0;
  if (digit_thread) {
#pragma unroll
    for (int w = 0; w < NW; w++) {
      cw[w] = s_hist[w][tid];
      tot += cw[w];
    }
    gt = totals[tid];  // keys of digit `tid` in the whole array -> start of the digit = scan over digits
  }
  uint32_t inc = tot, ginc = gt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    const uint32_t g = __shfl_up(ginc, off, 64);
    if (lane >= off) {
      inc += t;
      ginc += g;
    }
  }
  if (digit_thread && lane == 63) {
    s_wtot[wid] = inc;
    s_gtot[wid] = ginc;
  }
  __syncthreads();
  if (digit_thread) {
    uint32_t loc = inc - tot, gbase = ginc - gt;
    for (int w = 0; w < wid; w++) {
      loc += s_wtot[w];
      gbase += s_gtot[w];
    }
    s_delta[tid] = gbase + hist[(size_t)tid * nblk + blockIdx.x] - loc;
    if (tid == RS_RADIX - 1) {
      s_cnt = loc + tot;                                     // end of the last digit's run = keys that take part
      if (n_kept && blockIdx.x == 0) *n_kept = gbase + gt;   // ... and in the whole array
    }
    uint32_t run = loc;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      s_hist[w][tid] = run;
      run += cw[w];
    }
  }
  __syncthreads();
  // Phase 3: exchange through LDS so that every digit's keys are contiguous, then store: consecutive lanes write
  // consecutive global addresses inside a digit run (a direct scatter writes 64 isolated 4-byte words per
  // wave instruction).
  __shared__ uint32_t s_key[RS_TILE];
  __shared__ uint32_t s_val[RS_TILE];
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    if (pre[r] != 0xFFFFFFFFu) {
      const uint32_t lp = s_hist[wid][(key[r] >> shift) & 0xFFu] + pre[r];
      s_key[lp] = key[r];
      s_val[lp] = val[r];
    }
  }
  __syncthreads();
  const uint32_t cnt = s_cnt;
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = r * NT + tid;
    if (i < cnt) {
      const uint32_t k = s_key[i];
      const uint32_t pos = i + s_delta[(k >> shift) & 0xFFu];
      kout[pos] = k;
      vout[pos] = s_val[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// instance emission in depth order
// ------------------------------------------------------------------------------------------------
// per-workgroup sums of tiles_touched in depth order
__global__ void __launch_bounds__(GS_BLOCK) sorted_block_sums_kernel(const uint32_t* __restrict__ order,
                                                                     const uint32_t* __restrict__ tiles_touched,
                                                                     const uint32_t* __restrict__ n_ordered,
                                                                     uint32_t* __restrict__ sums) {
  __shared__ uint32_t red[GS_BLOCK / 64];
  const uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x;
  uint32_t v = (i < *n_ordered) ? tiles_touched[order[i]] : 0u;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Instance emission (duplicateWithKeys, rasterizer_impl.cu:70-111) in depth order.  The unit of work is one
// TILE ROW of one Gaussian: a workgroup takes 256 consecutive Gaussians of the depth order, expands them into
// their rectangle rows (binary search over the row-count prefix in LDS), evaluates each row's column span once
// (whole row, or the ellipse span of gs_tilecull.h), scans the span lengths and then writes the instances with
// one binary search per instance - coalesced stores, no lane serialised on a Gaussian covering thousands of
// tiles, and no per-instance re-evaluation of the spans.  Rows are processed DUP_RC at a time to bound LDS.
#define DUP_RC 1024
__device__ __forceinline__ uint32_t block_exclusive_scan256(uint32_t v, uint32_t* s_wsum, uint32_t& total) {
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

__global__ void __launch_bounds__(GS_BLOCK) duplicate_kernel(GeomView g, const uint32_t* __restrict__ n_ordered, uint32_t grid_x,
                                                             uint32_t grid_y, int tile_cull,
                                                             const float* __restrict__ depth_limit,
                                                             const uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ block_base,
                                                             uint32_t* __restrict__ tkeys, uint32_t* __restrict__ tvals) {
  __shared__ uint32_t s_wsum[GS_BLOCK / 64];
  __shared__ uint32_t s_rowoff[GS_BLOCK + 1];  // exclusive prefix of rows per Gaussian
  __shared__ uint32_t s_id[GS_BLOCK];
  __shared__ uint32_t s_rmin[GS_BLOCK];
  __shared__ uint32_t s_rmax[GS_BLOCK];
  __shared__ TileCull s_cull[GS_BLOCK];
  __shared__ float s_depth[GS_BLOCK];
  __shared__ int s_dq[GS_BLOCK];
  __shared__ uint32_t s_span_off[DUP_RC + 1];  // exclusive prefix of span lengths within the row chunk
  __shared__ uint32_t s_span_key[DUP_RC];      // first tile id of the span
  __shared__ uint32_t s_span_own[DUP_RC];      // owner (index into s_id)
  if (g.hdr->overflow) return;
  const int tid = threadIdx.x;
  const uint32_t i = blockIdx.x * GS_BLOCK + tid;
  uint32_t tiles = 0, rows = 0;
  s_dq[tid] = 0;  // (only Gaussians with culled spans AND depth limits carry a verdict; everything else: spans as they are)
  if (i < *n_ordered) {
    const uint32_t id = order[i];
    // (the count comes from tiles_touched: a Gaussian the depth limits removed entirely has no record at all, and the
    // one-workgroup sort of small scenes keeps such Gaussians in the order)
    tiles = g.tiles_touched[id];
    const float4* rec = reinterpret_cast<const float4*>(&g.splat[id]);
    uint4 tail = make_uint4(0u, 0u, 0u, 0u);
    if (tiles) tail = reinterpret_cast<const uint4*>(rec)[3];  // rect_min, rect_max, tiles, clamped
    s_id[tid] = id;
    s_rmin[tid] = tail.x;
    s_rmax[tid] = tail.y;
    if (tiles) {
      rows = (tail.y >> 16) - (tail.x >> 16);
      if (tile_cull) {
        const float4 ra = rec[0], rc = rec[1];
        s_cull[tid] = tilecull_setup(1, ra.x, ra.y, rc.x, rc.y, rc.z, rc.w);
        s_depth[tid] = ra.z;
        s_dq[tid] = depth_limit ? (int)((tail.w >> 8) & 3u) : 0;  // the preprocess kernel's verdict (gs_tilecull.h)
      } else {
        s_cull[tid].mode = 0;
      }
    }
  }
  uint32_t total_rows, expected;
  const uint32_t my_rowoff = block_exclusive_scan256(rows, s_wsum, total_rows);
  s_rowoff[tid] = my_rowoff;
  if (tid == 0) s_rowoff[GS_BLOCK] = total_rows;
  (void)block_exclusive_scan256(tiles, s_wsum, expected);  // what the prefix sum reserved for this workgroup
  uint32_t written = 0;                                    // instances emitted so far (uniform)
  const uint32_t base = block_base[blockIdx.x];
  uint32_t last_key = 0, last_own = 0;
  for (uint32_t rbase = 0; rbase < total_rows; rbase += DUP_RC) {
    const uint32_t nrow = min((uint32_t)DUP_RC, total_rows - rbase);
    // each thread evaluates DUP_RC / 256 consecutive rows
    uint32_t n4[DUP_RC / GS_BLOCK], local = 0;
#pragma unroll
    for (int j = 0; j < DUP_RC / GS_BLOCK; j++) {
      const uint32_t r = tid * (DUP_RC / GS_BLOCK) + j;
      uint32_t n = 0;
      if (r < nrow) {
        const uint32_t rs = rbase + r;
        int lo = 0, hi = GS_BLOCK - 1;  // largest owner with s_rowoff[owner] <= rs
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const int mid = (lo + hi + 1) >> 1;
          if (s_rowoff[mid] <= rs) lo = mid; else hi = mid - 1;
        }
        const uint32_t rmin = s_rmin[lo], rmax = s_rmax[lo];
        const uint32_t ty = (rmin >> 16) + (rs - s_rowoff[lo]);
        uint32_t tx0;
        n = tilecull_row_span(s_cull[lo], ty, rmin & 0xFFFFu, rmax & 0xFFFFu, tx0);
        const int dq = depth_limit ? s_dq[lo] : 0;
        if (dq == 2) n = tilecull_trim_span(depth_limit, grid_x, ty, s_depth[lo], tx0, n);
        if (dq == 3) n = tilecull_trim_span_segments(depth_limit + (size_t)grid_x * grid_y, grid_x, ty, s_depth[lo], tx0, n);
        s_span_key[r] = ty * grid_x + tx0;
        s_span_own[r] = (uint32_t)lo;
      }
      n4[j] = n;
      local += n;
    }
    uint32_t chunk_total;
    uint32_t off = block_exclusive_scan256(local, s_wsum, chunk_total);
#pragma unroll
    for (int j = 0; j < DUP_RC / GS_BLOCK; j++) {
      const uint32_t r = tid * (DUP_RC / GS_BLOCK) + j;
      if (r < nrow) s_span_off[r] = off;
      off += n4[j];
    }
    if (tid == 0) s_span_off[nrow] = chunk_total;
    __syncthreads();
    // never write past what the prefix sum reserved (the two evaluations of the spans agree; this keeps the
    // kernel memory-safe even if they did not)
    const uint32_t room = expected - min(expected, written);
    const uint32_t emit = min(chunk_total, room);
    for (uint32_t k = tid; k < emit; k += GS_BLOCK) {
      int lo = 0, hi = (int)nrow - 1;  // largest row with s_span_off[row] <= k
#pragma unroll
      for (int it = 0; it < 10; it++) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_span_off[mid] <= k) lo = mid; else hi = mid - 1;
      }
      tkeys[base + written + k] = s_span_key[lo] + (k - s_span_off[lo]);
      tvals[base + written + k] = s_id[s_span_own[lo]];
    }
    if (nrow > 0) {
      last_key = s_span_key[nrow - 1];
      last_own = s_span_own[nrow - 1];
    }
    written += emit;
    __syncthreads();  // the chunk arrays are rewritten by the next iteration
  }
  // (unreachable when both evaluations agree) fill what is left with a valid instance of this workgroup
  for (uint32_t k = written + tid; k < expected; k += GS_BLOCK) {
    tkeys[base + k] = last_key;
    tvals[base + k] = s_id[last_own];
  }
}

// rasterizer_impl.cu:116-138 (keys hold the tile id only)
__global__ void __launch_bounds__(GS_BLOCK) tile_ranges_kernel(const uint32_t* __restrict__ tkeys, const uint32_t* n_dev,
                                                               uint2* __restrict__ ranges) {
  // four keys per thread (one 16-byte load + the key in front of them)
  const uint32_t L = *n_dev;
  const uint32_t i0 = (blockIdx.x * GS_BLOCK + threadIdx.x) * 4u;
  if (i0 >= L) return;
  uint32_t k[4];
  if (i0 + 3 < L) {
    const uint4 q = reinterpret_cast<const uint4*>(tkeys)[i0 >> 2];
    k[0] = q.x; k[1] = q.y; k[2] = q.z; k[3] = q.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; e++) k[e] = (i0 + e < L) ? tkeys[i0 + e] : 0u;
  }
  uint32_t prev = i0 ? tkeys[i0 - 1] : 0u;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const uint32_t idx = i0 + e;
    if (idx >= L) break;
    const uint32_t cur = k[e];
    if (idx == 0) {
      ranges[cur].x = 0;
    } else if (cur != prev) {
      ranges[prev].y = idx;
      ranges[cur].x = idx;
    }
    if (idx == L - 1) ranges[cur].y = L;
    prev = cur;
  }
}

// ------------------------------------------------------------------------------------------------
int launch_bin_prepare(const GeomView& g, int64_t capacity, uint2* ranges, int T, hipStream_t s) {
  uint32_t cap32 = capacity > 0xFFFFFFFFll ? 0xFFFFFFFFu : (capacity < 0 ? 0u : (uint32_t)capacity);
  hipLaunchKernelGGL(bin_prepare_kernel, dim3((uint32_t)((T + GS_BLOCK - 1) / GS_BLOCK > 0 ? (T + GS_BLOCK - 1) / GS_BLOCK : 1)),
                     dim3(GS_BLOCK), 0, s, g.hdr, cap32, ranges, T);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Small inputs (the depth order of <= 16 384 Gaussians: BASELINE config 1 has 10 000): the whole LSD sort in ONE workgroup.
// Twelve launches of ~4.8 us each are pure launch latency at that size (58 of the 515 us of a C1 step); here the 16 waves
// keep all keys in registers (16 per lane, wave w owns the contiguous slice [1024 w, 1024 w + 1024)), rank them per pass
// with the same ballot multi-split as rs_scatter_kernel and exchange them through LDS.  Same stable LSD passes, same
// result bit for bit.
// ------------------------------------------------------------------------------------------------------------------
#define RS_SMALL_ITEMS 16
#define RS_SMALL_MAX (1024 * RS_SMALL_ITEMS)
__global__ void __launch_bounds__(1024) rs_small_sort_kernel(const uint32_t* __restrict__ kin, const uint32_t* n_dev,
                                                             uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                             int end_bit, uint32_t* __restrict__ n_kept) {
  constexpr int NT = 1024, NW = NT / 64, ITEMS = RS_SMALL_ITEMS;
  __shared__ uint32_t s_hist[NW][RS_RADIX];
  __shared__ uint32_t s_key[RS_SMALL_MAX];
  __shared__ uint32_t s_val[RS_SMALL_MAX];
  __shared__ uint32_t s_wtot[RS_RADIX / 64];
  const uint32_t n = min(*n_dev, (uint32_t)RS_SMALL_MAX);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (n_kept && tid == 0) *n_kept = n;  // (nothing is dropped at this size: Gaussians without instances sort to the end)
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint32_t w0 = (uint32_t)wid * (64 * ITEMS);
  uint32_t key[ITEMS], val[ITEMS], pre[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = w0 + r * 64 + lane;
    key[r] = i < n ? kin[i] : 0xFFFFFFFFu;
    val[r] = i;
  }
  for (int shift = 0; shift < end_bit; shift += RS_BITS) {
    for (int k = tid; k < NW * RS_RADIX; k += NT) (&s_hist[0][0])[k] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      const bool valid = w0 + r * 64 + lane < n;
      const uint32_t d = (key[r] >> shift) & 0xFFu;
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < RS_BITS; b++) {
        const unsigned long long bal = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? bal : ~bal;
      }
      const uint32_t rank = __popcll(peers & lt_mask);
      const uint32_t before = s_hist[wid][d];  // (one wave, program order: see rs_scatter_kernel)
      pre[r] = before + rank;
      if (valid && rank == 0) s_hist[wid][d] = before + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    // thread d < 256 owns digit d: exclusive offset of the digit, then of each wave's share of it
    const bool digit_thread = tid < RS_RADIX;
    uint32_t cw[NW], tot = 0;
    if (digit_thread) {
#pragma unroll
      for (int w = 0; w < NW; w++) {
        cw[w] = s_hist[w][tid];
        tot += cw[w];
      }
    }
    uint32_t inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (digit_thread && lane == 63) s_wtot[wid] = inc;
    __syncthreads();
    if (digit_thread) {
      uint32_t run = inc - tot;
      for (int w = 0; w < wid; w++) run += s_wtot[w];
#pragma unroll
      for (int w = 0; w < NW; w++) {
        s_hist[w][tid] = run;
        run += cw[w];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      if (w0 + r * 64 + lane < n) {
        const uint32_t lp = s_hist[wid][(key[r] >> shift) & 0xFFu] + pre[r];
        s_key[lp] = key[r];
        s_val[lp] = val[r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      const uint32_t i = w0 + r * 64 + lane;
      key[r] = i < n ? s_key[i] : 0xFFFFFFFFu;
      val[r] = i < n ? s_val[i] : 0u;
    }
    // (the next pass starts by zeroing s_hist, last read before the barrier above; s_key / s_val are rewritten only
    // after two more barriers)
  }
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = w0 + r * 64 + lane;
    if (i < n) {
      kout[i] = key[r];
      vout[i] = val[r];
    }
  }
}

int launch_radix_sort(const SortBufs& b, const uint32_t* n_dev, int64_t n_bound, int end_bit, int start_buf,
                      const uint32_t* first_keys, hipStream_t s, int debug, uint32_t* n_kept) {
  int cur = start_buf;
  if (n_bound <= 0) return 0;
  if (first_keys != nullptr && n_bound <= RS_SMALL_MAX) {
    const int passes = (end_bit + RS_BITS - 1) / RS_BITS;
    const int fin = start_buf ^ (passes & 1);  // where the pass-by-pass form would leave the result
    hipLaunchKernelGGL(rs_small_sort_kernel, dim3(1), dim3(1024), 0, s, first_keys, n_dev, b.keys[fin], b.vals[fin], end_bit,
                       n_kept);
    GS_LAUNCH_CHECK(s, debug);
    return 0;
  }
  const uint32_t nblk = (uint32_t)((n_bound + RS_TILE - 1) / RS_TILE);
  bool first = true;
  for (int shift = 0; shift < end_bit; shift += RS_BITS) {
    const bool ext = first && first_keys != nullptr;
    const uint32_t* kin = ext ? first_keys : b.keys[cur];
    const uint32_t* vin = ext ? nullptr : b.vals[cur];
    const int drop = ext && n_kept ? 1 : 0;  // the first pass of the depth sort filters (see rs_hist_kernel)
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nblk), dim3(RS_HIST_THREADS), 0, s, kin, n_dev, shift, b.hist, nblk, drop);
    GS_LAUNCH_CHECK(s, debug);
    hipLaunchKernelGGL(rs_rowscan_kernel, dim3(RS_RADIX), dim3(GS_BLOCK), 0, s, b.hist, nblk, b.scan_tmp);
    GS_LAUNCH_CHECK(s, debug);
    // 512 threads per 4096-key tile (8 ranking rounds per wave): 0.213 ms for the two instance passes against 0.226 with
    // 256 threads and 0.217 with 1024 - the pass is bound by one workgroup's dependent chain, not by throughput
    hipLaunchKernelGGL(rs_scatter_kernel<RS_SCATTER_THREADS>, dim3(nblk), dim3(RS_SCATTER_THREADS), 0, s, kin, vin,
                       b.keys[cur ^ 1], b.vals[cur ^ 1], n_dev, shift, b.hist, nblk, b.scan_tmp, drop, drop ? n_kept : nullptr);
    GS_LAUNCH_CHECK(s, debug);
    if (drop) n_dev = n_kept;  // the later passes see the survivors only
    cur ^= 1;
    first = false;
  }
  return 0;
}

int launch_emit_instances(const GeomView& g, int P, const uint32_t* n_ordered, int grid_x, int grid_y, int tile_cull,
                          const float* tile_depth_limit, const uint32_t* order, uint32_t* tkeys, uint32_t* tvals, hipStream_t s,
                          int debug) {
  const int nb = (P + GS_BLOCK - 1) / GS_BLOCK;
  hipLaunchKernelGGL(sorted_block_sums_kernel, dim3(nb), dim3(GS_BLOCK), 0, s, order, g.tiles_touched, n_ordered, g.sorted_sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, g.sorted_sums, nb);
  GS_LAUNCH_CHECK(s, debug);
  hipLaunchKernelGGL(duplicate_kernel, dim3(nb), dim3(GS_BLOCK), 0, s, g, n_ordered, (uint32_t)grid_x, (uint32_t)grid_y,
                     tile_cull, tile_cull ? tile_depth_limit : nullptr, order, g.sorted_sums, tkeys, tvals);
  GS_LAUNCH_CHECK(s, debug);
  return 0;
}

int launch_tile_ranges(const uint32_t* tkeys, const uint32_t* n_dev, int64_t n_bound, uint2* ranges, int T,
                       hipStream_t s) {
  // `ranges` was zeroed by launch_bin_prepare at the start of the phase
  if (n_bound > 0)
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((uint32_t)((n_bound + 4 * GS_BLOCK - 1) / (4 * GS_BLOCK))), dim3(GS_BLOCK), 0, s,
                       tkeys, n_dev, ranges);
  return 0;
}
