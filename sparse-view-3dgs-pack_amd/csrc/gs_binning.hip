// gs_binning.hip - tile binning: key duplication, stable LSD radix sort, tile ranges.
//
// Replaces duplicateWithKeys (rasterizer_impl.cu:70-111), cub::DeviceRadixSort::SortPairs on the low
// 32+ceil_log2(T) key bits (rasterizer_impl.cu:306-311) and identifyTileRanges (:116-138).
// Integer/byte work, HBM-bound: everything here must reproduce the reference ordering exactly
// (stable sort: equal (tile, depth-bits) keys stay in ascending Gaussian index).
//
// All kernels read the instance count R from device memory (GeomHeader.num_rendered) so that the
// host never has to wait for it; grids are sized from a host-side upper bound.
#include "gs_common.h"

// ------------------------------------------------------------------------------------------------
// duplicate: one workgroup per 256 Gaussians.  A workgroup-local exclusive scan of tiles_touched
// gives every Gaussian its slot range; then ALL 256 lanes walk the workgroup's instance range
// cooperatively (binary search of the owning Gaussian in LDS), so stores are fully coalesced and a
// Gaussian covering thousands of tiles does not serialise one lane (the reference loops one thread
// over all of a Gaussian's tiles).
// ------------------------------------------------------------------------------------------------
__global__ void bin_prepare_kernel(GeomHeader* hdr, uint32_t capacity) {
  const uint32_t R = hdr->num_rendered;
  const bool ovf = R > capacity;
  hdr->overflow = ovf ? 1u : 0u;
  hdr->sort_n = ovf ? 0u : R;
}

__global__ void __launch_bounds__(GS_BLOCK) duplicate_kernel(GeomView g, int P, uint32_t grid_x, uint64_t* keys,
                                                             uint32_t* vals) {
  __shared__ uint32_t s_off[GS_BLOCK + 1];  // exclusive local offsets
  __shared__ uint32_t s_wsum[GS_BLOCK / 64];
  __shared__ uint32_t s_depth[GS_BLOCK];
  __shared__ uint32_t s_rmin[GS_BLOCK];
  __shared__ uint32_t s_w[GS_BLOCK];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int idx = blockIdx.x * GS_BLOCK + tid;
  uint32_t tiles = 0;
  if (idx < P) {
    const Splat* sp = &g.splat[idx];
    // one 16-B load covers rect_min, rect_max, tiles, clamped
    const uint4 tail = reinterpret_cast<const uint4*>(sp)[3];
    tiles = tail.z;
    s_rmin[tid] = tail.x;
    s_w[tid] = (tail.y & 0xFFFFu) - (tail.x & 0xFFFFu);
    s_depth[tid] = __float_as_uint(sp->depth);
  }
  uint32_t inc = tiles;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) s_wsum[wid] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wid; w++) woff += s_wsum[w];
  const uint32_t base = g.block_sums[blockIdx.x];
  s_off[tid] = woff + inc - tiles;
  if (tid == GS_BLOCK - 1) s_off[GS_BLOCK] = woff + inc;
  if (idx < P) g.point_offsets[idx] = base + woff + inc;  // inclusive scan, as the reference stores it
  __syncthreads();
  const uint32_t total = s_off[GS_BLOCK];
  if (total == 0 || g.hdr->overflow) return;
  for (uint32_t i = tid; i < total; i += GS_BLOCK) {
    // largest gi with s_off[gi] <= i
    int lo = 0, hi = GS_BLOCK - 1;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      int mid = (lo + hi + 1) >> 1;
      if (s_off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const uint32_t k = i - s_off[lo];
    const uint32_t w = s_w[lo];
    const uint32_t rmin = s_rmin[lo];
    const uint32_t ty = (rmin >> 16) + k / w;
    const uint32_t tx = (rmin & 0xFFFFu) + k % w;
    uint64_t key = (uint64_t)(ty * grid_x + tx);
    key <<= 32;
    key |= s_depth[lo];
    keys[base + i] = key;
    vals[base + i] = blockIdx.x * GS_BLOCK + lo;
  }
}

// ------------------------------------------------------------------------------------------------
// LSD radix sort, 8-bit digits, stable.  Per pass: (a) per-workgroup digit histograms
// (digit-major table), (b) exclusive scan of the table, (c) stable scatter.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GS_BLOCK) rs_hist_kernel(const uint64_t* __restrict__ keys, const uint32_t* n_dev,
                                                           int shift, uint32_t* __restrict__ hist, uint32_t nblk) {
  __shared__ uint32_t h[RS_RADIX];
  const uint32_t n = *n_dev;
  const uint32_t t0 = blockIdx.x * RS_TILE;
  h[threadIdx.x] = 0;
  __syncthreads();
  if (t0 < n) {
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
      uint32_t i = t0 + r * GS_BLOCK + threadIdx.x;
      if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 0xFFu], 1u);
    }
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// generic exclusive scan of u32 data[n] (n known on host): reduce / scan-of-sums / downsweep
__global__ void __launch_bounds__(GS_BLOCK) scan_reduce_kernel(const uint32_t* __restrict__ data, uint32_t n,
                                                               uint32_t* __restrict__ sums) {
  __shared__ uint32_t red[GS_BLOCK / 64];
  const uint32_t t0 = blockIdx.x * RS_TILE;
  uint32_t v = 0;
  const uint32_t i0 = t0 + threadIdx.x * RS_ITEMS;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++)
    if (i0 + r < n) v += data[i0 + r];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
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
}
__global__ void __launch_bounds__(GS_BLOCK) scan_down_kernel(uint32_t* __restrict__ data, uint32_t n,
                                                             const uint32_t* __restrict__ sums) {
  __shared__ uint32_t wsum[GS_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t i0 = blockIdx.x * RS_TILE + tid * RS_ITEMS;
  uint32_t v[RS_ITEMS];
  uint32_t tsum = 0;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    v[r] = (i0 + r < n) ? data[i0 + r] : 0u;
    tsum += v[r];
  }
  uint32_t inc = tsum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t run = sums[blockIdx.x] + inc - tsum;
  for (int w = 0; w < wid; w++) run += wsum[w];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    if (i0 + r < n) data[i0 + r] = run;
    run += v[r];
  }
}

// stable scatter of one 4096-key tile: 16 rounds of 256 keys in index order; inside a round the
// rank among equal digits = (same-digit lanes below me in my wave) + (same-digit counts of lower waves)
__global__ void __launch_bounds__(GS_BLOCK) rs_scatter_kernel(const uint64_t* __restrict__ kin,
                                                              const uint32_t* __restrict__ vin,
                                                              uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                              const uint32_t* n_dev, int shift,
                                                              const uint32_t* __restrict__ hist, uint32_t nblk) {
  __shared__ uint32_t s_base[RS_RADIX];          // running output position per digit
  __shared__ uint32_t s_cnt[GS_BLOCK / 64][RS_RADIX];  // per-wave digit counts of the current round
  const uint32_t n = *n_dev;
  const uint32_t t0 = blockIdx.x * RS_TILE;
  if (t0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  s_base[tid] = hist[(size_t)tid * nblk + blockIdx.x];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < RS_ITEMS; r++) {
    const uint32_t i = t0 + r * GS_BLOCK + tid;
    const bool valid = i < n;
    uint64_t key = 0;
    uint32_t val = 0;
    if (valid) {
      key = kin[i];
      val = vin[i];
    }
    const uint32_t d = valid ? ((uint32_t)(key >> shift) & 0xFFu) : 0u;
#pragma unroll
    for (int w = 0; w < GS_BLOCK / 64; w++) s_cnt[w][tid] = 0;
    __syncthreads();
    // lanes of this wave holding the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RS_BITS; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t rank_in_wave = __popcll(peers & lt_mask);
    if (valid && rank_in_wave == 0) s_cnt[wid][d] = __popcll(peers);
    __syncthreads();
    if (valid) {
      uint32_t pos = s_base[d] + rank_in_wave;
      for (int w = 0; w < wid; w++) pos += s_cnt[w][d];
      kout[pos] = key;
      vout[pos] = val;
    }
    __syncthreads();
    s_base[tid] += s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
    // (the zeroing of s_cnt at the top of the next round is ordered by the barrier that follows it)
    __syncthreads();
  }
}

// rasterizer_impl.cu:116-138
__global__ void __launch_bounds__(GS_BLOCK) tile_ranges_kernel(const uint64_t* __restrict__ keys, const uint32_t* n_dev,
                                                               uint2* __restrict__ ranges) {
  const uint32_t L = *n_dev;
  const uint32_t idx = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (idx >= L) return;
  const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
  if (idx == 0)
    ranges[currtile].x = 0;
  else {
    const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
    if (currtile != prevtile) {
      ranges[prevtile].y = idx;
      ranges[currtile].x = idx;
    }
  }
  if (idx == L - 1) ranges[currtile].y = L;
}

// ------------------------------------------------------------------------------------------------
int launch_bin_prepare(const GeomView& g, int64_t capacity, hipStream_t s) {
  uint32_t cap32 = capacity > 0xFFFFFFFFll ? 0xFFFFFFFFu : (capacity < 0 ? 0u : (uint32_t)capacity);
  hipLaunchKernelGGL(bin_prepare_kernel, dim3(1), dim3(1), 0, s, g.hdr, cap32);
  return 0;
}
int launch_duplicate(const GeomView& g, int P, int grid_x, const BinView& b, int buf, hipStream_t s) {
  const int nb = (P + GS_BLOCK - 1) / GS_BLOCK;
  hipLaunchKernelGGL(duplicate_kernel, dim3(nb), dim3(GS_BLOCK), 0, s, g, P, (uint32_t)grid_x, b.keys[buf], b.vals[buf]);
  return 0;
}

static int exclusive_scan_u32(uint32_t* data, uint32_t n, uint32_t* tmp, hipStream_t s) {
  const uint32_t nb = (n + RS_TILE - 1) / RS_TILE;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(GS_BLOCK), 0, s, data, n, tmp);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, tmp, (int)nb);
  hipLaunchKernelGGL(scan_down_kernel, dim3(nb), dim3(GS_BLOCK), 0, s, data, n, tmp);
  return 0;
}

int launch_radix_sort(const BinView& b, const uint32_t* n_dev, int64_t n_bound, int end_bit, int start_buf,
                      hipStream_t s, int debug) {
  int cur = start_buf;
  if (n_bound > 0) {
    const uint32_t nblk = (uint32_t)((n_bound + RS_TILE - 1) / RS_TILE);  // <= b.nblk
    const uint32_t hist_n = nblk * RS_RADIX;
    for (int shift = 0; shift < end_bit; shift += RS_BITS) {
      hipLaunchKernelGGL(rs_hist_kernel, dim3(nblk), dim3(GS_BLOCK), 0, s, b.keys[cur], n_dev, shift, b.hist, nblk);
      GS_LAUNCH_CHECK(s, debug);
      exclusive_scan_u32(b.hist, hist_n, b.scan_tmp, s);
      GS_LAUNCH_CHECK(s, debug);
      hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblk), dim3(GS_BLOCK), 0, s, b.keys[cur], b.vals[cur],
                         b.keys[cur ^ 1], b.vals[cur ^ 1], n_dev, shift, b.hist, nblk);
      GS_LAUNCH_CHECK(s, debug);
      cur ^= 1;
    }
  }
  return 0;
}

int launch_tile_ranges(const uint64_t* keys, const uint32_t* n_dev, int64_t n_bound, uint2* ranges, int T,
                       hipStream_t s) {
  hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)T, s);
  if (e != hipSuccess) return (int)e;
  if (n_bound > 0)
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((uint32_t)((n_bound + GS_BLOCK - 1) / GS_BLOCK)), dim3(GS_BLOCK), 0, s,
                       keys, n_dev, ranges);
  return 0;
}
