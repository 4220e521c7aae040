// gs_binning.hip - the stable LSD radix sort of (u32 key, u32 value) pairs behind the tile binning of tile_cull = 0 / 1.
// Integer/byte work, HBM-bound.
//
// The reference sorts R = sum(tiles_touched) 12-byte pairs on 45 key bits (cub::DeviceRadixSort::SortPairs,
// rasterizer_impl.cu:306-311: 6 radix passes, ~24 B x R each).  Here the two key fields are separated:
//   1. the P Gaussians (not the R instances; R/P ~ 20) are stably sorted by their 32 depth bits
//      (Gaussians without instances dropped by the first pass) - 4 passes over 8 B x P: this file;
//   2. in that depth order every Gaussian emits one ENTRY per 4 x 4-tile region it reaches (region id + 16-bit tile mask),
//      the entries are stably partitioned by region id with the same passes (1-2 over 8 B x entries), and the regions are
//      expanded into the tile lists: gs_tilebin.hip.  Every tile's list comes out in (depth, index) order - the total order
//      of the 64-bit sort - and point_list / ranges are the reference's bit for bit.
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
// The table's row stride is the number of tiles the keys really fill - ceil(n / RS_TILE) with n read on the device - not the
// host's bound: the partition of the region entries is launched for the binning CAPACITY (instances) but sorts a fifth of that
// many entries, and a workgroup beyond the last tile used to write its 512 zero counts into a 4.5 x larger table (94 MB of
// scattered 4-byte writes per pass at C3, profiles/r05_binning_counters.csv).
template <int BITS>
__global__ void __launch_bounds__(RS_HIST_THREADS) rs_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* n_dev,
                                                                  int shift, uint32_t* __restrict__ hist, int drop_max) {
  constexpr int RADIX = 1 << BITS;
  __shared__ uint32_t h[RS_HIST_THREADS / 64][RADIX];  // one private histogram per wave
  const uint32_t n = *n_dev;
  const uint32_t nblk = (n + RS_TILE - 1) / RS_TILE;
  if (blockIdx.x >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  constexpr int NW = RS_HIST_THREADS / 64, PER_WAVE = RS_TILE / NW, ROUNDS = PER_WAVE / 64;
#pragma unroll
  for (int k = tid; k < NW * RADIX; k += RS_HIST_THREADS) (&h[0][0])[k] = 0;
  __syncthreads();
  const uint32_t w0 = blockIdx.x * RS_TILE + wid * PER_WAVE;
  if (w0 < n) {
    uint32_t k[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
      const uint32_t i = w0 + r * 64 + lane;
      k[r] = i < n ? keys[i] : 0u;
    }
    // the lanes of a wave that share a digit elect a leader that adds their number (ballot match, as the scatter kernel ranks):
    // with LDS atomics the high digits of a depth - a handful of values - put 64 lanes on one counter (70 % bank conflicts)
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
      const uint32_t i = w0 + r * 64 + lane;
      const bool valid = i < n && !(drop_max && k[r] == 0xFFFFFFFFu);
      const uint32_t d = (k[r] >> shift) & (uint32_t)(RADIX - 1);
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < BITS; b++) {
        const unsigned long long bal = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? bal : ~bal;
      }
      if (valid && (peers & lt_mask) == 0ull) h[wid][d] += (uint32_t)__popcll(peers);   // (one lane per digit: no atomic needed)
    }
  }
  __syncthreads();
  if (tid < RADIX) {
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
__global__ void __launch_bounds__(GS_BLOCK) rs_rowscan_kernel(uint32_t* __restrict__ hist, const uint32_t* __restrict__ n_dev,
                                                              uint32_t* __restrict__ totals) {
  constexpr int IT = 8;
  __shared__ uint32_t wsum[GS_BLOCK / 64];
  const uint32_t nblk = (*n_dev + RS_TILE - 1) / RS_TILE;
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
template <int NT, int BITS>
__global__ void __launch_bounds__(NT) rs_scatter_kernel(const uint32_t* __restrict__ kin,
                                                        const uint32_t* __restrict__ vin,  // NULL: value = index
                                                        uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                        const uint32_t* n_dev, int shift,
                                                        const uint32_t* __restrict__ hist,
                                                        const uint32_t* __restrict__ totals, int drop_max,
                                                        uint32_t* __restrict__ n_kept) {
  constexpr int RADIX = 1 << BITS;
  constexpr int NW = NT / 64;               // waves; wave w ranks the contiguous RS_TILE / NW keys [w0, w0 + ...)
  constexpr int ITEMS = RS_TILE / NT;       // keys per thread = ranking rounds per wave
  static_assert(NT >= RADIX && RS_TILE % NT == 0, "one thread per digit in phase 2");
  __shared__ uint32_t s_hist[NW][RADIX];  // phase 1: wave digit counts; phase 3: output bases
  const uint32_t n = *n_dev;
  const uint32_t nblk = (n + RS_TILE - 1) / RS_TILE;   // (the table's row stride: see rs_hist_kernel)
  const uint32_t t0 = blockIdx.x * RS_TILE;
  if (t0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#pragma unroll
  for (int k = tid; k < NW * RADIX; k += NT) (&s_hist[0][0])[k] = 0;
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
    const uint32_t d = (key[r] >> shift) & (uint32_t)(RADIX - 1);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
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
  __shared__ uint32_t s_delta[RADIX];  // global start - local start
  __shared__ uint32_t s_wtot[RADIX / 64];
  __shared__ uint32_t s_gtot[RADIX / 64];
  const bool digit_thread = tid < RADIX;
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
    if (tid == RADIX - 1) {
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
      const uint32_t lp = s_hist[wid][(key[r] >> shift) & (uint32_t)(RADIX - 1)] + pre[r];
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
      const uint32_t pos = i + s_delta[(k >> shift) & (uint32_t)(RADIX - 1)];
      kout[pos] = k;
      vout[pos] = s_val[i];
    }
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

// digit_bits: 8, or 9 for a sort that is then ONE pass over keys below 512 (the partition of the region entries at 1080p:
// 510 regions) - after it b.scan_tmp[d] holds the number of keys of digit d
int launch_radix_sort(const SortBufs& b, const uint32_t* n_dev, int64_t n_bound, int end_bit, int start_buf,
                      const uint32_t* first_keys, hipStream_t s, int debug, uint32_t* n_kept, int digit_bits) {
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
  if (digit_bits == 9) {
    if (end_bit > 9 || first_keys != nullptr) return GS_E_SHAPE;
    hipLaunchKernelGGL(rs_hist_kernel<9>, dim3(nblk), dim3(RS_HIST_THREADS), 0, s, b.keys[cur], n_dev, 0, b.hist, 0);
    GS_LAUNCH_CHECK(s, debug);
    hipLaunchKernelGGL(rs_rowscan_kernel, dim3(512), dim3(GS_BLOCK), 0, s, b.hist, n_dev, b.scan_tmp);
    GS_LAUNCH_CHECK(s, debug);
    hipLaunchKernelGGL((rs_scatter_kernel<RS_SCATTER_THREADS, 9>), dim3(nblk), dim3(RS_SCATTER_THREADS), 0, s, b.keys[cur], b.vals[cur],
                       b.keys[cur ^ 1], b.vals[cur ^ 1], n_dev, 0, b.hist, b.scan_tmp, 0, (uint32_t*)nullptr);
    GS_LAUNCH_CHECK(s, debug);
    return 0;
  }
  bool first = true;
  // (round 5 tried to count the NEXT pass's per-tile digits in this pass's scatter kernel - one-way atomics where the keys land,
  //  wave-aggregated - to save the histogram launch of passes 2-4: the counters of neighbouring tiles share cache lines and
  //  device-scope atomics on one line serialise at the memory side: a pass of 8 us took 0.3-1.9 ms.  Removed.)
  for (int shift = 0; shift < end_bit; shift += RS_BITS) {
    const bool ext = first && first_keys != nullptr;
    const uint32_t* kin = ext ? first_keys : b.keys[cur];
    const uint32_t* vin = ext ? nullptr : b.vals[cur];
    const int drop = ext && n_kept ? 1 : 0;  // the first pass of the depth sort filters (see rs_hist_kernel)
    hipLaunchKernelGGL(rs_hist_kernel<RS_BITS>, dim3(nblk), dim3(RS_HIST_THREADS), 0, s, kin, n_dev, shift, b.hist, drop);
    GS_LAUNCH_CHECK(s, debug);
    hipLaunchKernelGGL(rs_rowscan_kernel, dim3(RS_RADIX), dim3(GS_BLOCK), 0, s, b.hist, n_dev, b.scan_tmp);
    GS_LAUNCH_CHECK(s, debug);
    // 512 threads per 4096-key tile (8 ranking rounds per wave): 0.213 ms for the two instance passes against 0.226 with
    // 256 threads and 0.217 with 1024 - the pass is bound by one workgroup's dependent chain, not by throughput
    hipLaunchKernelGGL((rs_scatter_kernel<RS_SCATTER_THREADS, RS_BITS>), dim3(nblk), dim3(RS_SCATTER_THREADS), 0, s, kin, vin,
                       b.keys[cur ^ 1], b.vals[cur ^ 1], n_dev, shift, b.hist, b.scan_tmp, drop, drop ? n_kept : nullptr);
    GS_LAUNCH_CHECK(s, debug);
    if (drop) n_dev = n_kept;  // the later passes see the survivors only
    cur ^= 1;
    first = false;
  }
  return 0;
}

int launch_scan_sums(uint32_t* sums, int nb, hipStream_t s) {
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, sums, nb);
  return 0;
}
