// gs_render_fwd_wave.hip - forward blend, "one wave64 per tile, four pixels per lane".
//
// Same decomposition as gs_render_bwd_wave.hip: a 64-thread workgroup (one wave) owns a 16x16 tile, lane l
// blends the pixel at the same position of each 8x8 quadrant.  Compared with the four-waves-per-tile kernel
// (since removed) the per-Gaussian LDS broadcast reads and loop overhead are paid once per tile instead
// of once per quadrant, batches are 64 entries (a tile stops within 64 entries of its last live pixel, not
// 256) and there is no workgroup barrier on which three waves wait for the slowest one.
// Replaces renderCUDA<3> forward (forward.cu:274-397); per-pixel arithmetic and stopping rules unchanged.
#include "gs_blend.h"
#include "gs_common.h"
#include "gs_tilecull.h"

#define WB 64

// FSGS: the older rasterizer generation of FSGS / DNGaussian (-confidence fork, forward.cu:262-380): out_invdepth
// receives depth = sum depth_i alpha_i T_i, out_extra receives alpha = sum alpha_i T_i.  The image state keeps the
// transmittance PRODUCT: that generation's backward reads T_final back as 1 - alpha (backward.cu:461), which on
// saturated pixels (T ~ 1e-4) keeps only ~3 digits of T and puts ~1e-3 of fp32 noise on its gradients; the product
// is the same quantity without the cancellation (tests/test_gpu_fsgs.py compares against both forms).
// CULL: per-entry ellipse-extent test against the tile (for the reference's bounding-square lists); with the
// culled lists of gs_tilecull.h every entry already passed a tighter test at emission time.
template <bool HAS_EXTRA, bool FSGS, bool CULL>
__global__ void __launch_bounds__(64) render_fwd_wave_kernel(const uint2* __restrict__ ranges,
                                                             const uint32_t* __restrict__ point_list, int W, int H,
                                                             int grid_x, const Splat* __restrict__ splat,
                                                             const float* __restrict__ bg, float* __restrict__ final_T,
                                                             uint32_t* __restrict__ n_contrib,
                                                             uint32_t* __restrict__ tile_work,
                                                             const uint32_t* __restrict__ order_hint,
                                                             const float* __restrict__ depth_limit,
                                                             float* __restrict__ stop_depth,
                                                             uint32_t* __restrict__ trunc_failed,
                                                             float* __restrict__ out_color,
                                                             float* __restrict__ out_invdepth,
                                                             float* __restrict__ out_extra) {
  __shared__ float4 s_a[WB];  // x, y, invdepth, cull extent y (CULL) / view depth (!CULL)
  __shared__ float4 s_c[WB];  // conic, opacity
  __shared__ float4 s_k[WB];  // rgb, cull extent x
  __shared__ float s_e[HAS_EXTRA ? WB : 1];  // 4th channel (N1: NIR albedo blended with the same weights)

  // XCD-aware mapping: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so tile
  // t = xcd * ceil(T/8) + id/8 keeps a contiguous band of the image - whose tiles share splat records - on one L2
  const int n_tiles = grid_x * ((H + TILE_Y - 1) / TILE_Y);
  const int per_xcd = (int)(gridDim.x >> 3);  // the grid is padded to a multiple of 8 workgroups
  int tile_sw = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  // optional scheduling hint (GsScratch.tile_order_hint): the same bands, each walked longest-tile-first as measured on
  // an earlier view; entry 0xFFFFFFFF = no tile for this workgroup
  if (order_hint) tile_sw = (int)order_hint[blockIdx.x];
  if (tile_sw < 0 || tile_sw >= n_tiles) return;
  const int tile = tile_sw;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int lane = threadIdx.x;
  const int px0 = tile_x * TILE_X + (lane & 7), py0 = tile_y * TILE_Y + (lane >> 3);
  const float pixfx0 = (float)px0, pixfy0 = (float)py0;
  const float tile_fx0 = (float)(tile_x * TILE_X), tile_fy0 = (float)(tile_y * TILE_Y);

  const uint2 range = ranges[tile];
  const int n = (int)(range.y - range.x);
  const int rounds = (n + WB - 1) / WB;

  // T[s] > 0: the pixel's transmittance, pixel alive.  T[s] < 0: the pixel is done - it saturated at an earlier entry (or
  // lies outside the image) - and |T[s]| is its final transmittance (forward.cu:360-364 leaves T untouched when
  // T (1 - alpha) < 1e-4 and stops).  "Done" in the sign bit instead of a flag: a flag assigned under a divergent branch
  // and carried around the loop is kept by the compiler as a 0/1 byte per lane and converted to a lane mask and back
  // several times per entry - a third of the loop's 60-90 scalar instructions (a SIMD issues one scalar instruction per
  // four cycles: tests/tools/valu_peak_probe.hip), with nested exec-mask regions on top.
  float T[4], C0[4], C1[4], C2[4], D[4], X[4];
  uint32_t last_contributor[4];
  bool inside[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int px = px0 + (s & 1) * 8, py = py0 + (s >> 1) * 8;
    inside[s] = px < W && py < H;
    T[s] = inside[s] ? 1.0f : -1.0f;
    C0[s] = C1[s] = C2[s] = D[s] = X[s] = 0.f;
    last_contributor[s] = 0;
  }

  float stop_z = __builtin_inff();  // view depth of the entry at which the tile's last pixel saturated
  float4 ra, rc, rk;
  ra = rc = rk = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < n) {
    const float4* rec = reinterpret_cast<const float4*>(&splat[point_list[range.x + lane]]);
    ra = rec[0]; rc = rec[1]; rk = rec[2];
  }
  for (int i = 0; i < rounds; i++) {
    // the whole tile is done (forward.cu:326-328)
    if (!__any(fmaxf(fmaxf(T[0], T[1]), fmaxf(T[2], T[3])) > 0.f)) break;
    __syncthreads();  // single wave: previous batch fully consumed
    s_a[lane] = make_float4(ra.x, ra.y, FSGS ? ra.z : ra.w, ra.z);
    s_c[lane] = blend_stage_conic(rc);  // (qa, qb, qc, opacity), see gs_blend.h
    if (!CULL) {
      s_k[lane] = rk;
    } else {
      // Exact-safe tile cull: alpha >= 1/255 needs power >= -L with L = ln(255 * opacity); the set
      // {d : d^T Q d <= 2L} is an ellipse whose half-extent along x is sqrt(2 L Sigma_xx), Sigma = Q^-1
      // (Sigma_xx = Q_yy / det Q).  A tile whose pixel range lies beyond that extent (+2 % and one pixel of
      // slack, far above the rounding of the per-pixel test) cannot receive a contribution, so the whole
      // iteration is skipped.  The list itself is untouched (the reference's bounding square stays the binning rule).
      const float det = rc.x * rc.z - rc.y * rc.y;
      const float L2 = 2.0f * __logf(255.0f * 1.0001f * fmaxf(rc.w, 1e-20f));  // < 0: opacity below 1/255, never blends
      const bool ok = det > 0.f && L2 >= 0.f;
      const float ex = ok ? sqrtf(L2 * rc.z / det) * 1.02f + 1.0f : (L2 >= 0.f ? 3.0e38f : -1.0f);
      const float ey = ok ? sqrtf(L2 * rc.x / det) * 1.02f + 1.0f : (L2 >= 0.f ? 3.0e38f : -1.0f);
      s_k[lane] = make_float4(rk.x, rk.y, rk.z, ex);
      s_a[lane].w = ey;
    }
    if (HAS_EXTRA) s_e[lane] = rk.w;
    __syncthreads();
    {
      const int nxt = (i + 1) * WB + lane;
      if (nxt < n) {
        const float4* rec = reinterpret_cast<const float4*>(&splat[point_list[range.x + nxt]]);
        ra = rec[0]; rc = rec[1]; rk = rec[2];
      }
    }
    const int cnt = min(WB, n - i * WB);
    bool running = true;  // (wave-uniform; the outer loop's own check ends the tile when the batch loop has been left)
    for (int j = 0; j < cnt && running; j++) {
      const uint32_t contributor = (uint32_t)(i * WB + j + 1);
      const float4 a = s_a[j];
      const float4 k = s_k[j];
      if (CULL) {  // wave-uniform: distance from the centre to the tile's pixel range, per axis
        const float ddx = fmaxf(fmaxf(tile_fx0 - a.x, a.x - (tile_fx0 + 15.0f)), 0.f);
        const float ddy = fmaxf(fmaxf(tile_fy0 - a.y, a.y - (tile_fy0 + 15.0f)), 0.f);
        if (ddx > k.w || ddy > a.w) continue;
      }
      const float4 co = s_c[j];
      float alpha[4];
      bool hit[4];
      unsigned long long hmask = 0ull;  // lane masks stay scalar (see the backward kernel)
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const float dx = a.x - (pixfx0 + (float)((s & 1) * 8));
        const float dy = a.y - (pixfy0 + (float)((s >> 1) * 8));
        const float p2 = blend_power2(co, dx, dy);
        alpha[s] = fminf(0.99f, co.w * blend_exp2(p2));
        const bool c1 = p2 <= 0.0f, c2 = alpha[s] >= 1.0f / 255.0f;
        const bool alive = T[s] > 0.f;
        hit[s] = alive && c1 && c2;
        hmask |= __ballot(c1) & __ballot(c2) & __ballot(alive);
      }
      if (hmask == 0ull) continue;
      // one exec-mask region per touched quadrant, no branch inside it (a fully branch-free, select-masked update of all
      // four pixels was measured slower twice: round 2 0.215 vs 0.204 ms, round 4 0.156 vs 0.137 ms - DESIGN_APPENDIX)
#pragma unroll
      for (int s = 0; s < 4; s++) {
        if (hit[s]) {
          const float test_T = T[s] * (1 - alpha[s]);
          const bool sat = test_T < 0.0001f;          // saturates here: this entry is NOT blended, the pixel is done
          const float w = sat ? 0.f : alpha[s] * T[s];
          C0[s] += k.x * w;
          C1[s] += k.y * w;
          C2[s] += k.z * w;
          D[s] += a.z * w;
          if (HAS_EXTRA) X[s] += s_e[j] * w;
          if (FSGS) X[s] += w;
          T[s] = sat ? -T[s] : test_T;
          last_contributor[s] = sat ? last_contributor[s] : contributor;
        }
      }
      running = __any(fmaxf(fmaxf(T[0], T[1]), fmaxf(T[2], T[3])) > 0.f);
      if (!running && !CULL) stop_z = a.w;
    }
  }
  if (lane == 0) {
    // depth-limited emission (gs_tilecull.h): the list of this tile was cut at depth_limit[tile] (+ margin).  All entries
    // up to that bound are present and in the reference's order, so the blend is the reference's iff it stopped inside
    // them; otherwise the caller repeats the forward with full lists.
    stop_depth[tile] = stop_z;
    if (depth_limit) {
      const float lim = depth_limit[tile];
      if (lim < __builtin_inff() && depth_beyond_limit(stop_z, lim)) *trunc_failed = 1u;  // stop_z = +inf: never saturated
    }
  }
  {
    // what the backward blend of this tile will cost: it walks the list back from the tile's deepest last contributor
    uint32_t lmax = max(max(last_contributor[0], last_contributor[1]), max(last_contributor[2], last_contributor[3]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, off, 64));
    if (lane == 0) tile_work[tile] = lmax;
  }
  const size_t HW = (size_t)H * W;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    if (inside[s]) {
      const int px = px0 + (s & 1) * 8, py = py0 + (s >> 1) * 8;
      const int pix_id = W * py + px;
      const float Tf = fabsf(T[s]);
      final_T[pix_id] = Tf;
      n_contrib[pix_id] = last_contributor[s];
      out_color[pix_id] = C0[s] + Tf * bg0;
      out_color[HW + pix_id] = C1[s] + Tf * bg1;
      out_color[2 * HW + pix_id] = C2[s] + Tf * bg2;
      if (out_invdepth) out_invdepth[pix_id] = D[s];
      if (HAS_EXTRA) out_extra[pix_id] = X[s] + Tf * bg0;  // the reference's NIR pass keeps channel 0 (bg[0])
      if (FSGS) out_extra[pix_id] = X[s];
    }
  }
}

int launch_render_fwd_wave(const uint2* ranges, const uint32_t* point_list, int W, int H, int grid_x, int grid_y,
                           const Splat* splat, const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_work,
                           const uint32_t* order_hint, const float* depth_limit, float* stop_depth, uint32_t* trunc_failed,
                           float* out_color, float* out_invdepth, float* out_extra, int fsgs, int cull, hipStream_t s) {
#define GS_FWD_WAVE(EX, FS, CU)                                                                                          \
  hipLaunchKernelGGL((render_fwd_wave_kernel<EX, FS, CU>), dim3(((grid_x * grid_y + 7) / 8) * 8), dim3(64), 0, s, ranges, point_list, W, H, \
                     grid_x, splat, bg, final_T, n_contrib, tile_work, order_hint, cull ? nullptr : depth_limit, stop_depth, trunc_failed, out_color,   \
                     out_invdepth, out_extra)
  if (fsgs) {
    if (cull) GS_FWD_WAVE(false, true, true); else GS_FWD_WAVE(false, true, false);
  } else if (out_extra) {
    if (cull) GS_FWD_WAVE(true, false, true); else GS_FWD_WAVE(true, false, false);
  } else {
    if (cull) GS_FWD_WAVE(false, false, true); else GS_FWD_WAVE(false, false, false);
  }
#undef GS_FWD_WAVE
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Launch order of the backward blend: longest tile first.  One wave owns a tile and runs at the pace of its dependent
// instruction chain, so a tile's time is proportional to the list entries it visits (tile_work, max/mean = 1.9 at the
// bench workload) and the kernel ends when the last wave does.  In image order the 2 x 4096 wave slots of the chip are
// kept 64 % busy on average (a long tile that starts in the second round finishes alone); handing the tiles out
// by decreasing work (list scheduling, longest processing time first) brings that to 97 % in the same model
// (tests/tools/tile_stats.py; measured 0.42 -> 0.33 ms).  Counting sort, T is a few thousand.
// ------------------------------------------------------------------------------------------------------------------
// XCD affinity is kept: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so the tiles of image
// band x = [x * ceil(T/8), ...) - which share splat records - are sorted among themselves (workgroup x of this kernel) and
// interleaved into slots x, x + 8, x + 16, ...: workgroup b of the backward takes tile_order[b], runs on XCD b % 8 and
// that XCD walks its own band longest-first.  Slots past a band's end hold 0xFFFFFFFF (grid padded to 8 * ceil(T/8)).
#define TO_THREADS 1024
#define TO_BUCKETS 2048
__device__ __forceinline__ void stop_depth_bound_item(const float* __restrict__ stop, float* __restrict__ out, int grid_x,
                                                      int grid_y, int i, float slack = 1.0f);
// The same launch also serves the two per-camera exports when the caller asked for them in GsScratch (tile_order_out,
// tile_depth_limit_out): the order is written twice, and the eight workgroups share the depth bounds among them.
__global__ void __launch_bounds__(TO_THREADS) tile_order_kernel(const uint32_t* __restrict__ tile_work,
                                                                uint32_t* __restrict__ tile_order, int T, int per_band,
                                                                uint32_t* __restrict__ order_out,
                                                                const float* __restrict__ stop_depth,
                                                                float* __restrict__ limit_out, int grid_x, int grid_y,
                                                                const GeomHeader* __restrict__ hdr,
                                                                const float* __restrict__ slack_dev,
                                                                uint32_t* status_host, const uint32_t* step_tag) {
  // GsScratch.status_host: this is the forward's last kernel and every status word is final (the blend before it raised
  // trunc_failed): one thread delivers the block straight into the caller's pinned memory - no copy command behind the forward
  if (status_host && blockIdx.x == 0 && threadIdx.x == 0) {
    const uint32_t nr = hdr->num_rendered, ov = hdr->overflow, tf = hdr->trunc_failed;
    volatile uint32_t* out = status_host;
    out[0] = nr; out[1] = ov; out[2] = tf; out[3] = hdr->zero;
    if (step_tag) {
      const uint32_t t = *step_tag;
      out[9] = gs_status_check(t, nr, ov, tf);
      __threadfence_system();
      out[8] = t;   // (the tag last: a poller that sees it finds the words above in place, and verifies the check word anyway)
    }
    __threadfence_system();
  }
  // a forward that ran out of binning capacity blended nothing and is about to be repeated with the SAME hints and
  // bounds (the geometry phase has already counted with them): it must not overwrite them with what it did not measure
  if (hdr->overflow) order_out = nullptr;
  if (blockIdx.x >= 8) {  // workgroups past the eight bands: the depth bounds, one item per thread
    const int i = (int)(blockIdx.x - 8) * TO_THREADS + (int)threadIdx.x;
    if (!hdr->overflow && i < (int)depth_limit_floats((uint32_t)grid_x, (uint32_t)grid_y))
      stop_depth_bound_item(stop_depth, limit_out, grid_x, grid_y, i, slack_dev ? fmaxf(*slack_dev, 1.0f) : 1.0f);
    return;
  }
  __shared__ uint32_t s_cnt[TO_BUCKETS];
  __shared__ uint32_t s_part[TO_THREADS / 64];
  __shared__ uint32_t s_max;
  const int tid = threadIdx.x;
  const int band = blockIdx.x;
  const int t_lo = band * per_band, t_hi = min(T, t_lo + per_band);
  uint32_t m = 0;
  for (int t = t_lo + tid; t < t_hi; t += TO_THREADS) m = max(m, tile_work[t]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
  if ((tid & 63) == 0) s_part[tid >> 6] = m;
  for (int b = tid; b < TO_BUCKETS; b += TO_THREADS) s_cnt[b] = 0;
  for (int i = tid; i < per_band; i += TO_THREADS) {  // slots past the band's end
    tile_order[band + 8 * i] = 0xFFFFFFFFu;
    if (order_out) order_out[band + 8 * i] = 0xFFFFFFFFu;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t mm = 0;
    for (int w = 0; w < TO_THREADS / 64; w++) mm = max(mm, s_part[w]);
    s_max = mm;
  }
  __syncthreads();
  int shift = 0;
  while ((s_max >> shift) >= TO_BUCKETS) shift++;
  // bucket 0 = the longest tiles
  for (int t = t_lo + tid; t < t_hi; t += TO_THREADS) atomicAdd(&s_cnt[TO_BUCKETS - 1 - (tile_work[t] >> shift)], 1u);
  __syncthreads();
  // exclusive scan of the 2048 counts: two per thread, wave scan, then the 16 wave totals
  const uint32_t c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
  uint32_t run = c0 + c1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)run, off, 64);
    if ((tid & 63) >= off) run += up;
  }
  if ((tid & 63) == 63) s_part[tid >> 6] = run;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < (tid >> 6); w++) base += s_part[w];
  const uint32_t excl = base + run - (c0 + c1);
  __syncthreads();
  s_cnt[2 * tid] = excl;
  s_cnt[2 * tid + 1] = excl + c0;
  __syncthreads();
  for (int t = t_lo + tid; t < t_hi; t += TO_THREADS) {
    const uint32_t pos = atomicAdd(&s_cnt[TO_BUCKETS - 1 - (tile_work[t] >> shift)], 1u);
    tile_order[band + 8 * pos] = (uint32_t)t;
    if (order_out) order_out[band + 8 * pos] = (uint32_t)t;
  }
}

// gs_export_tile_stop_depth: per tile the largest stop depth of its 3 x 3 neighbourhood, then per aligned run of four
// tiles of a row the largest of those bounds (reasons: gs_tilecull.h); one launch, both straight from the stop depths
__device__ __forceinline__ void stop_depth_bound_item(const float* __restrict__ stop, float* __restrict__ out, int grid_x,
                                                      int grid_y, int i, float slack) {
  const int T = grid_x * grid_y, segs_x = (int)depth_limit_segs_x((uint32_t)grid_x);
  int x0, x1, ty;  // tile columns whose 3 x 3 neighbourhoods are merged
  if (i < T) {
    x0 = x1 = i % grid_x;
    ty = i / grid_x;
  } else {
    const int j = i - T;
    x0 = 4 * (j % segs_x);
    x1 = min(x0 + 3, grid_x - 1);
    ty = j / segs_x;
  }
  float hi = -__builtin_inff();
  for (int y = max(ty - 1, 0); y <= min(ty + 1, grid_y - 1); y++)
    for (int x = max(x0 - 1, 0); x <= min(x1 + 1, grid_x - 1); x++) hi = fmaxf(hi, stop[y * grid_x + x]);
  out[i] = hi * slack;  // (GsScratch.tile_depth_limit_slack >= 1: +inf stays +inf)
}
__global__ void __launch_bounds__(256) stop_depth_bounds_kernel(const float* __restrict__ stop, float* __restrict__ out,
                                                                int grid_x, int grid_y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < (int)depth_limit_floats((uint32_t)grid_x, (uint32_t)grid_y)) stop_depth_bound_item(stop, out, grid_x, grid_y, i);
}

int launch_export_stop_depth(const float* stop_depth, float* out, int grid_x, int grid_y, hipStream_t s) {
  const int n = (int)depth_limit_floats((uint32_t)grid_x, (uint32_t)grid_y);
  hipLaunchKernelGGL(stop_depth_bounds_kernel, dim3((n + 255) / 256), dim3(256), 0, s, stop_depth, out, grid_x, grid_y);
  return 0;
}

int launch_tile_order(const uint32_t* tile_work, uint32_t* tile_order, int T, uint32_t* order_out, const float* stop_depth,
                      float* limit_out, int grid_x, int grid_y, const GeomHeader* hdr, hipStream_t s, const float* slack_dev,
                      uint32_t* status_host, const uint32_t* step_tag) {
  const int extra = limit_out ? ((int)depth_limit_floats((uint32_t)grid_x, (uint32_t)grid_y) + TO_THREADS - 1) / TO_THREADS : 0;
  hipLaunchKernelGGL(tile_order_kernel, dim3(8 + extra), dim3(TO_THREADS), 0, s, tile_work, tile_order, T, (T + 7) / 8, order_out,
                     stop_depth, limit_out, grid_x, grid_y, hdr, slack_dev, status_host, step_tag);
  return 0;
}
