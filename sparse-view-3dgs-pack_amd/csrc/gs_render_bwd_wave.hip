// gs_render_bwd_wave.hip - backward blend, "one wave64 per tile, four pixels per lane".
//
// Measured on the first-generation backward kernel (since removed: one pixel per lane, four waves per tile): of
// 1.40 ms, the per-pair alpha test costs 0.31 ms, the gradient math 0.30 ms and the cross-lane reduction of
// the ten per-Gaussian sums 0.61 ms (+0.18 ms LDS accumulate / flush).  The reduction is paid once per
// (wave, Gaussian) pair with a touched lane, i.e. up to four times per (tile, Gaussian).
// Here ONE wave owns the whole 16x16 tile: lane l holds the pixel at the same position of each of the four
// 8x8 quadrants.  A Gaussian's contributions to up to four pixels are summed in registers for free, the
// wave reduces once per (tile, Gaussian), and - because the wave total IS the tile total - the ten sums go
// straight to memory as global atomics into one Gaussian's 64-byte gradient row (3 instructions, 10 lanes): no
// LDS accumulator, no zeroing, no flush pass, no workgroup barrier (a 64-thread workgroup is one wave).
// Replaces renderCUDA<3> backward (backward.cu:452-638); per-pixel recurrences are unchanged.
#include "gs_blend.h"
#include "gs_common.h"

template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dppw(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
// v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b; the sum then holds a's lane-pair sums in lanes
// 0..31 and b's in lanes 32..63: TWO values halve their lane count for one swap + one add
__device__ __forceinline__ float swap32_add(float a, float b) {
  const uint2v r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  // NB: element access as r[0]/r[1] + __uint_as_float; `bit_cast(float, r.x) + bit_cast(float, r.y)` is miscompiled
  // by hipcc 7.2 into x + x (checked in the ISA)
  const unsigned x = r[0], y = r[1];
  return __uint_as_float(x) + __uint_as_float(y);
}
// v_permlane16_swap: odd rows of a <-> even rows of b; with a = (u | w), b = (x | y) as produced by swap32_add the
// sum holds u, x, w, y in rows 0..3 (16 lanes each)
__device__ __forceinline__ float swap16_add(float a, float b) {
  const uint2v r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  // NB: element access as r[0]/r[1] + __uint_as_float; `bit_cast(float, r.x) + bit_cast(float, r.y)` is miscompiled
  // by hipcc 7.2 into x + x (checked in the ISA)
  const unsigned x = r[0], y = r[1];
  return __uint_as_float(x) + __uint_as_float(y);
}
// inclusive scan inside each row of 16 lanes: lane 15 of every row ends with the row total
__device__ __forceinline__ float row_total_in_lane15(float v) {
  v += dppw<0x111, 0xf, true>(v);  // row_shr:1
  v += dppw<0x112, 0xf, true>(v);  // row_shr:2
  v += dppw<0x114, 0xf, true>(v);  // row_shr:4
  v += dppw<0x118, 0xf, true>(v);  // row_shr:8
  return v;
}
// the same scan in the other direction: lane 0 of every row ends with the row total
__device__ __forceinline__ float row_total_in_lane0(float v) {
  v += dppw<0x101, 0xf, true>(v);  // row_shl:1
  v += dppw<0x102, 0xf, true>(v);  // row_shl:2
  v += dppw<0x104, 0xf, true>(v);  // row_shl:4
  v += dppw<0x108, 0xf, true>(v);  // row_shl:8
  return v;
}

#define WB 64  // batch = one list entry per lane

// FSGS: the older generation's backward (-confidence fork, backward.cu:414-600) = the depth channel carries the
// view-space depth instead of its inverse and the 4th channel is the alpha image: a constant colour 1 without a
// background term (accum_alpha_rec = last_alpha + (1 - last_alpha) accum_alpha_rec, dL_dopa += (1 - accum) dL_dalpha).
template <bool HAS_INVDEPTH, bool HAS_EXTRA, bool FSGS>
__global__ void __launch_bounds__(64, (HAS_INVDEPTH || HAS_EXTRA) ? 3 : 4) render_bwd_wave_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int grid_x,
    const Splat* __restrict__ splat, const float* __restrict__ bg, const float* __restrict__ final_Ts,
    const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ tile_work,
    const uint32_t* __restrict__ tile_order, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_invdepths, const float* __restrict__ dL_dextra, gs_row_t* __restrict__ grad_rows) {
  __shared__ float4 s_a[WB];  // x, y, invdepth, -
  __shared__ float4 s_c[WB];  // conic, opacity
  __shared__ float4 s_k[WB];  // rgb, 4th channel
  __shared__ uint32_t s_id[WB];

  // longest tile first (tile_order_kernel in gs_render_fwd_wave.hip): the hardware hands workgroups to free wave slots
  // in index order, so this is list scheduling by decreasing work - 64 % -> 97 % of the wave slots busy in the model -
  // and slot b holds a tile of image band b % 8, so that a band's splat records stay in one XCD's L2
  const uint32_t tile_u = tile_order[blockIdx.x];
  if (tile_u == 0xFFFFFFFFu) return;
  const int tile = (int)tile_u;
  const uint32_t lmax = tile_work[tile];  // deepest last contributor of the tile's pixels
  if (lmax == 0) return;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int lane = threadIdx.x;
  const uint2 range = ranges[tile];
  const int n = (int)(range.y - range.x);
  if (n == 0) return;

  // slot s = quadrant (s&1, s>>1); this lane's pixel inside every quadrant is (lane&7, lane>>3)
  const int px0 = tile_x * TILE_X + (lane & 7), py0 = tile_y * TILE_Y + (lane >> 3);
  const float pixfx0 = (float)px0, pixfy0 = (float)py0;
  const size_t HW = (size_t)H * W;

  // Per pixel the reference runs one "colour behind me" recurrence per channel (accum_rec[ch], last_color[ch],
  // backward.cu:573-591) and then contracts with dL_dpixel[ch].  The contraction commutes with the recurrence (it is
  // linear in the colour), so ONE scalar recurrence per pixel carries the same information:
  //   kd = sum_ch c_ch dL_dpixel_ch            (this contributor's colour, seen through the pixel's cotangent)
  //   A  = last_alpha * (kd_prev - A) + A      (= sum_ch accum_rec[ch] dL_dpixel_ch)
  //   dL_dalpha = kd - A
  // 6 VALU ops instead of 15 per (pixel, Gaussian) and 2 state registers instead of 6 (8 with depth + 4th channel),
  // which the depth and 4th channels join for one fma each.
  float T[4], Tbg[4], dLp0[4], dLp1[4], dLp2[4], dLinv[4], dLpX[4];  // Tbg = T_final * (bg . dL_dpixel)
  float A[4], kd_prev[4], last_alpha[4];
  uint32_t lastc[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int px = px0 + (s & 1) * 8, py = py0 + (s >> 1) * 8;
    const bool inside = px < W && py < H;
    const int pix_id = W * py + px;
    T[s] = inside ? final_Ts[pix_id] : 0.f;
    lastc[s] = inside ? n_contrib[pix_id] : 0u;
    dLp0[s] = inside ? dL_dpixels[pix_id] : 0.f;
    dLp1[s] = inside ? dL_dpixels[HW + pix_id] : 0.f;
    dLp2[s] = inside ? dL_dpixels[2 * HW + pix_id] : 0.f;
    dLinv[s] = (HAS_INVDEPTH && inside) ? dL_invdepths[pix_id] : 0.f;
    dLpX[s] = (HAS_EXTRA && inside) ? dL_dextra[pix_id] : 0.f;
    float b = 0;
    if (HAS_EXTRA && !FSGS) b += bg[0] * dLpX[s];  // the 4th channel's image is X + T bg[0]
    b += bg[0] * dLp0[s];
    b += bg[1] * dLp1[s];
    b += bg[2] * dLp2[s];
    Tbg[s] = T[s] * b;
    A[s] = kd_prev[s] = last_alpha[s] = 0.f;
  }
  // After the reduction the twelve totals of an entry sit in THREE lanes of every 16-lane row - lane 15: value q of t0,
  // lane 0: value 4 + q of t1, lane 7: t2's value of row q - so that ONE atomic instruction (12 lanes, one 64-byte gradient
  // row) carries them all; per-lane slot and factor are fixed for the whole kernel.  (Three instructions with four lanes each
  // were three requests to the memory-side atomic units per entry: 2.9 M per view at the bench workload, against the ~20 G
  // requests/s those units retire - MI355X_MICROARCH.md "Global float atomics".)
  const int rq = lane >> 4, lr = lane & 15;
  const float rowscale0 = rq == 0 ? (0.5f * W) / GS_LOG2E : (rq == 1 ? (0.5f * H) / GS_LOG2E : -0.5f);
  const float rowscale1 = rq == 0 ? -0.5f : 1.0f;
  const bool live2 = rq == 0 || (HAS_EXTRA && !FSGS && rq == 1) || (HAS_INVDEPTH && rq == 2);
  const int lane_slot = lr == 15 ? rq : (lr == 0 ? 4 + rq : (rq == 0 ? GR_CB : (rq == 1 ? GR_EXTRA : (rq == 2 ? GR_ID : GR_N))));
  const float lane_scale = lr == 15 ? rowscale0 : (lr == 0 ? rowscale1 : 1.0f);
  const bool lane_adds = lr == 15 || lr == 0 || (lr == 7 && live2);
  const int q0 = n - (int)lmax;  // entries q < q0 (counted from the back) are behind every pixel's last contributor
  const int rounds = (n + WB - 1) / WB;

  // prefetch the first needed batch
  float4 ra, rc, rk;
  uint32_t rid = 0;
  ra = rc = rk = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const int q = (q0 / WB) * WB + lane;
    if (q < n) {
      rid = point_list[range.y - q - 1];
      const float4* rec = reinterpret_cast<const float4*>(&splat[rid]);
      ra = rec[0]; rc = rec[1]; rk = rec[2];
    }
  }
  for (int i = q0 / WB; i < rounds; i++) {
    __syncthreads();  // (single wave) previous batch fully consumed
    s_id[lane] = rid;
    s_a[lane] = make_float4(ra.x, ra.y, FSGS ? ra.z : ra.w, 0.f);
    s_c[lane] = blend_stage_conic(rc);  // (qa, qb, qc, opacity), see gs_blend.h
    s_k[lane] = FSGS ? make_float4(rk.x, rk.y, rk.z, 1.0f) : rk;
    __syncthreads();
    {
      const int q = (i + 1) * WB + lane;
      if (q < n) {
        rid = point_list[range.y - q - 1];
        const float4* rec = reinterpret_cast<const float4*>(&splat[rid]);
        ra = rec[0]; rc = rec[1]; rk = rec[2];
      }
    }
    const int cnt = min(WB, n - i * WB);
    const int jbeg = max(0, q0 - i * WB);
    for (int j = jbeg; j < cnt; j++) {
      const uint32_t contributor = (uint32_t)(n - 1 - (i * WB + j));
      const float4 a = s_a[j];
      const float4 co = s_c[j];
      // alpha test for the four pixels of this lane
      // (the four lane masks are kept as 64-bit scalars: with per-lane bools OR-ed together the compiler materialises
      // them as 0/1 bytes in VGPRs - four v_cndmask, shifts and a bit-op per entry - and re-derives the masks afterwards)
      float G[4], alpha[4];
      bool valid[4];
      unsigned long long vmask[4];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const float dx = a.x - (pixfx0 + (float)((s & 1) * 8));
        const float dy = a.y - (pixfy0 + (float)((s >> 1) * 8));
        const float p2 = blend_power2(co, dx, dy);
        G[s] = blend_exp2(p2);
        alpha[s] = fminf(0.99f, co.w * G[s]);
        const bool c0 = contributor < lastc[s], c1 = p2 <= 0.0f, c2 = alpha[s] >= 1.0f / 255.0f;
        valid[s] = c0 && c1 && c2;
        vmask[s] = __ballot(c0) & __ballot(c1) & __ballot(c2);  // ballot of a compare IS the compare's scalar result
      }
      if ((vmask[0] | vmask[1] | vmask[2] | vmask[3]) == 0ull) continue;

      // Raw sums; the constant factors of backward.cu:617-635 are applied once per (tile, Gaussian) after the
      // reduction:  mean2D.x = (0.5 W / log2e) * sum h (2 qa dx + qb dy),  conic.xx = -0.5 * sum h dx^2, ...
      // with h = opacity * G * dL_dalpha.
      float v_mx = 0.f, v_my = 0.f, v_cxx = 0.f, v_cxy = 0.f, v_cyy = 0.f, v_op = 0.f;
      float v_c0 = 0.f, v_c1 = 0.f, v_c2 = 0.f, v_id = 0.f, v_x = 0.f;
      const float4 k = s_k[j];
      const float q2a = co.x + co.x, q2c = co.z + co.z;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        // (no separate wave-uniform "quadrant untouched" test: the exec-mask region's own s_cbranch_execz is that test, and a
        // scalar instruction costs 2.7 vector issue slots - tests/tools/valu_peak_probe.hip)
        if (valid[s]) {
          const float dx = a.x - (pixfx0 + (float)((s & 1) * 8));
          const float dy = a.y - (pixfy0 + (float)((s >> 1) * 8));
          const float om = 1.f - alpha[s];
          float rinv = __builtin_amdgcn_rcpf(om);
          rinv = fmaf(rinv, fmaf(-om, rinv, 1.0f), rinv);
          T[s] = T[s] * rinv;
          const float w = alpha[s] * T[s];  // dchannel_dcolor
          A[s] = fmaf(last_alpha[s], kd_prev[s] - A[s], A[s]);
          float kd = k.x * dLp0[s];
          kd = fmaf(k.y, dLp1[s], kd);
          kd = fmaf(k.z, dLp2[s], kd);
          v_c0 = fmaf(w, dLp0[s], v_c0);
          v_c1 = fmaf(w, dLp1[s], v_c1);
          v_c2 = fmaf(w, dLp2[s], v_c2);
          if (HAS_INVDEPTH) {
            kd = fmaf(a.z, dLinv[s], kd);
            v_id = fmaf(w, dLinv[s], v_id);
          }
          if (HAS_EXTRA) {
            kd = fmaf(k.w, dLpX[s], kd);
            v_x = fmaf(w, dLpX[s], v_x);
          }
          kd_prev[s] = kd;
          const float dL_dalpha = fmaf(kd - A[s], T[s], -Tbg[s] * rinv);  // *T, then + (-T_final/(1-alpha)) * bg_dot_dpixel
          last_alpha[s] = alpha[s];
          v_op = fmaf(G[s], dL_dalpha, v_op);
          const float h = co.w * G[s] * dL_dalpha;
          const float hx = h * dx, hy = h * dy;
          v_cxx = fmaf(hx, dx, v_cxx);
          v_cxy = fmaf(hx, dy, v_cxy);
          v_cyy = fmaf(hy, dy, v_cyy);
          v_mx = fmaf(q2a, hx, fmaf(co.y, hy, v_mx));
          v_my = fmaf(q2c, hy, fmaf(co.y, hx, v_my));
        }
      }
      // One reduction per (tile, Gaussian): 5 + 3 swap-adds pack the ten sums into three registers holding one
      // value per row of 16 lanes, 3 x 4 row shifts finish them (28 VALU ops instead of 10 x 6 DPP adds), and
      // lane 15 of every row adds its total to the Gaussian's 64-byte gradient row.
      const float s0 = swap32_add(v_mx, v_cxx), s1 = swap32_add(v_my, v_cxy);   // (0|2) (1|3)
      const float s2 = swap32_add(v_cyy, v_c0), s3 = swap32_add(v_op, v_c1);    // (4|6) (5|7)
      const float s4 = swap32_add(v_c2, v_id);                                  // (8|9)
      const float t0 = row_total_in_lane15(swap16_add(s0, s1));                 // rows: 0 1 2 3   (total in lane 15)
      const float t1 = row_total_in_lane0(swap16_add(s2, s3));                  // rows: 4 5 6 7   (total in lane 0)
      const float s5 = HAS_EXTRA ? swap32_add(v_x, 0.f) : 0.f;                  // (10|-)
      const float t2 = row_total_in_lane15(swap16_add(s4, s5));                 // rows: 8 10 9 -  (total in lane 15 ...
      const float t2m = dppw<0x128, 0xf, true>(t2);                             //  ... row_ror:8: lane 15 -> lane 7)
      // (no "!= 0" test: an entry that reaches this point has a touched pixel, its sums are non-zero in practice)
      const float tot = (lr == 15 ? t0 : (lr == 0 ? t1 : t2m)) * lane_scale;
      // float64 slots (gs_common.h): the tile's fp32 total joins the Gaussian's row without a rounding that depends on
      // which tiles came before it - one global_atomic_add_f64 instruction, twelve lanes, one 128-byte line
      if (lane_adds) atomicAdd(grad_rows + (size_t)s_id[j] * GR_STRIDE + lane_slot, (double)tot);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Developer statistics (gs_debug_blend_stats): what the backward blend's loop meets.  Walks every tile's list exactly as
// render_bwd_wave_kernel does (same entries, same alpha test) and counts:
//   0 entries visited | 1 entries with at least one valid pixel | 2 (entry, quadrant) pairs with a valid pixel |
//   3 valid (entry, pixel) pairs | 4 tiles with work | 5 list entries in those tiles | 6 entries whose every quadrant is
//   behind its last contributor or untouched by the ellipse's bounding box (what a cheap scalar pre-test would skip)
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) blend_stats_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                         int W, int H, int grid_x, const Splat* __restrict__ splat,
                                                         const uint32_t* __restrict__ n_contrib,
                                                         const uint32_t* __restrict__ tile_work,
                                                         unsigned long long* __restrict__ out) {
  const int tile = blockIdx.x;
  const uint32_t lmax = tile_work[tile];
  if (lmax == 0) return;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int lane = threadIdx.x;
  const uint2 range = ranges[tile];
  const int n = (int)(range.y - range.x);
  if (n == 0) return;
  const int px0 = tile_x * TILE_X + (lane & 7), py0 = tile_y * TILE_Y + (lane >> 3);
  uint32_t lastc[4];
  for (int s = 0; s < 4; s++) {
    const int px = px0 + (s & 1) * 8, py = py0 + (s >> 1) * 8;
    lastc[s] = (px < W && py < H) ? n_contrib[W * py + px] : 0u;
  }
  unsigned long long c_vis = 0, c_any = 0, c_quad = 0, c_pix = 0;
  for (int q = n - (int)lmax; q < n; q++) {
    const uint32_t contributor = (uint32_t)(n - 1 - q);
    const uint32_t id = point_list[range.y - q - 1];
    const float4* rec = reinterpret_cast<const float4*>(&splat[id]);
    const float4 ra = rec[0];
    const float4 co = blend_stage_conic(rec[1]);
    c_vis++;
    int quads = 0;
    for (int s = 0; s < 4; s++) {
      const float dx = ra.x - (float)(px0 + (s & 1) * 8), dy = ra.y - (float)(py0 + (s >> 1) * 8);
      const float p2 = blend_power2(co, dx, dy);
      const float alpha = fminf(0.99f, co.w * blend_exp2(p2));
      const bool valid = contributor < lastc[s] && p2 <= 0.0f && alpha >= 1.0f / 255.0f;
      const unsigned long long m = __ballot(valid);
      if (m) quads++;
      c_pix += (unsigned long long)__popcll(m);
    }
    c_quad += quads;
    c_any += quads ? 1 : 0;
  }
  if (lane == 0) {
    atomicAdd(&out[0], c_vis);
    atomicAdd(&out[1], c_any);
    atomicAdd(&out[2], c_quad);
    atomicAdd(&out[3], c_pix);
    atomicAdd(&out[4], 1ull);
    atomicAdd(&out[5], (unsigned long long)n);
  }
}
int launch_blend_stats(const uint2* ranges, const uint32_t* point_list, int W, int H, int grid_x, int grid_y, const Splat* splat,
                       const uint32_t* n_contrib, const uint32_t* tile_work, unsigned long long* out, hipStream_t s) {
  hipLaunchKernelGGL(blend_stats_kernel, dim3(grid_x * grid_y), dim3(64), 0, s, ranges, point_list, W, H, grid_x, splat, n_contrib,
                     tile_work, out);
  return 0;
}

// zero fill of the gradient rows (GR_STRIDE = 16 float64 slots, twelve in use = six float4 per row): one thread per
// Gaussian.  With tiles_touched given only the rows of Gaussians that emitted instances are cleared: no atomic lands
// anywhere else and the chain kernel does not read the others (PreprocessBwdArgs.skip_uninstanced) - a fifth of the rows
// with depth-limited lists.  (One thread per float4 - 8 M threads at 1 M Gaussians, seven eighths of them leaving after
// the flag load - took 26 us; this takes 7.)
__global__ void __launch_bounds__(GS_BLOCK) zero_rows_kernel(float4* __restrict__ p, size_t P,
                                                             const uint32_t* __restrict__ tiles_touched) {
  const size_t i = (size_t)blockIdx.x * GS_BLOCK + threadIdx.x;
  if (i >= P || (tiles_touched && tiles_touched[i] == 0)) return;
  float4* w = p + i * (GR_ROW_BYTES / 16);
  w[0] = w[1] = w[2] = w[3] = w[4] = w[5] = make_float4(0.f, 0.f, 0.f, 0.f);
}
int launch_zero_rows(gs_row_t* rows, size_t P, const uint32_t* tiles_touched, hipStream_t s) {
  static_assert(GR_ROW_BYTES == 128, "eight float4 per row, six in use");
  if (P) hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((P + GS_BLOCK - 1) / GS_BLOCK)), dim3(GS_BLOCK), 0, s,
                            reinterpret_cast<float4*>(rows), P, tiles_touched);
  return 0;
}

int launch_render_bwd_wave(const uint2* ranges, const uint32_t* point_list, int W, int H, int grid_x, int grid_y,
                           const Splat* splat, const float* bg, const float* final_T, const uint32_t* n_contrib,
                           const uint32_t* tile_work, const uint32_t* tile_order, const float* dL_dpix,
                           const float* dL_dinvdepth, const float* dL_dextra, gs_row_t* grad_rows, int fsgs, hipStream_t s) {
#define GS_BWD_WAVE(ID, EX, FS)                                                                                           \
  hipLaunchKernelGGL((render_bwd_wave_kernel<ID, EX, FS>), dim3(((grid_x * grid_y + 7) / 8) * 8), dim3(64), 0, s, ranges, point_list, W, H, \
                     grid_x, splat, bg, final_T, n_contrib, tile_work, tile_order, dL_dpix, dL_dinvdepth, dL_dextra, grad_rows)
  if (fsgs) GS_BWD_WAVE(true, true, true);
  else if (dL_dinvdepth && dL_dextra) GS_BWD_WAVE(true, true, false);
  else if (dL_dinvdepth) GS_BWD_WAVE(true, false, false);
  else if (dL_dextra) GS_BWD_WAVE(false, true, false);
  else GS_BWD_WAVE(false, false, false);
#undef GS_BWD_WAVE
  return 0;
}
