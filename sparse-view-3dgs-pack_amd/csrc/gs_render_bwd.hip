// gs_render_bwd.hip - per-tile back-to-front replay producing per-Gaussian 2D gradients.
//
// Replaces renderCUDA<3> backward (backward.cu:452-638).  Same tiling as the forward kernel
// (256 threads = 4 wave64, one 8x8 pixel quadrant per wave, 256-entry batches staged through LDS).
// The reference issues 10 global float atomicAdd per (pixel, Gaussian) pair; on MI355X scattered
// float atomics run ~17x below the streaming rate, so the accumulation is restructured:
//   1. every wave reduces its 64 lanes' contributions with DPP row shifts / row broadcasts
//      (6 v_add_f32_dpp per value, no LDS traffic) - skipped entirely when no lane of the wave
//      is touched by the Gaussian (wave-uniform ballot);
//   2. lane 63 adds the 10 wave totals into a per-batch LDS accumulator (ds_add_f32);
//   3. after the batch, the workgroup flushes the LDS rows with global atomics shaped so that 16
//      consecutive lanes cover ONE Gaussian's 64-byte gradient row (one memory-side atomic request
//      per (tile, Gaussian) instead of 2560), skipping rows nobody touched.
// Entries behind every pixel's last contributor are never visited: the loop starts at the first
// entry some pixel of the wave (batch: of the tile) actually blended in the forward pass.
#include <stdlib.h>

#include "gs_blend.h"
#include "gs_common.h"

template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
// sum over the 64 lanes; the total is valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_mov<0x111, 0xf, true>(v);   // row_shr:1
  v += dpp_mov<0x112, 0xf, true>(v);   // row_shr:2
  v += dpp_mov<0x114, 0xf, true>(v);   // row_shr:4
  v += dpp_mov<0x118, 0xf, true>(v);   // row_shr:8   -> lane 15 of each row holds the row total
  v += dpp_mov<0x142, 0xa, false>(v);  // row_bcast:15 into rows 1,3
  v += dpp_mov<0x143, 0xc, false>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave total
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
  return v;
}

#define ACC_STRIDE 11  // 10 values + touched flag per entry; rows [j][11]: odd stride -> conflict-free flush reads

// VARIANT: 0 = product; 1/2 = timing experiments only (wrong results): 1 drops the cross-lane reduction and the
// accumulator traffic, 2 additionally drops the per-pair gradient math (alpha test only)
template <bool HAS_INVDEPTH, int VARIANT>
__global__ void __launch_bounds__(GS_BLOCK) render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int grid_x,
    const Splat* __restrict__ splat, const float* __restrict__ bg, const float* __restrict__ final_Ts,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_invdepths, float* __restrict__ grad_rows) {
  __shared__ float4 s_a[GS_BLOCK];  // x, y, invdepth, -
  __shared__ float4 s_c[GS_BLOCK];  // conic, opacity
  __shared__ float4 s_k[GS_BLOCK];  // rgb
  __shared__ uint32_t s_id[GS_BLOCK];
  __shared__ float s_acc[GS_BLOCK * ACC_STRIDE];
  __shared__ uint32_t s_wmax[GS_BLOCK / 64];

  const int tile = blockIdx.x;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lx = (wid & 1) * 8 + (lane & 7), ly = (wid >> 1) * 8 + (lane >> 3);
  const int px = tile_x * TILE_X + lx, py = tile_y * TILE_Y + ly;
  const bool inside = px < W && py < H;
  const int pix_id = W * py + px;
  const float pixfx = (float)px, pixfy = (float)py;

  const uint2 range = ranges[tile];
  const int n = (int)(range.y - range.x);
  if (n == 0) return;

  const float T_final = inside ? final_Ts[pix_id] : 0.f;
  float T = T_final;
  const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0u;

  float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLinv = 0.f;
  if (inside) {
    const size_t HW = (size_t)H * W;
    dLp0 = dL_dpixels[pix_id];
    dLp1 = dL_dpixels[HW + pix_id];
    dLp2 = dL_dpixels[2 * HW + pix_id];
    if (HAS_INVDEPTH) dLinv = dL_invdepths[pix_id];
  }
  float bg_dot_dpixel = 0;
  bg_dot_dpixel += bg[0] * dLp0;
  bg_dot_dpixel += bg[1] * dLp1;
  bg_dot_dpixel += bg[2] * dLp2;

  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accD = 0.f;  // accum_rec / accum_invdepth_rec
  float lastc0 = 0.f, lastc1 = 0.f, lastc2 = 0.f, lastD = 0.f, last_alpha = 0.f;

  const float ddelx_dx = 0.5f * W;
  const float ddely_dy = 0.5f * H;

  // first list position (counted from the back) any pixel of this wave / tile needs
  const uint32_t wmax = __builtin_amdgcn_readfirstlane(wave_max_u32(last_contributor));
  if (lane == 0) s_wmax[wid] = wmax;
  __syncthreads();
  const uint32_t bmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
  if (bmax == 0) return;
  const int q0_wave = n - (int)wmax;   // entries q < q0 are behind every pixel's last contributor
  const int q0_block = n - (int)bmax;
  const int rounds = (n + GS_BLOCK - 1) / GS_BLOCK;

  for (int i = q0_block / GS_BLOCK; i < rounds; i++) {
    __syncthreads();
    {
      const int q = i * GS_BLOCK + tid;
      if (q < n) {
        const uint32_t id = point_list[range.y - q - 1];
        const float4* rec = reinterpret_cast<const float4*>(&splat[id]);
        const float4 a = rec[0];
        s_id[tid] = id;
        s_a[tid] = make_float4(a.x, a.y, a.w, 0.f);
        s_c[tid] = rec[1];
        s_k[tid] = rec[2];
      }
#pragma unroll
      for (int k = 0; k < ACC_STRIDE; k++) s_acc[k * GS_BLOCK + tid] = 0.f;
    }
    __syncthreads();
    const int cnt = min(GS_BLOCK, n - i * GS_BLOCK);
    const int jbeg = max(0, q0_wave - i * GS_BLOCK);
    for (int j = jbeg; j < cnt; j++) {
      const uint32_t contributor = (uint32_t)(n - 1 - (i * GS_BLOCK + j));
      const float4 a = s_a[j];
      const float4 co = s_c[j];
      const float dx = a.x - pixfx, dy = a.y - pixfy;
      const float power = blend_power2(blend_stage_conic(co), dx, dy);  // same decision arithmetic as every blend kernel
      const float G = blend_exp2(power);
      const float alpha = fminf(0.99f, co.w * G);
      const bool valid = (contributor < last_contributor) && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
      if (!__any(valid)) continue;
      if (VARIANT == 2) { T += G; continue; }

      float v_mx = 0.f, v_my = 0.f, v_cxx = 0.f, v_cxy = 0.f, v_cyy = 0.f, v_op = 0.f;
      float v_c0 = 0.f, v_c1 = 0.f, v_c2 = 0.f, v_id = 0.f;
      if (valid) {
        // 1/(1-alpha): hardware reciprocal + one Newton step (the two divisions of backward.cu:567,613)
        const float om = 1.f - alpha;
        float rinv = __builtin_amdgcn_rcpf(om);
        rinv = fmaf(rinv, fmaf(-om, rinv, 1.0f), rinv);
        T = T * rinv;
        const float dchannel_dcolor = alpha * T;
        const float4 k = s_k[j];
        float dL_dalpha = 0.0f;
        const float oma = 1.f - last_alpha;
        acc0 = last_alpha * lastc0 + oma * acc0;
        lastc0 = k.x;
        dL_dalpha += (k.x - acc0) * dLp0;
        v_c0 = dchannel_dcolor * dLp0;
        acc1 = last_alpha * lastc1 + oma * acc1;
        lastc1 = k.y;
        dL_dalpha += (k.y - acc1) * dLp1;
        v_c1 = dchannel_dcolor * dLp1;
        acc2 = last_alpha * lastc2 + oma * acc2;
        lastc2 = k.z;
        dL_dalpha += (k.z - acc2) * dLp2;
        v_c2 = dchannel_dcolor * dLp2;
        if (HAS_INVDEPTH) {
          const float invd = a.z;
          accD = last_alpha * lastD + oma * accD;
          lastD = invd;
          dL_dalpha += (invd - accD) * dLinv;
          v_id = dchannel_dcolor * dLinv;
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final * rinv) * bg_dot_dpixel;
        const float dL_dG = co.w * dL_dalpha;
        const float gdx = G * dx;
        const float gdy = G * dy;
        const float dG_ddelx = -gdx * co.x - gdy * co.y;
        const float dG_ddely = -gdy * co.z - gdx * co.y;
        v_mx = dL_dG * dG_ddelx * ddelx_dx;
        v_my = dL_dG * dG_ddely * ddely_dy;
        v_cxx = -0.5f * gdx * dx * dL_dG;
        v_cxy = -0.5f * gdx * dy * dL_dG;
        v_cyy = -0.5f * gdy * dy * dL_dG;
        v_op = G * dL_dalpha;
      }
      if (VARIANT == 4) {  // every touched lane adds straight into the LDS accumulator (no cross-lane reduction)
        if (valid) {
          float* row = &s_acc[j * ACC_STRIDE];
          atomicAdd(&row[GR_MX], v_mx);
          atomicAdd(&row[GR_MY], v_my);
          atomicAdd(&row[GR_CXX], v_cxx);
          atomicAdd(&row[GR_CXY], v_cxy);
          atomicAdd(&row[GR_CYY], v_cyy);
          atomicAdd(&row[GR_OP], v_op);
          atomicAdd(&row[GR_CR], v_c0);
          atomicAdd(&row[GR_CG], v_c1);
          atomicAdd(&row[GR_CB], v_c2);
          if (HAS_INVDEPTH) atomicAdd(&row[GR_ID], v_id);
          row[GR_N] = 1.0f;
        }
        continue;
      }
      if (VARIANT == 1) { T += (v_mx + v_my + v_cxx + v_cxy + v_cyy + v_op + v_c0 + v_c1 + v_c2 + v_id) * 1e-30f; continue; }
      v_mx = wave_sum_to_lane63(v_mx);
      v_my = wave_sum_to_lane63(v_my);
      v_cxx = wave_sum_to_lane63(v_cxx);
      v_cxy = wave_sum_to_lane63(v_cxy);
      v_cyy = wave_sum_to_lane63(v_cyy);
      v_op = wave_sum_to_lane63(v_op);
      v_c0 = wave_sum_to_lane63(v_c0);
      v_c1 = wave_sum_to_lane63(v_c1);
      v_c2 = wave_sum_to_lane63(v_c2);
      if (HAS_INVDEPTH) v_id = wave_sum_to_lane63(v_id);
      if (VARIANT == 3) { T += (v_mx + v_my + v_cxx + v_cxy + v_cyy + v_op + v_c0 + v_c1 + v_c2 + v_id) * 1e-30f; continue; }
      if (lane == 63) {
        atomicAdd(&s_acc[j * ACC_STRIDE + GR_MX], v_mx);
        atomicAdd(&s_acc[j * ACC_STRIDE + GR_MY], v_my);
        atomicAdd(&s_acc[j * ACC_STRIDE + GR_CXX], v_cxx);
        atomicAdd(&s_acc[j * ACC_STRIDE + GR_CXY], v_cxy);
        atomicAdd(&s_acc[j * ACC_STRIDE + GR_CYY], v_cyy);
        atomicAdd(&s_acc[j * ACC_STRIDE + GR_OP], v_op);
        atomicAdd(&s_acc[j * ACC_STRIDE + GR_CR], v_c0);
        atomicAdd(&s_acc[j * ACC_STRIDE + GR_CG], v_c1);
        atomicAdd(&s_acc[j * ACC_STRIDE + GR_CB], v_c2);
        if (HAS_INVDEPTH) atomicAdd(&s_acc[j * ACC_STRIDE + GR_ID], v_id);
        s_acc[j * ACC_STRIDE + GR_N] = 1.0f;  // touched flag
      }
    }
    __syncthreads();
    // flush: 16 consecutive lanes = one Gaussian's 64-B gradient row
    for (int e = tid; e < cnt * GR_STRIDE; e += GS_BLOCK) {
      const int j = e / GR_STRIDE, k = e % GR_STRIDE;
      if (k < GR_N && s_acc[j * ACC_STRIDE + GR_N] != 0.0f) {
        const float v = s_acc[j * ACC_STRIDE + k];
        if (v != 0.0f) atomicAdd(&grad_rows[(size_t)s_id[j] * GR_STRIDE + k], v);
      }
    }
  }
  if (VARIANT != 0 && T == 123.456f) atomicAdd(&grad_rows[0], T + acc0 + acc1 + acc2 + accD);
}

int launch_render_bwd(const uint2* ranges, const uint32_t* point_list, int W, int H, int grid_x, int grid_y,
                      const Splat* splat, const float* bg, const float* final_T, const uint32_t* n_contrib,
                      const float* dL_dpix, const float* dL_dinvdepth, float* grad_rows, hipStream_t s) {
  static const int variant = getenv("GS_BWD_VARIANT") ? atoi(getenv("GS_BWD_VARIANT")) : 0;
#define GS_LAUNCH_BWD(INV, VAR)                                                                                          \
  hipLaunchKernelGGL((render_bwd_kernel<INV, VAR>), dim3(grid_x * grid_y), dim3(GS_BLOCK), 0, s, ranges, point_list, W, H, \
                     grid_x, splat, bg, final_T, n_contrib, dL_dpix, dL_dinvdepth, grad_rows)
  if (variant == 1) GS_LAUNCH_BWD(true, 1);
  else if (variant == 2) GS_LAUNCH_BWD(true, 2);
  else if (variant == 3) GS_LAUNCH_BWD(true, 3);
  else if (variant == 4) GS_LAUNCH_BWD(true, 4);
  else if (dL_dinvdepth) GS_LAUNCH_BWD(true, 0);
  else GS_LAUNCH_BWD(false, 0);
#undef GS_LAUNCH_BWD
  return 0;
}
