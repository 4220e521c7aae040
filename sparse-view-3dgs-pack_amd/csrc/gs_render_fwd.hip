// gs_render_fwd.hip - per-tile front-to-back alpha blend (the forward hot kernel).
//
// Replaces renderCUDA<3> forward (forward.cu:274-397).  One 256-thread workgroup (4 wave64) per
// 16x16 tile; each wave owns an 8x8 pixel quadrant (compact footprint -> fewer live lanes per
// Gaussian than a 16x4 strip).  The tile's depth-sorted list is consumed in batches of 256
// entries: every lane gathers ONE 64-byte splat record (a single cache line, mostly served from
// L2 / Infinity Cache) into LDS - position, conic, opacity, colour AND inverse depth (the reference
// stages only 28 B and re-reads colour/depth from global memory per pixel per contributor).  The next
// batch's gather is issued before the current batch is blended, so its latency hides behind the
// VALU work.  In the blend loop all lanes read the same LDS address (broadcast reads).
//
// Bound: VALU (~22 issue slots + one v_exp_f32 per pixel x Gaussian pair); compulsory HBM traffic
// is 4 B (id) + <=64 B (record) per instance and 24 B per pixel.
#include "gs_blend.h"
#include "gs_common.h"

struct __attribute__((aligned(16))) StageRec {
  float4 a;  // x, y, invdepth, -
  float4 c;  // cxx, cxy, cyy, opacity
  float4 k;  // r, g, b, -
};

__global__ void __launch_bounds__(GS_BLOCK) render_fwd_kernel(const uint2* __restrict__ ranges,
                                                              const uint32_t* __restrict__ point_list, int W, int H,
                                                              int grid_x, const Splat* __restrict__ splat,
                                                              const float* __restrict__ bg,
                                                              float* __restrict__ final_T,
                                                              uint32_t* __restrict__ n_contrib,
                                                              float* __restrict__ out_color,
                                                              float* __restrict__ out_invdepth) {
  __shared__ float4 s_a[GS_BLOCK];
  __shared__ float4 s_c[GS_BLOCK];
  __shared__ float4 s_k[GS_BLOCK];

  const int tile = blockIdx.x;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lx = (wid & 1) * 8 + (lane & 7), ly = (wid >> 1) * 8 + (lane >> 3);
  const int px = tile_x * TILE_X + lx, py = tile_y * TILE_Y + ly;
  const bool inside = px < W && py < H;
  const float pixfx = (float)px, pixfy = (float)py;
  bool done = !inside;

  const uint2 range = ranges[tile];
  const int n = (int)(range.y - range.x);
  const int rounds = (n + GS_BLOCK - 1) / GS_BLOCK;

  float T = 1.0f;
  uint32_t contributor = 0, last_contributor = 0;
  float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;

  // prefetch batch 0
  float4 ra, rc, rk;
  ra = rc = rk = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < n) {
    const float4* rec = reinterpret_cast<const float4*>(&splat[point_list[range.x + tid]]);
    ra = rec[0]; rc = rec[1]; rk = rec[2];
  }
  for (int i = 0; i < rounds; i++) {
    // end if the whole tile is done (forward.cu:326-328)
    if (__syncthreads_count(done) == GS_BLOCK) break;
    s_a[tid] = make_float4(ra.x, ra.y, ra.w, 0.f);
    s_c[tid] = rc;
    s_k[tid] = rk;
    __syncthreads();
    {  // issue the next batch's gather now; it completes while this batch is blended
      const int nxt = (i + 1) * GS_BLOCK + tid;
      if (nxt < n) {
        const float4* rec = reinterpret_cast<const float4*>(&splat[point_list[range.x + nxt]]);
        ra = rec[0]; rc = rec[1]; rk = rec[2];
      }
    }
    const int cnt = min(GS_BLOCK, n - i * GS_BLOCK);
    if (!done) {
      for (int j = 0; j < cnt; j++) {
        contributor++;
        const float4 a = s_a[j];
        const float4 co = s_c[j];
        const float dx = a.x - pixfx, dy = a.y - pixfy;
        const float power = blend_power2(blend_stage_conic(co), dx, dy);  // same decision arithmetic as every blend kernel
        if (power > 0.0f) continue;
        const float alpha = fminf(0.99f, co.w * blend_exp2(power));
        if (alpha < 1.0f / 255.0f) continue;
        const float test_T = T * (1 - alpha);
        if (test_T < 0.0001f) {
          done = true;
          break;
        }
        const float4 k = s_k[j];
        const float w = alpha * T;
        C0 += k.x * w;
        C1 += k.y * w;
        C2 += k.z * w;
        D += a.z * w;
        T = test_T;
        last_contributor = contributor;
      }
    }
  }
  if (inside) {
    const int pix_id = W * py + px;
    final_T[pix_id] = T;
    n_contrib[pix_id] = last_contributor;
    const size_t HW = (size_t)H * W;
    out_color[pix_id] = C0 + T * bg[0];
    out_color[HW + pix_id] = C1 + T * bg[1];
    out_color[2 * HW + pix_id] = C2 + T * bg[2];
    if (out_invdepth) out_invdepth[pix_id] = D;
  }
}

int launch_render_fwd(const uint2* ranges, const uint32_t* point_list, int W, int H, int grid_x, int grid_y,
                      const Splat* splat, const float* bg, float* final_T, uint32_t* n_contrib, float* out_color,
                      float* out_invdepth, hipStream_t s) {
  hipLaunchKernelGGL(render_fwd_kernel, dim3(grid_x * grid_y), dim3(GS_BLOCK), 0, s, ranges, point_list, W, H, grid_x,
                     splat, bg, final_T, n_contrib, out_color, out_invdepth);
  return 0;
}
