// gs_loss.hip - image-space loss kernels of LGDWT-GS (all HBM-bound streaming kernels).
//
//   L1                          LGDWT-GS/utils/loss_utils.py:40-41
//   Haar DWT ('db1','symmetric')  the arithmetic of pytorch_wavelets.DWTForward as used by
//                               LGDWT-GS/utils/loss_utils.py:106-153 (two separable stages with
//                               c = float32(0.70710678...), first along W then along H; odd sizes repeat
//                               the last sample; LH = low along W / high along H)
//   global 2-level DWT L1 loss  LGDWT-GS/train.py:132-164 - ONE pass over pred and gt computes all eight
//                               sub-band L1 sums (the reference runs a wasted J=2 transform plus two J=1
//                               transforms per image and materialises 16 band tensors)
//   ELF map / patch selection   LGDWT-GS/utils/loss_utils.py:336-442
//   fused SSIM                  fused-ssim/ssim.cu:187-366 (11-tap sigma=1.5 separable window, zero padding)
#include "gs_common.h"
#include "gs_prof.h"

#define HC 0.7071067811865476f

struct Bands {
  float ll, lh, hl, hh;
};
__device__ __forceinline__ Bands haar_block(float a, float b, float c, float d) {
  const float lo_t = HC * a + HC * b, hi_t = HC * a - HC * b;
  const float lo_b = HC * c + HC * d, hi_b = HC * c - HC * d;
  Bands r;
  r.ll = HC * lo_t + HC * lo_b;
  r.lh = HC * lo_t - HC * lo_b;
  r.hl = HC * hi_t + HC * hi_b;
  r.hh = HC * hi_t - HC * hi_b;
  return r;
}
__device__ __forceinline__ void haar_block_adj(float gll, float glh, float ghl, float ghh, float& da, float& db,
                                               float& dc, float& dd) {
  const float dlo_t = HC * gll + HC * glh, dlo_b = HC * gll - HC * glh;
  const float dhi_t = HC * ghl + HC * ghh, dhi_b = HC * ghl - HC * ghh;
  da = HC * dlo_t + HC * dhi_t;
  db = HC * dlo_t - HC * dhi_t;
  dc = HC * dlo_b + HC * dhi_b;
  dd = HC * dlo_b - HC * dhi_b;
}
__device__ __forceinline__ float sgnf(float x) { return (float)((x > 0.f) - (x < 0.f)); }
__device__ __forceinline__ int cdiv2(int n) { return (n + 1) >> 1; }

template <int N>
__device__ __forceinline__ void block_sum_atomic(float (&v)[N], float* out) {
  __shared__ float red[GS_BLOCK / 64][N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    float x = v[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = x;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    float s = 0.f;
    for (int w = 0; w < GS_BLOCK / 64; w++) s += red[w][threadIdx.x];
    if (s != 0.f) atomicAdd(&out[threadIdx.x], s);
  }
}

// ------------------------------------------------------------------------------------------------ L1
// partials != NULL: workgroup w stores its sum to partials[w] instead of adding it to *sum atomically (a caller that wants
// the total independent of the order in which the workgroups finish adds the partials itself: gs_lgdwt_combine_pp)
__global__ void __launch_bounds__(GS_BLOCK) l1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          int64_t n, float* sum, float* __restrict__ partials) {
  float acc[1] = {0.f};
  const int64_t n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  for (int64_t i = (int64_t)blockIdx.x * GS_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * GS_BLOCK) {
    const float4 x = a4[i], y = b4[i];
    acc[0] += fabsf(x.x - y.x) + fabsf(x.y - y.y) + fabsf(x.z - y.z) + fabsf(x.w - y.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc[0] += fabsf(a[n4 * 4 + threadIdx.x] - b[n4 * 4 + threadIdx.x]);
  if (partials) {
    __shared__ float red[GS_BLOCK / 64];
    float x = acc[0];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < GS_BLOCK / 64; w++) t += red[w];
      partials[blockIdx.x] = t;
    }
    return;
  }
  block_sum_atomic<1>(acc, sum);
}
__global__ void __launch_bounds__(GS_BLOCK) l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          int64_t n, float coef, const float* __restrict__ coef_dev,
                                                          float* __restrict__ g, int accumulate) {
  if (coef_dev) coef *= coef_dev[0];
  for (int64_t i = (int64_t)blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * GS_BLOCK) {
    const float v = coef * sgnf(a[i] - b[i]);
    g[i] = accumulate ? g[i] + v : v;
  }
}

// Depth regularisation of the step (LGDWT-GS/train.py:204-216): Ll1depth_pure = mean |(invDepth - mono) * mask|.  One pass
// over the three images gives both the term (per-workgroup partial sums, added by the caller in index order) and its
// gradient coef * sign((d - m) k) k - which does not depend on the sum, so the train step needs no second launch.
__global__ void __launch_bounds__(GS_BLOCK) depth_l1_kernel(const float* __restrict__ d, const float* __restrict__ m,
                                                            const float* __restrict__ k, int64_t n, float* __restrict__ partials,
                                                            float coef, const float* __restrict__ coef_dev,
                                                            float* __restrict__ grad) {
  if (coef_dev) coef *= coef_dev[0];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * GS_BLOCK) {
    const float w = k ? k[i] : 1.0f;
    const float v = (d[i] - m[i]) * w;
    acc += fabsf(v);
    if (grad) grad[i] = coef * sgnf(v) * w;
  }
  if (partials) {
    __shared__ float red[GS_BLOCK / 64];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < GS_BLOCK / 64; w++) t += red[w];
      partials[blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------ single-level DWT
__global__ void __launch_bounds__(GS_BLOCK) dwt_fwd_kernel(const float* __restrict__ x, int NC, int H, int W, float* ll,
                                                           float* lh, float* hl, float* hh) {
  const int h = cdiv2(H), w = cdiv2(W);
  const int64_t total = (int64_t)NC * h * w;
  for (int64_t o = (int64_t)blockIdx.x * GS_BLOCK + threadIdx.x; o < total; o += (int64_t)gridDim.x * GS_BLOCK) {
    const int j = (int)(o % w), i = (int)((o / w) % h), c = (int)(o / ((int64_t)w * h));
    const float* p = x + (size_t)c * H * W;
    const int y0 = 2 * i, y1 = min(2 * i + 1, H - 1), x0 = 2 * j, x1 = min(2 * j + 1, W - 1);
    const Bands b = haar_block(p[(size_t)y0 * W + x0], p[(size_t)y0 * W + x1], p[(size_t)y1 * W + x0], p[(size_t)y1 * W + x1]);
    if (ll) ll[o] = b.ll;
    if (lh) lh[o] = b.lh;
    if (hl) hl[o] = b.hl;
    if (hh) hh[o] = b.hh;
  }
}
// adjoint: one thread per 2x2 block owns its four output pixels (padded duplicates fold onto the last row/col)
__global__ void __launch_bounds__(GS_BLOCK) dwt_bwd_kernel(const float* dll, const float* dlh, const float* dhl,
                                                           const float* dhh, int NC, int H, int W, float* __restrict__ dx) {
  const int h = cdiv2(H), w = cdiv2(W);
  const int64_t total = (int64_t)NC * h * w;
  for (int64_t o = (int64_t)blockIdx.x * GS_BLOCK + threadIdx.x; o < total; o += (int64_t)gridDim.x * GS_BLOCK) {
    const int j = (int)(o % w), i = (int)((o / w) % h), c = (int)(o / ((int64_t)w * h));
    float da, db, dc, dd;
    haar_block_adj(dll ? dll[o] : 0.f, dlh ? dlh[o] : 0.f, dhl ? dhl[o] : 0.f, dhh ? dhh[o] : 0.f, da, db, dc, dd);
    float* p = dx + (size_t)c * H * W;
    const int y0 = 2 * i, x0 = 2 * j;
    const bool py = 2 * i + 1 >= H, pxp = 2 * j + 1 >= W;  // padded row / column
    // clamped targets: fold in the oracle's order  p[y0,x0]+=da; p[y0,x1]+=db; p[y1,x0]+=dc; p[y1,x1]+=dd
    if (!py && !pxp) {
      p[(size_t)y0 * W + x0] = da;
      p[(size_t)y0 * W + x0 + 1] = db;
      p[(size_t)(y0 + 1) * W + x0] = dc;
      p[(size_t)(y0 + 1) * W + x0 + 1] = dd;
    } else if (py && !pxp) {
      p[(size_t)y0 * W + x0] = da + dc;
      p[(size_t)y0 * W + x0 + 1] = db + dd;
    } else if (!py && pxp) {
      p[(size_t)y0 * W + x0] = da + db;
      p[(size_t)(y0 + 1) * W + x0] = dc + dd;
    } else {
      p[(size_t)y0 * W + x0] = ((da + db) + dc) + dd;
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused 2-level DWT L1
// One thread per level-2 coefficient = a 4x4 pixel block of one channel.  It forms the four level-1
// blocks (bands of pred and gt separately, as the reference does) and the level-2 block of their LL1.
// The same 16 pixels also give the plain L1 term (loss_utils.py:40-41) for free, so the fused criterion gets both
// from one read of the two images.  FAST (H, W multiples of 4, 16-byte aligned rows) moves whole float4 rows.
struct Px44 {
  float v[4][4];  // [row][col] of the 4x4 pixel block (padding already replicated)
};
struct Blk44 {
  float l1[2][2][4];  // [r][c][band] level-1 bands (r,c = position inside the level-2 block)
  float l2[4];
};
template <bool FAST>
__device__ __forceinline__ void load_px44(const float* __restrict__ p, int H, int W, int i2, int j2, int h1, int w1,
                                          Px44& o) {
  if (FAST) {
#pragma unroll
    for (int y = 0; y < 4; y++) {
      const float4 q = reinterpret_cast<const float4*>(p + (size_t)(4 * i2 + y) * W)[j2];
      o.v[y][0] = q.x; o.v[y][1] = q.y; o.v[y][2] = q.z; o.v[y][3] = q.w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int r1 = min(2 * i2 + r, h1 - 1), c1 = min(2 * j2 + c, w1 - 1);  // symmetric repeat of LL1's last sample
        const int y0 = 2 * r1, y1 = min(2 * r1 + 1, H - 1), x0 = 2 * c1, x1 = min(2 * c1 + 1, W - 1);
        o.v[2 * r][2 * c] = p[(size_t)y0 * W + x0];
        o.v[2 * r][2 * c + 1] = p[(size_t)y0 * W + x1];
        o.v[2 * r + 1][2 * c] = p[(size_t)y1 * W + x0];
        o.v[2 * r + 1][2 * c + 1] = p[(size_t)y1 * W + x1];
      }
  }
}
__device__ __forceinline__ void bands_of(const Px44& x, Blk44& o) {
  float ll[2][2];
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const Bands b = haar_block(x.v[2 * r][2 * c], x.v[2 * r][2 * c + 1], x.v[2 * r + 1][2 * c], x.v[2 * r + 1][2 * c + 1]);
      o.l1[r][c][0] = b.ll; o.l1[r][c][1] = b.lh; o.l1[r][c][2] = b.hl; o.l1[r][c][3] = b.hh;
      ll[r][c] = b.ll;
    }
  const Bands b2 = haar_block(ll[0][0], ll[0][1], ll[1][0], ll[1][1]);
  o.l2[0] = b2.ll; o.l2[1] = b2.lh; o.l2[2] = b2.hl; o.l2[3] = b2.hh;
}

// sums[0..7] += band L1 sums; sums[8] (only when l1_sum != nullptr, added there) += sum |pred - gt|
// clamped_out (FAST only): `pred` is the un-clamped render - the 4x4 block is clamped to [0, 1] as it is loaded and written
// there, which is the torch.clamp pass of gaussian_renderer/__init__.py:119 without its own read of the image
// patch_mask (FAST, patch size a multiple of 4, l1_sum given): the patch-DWT term (loss_utils.py:395-442) from the same
// pass - its three sums are the level-1 LH / HL / HH differences of THE SAME 2x2 blocks, restricted to the selected
// patches (a patch is a multiple of 4 pixels wide and starts on one: a thread's 4x4 block lies inside one patch), so
// patch_dwt_kernel<false> and its launch disappear: patch_sums[0..2] += sum over selected patches of |d LH|, |d HL|, |d HH|.
template <bool FAST>
__global__ void __launch_bounds__(GS_BLOCK) dwt2_l1_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                               int C, int H, int W, float* band_sums, float* l1_sum,
                                                               float* __restrict__ clamped_out,
                                                               const uint8_t* __restrict__ patch_mask, int ps,
                                                               float* patch_sums, float* __restrict__ partials) {
  const int h1 = cdiv2(H), w1 = cdiv2(W), h2 = cdiv2(h1), w2 = cdiv2(w1);
  const int64_t total = (int64_t)C * h2 * w2;
  const int pnx = patch_mask ? W / ps : 0, pny = patch_mask ? H / ps : 0;
  float s[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t o = (int64_t)blockIdx.x * GS_BLOCK + threadIdx.x; o < total; o += (int64_t)gridDim.x * GS_BLOCK) {
    const int j2 = (int)(o % w2), i2 = (int)((o / w2) % h2), c = (int)(o / ((int64_t)w2 * h2));
    Px44 pa, pb;
    load_px44<FAST>(pred + (size_t)c * H * W, H, W, i2, j2, h1, w1, pa);
    load_px44<FAST>(gt + (size_t)c * H * W, H, W, i2, j2, h1, w1, pb);
    if (FAST && clamped_out) {
#pragma unroll
      for (int y = 0; y < 4; y++) {
#pragma unroll
        for (int x = 0; x < 4; x++) pa.v[y][x] = fminf(fmaxf(pa.v[y][x], 0.f), 1.f);  // (torch.clamp: NaN stays NaN - v_max/v_min drop it; renders are finite)
        reinterpret_cast<float4*>(clamped_out + (size_t)c * H * W + (size_t)(4 * i2 + y) * W)[j2] =
            make_float4(pa.v[y][0], pa.v[y][1], pa.v[y][2], pa.v[y][3]);
      }
    }
    Blk44 a, b;
    bands_of(pa, a);
    bands_of(pb, b);
    bool in_patch = false;
    if (FAST && patch_mask) {
      const int py = (4 * i2) / ps, px = (4 * j2) / ps;
      in_patch = py < pny && px < pnx && patch_mask[py * pnx + px] != 0;
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int cc = 0; cc < 2; cc++)
        if (FAST || (2 * i2 + r < h1 && 2 * j2 + cc < w1)) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float d = fabsf(a.l1[r][cc][k] - b.l1[r][cc][k]);
            s[k] += d;
            if (FAST && k > 0 && in_patch) s[8 + k] += d;   // (9, 10, 11 = LH, HL, HH of the selected patches)
          }
        }
#pragma unroll
    for (int k = 0; k < 4; k++) s[4 + k] += fabsf(a.l2[k] - b.l2[k]);
    if (l1_sum) {
#pragma unroll
      for (int y = 0; y < 4; y++)
#pragma unroll
        for (int x = 0; x < 4; x++)
          if (FAST || (4 * i2 + y < H && 4 * j2 + x < W)) s[8] += fabsf(pa.v[y][x] - pb.v[y][x]);
    }
  }
  if (l1_sum) {
    // the nine (twelve with the patch term) sums go to two (three) places: one reduction, then route them
    __shared__ float red12[GS_BLOCK / 64][12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
      float x = s[k];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off, 64);
      if ((threadIdx.x & 63) == 0) red12[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
      float t = 0.f;
      for (int w = 0; w < GS_BLOCK / 64; w++) t += red12[w][threadIdx.x];
      if (partials)   // (deterministic form: the caller adds the workgroups' sums in index order, gs_lgdwt_combine_pp)
        partials[(size_t)blockIdx.x * 12 + threadIdx.x] = t;
      else if (t != 0.f)
        atomicAdd(threadIdx.x < 8 ? &band_sums[threadIdx.x] : (threadIdx.x == 8 ? l1_sum : &patch_sums[threadIdx.x - 9]), t);
    }
  } else {
    float s8[8];
#pragma unroll
    for (int k = 0; k < 8; k++) s8[k] = s[k];
    block_sum_atomic<8>(s8, band_sums);
  }
}

// grad (+= if accumulate) = DWT adjoint of the band signs (coef[8]) [+ l1_coef[0] * sign(pred - gt)]
template <bool FAST>
__global__ void __launch_bounds__(GS_BLOCK) dwt2_l1_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                               int C, int H, int W, const float* __restrict__ coef,
                                                               const float* __restrict__ l1_coef, float* __restrict__ grad,
                                                               int accumulate, const uint8_t* __restrict__ patch_mask, int ps,
                                                               const float* __restrict__ patch_coef) {
  // patch_mask (FAST, patch size a multiple of 4): the gradient of the patch-DWT term from the same pass - inside a selected
  // patch the level-1 LH / HL / HH signs take the patch coefficients on top of the global ones (see dwt2_l1_fwd_kernel)
  const int h1 = cdiv2(H), w1 = cdiv2(W), h2 = cdiv2(h1), w2 = cdiv2(w1);
  const int64_t total = (int64_t)C * h2 * w2;
  const int pnx = patch_mask ? W / ps : 0, pny = patch_mask ? H / ps : 0;
  float cfg[8];
#pragma unroll
  for (int k = 0; k < 8; k++) cfg[k] = coef[k];
  float cp[3] = {0.f, 0.f, 0.f};
  if (patch_mask) { cp[0] = patch_coef[0]; cp[1] = patch_coef[1]; cp[2] = patch_coef[2]; }
  const float c1l = l1_coef ? l1_coef[0] : 0.f;
  for (int64_t o = (int64_t)blockIdx.x * GS_BLOCK + threadIdx.x; o < total; o += (int64_t)gridDim.x * GS_BLOCK) {
    const int j2 = (int)(o % w2), i2 = (int)((o / w2) % h2), c = (int)(o / ((int64_t)w2 * h2));
    Px44 pa, pb;
    load_px44<FAST>(pred + (size_t)c * H * W, H, W, i2, j2, h1, w1, pa);
    load_px44<FAST>(gt + (size_t)c * H * W, H, W, i2, j2, h1, w1, pb);
    Blk44 a, b;
    bands_of(pa, a);
    bands_of(pb, b);
    float cf[8];
#pragma unroll
    for (int k = 0; k < 8; k++) cf[k] = cfg[k];
    if (FAST && patch_mask) {
      const int py = (4 * i2) / ps, px = (4 * j2) / ps;
      if (py < pny && px < pnx && patch_mask[py * pnx + px] != 0) {
        cf[1] += cp[0];
        cf[2] += cp[1];
        cf[3] += cp[2];
      }
    }
    // level-2 adjoint -> gradient of the four LL1 inputs (padded duplicates fold onto the last sample)
    float dl[2][2];
    {
      float da, db, dc, dd;
      haar_block_adj(cf[4] * sgnf(a.l2[0] - b.l2[0]), cf[5] * sgnf(a.l2[1] - b.l2[1]), cf[6] * sgnf(a.l2[2] - b.l2[2]),
                     cf[7] * sgnf(a.l2[3] - b.l2[3]), da, db, dc, dd);
      const bool pr = !FAST && 2 * i2 + 1 >= h1, pc = !FAST && 2 * j2 + 1 >= w1;
      dl[0][0] = da; dl[0][1] = db; dl[1][0] = dc; dl[1][1] = dd;
      if (pr && pc) { dl[0][0] = ((da + db) + dc) + dd; }
      else if (pr) { dl[0][0] = da + dc; dl[0][1] = db + dd; }
      else if (pc) { dl[0][0] = da + db; dl[1][0] = dc + dd; }
    }
    float* gp = grad + (size_t)c * H * W;
    float out[4][4];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int cc = 0; cc < 2; cc++) {
        const int r1 = 2 * i2 + r, c1 = 2 * j2 + cc;
        if (!FAST && (r1 >= h1 || c1 >= w1)) continue;
        float da, db, dc, dd;
        haar_block_adj(cf[0] * sgnf(a.l1[r][cc][0] - b.l1[r][cc][0]) + dl[r][cc], cf[1] * sgnf(a.l1[r][cc][1] - b.l1[r][cc][1]),
                       cf[2] * sgnf(a.l1[r][cc][2] - b.l1[r][cc][2]), cf[3] * sgnf(a.l1[r][cc][3] - b.l1[r][cc][3]), da, db,
                       dc, dd);
        const int y0 = 2 * r1, x0 = 2 * c1;
        const bool py = !FAST && y0 + 1 >= H, pxp = !FAST && x0 + 1 >= W;
        float v00 = da, v01 = db, v10 = dc, v11 = dd;
        if (py && pxp) v00 = ((da + db) + dc) + dd;
        else if (py) { v00 = da + dc; v01 = db + dd; }
        else if (pxp) { v00 = da + db; v10 = dc + dd; }
        // the plain-L1 gradient lands on the real pixels only (r1, c1 are not clamped here, so pa holds them)
        v00 += c1l * sgnf(pa.v[2 * r][2 * cc] - pb.v[2 * r][2 * cc]);
        v01 += c1l * sgnf(pa.v[2 * r][2 * cc + 1] - pb.v[2 * r][2 * cc + 1]);
        v10 += c1l * sgnf(pa.v[2 * r + 1][2 * cc] - pb.v[2 * r + 1][2 * cc]);
        v11 += c1l * sgnf(pa.v[2 * r + 1][2 * cc + 1] - pb.v[2 * r + 1][2 * cc + 1]);
        if (FAST) {
          out[2 * r][2 * cc] = v00; out[2 * r][2 * cc + 1] = v01; out[2 * r + 1][2 * cc] = v10; out[2 * r + 1][2 * cc + 1] = v11;
        } else {
          float* q = gp + (size_t)y0 * W + x0;
          if (accumulate) {
            q[0] += v00;
            if (!pxp) q[1] += v01;
            if (!py) { q[W] += v10; if (!pxp) q[W + 1] += v11; }
          } else {
            q[0] = v00;
            if (!pxp) q[1] = v01;
            if (!py) { q[W] = v10; if (!pxp) q[W + 1] = v11; }
          }
        }
      }
    if (FAST) {
#pragma unroll
      for (int y = 0; y < 4; y++) {
        float4* q = reinterpret_cast<float4*>(gp + (size_t)(4 * i2 + y) * W) + j2;
        float4 v = make_float4(out[y][0], out[y][1], out[y][2], out[y][3]);
        if (accumulate) { const float4 old = *q; v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
        *q = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ ELF
__global__ void __launch_bounds__(GS_BLOCK) elf_low_kernel(const float* __restrict__ img, int C, int H, int W,
                                                           float* __restrict__ low) {
  const int h = cdiv2(H), w = cdiv2(W);
  const int o = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (o >= h * w) return;
  const int j = o % w, i = o / w;
  const int y0 = 2 * i, y1 = min(2 * i + 1, H - 1), x0 = 2 * j, x1 = min(2 * j + 1, W - 1);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int c = 0; c < C; c++) {
    const float* p = img + (size_t)c * H * W;
    const Bands b = haar_block(p[(size_t)y0 * W + x0], p[(size_t)y0 * W + x1], p[(size_t)y1 * W + x0], p[(size_t)y1 * W + x1]);
    s0 += fabsf(b.ll); s1 += fabsf(b.lh); s2 += fabsf(b.hl); s3 += fabsf(b.hh);
  }
  const float HF = s1 + s2 + s3;
  low[o] = s0 / (s0 + HF + 1e-8f);
}
// F.interpolate(size=(H,W), mode='bilinear', align_corners=False)
__global__ void __launch_bounds__(GS_BLOCK) bilinear_up_kernel(const float* __restrict__ low, int h, int w, int H, int W,
                                                               float* __restrict__ out) {
  const int o = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (o >= H * W) return;
  const int x = o % W, y = o / W;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  float sy = sh * ((float)y + 0.5f) - 0.5f;
  if (sy < 0) sy = 0;
  const int y0 = (int)sy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
  float sx = sw * ((float)x + 0.5f) - 0.5f;
  if (sx < 0) sx = 0;
  const int x0 = (int)sx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
  out[o] = ly0 * (lx0 * low[(size_t)y0 * w + x0] + lx1 * low[(size_t)y0 * w + x1]) +
           ly1 * (lx0 * low[(size_t)y1 * w + x0] + lx1 * low[(size_t)y1 * w + x1]);
}
// one workgroup per patch
__global__ void __launch_bounds__(GS_BLOCK) patch_means_kernel(const float* __restrict__ elf, int H, int W, int ps,
                                                               float* __restrict__ means) {
  const int nx = W / ps;
  const int px = blockIdx.x % nx, py = blockIdx.x / nx;
  float acc[1] = {0.f};
  for (int k = threadIdx.x; k < ps * ps; k += GS_BLOCK) {
    const int y = k / ps, x = k % ps;
    acc[0] += elf[(size_t)(py * ps + y) * W + px * ps + x];
  }
  __shared__ float red[GS_BLOCK / 64];
  float v = acc[0];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) means[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) / ((float)ps * (float)ps);
}

// patch DWT loss: grid.x = patch, grid.y = chunk of the patch's C*hp*hp level-1 blocks
template <bool BWD>
__global__ void __launch_bounds__(GS_BLOCK) patch_dwt_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                             int C, int H, int W, int ps, const uint8_t* __restrict__ mask,
                                                             float* sums, const float* __restrict__ coef,
                                                             float* __restrict__ grad) {
  const int patch = blockIdx.x;
  if (!mask[patch]) return;
  const int nx = W / ps, hp = cdiv2(ps);
  const int px = patch % nx, py = patch / nx;
  const int total = C * hp * hp;
  float s[3] = {0.f, 0.f, 0.f};
  float cf0 = 0.f, cf1 = 0.f, cf2 = 0.f;
  if (BWD) { cf0 = coef[0]; cf1 = coef[1]; cf2 = coef[2]; }
  for (int o = blockIdx.y * GS_BLOCK + threadIdx.x; o < total; o += gridDim.y * GS_BLOCK) {
    const int j = o % hp, i = (o / hp) % hp, c = o / (hp * hp);
    const int y0 = py * ps + 2 * i, x0 = px * ps + 2 * j;
    const bool pyb = 2 * i + 1 >= ps, pxb = 2 * j + 1 >= ps;
    const int y1 = pyb ? y0 : y0 + 1, x1 = pxb ? x0 : x0 + 1;
    const float* p = pred + (size_t)c * H * W;
    const float* q = gt + (size_t)c * H * W;
    const Bands a = haar_block(p[(size_t)y0 * W + x0], p[(size_t)y0 * W + x1], p[(size_t)y1 * W + x0], p[(size_t)y1 * W + x1]);
    const Bands b = haar_block(q[(size_t)y0 * W + x0], q[(size_t)y0 * W + x1], q[(size_t)y1 * W + x0], q[(size_t)y1 * W + x1]);
    if (!BWD) {
      s[0] += fabsf(a.lh - b.lh);
      s[1] += fabsf(a.hl - b.hl);
      s[2] += fabsf(a.hh - b.hh);
    } else {
      float da, db, dc, dd;
      haar_block_adj(0.f, cf0 * sgnf(a.lh - b.lh), cf1 * sgnf(a.hl - b.hl), cf2 * sgnf(a.hh - b.hh), da, db, dc, dd);
      float* g = grad + (size_t)c * H * W;
      float v00 = da, v01 = db, v10 = dc, v11 = dd;
      if (pyb && pxb) v00 = ((da + db) + dc) + dd;
      else if (pyb) { v00 = da + dc; v01 = db + dd; }
      else if (pxb) { v00 = da + db; v10 = dc + dd; }
      g[(size_t)y0 * W + x0] += v00;
      if (!pxb) g[(size_t)y0 * W + x1] += v01;
      if (!pyb) { g[(size_t)y1 * W + x0] += v10; if (!pxb) g[(size_t)y1 * W + x1] += v11; }
    }
  }
  if (!BWD) block_sum_atomic<3>(s, sums);
}

// ------------------------------------------------------------------------------------------------ SSIM
// 32x32 output tile per 256-thread workgroup; 42x42 input halo tile in LDS, horizontal pass to LDS,
// vertical pass from LDS (both register-blocked, see ssim_conv_tile).  fused-ssim/ssim.cu:9-19:
__constant__ float GW[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                             0.10936068743467331f,  0.21300552785396576f,  0.26601171493530273f,
                             0.21300552785396576f,  0.10936068743467331f,  0.036000773310661316f,
                             0.0075987582094967365f, 0.001028380123898387f};
#define ST 32
#define SH (ST + 10)

#define HS 33  // row stride of the horizontal-pass buffer (odd): the 32 lanes of a store group (8 column groups x 4 rows)
               // and of a vertical-pass load group (one row) each fall on 32 different banks

// Workgroup -> tile mapping of the SSIM kernels.  The hardware deals consecutive workgroups out to the 8 XCDs in turn, each
// with its own L2; with the natural (x, y, plane) order horizontally adjacent tiles - which share 16 of a halo row's 48
// floats - land on different XCDs and every XCD fetches its own copy of the shared lines (PMC: 2x the bytes the tiles
// span).  Here launch slot b belongs to XCD b % 8 and XCD k works through the k-th contiguous eighth of the tiles in raster
// order: neighbours run on the same XCD at about the same time and meet in its L2.
struct SsimTile { int bx, by, z, linear; bool live; };
__device__ __forceinline__ SsimTile ssim_tile_of_block(int gx, int gy, int gz) {
  const int n = gx * gy * gz, per = (n + 7) / 8;
  const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  SsimTile r;
  r.live = (int)(blockIdx.x >> 3) < per && t < n;
  r.linear = t;
  r.z = t / (gx * gy);
  const int rem = t - r.z * gx * gy;
  r.by = (rem / gx) * ST;
  r.bx = (rem % gx) * ST;
  return r;
}
static inline unsigned ssim_launch_blocks(int W, int H, int planes) {
  const int n = ((W + ST - 1) / ST) * ((H + ST - 1) / ST) * planes;
  return (unsigned)(((n + 7) / 8) * 8);
}

// Separable 11-tap convolution of NQ quantities over the 42x42 halo tile.  Register-blocked: a thread produces 4
// adjacent outputs along the pass direction from 14 inputs (sliding window), i.e. 3.5 LDS reads per output and
// quantity instead of 11; in the forward the five quantities (a, a^2, b, b^2, ab) are formed from the two image
// planes on the fly, so only those two are staged.  Every output is still the sum over t = 0..10 in ascending
// order (the order of fused-ssim/ssim.cu:60-100 and of the checker).  out[j] belongs to tile row 4 * (tid / 32) + j,
// column tid % 32.
// `tile` and `hor` SHARE their LDS (the horizontal results are held in registers across a barrier and then overwrite
// the input tile): 27.7 KB per workgroup in the forward instead of 48 KB, i.e. 5 workgroups per CU instead of 3.
template <int NQ, bool FWD>
__device__ __forceinline__ void ssim_conv_tile(const float (*tile)[SH][SH + 1], float (*hor)[SH][HS], float (&out)[4][NQ]) {
  // the taps contract to FMAs here although the file is built with -ffp-contract=off: nvcc contracts the reference's
  // `sum += G * val` (fused-ssim/ssim.cu:60-100) the same way, and the kernel is VALU-issue-bound (half the instructions)
#pragma clang fp contract(fast)
  constexpr int NITEM = SH * (ST / 4);                      // 336 (row, 4-column group) items
  constexpr int NR = (NITEM + GS_BLOCK - 1) / GS_BLOCK;     // 2 rounds
  float acc[NR][4][NQ];
#pragma unroll
  for (int rd = 0; rd < NR; rd++) {
    const int item = threadIdx.x + rd * GS_BLOCK;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < NQ; q++) acc[rd][j][q] = 0.f;
    if (item < NITEM) {
      const int r = item / (ST / 4), c0 = (item % (ST / 4)) * 4;
#pragma unroll
      for (int e = 0; e < 14; e++) {
        float val[NQ];
        if (FWD) {
          const float a = tile[0][r][c0 + e], b = tile[1][r][c0 + e];
          val[0] = a; val[1] = a * a; val[2] = b; val[3 % NQ] = b * b; val[4 % NQ] = a * b;
        } else {
#pragma unroll
          for (int q = 0; q < NQ; q++) val[q] = tile[q][r][c0 + e];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int t = e - j;
          if (t >= 0 && t < 11) {
#pragma unroll
            for (int q = 0; q < NQ; q++) acc[rd][j][q] += GW[t] * val[q];
          }
        }
      }
    }
  }
  __syncthreads();  // every read of `tile` is done: `hor` may overwrite it
#pragma unroll
  for (int rd = 0; rd < NR; rd++) {
    const int item = threadIdx.x + rd * GS_BLOCK;
    if (item < NITEM) {
      const int r = item / (ST / 4), c0 = (item % (ST / 4)) * 4;
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < NQ; q++) hor[q][r][c0 + j] = acc[rd][j][q];
    }
  }
  __syncthreads();
  const int lx = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 4;
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int q = 0; q < NQ; q++) out[j][q] = 0.f;
#pragma unroll
  for (int e = 0; e < 14; e++) {
    float val[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) val[q] = hor[q][r0 + e][lx];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int t = e - j;
      if (t >= 0 && t < 11) {
#pragma unroll
        for (int q = 0; q < NQ; q++) out[j][q] += GW[t] * val[q];
      }
    }
  }
}

__global__ void __launch_bounds__(GS_BLOCK) ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                            int H, int W, float C1, float C2, float* __restrict__ ssim_map,
                                                            float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                            float* __restrict__ dm_dsigma12, float* sum_out, float* __restrict__ partials,
                                                            int planes) {
  constexpr int LDS_WORDS = (2 * SH * (SH + 1) > 5 * SH * HS) ? 2 * SH * (SH + 1) : 5 * SH * HS;
  __shared__ float lds_buf[LDS_WORDS];
  float (*tile)[SH][SH + 1] = reinterpret_cast<float (*)[SH][SH + 1]>(lds_buf);
  float (*hor)[SH][HS] = reinterpret_cast<float (*)[SH][HS]>(lds_buf);
  const SsimTile tl = ssim_tile_of_block((W + ST - 1) / ST, (H + ST - 1) / ST, planes);
  if (!tl.live) return;
  const size_t plane = (size_t)tl.z * H * W;
  const int bx = tl.bx, by = tl.by;
  // Halo tile loads.  Rows are 42 floats starting at column bx - 5: with W a multiple of 4 (and so every plane and row
  // 16-byte aligned) the aligned span [bx - 8, bx + 40) is fetched as 12 float4 per row - 2 vector loads per thread and
  // plane instead of 7 scalar ones (a float4 lies entirely inside or entirely outside the image); else element-wise.
  const bool vec = (W & 3) == 0 && ((((uintptr_t)img1) | ((uintptr_t)img2)) & 15) == 0;
  if (vec) {
    constexpr int VPR = 12, NV = SH * VPR, NVI = (NV + GS_BLOCK - 1) / GS_BLOCK;
    float4 a[NVI], b[NVI];
#pragma unroll
    for (int it = 0; it < NVI; it++) {
      const int v = threadIdx.x + it * GS_BLOCK;
      const int r = v / VPR, q = v % VPR;
      const int y = by + r - 5, x4 = bx - 8 + 4 * q;
      a[it] = b[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < NV && y >= 0 && y < H && x4 >= 0 && x4 + 3 < W) {
        a[it] = *reinterpret_cast<const float4*>(img1 + plane + (size_t)y * W + x4);
        b[it] = *reinterpret_cast<const float4*>(img2 + plane + (size_t)y * W + x4);
      }
    }
#pragma unroll
    for (int it = 0; it < NVI; it++) {
      const int v = threadIdx.x + it * GS_BLOCK;
      if (v < NV) {
        const int r = v / VPR, c0 = 4 * (v % VPR) - 3;  // tile column of the vector's first element
        const float av[4] = {a[it].x, a[it].y, a[it].z, a[it].w}, bv[4] = {b[it].x, b[it].y, b[it].z, b[it].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int c = c0 + e;
          if (c >= 0 && c < SH) {
            tile[0][r][c] = av[e];
            tile[1][r][c] = bv[e];
          }
        }
      }
    }
  } else {  // all global loads of the halo tile are in flight before the first LDS store (a rolled loop waits per round)
    constexpr int NIT = (SH * SH + GS_BLOCK - 1) / GS_BLOCK;
    float a[NIT], b[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int k = threadIdx.x + it * GS_BLOCK;
      const int r = k / SH, c = k % SH;
      const int y = by + r - 5, x = bx + c - 5;
      a[it] = 0.f;
      b[it] = 0.f;
      if (k < SH * SH && x >= 0 && x < W && y >= 0 && y < H) {
        a[it] = img1[plane + (size_t)y * W + x];
        b[it] = img2[plane + (size_t)y * W + x];
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int k = threadIdx.x + it * GS_BLOCK;
      if (k < SH * SH) {
        tile[0][k / SH][k % SH] = a[it];
        tile[1][k / SH][k % SH] = b[it];
      }
    }
  }
  __syncthreads();
  float out[4][5];
  ssim_conv_tile<5, true>(tile, hor, out);
  const int lx = threadIdx.x & 31, ly0 = threadIdx.x >> 5;
  float msum = 0.f;
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int x = bx + lx, y = by + 4 * ly0 + m;
    if (x >= W || y >= H) continue;
    const float mu1 = out[m][0], mu2 = out[m][2];
    const float sigma1_sq = out[m][1] - mu1 * mu1;
    const float sigma2_sq = out[m][3] - mu2 * mu2;
    const float sigma12 = out[m][4] - mu1 * mu2;
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
    const float Cc = (2.0f * mu1_mu2 + C1);
    const float D = (2.0f * sigma12 + C2);
    const float A = (mu1_sq + mu2_sq + C1);
    const float B = (sigma1_sq + sigma2_sq + C2);
    const size_t o = plane + (size_t)y * W + x;
    // fused-ssim/ssim.cu:140-160 with ONE division: 1/(AB), 1/A = B/(AB), 1/B = A/(AB) (the reference divides seven
    // times per pixel; an IEEE division is ~10 VALU instructions and this kernel is issue-bound) - a few ulp apart
    const float inv_AB = 1.0f / (A * B);
    const float inv_A = B * inv_AB, inv_B = A * inv_AB;
    const float mval = (Cc * D) * inv_AB;
    msum += mval;
    if (ssim_map) ssim_map[o] = mval;
    if (dm_dmu1) {
      dm_dmu1[o] = (mu2 * 2.0f * D) * inv_AB - (mu2 * 2.0f * Cc) * inv_AB - (mu1 * 2.0f * Cc * D) * inv_AB * inv_A +
                   (mu1 * 2.0f * Cc * D) * inv_AB * inv_B;
      dm_dsigma1_sq[o] = (-Cc * D) * inv_AB * inv_B;
      dm_dsigma12[o] = (2 * Cc) * inv_AB;
    }
  }
  if (partials) {
    // one plain store per workgroup; gs_lgdwt_combine_p adds them up.  (An atomic on ONE address per workgroup is
    // serialised device-wide at ~14 ns each on this 8-XCD part: 6120 of them took 86 us of a 98-us kernel.)
    __shared__ float red[GS_BLOCK / 64];
    float x = msum;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0)
      partials[tl.linear] = (red[0] + red[1]) + (red[2] + red[3]);
  } else if (sum_out) {  // uniform branch: mean SSIM without a second pass over the map
    float acc[1] = {msum};
    __syncthreads();
    block_sum_atomic<1>(acc, sum_out);
  }
}

__global__ void __launch_bounds__(GS_BLOCK) ssim_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                            int H, int W, const float* __restrict__ dL_dmap,
                                                            const float* __restrict__ dm_dmu1,
                                                            const float* __restrict__ dm_dsigma1_sq,
                                                            const float* __restrict__ dm_dsigma12,
                                                            float* __restrict__ dL_dimg1, const float* __restrict__ coef_dev,
                                                            int accumulate, const float* __restrict__ clamp_src, int planes) {
  constexpr int LDS_WORDS = (3 * SH * (SH + 1) > 3 * SH * HS) ? 3 * SH * (SH + 1) : 3 * SH * HS;
  __shared__ float lds_buf[LDS_WORDS];
  float (*tile)[SH][SH + 1] = reinterpret_cast<float (*)[SH][SH + 1]>(lds_buf);
  float (*hor)[SH][HS] = reinterpret_cast<float (*)[SH][HS]>(lds_buf);
  const float gu = coef_dev ? coef_dev[0] : 0.f;  // uniform dL/dssim_map (mean reduction upstream)
  const SsimTile tl = ssim_tile_of_block((W + ST - 1) / ST, (H + ST - 1) / ST, planes);
  if (!tl.live) return;
  const size_t plane = (size_t)tl.z * H * W;
  const int bx = tl.bx, by = tl.by;
  const bool vec = (W & 3) == 0 && !dL_dmap &&
                   ((((uintptr_t)dm_dmu1) | ((uintptr_t)dm_dsigma1_sq) | ((uintptr_t)dm_dsigma12)) & 15) == 0;
  if (vec) {  // (see ssim_fwd_kernel; uniform upstream gradient only - the train step's case)
    constexpr int VPR = 12, NV = SH * VPR, NVI = (NV + GS_BLOCK - 1) / GS_BLOCK;
    float4 a[NVI], b[NVI], d[NVI];
#pragma unroll
    for (int it = 0; it < NVI; it++) {
      const int v = threadIdx.x + it * GS_BLOCK;
      const int r = v / VPR, q = v % VPR;
      const int y = by + r - 5, x4 = bx - 8 + 4 * q;
      a[it] = b[it] = d[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < NV && y >= 0 && y < H && x4 >= 0 && x4 + 3 < W) {
        const size_t o = plane + (size_t)y * W + x4;
        a[it] = *reinterpret_cast<const float4*>(dm_dmu1 + o);
        b[it] = *reinterpret_cast<const float4*>(dm_dsigma1_sq + o);
        d[it] = *reinterpret_cast<const float4*>(dm_dsigma12 + o);
      }
    }
#pragma unroll
    for (int it = 0; it < NVI; it++) {
      const int v = threadIdx.x + it * GS_BLOCK;
      if (v < NV) {
        const int r = v / VPR, c0 = 4 * (v % VPR) - 3;
        const float av[4] = {a[it].x, a[it].y, a[it].z, a[it].w}, bv[4] = {b[it].x, b[it].y, b[it].z, b[it].w};
        const float dv[4] = {d[it].x, d[it].y, d[it].z, d[it].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int c = c0 + e;
          if (c >= 0 && c < SH) {
            tile[0][r][c] = av[e] * gu;
            tile[1][r][c] = bv[e] * gu;
            tile[2][r][c] = dv[e] * gu;
          }
        }
      }
    }
  } else {
    constexpr int NIT = (SH * SH + GS_BLOCK - 1) / GS_BLOCK;
    float g[NIT], a[NIT], b[NIT], d[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int k = threadIdx.x + it * GS_BLOCK;
      const int r = k / SH, c = k % SH;
      const int y = by + r - 5, x = bx + c - 5;
      g[it] = a[it] = b[it] = d[it] = 0.f;
      if (k < SH * SH && x >= 0 && x < W && y >= 0 && y < H) {
        const size_t o = plane + (size_t)y * W + x;
        g[it] = dL_dmap ? dL_dmap[o] : gu;
        a[it] = dm_dmu1[o];
        b[it] = dm_dsigma1_sq[o];
        d[it] = dm_dsigma12[o];
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int k = threadIdx.x + it * GS_BLOCK;
      if (k < SH * SH) {
        tile[0][k / SH][k % SH] = a[it] * g[it];
        tile[1][k / SH][k % SH] = b[it] * g[it];
        tile[2][k / SH][k % SH] = d[it] * g[it];
      }
    }
  }
  __syncthreads();
  float out[4][3];
  ssim_conv_tile<3, false>(tile, hor, out);
  const int lx = threadIdx.x & 31, ly0 = threadIdx.x >> 5;
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int x = bx + lx, y = by + 4 * ly0 + m;
    if (x >= W || y >= H) continue;
    const size_t o = plane + (size_t)y * W + x;
    // with clamp_src, img1 IS clamp(clamp_src, 0, 1) (gsplat.h): formed here instead of read (25 MB less at 1080p)
    const float r = clamp_src ? clamp_src[o] : 0.f;
    const float a1 = clamp_src ? (r < 0.f ? 0.f : (r > 1.f ? 1.f : r)) : img1[o];
    float dL = 0.0f;
    dL += out[m][0];
    dL += a1 * 2.0f * out[m][1];
    dL += img2[o] * out[m][2];
    if (accumulate) dL += dL_dimg1[o];
    // gradient of clamp(x, 0, 1) folded in: zero where the un-clamped source was outside [0, 1]
    if (clamp_src && (r < 0.f || r > 1.f)) dL = 0.f;
    dL_dimg1[o] = dL;
  }
}

// ------------------------------------------------------------------------------------------------ entry points
static inline int nblocks(int64_t n, int per = GS_BLOCK, int cap = 8192) {
  int64_t b = (n + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

extern "C" {

int gs_l1_fwd(const float* a, const float* b, int64_t n, float* sum, void* stream) {
  if (!a || !b || !sum) return GS_E_NULL;
  if (n <= 0) return GS_OK;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_L1, s);
  if ((((uintptr_t)a | (uintptr_t)b) & 15) != 0) return GS_E_UNSUPPORTED;  // torch allocations are 256-B aligned
  hipLaunchKernelGGL(l1_fwd_kernel, dim3(nblocks(n / 4 + 1, GS_BLOCK, 512)), dim3(GS_BLOCK), 0, s, a, b, n, sum, (float*)nullptr);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int64_t gs_l1_partials_count(int64_t n) { return n <= 0 ? 0 : (int64_t)nblocks(n / 4 + 1, GS_BLOCK, 512); }
int gs_l1_fwd_p(const float* a, const float* b, int64_t n, float* partials, void* stream) {
  if (!a || !b || !partials) return GS_E_NULL;
  if (n <= 0) return GS_OK;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_L1, s);
  if ((((uintptr_t)a | (uintptr_t)b) & 15) != 0) return GS_E_UNSUPPORTED;
  hipLaunchKernelGGL(l1_fwd_kernel, dim3(nblocks(n / 4 + 1, GS_BLOCK, 512)), dim3(GS_BLOCK), 0, s, a, b, n, (float*)nullptr, partials);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int64_t gs_depth_l1_partials_count(int64_t n) { return n <= 0 ? 0 : (int64_t)nblocks(n, 4 * GS_BLOCK, 1024); }
int gs_depth_l1(const float* invdepth, const float* mono, const float* mask, int64_t n, float* partials, float coef,
                const float* coef_dev, float* grad, void* stream) {
  if (!invdepth || !mono || (!partials && !grad)) return GS_E_NULL;
  if (n <= 0) return GS_OK;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_L1, s);
  hipLaunchKernelGGL(depth_l1_kernel, dim3(nblocks(n, 4 * GS_BLOCK, 1024)), dim3(GS_BLOCK), 0, s, invdepth, mono, mask, n, partials,
                     coef, coef_dev, grad);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_l1_bwd(const float* a, const float* b, int64_t n, float coef, float* g, int32_t accumulate, void* stream) {
  if (!a || !b || !g) return GS_E_NULL;
  if (n <= 0) return GS_OK;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_L1, s);
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(nblocks(n)), dim3(GS_BLOCK), 0, s, a, b, n, coef, (const float*)nullptr, g, accumulate);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_dwt_haar_fwd(const float* x, int32_t NC, int32_t H, int32_t W, float* ll, float* lh, float* hl, float* hh,
                    void* stream) {
  if (!x) return GS_E_NULL;
  if (NC <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT1, s);
  const int64_t total = (int64_t)NC * ((H + 1) / 2) * ((W + 1) / 2);
  hipLaunchKernelGGL(dwt_fwd_kernel, dim3(nblocks(total)), dim3(GS_BLOCK), 0, s, x, NC, H, W, ll, lh, hl, hh);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_dwt_haar_bwd(const float* dll, const float* dlh, const float* dhl, const float* dhh, int32_t NC, int32_t H,
                    int32_t W, float* dx, void* stream) {
  if (!dx) return GS_E_NULL;
  if (NC <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT1, s);
  const int64_t total = (int64_t)NC * ((H + 1) / 2) * ((W + 1) / 2);
  hipLaunchKernelGGL(dwt_bwd_kernel, dim3(nblocks(total)), dim3(GS_BLOCK), 0, s, dll, dlh, dhl, dhh, NC, H, W, dx);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
static inline bool dwt2_fast(const void* a, const void* b, const void* g, int H, int W) {
  return (H % 4) == 0 && (W % 4) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)g) & 15) == 0);
}
static inline unsigned dwt2_fwd_workgroups(int32_t C, int32_t H, int32_t W) {
  const int h2 = ((H + 1) / 2 + 1) / 2, w2 = ((W + 1) / 2 + 1) / 2;
  return (unsigned)nblocks((int64_t)C * h2 * w2, GS_BLOCK, 512);
}
static int dwt2_l1_fwd_launch(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, float* band_sums,
                              float* l1_sum, hipStream_t s, float* clamped_out = nullptr, const uint8_t* patch_mask = nullptr,
                              int ps = 0, float* patch_sums = nullptr, float* partials = nullptr) {
  const int h2 = ((H + 1) / 2 + 1) / 2, w2 = ((W + 1) / 2 + 1) / 2;
  // few, long-running workgroups: every workgroup ends in nine atomics on one cache line, and at 1500 workgroups
  // (1080p) their serialisation was 40 % of the kernel (25.9 us at a 4096 cap, 15.1 us at 512)
  const dim3 grid(nblocks((int64_t)C * h2 * w2, GS_BLOCK, 512));
  if (dwt2_fast(pred, gt, clamped_out, H, W))
    hipLaunchKernelGGL(dwt2_l1_fwd_kernel<true>, grid, dim3(GS_BLOCK), 0, s, pred, gt, C, H, W, band_sums, l1_sum, clamped_out,
                       patch_mask, ps, patch_sums, partials);
  else if (clamped_out || patch_mask || partials)
    return GS_E_UNSUPPORTED;  // (H, W multiples of 4 and 16-byte aligned planes only: clamp with torch otherwise)
  else
    hipLaunchKernelGGL(dwt2_l1_fwd_kernel<false>, grid, dim3(GS_BLOCK), 0, s, pred, gt, C, H, W, band_sums, l1_sum,
                       (float*)nullptr, (const uint8_t*)nullptr, 0, (float*)nullptr, (float*)nullptr);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
static int dwt2_l1_bwd_launch(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, const float* coef_dev,
                              const float* l1_coef_dev, float* grad_pred, int32_t accumulate, hipStream_t s,
                              const uint8_t* patch_mask = nullptr, int ps = 0, const float* patch_coef = nullptr) {
  const int h2 = ((H + 1) / 2 + 1) / 2, w2 = ((W + 1) / 2 + 1) / 2;
  const dim3 grid(nblocks((int64_t)C * h2 * w2));
  if (dwt2_fast(pred, gt, grad_pred, H, W))
    hipLaunchKernelGGL(dwt2_l1_bwd_kernel<true>, grid, dim3(GS_BLOCK), 0, s, pred, gt, C, H, W, coef_dev, l1_coef_dev,
                       grad_pred, accumulate, patch_mask, ps, patch_coef);
  else if (patch_mask)
    return GS_E_UNSUPPORTED;
  else
    hipLaunchKernelGGL(dwt2_l1_bwd_kernel<false>, grid, dim3(GS_BLOCK), 0, s, pred, gt, C, H, W, coef_dev, l1_coef_dev,
                       grad_pred, accumulate, (const uint8_t*)nullptr, 0, (const float*)nullptr);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_dwt2_l1_fwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, float* band_sums, void* stream) {
  if (!pred || !gt || !band_sums) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT2_FWD, s);
  return dwt2_l1_fwd_launch(pred, gt, C, H, W, band_sums, nullptr, s);
}
int gs_dwt2_l1_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, const float* coef_dev,
                   float* grad_pred, int32_t accumulate, void* stream) {
  if (!pred || !gt || !coef_dev || !grad_pred) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT2_BWD, s);
  return dwt2_l1_bwd_launch(pred, gt, C, H, W, coef_dev, nullptr, grad_pred, accumulate, s);
}
int gs_l1_dwt2_fwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, float* l1_sum, float* band_sums,
                   void* stream) {
  if (!pred || !gt || !band_sums || !l1_sum) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT2_FWD, s);
  return dwt2_l1_fwd_launch(pred, gt, C, H, W, band_sums, l1_sum, s);
}
int gs_l1_dwt2_fwd_clamp(const float* raw, const float* gt, int32_t C, int32_t H, int32_t W, float* l1_sum, float* band_sums,
                         float* clamped_out, void* stream) {
  if (!raw || !gt || !band_sums || !l1_sum || !clamped_out) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT2_FWD, s);
  return dwt2_l1_fwd_launch(raw, gt, C, H, W, band_sums, l1_sum, s, clamped_out);
}
int gs_l1_dwt2_patch_fwd_clamp(const float* raw, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps, const uint8_t* mask,
                               float* l1_sum, float* band_sums, float* patch_sums, float* clamped_out, void* stream) {
  if (!raw || !gt || !band_sums || !l1_sum || !clamped_out || !mask || !patch_sums) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0 || ps <= 0 || H < ps || W < ps) return GS_E_SHAPE;
  if (ps % 4 != 0) return GS_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT2_FWD, s);
  return dwt2_l1_fwd_launch(raw, gt, C, H, W, band_sums, l1_sum, s, clamped_out, mask, ps, patch_sums);
}
int64_t gs_dwt_partials_count(int32_t C, int32_t H, int32_t W) {
  return (C <= 0 || H <= 0 || W <= 0) ? 0 : (int64_t)dwt2_fwd_workgroups(C, H, W);
}
int gs_l1_dwt2_patch_fwd_clamp_p(const float* raw, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps, const uint8_t* mask,
                                 float* partials, float* clamped_out, void* stream) {
  if (!raw || !gt || !partials || !clamped_out) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0 || ps < 0) return GS_E_SHAPE;
  if ((ps > 0) != (mask != nullptr)) return GS_E_NULL;
  if (ps > 0 && (H < ps || W < ps)) return GS_E_SHAPE;
  if (ps % 4 != 0) return GS_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT2_FWD, s);
  float dummy_target = 0.f;  // (never dereferenced with partials given; the kernel only tests l1_sum for NULL)
  return dwt2_l1_fwd_launch(raw, gt, C, H, W, &dummy_target, &dummy_target, s, clamped_out, mask, ps, &dummy_target, partials);
}
int gs_l1_dwt2_patch_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps, const uint8_t* mask,
                         const float* l1_coef_dev, const float* coef_dev, const float* patch_coef_dev, float* grad_pred,
                         int32_t accumulate, void* stream) {
  if (!pred || !gt || !coef_dev || !l1_coef_dev || !grad_pred || !mask || !patch_coef_dev) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0 || ps <= 0 || H < ps || W < ps) return GS_E_SHAPE;
  if (ps % 4 != 0) return GS_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT2_BWD, s);
  return dwt2_l1_bwd_launch(pred, gt, C, H, W, coef_dev, l1_coef_dev, grad_pred, accumulate, s, mask, ps, patch_coef_dev);
}
int gs_l1_dwt2_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, const float* l1_coef_dev,
                   const float* coef_dev, float* grad_pred, int32_t accumulate, void* stream) {
  if (!pred || !gt || !coef_dev || !l1_coef_dev || !grad_pred) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_DWT2_BWD, s);
  return dwt2_l1_bwd_launch(pred, gt, C, H, W, coef_dev, l1_coef_dev, grad_pred, accumulate, s);
}
int gs_elf_map(const float* img, int32_t C, int32_t H, int32_t W, float* elf_low, float* elf, void* stream) {
  if (!img || !elf || !elf_low) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_ELF, s);
  const int h = (H + 1) / 2, w = (W + 1) / 2;
  hipLaunchKernelGGL(elf_low_kernel, dim3((h * w + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, img, C, H, W, elf_low);
  hipLaunchKernelGGL(bilinear_up_kernel, dim3((H * W + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, elf_low, h, w, H, W, elf);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_patch_means(const float* elf, int32_t H, int32_t W, int32_t ps, float* means, void* stream) {
  if (!elf || !means) return GS_E_NULL;
  if (ps <= 0 || H < ps || W < ps) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_ELF, s);
  hipLaunchKernelGGL(patch_means_kernel, dim3((H / ps) * (W / ps)), dim3(GS_BLOCK), 0, s, elf, H, W, ps, means);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_patch_dwt_fwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps, const uint8_t* mask,
                     float* sums, void* stream) {
  if (!pred || !gt || !mask || !sums) return GS_E_NULL;
  if (ps <= 0 || H < ps || W < ps || C <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_PATCH, s);
  const int hp = (ps + 1) / 2;
  const int chunks = nblocks((int64_t)C * hp * hp, GS_BLOCK * 4, 64);
  hipLaunchKernelGGL(patch_dwt_kernel<false>, dim3((H / ps) * (W / ps), chunks), dim3(GS_BLOCK), 0, s, pred, gt, C, H, W, ps,
                     mask, sums, (const float*)nullptr, (float*)nullptr);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_patch_dwt_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps, const uint8_t* mask,
                     const float* coef_dev, float* grad_pred, int32_t accumulate, void* stream) {
  if (!pred || !gt || !mask || !coef_dev || !grad_pred) return GS_E_NULL;
  if (ps <= 0 || H < ps || W < ps || C <= 0) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_PATCH, s);
  if (!accumulate) GS_HIP_CHECK(hipMemsetAsync(grad_pred, 0, sizeof(float) * (size_t)C * H * W, s));
  const int hp = (ps + 1) / 2;
  const int chunks = nblocks((int64_t)C * hp * hp, GS_BLOCK * 4, 64);
  hipLaunchKernelGGL(patch_dwt_kernel<true>, dim3((H / ps) * (W / ps), chunks), dim3(GS_BLOCK), 0, s, pred, gt, C, H, W, ps,
                     mask, (float*)nullptr, coef_dev, grad_pred);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_ssim_fwd(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float C1, float C2,
                float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
  if (!img1 || !img2 || !ssim_map) return GS_E_NULL;
  if (dm_dmu1 && (!dm_dsigma1_sq || !dm_dsigma12)) return GS_E_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (int64_t)B * C > 65535) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_SSIM_FWD, s);
  hipLaunchKernelGGL(ssim_fwd_kernel, dim3(ssim_launch_blocks(W, H, B * C)), dim3(GS_BLOCK), 0, s, img1, img2, H, W,
                     C1, C2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, (float*)nullptr, (float*)nullptr, B * C);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_ssim_bwd(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float, float,
                const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                float* dL_dimg1, void* stream) {
  if (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1) return GS_E_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (int64_t)B * C > 65535) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_SSIM_BWD, s);
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3(ssim_launch_blocks(W, H, B * C)), dim3(GS_BLOCK), 0, s, img1, img2, H, W,
                     dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1, (const float*)nullptr, 0, (const float*)nullptr, B * C);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}

// ------------------------------------------------------------------------------------------------ fused criterion
// Entry points used by the one-Function form of the LGDWT-GS loss (gsplat_amd/losses.py::FusedLGDWTLoss): the
// per-term kernels above, but (1) coefficients come from DEVICE memory (no host sync on the running-mean DWT
// scale), (2) every term accumulates into ONE image-gradient buffer, (3) SSIM's mean and the clamp(0,1) backward
// are folded in, so no torch elementwise pass touches an image.

int gs_l1_bwd_dev(const float* a, const float* b, int64_t n, const float* coef_dev, float* grad_a, int32_t accumulate,
                  void* stream) {
  if (!a || !b || !grad_a || !coef_dev) return GS_E_NULL;
  if (n <= 0) return GS_OK;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_L1, s);
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(nblocks(n)), dim3(GS_BLOCK), 0, s, a, b, n, 1.0f, coef_dev, grad_a, accumulate);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}

int gs_ssim_fwd_sum(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float C1, float C2,
                    float* sum_out, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
  if (!img1 || !img2 || !sum_out) return GS_E_NULL;
  if (dm_dmu1 && (!dm_dsigma1_sq || !dm_dsigma12)) return GS_E_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (int64_t)B * C > 65535) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_SSIM_FWD, s);
  hipLaunchKernelGGL(ssim_fwd_kernel, dim3(ssim_launch_blocks(W, H, B * C)), dim3(GS_BLOCK), 0, s, img1, img2, H, W,
                     C1, C2, (float*)nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, sum_out, (float*)nullptr, B * C);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}

int64_t gs_ssim_partials_count(int32_t B, int32_t C, int32_t H, int32_t W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return (int64_t)((W + ST - 1) / ST) * ((H + ST - 1) / ST) * B * C;
}

int gs_ssim_fwd_partials(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float C1, float C2,
                         float* partials, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
  if (!img1 || !img2 || !partials) return GS_E_NULL;
  if (dm_dmu1 && (!dm_dsigma1_sq || !dm_dsigma12)) return GS_E_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (int64_t)B * C > 65535) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_SSIM_FWD, s);
  hipLaunchKernelGGL(ssim_fwd_kernel, dim3(ssim_launch_blocks(W, H, B * C)), dim3(GS_BLOCK), 0, s, img1, img2, H, W,
                     C1, C2, (float*)nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, (float*)nullptr, partials, B * C);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}

int gs_ssim_bwd_uniform(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W,
                        const float* coef_dev, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                        float* dL_dimg1, int32_t accumulate, const float* clamp_src, void* stream) {
  if (!img1 || !img2 || !coef_dev || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1) return GS_E_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (int64_t)B * C > 65535) return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_SSIM_BWD, s);
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3(ssim_launch_blocks(W, H, B * C)), dim3(GS_BLOCK), 0, s, img1, img2, H, W,
                     (const float*)nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1, coef_dev, accumulate, clamp_src, B * C);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}

// loss composition of LGDWT-GS/train.py:188-202 on the device (one thread): the reference pulls base/dwt to the
// host with .item() every iteration; here the running mean lives in device memory and the coefficient vector of
// the backward kernels is produced in the same launch.
//   sums[16] : 0 l1_sum | 1 ssim_sum | 2..9 band_sums | 10..12 patch_sums | 13 n_selected_patches
//   out[24]  : 0 loss | 1 base | 2 dwt | 3 patch | 4 dwt_scale | 5 l1 | 6 ssim | 7 running mean before the call |
//              8 c_l1 | 9 c_ssim | 10..17 c_band[8] | 18..20 c_patch[3]      (dL/d term-sum, for upstream grad 1)
__global__ void __launch_bounds__(1024) lgdwt_combine_kernel(const float* __restrict__ sums_in, float* running_mean,
                                                             GsLgdwtParams p, float* __restrict__ out,
                                                             const float* __restrict__ ssim_partials, int n_partials,
                                                             const float* __restrict__ dwt_partials, int n_dwt,
                                                             const float* __restrict__ l1_partials, int n_l1) {
  // the per-workgroup partial sums of gs_l1_dwt2_patch_fwd_clamp_p (rows of 12: 8 bands, L1, 3 patch) and of gs_l1_fwd_p,
  // added in a fixed order (one wave per slot): the same total whatever order the workgroups finished in
  __shared__ float s_extra[13];
  {  // wave k (of the 16) owns slot k: its lanes add strided partials, then a fixed shuffle tree - the same shape every run
    const int slot = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (slot < 13) {
      float t = 0.f;
      if (slot < 12) {
        for (int w = lane; w < n_dwt; w += 64) t += dwt_partials[(size_t)w * 12 + slot];
      } else {
        for (int w = lane; w < n_l1; w += 64) t += l1_partials[w];
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) t += __shfl_down(t, off, 64);
      if (lane == 0) s_extra[slot] = t;
    }
  }
  __syncthreads();
  float sums[16];
#pragma unroll
  for (int k = 0; k < 16; k++) sums[k] = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 16; k++) sums[k] = sums_in[k];
    sums[0] += s_extra[8] + s_extra[12];
#pragma unroll
    for (int k = 0; k < 8; k++) sums[2 + k] += s_extra[k];
#pragma unroll
    for (int k = 0; k < 3; k++) sums[10 + k] += s_extra[9 + k];
  }
  // SSIM sum = sums[1] + the per-workgroup partials of gs_ssim_fwd_partials, added in a fixed order
  __shared__ float red[16];
  float part = 0.f;
  for (int i = threadIdx.x; i < n_partials; i += 1024) part += ssim_partials[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_down(part, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float l1 = sums[0] / p.n_pix;
  float ptot = 0.f;
  for (int w = 0; w < 16; w++) ptot += red[w];
  const float ssim = (sums[1] + ptot) / p.n_pix;
  const float w_l1 = p.custom_base ? p.w_l1 : 1.0f - p.lambda_dssim, w_ssim = p.custom_base ? p.w_ssim : p.lambda_dssim;
  const float base = w_l1 * l1 + w_ssim * (1.0f - ssim);
  float loss = base;
  float dwt = 0.f, scale = 0.f, patch = 0.f;
  for (int k = 0; k < 24; k++) out[k] = 0.f;
  out[7] = running_mean[0];  // the running mean BEFORE this view (a caller that has to take the view back restores it)
  if (p.dwt_enable) {
    for (int k = 0; k < 8; k++) {
      const float cnt = k < 4 ? p.n_band1 : p.n_band2;
      dwt += p.dwt_w[k] * (sums[2 + k] / cnt);
    }
    const float ratio = base / (dwt + 1e-8f);
    const float m = 0.95f * running_mean[0] + 0.05f * ratio;
    running_mean[0] = m;
    scale = fminf(fmaxf(m, 0.1f), 10.0f);
    loss = base + scale * dwt;
    for (int k = 0; k < 8; k++) out[10 + k] = scale * p.dwt_w[k] / (k < 4 ? p.n_band1 : p.n_band2);
  }
  if (p.patch_enable) {
    const float denom = fmaxf(sums[13] * p.patch_elems_per_sel, 1.0f);
    for (int k = 0; k < 3; k++) {
      patch += p.patch_w[k] * (sums[10 + k] / denom);
      out[18 + k] = p.patch_weight * p.patch_w[k] / denom;
    }
    loss = loss + p.patch_weight * patch;
  }
  out[0] = loss; out[1] = base; out[2] = dwt; out[3] = patch; out[4] = scale; out[5] = l1; out[6] = ssim;
  out[8] = w_l1 / p.n_pix;
  out[9] = -w_ssim / p.n_pix;
  if (p.reset_sums) {  // every sum has been read (this thread read them all): ready for the next view's accumulation
    float* z = const_cast<float*>(sums_in);
    for (int k = 0; k < 13; k++) z[k] = 0.f;
  }
}

int gs_lgdwt_combine_pp(const float* sums, const float* ssim_partials, int64_t n_partials, const float* dwt_partials,
                        int64_t n_dwt, const float* l1_partials, int64_t n_l1, float* running_mean, const GsLgdwtParams* p,
                        float* out, void* stream) {
  if (!sums || !running_mean || !p || !out) return GS_E_NULL;
  if (n_partials < 0 || n_partials > 0x7FFFFFFF || (n_partials > 0 && !ssim_partials)) return GS_E_SHAPE;
  if (n_dwt < 0 || n_dwt > 65536 || (n_dwt > 0 && !dwt_partials) || n_l1 < 0 || n_l1 > 65536 || (n_l1 > 0 && !l1_partials))
    return GS_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  hipLaunchKernelGGL(lgdwt_combine_kernel, dim3(1), dim3(1024), 0, s, sums, running_mean, *p, out, ssim_partials,
                     (int)n_partials, dwt_partials, (int)n_dwt, l1_partials, (int)n_l1);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
int gs_lgdwt_combine_p(const float* sums, const float* ssim_partials, int64_t n_partials, float* running_mean,
                       const GsLgdwtParams* p, float* out, void* stream) {
  return gs_lgdwt_combine_pp(sums, ssim_partials, n_partials, nullptr, 0, nullptr, 0, running_mean, p, out, stream);
}
int gs_lgdwt_combine(const float* sums, float* running_mean, const GsLgdwtParams* p, float* out, void* stream) {
  return gs_lgdwt_combine_p(sums, nullptr, 0, running_mean, p, out, stream);
}
}
