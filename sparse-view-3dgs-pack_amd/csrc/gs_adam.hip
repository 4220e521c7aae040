// gs_adam.hip - fused Adam over the flat parameter buffer (SURVEY.md 8f-1 "next" row).
//
// The reference steps six torch tensors with torch.optim.Adam(lr=0, eps=1e-15) and per-group learning rates
// (LGDWT-GS/scene/gaussian_model.py:183-193, train.py:279-288).  Here parameters, gradients and both moment
// buffers are each ONE flat fp32 array (59 floats per Gaussian); a launch updates all of it in one streaming
// pass (28 B/element: read p,g,m,v, write p,m,v), the learning rate being looked up from a small segment table.
// A segment may alternate between two rates with a period (the [P,16,3] SH block: 3 DC floats at feature_lr,
// 45 at feature_lr/20), which lets the model keep DC and rest coefficients interleaved as the rasterizer wants
// them - no torch.cat per step.
// Arithmetic = torch.optim.Adam (no amsgrad, no weight decay):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include <math.h>

#include "gs_common.h"
#include "gs_prof.h"

#define ADAM_MAX_SEG 8
struct AdamSegs {
  int n;
  long long begin[ADAM_MAX_SEG], end[ADAM_MAX_SEG];
  float lr_a[ADAM_MAX_SEG], lr_b[ADAM_MAX_SEG];
  int period[ADAM_MAX_SEG], split[ADAM_MAX_SEG];
  float inv_bc1[ADAM_MAX_SEG], inv_sqrt_bc2[ADAM_MAX_SEG];  // bias corrections of the segment's own step count
  int row_width[ADAM_MAX_SEG];                              // gs_adam_step_masked: floats per Gaussian (0: not masked)
};
// gs_adam_step_masked ("sparse_adam", train.py:282-284): is element i one of a row that is stepped?
__device__ __forceinline__ bool adam_row_on(const AdamSegs& s, int k, long long i, const float* __restrict__ row_mask) {
  if (!row_mask || s.row_width[k] <= 0) return true;
  return row_mask[(i - s.begin[k]) / s.row_width[k]] > 0.f;
}

// step size lr / (1 - b1^t) and 1 / sqrt(1 - b2^t) of element i; false = covered by no segment (left untouched)
__device__ __forceinline__ bool adam_coef(const AdamSegs& s, long long i, float& lr_bc1, float& inv_sqrt_bc2,
                                          const float* __restrict__ row_mask = nullptr) {
  bool hit = false;
#pragma unroll
  for (int k = 0; k < ADAM_MAX_SEG; k++)
    if (k < s.n && i >= s.begin[k] && i < s.end[k]) {
      const float lr = (s.period[k] > 0 && (int)((i - s.begin[k]) % s.period[k]) >= s.split[k]) ? s.lr_b[k] : s.lr_a[k];
      lr_bc1 = lr * s.inv_bc1[k];
      inv_sqrt_bc2 = s.inv_sqrt_bc2[k];
      hit = adam_row_on(s, k, i, row_mask);
    }
  return hit;
}

__global__ void __launch_bounds__(GS_BLOCK) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n,
                                                        AdamSegs segs, float b1, float b2, float eps,
                                                        const float* __restrict__ gate, const float* __restrict__ row_mask) {
  if (gate && *gate != 0.0f) return;  // gs_adam_step_gated: some rank's view was invalid - nobody steps
  const long long n4 = n >> 2;
  for (long long i4 = (long long)blockIdx.x * GS_BLOCK + threadIdx.x; i4 < n4; i4 += (long long)gridDim.x * GS_BLOCK) {
    float4 pp = reinterpret_cast<float4*>(p)[i4];
    const float4 gg = reinterpret_cast<const float4*>(g)[i4];
    float4 mm = reinterpret_cast<float4*>(m)[i4];
    float4 vv = reinterpret_cast<float4*>(v)[i4];
    float* pe = reinterpret_cast<float*>(&pp);
    const float* ge = reinterpret_cast<const float*>(&gg);
    float* me = reinterpret_cast<float*>(&mm);
    float* ve = reinterpret_cast<float*>(&vv);
    // Fast path: the four elements lie in one segment (always, except where a segment boundary is not a multiple
    // of 4): one table walk and one 64-bit modulo per vector instead of four
    const long long i0 = i4 * 4;
    int seg = -1;
#pragma unroll
    for (int k = 0; k < ADAM_MAX_SEG; k++)
      if (k < segs.n && i0 >= segs.begin[k] && i0 + 3 < segs.end[k]) seg = k;
    if (seg >= 0) {
      const float inv_sqrt_bc2 = segs.inv_sqrt_bc2[seg];
      const float lra = segs.lr_a[seg] * segs.inv_bc1[seg], lrb = segs.lr_b[seg] * segs.inv_bc1[seg];
      const int period = segs.period[seg], split = segs.split[seg];
      int ph = period > 0 ? (int)((i0 - segs.begin[seg]) % period) : 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float lr_bc1 = (period > 0 && ph >= split) ? lrb : lra;
        ph = (ph + 1 == period) ? 0 : ph + 1;
        if (!adam_row_on(segs, seg, i0 + k, row_mask)) continue;
        me[k] = b1 * me[k] + (1.f - b1) * ge[k];
        ve[k] = b2 * ve[k] + (1.f - b2) * ge[k] * ge[k];
        const float denom = sqrtf(ve[k]) * inv_sqrt_bc2 + eps;
        pe[k] = pe[k] - lr_bc1 * (me[k] / denom);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float lr_bc1, inv_sqrt_bc2;
        if (!adam_coef(segs, i0 + k, lr_bc1, inv_sqrt_bc2, row_mask)) continue;
        me[k] = b1 * me[k] + (1.f - b1) * ge[k];
        ve[k] = b2 * ve[k] + (1.f - b2) * ge[k] * ge[k];
        const float denom = sqrtf(ve[k]) * inv_sqrt_bc2 + eps;
        pe[k] = pe[k] - lr_bc1 * (me[k] / denom);
      }
    }
    reinterpret_cast<float4*>(p)[i4] = pp;
    reinterpret_cast<float4*>(m)[i4] = mm;
    reinterpret_cast<float4*>(v)[i4] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const long long i = n4 * 4 + threadIdx.x;
    float lr_bc1, inv_sqrt_bc2;
    if (adam_coef(segs, i, lr_bc1, inv_sqrt_bc2, row_mask)) {
      const float gi = g[i];
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      p[i] = p[i] - lr_bc1 * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
  }
}

extern "C" int gs_adam_step_gated(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const GsAdamSeg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                                  const float* gate, void* stream);
extern "C" int gs_adam_step_masked(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const GsAdamSeg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                                   const float* gate, const float* row_mask, void* stream);
extern "C" int gs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                            const GsAdamSeg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                            void* stream) {
  return gs_adam_step_gated(params, grads, exp_avg, exp_avg_sq, n, segs, nseg, beta1, beta2, eps, step, nullptr, stream);
}
extern "C" int gs_adam_step_gated(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const GsAdamSeg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                                  const float* gate, void* stream) {
  return gs_adam_step_masked(params, grads, exp_avg, exp_avg_sq, n, segs, nseg, beta1, beta2, eps, step, gate, nullptr, stream);
}
extern "C" int gs_adam_step_masked(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const GsAdamSeg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                                   const float* gate, const float* row_mask, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || (nseg > 0 && !segs)) return GS_E_NULL;
  if (n < 0 || nseg < 0 || nseg > ADAM_MAX_SEG || step < 1) return GS_E_SHAPE;
  if (n == 0) return GS_OK;
  if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
    return GS_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_ADAM, s);
  AdamSegs a;
  a.n = nseg;
  for (int k = 0; k < ADAM_MAX_SEG; k++) {
    const bool on = k < nseg;
    a.begin[k] = on ? segs[k].begin : 0;
    a.end[k] = on ? segs[k].end : 0;
    a.lr_a[k] = on ? segs[k].lr_a : 0.f;
    a.lr_b[k] = on ? segs[k].lr_b : 0.f;
    a.period[k] = on ? segs[k].period : 0;
    a.split[k] = on ? segs[k].split : 0;
    a.row_width[k] = on ? segs[k].row_width : 0;
    const int st = (on && segs[k].step > 0) ? segs[k].step : step;
    const double bc1 = 1.0 - pow((double)beta1, (double)st), bc2 = 1.0 - pow((double)beta2, (double)st);
    a.inv_bc1[k] = (float)(1.0 / bc1);
    a.inv_sqrt_bc2[k] = (float)(1.0 / sqrt(bc2));
  }
  const long long n4 = (n + 3) / 4;
  long long blocks = (n4 + GS_BLOCK - 1) / GS_BLOCK;
  // one float4 per thread up to 2^20 workgroups: 0.292 ms at 59 M elements against 0.315 with an 8192-workgroup
  // grid-stride loop (and 0.331 with 2048) - short-lived workgroups keep more requests in flight
  if (blocks > (1ll << 20)) blocks = 1ll << 20;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(GS_BLOCK), 0, s, params, grads, exp_avg, exp_avg_sq,
                     (long long)n, a, beta1, beta2, eps, gate, row_mask);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
