// gs_preprocess_bwd.hip - backward per-Gaussian stage (HBM-bound streaming kernel).
// Compiled with -ffp-contract=off: see gs_math.h.
//
// Does the work of computeCov2DCUDA (backward.cu:147-326) and preprocessCUDA<3> backward (backward.cu:398-449,
// with the SH backward :23-142 and the cov3D->scale/quaternion backward :330-393) in ONE pass:
// the reference launches two kernels that both re-read means / radii and round-trip dL_dmeans and
// dL_dcov3D through HBM; here dL/dSigma stays in registers between the two halves.  The mean gradient is the sum
// (cov2D part) + (projection part) + (SH view-direction part).
// The kernel writes EVERY row of every output (zeros for culled Gaussians), which replaces the
// reference's 304 B/Gaussian of cudaMemset (rasterize_points.cu:163-172).
// Per visible Gaussian: reads 80 B record (chain_kernel: 96 B float64 sums + 44 B parameters in, 80 B out) + 64 B splat + 24 B cov3D + 12+12+16+4 inputs
// + 192 B SH; writes 12+12+192+12+4+12+16+24 B.
#include "gs_common.h"
#include "gs_math.h"
#include "gs_backward_math.h"

// ------------------------------------------------------------------------------------------------------------------
// chain_kernel: rows (float64 blend sums) -> records (fp32 results of the float64 covariance chain), one thread per
// Gaussian, only those with sums do anything.  Registers are not an issue here (2 waves per SIMD allowed: the chain's
// ~170 VGPRs), the launch is ~10 us at 1 M Gaussians.  With clean_rows it zeroes each row it has read, so the rows are
// all-zero again for the next view's blend backward and no clear launch is needed (GsStepState.rows_clean).
// ------------------------------------------------------------------------------------------------------------------
// Only a fifth of the Gaussians have sums on depth-limited lists, scattered over the index range: one thread per Gaussian
// would run the ~700 double-precision instructions of the chain in EVERY wave at 20 % lane occupancy (measured: 33 us).  So
// ONE WAVE takes CHAIN_PER_WAVE consecutive Gaussians, loads their flags at once, compacts the indices of those with sums
// into its LDS list (ballots, no barrier: a workgroup is one wave) and its lanes then take one list entry each: nearly full
// waves, a fifth of the instruction issue, and 3 900 independent waves to hide the two dependent round trips behind.
#define CHAIN_PER_WAVE 256
__global__ void __launch_bounds__(64, 3) chain_kernel(PreprocessBwdArgs a, const GeomHeader* hdr) {
  __shared__ int s_list[CHAIN_PER_WAVE];
  const int lane = threadIdx.x;
  const int base = blockIdx.x * CHAIN_PER_WAVE;
  bool has[CHAIN_PER_WAVE / 64];
#pragma unroll
  for (int k = 0; k < CHAIN_PER_WAVE / 64; k++) {   // (all flag loads in flight before the first is used)
    const int idx = base + k * 64 + lane;
    has[k] = idx < a.P && a.radii[idx] > 0 && !(a.skip_uninstanced && a.tiles_touched[idx] == 0);
  }
  int n = 0;
#pragma unroll
  for (int k = 0; k < CHAIN_PER_WAVE / 64; k++) {
    const unsigned long long m = __ballot(has[k]);
    if (has[k]) s_list[n + __popcll(m & ((1ull << lane) - 1ull))] = base + k * 64 + lane;
    n += __popcll(m);
  }
  __syncthreads();  // (one wave: orders the LDS writes before the reads below)
  // the forward ran out of binning capacity or its depth limits proved too tight (possible only when the caller did not
  // re-run it: deferred verdict, replayed graph): the step kernels do nothing - only leave the rows clean
  const bool failed = hdr && (hdr->overflow | hdr->trunc_failed) != 0u;
  // fused multispectral step: the 4th channel is sigmoid(raw) * clamp(gain) - fold clamp(gain) into the record's gradient
  // and collect this wave's share of dL/dgain (see PreprocessBwdArgs)
  float gc = 1.0f, dgain = 0.f;
  bool gain_free = false;
  if (a.gain_partials) {
    const float gain = *a.extra_gain;
    gc = clamp_gain(gain);
    gain_free = gain >= 0.1f && gain <= 10.0f;
  }
  for (int j = lane; j < n; j += 64) {
    const int idx = s_list[j];
    gs_row_t* row = const_cast<gs_row_t*>(a.grad_rows) + (size_t)idx * GR_STRIDE;
    float* rec = a.grad_recs + (size_t)idx * GC_STRIDE;
    if (!failed) {
      chain_from_row(a, idx, row, rec);
      if (a.gain_partials) {
        const float dextra = rec[GC_EXTRA];
        if (gain_free) dgain += dextra * (1.0f / (1.0f + expf(-a.extra_raw[idx])));
        rec[GC_EXTRA] = dextra * gc;
      }
    }
    if (a.clean_rows) {
      float4* w = reinterpret_cast<float4*>(row);
      w[0] = w[1] = w[2] = w[3] = w[4] = w[5] = make_float4(0.f, 0.f, 0.f, 0.f);  // the twelve slots in use
    }
  }
  if (a.gain_partials) {  // (a fixed tree over the 64 lanes: the same bits whenever the wave runs)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dgain += __shfl_down(dgain, off, 64);
    if (lane == 0) a.gain_partials[blockIdx.x] = dgain;
  }
}
int launch_chain(const PreprocessBwdArgs& a, const GeomHeader* hdr, hipStream_t s) {
  hipLaunchKernelGGL(chain_kernel, dim3((a.P + CHAIN_PER_WAVE - 1) / CHAIN_PER_WAVE), dim3(64), 0, s, a, hdr);
  return 0;
}

__global__ void __launch_bounds__(GS_BLOCK, 4) preprocess_bwd_kernel(PreprocessBwdArgs a) {
  // dL_dsh rows (192 B per Gaussian) leave through LDS so that every store instruction of a wave covers 1 KiB of
  // consecutive addresses; written per lane (12 x 16 B at a 192-B stride) the same bytes cost 2.3x the HBM write
  // traffic (rocprofv3 WRITE_SIZE, profiles/).  Row stride 19 words (odd): conflict-free per-lane writes.
  __shared__ float s_sh[GS_BLOCK * SH_LDS_ROW];
  const int idx_raw = blockIdx.x * GS_BLOCK + threadIdx.x;
  const bool in_range = idx_raw < a.P;
  const int idx = in_range ? idx_raw : a.P - 1;  // out-of-range lanes shadow the last Gaussian and store nothing
  const bool visible = a.radii[idx] > 0;

  GeomBack gb = {};  // zeros: what a culled Gaussian stores
  // dL_dsh sink: zero row first (culled Gaussians and coefficients above the active degree stay 0)
  const bool sh_lds = a.shs && a.out.dL_dsh && a.M == 16;
  const bool sh_global = a.shs && a.out.dL_dsh && !sh_lds && in_range;  // generic M: straight to the global row
  ShSink dsh{sh_global ? a.out.dL_dsh + (size_t)idx * a.M * 3 : s_sh + threadIdx.x * SH_LDS_ROW, !sh_global};
  {
    const int nfl_row = sh_global ? a.M * 3 : SH_LDS_ROW;
    for (int k = 0; k < nfl_row; k++) dsh.p[k] = 0.f;
  }

  const bool active = visible && !(a.skip_uninstanced && a.tiles_touched[idx] == 0);  // see PreprocessBwdArgs
  if (active) geometry_backward(a, idx, gb);

  // geometry outputs leave first (their registers are free again before the 48 SH coefficients arrive); every row is
  // written, zeros when culled
  const GsGrads& o = a.out;
  if (in_range) {
    if (o.dL_dmeans2D) {
      o.dL_dmeans2D[3 * idx] = gb.dmean2D_x;
      o.dL_dmeans2D[3 * idx + 1] = gb.dmean2D_y;
      o.dL_dmeans2D[3 * idx + 2] = 0.f;
    }
    if (o.dL_dcolors) {
      o.dL_dcolors[3 * idx] = gb.dcolor.x;
      o.dL_dcolors[3 * idx + 1] = gb.dcolor.y;
      o.dL_dcolors[3 * idx + 2] = gb.dcolor.z;
    }
    if (o.dL_dopacity) o.dL_dopacity[idx] = gb.dop;
    if (o.dL_dextra) o.dL_dextra[idx] = gb.dextra;
    if (o.dL_dcov3D) {
      float2* cd = reinterpret_cast<float2*>(o.dL_dcov3D + (size_t)idx * 6);
      cd[0] = make_float2(gb.dcov[0], gb.dcov[1]);
      cd[1] = make_float2(gb.dcov[2], gb.dcov[3]);
      cd[2] = make_float2(gb.dcov[4], gb.dcov[5]);
    }
    if (a.scales) {
      if (o.dL_dscales) {
        o.dL_dscales[3 * idx] = gb.dscale.x;
        o.dL_dscales[3 * idx + 1] = gb.dscale.y;
        o.dL_dscales[3 * idx + 2] = gb.dscale.z;
      }
      if (o.dL_drotations) reinterpret_cast<float4*>(o.dL_drotations)[idx] = make_float4(gb.dq[0], gb.dq[1], gb.dq[2], gb.dq[3]);
    }
  }

  if (active && a.shs) gb.dmean = gb.dmean + sh_backward_row(a, idx, gb.dcolor, dsh);
  if (in_range && o.dL_dmeans3D) {
    o.dL_dmeans3D[3 * idx] = gb.dmean.x;
    o.dL_dmeans3D[3 * idx + 1] = gb.dmean.y;
    o.dL_dmeans3D[3 * idx + 2] = gb.dmean.z;
  }
  if (sh_lds) {
    __syncthreads();
    const int first = blockIdx.x * GS_BLOCK;
    const int nfl = min(GS_BLOCK, a.P - first) * 48;  // floats this workgroup owns, a multiple of 4
    float* dst = o.dL_dsh + (size_t)first * 48;
    for (int j = 4 * threadIdx.x; j < nfl; j += 4 * GS_BLOCK) {
      const int r = j / 48, c = j - r * 48;  // c is a multiple of 4: the four floats are in one row
      const float* src = s_sh + r * SH_LDS_ROW;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int k = (c + e) / 3, ch = (c + e) - 3 * k;
        v[e] = src[k] * src[16 + ch];  // the very product sh_backward's generic path forms: basis_k * dL_dRGB[ch]
      }
      *reinterpret_cast<float4*>(dst + j) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

int launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((a.P + GS_BLOCK - 1) / GS_BLOCK), dim3(GS_BLOCK), 0, s, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// gs_backward_step: the same per-Gaussian stage with the tail of a single-GPU train step fused in.  Every gradient of
// a Gaussian is final when its thread has run the chain rule, so the thread goes on - activation backward, view
// statistics, Adam - while the values are still in registers (dL_dsh: in the workgroup's LDS rows).  Element
// arithmetic = act_bwd_kernel / densify_stats_kernel (gs_model.hip) and adam_kernel (gs_adam.hip), same order of
// operations, so the fused step and the three-kernel step agree bit for bit given the same blend sums.
// ------------------------------------------------------------------------------------------------------------------
GS_DEV void adam_update(float& p, float g, float& m, float& v, float lr_bc1, float inv_sqrt_bc2, float b1, float b2, float eps) {
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p = p - lr_bc1 * (m / denom);
}
// Adam over the workgroup's contiguous piece of one parameter row array: `cnt` Gaussians x N floats starting at element
// `first * N` of p / m / v, gradients from the LDS image s_g[cnt * N] (same element order).  One float4 per thread and
// iteration when the piece starts on 16 bytes (always when P is a multiple of 4; after a densification a row's segment
// of the flat buffer may start anywhere), else element-wise.  LR_SPLIT > 0: elements with (index % 48) < LR_SPLIT take
// lr_a, the others lr_b (the SH block: 3 DC floats at feature_lr, 45 at feature_lr / 20).
#ifndef GS_PHASE1_UNROLL
#define GS_PHASE1_UNROLL 3
#endif
#ifndef GS_PHASE1_NT
#define GS_PHASE1_NT 1
#endif
typedef float gs_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const gs_f4v v = __builtin_nontemporal_load(reinterpret_cast<const gs_f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float* p, float4 v) {
  gs_f4v w;
  w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<gs_f4v*>(p));
}
// SPARSE ("sparse_adam", GsStepState.sparse; train.py:282-284): bit 1 of sel[r] = "Gaussian r is not visible in this view" -
// its elements are left alone in every mode, and a float4 that holds nothing else is neither read nor written.
// MODE (the two-phase step, StepArgs.phase; bit 0 of sel[r] = "Gaussian r of the workgroup's piece has NO instances"): the two
// phases split the float4s of a piece between them - 1: only float4s ALL of whose elements belong to Gaussians without
// instances (zero gradient); 2: the float4s with at least one element of a Gaussian WITH instances, every element of
// them (the gradient image holds zeros for the others) - so every float4 is read and written once per step, as in
// MODE 0.  (Unaligned pieces go element by element: 1 takes the flagged elements, 2 the others.)
template <int N, int LR_SPLIT, int MODE, typename G>
__device__ __forceinline__ void adam_block(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, int first,
                                           int cnt, const G& grad, float lr_a, float lr_b, float isb, float b1, float b2,
                                           float eps, const unsigned char* sel = nullptr) {
  float* pf = p + (size_t)first * N;
  float* mf = m + (size_t)first * N;
  float* vf = v + (size_t)first * N;
  const int nfl = cnt * N;
  const bool vec = (((((uintptr_t)pf) | ((uintptr_t)mf) | ((uintptr_t)vf)) & 15) == 0) && (nfl % 4 == 0);
  if (vec) {
    // U float4 triples in flight per thread: the loads of a group are all issued before the first is consumed (a plain
    // loop waits out the HBM latency once per float4: 12 round trips per thread for the SH block)
    constexpr int U = MODE == 1 ? GS_PHASE1_UNROLL : 3;  // (phase 1 shares the SIMDs with the backward blend: few registers)
    for (int j0 = 4 * threadIdx.x; j0 < nfl; j0 += 4 * GS_BLOCK * U) {
      float4 p4[U], m4[U], v4[U];
      unsigned char es[U][4] = {};
      bool any[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int j = j0 + u * 4 * GS_BLOCK;
        any[u] = j < nfl;
        if ((MODE != 0 || sel) && j < nfl) {
#pragma unroll
          for (int e = 0; e < 4; e++) es[u][e] = sel[(j + e) / N];
          const unsigned char all = es[u][0] & es[u][1] & es[u][2] & es[u][3];
          const bool all_flagged = (all & 1) != 0;
          any[u] = (MODE == 0 || (MODE == 1 ? all_flagged : !all_flagged)) && (all & 2) == 0;
        }
        if (any[u]) {
          if (MODE == 1 && GS_PHASE1_NT) {  // phase 1 streams next to the backward blend: keep its lines out of that kernel's L2
            p4[u] = nt_load4(pf + j); m4[u] = nt_load4(mf + j); v4[u] = nt_load4(vf + j);
          } else {
            p4[u] = *reinterpret_cast<const float4*>(pf + j);
            m4[u] = *reinterpret_cast<const float4*>(mf + j);
            v4[u] = *reinterpret_cast<const float4*>(vf + j);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int j = j0 + u * 4 * GS_BLOCK;
        if (any[u]) {
          float* pe = reinterpret_cast<float*>(&p4[u]);
          float* me = reinterpret_cast<float*>(&m4[u]);
          float* ve = reinterpret_cast<float*>(&v4[u]);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float lr = (LR_SPLIT > 0 && ((j + e) % 48) >= LR_SPLIT) ? lr_b : lr_a;
            if (!(es[u][e] & 2)) adam_update(pe[e], grad(j + e), me[e], ve[e], lr, isb, b1, b2, eps);
            // phase 1 runs next to the backward blend, which needs the registers: one element's IEEE sqrt / division
            // temporaries at a time instead of four interleaved (the kernel waits on HBM, not on issue)
            if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
          }
          if (MODE == 1 && GS_PHASE1_NT) {
            nt_store4(pf + j, p4[u]); nt_store4(mf + j, m4[u]); nt_store4(vf + j, v4[u]);
          } else {
            *reinterpret_cast<float4*>(pf + j) = p4[u];
            *reinterpret_cast<float4*>(mf + j) = m4[u];
            *reinterpret_cast<float4*>(vf + j) = v4[u];
          }
        }
      }
    }
  } else {
    for (int j = threadIdx.x; j < nfl; j += GS_BLOCK) {
      if (MODE != 0 && ((sel[j / N] & 1) != 0) != (MODE == 1)) continue;
      if (sel && (sel[j / N] & 2)) continue;
      float pe = pf[j], me = mf[j], ve = vf[j];
      const float lr = (LR_SPLIT > 0 && (j % 48) >= LR_SPLIT) ? lr_b : lr_a;
      adam_update(pe, grad(j), me, ve, lr, isb, b1, b2, eps);
      pf[j] = pe;
      mf[j] = me;
      vf[j] = ve;
    }
  }
}

// data-parallel form: the workgroup's piece of one gradient row array written out (same element order as adam_block reads)
template <int N, typename G>
__device__ __forceinline__ void grad_block(float* __restrict__ out, int first, int cnt, const G& grad) {
  float* of = out + (size_t)first * N;
  const int nfl = cnt * N;
  if ((((uintptr_t)of) & 15) == 0 && (nfl % 4 == 0)) {
    for (int j = 4 * threadIdx.x; j < nfl; j += 4 * GS_BLOCK)
      *reinterpret_cast<float4*>(of + j) = make_float4(grad(j), grad(j + 1), grad(j + 2), grad(j + 3));
  } else {
    for (int j = threadIdx.x; j < nfl; j += GS_BLOCK) of[j] = grad(j);
  }
}

// LDS image of the 11 non-SH gradients of the workgroup's Gaussians, row arrays back to back in their own element order
#define SG_XYZ 0
#define SG_OPAC (3 * GS_BLOCK)
#define SG_SCALE (4 * GS_BLOCK)
#define SG_ROT (7 * GS_BLOCK)
#define SG_EXTRA (11 * GS_BLOCK)
#define SG_TOTAL (12 * GS_BLOCK)

// dL/dgain of the fused multispectral step: chain_kernel left one partial sum per wave of 256 Gaussians; the FIRST workgroup
// of the per-Gaussian kernel adds them in index order (fixed shape: the same bits every run) and steps the gain - or stores
// the gradient (data-parallel form).  No other workgroup reads the gain (its clamp is folded into the records).
static_assert(CHAIN_PER_WAVE == GS_BLOCK, "one partial per workgroup of the streaming kernels");
__device__ __forceinline__ void gain_step(const StepArgs& sa, bool grads_out, float b1, float b2, float eps) {
  __shared__ float s_red[GS_BLOCK];
  const GsStepState& st = sa.st;
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (unsigned j = tid; j < gridDim.x; j += GS_BLOCK) acc += sa.gain_partials[j];
  s_red[tid] = acc;
  __syncthreads();
  for (int w = GS_BLOCK / 2; w > 0; w >>= 1) {
    if (tid < w) s_red[tid] += s_red[tid + w];
    __syncthreads();
  }
  if (tid == 0) {
    const float g = s_red[0];
    if (grads_out) {
      *st.grad_out_gain = g;
    } else if (st.step_gain > 0) {
      float p = *st.gain, m = *st.gain_m, v = *st.gain_v;
      adam_update(p, g, m, v, sa.x_lr_bc1[1], sa.x_inv_sqrt_bc2[1], b1, b2, eps);
      *st.gain = p; *st.gain_m = m; *st.gain_v = v;
    }
  }
}

// PHASE (StepArgs.phase; the two-phase step, gs_step_uninstanced): 0 = every Gaussian.  2 = the Gaussians WITH instances
// (statistics; Adam on every float4 that holds one of their elements) - the others are stepped by step_uninstanced_kernel
// below (phase 1), which needs nothing but the forward's geometry stage and runs on a side stream next to the backward
// blend: a pure HBM stream (parameters and moments of four fifths of the Gaussians with depth-limited lists) beside a
// kernel that is bound by vector issue.  Same arithmetic per element in every phase.
template <int PHASE>
__global__ void __launch_bounds__(GS_BLOCK, 4) preprocess_bwd_step_kernel(PreprocessBwdArgs a, StepArgs sa) {
  __shared__ float s_sh[GS_BLOCK * SH_LDS_ROW];
  __shared__ float s_g[SG_TOTAL];
  __shared__ unsigned char s_sel_buf[GS_BLOCK];
  const GsStepState& st = sa.st;
  const bool sparse = st.sparse != 0;
  unsigned char* const s_sel = (PHASE != 0 || sparse) ? s_sel_buf : nullptr;   // (adam_block's flags; phase 0 needs them only when sparse)
  const bool grads_out = st.grad_out[0] != nullptr;  // data-parallel form: gradients out, no Adam (gsplat.h)
  // the forward ran out of binning capacity (possible only when the caller did not re-run it: a replayed graph): the
  // image was not rendered, so nothing may be updated - the host sees the flag and repeats the step eagerly
  const bool failed = (sa.hdr->overflow | sa.hdr->trunc_failed) != 0u;
  if (grads_out && st.fail_flag && blockIdx.x == 0 && threadIdx.x == 0) *st.fail_flag = failed ? 1.0f : 0.0f;
  if (failed) {
    // (the rows the blend backward accumulated into for this invalid view were cleaned by chain_kernel)
    if (grads_out && st.max_radii2D) {  // this view contributes no statistics (the sum over ranks must stay finite)
      const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
      if (i < a.P) {
        st.xyz_gradient_accum[i] = 0.f;
        st.denom[i] = 0.f;
      }
    }
    if (grads_out && st.grad_mask) {  // ... and no gradients: an empty mask (the step is repeated; nothing is applied)
      const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
      if (i < a.P) st.grad_mask[i] = 0;
    }
    return;
  }
  // step-dependent constants from device memory when the launch is replayed from a captured graph
  if (st.coef_dev) {
#pragma unroll
    for (int k = 0; k < 6; k++) sa.lr_bc1[k] = st.coef_dev[k];
#pragma unroll
    for (int k = 0; k < 5; k++) sa.inv_sqrt_bc2[k] = st.coef_dev[6 + k];
    if (st.extra) {
      sa.x_lr_bc1[0] = st.coef_dev[11]; sa.x_inv_sqrt_bc2[0] = st.coef_dev[12];
      sa.x_lr_bc1[1] = st.coef_dev[13]; sa.x_inv_sqrt_bc2[1] = st.coef_dev[14];
    }
  }
  const int tid = threadIdx.x;
  const int idx_raw = blockIdx.x * GS_BLOCK + tid;
  const bool in_range = idx_raw < a.P;
  const int idx = in_range ? idx_raw : a.P - 1;
  const int radius = a.radii[idx];
  const bool visible = radius > 0;
  const float b1 = st.beta1, b2 = st.beta2, eps = st.eps;
  // a Gaussian that emitted no instance (culled spans, depth limits) has all-zero blend sums, hence zero gradients: it
  // still counts as seen (statistics) and still takes its Adam step, but its records, sums and SH row are not read
  const bool instanced = !(a.skip_uninstanced && a.tiles_touched[idx] == 0);
  const bool mine = PHASE == 0 || (in_range && instanced);  // this launch does this Gaussian's statistics
  // adam_block's flags: bit 0 = a Gaussian WITHOUT instances (the other phase's), bit 1 = not stepped at all (sparse_adam: not visible)
  if (s_sel) s_sel[tid] = (unsigned char)(((PHASE != 0 && in_range && !instanced) ? 1 : 0) | ((sparse && !visible) ? 2 : 0));
  const bool active = visible && instanced;

  GeomBack gb = {};
  ShSink dsh{s_sh + tid * SH_LDS_ROW, true};
#pragma unroll
  for (int k = 0; k < SH_LDS_ROW; k++) dsh.p[k] = 0.f;
  if (active) geometry_backward(a, idx, gb);

  // ---- view statistics (train.py:266-268, gaussian_model.py:471-473)
  if (in_range && st.max_radii2D && mine) {
    const float gnorm = sqrtf(gb.dmean2D_x * gb.dmean2D_x + gb.dmean2D_y * gb.dmean2D_y);
    if (grads_out) {  // this view's increments, assigned (the caller sums them over ranks)
      if (visible) st.max_radii2D[idx] = fmaxf(st.max_radii2D[idx], (float)radius);
      st.xyz_gradient_accum[idx] = visible ? gnorm : 0.f;
      st.denom[idx] = visible ? 1.0f : 0.f;
    } else if (visible) {
      st.max_radii2D[idx] = fmaxf(st.max_radii2D[idx], (float)radius);
      st.xyz_gradient_accum[idx] += gnorm;
      st.denom[idx] += 1.0f;
    }
  }
  // ---- phase 2 only steps float4s that hold an element of a Gaussian with instances: a workgroup without one is done
  const bool gain_block = st.extra && blockIdx.x == 0;  // (gain_step below is this workgroup's)
  if (PHASE == 2 && !grads_out && !gain_block) {
    if (!__syncthreads_or((in_range && instanced) ? 1 : 0)) return;
  }
  // ... and with sparse_adam the one-launch form has nothing to do for a workgroup without a visible Gaussian
  if (PHASE == 0 && sparse && !grads_out && !gain_block) {
    if (!__syncthreads_or((in_range && visible) ? 1 : 0)) return;
  }
  // ---- activation backward of this Gaussian's rows into the LDS gradient image
  {
    if (PHASE == 2 && !mine) {  // (stepped by phase 1: zero gradient, no parameter read)
#pragma unroll
      for (int k = 0; k < 3; k++) s_g[SG_XYZ + 3 * tid + k] = s_g[SG_SCALE + 3 * tid + k] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) s_g[SG_ROT + 4 * tid + k] = 0.f;
      s_g[SG_OPAC + tid] = 0.f;
      s_g[SG_EXTRA + tid] = 0.f;
    } else {
      // opacity = sigmoid(raw): grad * s * (1 - s)
      const float sg = 1.0f / (1.0f + expf(-st.opacity[idx]));
      s_g[SG_OPAC + tid] = gb.dop * (1.0f - sg) * sg;
      // scaling = exp(raw): grad * result
      s_g[SG_SCALE + 3 * tid] = gb.dscale.x * expf(st.scaling[3 * (size_t)idx]);
      s_g[SG_SCALE + 3 * tid + 1] = gb.dscale.y * expf(st.scaling[3 * (size_t)idx + 1]);
      s_g[SG_SCALE + 3 * tid + 2] = gb.dscale.z * expf(st.scaling[3 * (size_t)idx + 2]);
      // rotation = q / max(|q|, 1e-12): dq = (g - v (v . g)) / |q|
      const float* qr = st.rotation + 4 * (size_t)idx;
      const float4 q = make_float4(qr[0], qr[1], qr[2], qr[3]);
      const float norm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
      float g[4];
      if (norm > 1e-12f) {
        const float inv = 1.0f / norm;
        const float vx = q.x * inv, vy = q.y * inv, vz = q.z * inv, vw = q.w * inv;
        const float dot = vx * gb.dq[0] + vy * gb.dq[1] + vz * gb.dq[2] + vw * gb.dq[3];
        g[0] = (gb.dq[0] - vx * dot) * inv;
        g[1] = (gb.dq[1] - vy * dot) * inv;
        g[2] = (gb.dq[2] - vz * dot) * inv;
        g[3] = (gb.dq[3] - vw * dot) * inv;
      } else {
        g[0] = gb.dq[0] / 1e-12f; g[1] = gb.dq[1] / 1e-12f; g[2] = gb.dq[2] / 1e-12f; g[3] = gb.dq[3] / 1e-12f;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) s_g[SG_ROT + 4 * tid + k] = g[k];
      // 4th channel = sigmoid(raw) * clamp(gain): gradient of the raw row (the record holds dL/dextra * clamp(gain))
      if (st.extra) {
        const float sx = 1.0f / (1.0f + expf(-st.extra[idx]));
        s_g[SG_EXTRA + tid] = (gb.dextra * (1.0f - sx)) * sx;
      }
      // ---- SH half: basis values and colour gradient to the LDS row; the view-direction part completes dL_dmean
      if (active) gb.dmean = gb.dmean + sh_backward_row(a, idx, gb.dcolor, dsh);
      s_g[SG_XYZ + 3 * tid] = gb.dmean.x;
      s_g[SG_XYZ + 3 * tid + 1] = gb.dmean.y;
      s_g[SG_XYZ + 3 * tid + 2] = gb.dmean.z;
    }
  }
  // (barrier: every thread of the workgroup has read its parameters - the rows may now change)
  // Dormant blocks (GsStepState.dormant: every moment of every row +0).  Does any Gaussian of the workgroup bring a gradient
  // element that is not zero?  (-0 counts as zero: it leaves a +0 moment at +0; NaN counts as non-zero.)  If none does and
  // the block is dormant, the update changes no bit: nothing is streamed.  If one does, the block stops being dormant.
  bool brings = false;
  if (active && in_range) {
#pragma unroll
    for (int k = 0; k < 3; k++) brings |= s_g[SG_XYZ + 3 * tid + k] != 0.f || s_g[SG_SCALE + 3 * tid + k] != 0.f || dsh.p[16 + k] != 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) brings |= s_g[SG_ROT + 4 * tid + k] != 0.f;
    brings |= s_g[SG_OPAC + tid] != 0.f || (st.extra && s_g[SG_EXTRA + tid] != 0.f);
  }
  const int block_brings = __syncthreads_or(brings ? 1 : 0);
  if (!grads_out && st.dormant) {
    if (block_brings) {
      if (tid == 0) st.dormant[blockIdx.x] = 0;
    } else if (eps > 0.f && !gain_block && st.dormant[blockIdx.x] != 0) {
      return;
    }
  }

  // ---- Adam over the workgroup's contiguous pieces of the five row arrays: coalesced float4 streams of p, m, v
  const int first = blockIdx.x * GS_BLOCK;
  const int cnt = min(GS_BLOCK, a.P - first);
  struct Lds {
    const float* p;
    __device__ __forceinline__ float operator()(int j) const { return p[j]; }
  };
  struct ShGrad {  // dL_dsh[r][k][ch] = basis_k * dL_dRGB[ch] from the 19-word rows
    const float* s;
    __device__ __forceinline__ float operator()(int j) const {
      const int r = j / 48, c = j - r * 48, k = c / 3, ch = c - 3 * k;
      const float* src = s + r * SH_LDS_ROW;
      return src[k] * src[16 + ch];
    }
  };
  if (st.extra && blockIdx.x == 0) gain_step(sa, grads_out, b1, b2, eps);
  if (grads_out && st.grad_mask && in_range) st.grad_mask[idx] = active ? 1 : 0;  // (what a sparse exchange has to move)
  if (grads_out) {
    // the workgroup's contiguous piece of each gradient row array, straight from the LDS image
    if (PHASE == 0) {
      if (st.extra) grad_block<1>(st.grad_out_extra, first, cnt, Lds{s_g + SG_EXTRA});
      grad_block<3>(st.grad_out[0], first, cnt, Lds{s_g + SG_XYZ});
      grad_block<1>(st.grad_out[2], first, cnt, Lds{s_g + SG_OPAC});
      grad_block<3>(st.grad_out[3], first, cnt, Lds{s_g + SG_SCALE});
      grad_block<4>(st.grad_out[4], first, cnt, Lds{s_g + SG_ROT});
      if (st.grad_out_rest) {  // split rows: one contiguous gradient per model tensor (_features_dc, _features_rest)
        struct ShGradDc {
          const float* s;
          __device__ __forceinline__ float operator()(int j) const {
            const int r = j / 3, ch = j - 3 * r;
            const float* src = s + r * SH_LDS_ROW;
            return src[0] * src[16 + ch];
          }
        };
        struct ShGradRest {
          const float* s;
          __device__ __forceinline__ float operator()(int j) const {
            const int r = j / 45, c = j - r * 45, k = c / 3, ch = c - 3 * k;
            const float* src = s + r * SH_LDS_ROW;
            return src[k + 1] * src[16 + ch];
          }
        };
        grad_block<3>(st.grad_out[1], first, cnt, ShGradDc{s_sh});
        grad_block<45>(st.grad_out_rest, first, cnt, ShGradRest{s_sh});
      } else {
        grad_block<48>(st.grad_out[1], first, cnt, ShGrad{s_sh});
      }
    }
    return;
  }
  constexpr int SEL = PHASE;
  if (st.step[0] > 0)
    adam_block<3, 0, SEL>(st.xyz, st.m[0], st.v[0], first, cnt, Lds{s_g + SG_XYZ}, sa.lr_bc1[0], 0.f, sa.inv_sqrt_bc2[0], b1, b2, eps, s_sel);
  if (st.step[2] > 0)
    adam_block<1, 0, SEL>(st.opacity, st.m[2], st.v[2], first, cnt, Lds{s_g + SG_OPAC}, sa.lr_bc1[3], 0.f, sa.inv_sqrt_bc2[2], b1, b2, eps, s_sel);
  if (st.step[3] > 0)
    adam_block<3, 0, SEL>(st.scaling, st.m[3], st.v[3], first, cnt, Lds{s_g + SG_SCALE}, sa.lr_bc1[4], 0.f, sa.inv_sqrt_bc2[3], b1, b2, eps, s_sel);
  if (st.step[4] > 0)
    adam_block<4, 0, SEL>(st.rotation, st.m[4], st.v[4], first, cnt, Lds{s_g + SG_ROT}, sa.lr_bc1[5], 0.f, sa.inv_sqrt_bc2[4], b1, b2, eps, s_sel);
  if (st.extra && st.step_extra > 0)
    adam_block<1, 0, SEL>(st.extra, st.extra_m, st.extra_v, first, cnt, Lds{s_g + SG_EXTRA}, sa.x_lr_bc1[0], 0.f, sa.x_inv_sqrt_bc2[0], b1, b2,
                          eps, s_sel);
  if (st.step[1] > 0)
    adam_block<48, 3, SEL>(st.features, st.m[1], st.v[1], first, cnt, ShGrad{s_sh}, sa.lr_bc1[1], sa.lr_bc1[2], sa.inv_sqrt_bc2[1], b1,
                           b2, eps, s_sel);
}

// Phase 1 as its own, THROTTLED kernel: it runs on a side stream next to the criterion and the backward blend, which need the
// wave slots - launched as one workgroup per Gaussian block it would take every slot first and the other stream's kernels
// would queue behind it (measured: no overlap at all, the step got slower).  A few workgroups per CU walking the blocks keep
// enough loads in flight for the HBM stream (3 float4 triples per thread) and leave the machine to the others.
__global__ void __launch_bounds__(GS_BLOCK, 4) step_uninstanced_kernel(PreprocessBwdArgs a, StepArgs sa, int nblocks) {
  __shared__ unsigned char s_sel[GS_BLOCK];
  const GsStepState& st = sa.st;
  if ((sa.hdr->overflow | sa.hdr->trunc_failed) != 0u) return;  // (as preprocess_bwd_step_kernel)
  if (st.coef_dev) {  // (wave-uniform values: kept in scalar registers - this kernel shares its SIMDs with the backward blend)
#pragma unroll
    for (int k = 0; k < 6; k++) sa.lr_bc1[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st.coef_dev[k])));
#pragma unroll
    for (int k = 0; k < 5; k++)
      sa.inv_sqrt_bc2[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st.coef_dev[6 + k])));
    if (st.extra) {
      sa.x_lr_bc1[0] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st.coef_dev[11])));
      sa.x_inv_sqrt_bc2[0] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st.coef_dev[12])));
    }
  }
  const float b1 = st.beta1, b2 = st.beta2, eps = st.eps;
  const bool sparse = st.sparse != 0;
  const int tid = threadIdx.x;
  struct Zero {
    __device__ __forceinline__ float operator()(int) const { return 0.f; }
  };
  // blocks are handed out by a cursor in the view's header (zeroed by the forward): with dormant blocks - cheap, and lying in
  // long runs when the rows are in spatial order - a fixed round-robin would leave most workgroups waiting for the unluckiest
  __shared__ int s_blk;
  for (;;) {
    if (tid == 0) s_blk = (int)atomicAdd(const_cast<uint32_t*>(&sa.hdr->pad[HDR_SIDE_CURSOR]), 1u);
    __syncthreads();
    const int blk = s_blk;
    if (blk >= nblocks) {
      // every workgroup draws exactly one ticket past the end: the one holding the last of the nblocks + gridDim.x tickets knows
      // that nobody will draw again and re-arms the cursor - a second gs_step_uninstanced on the same geometry state (a retry,
      // a probe that re-times the launch) then steps again instead of silently doing nothing
      if (tid == 0 && blk == nblocks + (int)gridDim.x - 1) const_cast<uint32_t*>(sa.hdr->pad)[HDR_SIDE_CURSOR] = 0u;
      break;
    }
    const int idx = blk * GS_BLOCK + tid;
    const bool in_range = idx < a.P;
    const bool mine = in_range && a.tiles_touched[idx] == 0;
    const int radius = (mine && (st.max_radii2D || sparse)) ? a.radii[idx] : 0;
    const bool frozen = sparse && radius <= 0;   // sparse_adam: a Gaussian that is not visible in this view is not stepped
    s_sel[tid] = (unsigned char)((mine ? 1 : 0) | (frozen ? 2 : 0));
    if (mine && st.max_radii2D) {  // seen, with a zero gradient (train.py:266-268)
      if (radius > 0) {
        st.max_radii2D[idx] = fmaxf(st.max_radii2D[idx], (float)radius);
        st.denom[idx] += 1.0f;   // (xyz_gradient_accum += 0)
      }
    }
    const int live = __syncthreads_or((mine && !frozen) ? 1 : 0);
    // a dormant block (GsStepState.dormant: every moment of every row +0): the zero-gradient update changes nothing;
    // a block none of whose Gaussians this launch steps (all of them instanced, or - sparse_adam - not visible) likewise
    if (!live || (st.dormant && eps > 0.f && st.dormant[blk] != 0)) {
      __syncthreads();  // (s_sel is rewritten by the next block)
      continue;
    }
    const int first = blk * GS_BLOCK;
    const int cnt = min(GS_BLOCK, a.P - first);
    if (st.step[0] > 0)
      adam_block<3, 0, 1>(st.xyz, st.m[0], st.v[0], first, cnt, Zero{}, sa.lr_bc1[0], 0.f, sa.inv_sqrt_bc2[0], b1, b2, eps, s_sel);
    if (st.step[2] > 0)
      adam_block<1, 0, 1>(st.opacity, st.m[2], st.v[2], first, cnt, Zero{}, sa.lr_bc1[3], 0.f, sa.inv_sqrt_bc2[2], b1, b2, eps, s_sel);
    if (st.step[3] > 0)
      adam_block<3, 0, 1>(st.scaling, st.m[3], st.v[3], first, cnt, Zero{}, sa.lr_bc1[4], 0.f, sa.inv_sqrt_bc2[3], b1, b2, eps, s_sel);
    if (st.step[4] > 0)
      adam_block<4, 0, 1>(st.rotation, st.m[4], st.v[4], first, cnt, Zero{}, sa.lr_bc1[5], 0.f, sa.inv_sqrt_bc2[4], b1, b2, eps, s_sel);
    if (st.extra && st.step_extra > 0)
      adam_block<1, 0, 1>(st.extra, st.extra_m, st.extra_v, first, cnt, Zero{}, sa.x_lr_bc1[0], 0.f, sa.x_inv_sqrt_bc2[0], b1, b2, eps, s_sel);
    if (st.step[1] > 0)
      adam_block<48, 3, 1>(st.features, st.m[1], st.v[1], first, cnt, Zero{}, sa.lr_bc1[1], sa.lr_bc1[2], sa.inv_sqrt_bc2[1], b1,
                           b2, eps, s_sel);
    __syncthreads();  // (s_sel is rewritten by the next block)
  }
}

int launch_preprocess_bwd_step(const PreprocessBwdArgs& a, const StepArgs& sa, hipStream_t s) {
  const dim3 grid((a.P + GS_BLOCK - 1) / GS_BLOCK), block(GS_BLOCK);
  if (sa.phase == 1) {
    const int nblocks = (int)grid.x;
    // grid of the throttled kernel.  Round 3 (round-robin blocks, every block streamed): C3 320 / 384 / 448 / 512 / 640 workgroups
    // -> step 1.01 / 0.975 / 0.996 / 0.995 / 1.02 ms, C4 (2 M) 384 / 512 / 768 -> 1.379 / 1.336 / 1.337: 384 below 1.5 M, 512 above.
    // Round 4 (block cursor, dormant blocks skipped, rows in spatial order - profiles/r04_side_kernel_experiments.txt):
    // C3 256 / 320 / 384 / 512 / 640 -> 0.914 / 0.911 / 0.912 / 0.929 / 0.938 ms; C4 256 / 320 / 384 / 448 / 512 / 640 / 768 ->
    // 1.093 / 1.061-1.067 / 1.092 / 1.107 / 1.113 / 1.116 / 1.135: 320 at every size.
    int wgs = sa.phase1_workgroups > 0 ? sa.phase1_workgroups : 320;
    if (wgs > nblocks) wgs = nblocks;
    hipLaunchKernelGGL(step_uninstanced_kernel, dim3(wgs), block, 0, s, a, sa, nblocks);
  } else if (sa.phase == 2) hipLaunchKernelGGL(preprocess_bwd_step_kernel<2>, grid, block, 0, s, a, sa);
  else hipLaunchKernelGGL(preprocess_bwd_step_kernel<0>, grid, block, 0, s, a, sa);
  return 0;
}
