// gs_knn.hip - distCUDA2: mean squared distance to the 3 nearest neighbours (initial Gaussian scales).
// Compiled with -ffp-contract=off (bit-exact against the CPU oracle).
//
// Replaces SimpleKNN::knn (simple-knn/simple_knn.cu:186-222): AABB reduce seeded with (0,0,0), 30-bit
// Morton codes (:46-62), stable sort by code, AABBs of 1024-point boxes (:79-118), pruned search (:148-184).
// MI355X structure: no hipMalloc / host round trips inside the call (the reference does cudaMalloc,
// two blocking D2H copies and a cudaFree); the sort reuses the library's stable LSD radix sort; points
// are gathered ONCE into Morton order (float4) so the search streams them; a candidate box that any
// query of a 256-query workgroup still needs is staged in LDS (12 KB) and scanned from there with
// broadcast reads instead of 1024 dependent global gathers per query.
#include <float.h>

#include "gs_common.h"
#include "gs_math.h"
#include "gs_prof.h"

#define KNN_BOX 1024

struct KnnHeader {
  float minn[3];
  float maxx[3];
  uint32_t n;
  uint32_t pad[57];
};
static_assert(sizeof(KnnHeader) == 256, "");

struct KnnTmp {
  KnnHeader* hdr;
  float* partial;  // [nblk][6]
  void* bin;       // radix sort workspace
  float* boxes;    // [nb][6]
  float4* sorted;  // [P] xyz in Morton order
  size_t nblk, nb;
};
#define KNN_MM_BLOCKS 512
static size_t knn_bytes(size_t P) {
  size_t nb = (P + KNN_BOX - 1) / KNN_BOX;
  return sizeof(KnnHeader) + gs_align(KNN_MM_BLOCKS * 24) + gs_align(sort_bytes(P)) + gs_align(nb * 24) + gs_align(16 * P);
}
static KnnTmp knn_view(void* buf, size_t P) {
  char* p = (char*)buf;
  KnnTmp t;
  t.nblk = KNN_MM_BLOCKS;
  t.nb = (P + KNN_BOX - 1) / KNN_BOX;
  t.hdr = (KnnHeader*)p; p += sizeof(KnnHeader);
  t.partial = (float*)p; p += gs_align(KNN_MM_BLOCKS * 24);
  t.bin = p; p += gs_align(sort_bytes(P));
  t.boxes = (float*)p; p += gs_align(t.nb * 24);
  t.sorted = (float4*)p;
  return t;
}

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// cub::DeviceReduce with init {0,0,0}: simple_knn.cu:192-201
__global__ void __launch_bounds__(GS_BLOCK) knn_minmax_partial(const float* __restrict__ xyz, int P, float* partial) {
  float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
  for (int i = blockIdx.x * GS_BLOCK + threadIdx.x; i < P; i += gridDim.x * GS_BLOCK) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float v = xyz[3 * (size_t)i + c];
      mn[c] = fminf(mn[c], v);
      mx[c] = fmaxf(mx[c], v);
    }
  }
  __shared__ float red[GS_BLOCK / 64][6];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    mn[c] = wave_min_f(mn[c]);
    mx[c] = wave_max_f(mx[c]);
  }
  if ((threadIdx.x & 63) == 0)
    for (int c = 0; c < 3; c++) {
      red[threadIdx.x >> 6][c] = mn[c];
      red[threadIdx.x >> 6][3 + c] = mx[c];
    }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int w = 1; w < GS_BLOCK / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
    partial[blockIdx.x * 6 + threadIdx.x] = v;
  }
}
__global__ void knn_minmax_final(const float* partial, int nblk, KnnHeader* hdr, uint32_t P) {
  const int c = threadIdx.x;
  if (c < 6) {
    float v = partial[c];
    for (int b = 1; b < nblk; b++) v = c < 3 ? fminf(v, partial[b * 6 + c]) : fmaxf(v, partial[b * 6 + c]);
    if (c < 3) hdr->minn[c] = v; else hdr->maxx[c - 3] = v;
  }
  if (c == 0) hdr->n = P;
}

// simple_knn.cu:46-62
GS_DEV uint32_t prep_morton(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
__global__ void __launch_bounds__(GS_BLOCK) knn_morton_kernel(const float* __restrict__ xyz, int P, const KnnHeader* hdr,
                                                              uint32_t* keys, uint32_t* vals) {
  const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (i >= P) return;
  const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
  const uint32_t x = prep_morton(f2u_sat(((px - hdr->minn[0]) / (hdr->maxx[0] - hdr->minn[0])) * ((1 << 10) - 1)));
  const uint32_t y = prep_morton(f2u_sat(((py - hdr->minn[1]) / (hdr->maxx[1] - hdr->minn[1])) * ((1 << 10) - 1)));
  const uint32_t z = prep_morton(f2u_sat(((pz - hdr->minn[2]) / (hdr->maxx[2] - hdr->minn[2])) * ((1 << 10) - 1)));
  keys[i] = x | (y << 1) | (z << 2);
  vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(GS_BLOCK) knn_gather_kernel(const float* __restrict__ xyz, int P,
                                                              const uint32_t* __restrict__ order, float4* sorted) {
  const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
  if (i >= P) return;
  const uint32_t id = order[i];
  sorted[i] = make_float4(xyz[3 * (size_t)id], xyz[3 * (size_t)id + 1], xyz[3 * (size_t)id + 2], 0.f);
}

// boxMinMax, simple_knn.cu:79-118 (one workgroup per 1024-point box)
__global__ void __launch_bounds__(GS_BLOCK) knn_box_kernel(const float4* __restrict__ sorted, int P, float* boxes) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  const int base = blockIdx.x * KNN_BOX;
  for (int k = threadIdx.x; k < KNN_BOX; k += GS_BLOCK) {
    const int i = base + k;
    if (i < P) {
      const float4 p = sorted[i];
      mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
  }
  __shared__ float red[GS_BLOCK / 64][6];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    mn[c] = wave_min_f(mn[c]);
    mx[c] = wave_max_f(mx[c]);
  }
  if ((threadIdx.x & 63) == 0)
    for (int c = 0; c < 3; c++) {
      red[threadIdx.x >> 6][c] = mn[c];
      red[threadIdx.x >> 6][3 + c] = mx[c];
    }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int w = 1; w < GS_BLOCK / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
    boxes[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

// simple_knn.cu:132-146
GS_DEV void update_kbest3(float rx, float ry, float rz, float4 pt, float* knn) {
  const float dx = pt.x - rx, dy = pt.y - ry, dz = pt.z - rz;
  float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    if (knn[j] > dist) {
      float t = knn[j];
      knn[j] = dist;
      dist = t;
    }
  }
}
// simple_knn.cu:120-130
GS_DEV float dist_box_point(const float* b, float px, float py, float pz) {
  float dx = 0.f, dy = 0.f, dz = 0.f;
  if (px < b[0] || px > b[3]) dx = fminf(fabsf(px - b[0]), fabsf(px - b[3]));
  if (py < b[1] || py > b[4]) dy = fminf(fabsf(py - b[1]), fabsf(py - b[4]));
  if (pz < b[2] || pz > b[5]) dz = fminf(fabsf(pz - b[2]), fabsf(pz - b[5]));
  return dx * dx + dy * dy + dz * dz;
}

// FSGS's fork also reports WHICH three points are nearest (FSGS/submodules/simple-knn/simple_knn.cu:132-189):
// the index travels with the distance through the same insertion, starts at 0 and is NOT cleared between the
// neighbour pass and the box pass.
GS_DEV void update_kbest3_idx(float rx, float ry, float rz, float4 pt, float* knn, int32_t* ind, int32_t pid) {
  const float dx = pt.x - rx, dy = pt.y - ry, dz = pt.z - rz;
  float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    if (knn[j] > dist) {
      const float t = knn[j];
      knn[j] = dist;
      dist = t;
      const int32_t ti = ind[j];
      ind[j] = pid;
      pid = ti;
    }
  }
}

// boxMeanDist, simple_knn.cu:148-184: 256 consecutive (Morton-ordered) queries per workgroup
template <bool WITH_IDX>
__global__ void __launch_bounds__(GS_BLOCK) knn_search_kernel(const float4* __restrict__ sorted, int P,
                                                              const uint32_t* __restrict__ order,
                                                              const float* __restrict__ boxes, int nb,
                                                              float* __restrict__ out, int32_t* __restrict__ nearest) {
  __shared__ float4 s_pts[KNN_BOX];
  __shared__ uint32_t s_ord[WITH_IDX ? KNN_BOX : 1];
  __shared__ float s_box[6];
  const int idx = blockIdx.x * GS_BLOCK + threadIdx.x;
  const bool active = idx < P;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  int32_t bi[3] = {0, 0, 0};
  float reject = FLT_MAX;
  if (active) {
    q = sorted[idx];
    for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
      if (i == idx) continue;
      if (WITH_IDX) update_kbest3_idx(q.x, q.y, q.z, sorted[i], best, bi, (int32_t)order[i]);
      else update_kbest3(q.x, q.y, q.z, sorted[i], best);
    }
    reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
  }
  for (int b = 0; b < nb; b++) {
    __syncthreads();  // previous box fully consumed
    if (threadIdx.x < 6) s_box[threadIdx.x] = boxes[b * 6 + threadIdx.x];
    __syncthreads();
    bool need = false;
    if (active) {
      const float dist = dist_box_point(s_box, q.x, q.y, q.z);
      need = !(dist > reject || dist > best[2]);
    }
    if (!__syncthreads_or(need)) continue;
    const int base = b * KNN_BOX;
    const int cnt = min(KNN_BOX, P - base);
    for (int k = threadIdx.x; k < cnt; k += GS_BLOCK) {
      s_pts[k] = sorted[base + k];
      if (WITH_IDX) s_ord[k] = order[base + k];
    }
    __syncthreads();
    if (need) {
      for (int k = 0; k < cnt; k++) {
        if (base + k == idx) continue;
        if (WITH_IDX) update_kbest3_idx(q.x, q.y, q.z, s_pts[k], best, bi, (int32_t)s_ord[k]);
        else update_kbest3(q.x, q.y, q.z, s_pts[k], best);
      }
    }
  }
  if (active) {
    const uint32_t me = order[idx];
    out[me] = (best[0] + best[1] + best[2]) / 3.0f;
    if (WITH_IDX) {
      nearest[3 * (size_t)me] = bi[0];
      nearest[3 * (size_t)me + 1] = bi[1];
      nearest[3 * (size_t)me + 2] = bi[2];
    }
  }
}

extern "C" {

size_t gs_knn_tmp_bytes(int32_t P) { return knn_bytes((size_t)(P > 0 ? P : 1)); }

int gs_knn_mean_dist2(const float* xyz, int32_t P, float* out, void* tmp, size_t tmp_bytes, void* stream) {
  return gs_knn_mean_dist2_idx(xyz, P, out, nullptr, tmp, tmp_bytes, stream);
}

int gs_knn_mean_dist2_idx(const float* xyz, int32_t P, float* out, int32_t* nearest, void* tmp, size_t tmp_bytes,
                          void* stream) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!xyz || !out || !tmp) return GS_E_NULL;
  if (tmp_bytes < knn_bytes((size_t)P)) return GS_E_SCRATCH;
  hipStream_t s = (hipStream_t)stream;
  GS_PROF(ST_KNN, s);
  KnnTmp t = knn_view(tmp, (size_t)P);
  SortBufs bv = sort_view(t.bin, (size_t)P);
  const int nblk_p = (P + GS_BLOCK - 1) / GS_BLOCK;
  const int mm_blocks = nblk_p < KNN_MM_BLOCKS ? nblk_p : KNN_MM_BLOCKS;
  hipLaunchKernelGGL(knn_minmax_partial, dim3(mm_blocks), dim3(GS_BLOCK), 0, s, xyz, P, t.partial);
  hipLaunchKernelGGL(knn_minmax_final, dim3(1), dim3(64), 0, s, t.partial, mm_blocks, t.hdr, (uint32_t)P);
  // 30-bit codes -> 4 passes (even): start in half 0, sorted list ends in half 0
  hipLaunchKernelGGL(knn_morton_kernel, dim3(nblk_p), dim3(GS_BLOCK), 0, s, xyz, P, t.hdr, bv.keys[0], bv.vals[0]);
  GS_LAUNCH_CHECK(s, 0);
  int rc = launch_radix_sort(bv, &t.hdr->n, P, 32, 0, nullptr, s, 0);
  if (rc) return rc;
  hipLaunchKernelGGL(knn_gather_kernel, dim3(nblk_p), dim3(GS_BLOCK), 0, s, xyz, P, bv.vals[0], t.sorted);
  hipLaunchKernelGGL(knn_box_kernel, dim3((unsigned)t.nb), dim3(GS_BLOCK), 0, s, t.sorted, P, t.boxes);
  if (nearest)
    hipLaunchKernelGGL(knn_search_kernel<true>, dim3(nblk_p), dim3(GS_BLOCK), 0, s, t.sorted, P, bv.vals[0], t.boxes,
                       (int)t.nb, out, nearest);
  else
    hipLaunchKernelGGL(knn_search_kernel<false>, dim3(nblk_p), dim3(GS_BLOCK), 0, s, t.sorted, P, bv.vals[0], t.boxes,
                       (int)t.nb, out, nearest);
  GS_LAUNCH_CHECK(s, 0);
  return GS_OK;
}
}
