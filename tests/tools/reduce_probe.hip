// developer probe: checks the 10-value wave reduction built from v_permlane32_swap / v_permlane16_swap / row_shr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dppw(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
__device__ __forceinline__ float swap32_add(float a, float b) {
  const uint2v r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  // NB: element access as r[0]/r[1] + __uint_as_float; `bit_cast(float, r.x) + bit_cast(float, r.y)` is miscompiled
  // by hipcc 7.2 into x + x (checked in the ISA)
  const unsigned x = r[0], y = r[1];
  return __uint_as_float(x) + __uint_as_float(y);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  const uint2v r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  // NB: element access as r[0]/r[1] + __uint_as_float; `bit_cast(float, r.x) + bit_cast(float, r.y)` is miscompiled
  // by hipcc 7.2 into x + x (checked in the ISA)
  const unsigned x = r[0], y = r[1];
  return __uint_as_float(x) + __uint_as_float(y);
}
__device__ __forceinline__ float row_total_in_lane15(float v) {
  v += dppw<0x111, 0xf, true>(v);
  v += dppw<0x112, 0xf, true>(v);
  v += dppw<0x114, 0xf, true>(v);
  v += dppw<0x118, 0xf, true>(v);
  return v;
}
__global__ void k(float* out) {
  float v[10];
  for (int i = 0; i < 10; i++) v[i] = (float)(threadIdx.x * 10 + i);
  const float s0 = swap32_add(v[0], v[2]), s1 = swap32_add(v[1], v[3]);
  const float s2 = swap32_add(v[4], v[6]), s3 = swap32_add(v[5], v[7]);
  const float s4 = swap32_add(v[8], v[9]);
  const float t0 = row_total_in_lane15(swap16_add(s0, s1));
  const float t1 = row_total_in_lane15(swap16_add(s2, s3));
  const float t2 = row_total_in_lane15(swap16_add(s4, 0.f));
  out[threadIdx.x] = t0; out[64 + threadIdx.x] = t1; out[128 + threadIdx.x] = t2;
}
int main() {
  float* d; hipMalloc(&d, 192 * 4); k<<<1, 64>>>(d); float h[192]; hipMemcpy(h, d, 768, hipMemcpyDeviceToHost);
  for (int r = 0; r < 3; r++) { printf("t%d lanes 15,31,47,63:", r); for (int q = 0; q < 4; q++) printf(" %.0f", h[r * 64 + 16 * q + 15]); printf("\n"); }
  printf("expected value k total = %d + 64 k\n", 20160);
  return 0;
}
