#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  uint2v r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r.x; out[64 + threadIdx.x] = r.y;
  uint2v q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + threadIdx.x] = q.x; out[192 + threadIdx.x] = q.y;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32.x", "swap32.y", "swap16.x", "swap16.y"};
  for (int r = 0; r < 4; r++) { printf("%s:", names[r]); for (int i = 0; i < 64; i += 4) printf(" %u", h[r * 64 + i]); printf("\n"); }
  return 0;
}
