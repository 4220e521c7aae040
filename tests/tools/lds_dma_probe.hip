// Developer probe: global_load_lds_dwordx4 on gfx950 - where does lane L's 16 bytes land in LDS?  (expected: base + 16 L)
// build: hipcc --offload-arch=gfx950 -O3 tests/tools/lds_dma_probe.hip -o tests/tools/_build/lds_dma_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, int active_mod) {
  __shared__ float4 buf[64];
  const int lane = threadIdx.x;
  buf[lane] = make_float4(-1.f, -1.f, -1.f, -1.f);
  __syncthreads();
  if (active_mod == 0 || (lane % active_mod) == 0) {   // exec-masked: inactive lanes must not load
    const float4* g = reinterpret_cast<const float4*>(src) + (63 - lane);   // lane L fetches float4 (63 - L)
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lds_ptr_t)buf, 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  reinterpret_cast<float4*>(dst)[lane] = buf[lane];
}
int main() {
  std::vector<float> h(256), o(256);
  for (int i = 0; i < 256; i++) h[i] = (float)i;
  float *d, *e;
  hipMalloc(&d, 1024); hipMalloc(&e, 1024);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  for (int mod : {0, 2}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, mod);
    hipMemcpy(o.data(), e, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int L = 0; L < 64; L++) {
      const bool act = mod == 0 || (L % mod) == 0;
      for (int c = 0; c < 4; c++) {
        const float want = act ? (float)(4 * (63 - L) + c) : -1.f;
        if (o[4 * L + c] != want) ok = 0;
      }
    }
    printf("active_mod %d: %s   lane0 %g %g %g %g  lane1 %g %g %g %g  lane63 %g %g %g %g\n", mod, ok ? "layout = base + 16*lane, masked lanes untouched" : "DIFFERENT",
           o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[252], o[253], o[254], o[255]);
  }
  return 0;
}
