// developer probe: what is the wave64 VALU issue ceiling of one MI355X SIMD as a function of the resident waves?
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak_probe valu_peak_probe.hip && ./valu_peak_probe [json-out]
//
// Every wave runs ITERS iterations of a block of 48 instructions issued from inline asm (so the compiler can neither
// fuse, pack nor reorder them):
//   fma       48 x v_fma_f32 on 8 independent accumulators (dependency distance 8)
//   fma_dep   48 x v_fma_f32 on ONE accumulator (the dependent-chain latency)
//   pk_fma    48 x v_pk_fma_f32 on 8 independent register pairs (two f32 lanes of work per instruction)
//   pk_mul    48 x v_pk_mul_f32
//   mix5_1    40 x v_fma_f32 + 8 x v_exp_f32 (the blend kernels' ratio: 5 VALU per transcendental)
//   blendmix  a stream shaped like the blend kernels' alpha test: v_sub, v_mul, v_fma x3, v_exp, v_mul, v_min, v_cmp x2
//   salu      48 x s_and_b64 on 8 independent SGPR pairs: the SCALAR issue rate (the blend loops keep their lane masks in
//             scalars: ballots, mask algebra, exec-mask regions - 60-90 scalar instructions per list entry)
//   valu_salu_1to1  48 x (v_fma_f32 ; s_and_b64) interleaved, 96 instructions: do the two pipes issue side by side?
// Waves per SIMD are forced by the LDS a workgroup asks for: a workgroup = 4 waves (one per SIMD of its CU) and
// w workgroups fit a CU, so every SIMD holds w waves.  The grid is 256 CUs x w x ROUNDS workgroups.
// Reported: G wave-instructions/s of the whole chip (events around the launch) and shader cycles per instruction per SIMD
// (s_memtime ticks of the slowest wave x resident waves / instructions per wave): the measured counterpart of the
// "2 cycles per wave64 VALU instruction" that 1 228.8 G wave-instr/s (1 024 SIMDs x 2.4 GHz / 2) assumes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

enum Kind { FMA = 0, FMA_DEP = 1, PK_FMA = 2, PK_MUL = 3, MIX5_1 = 4, BLENDMIX = 5, SALU = 6, VS_MIX = 7, NKIND = 8 };
static const char* kind_name[NKIND] = {"fma", "fma_dep", "pk_fma", "pk_mul", "mix5_1", "blendmix", "salu", "valu_salu_1to1"};
// VALU instructions per block of each kind (all 48), and how many f32 lane-operations one instruction performs
static const int kind_insts[NKIND] = {48, 48, 48, 48, 48, 48, 48, 96};

// one asm statement per 48-instruction block: between separate asm statements the compiler inserts s_nop hazard padding
#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define X6(blk) blk blk blk blk blk blk
#define OP_FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_FMA_DEP(i) "v_fma_f32 %0, %8, %9, %0\n"
#define OP_PKFMA(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define MIX8 "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n" \
             "v_fma_f32 %4, %8, %9, %4\n v_exp_f32 %5, %5\n"
#define ACC8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define PACC8 "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)

template <int KIND>
__global__ void __launch_bounds__(256) probe_kernel(float* __restrict__ out, unsigned long long* __restrict__ ticks,
                                                    int iters, float x, float y) {
  extern __shared__ float lds[];  // only its size matters (occupancy control)
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
        a7 = a0 + 7.f;
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4},
          p7 = {a7, a6};
  const float2v x2 = {x, x}, y2 = {y, y};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (KIND == FMA) {
      asm volatile(X6(R8(OP_FMA)) : ACC8 : "v"(x), "v"(y));
    } else if (KIND == FMA_DEP) {
      asm volatile(X6(R8(OP_FMA_DEP)) : ACC8 : "v"(x), "v"(y));
    } else if (KIND == PK_FMA) {
      asm volatile(X6(R8(OP_PKFMA)) : PACC8 : "v"(x2), "v"(y2));
    } else if (KIND == PK_MUL) {
      asm volatile(X6(R8(OP_PKMUL)) : PACC8 : "v"(x2), "v"(y2));
    } else if (KIND == SALU) {
      unsigned long long m0 = i, m1 = i + 1, m2 = i + 2, m3 = i + 3, m4 = i + 4, m5 = i + 5, m6 = i + 6, m7 = i + 7, mm = ~0ull;
#define OP_SAND(i) "s_and_b64 %" #i ", %" #i ", %8\n"
      asm volatile(X6(R8(OP_SAND)) : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3), "+s"(m4), "+s"(m5), "+s"(m6), "+s"(m7) : "s"(mm));
      a4 += (float)(m0 & 1);
    } else if (KIND == VS_MIX) {
      unsigned long long m0 = i, m1 = i + 1, m2 = i + 2, m3 = i + 3, mm = ~0ull;
#define OP_VS(i) "v_fma_f32 %" #i ", %12, %13, %" #i "\n s_and_b64 %8, %8, %14\n"
#define OP_VS2(i) "v_fma_f32 %" #i ", %12, %13, %" #i "\n s_and_b64 %9, %9, %14\n"
      asm volatile(X6(OP_VS(0) OP_VS2(1) OP_VS(2) OP_VS2(3) OP_VS(4) OP_VS2(5) OP_VS(6) OP_VS2(7))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(m0), "+s"(m1), "+s"(m2),
                     "+s"(m3)
                   : "v"(x), "v"(y), "s"(mm));
      a4 += (float)((m0 ^ m1) & 1);
    } else if (KIND == MIX5_1) {
      asm volatile(MIX8 MIX8 MIX8 MIX8 MIX8 MIX8 MIX8 MIX8 : ACC8 : "v"(x), "v"(y));
    } else {
      // four independent "pixels", each 11 VALU + 1 v_exp_f32 (sub, sub, mul, fma, mul, mul, fma, exp, mul, min, cmp, cmp)
#define PIX(d0, d1, t_, p_, g_, al_, m0_, m1_)                                                                       \
  "v_sub_f32 " d0 ", %24, %25\n v_sub_f32 " d1 ", %24, %26\n v_mul_f32 " t_ ", %27, " d0 "\n v_fma_f32 " t_ ", %28, " d1 ", " t_ "\n" \
  "v_mul_f32 " p_ ", %27, " d1 "\n v_mul_f32 " p_ ", " p_ ", " d1 "\n v_fma_f32 " p_ ", " t_ ", " d0 ", " p_ "\n v_exp_f32 " g_ ", " p_ "\n" \
  "v_mul_f32 " al_ ", %28, " g_ "\n v_min_f32 " al_ ", %29, " al_ "\n v_cmp_ge_f32 " m0_ ", %30, " p_ "\n v_cmp_le_f32 " m1_ ", %30, " al_ "\n"
      float r[16];
      unsigned long long m[8];
      asm volatile(PIX("%0", "%1", "%2", "%3", "%4", "%5", "%16", "%17") PIX("%6", "%7", "%8", "%9", "%10", "%11", "%18", "%19")
                   PIX("%12", "%13", "%14", "%15", "%0", "%1", "%20", "%21") PIX("%2", "%3", "%4", "%5", "%6", "%7", "%22", "%23")
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
                     "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]),
                     "=&s"(m[0]), "=&s"(m[1]), "=&s"(m[2]), "=&s"(m[3]), "=&s"(m[4]), "=&s"(m[5]), "=&s"(m[6]), "=&s"(m[7])
                   : "v"(a0), "v"(a1), "v"(a2), "v"(x), "v"(y), "v"(a6), "v"(a7));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
            p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
  if (s == 123.456f) out[0] = s + lds[threadIdx.x];  // keep everything alive
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Result { double ginst, cyc_per_inst, ms; unsigned long long max_ticks; };

template <int KIND>
static Result run(int w, int iters, int rounds, float* d_out, unsigned long long* d_ticks, int n_cu, double clock_hz) {
  // w workgroups per CU: each asks for just under 1/w of the 160 KB LDS (64 KB cap per workgroup: w = 1, 2 use registers
  // instead?  no - 64 KB x 2 = 128 KB < 160 KB would admit only 2; for w = 1 the grid itself has one workgroup per CU)
  const size_t lds = (size_t)(160 * 1024 / w) - 1024;
  const int grid = n_cu * w * rounds;
  CHECK(hipFuncSetAttribute((const void*)probe_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  probe_kernel<KIND><<<grid, 256, lds>>>(d_out, d_ticks, iters / 8, 1.0001f, 1e-6f);  // warm-up
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  probe_kernel<KIND><<<grid, 256, lds>>>(d_out, d_ticks, iters, 1.0001f, 1e-6f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)grid * 4);
  CHECK(hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long mx = 0;
  double mean = 0;
  for (auto t : h) { mx = t > mx ? t : mx; mean += (double)t; }
  mean /= (double)h.size();
  const int per_block = kind_insts[KIND];
  const double insts_per_wave = (double)iters * per_block;
  const double total = insts_per_wave * (double)grid * 4.0;
  Result r;
  r.ms = ms;
  r.ginst = total / (ms * 1e-3) / 1e9;
  // a SIMD holds w waves at a time; while they run (mean ticks) it retires w x insts_per_wave instructions
  r.cyc_per_inst = mean / (insts_per_wave * w);
  r.max_ticks = mx;
  (void)clock_hz;
  return r;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const double clock_hz = prop.clockRate * 1e3;
  printf("device %s, %d CUs, clockRate %.0f MHz\n", prop.name, n_cu, clock_hz / 1e6);
  float* d_out;
  unsigned long long* d_ticks;
  CHECK(hipMalloc(&d_out, 1024));
  CHECK(hipMalloc(&d_ticks, (size_t)n_cu * 8 * 8 * 4 * 8 + 1024));
  const int ws[] = {1, 2, 3, 4, 8};
  const int iters = 1 << 14, rounds = 4;
  FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (js) fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"results\": [\n", prop.name, n_cu, n_cu * 4);
  printf("%-9s %5s %10s %12s %14s\n", "kind", "w/SIMD", "ms", "G winst/s", "cyc/inst/SIMD");
  bool first = true;
  for (int k = 0; k < NKIND; k++) {
    for (int wi = 0; wi < 5; wi++) {
      const int w = ws[wi];
      Result r;
      switch (k) {
        case FMA: r = run<FMA>(w, iters, rounds, d_out, d_ticks, n_cu, clock_hz); break;
        case FMA_DEP: r = run<FMA_DEP>(w, iters, rounds, d_out, d_ticks, n_cu, clock_hz); break;
        case PK_FMA: r = run<PK_FMA>(w, iters, rounds, d_out, d_ticks, n_cu, clock_hz); break;
        case PK_MUL: r = run<PK_MUL>(w, iters, rounds, d_out, d_ticks, n_cu, clock_hz); break;
        case MIX5_1: r = run<MIX5_1>(w, iters, rounds, d_out, d_ticks, n_cu, clock_hz); break;
        case SALU: r = run<SALU>(w, iters, rounds, d_out, d_ticks, n_cu, clock_hz); break;
        case VS_MIX: r = run<VS_MIX>(w, iters, rounds, d_out, d_ticks, n_cu, clock_hz); break;
        default: r = run<BLENDMIX>(w, iters, rounds, d_out, d_ticks, n_cu, clock_hz); break;
      }
      printf("%-9s %5d %10.3f %12.1f %14.2f\n", kind_name[k], w, r.ms, r.ginst, r.cyc_per_inst);
      if (js) {
        fprintf(js, "%s {\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"g_wave_inst_per_s\": %.2f, \"cycles_per_inst_per_simd\": %.3f}",
                first ? "" : ",\n", kind_name[k], w, r.ms, r.ginst, r.cyc_per_inst);
        first = false;
      }
    }
  }
  if (js) { fprintf(js, "\n]}\n"); fclose(js); }
  return 0;
}
