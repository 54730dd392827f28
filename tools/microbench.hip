// microbench.hip -- instruction-rate probes that size the integer-ALU bound of
// the NTT butterflies on gfx950 (DESIGN.md "roofline").  Dev tool, not product.
//   build: hipcc --offload-arch=gfx950 -O3 -I../gpqhe_amd/csrc microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "modarith.hpp"
using namespace gpq;

#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)

constexpr int ITER = 4096;

template <int OP>
__global__ __launch_bounds__(256) void probe(uint64_t *out, uint64_t seed, PrimeK k) {
  uint64_t v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = seed * (threadIdx.x + 1 + 64 * i) + blockIdx.x;
  uint32_t m = (uint32_t)seed | 1;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) v[i] = (uint32_t)v[i] + m + (v[i] & 0xffffffff00000000ull);             // v_add_u32
      if (OP == 1) v[i] = (uint64_t)((uint32_t)v[i] * m) | (v[i] & 0xffffffff00000000ull);  // v_mul_lo_u32
      if (OP == 2) v[i] = mad_u64((uint32_t)v[i], m, v[i]);                                 // v_mad_u64_u32
      if (OP == 3) v[i] = v[i] + k.p;                                                       // 64-bit add
      if (OP == 4) v[i] = mulmod_lazy(v[i] & 0x3fffffffffffffffull, k.p - 1 - i, k);        // 7 mads + fold
      if (OP == 5) v[i] = csub(v[i], k.p4) + m;                                             // conditional subtract
    }
    if (OP == 6) {  // 4 CT butterflies on the 8 values
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] &= 0x3fffffffffffffffull; v[i + 4] &= 0x3fffffffffffffffull; ct_bfly(v[i], v[i + 4], k.p - 3 - i, k); }
    }
    if (OP == 7) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] &= 0x1fffffffffffffffull; v[i + 4] &= 0x1fffffffffffffffull; gs_bfly(v[i], v[i + 4], k.p - 3 - i, k); }
    }
  }
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= v[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
int run(const char *name, double ops_per_iter, uint64_t *d_out, PrimeK k) {
  const int blocks = 256 * 8;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 0x9e3779b97f4a7c15ull, k);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 0x9e3779b97f4a7c15ull + r, k);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  double total = 5.0 * blocks * 256.0 * ITER * ops_per_iter;
  printf("%-28s %8.3f ms  %10.2f Gop/s  (%.3f lane-op per CU-cycle-lane at 2.4GHz)\n", name, ms / 5, total / (ms * 1e-3) / 1e9,
         total / (ms * 1e-3) / (256.0 * 128 * 2.4e9));
  return 0;
}

int main() {
  uint64_t *d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 8));
  PrimeK k; k.p = 576460752308273153ull; k.p2 = 2 * k.p; k.p4 = 4 * k.p; k.c = (uint32_t)(k.p - (1ull << 59)); k.c1 = k.c + 1;
  k.kx0 = k.c1; k.kx1 = (uint64_t)k.c1 - k.p4; k.ky = k.p4 - 2 * (uint64_t)k.c1;
  run<0>("v_add_u32", 8, d_out, k);
  run<1>("v_mul_lo_u32", 8, d_out, k);
  run<2>("v_mad_u64_u32", 8, d_out, k);
  run<3>("add u64", 8, d_out, k);
  run<4>("mulmod_lazy", 8, d_out, k);
  run<5>("csub+add", 8, d_out, k);
  run<6>("ct_bfly", 4, d_out, k);
  run<7>("gs_bfly", 4, d_out, k);
  return 0;
}
