// issue_probe.hip -- the issue rate of the library's OWN butterfly code when nothing but instruction issue is in the way (dev tool).
//
// A wave keeps eight coefficients and seven split-twiddle pairs in registers and runs the register groups of ntt_kernels.hpp
// (ct_group / gs_group of the wide-split class, the variable x variable products of the fused middles) ITER times; no memory traffic
// inside the loop.  Every wave reads the shader-clock counter (s_memtime) around its loop, so the result is in CYCLES and does not
// depend on the clock the part happens to run at; the wall time next to it gives that clock.  One workgroup of 256 W threads per CU
// puts exactly W waves on every SIMD; with W waves resident per SIMD
//     cycles per VALU wave-instruction on one SIMD = elapsed cycles / (W x VALU instructions of one wave's loop)
// where the instruction count of the loop body is taken from this file's own assembly (tools/valu_bound.py parses it).
//   issue_probe            prints one line per (mix, waves per SIMD)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gpqhe_amd/csrc tools/issue_probe.hip -o tools/issue_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "ntt_kernels.hpp"
using namespace gpq;
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
constexpr int ITER = 512;

// MIX 0: forward group (three wide-split CT stages on 8 coefficients = 12 butterflies)
// MIX 1: inverse group (three wide GS stages = 12 butterflies)
// MIX 2: what a fused middle does per 8 coefficients of one polynomial pair: forward group, one lazy product, inverse group
template <int MIX>
__global__ __launch_bounds__(1024) void probe(uint64_t *out, const LimbTab *tabs, const TwW *tw, unsigned long long *cycles) {
  extern __shared__ uint64_t pad[];                   // sized by the host: one workgroup per CU
  const PrimeK k = tabs[0].k;
  uint64_t x[8];
  TwW t[7];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = (tabs[0].ninv * (threadIdx.x + 1 + 64 * e) + blockIdx.x) % k.p;
#pragma unroll
  for (int j = 0; j < 7; ++j) t[j] = tw[(threadIdx.x & 63) * 7 + j];
  if (threadIdx.x == 0xffffff) pad[0] = x[0];
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (MIX == 0 || MIX == 2) ct_group<3, 2, 0, 5>(x, t, k);
    if (MIX == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = TwTraits<TwW>::inv_from4(mulmod_lazy(TwTraits<TwW>::left(x[e], k), TwTraits<TwW>::right(x[e ^ 1], k), k), k);
    }
    if (MIX == 1 || MIX == 2) gs_group<3, 2, 0>(x, t, k);
    if (MIX == 1) {                                    // keep the inverse-only loop inside its lazy range [0, 4p)
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = csub4(x[e], k);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t acc = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc ^= x[e];
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MIX>
int run(const char *name, int waves_per_simd, uint64_t *d_out, const LimbTab *d_tab, const TwW *d_tw, unsigned long long *d_cyc) {
  // ONE workgroup of 256 W threads per CU (100 KB of LDS keeps a second one out): its 4 W waves go round-robin to the CU's four
  // SIMDs, so every SIMD of the chip holds exactly W waves of the probe -- several small workgroups per CU do not spread evenly
  const size_t lds = 100 * 1024;
  const int blocks = 256, threads = 256 * waves_per_simd;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe<MIX>, dim3(blocks), dim3(threads), lds, 0, d_out, d_tab, d_tw, d_cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(probe<MIX>, dim3(blocks), dim3(threads), lds, 0, d_out, d_tab, d_tw, d_cyc);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  static unsigned long long h[256 * 16];
  const int waves = blocks * 4 * waves_per_simd;
  for (int blk = 0; blk < blocks; ++blk)
    CHECK(hipMemcpy(h + blk * 4 * waves_per_simd, d_cyc + blk * 16, sizeof(unsigned long long) * 4 * waves_per_simd, hipMemcpyDeviceToHost));
  double sum = 0; unsigned long long mx = 0, mn = ~0ull;
  for (int i = 0; i < waves; ++i) { sum += (double)h[i]; if (h[i] > mx) mx = h[i]; if (h[i] < mn) mn = h[i]; }
  const double mean = sum / waves;
  printf("mix %d %-44s waves/SIMD %d  iters %d  cycles/iter/wave mean %.1f max %.1f  kernel %.3f ms  implied clock %.0f MHz\n", MIX, name, waves_per_simd, ITER,
         mean / ITER, (double)mx / ITER, ms, (double)mx / (ms * 1e-3) / 1e6);   // (clock: the longest wave's cycles over the kernel's wall time, a lower bound)
  (void)mn;
  return 0;
}

int main(int argc, char **argv) {
  // the first prime of the n = 2^16 chain and plausible twiddle pairs: values only matter for staying in range
  const uint64_t p = 576460752308273153ull;
  LimbTab tab;
  memset((void *)&tab, 0, sizeof tab);
  tab.k.p = p; tab.k.p2 = 2 * p; tab.k.p4 = 4 * p; tab.k.c = (uint32_t)(p - (1ull << 59)); tab.k.c1 = tab.k.c + 1;
  tab.k.kx0 = tab.k.c1; tab.k.kx1 = (uint64_t)tab.k.c1 - 4 * p; tab.k.ky = 4 * p - 2 * (uint64_t)tab.k.c1;
  tab.k.kys = 2 * p - 2 * (uint64_t)tab.k.c1;
  tab.k.p3 = 3 * p; tab.k.np3 = (uint64_t)0 - 3 * p; tab.k.kx1x = (uint64_t)tab.k.c1 - 3 * p; tab.k.kyx = 3 * p - 2 * (uint64_t)tab.k.c1;
  tab.k.np = (uint64_t)0 - p; tab.k.np2 = (uint64_t)0 - 2 * p; tab.k.np4 = (uint64_t)0 - 4 * p;
  tab.ninv = 281474976710656ull % p;
  TwW htw[64 * 7];
  uint64_t st = 12345;
  for (auto &w : htw) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const uint64_t v = st % p;
    w.x = p - v; w.y = p - (uint64_t)((((unsigned __int128)v) << 31) % p);
  }
  LimbTab *d_tab; TwW *d_tw; uint64_t *d_out; unsigned long long *d_cyc;
  CHECK(hipMalloc(&d_tab, sizeof tab)); CHECK(hipMalloc(&d_tw, sizeof htw)); CHECK(hipMalloc(&d_out, 256 * 1024 * 8)); CHECK(hipMalloc(&d_cyc, 256 * 16 * 8));
  CHECK(hipMemcpy(d_tab, &tab, sizeof tab, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_tw, htw, sizeof htw, hipMemcpyHostToDevice));
  if (argc > 1) {                                     // issue_probe long [waves]: ~2.5 s of the fused-middle mix, for rocm-smi to sample clock and power
    const int w = argc > 2 ? atoi(argv[2]) : 4, reps = 1000;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256 * w), 100 * 1024, 0, d_out, d_tab, d_tw, d_cyc);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    printf("long mix 2 waves/SIMD %d  launches %d  iters %d  waves %d  total %.3f ms\n", w, reps, ITER, 256 * 4 * w, ms);
    return 0;
  }
  for (int w : {4, 3, 2, 1}) {
    run<0>("forward group (12 wide-split CT butterflies)", w, d_out, d_tab, d_tw, d_cyc);
    run<1>("inverse group (12 wide GS butterflies) + csub", w, d_out, d_tab, d_tw, d_cyc);
    run<2>("forward + 8 lazy products + inverse", w, d_out, d_tab, d_tw, d_cyc);
  }
  return 0;
}
