// bfly_lab.hip -- A/B of butterfly formulations in one binary (dev tool).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "modarith.hpp"
using namespace gpq;
typedef unsigned __int128 u128;
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
constexpr int ITER = 4096;

// v0: first formulation (128-bit product by the compiler, 64-bit subtracts)
__device__ __forceinline__ uint64_t mulmod_v0(uint64_t a, uint64_t w, const PrimeK &k) {
  const u128 x = (u128)a * w;
  const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  const uint64_t xh = (hi << 5) | (lo >> 59);
  const uint64_t xl = lo & ((1ull << 59) - 1);
  const uint64_t t0 = mad_u64(k.c, (uint32_t)xh, 0);
  const uint64_t t1 = mad_u64(k.c, (uint32_t)(xh >> 32), t0 >> 32);
  const uint64_t tl = ((t1 & ((1u << 27) - 1)) << 32) | (uint32_t)t0;
  const uint32_t th = (uint32_t)(t1 >> 27);
  return mad_u64(k.c, th, xl) + (k.p - tl);
}
__device__ __forceinline__ void ct_v0(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t t = mulmod_v0(y, w, k);
  const uint64_t xr = csub(x, k.p4);
  x = xr + t; y = xr + k.p4 - t;
}
// v2: v1 + approximate conditional subtract (compare high words only)
__device__ __forceinline__ void ct_v2(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t t = mulmod_raw(y, w, k);
  const bool ge = (uint32_t)(x >> 32) > (uint32_t)(k.p4 >> 32);
  const uint64_t xs = x + (ge ? k.kx1 : k.kx0);
  x = xs + t; y = xs + k.ky - t;
}
// v6: v1 with the middle column pinned (no re-association) ; v7: v6 + approx csub
__device__ __forceinline__ uint64_t mulmod_pin(uint64_t a, uint64_t w, const PrimeK &k) {
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
  const uint64_t m00 = mad_u64(a0, w0, 0);
  uint64_t mid = mad_u64(a0, w1, (uint32_t)(m00 >> 32));
  asm volatile("" : "+v"(mid));
  mid = mad_u64(a1, w0, mid);
  const uint64_t hi = mad_u64(a1, w1, (uint32_t)(mid >> 32));
  const uint32_t midlo = (uint32_t)mid, hilo = (uint32_t)hi, hihi = (uint32_t)(hi >> 32);
  const uint32_t xh0 = __builtin_amdgcn_alignbit(hilo, midlo, 27);
  const uint32_t xh1 = __builtin_amdgcn_alignbit(hihi, hilo, 27);
  const uint64_t xl = pack64((uint32_t)m00, midlo & 0x7ffffffu);
  const uint64_t t0 = mad_u64(k.c, xh0, 0);
  const uint64_t t1 = mad_u64(k.c, xh1, (uint32_t)(t0 >> 32));
  const uint32_t t1lo = (uint32_t)t1, t1hi = (uint32_t)(t1 >> 32);
  const uint32_t th = __builtin_amdgcn_alignbit(t1hi, t1lo, 27);
  const uint64_t ntl = pack64(~(uint32_t)t0, ~t1lo & 0x7ffffffu);
  return mad_u64(k.c, th, xl) + ntl;
}
__device__ __forceinline__ void ct_v6(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t t = mulmod_pin(y, w, k);
  const uint64_t xs = x + (x >= k.p4 ? k.kx1 : k.kx0);
  x = xs + t; y = xs + k.ky - t;
}
__device__ __forceinline__ void ct_v7(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t t = mulmod_pin(y, w, k);
  const bool ge = (uint32_t)(x >> 32) > (uint32_t)(k.p4 >> 32);
  const uint64_t xs = x + (ge ? k.kx1 : k.kx0);
  x = xs + t; y = xs + k.ky - t;
}
// v8: v7 with y' = (xs + ky + 1) + ~t
__device__ __forceinline__ void ct_v8(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t t = mulmod_pin(y, w, k);
  const bool ge = (uint32_t)(x >> 32) > (uint32_t)(k.p4 >> 32);
  const uint64_t xs = x + (ge ? k.kx1 : k.kx0);
  x = xs + t; y = (xs + (k.ky + 1)) + ~t;
}
// v10/v11: two consecutive stages on 4 values (radix-4 shape) -- per-stage csub(4p) vs one csub(6p) per two stages
__device__ __forceinline__ void ct_nocsub(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k, uint64_t kc1, uint64_t ky3) {
  const uint64_t t = mulmod_pin(y, w, k);       // x' = x + t + (c+1), y' = x + 3p - t - (c+1)
  const uint64_t xs = x + kc1;
  x = xs + t; y = xs + ky3 - t;
}
__device__ __forceinline__ void ct_csub6(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k, uint64_t p6, uint64_t k61, uint64_t k60, uint64_t ky3) {
  const uint64_t t = mulmod_pin(y, w, k);
  const uint64_t xs = x + (x >= p6 ? k61 : k60);
  x = xs + t; y = xs + ky3 - t;
}
// v3: no conditional subtract at all (bounds not kept; timing only)
__device__ __forceinline__ void ct_v3(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t t = mulmod_raw(y, w, k);
  const uint64_t xs = x;
  x = xs + t; y = xs + k.ky - t;
}
// v4: multiply only
__device__ __forceinline__ void ct_v4(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  y = mulmod_raw(y, w, k); x ^= y;
}
// v5: only the 4-mad product
__device__ __forceinline__ void ct_v5(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint32_t a0 = (uint32_t)y, a1 = (uint32_t)(y >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
  const uint64_t m00 = mad_u64(a0, w0, 0);
  uint64_t mid = mad_u64(a0, w1, (uint32_t)(m00 >> 32));
  mid = mad_u64(a1, w0, mid);
  const uint64_t hi = mad_u64(a1, w1, (uint32_t)(mid >> 32));
  y = hi ^ mid; x ^= m00;
}

// v12..: split-twiddle multiply.  (w, w2 = w*2^31 mod p) both precomputed, a = ah*2^31 + al:
//   a*w == al*w + ah*w2; for a < 4p the sum is < 2^90.6, so ONE fold (th < 2^31.6) finishes: 5 mads.
// Twiddles negated (p-w, p-w2) so that the fold's subtraction turns into the complement trick.
__device__ __forceinline__ uint64_t mulmod_split(uint64_t a, uint64_t wn, uint64_t w2n, const PrimeK &k) {
  const uint32_t al = (uint32_t)a & 0x7fffffffu;
  const uint32_t ah = __builtin_amdgcn_alignbit((uint32_t)(a >> 32), (uint32_t)a, 31);
  uint64_t t0 = mad_u64(al, (uint32_t)wn, 0);
  t0 = mad_u64(ah, (uint32_t)w2n, t0);
  asm volatile("" : "+v"(t0));
  uint64_t t1 = mad_u64(al, (uint32_t)(wn >> 32), (uint32_t)(t0 >> 32));
  asm volatile("" : "+v"(t1));
  t1 = mad_u64(ah, (uint32_t)(w2n >> 32), t1);
  const uint32_t t1lo = (uint32_t)t1, t1hi = (uint32_t)(t1 >> 32);
  const uint32_t th = __builtin_amdgcn_alignbit(t1hi, t1lo, 27);
  const uint64_t ntl = pack64(~(uint32_t)t0, ~t1lo & 0x7ffffffu);
  return mad_u64(k.c, th, ntl);            // a*w == this + (c+1)  (mod p), < 2p for c < 2^27.3
}
// data in [0,4p): xs = csub(x,2p) + (c+1)
__device__ __forceinline__ void ct_split(uint64_t &x, uint64_t &y, uint64_t wn, uint64_t w2n, const PrimeK &k, uint64_t kx0, uint64_t kx1, uint64_t ky) {
  const uint64_t t = mulmod_split(y, wn, w2n, k);
  const uint64_t xs = x + (x >= k.p2 ? kx1 : kx0);
  x = xs + t; y = xs + ky - t;
}
// alternate-stage variant (wide split class, c < 2^27): stage A adds without the conditional subtract, stage B subtracts 4p
__device__ __forceinline__ void ct_split_nocsub(uint64_t &x, uint64_t &y, uint64_t wn, uint64_t w2n, const PrimeK &k, uint64_t kx0, uint64_t ky) {
  const uint64_t t = mulmod_split(y, wn, w2n, k);
  const uint64_t xs = x + kx0;
  x = xs + t; y = xs + ky - t;
}
__device__ __forceinline__ void ct_split_csub4(uint64_t &x, uint64_t &y, uint64_t wn, uint64_t w2n, const PrimeK &k, uint64_t kx0, uint64_t kx14, uint64_t ky) {
  const uint64_t t = mulmod_split(y, wn, w2n, k);
  const uint64_t xs = x + (x >= k.p4 ? kx14 : kx0);
  x = xs + t; y = xs + ky - t;
}
// data in [0,2p)
__device__ __forceinline__ void gs_split(uint64_t &x, uint64_t &y, uint64_t wn, uint64_t w2n, const PrimeK &k) {
  const uint64_t v = x + y;
  const uint64_t d = x + k.p2 - y;
  x = v + (v >= k.p2 ? (uint64_t)0 - k.p2 : (uint64_t)0);
  y = mulmod_split(d, wn, w2n, k) + k.c1;
}

template <int V>
__global__ __launch_bounds__(256) void probe(uint64_t *out, uint64_t seed, PrimeK k, PrimeK k2) {
  uint64_t v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = seed * (threadIdx.x + 1 + 64 * i) + blockIdx.x;
  const uint64_t p6 = 6 * k.p, ky3 = 3 * k.p - 2 * (uint64_t)k.c1, k60 = k.c1, k61 = (uint64_t)k.c1 - p6;
  if (V == 10 || V == 11 || V == 15 || V == 16) {
    for (int it = 0; it < ITER / 2; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] &= (V >= 15 ? 0x0fffffffffffffffull : 0x3fffffffffffffffull);
      // stage 1: pairs (0,2),(1,3),(4,6),(5,7); stage 2: pairs (0,1),(2,3),(4,5),(6,7)
#pragma unroll
      for (int g = 0; g < 8; g += 4) {
        const uint64_t w = k.p - 3 - g;
        const uint64_t w2 = w - 77 - seed, kx14 = (uint64_t)k.c1 - k.p4;
        if (V == 15) { ct_split(v[g], v[g + 2], w, w2, k, k2.kx0, k2.kx1, k2.ky); ct_split(v[g + 1], v[g + 3], w, w2, k, k2.kx0, k2.kx1, k2.ky);
                       ct_split(v[g], v[g + 1], w - 1, w2 - 1, k, k2.kx0, k2.kx1, k2.ky); ct_split(v[g + 2], v[g + 3], w - 2, w2 - 2, k, k2.kx0, k2.kx1, k2.ky); }
        else if (V == 16) { ct_split_nocsub(v[g], v[g + 2], w, w2, k, k2.kx0, k2.ky); ct_split_nocsub(v[g + 1], v[g + 3], w, w2, k, k2.kx0, k2.ky);
                            ct_split_csub4(v[g], v[g + 1], w - 1, w2 - 1, k, k2.kx0, kx14, k2.ky); ct_split_csub4(v[g + 2], v[g + 3], w - 2, w2 - 2, k, k2.kx0, kx14, k2.ky); }
        else if (V == 10) { ct_v6(v[g], v[g + 2], w, k); ct_v6(v[g + 1], v[g + 3], w, k); ct_v6(v[g], v[g + 1], w - 1, k); ct_v6(v[g + 2], v[g + 3], w - 2, k); }
        else { ct_nocsub(v[g], v[g + 2], w, k, k60, ky3); ct_nocsub(v[g + 1], v[g + 3], w, k, k60, ky3);
               ct_csub6(v[g], v[g + 1], w - 1, k, p6, k61, k60, ky3); ct_csub6(v[g + 2], v[g + 3], w - 2, k, p6, k61, k60, ky3); }
      }
    }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= v[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    return;
  }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] &= 0x3fffffffffffffffull; v[i + 4] &= 0x3fffffffffffffffull;
      const uint64_t w = k.p - 3 - i;
      if (V == 0) ct_v0(v[i], v[i + 4], w, k);
      if (V == 1) ct_bfly(v[i], v[i + 4], w, k);
      if (V == 2) ct_v2(v[i], v[i + 4], w, k);
      if (V == 3) ct_v3(v[i], v[i + 4], w, k);
      if (V == 4) ct_v4(v[i], v[i + 4], w, k);
      if (V == 5) ct_v5(v[i], v[i + 4], w, k);
      if (V == 6) gs_bfly(v[i], v[i + 4], w, k);
      if (V == 7) ct_v6(v[i], v[i + 4], w, k);
      if (V == 8) ct_v7(v[i], v[i + 4], w, k);
      if (V == 9) ct_v8(v[i], v[i + 4], w, k);
      if (V == 12) { v[i] &= 0x0fffffffffffffffull; v[i + 4] &= 0x0fffffffffffffffull; ct_split(v[i], v[i + 4], w, w - 77 - seed, k, k2.kx0, k2.kx1, k2.ky); }
      if (V == 13) { v[i] &= 0x0fffffffffffffffull; v[i + 4] &= 0x0fffffffffffffffull; gs_split(v[i], v[i + 4], w, w - 77 - seed, k); }
      if (V == 14) { v[i + 4] = mulmod_split(v[i + 4] & 0x0fffffffffffffffull, w, w - 77 - seed, k); v[i] ^= v[i + 4]; }
    }
  }
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= v[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V>
int run(const char *name, uint64_t *d_out, PrimeK k, int blocks_per_cu) {
  PrimeK k2 = k; k2.kx0 = k.c1; k2.kx1 = (uint64_t)k.c1 - k.p2; k2.ky = k.p2 - 2 * (uint64_t)k.c1;
  const int blocks = 256 * blocks_per_cu;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, d_out, 0x9e3779b97f4a7c15ull, k, k2);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, d_out, 0x9e3779b97f4a7c15ull + r, k, k2);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  double total = 5.0 * blocks * 256.0 * ITER * 4;
  printf("%-34s waves/SIMD=%d %8.3f ms  %8.1f Gbfly/s\n", name, blocks_per_cu, ms / 5, total / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  uint64_t *d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 8));
  PrimeK k; k.p = 576460752308273153ull; k.p2 = 2 * k.p; k.p4 = 4 * k.p; k.c = (uint32_t)(k.p - (1ull << 59)); k.c1 = k.c + 1;
  k.kx0 = k.c1; k.kx1 = (uint64_t)k.c1 - k.p4; k.ky = k.p4 - 2 * (uint64_t)k.c1;
  for (int w : {8, 4, 2}) {
    if (w == 8) { run<0>("ct v0 (first formulation)", d_out, k, 8); run<1>("ct v1 (current)", d_out, k, 8); run<2>("ct v2 (approx csub)", d_out, k, 8);
                  run<3>("ct v3 (no csub)", d_out, k, 8); run<4>("mulmod only", d_out, k, 8); run<5>("4-mad product only", d_out, k, 8); run<6>("gs (current)", d_out, k, 8);
                  run<7>("ct v6 (pinned mid)", d_out, k, 8); run<8>("ct v7 (pinned + approx csub)", d_out, k, 8); run<9>("ct v8 (v7, y via ~t)", d_out, k, 8);
                  run<12>("ct split-twiddle (5 mads)", d_out, k, 8); run<13>("gs split-twiddle (5 mads)", d_out, k, 8); run<14>("mulmod split only", d_out, k, 8);
                  run<10>("2 stages, csub(4p) each", d_out, k, 8); run<11>("2 stages, one csub(6p)", d_out, k, 8);
                  run<15>("split, 2 stages, csub(2p) each", d_out, k, 8); run<16>("split, 2 stages, one csub(4p)", d_out, k, 8); }
    if (w == 4) { run<0>("ct v0", d_out, k, 4); run<1>("ct v1", d_out, k, 4); run<15>("split, 2 stages, csub(2p) each", d_out, k, 4); run<16>("split, 2 stages, one csub(4p)", d_out, k, 4); }
    if (w == 2) { run<0>("ct v0", d_out, k, 2); run<1>("ct v1", d_out, k, 2); }
  }
  return 0;
}
