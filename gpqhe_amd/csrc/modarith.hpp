// modarith.hpp -- 64-bit modular arithmetic for the GPQHE prime tower on CDNA4.
//
// Replaces the reference's word-level reductions on the device side:
//   montgomery_reduce  src/reduce.c:59-66   (twiddle multiply in ntt/invntt, src/ntt.c:32-35)
//   barrett_reduce     src/reduce.c:88-106  (poly_rns_mul / poly_rns_add, src/poly.c:71-82)
// Only canonical outputs are compared with the reference, so the reduction
// algorithm is free.  Every prime of the reference's chain has the shape
//   p = 2^59 + c,  c = 1 + k*2n  small      (src/precomp.c:358, :372-376)
// so 2^59 == -c (mod p) and a 122-bit product folds to 64 bits with three
// 32x32 multiplies instead of the seven of a Montgomery REDC.  With the four
// of the product itself a modular multiply is 7 v_mad_u64_u32.
//
// Lazy ranges (Harvey-style), all values unsigned 64-bit:
//   mulmod_lazy(a, w): a < 8p, w < p   ->  result in (0, 4p), == a*w (mod p)
//   butterflies keep data in [0, 8p) (forward) / [0, 4p) (inverse)
// Bounds hold for c < GPQ_FOLD_CMAX (checked at context creation):
//   x = a*w < 8p^2, xh = x>>59 < 8p(1+c/2^59) < 2^62.01
//   t = c*xh < 2^90.3, th = t>>59 < 2^31.3 (fits 32 bits), u = c*th < 8c^2(1+eps) < 2^59.5
//   r = xl + u + p - tl  in (p-2^59, 2^59 + 2^59.5 + p)  subset (0, 3.42p)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GPQ_FOLD_CMAX 319000000u /* < 2^28.25 */

namespace gpq {

typedef unsigned __int128 u128;

struct PrimeK {        // per-limb constants handed to kernels (uniform per block)
  uint64_t p;          // modulus
  uint64_t p2, p4;     // 2p, 4p
  uint32_t c;          // p - 2^59
  uint32_t pad;
};

__device__ __forceinline__ uint64_t mad_u64(uint32_t a, uint32_t b, uint64_t acc) {
  return (uint64_t)a * b + acc;  // v_mad_u64_u32
}

// a*w mod p, lazily reduced; see header for ranges.
__device__ __forceinline__ uint64_t mulmod_lazy(uint64_t a, uint64_t w, const PrimeK &k) {
  const u128 x = (u128)a * w;
  const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  const uint64_t xh = (hi << 5) | (lo >> 59);
  const uint64_t xl = lo & ((1ull << 59) - 1);
  const uint64_t t0 = mad_u64(k.c, (uint32_t)xh, 0);
  const uint64_t t1 = mad_u64(k.c, (uint32_t)(xh >> 32), t0 >> 32);  // t = t1 : lo32(t0)
  const uint64_t tl = ((t1 & ((1u << 27) - 1)) << 32) | (uint32_t)t0; // t mod 2^59
  const uint32_t th = (uint32_t)(t1 >> 27);                           // t >> 59
  return mad_u64(k.c, th, xl) + (k.p - tl);
}

// x - m if x >= m else x
__device__ __forceinline__ uint64_t csub(uint64_t x, uint64_t m) {
  const uint64_t d = x - m;
  return x >= m ? d : x;
}

// value < 8p -> [0,p)
__device__ __forceinline__ uint64_t canon8(uint64_t x, const PrimeK &k) {
  return csub(csub(csub(x, k.p4), k.p2), k.p);
}
// value < 4p -> [0,p)
__device__ __forceinline__ uint64_t canon4(uint64_t x, const PrimeK &k) {
  return csub(csub(x, k.p2), k.p);
}

// Cooley-Tukey butterfly of src/ntt.c:45-49 in lazy form.
// in: x,y < 8p ; out: x,y < 8p
__device__ __forceinline__ void ct_bfly(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t t = mulmod_lazy(y, w, k);  // < 4p
  const uint64_t xr = csub(x, k.p4);        // < 4p
  x = xr + t;
  y = xr + k.p4 - t;
}

// Gentleman-Sande butterfly of src/ntt.c:63-68 in lazy form.
// in: x,y < 4p ; out: x,y < 4p
__device__ __forceinline__ void gs_bfly(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t s = csub(x + y, k.p4);
  const uint64_t d = x + k.p4 - y;  // (0, 8p)
  x = s;
  y = mulmod_lazy(d, w, k);
}

// Exact a*b mod p for canonical a,b (poly_rns_mul, src/poly.c:77-82).
__device__ __forceinline__ uint64_t mulmod_canon(uint64_t a, uint64_t b, const PrimeK &k) {
  return canon4(mulmod_lazy(a, b, k), k);
}
// Exact a+b mod p for canonical a,b (poly_rns_add, src/poly.c:71-76).
__device__ __forceinline__ uint64_t addmod_canon(uint64_t a, uint64_t b, const PrimeK &k) {
  return csub(a + b, k.p);
}

}  // namespace gpq
