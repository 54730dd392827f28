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
// Split-twiddle multiply (below): one fold finishes, and the butterflies' ranges (forward < 6p, inverse < 3p) close, when
// 3.5*2^30*c < 2^60, i.e. c < 2^30/3.5 = 2^28.19.  (Round 5: until then the class kept data < 4p / < 2p, which needs 3*2^30*c < 2^59 --
// c < 2^27.415 -- and left the last 13 of the 44 limbs of n = 2^17 to the 7-mad class: +25 % time per limb, profiles/r05.)
#define GPQ_SPLIT_CMAX 306000000u /* c < this: every prime of the chains up to n = 2^17 */
// Forward butterflies that skip every other conditional subtraction (ct_bfly_wide below) need c < 2^27.
#define GPQ_WIDE_CMAX 134217000u
// The margins of the split class are narrow (0.25 % and 2.5 %): raising GPQ_SPLIT_CMAX must fail here, not in a product.
//   7 * 2^29 * c < 2^60   one fold finishes: a multiplicand < 6p has th <= 3.5 * 2^30 and t = c*th + 2^59 + (c+1) stays below 3p
//   12 * c < 2^32         variable * variable products of the middle kernels, left < 2p, right < 6p (TwTraits<TwS>): the 7-mad fold's th fits 32 bits
//   12 * c^2 < 2^60       ... and such a product leaves below 4p (tests/test_lazy_ranges.py: the integer model of all three)
static_assert(7ull * (1ull << 29) * GPQ_SPLIT_CMAX < (1ull << 60), "GPQ_SPLIT_CMAX: the single fold of mulmod_split no longer finishes below 3p");
static_assert(12ull * GPQ_SPLIT_CMAX < (1ull << 32), "GPQ_SPLIT_CMAX: th of mulmod_lazy(left < 2p, right < 6p) no longer fits 32 bits");
static_assert(12ull * GPQ_SPLIT_CMAX * GPQ_SPLIT_CMAX < (1ull << 60), "GPQ_SPLIT_CMAX: a product of the split class no longer leaves below 4p");
static_assert(GPQ_WIDE_CMAX <= (1u << 27) && GPQ_WIDE_CMAX <= GPQ_SPLIT_CMAX && GPQ_SPLIT_CMAX <= GPQ_FOLD_CMAX, "butterfly classes must nest: wide <= split <= 7-mad");

#ifndef GPQ_PIN_VOLATILE
#define GPQ_PIN_VOLATILE volatile
#endif

namespace gpq {

typedef unsigned __int128 u128;

struct PrimeK {        // per-limb constants handed to kernels (uniform per block)
  uint64_t p;          // modulus
  uint64_t p2, p4;     // 2p, 4p
  uint64_t kx0, kx1;   // c+1 and c+1-4p (mod 2^64): the two addends of the CT x-leg select
  uint64_t ky;         // 4p - 2(c+1)
  uint64_t kys;        // wide-split forward butterflies: 2p - 2(c+1)
  uint64_t p3, np3;    // 3p and -3p (mod 2^64): the split class subtracts 3p (forward data < 6p, inverse data < 3p)
  uint64_t kx1x, kyx;  // split-twiddle forward butterflies: c+1-3p (mod 2^64) and 3p - 2(c+1)
  uint64_t np, np2, np4;  // -p, -2p, -4p (mod 2^64): a conditional subtraction is an ADD of one of these or of zero (csub1/2/4)
  uint32_t c;          // p - 2^59
  uint32_t c1;         // c + 1
};

__device__ __forceinline__ uint64_t mad_u64(uint32_t a, uint32_t b, uint64_t acc) {
  return (uint64_t)a * b + acc;  // v_mad_u64_u32
}
__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
// ~x & 0x7ffffff in ONE instruction (v_bitop3_b32, truth table 0x0c = ~a & b).  Written as plain C the compiler emits
// v_and + v_xor with two 32-bit literals inside the big kernels (VOP3 takes no literal and it would not spend an SGPR on
// the mask): one VALU instruction of 21.5 per butterfly, -3 % on tensor_mid8 and +1.7 % he_mul/s in an interleaved A/B
// (profiles/r02/v6_instruction_trims_ab.txt).  The same table shows what did NOT pay: forming the carry word {x >> 32, 0} of
// a column sum with one v_pk_mov_b32 (op_sel picks the high dword, the inline constant 0 the other half) instead of the two
// v_mov the middle kernels spend on it removes another 2-5 % of their VALU instructions and makes them 3 % SLOWER: the packed
// move issues like a 64-bit operation.
__device__ __forceinline__ uint32_t not_low27(uint32_t x) { return __builtin_amdgcn_bitop3_b32(x, 0x7ffffffu, x, 0x0c); }

// Core of the modular multiply: returns T' with  a*w == T' + (c+1)  (mod p).
//   x = a*w by 32-bit pieces (a < 2^62.1, w < 2^60: the middle column cannot overflow),
//   xh = x >> 59, xl = x mod 2^59, t = c*xh, th = t >> 59, tl = t mod 2^59,
//   x == xl - t == xl + c*th - tl == xl + c*th + (2^59-1-tl) + (c+1) - p.
// The complement 2^59-1-tl is two bit operations where -tl would be a 64-bit subtract.
// T' < 2^59 + 2^59 + 2^59.5 < 3.42p - (c+1).
template <bool PIN>
__device__ __forceinline__ uint64_t mulmod_raw_t(uint64_t a, uint64_t w, const PrimeK &k) {
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
  const uint64_t m00 = mad_u64(a0, w0, 0);
  uint64_t mid = mad_u64(a0, w1, (uint32_t)(m00 >> 32));
  // PIN keeps the carry inside the mad: stops hipcc re-associating it into a mad + 64-bit add.
  // Measured (profiles/r01): +5 % on forward butterflies, a loss on the inverse ones, hence per call site.
  if (PIN) asm GPQ_PIN_VOLATILE("" : "+v"(mid));
  mid = mad_u64(a1, w0, mid);                                   // < 2^60 + 2^62.1 : fits
  const uint64_t hi = mad_u64(a1, w1, (uint32_t)(mid >> 32));   // x >> 64
  const uint32_t midlo = (uint32_t)mid, hilo = (uint32_t)hi, hihi = (uint32_t)(hi >> 32);
  const uint32_t xh0 = __builtin_amdgcn_alignbit(hilo, midlo, 27);
  const uint32_t xh1 = __builtin_amdgcn_alignbit(hihi, hilo, 27);
  const uint64_t xl = pack64((uint32_t)m00, midlo & 0x7ffffffu);
  const uint64_t t0 = mad_u64(k.c, xh0, 0);
  const uint64_t t1 = mad_u64(k.c, xh1, (uint32_t)(t0 >> 32)); // t = t1 : lo32(t0)
  const uint32_t t1lo = (uint32_t)t1, t1hi = (uint32_t)(t1 >> 32);
  const uint32_t th = __builtin_amdgcn_alignbit(t1hi, t1lo, 27);
  const uint64_t ntl = pack64(~(uint32_t)t0, not_low27(t1lo));  // 2^59 - 1 - tl
  return mad_u64(k.c, th, xl) + ntl;
}

#ifndef GPQ_PIN_CT
#define GPQ_PIN_CT true
#endif
#ifndef GPQ_PIN_GS
#define GPQ_PIN_GS true
#endif
__device__ __forceinline__ uint64_t mulmod_raw(uint64_t a, uint64_t w, const PrimeK &k) { return mulmod_raw_t<false>(a, w, k); }

// a*w mod p, lazily reduced: a < 8p, w < p -> (0, 4p)
__device__ __forceinline__ uint64_t mulmod_lazy(uint64_t a, uint64_t w, const PrimeK &k) {
  return mulmod_raw(a, w, k) + k.c1;
}

// x - m if x >= m else x, for m = p, 2p, 4p: compare, select the addend (-m or 0), one 64-bit add = 4 VALU instructions.
// Written as `x >= m ? x - m : x` (or with `0 - m` the compiler can see through) it becomes compare, select, v_sub_co,
// v_subb_co: 5 instructions and a carry hazard.  The negated moduli come from the table so that they stay opaque.
__device__ __forceinline__ uint64_t csub_by(uint64_t x, uint64_t m, uint64_t neg_m) { return x + (x >= m ? neg_m : (uint64_t)0); }
__device__ __forceinline__ uint64_t csub1(uint64_t x, const PrimeK &k) { return csub_by(x, k.p, k.np); }
__device__ __forceinline__ uint64_t csub2(uint64_t x, const PrimeK &k) { return csub_by(x, k.p2, k.np2); }
__device__ __forceinline__ uint64_t csub4(uint64_t x, const PrimeK &k) { return csub_by(x, k.p4, k.np4); }
__device__ __forceinline__ uint64_t csub3(uint64_t x, const PrimeK &k) { return csub_by(x, k.p3, k.np3); }

// value < 8p -> [0,p)
__device__ __forceinline__ uint64_t canon8(uint64_t x, const PrimeK &k) {
  return csub1(csub2(csub4(x, k), k), k);
}
// value < 4p -> [0,p)
__device__ __forceinline__ uint64_t canon4(uint64_t x, const PrimeK &k) {
  return csub1(csub2(x, k), k);
}

// Cooley-Tukey butterfly of src/ntt.c:45-49 in lazy form.
// in: x,y < 8p ; out: x,y < 8p.  With t = T' + (c+1) and xr = csub(x, 4p):
//   x' = xr + t = xs + T',  y' = xr + 4p - t = xs + (4p - 2(c+1)) - T',  xs = xr + (c+1)
// so the (c+1) rides on the constant the conditional subtract selects anyway.
__device__ __forceinline__ void ct_bfly(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t t = mulmod_raw_t<GPQ_PIN_CT>(y, w, k);
  const uint64_t xs = x + (x >= k.p4 ? k.kx1 : k.kx0);
  x = xs + t;
  y = xs + k.ky - t;
}

// Gentleman-Sande butterfly of src/ntt.c:63-68 in lazy form.
// in: x,y < 4p ; out: x,y < 4p
__device__ __forceinline__ void gs_bfly(uint64_t &x, uint64_t &y, uint64_t w, const PrimeK &k) {
  const uint64_t v = x + y;
  const uint64_t d = x + k.p4 - y;  // (0, 8p)
  x = csub4(v, k);
  y = mulmod_raw_t<GPQ_PIN_GS>(d, w, k) + k.c1;
}

// ---------------------------------------------------------------------------
// Split-twiddle multiply: when the multiplier is a table constant w, the table also
// carries w2 = w*2^31 mod p, and with a = ah*2^31 + al
//   a*w == al*w + ah*w2  (mod p),   al < 2^31, ah <= 2^30 for a < 4p
// is a 92-bit sum of two 32x60 products (4 mads, the column carries ride in the mad
// addends) whose part above 2^59 fits 32 bits: ONE fold by c finishes.  5 v_mad_u64_u32
// instead of 7 and no 128-bit shift.  Both constants are stored negated (p-w, p-w2), so
//   a*w == -(tl + 2^59*th) == c*th - tl == c*th + (2^59-1-tl) + (c+1) - p
// and the subtraction is again a complement.  Returns T' with a*w == T' + (c+1) (mod p).
// For a multiplicand a < 6p:  ah <= 3*2^29,  th <= 3.5*2^30,  T' <= c*th + 2^59 - 1 < 3p - (c+1)   for c < GPQ_SPLIT_CMAX,
// so t = T' + (c+1) < 3p and the ranges [0, 6p) forward / [0, 3p) inverse close with ONE conditional subtraction (of 3p) per
// butterfly:  forward  xr = x - [x >= 3p] 3p < 3p,  x' = xr + t < 6p,  y' = xr + 3p - t in (0, 6p);
//             inverse  v = x + y < 6p,  x' = v - [v >= 3p] 3p < 3p,  d = x + 3p - y in (0, 6p),  y' = t < 3p.
// (tests/test_lazy_ranges.py is the integer model; measured in tools/bfly_lab at the 4p / 2p ranges: +20 % CT, +27 % GS over the 7-mad form.)
// ---------------------------------------------------------------------------
typedef ulonglong2 TwS;   // .x = p - w, .y = p - (w*2^31 mod p)

// Same storage, other butterflies: the forward stages of limbs with c < GPQ_WIDE_CMAX (ct_bfly_wide).
struct alignas(16) TwW { uint64_t x, y; };

template <typename W>
__device__ __forceinline__ uint64_t mulmod_split(uint64_t a, const W &w, const PrimeK &k) {
  const uint32_t al = (uint32_t)a & 0x7fffffffu;
  const uint32_t ah = __builtin_amdgcn_alignbit((uint32_t)(a >> 32), (uint32_t)a, 31);
  uint64_t t0 = mad_u64(al, (uint32_t)w.x, 0);
  t0 = mad_u64(ah, (uint32_t)w.y, t0);                         // < 2^63 + 2^63 - ... : al < 2^31, ah < 2^31 (a < 8p), constants < 2^32
  asm GPQ_PIN_VOLATILE("" : "+v"(t0));                                 // keep the carries inside the mads (see mulmod_raw_t)
  uint64_t t1 = mad_u64(al, (uint32_t)(w.x >> 32), (uint32_t)(t0 >> 32));
  asm GPQ_PIN_VOLATILE("" : "+v"(t1));
  t1 = mad_u64(ah, (uint32_t)(w.y >> 32), t1);                 // sum >> 32, < 0.75p
  const uint32_t t1lo = (uint32_t)t1, t1hi = (uint32_t)(t1 >> 32);
  const uint32_t th = __builtin_amdgcn_alignbit(t1hi, t1lo, 27);
  const uint64_t ntl = pack64(~(uint32_t)t0, not_low27(t1lo));  // 2^59 - 1 - tl
  return mad_u64(k.c, th, ntl);
}

// Cooley-Tukey, split twiddle.  in: x,y < 6p ; out: x,y < 6p.
__device__ __forceinline__ void ct_bfly(uint64_t &x, uint64_t &y, const TwS &w, const PrimeK &k) {
  const uint64_t t = mulmod_split(y, w, k);
  const uint64_t xs = x + (x >= k.p3 ? k.kx1x : k.kx0);
  x = xs + t;
  y = xs + k.kyx - t;
}
// Gentleman-Sande, split twiddle.  in: x,y < 3p ; out: x,y < 3p.
template <typename W>
__device__ __forceinline__ void gs_bfly_split(uint64_t &x, uint64_t &y, const W &w, const PrimeK &k) {
  const uint64_t v = x + y;
  const uint64_t d = x + k.p3 - y;  // (0, 6p)
  x = csub3(v, k);
  y = mulmod_split(d, w, k) + k.c1;
}
__device__ __forceinline__ void gs_bfly(uint64_t &x, uint64_t &y, const TwS &w, const PrimeK &k) { gs_bfly_split(x, y, w, k); }

// Gentleman-Sande for the wide class (c < GPQ_WIDE_CMAX): inverse data lives in [0, 4p) instead of [0, 2p).
//   v = x + y < 8p;   d = x + 4p - y in (0, 8p): a legal multiplicand (8p - 1 = 2^62 + 8c - 1 <= 2^62 + 2^31 - 2 for c < 2^27),
//   so the product leg y' = T' + (c+1) < 2p as in ct_bfly_wide;   sum leg x' = v - [v >= 4p] 4p < 4p.
// What the wider range buys: a butterfly whose two inputs are BOTH product legs of the stage before (each < 2p) has
// v < 4p already and needs no conditional subtraction (CSUB = false).  Inside a register group that is known at compile time
// -- the pair (e, e + 2^B) of stage B shares bit B-1, and bit B-1 set means "product leg of the previous stage" -- so half
// of the butterflies of every stage but a group's first drop their compare / select / add (a quarter of a GS butterfly's
// issue cost).  The products of the fused kernels enter at < 4p without any subtraction either.
template <bool CSUB>
__device__ __forceinline__ void gs_bfly_wide(uint64_t &x, uint64_t &y, const TwW &w, const PrimeK &k) {
  const uint64_t v = x + y;
  const uint64_t d = x + k.p4 - y;
  x = CSUB ? csub4(v, k) : v;
  y = mulmod_split(d, w, k) + k.c1;
}
__device__ __forceinline__ void gs_bfly(uint64_t &x, uint64_t &y, const TwW &w, const PrimeK &k) { gs_bfly_wide<true>(x, y, w, k); }

// Cooley-Tukey, split twiddle, one conditional subtraction per TWO stages (c < GPQ_WIDE_CMAX <= 2^27).
// The split multiply only needs th < 2^32, i.e. al + ah <= 2^32 - 2, which holds for every multiplicand
// a <= 2^62 + 2^31 - 2 (above 2^62 the low part al is what exceeds 2^62: small), and then
//   t = T' + (c+1) <= c (2^32 - 2) + 2^59 + c < 2p          for c <= 2^27.
// Stage kinds alternate by the position s of the index bit a stage works on (len = 2^s):
//   s odd  "A": no subtraction.   in: x, y < 6p          out: x' = x + t < 8p,  y' = x + 2p - t <= 8p - c - 2
//   s even "B": subtract 4p.      in: x, y < 8p          out: x' = xr + t < 6p, y' = xr + 2p - t < 6p     (xr = csub(x, 4p) < 4p)
// 8p - c - 2 = 2^62 + 7c - 2 is a legal multiplicand (7c < 2^31).  Two A stages never follow each other; the
// last stage (s = 0) is B, so a finished forward transform is < 6p; canonical inputs may enter at either kind.
// Measured (tools/bfly_lab): +8 % over a subtraction in every stage.
template <bool CSUB>
__device__ __forceinline__ void ct_bfly_wide(uint64_t &x, uint64_t &y, const TwW &w, const PrimeK &k) {
  const uint64_t t = mulmod_split(y, w, k);
  const uint64_t xs = x + (CSUB ? (x >= k.p4 ? k.kx1 : k.kx0) : k.kx0);
  x = xs + t;
  y = xs + k.kys - t;
}

// Exact a*b mod p for canonical a,b (poly_rns_mul, src/poly.c:77-82).
__device__ __forceinline__ uint64_t mulmod_canon(uint64_t a, uint64_t b, const PrimeK &k) {
  return canon4(mulmod_lazy(a, b, k), k);
}
// v mod p for v < 16p without compares: v = q 2^59 + low, v - q p = low - q c in (-16c, 2^59); plus p when negative.
// (7 VALU instructions and no VCC/SGPR-pair hazards, against 5 per conditional subtraction.)
__device__ __forceinline__ uint64_t canon_fold(uint64_t v, uint64_t p, uint32_t c) {
  const int q = (int)(v >> 59);
  int64_t r = (int64_t)(-(int)c) * q + (int64_t)(v & ((1ull << 59) - 1));
  r += (r >> 63) & (int64_t)p;
  return (uint64_t)r;
}

// Exact a*b mod p for a < 8p, b canonical.
__device__ __forceinline__ uint64_t mulmod_canon_lazy(uint64_t a, uint64_t b, const PrimeK &k) {
  return canon4(mulmod_lazy(a, b, k), k);
}
// Exact a+b mod p for a, b in [0, p] (poly_rns_add, src/poly.c:71-76: barrett_reduce of the sum is canonical; operands that
// come out of ntt may be p itself, src/ntt.c:47).
__device__ __forceinline__ uint64_t addmod_canon(uint64_t a, uint64_t b, const PrimeK &k) {
  return canon4(a + b, k);
}

}  // namespace gpq
