// ntt_kernels.hpp -- device side of the negacyclic NTT / INTT over the RNS
// prime tower (reference: src/ntt.c:37-73) and of the fused he_mul RNS core
// (reference limb loops: src/he-mult.c:116-138 and :58-66).
//
// Decomposition.  The reference runs logn radix-2 stages over one limb
// (len = n/2 .. 1 forward, 1 .. n/2 inverse), each stage pairing a[j] with
// a[j+len] and using zetas[n/(2len) + j/(2len)].  The same butterflies at the
// same positions are grouped here into two passes so that every coefficient
// crosses HBM twice per transform instead of logn times:
//
//   strided pass     stages with len >= 256: element index i = r*256 + col;
//                    a workgroup owns 2^M1 rows x 16 columns (M1 = logn-8)
//   contiguous pass  stages with len <= 128: inside 256-element blocks;
//                    a wave owns 4 consecutive blocks, no workgroup barrier
//   (n = 2^17: len >= 512 / len <= 256, rows of 512 and blocks of 512: same
//   256-row tiles, 9 low stages in the 8-per-lane kernels further down)
// The low stages are fused with what surrounds them wherever a limb loop of
// the reference allows it: tensor_mid8 (he_mul tensor stage),
// keyswitch_mid8x2 (he_relin, he_swk), polymul_mid8 (poly_mul),
// mulpt_mid8 (he_mulpt).
//
// Inside a pass a thread keeps 2^EL coefficients in registers and runs EL
// (or fewer) stages on them before exchanging through LDS (padded, conflict
// free both ways).  Arithmetic is lazy (modarith.hpp); only the last stage of
// a transform canonicalises, so results are bit-identical with src/ntt.c.
//
// Twiddle tables on the device are in standard form (the reference stores
// zeta*2^64 mod p; converted once at context creation) and keep the
// reference's bit-reversed indexing, so index formulas below read like
// src/ntt.c: stage with len uses table[n/(2len) + i/(2len)].
#pragma once
#include "modarith.hpp"
#include "tables.hpp"

namespace gpq {

// One slab word with the launch's cache policy (PassArgs::nt picks the kernel's NT instantiation on the host).  A template parameter, not an
// argument: as a run-time select the compiler folds the two loads into one plain load before inlining and the modifier is gone.
template <bool NT> __device__ __forceinline__ uint64_t slab_ld(const uint64_t *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT> __device__ __forceinline__ void slab_st(uint64_t *p, uint64_t v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}


// ---------------------------------------------------------------------------
// Register groups.  x[] holds E = 2^EL coefficients; register bit b stands for
// global index bit (rs + b); ibase is the global index of x[0].  A stage on
// register bit b is the reference stage with len = 2^(rs+b).
// ---------------------------------------------------------------------------
// Number of distinct twiddles a group over register bits BHI..BLO touches.
constexpr int tw_count(int EL, int BHI, int BLO) { return (1 << (EL - BLO)) - (1 << (EL - BHI - 1)); }

// Fetch the twiddles of a group into registers ahead of the butterflies: stage on
// bit b uses table[n/(2 len) + i/(2 len)], i.e. 2^(EL-b-1) consecutive entries per thread.
// Offset of bit b's twiddles inside a group's tw[] (bits above b come first).
constexpr int tw_off(int EL, int BHI, int b) { return (1 << (EL - b - 1)) - (1 << (EL - BHI - 1)); }

// The stages of a group are unrolled by template recursion so that every
// register index is a compile-time constant (a plain nested loop over 32
// registers is not always unrolled completely, and a runtime-indexed register
// array goes to scratch: measured 8x slower on the n=2^17 forward pass).
// TW is the table entry type: uint64_t (plain twiddle, 7-mad multiply, data < 8p / < 4p) or
// TwS (split pair, 5-mad multiply, data < 6p / < 3p); the butterflies are overloaded on it.
template <int EL, int BHI, int B, int BLO, bool UNIFORM, int NTW, typename TW>
__device__ __forceinline__ void load_tw_bits(TW (&tw)[NTW], unsigned ibase, int rs, unsigned logn,
                                             const TW *__restrict__ w) {
  const int sh = rs + B + 1;                                          // i / (2 len)
  const TW *wp = w + (1u << (logn - sh)) + (UNIFORM ? 0u : (ibase >> sh));
#pragma unroll
  for (int j = 0; j < (1 << (EL - B - 1)); ++j) tw[tw_off(EL, BHI, B) + j] = wp[j];
  if constexpr (B > BLO) load_tw_bits<EL, BHI, B - 1, BLO, UNIFORM>(tw, ibase, rs, logn, w);
}

// Fetch the twiddles of a group into registers ahead of the butterflies: stage on
// bit b uses table[n/(2 len) + i/(2 len)], i.e. 2^(EL-b-1) consecutive entries per thread.
template <int EL, int BHI, int BLO, bool UNIFORM, typename TW>
__device__ __forceinline__ void load_tw(TW (&tw)[tw_count(EL, BHI, BLO)], unsigned ibase, int rs, unsigned logn,
                                        const TW *__restrict__ w) {
  load_tw_bits<EL, BHI, BHI, BLO, UNIFORM>(tw, ibase, rs, logn, w);
}

// One forward butterfly of the stage on index bit S (len = 2^S).  Only the wide-split butterflies care about S.
// (Leaving out the even stage's conditional subtraction among a transform's first two stages -- canonical inputs never
// trigger it -- was measured: -2 % instructions in the forward strided pass, no change in its time: that pass waits on
// memory, profiles/r02/v6_instruction_trims_ab.txt.)
template <int S, typename TW>
__device__ __forceinline__ void ct_stage(uint64_t &x, uint64_t &y, const TW &w, const PrimeK &k) { ct_bfly(x, y, w, k); }
template <int S>
__device__ __forceinline__ void ct_stage(uint64_t &x, uint64_t &y, const TwW &w, const PrimeK &k) { ct_bfly_wide<(S & 1) == 0>(x, y, w, k); }

// RS = index bit that register bit 0 stands for.
template <int EL, int BHI, int B, int BLO, int RS, int NTW, typename TW>
__device__ __forceinline__ void ct_bits(uint64_t (&x)[1 << EL], const TW (&tw)[NTW], const PrimeK &k) {
#pragma unroll
  for (int e = 0; e < (1 << EL); ++e)
    if (!(e & (1 << B))) ct_stage<RS + B>(x[e], x[e + (1 << B)], tw[tw_off(EL, BHI, B) + (e >> (B + 1))], k);
  if constexpr (B > BLO) ct_bits<EL, BHI, B - 1, BLO, RS>(x, tw, k);
}
template <int EL, int BHI, int BLO, int RS, typename TW>
__device__ __forceinline__ void ct_group(uint64_t (&x)[1 << EL], const TW (&tw)[tw_count(EL, BHI, BLO)],
                                         const PrimeK &k) {
  ct_bits<EL, BHI, BHI, BLO, RS>(x, tw, k);
}

// One inverse butterfly; BOTH_PRODUCT_LEGS = its two inputs are product legs of the previous stage of the same register
// group (only the wide class makes use of it: gs_bfly_wide).
template <bool BOTH_PRODUCT_LEGS, typename TW>
__device__ __forceinline__ void gs_stage(uint64_t &x, uint64_t &y, const TW &w, const PrimeK &k) { gs_bfly(x, y, w, k); }
template <bool BOTH_PRODUCT_LEGS>
__device__ __forceinline__ void gs_stage(uint64_t &x, uint64_t &y, const TwW &w, const PrimeK &k) { gs_bfly_wide<!BOTH_PRODUCT_LEGS>(x, y, w, k); }

// Inverse: stages run from bit BLO up to BHI; tw[] is laid out as load_tw fills it (BHI first).  NTH = how many stages of
// the group ran before this one (0: inputs come from outside the group, any leg).  CANON_IN = the group holds a transform's
// FIRST stages and its inputs are canonical (< p): the first stage leaves everything < 2p, the second < 4p, so for the wide
// class neither needs a conditional subtraction at all.
template <int EL, int BHI, int B, int NTH, bool CANON_IN, int NTW, typename TW>
__device__ __forceinline__ void gs_bits(uint64_t (&x)[1 << EL], const TW (&tw)[NTW], const PrimeK &k) {
#pragma unroll
  for (int e = 0; e < (1 << EL); ++e)
    if (!(e & (1 << B))) {
      const TW &w = tw[tw_off(EL, BHI, B) + (e >> (B + 1))];
      // e is a constant after unrolling: the pair (e, e + 2^B) shares bit B-1, set = both came out of a multiplication
      if ((CANON_IN && NTH < 2) || (NTH > 0 && B > 0 && ((e >> (B > 0 ? B - 1 : 0)) & 1))) gs_stage<true>(x[e], x[e + (1 << B)], w, k);
      else gs_stage<false>(x[e], x[e + (1 << B)], w, k);
    }
  if constexpr (B < BHI) gs_bits<EL, BHI, B + 1, NTH + 1, CANON_IN>(x, tw, k);
}
template <int EL, int BHI, int BLO, bool CANON_IN = false, typename TW>
__device__ __forceinline__ void gs_group(uint64_t (&x)[1 << EL], const TW (&tw)[tw_count(EL, BHI, BLO)],
                                         const PrimeK &k) {
  gs_bits<EL, BHI, BLO, 0, CANON_IN>(x, tw, k);
}

// Last inverse stage (len = n/2, twiddle winv[1]) with the n^-1 scaling of
// src/ntt.c:71-72 folded in, canonical outputs.  in: x,y < 4p.
// The two constants are taken by value (LastK is copied out of the LimbTab at kernel entry, next to
// PrimeK): read through the table reference inside the unrolled tail they were re-fetched with
// vector loads for every pair.
template <typename TW> struct LastK;
template <> struct LastK<uint64_t> {
  uint64_t ninv, winv1_ninv;
  __device__ __forceinline__ explicit LastK(const LimbTab &t) : ninv(t.ninv), winv1_ninv(t.winv1_ninv) {}
};
template <> struct LastK<TwS> {
  TwS ninv, winv1_ninv;
  __device__ __forceinline__ explicit LastK(const LimbTab &t) : ninv(t.ninv_s), winv1_ninv(t.winv1_ninv_s) {}
};
__device__ __forceinline__ void gs_last(uint64_t &x, uint64_t &y, const LastK<uint64_t> &t, const PrimeK &k) {
  const uint64_t s = x + y;                 // < 8p
  const uint64_t d = x + k.p4 - y;          // (0, 8p)
  x = canon4(mulmod_lazy(s, t.ninv, k), k);
  y = canon4(mulmod_lazy(d, t.winv1_ninv, k), k);
}
__device__ __forceinline__ void gs_last(uint64_t &x, uint64_t &y, const LastK<TwS> &t, const PrimeK &k) {   // in: x,y < 3p
  const uint64_t s = x + y;                 // < 6p
  const uint64_t d = x + k.p3 - y;          // (0, 6p)
  x = canon4(mulmod_split(s, t.ninv, k) + k.c1, k);            // products < 3p
  y = canon4(mulmod_split(d, t.winv1_ninv, k) + k.c1, k);
}

// Per twiddle type: the table pointers of a launch and the lazy ranges the butterflies keep.
template <typename TW> struct TwTraits;
template <> struct TwTraits<uint64_t> {
  __device__ static __forceinline__ const uint64_t *table(const PassArgs &a, bool inv) { return inv ? a.winv : a.w; }
  __device__ static __forceinline__ uint64_t canon_fwd(uint64_t x, const PrimeK &k) { return canon8(x, k); }   // forward data < 8p
  __device__ static __forceinline__ uint64_t canon_inv(uint64_t x, const PrimeK &k) { return canon4(x, k); }   // inverse data < 4p
  // into the inverse range from a value < 4p / < 8p
  __device__ static __forceinline__ uint64_t inv_from4(uint64_t x, const PrimeK &) { return x; }
  __device__ static __forceinline__ uint64_t inv_from8(uint64_t x, const PrimeK &k) { return csub4(x, k); }
  // a forward-range value as the left (< 2p) / right (< 4p) operand of a variable*variable mulmod_lazy
  __device__ static __forceinline__ uint64_t left(uint64_t x, const PrimeK &k) { return csub2(csub4(x, k), k); }
  __device__ static __forceinline__ uint64_t right(uint64_t x, const PrimeK &k) { return csub4(x, k); }
};
template <> struct TwTraits<TwS> {
  __device__ static __forceinline__ const TwS *table(const PassArgs &a, bool inv) { return inv ? a.winvs : a.ws; }
  __device__ static __forceinline__ uint64_t canon_fwd(uint64_t x, const PrimeK &k) { return canon_fold(x, k.p, k.c); }   // forward data < 6p
  __device__ static __forceinline__ uint64_t canon_inv(uint64_t x, const PrimeK &k) { return canon4(x, k); }   // inverse data < 3p
  __device__ static __forceinline__ uint64_t inv_from4(uint64_t x, const PrimeK &k) { return csub2(x, k); }
  __device__ static __forceinline__ uint64_t inv_from8(uint64_t x, const PrimeK &k) { return csub2(csub4(x, k), k); }
  // products (mulmod_lazy: th of the 7-mad fold must fit 32 bits, c * (a b >> 59) < 2^91): left < 2p, right < 6p as it comes: 12 p^2, th < 12 c < 2^32
  // for c < GPQ_SPLIT_CMAX, and the product leaves below 4p (12 c^2 < 1.95 * 2^59)
  __device__ static __forceinline__ uint64_t left(uint64_t x, const PrimeK &k) { return csub2(csub4(x, k), k); }
  __device__ static __forceinline__ uint64_t right(uint64_t x, const PrimeK &) { return x; }
};

template <> struct LastK<TwW> : LastK<TwS> {      // the same two constants as the split class
  __device__ __forceinline__ explicit LastK(const LimbTab &t) : LastK<TwS>(t) {}
};
__device__ __forceinline__ void gs_last(uint64_t &x, uint64_t &y, const LastK<TwW> &t, const PrimeK &k) {   // in: x,y < 4p
  const uint64_t s = x + y;                 // < 8p: a legal multiplicand for the wide class (gs_bfly_wide)
  const uint64_t d = x + k.p4 - y;          // (0, 8p)
  x = csub1(mulmod_split(s, t.ninv, k) + k.c1, k);
  y = csub1(mulmod_split(d, t.winv1_ninv, k) + k.c1, k);
}
// Wide-split limbs: forward data < 6p after a finished transform (< 8p inside one), inverse data < 4p (gs_bfly_wide).
template <> struct TwTraits<TwW> {
  __device__ static __forceinline__ const TwW *table(const PassArgs &a, bool inv) { return reinterpret_cast<const TwW *>(inv ? a.winvs : a.ws); }
  __device__ static __forceinline__ uint64_t canon_fwd(uint64_t x, const PrimeK &k) { return canon_fold(x, k.p, k.c); }
  __device__ static __forceinline__ uint64_t canon_inv(uint64_t x, const PrimeK &k) { return canon4(x, k); }   // inverse data < 4p
  __device__ static __forceinline__ uint64_t inv_from4(uint64_t x, const PrimeK &) { return x; }                  // a product (< 4p) enters the inverse stages as it is
  __device__ static __forceinline__ uint64_t inv_from8(uint64_t x, const PrimeK &k) { return csub4(x, k); }
  // products (mulmod_lazy wants a*b < 2^122.8): left < 4p, right < 6p as it comes: 24 p^2 = 2^122.6
  __device__ static __forceinline__ uint64_t left(uint64_t x, const PrimeK &k) { return csub4(x, k); }
  __device__ static __forceinline__ uint64_t right(uint64_t x, const PrimeK &) { return x; }
};

// Zero watch of the canonical forward stores (PassArgs::zflag): m = min over the lane's outputs of (lo | hi); m == 0 <=> one of
// them is the residue 0.  2 VALU instructions per coefficient; the atomic runs only in the (rare) lanes that saw one.
__device__ __forceinline__ void zero_watch(uint32_t &m, uint64_t x) {
  const uint32_t t = (uint32_t)x | (uint32_t)(x >> 32);
  m = t < m ? t : m;
}
__device__ __forceinline__ void zero_note(const PassArgs &a, uint32_t m, unsigned poly, unsigned limb_in_launch) {
  if (m == 0 && a.zflag) atomicOr(a.zflag + (size_t)poly * a.zstride + limb_in_launch, 1u);
}

// ---------------------------------------------------------------------------
// Strided pass: the M1 = logn-8 stages with len >= 256.
// Tile = 2^M1 rows x 16 columns, T = 2^(M1+4-EL) threads, tid = col + 16*q.
//   group A: rows q + 2^(M1-EL)*e      (register bits = top EL row bits)
//   group B: rows (q << EL) + e        (register bits = low EL row bits, the
//                                       low M1-EL of them still to be done)
// Forward runs A then B, inverse B then A.  Group A's twiddles are the same
// for the whole grid (scalar loads); group B's are fetched at kernel entry.
// ---------------------------------------------------------------------------
#ifndef GPQ_STRIDED_CB
#define GPQ_STRIDED_CB 4      /* log2 of the columns of a tile: 16 columns = 128-byte row segments */
#endif
template <int M1, int EL>
struct StridedGeom {
  static constexpr int CB = M1 <= 8 ? GPQ_STRIDED_CB : 4; // (the 512-row tiles of n = 2^17 fill the LDS with 16 columns already)
  static constexpr int C = 1 << CB;
  static constexpr int E = 1 << EL;
  static constexpr int T = 1 << (M1 + CB - EL);
  static constexpr int S2 = M1 - EL;                      // stages left for group B
  static constexpr int BB = S2 > 0 ? S2 - 1 : 0;          // highest register bit group B works on
  static constexpr int LDS_ELEMS = (1 << (M1 + CB)) + ((S2 > 0) ? (C << S2) : 0);
  __device__ static __forceinline__ unsigned pad(unsigned l) { return l + ((l >> (EL + CB)) << CB); }
};

// CW = log2 of the row length the low kernels own: 8, or 9 for n = 2^17 (256-row tiles over 512-coefficient rows).
template <int M1, int EL, bool INV, bool CANON_OUT, typename TW, int CW = 8, bool NT = false>
__global__ __launch_bounds__((StridedGeom<M1, EL>::T)) void strided_pass(PassArgs a) {
  using G = StridedGeom<M1, EL>;
  constexpr int E = G::E;
  constexpr unsigned logn = M1 + CW;
  __shared__ uint64_t lds[G::S2 > 0 ? G::LDS_ELEMS : 1];

  // Workgroups go to the 8 XCDs round-robin by linear id.  The forward pass gives each XCD a
  // contiguous run of tiles instead, so the 16 column tiles that share the same 2-KB rows run on one
  // XCD together: -10 % on the forward pass (profiles/r01), nothing on the in-place inverse pass.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (!INV) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), tot = gridDim.x * gridDim.y * gridDim.z;
    const unsigned logical = (lin & 7) * (tot >> 3) + (lin >> 3);     // gridDim.x = 8 or 16: tot is a multiple of 8
    bx = logical % gridDim.x; by = (logical / gridDim.x) % gridDim.y; bz = logical / (gridDim.x * gridDim.y);
  }
  const unsigned limb = a.limb0 + bz;
  const LimbTab &tab = a.tabs[limb];
  const PrimeK k = tab.k;
  const LastK<TW> last(tab);
  const TW *__restrict__ wt = TwTraits<TW>::table(a, INV) + ((size_t)limb << logn);
  const unsigned slab = by % a.nslab, poly = by / a.nslab;
  const size_t off = (size_t)poly * a.poly_stride + ((size_t)bz << logn);
  const uint64_t *__restrict__ src = a.src[slab] + off;
  uint64_t *__restrict__ dst = a.dst[slab] + off;

  const unsigned tid = threadIdx.x;
  constexpr int CB = G::CB;
  const unsigned col = (bx << CB) + (tid & (G::C - 1));
  const unsigned q = tid >> CB;
  // group A: element e at row q + (e << S2); group B: row (q << EL) + e
  const unsigned iA = (q << CW) + col, iB = (q << (EL + CW)) + col;
  constexpr unsigned strideA = 1u << (G::S2 + CW), strideB = 1u << CW;
  uint64_t x[E];
  TW twB[tw_count(EL, G::BB, 0)];

  if (!INV) {
    TW twA[tw_count(EL, EL - 1, 0)];
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = slab_ld<NT>(&src[iA + e * strideA]);
    load_tw<EL, EL - 1, 0, true>(twA, iA, G::S2 + CW, logn, wt);
    if (G::S2 > 0) load_tw<EL, G::BB, 0, false>(twB, iB, CW, logn, wt);
    ct_group<EL, EL - 1, 0, G::S2 + CW>(x, twA, k);
    if (G::S2 > 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) lds[G::pad(tid + e * G::T)] = x[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = lds[G::pad((q << (EL + CB)) + (e << CB) + (tid & (G::C - 1)))];
      ct_group<EL, G::BB, 0, CW>(x, twB, k);
#pragma unroll
      for (int e = 0; e < E; ++e) slab_st<NT>(&dst[iB + e * strideB], CANON_OUT ? TwTraits<TW>::canon_fwd(x[e], k) : x[e]);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) slab_st<NT>(&dst[iA + e * strideA], CANON_OUT ? TwTraits<TW>::canon_fwd(x[e], k) : x[e]);
    }
  } else {
    // top EL row bits; the very last one (len = n/2) carries the n^-1 scaling and is done by gs_last
    TW twA[tw_count(EL, (EL > 1 ? EL - 2 : 0), 0)];
    if (G::S2 > 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = slab_ld<NT>(&src[iB + e * strideB]);
      load_tw<EL, G::BB, 0, false>(twB, iB, CW, logn, wt);
      if (EL > 1) load_tw<EL, (EL > 1 ? EL - 2 : 0), 0, true>(twA, iA, G::S2 + CW, logn, wt);
      gs_group<EL, G::BB, 0>(x, twB, k);
#pragma unroll
      for (int e = 0; e < E; ++e) lds[G::pad((q << (EL + CB)) + (e << CB) + (tid & (G::C - 1)))] = x[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = lds[G::pad(tid + e * G::T)];
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = slab_ld<NT>(&src[iA + e * strideA]);
      if (EL > 1) load_tw<EL, (EL > 1 ? EL - 2 : 0), 0, true>(twA, iA, G::S2 + CW, logn, wt);
    }
    if (EL > 1) gs_group<EL, (EL > 1 ? EL - 2 : 0), 0>(x, twA, k);
#pragma unroll
    for (int e = 0; e < E / 2; ++e) gs_last(x[e], x[e + E / 2], last, k);
#pragma unroll
    for (int e = 0; e < E; ++e) slab_st<NT>(&dst[iA + e * strideA], x[e]);
  }
}

// ---------------------------------------------------------------------------
// Contiguous pass: the 8 stages with len <= 128, inside 256-element blocks.
// A wave owns 4 consecutive blocks (1024 coefficients); lane = j + 16*blk.
//   group H: k = j + 16*e   (register bits = k bits 7..4)
//   group L: k = 16*j + e   (register bits = k bits 3..0)
// The H<->L exchange stays inside the wave's own LDS region.
// ---------------------------------------------------------------------------
constexpr int CONTIG_WAVES = 4;                       // waves per workgroup
constexpr int CONTIG_LDS_PER_WAVE = 1024 + 64;        // padded: l + (l >> 4)

// LDS traffic of one wave is ordered by the hardware; this only stops the
// compiler from moving the wave's LDS reads across its LDS writes.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct ContigLane {
  unsigned j, blk, hbase, lbase;                      // hbase/lbase: index of x[0] inside the wave's 1024 coefficients
  uint64_t *lds;
  __device__ __forceinline__ ContigLane(uint64_t *wave_lds) {
    const unsigned lane = threadIdx.x & 63;
    j = lane & 15; blk = lane >> 4;
    hbase = (blk << 8) + j;
    lbase = (blk << 8) + (j << 4);
    lds = wave_lds;
  }
  __device__ static __forceinline__ unsigned pad(unsigned l) { return l + (l >> 4); }
  // registers in H layout -> registers in L layout
  __device__ __forceinline__ void h_to_l(uint64_t (&x)[16]) const {
#pragma unroll
    for (int e = 0; e < 16; ++e) lds[pad(hbase + 16 * e)] = x[e];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = lds[pad(lbase + e)];
    wave_lds_sync();
  }
  __device__ __forceinline__ void l_to_h(uint64_t (&x)[16]) const {
#pragma unroll
    for (int e = 0; e < 16; ++e) lds[pad(lbase + e)] = x[e];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = lds[pad(hbase + 16 * e)];
    wave_lds_sync();
  }
};

// The 30 twiddles one lane needs for the 8 low stages of one direction: 15 for the H group, 15 for
// the L group.  Kernels that hold several polynomials run one group on all of them before fetching
// the other group's twiddles, so only 15 entries are live at a time (split pairs are 4 VGPRs each).
template <typename TW>
struct ContigTw {
  TW t[15];
  __device__ __forceinline__ void load_h(const ContigLane &ln, unsigned wave0, unsigned logn, const TW *__restrict__ w) {
    load_tw<4, 3, 0, false>(t, wave0 + ln.hbase, 4, logn, w);
  }
  __device__ __forceinline__ void load_l(const ContigLane &ln, unsigned wave0, unsigned logn, const TW *__restrict__ w) {
    load_tw<4, 3, 0, false>(t, wave0 + ln.lbase, 0, logn, w);
  }
};

template <bool NT> __device__ __forceinline__ void load_h(uint64_t (&x)[16], const uint64_t *__restrict__ p, const ContigLane &ln) {
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = slab_ld<NT>(&p[ln.hbase + 16 * e]);
}
template <bool NT> __device__ __forceinline__ void store_h(uint64_t *__restrict__ p, const uint64_t (&x)[16], const ContigLane &ln) {
#pragma unroll
  for (int e = 0; e < 16; ++e) slab_st<NT>(&p[ln.hbase + 16 * e], x[e]);
}
__device__ __forceinline__ void load_l(uint64_t (&x)[16], const uint64_t *__restrict__ p, const ContigLane &ln) {
  const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(p + ln.lbase);
#pragma unroll
  for (int e = 0; e < 8; ++e) { ulonglong2 t = v[e]; x[2 * e] = t.x; x[2 * e + 1] = t.y; }
}
struct ContigBlock {       // per-workgroup addressing shared by the contiguous kernels
  unsigned wave0;          // limb-relative index of the wave's first coefficient
  unsigned limb;           // prime index
  size_t off;              // offset of the limb inside a slab + wave0
  size_t toff;             // offset of the limb inside the twiddle tables
  unsigned slab, poly;
  __device__ __forceinline__ ContigBlock(const PassArgs &a) {
    const unsigned wave = threadIdx.x >> 6;
    wave0 = (blockIdx.x * CONTIG_WAVES + wave) << 10;
    limb = a.limb0 + blockIdx.z;
    slab = blockIdx.y % a.nslab; poly = blockIdx.y / a.nslab;
    off = (size_t)poly * a.poly_stride + ((size_t)blockIdx.z << a.logn) + wave0;
    toff = (size_t)limb << a.logn;
  }
};

// Forward: low 8 stages, canonical output (end of gpq_ntt).  Inverse: low 8
// stages of the inverse transform (feeds the strided inverse pass).
// A pass reads as many twiddle bytes as data bytes (every table entry is used
// exactly once per limb), so one workgroup walks CONTIG_POLYS polynomials of
// the same limb and tile with the 30 twiddles of a lane held in registers, and
// fetches the next polynomial's coefficients while it works on the current one.
#ifndef GPQ_CONTIG_POLYS
#define GPQ_CONTIG_POLYS 2   /* 2 polynomials per workgroup at 3 waves per SIMD beat 4 at 2 (split twiddles): -13 % on both passes */
#endif
constexpr int CONTIG_POLYS = GPQ_CONTIG_POLYS;

#ifndef GPQ_CONTIG_MINWAVES
#define GPQ_CONTIG_MINWAVES 3
#endif
// (the plain-twiddle form keeps its 30 twiddles and a prefetched polynomial in registers: 2 waves per SIMD)
template <bool INV, typename TW, bool NT = false>
__global__ __launch_bounds__(CONTIG_WAVES * 64, (sizeof(TW) == 8 ? 2 : GPQ_CONTIG_MINWAVES)) void contig_pass(PassArgs a, unsigned polys) {
  using TT = TwTraits<TW>;
  __shared__ uint64_t lds[CONTIG_WAVES * CONTIG_LDS_PER_WAVE];
  ContigLane ln(lds + (threadIdx.x >> 6) * CONTIG_LDS_PER_WAVE);
  const unsigned wave0 = (blockIdx.x * CONTIG_WAVES + (threadIdx.x >> 6)) << 10;
  const unsigned limb = a.limb0 + blockIdx.z;
  const LimbTab &tab = a.tabs[limb];
  const PrimeK k = tab.k;
  const unsigned p0 = blockIdx.y * CONTIG_POLYS;
  const unsigned cnt = polys - p0 < CONTIG_POLYS ? polys - p0 : CONTIG_POLYS;
  const size_t off = (size_t)p0 * a.poly_stride + ((size_t)blockIdx.z << a.logn) + wave0;
  const uint64_t *__restrict__ src = a.src[0] + off;
  uint64_t *__restrict__ dst = a.dst[0] + off;
  const TW *__restrict__ wt = TT::table(a, INV) + ((size_t)limb << a.logn);
  if constexpr (sizeof(TW) == 8) {
    // plain twiddles: all 30 of a lane stay in registers and the next polynomial's coefficients
    // are fetched under the current one's butterflies
    uint64_t x[16], nx[16];
    ContigTw<TW> twh, twl;
    // Global accesses use the H layout only (16 lanes = 128 contiguous bytes); the L layout
    // (each lane on its own 128-byte line) is reached through one more LDS exchange instead.
    load_h<NT>(x, src, ln);
    twh.load_h(ln, wave0, a.logn, wt);
    twl.load_l(ln, wave0, a.logn, wt);
    for (unsigned i = 0; i < cnt; ++i) {
      if (i + 1 < cnt) load_h<NT>(nx, src + (size_t)(i + 1) * a.poly_stride, ln);
      if (!INV) {
        ct_group<4, 3, 0, 4>(x, twh.t, k);
        ln.h_to_l(x);
        ct_group<4, 3, 0, 0>(x, twl.t, k);
        ln.l_to_h(x);
        uint32_t zm = ~0u;
#pragma unroll
        for (int e = 0; e < 16; ++e) { x[e] = TT::canon_fwd(x[e], k); zero_watch(zm, x[e]); }
        zero_note(a, zm, p0 + i, blockIdx.z);
      } else {
        ln.h_to_l(x);
        gs_group<4, 3, 0, true>(x, twl.t, k);      // gpq_invntt's first stages: canonical inputs
        ln.l_to_h(x);
        gs_group<4, 3, 0>(x, twh.t, k);
      }
      store_h<NT>(dst + (size_t)i * a.poly_stride, x, ln);
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] = nx[e];
    }
  } else {
    // split pairs are 4 VGPRs each: the CONTIG_POLYS polynomials are held together and one twiddle
    // group (15 pairs) at a time runs over all of them, as in tensor_mid8
    uint64_t x[CONTIG_POLYS][16];
    ContigTw<TW> tw;
#pragma unroll
    for (int j = 0; j < CONTIG_POLYS; ++j)
      if (j < (int)cnt) load_h<NT>(x[j], src + (size_t)j * a.poly_stride, ln);
    if (!INV) {
      tw.load_h(ln, wave0, a.logn, wt);
#pragma unroll
      for (int j = 0; j < CONTIG_POLYS; ++j) if (j < (int)cnt) ct_group<4, 3, 0, 4>(x[j], tw.t, k);
      tw.load_l(ln, wave0, a.logn, wt);
#pragma unroll
      for (int j = 0; j < CONTIG_POLYS; ++j)
        if (j < (int)cnt) {
          ln.h_to_l(x[j]);
          ct_group<4, 3, 0, 0>(x[j], tw.t, k);
          ln.l_to_h(x[j]);
          uint32_t zm = ~0u;
#pragma unroll
          for (int e = 0; e < 16; ++e) { x[j][e] = TT::canon_fwd(x[j][e], k); zero_watch(zm, x[j][e]); }
          zero_note(a, zm, p0 + j, blockIdx.z);
          store_h<NT>(dst + (size_t)j * a.poly_stride, x[j], ln);
        }
    } else {
      tw.load_l(ln, wave0, a.logn, wt);
#pragma unroll
      for (int j = 0; j < CONTIG_POLYS; ++j)
        if (j < (int)cnt) {
          ln.h_to_l(x[j]);
          gs_group<4, 3, 0, true>(x[j], tw.t, k);      // gpq_invntt's first stages: canonical inputs
          ln.l_to_h(x[j]);
        }
      tw.load_h(ln, wave0, a.logn, wt);
#pragma unroll
      for (int j = 0; j < CONTIG_POLYS; ++j)
        if (j < (int)cnt) {
          gs_group<4, 3, 0>(x[j], tw.t, k);
          store_h<NT>(dst + (size_t)j * a.poly_stride, x[j], ln);
        }
    }
  }
}

// ---------------------------------------------------------------------------
// Fused middle of the he_mul tensor stage (src/he-mult.c:121-136), per limb:
//   low 8 forward stages of ct1.c0, ct1.c1, ct2.c0, ct2.c1
//   d0 = c0*c0', d2 = c1*c1', d1 = c0*c1' + c1*c0'   (the add moved in front
//   of the inverse transform: INTT is linear, canonical results are identical)
//   low 8 inverse stages of d0, d1, d2
// src[0..3] = a0,a1,b0,b1 after the strided forward pass; dst[0..2] = d0,d1,d2.  Kernel: tensor_mid8 below.
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Fused middle of the key-switch inner product (src/he-mult.c:60-64 ==
// src/he-automorphism.c:61-65): src[0] = d2 after the strided forward pass,
// evk0/evk1 = NTT-domain key limbs (shared by the whole batch, no poly
// stride), dst[0..1] = low inverse stages of d2*evk.p0 and d2*evk.p1.
// ---------------------------------------------------------------------------
struct KeyswitchArgs { PassArgs p; const uint64_t *evk0; const uint64_t *evk1; };

// ---------------------------------------------------------------------------
// The low stages with 8 coefficients per lane instead of 16.
// The 16-per-lane forms hold 4 (3) polynomials x 16 coefficients x 2 VGPRs and run at 2 (3) waves per
// SIMD, where the dependent instruction chain of a butterfly is not covered (profiles/r01: 20 % more cycles
// per VALU instruction with arithmetic only than the same mix at 8 waves).  Here a wave owns
// 512 consecutive coefficients and the LOW = 8 or 9 low stages run as three register groups with two
// exchanges through the wave's own LDS region.  LOW = 8 (blocks of 256, two per wave; lane = j + 32 blk):
//   H: k = j + 32 e                      index bits 7..5     (3 stages)
//   M: k = 32 (j >> 2) + 4 e + (j & 3)   bits 4..2           (3 stages)
//   L: k = 8 j + e                       bits 1..0           (2 stages; bit 2 rides along)
// LOW = 9 (one block of 512 per wave; j = lane): H: k = j + 64 e (bits 8..6), M: k = 64 (j >> 3) + 8 e + (j & 7)
// (bits 5..3), L: k = 8 j + e (bits 2..0): three full groups.  With LOW = 9 the strided passes of n = 2^17 keep
// the 256-row tiles of n = 2^16 (512-coefficient rows) instead of 512-row tiles at 2 waves per SIMD.
// Global accesses use the H layout (32 or 64 lanes = 256 or 512 contiguous bytes).
// ---------------------------------------------------------------------------
constexpr int LANE8_LDS_PER_WAVE = 576;     // 512 + padding (both paddings end below 576)

template <int LOW>
struct Lane8 {
  static constexpr int JB = LOW - 3;        // log2 lanes per block = log2 of the H stride
  static constexpr int MB = LOW - 6;        // log2 of the M stride
  static constexpr int LHI = LOW - 7;       // highest register bit the L group works on
  unsigned hk, mk, lk;      // index of x[0] inside the wave's 512 coefficients, per layout
  uint64_t *lds;
  __device__ __forceinline__ Lane8(uint64_t *wave_lds) {
    const unsigned lane = threadIdx.x & 63, j = lane & ((1u << JB) - 1), blk = lane >> JB;
    hk = (blk << LOW) + j;
    mk = (blk << LOW) + ((j >> MB) << JB) + (j & ((1u << MB) - 1));
    lk = (blk << LOW) + (j << 3);
    lds = wave_lds;
  }
  // H<->M exchange: rows of 2^JB words padded by 2^MB (M reads walk 2^MB e + (j mod 2^MB) inside row j >> MB);
  // M<->L exchange: one word per 8 (L reads are 9 j + e).  Both conflict-free per half-wave (0 measured).
  __device__ static __forceinline__ unsigned pad1(unsigned l) { return l + ((l >> JB) << MB); }
  __device__ static __forceinline__ unsigned pad2(unsigned l) { return l + (l >> 3); }
  __device__ __forceinline__ void h_to_m(uint64_t (&x)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[pad1(hk + (e << JB))] = x[e];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds[pad1(mk + (e << MB))];
    wave_lds_sync();
  }
  __device__ __forceinline__ void m_to_l(uint64_t (&x)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[pad2(mk + (e << MB))] = x[e];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds[pad2(lk + e)];
    wave_lds_sync();
  }
  __device__ __forceinline__ void l_to_m(uint64_t (&x)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[pad2(lk + e)] = x[e];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds[pad2(mk + (e << MB))];
    wave_lds_sync();
  }
  __device__ __forceinline__ void m_to_h(uint64_t (&x)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[pad1(mk + (e << MB))] = x[e];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds[pad1(hk + (e << JB))];
    wave_lds_sync();
  }
  // direct L<->H for the standalone passes (global accesses stay in the H layout): padding 2, where a few lanes of the
  // H side share a bank pair with another one
  __device__ __forceinline__ void l_to_h(uint64_t (&x)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[pad2(lk + e)] = x[e];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds[pad2(hk + (e << JB))];
    wave_lds_sync();
  }
  __device__ __forceinline__ void h_to_l(uint64_t (&x)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[pad2(hk + (e << JB))] = x[e];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds[pad2(lk + e)];
    wave_lds_sync();
  }
  __device__ __forceinline__ void load_l(uint64_t (&x)[8], const uint64_t *__restrict__ p) const {
    const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(p + lk);
#pragma unroll
    for (int e = 0; e < 4; ++e) { ulonglong2 t = v[e]; x[2 * e] = t.x; x[2 * e + 1] = t.y; }
  }
  // the stages of the three groups (forward: H, M, L; inverse: L, M, H)
  template <typename TW> __device__ static __forceinline__ void ct_h(uint64_t (&x)[8], const TW (&t)[7], const PrimeK &k) { ct_group<3, 2, 0, LOW - 3>(x, t, k); }
  template <typename TW> __device__ static __forceinline__ void ct_m(uint64_t (&x)[8], const TW (&t)[7], const PrimeK &k) { ct_group<3, 2, 0, LOW - 6>(x, t, k); }
  template <typename TW> __device__ static __forceinline__ void ct_l(uint64_t (&x)[8], const TW (&u)[tw_count(3, LHI, 0)], const PrimeK &k) { ct_group<3, LHI, 0, 0>(x, u, k); }
  template <typename TW> __device__ static __forceinline__ void gs_hm(uint64_t (&x)[8], const TW (&t)[7], const PrimeK &k) { gs_group<3, 2, 0>(x, t, k); }
  template <typename TW> __device__ static __forceinline__ void gs_l(uint64_t (&x)[8], const TW (&u)[tw_count(3, LHI, 0)], const PrimeK &k) { gs_group<3, LHI, 0>(x, u, k); }
  // ... the same group when it opens a standalone inverse transform (canonical inputs)
  template <typename TW> __device__ static __forceinline__ void gs_l_canon(uint64_t (&x)[8], const TW (&u)[tw_count(3, LHI, 0)], const PrimeK &k) { gs_group<3, LHI, 0, true>(x, u, k); }
};

// ... with the slab accesses (H layout) under the launch's cache policy
template <int LOW, bool NT>
struct Lane8N : Lane8<LOW> {
  using Lane8<LOW>::hk;
  using Lane8<LOW>::JB;
  __device__ __forceinline__ Lane8N(uint64_t *wave_lds) : Lane8<LOW>(wave_lds) {}
  __device__ __forceinline__ void load_h(uint64_t (&x)[8], const uint64_t *__restrict__ p) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = slab_ld<NT>(&p[hk + (e << JB)]);
  }
  __device__ __forceinline__ void store_h(uint64_t *__restrict__ p, const uint64_t (&x)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) slab_st<NT>(&p[hk + (e << JB)], x[e]);
  }
};

// twiddles of one group of one direction: 7 for H and M, 6 or 7 for L
template <typename TW, int LOW>
struct Tw8 {
  using L8 = Lane8<LOW>;
  TW t[7];
  TW u[tw_count(3, L8::LHI, 0)];
  __device__ __forceinline__ void load_h(const L8 &ln, unsigned wave0, unsigned logn, const TW *__restrict__ w) {
    load_tw<3, 2, 0, false>(t, wave0 + ln.hk, LOW - 3, logn, w);
  }
  __device__ __forceinline__ void load_m(const L8 &ln, unsigned wave0, unsigned logn, const TW *__restrict__ w) {
    load_tw<3, 2, 0, false>(t, wave0 + ln.mk, LOW - 6, logn, w);
  }
  __device__ __forceinline__ void load_l(const L8 &ln, unsigned wave0, unsigned logn, const TW *__restrict__ w) {
    load_tw<3, L8::LHI, 0, false>(u, wave0 + ln.lk, 0, logn, w);
  }
};

struct Block8 {             // per-workgroup addressing: 4 waves x 512 coefficients
  unsigned wave0, limb;
  size_t off, toff;
  __device__ __forceinline__ Block8(const PassArgs &a) {
    wave0 = (blockIdx.x * CONTIG_WAVES + (threadIdx.x >> 6)) << 9;
    limb = a.limb0 + blockIdx.z;
    off = (size_t)blockIdx.y * a.poly_stride + ((size_t)blockIdx.z << a.logn) + wave0;
    toff = (size_t)limb << a.logn;
  }
};

#ifndef GPQ_MID8_MINWAVES
#define GPQ_MID8_MINWAVES 3
#endif
template <typename TW, int LOW, bool NT = false>
__global__ __launch_bounds__(CONTIG_WAVES * 64, GPQ_MID8_MINWAVES) void tensor_mid8(PassArgs a) {
  using TT = TwTraits<TW>;
  using L8 = Lane8N<LOW, NT>;
  __shared__ uint64_t lds[CONTIG_WAVES * LANE8_LDS_PER_WAVE];
  L8 ln(lds + (threadIdx.x >> 6) * LANE8_LDS_PER_WAVE);
  const Block8 cb(a);
  const PrimeK k = a.tabs[cb.limb].k;
  const TW *__restrict__ wf = TT::table(a, false) + cb.toff, *__restrict__ wi = TT::table(a, true) + cb.toff;
  uint64_t a0[8], a1[8], b0[8], b1[8];
  Tw8<TW, LOW> tw;
  ln.load_h(a0, a.src[0] + cb.off);
  tw.load_h(ln, cb.wave0, a.logn, wf);
  ln.load_h(b0, a.src[2] + cb.off);
  ln.load_h(a1, a.src[1] + cb.off);
  ln.load_h(b1, a.src[3] + cb.off);
  L8::ct_h(a0, tw.t, k);
  L8::ct_h(b0, tw.t, k);
  L8::ct_h(a1, tw.t, k);
  L8::ct_h(b1, tw.t, k);
  tw.load_m(ln, cb.wave0, a.logn, wf);
  ln.h_to_m(a0); L8::ct_m(a0, tw.t, k);
  ln.h_to_m(b0); L8::ct_m(b0, tw.t, k);
  ln.h_to_m(a1); L8::ct_m(a1, tw.t, k);
  ln.h_to_m(b1); L8::ct_m(b1, tw.t, k);
  tw.load_l(ln, cb.wave0, a.logn, wf);
  ln.m_to_l(a0); L8::ct_l(a0, tw.u, k);
  ln.m_to_l(b0); L8::ct_l(b0, tw.u, k);
  ln.m_to_l(a1); L8::ct_l(a1, tw.u, k);
  ln.m_to_l(b1); L8::ct_l(b1, tw.u, k);
  tw.load_l(ln, cb.wave0, a.logn, wi);               // inverse twiddles arrive under the products
  uint64_t d1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {                      // products: TwTraits::left / right put the operands where mulmod_lazy's second fold fits 32 bits and the product leaves
                                                     // below 4p -- 7-mad class: left < 2p, right < 4p (8 p^2); split: left < 2p, right < 6p (12 p^2, 12 c^2 < 2^60); wide: left < 4p, right < 6p
    const uint64_t u0 = TT::left(a0[e], k), u1 = TT::left(a1[e], k);
    const uint64_t v0 = TT::right(b0[e], k), v1 = TT::right(b1[e], k);
    a0[e] = TT::inv_from4(mulmod_lazy(u0, v0, k), k);                                   // d0
    d1[e] = TT::inv_from8(mulmod_lazy(u0, v1, k) + mulmod_lazy(u1, v0, k), k);          // d1
    a1[e] = TT::inv_from4(mulmod_lazy(u1, v1, k), k);                                   // d2
  }
  L8::gs_l(a0, tw.u, k); ln.l_to_m(a0);
  L8::gs_l(d1, tw.u, k); ln.l_to_m(d1);
  L8::gs_l(a1, tw.u, k); ln.l_to_m(a1);
  tw.load_m(ln, cb.wave0, a.logn, wi);
  L8::gs_hm(a0, tw.t, k); ln.m_to_h(a0);
  L8::gs_hm(d1, tw.t, k); ln.m_to_h(d1);
  L8::gs_hm(a1, tw.t, k); ln.m_to_h(a1);
  tw.load_h(ln, cb.wave0, a.logn, wi);
  L8::gs_hm(a0, tw.t, k);
  ln.store_h(a.dst[0] + cb.off, a0);
  L8::gs_hm(d1, tw.t, k);
  ln.store_h(a.dst[1] + cb.off, d1);
  L8::gs_hm(a1, tw.t, k);
  ln.store_h(a.dst[2] + cb.off, a1);
}

// The tensor stage of a SQUARING, he_mul(&ct, &ct, &ct, rlk) (src/he-algo.c:151 and the repeated squarings of he_exp / he_inv call
// src/he-mult.c:88-156 with both operands the same ciphertext): two forward low-halves instead of four,
//   d0 = c0 c0, d2 = c1 c1, d1 = c0 c1 + c1 c0 = 2 c0 c1   (the same residues as the general kernel gives for b = a),
// three inverse low-halves.  src[0..1] = c0, c1 after the strided forward pass; dst[0..2] = d0, d1, d2.
template <typename TW, int LOW, bool NT = false>
__global__ __launch_bounds__(CONTIG_WAVES * 64, GPQ_MID8_MINWAVES) void tensor_sq_mid8(PassArgs a) {
  using TT = TwTraits<TW>;
  using L8 = Lane8N<LOW, NT>;
  __shared__ uint64_t lds[CONTIG_WAVES * LANE8_LDS_PER_WAVE];
  L8 ln(lds + (threadIdx.x >> 6) * LANE8_LDS_PER_WAVE);
  const Block8 cb(a);
  const PrimeK k = a.tabs[cb.limb].k;
  const TW *__restrict__ wf = TT::table(a, false) + cb.toff, *__restrict__ wi = TT::table(a, true) + cb.toff;
  uint64_t a0[8], a1[8];
  Tw8<TW, LOW> tw;
  ln.load_h(a0, a.src[0] + cb.off);
  tw.load_h(ln, cb.wave0, a.logn, wf);
  ln.load_h(a1, a.src[1] + cb.off);
  L8::ct_h(a0, tw.t, k);
  L8::ct_h(a1, tw.t, k);
  tw.load_m(ln, cb.wave0, a.logn, wf);
  ln.h_to_m(a0); L8::ct_m(a0, tw.t, k);
  ln.h_to_m(a1); L8::ct_m(a1, tw.t, k);
  tw.load_l(ln, cb.wave0, a.logn, wf);
  ln.m_to_l(a0); L8::ct_l(a0, tw.u, k);
  ln.m_to_l(a1); L8::ct_l(a1, tw.u, k);
  tw.load_l(ln, cb.wave0, a.logn, wi);
  uint64_t d1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {                      // left operand < 2p or < 4p, right operand as it comes: the ranges of tensor_mid8
    const uint64_t u0 = TT::left(a0[e], k), u1 = TT::left(a1[e], k);
    const uint64_t v0 = TT::right(a0[e], k), v1 = TT::right(a1[e], k);
    const uint64_t cross = mulmod_lazy(u0, v1, k);                                      // c0 c1, in (0, 4p)
    a0[e] = TT::inv_from4(mulmod_lazy(u0, v0, k), k);                                   // d0
    d1[e] = TT::inv_from8(cross + cross, k);                                            // d1
    a1[e] = TT::inv_from4(mulmod_lazy(u1, v1, k), k);                                   // d2
  }
  L8::gs_l(a0, tw.u, k); ln.l_to_m(a0);
  L8::gs_l(d1, tw.u, k); ln.l_to_m(d1);
  L8::gs_l(a1, tw.u, k); ln.l_to_m(a1);
  tw.load_m(ln, cb.wave0, a.logn, wi);
  L8::gs_hm(a0, tw.t, k); ln.m_to_h(a0);
  L8::gs_hm(d1, tw.t, k); ln.m_to_h(d1);
  L8::gs_hm(a1, tw.t, k); ln.m_to_h(a1);
  tw.load_h(ln, cb.wave0, a.logn, wi);
  L8::gs_hm(a0, tw.t, k);
  ln.store_h(a.dst[0] + cb.off, a0);
  L8::gs_hm(d1, tw.t, k);
  ln.store_h(a.dst[1] + cb.off, d1);
  L8::gs_hm(a1, tw.t, k);
  ln.store_h(a.dst[2] + cb.off, a1);
}

// Middle of poly_mul's limb loop (src/poly.c:96-103): low forward stages of a and b, a (*) b, low inverse stages.
// src[0], src[1] = a, b after the strided forward pass; dst[0] = r before the strided inverse pass (r may be a or b: a
// workgroup reads its 2048 coefficients of both before it writes them).
template <typename TW, int LOW, bool NT = false>
__global__ __launch_bounds__(CONTIG_WAVES * 64, 4) void polymul_mid8(PassArgs a) {
  using TT = TwTraits<TW>;
  using L8 = Lane8N<LOW, NT>;
  __shared__ uint64_t lds[CONTIG_WAVES * LANE8_LDS_PER_WAVE];
  L8 ln(lds + (threadIdx.x >> 6) * LANE8_LDS_PER_WAVE);
  const Block8 cb(a);
  const PrimeK k = a.tabs[cb.limb].k;
  const TW *__restrict__ wf = TT::table(a, false) + cb.toff, *__restrict__ wi = TT::table(a, true) + cb.toff;
  uint64_t x[8], y[8];
  Tw8<TW, LOW> tw;
  ln.load_h(x, a.src[0] + cb.off);
  tw.load_h(ln, cb.wave0, a.logn, wf);
  ln.load_h(y, a.src[1] + cb.off);
  L8::ct_h(x, tw.t, k);
  L8::ct_h(y, tw.t, k);
  tw.load_m(ln, cb.wave0, a.logn, wf);
  ln.h_to_m(x); L8::ct_m(x, tw.t, k);
  ln.h_to_m(y); L8::ct_m(y, tw.t, k);
  tw.load_l(ln, cb.wave0, a.logn, wf);
  ln.m_to_l(x); L8::ct_l(x, tw.u, k);
  ln.m_to_l(y); L8::ct_l(y, tw.u, k);
  tw.load_l(ln, cb.wave0, a.logn, wi);
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = TT::inv_from4(mulmod_lazy(TT::left(x[e], k), TT::right(y[e], k), k), k);
  L8::gs_l(x, tw.u, k); ln.l_to_m(x);
  tw.load_m(ln, cb.wave0, a.logn, wi);
  L8::gs_hm(x, tw.t, k); ln.m_to_h(x);
  tw.load_h(ln, cb.wave0, a.logn, wi);
  L8::gs_hm(x, tw.t, k);
  ln.store_h(a.dst[0] + cb.off, x);
}

// Middle of he_mulpt's limb loop (src/he-mult.c:179-185): low forward stages of m, c0, c1, then m (*) c0 and m (*) c1,
// low inverse stages of both.  src[0..2] = m, c0, c1 after the strided forward pass; dst[0..1] may be c0, c1.
template <typename TW, int LOW, bool NT = false>
__global__ __launch_bounds__(CONTIG_WAVES * 64, 3) void mulpt_mid8(PassArgs a) {
  using TT = TwTraits<TW>;
  using L8 = Lane8N<LOW, NT>;
  __shared__ uint64_t lds[CONTIG_WAVES * LANE8_LDS_PER_WAVE];
  L8 ln(lds + (threadIdx.x >> 6) * LANE8_LDS_PER_WAVE);
  const Block8 cb(a);
  const PrimeK k = a.tabs[cb.limb].k;
  const TW *__restrict__ wf = TT::table(a, false) + cb.toff, *__restrict__ wi = TT::table(a, true) + cb.toff;
  uint64_t m[8], x[8], y[8];
  Tw8<TW, LOW> tw;
  ln.load_h(m, a.src[0] + cb.off);
  tw.load_h(ln, cb.wave0, a.logn, wf);
  ln.load_h(x, a.src[1] + cb.off);
  ln.load_h(y, a.src[2] + cb.off);
  L8::ct_h(m, tw.t, k);
  L8::ct_h(x, tw.t, k);
  L8::ct_h(y, tw.t, k);
  tw.load_m(ln, cb.wave0, a.logn, wf);
  ln.h_to_m(m); L8::ct_m(m, tw.t, k);
  ln.h_to_m(x); L8::ct_m(x, tw.t, k);
  ln.h_to_m(y); L8::ct_m(y, tw.t, k);
  tw.load_l(ln, cb.wave0, a.logn, wf);
  ln.m_to_l(m); L8::ct_l(m, tw.u, k);
  ln.m_to_l(x); L8::ct_l(x, tw.u, k);
  ln.m_to_l(y); L8::ct_l(y, tw.u, k);
  tw.load_l(ln, cb.wave0, a.logn, wi);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint64_t u = TT::left(m[e], k);
    x[e] = TT::inv_from4(mulmod_lazy(u, TT::right(x[e], k), k), k);
    y[e] = TT::inv_from4(mulmod_lazy(u, TT::right(y[e], k), k), k);
  }
  L8::gs_l(x, tw.u, k); ln.l_to_m(x);
  L8::gs_l(y, tw.u, k); ln.l_to_m(y);
  tw.load_m(ln, cb.wave0, a.logn, wi);
  L8::gs_hm(x, tw.t, k); ln.m_to_h(x);
  L8::gs_hm(y, tw.t, k); ln.m_to_h(y);
  tw.load_h(ln, cb.wave0, a.logn, wi);
  L8::gs_hm(x, tw.t, k);
  ln.store_h(a.dst[0] + cb.off, x);
  L8::gs_hm(y, tw.t, k);
  ln.store_h(a.dst[1] + cb.off, y);
}

// Key switch with TWO polynomials of the same limb and tile per workgroup: the key limbs (shared by the batch) and every
// twiddle group are fetched once for both -- per polynomial the 16-per-lane kernel reads 960 B of twiddle pairs and 256 B
// of key per lane against 384 B of its own data -- and the four products of a pair run like the tensor stage (2 forward
// low-halves, 4 inverse ones).  blockIdx.y = pair of polynomials starting at `first`; an odd last polynomial runs alone in the TWO = false instantiation.
template <typename TW, int LOW, bool TWO = true, bool NT = false>
__global__ __launch_bounds__(CONTIG_WAVES * 64, 3) void keyswitch_mid8x2(KeyswitchArgs ka, unsigned first) {
  using TT = TwTraits<TW>;
  using L8 = Lane8N<LOW, NT>;
  __shared__ uint64_t lds[CONTIG_WAVES * LANE8_LDS_PER_WAVE];
  const PassArgs &a = ka.p;
  L8 ln(lds + (threadIdx.x >> 6) * LANE8_LDS_PER_WAVE);
  const unsigned wave0 = (blockIdx.x * CONTIG_WAVES + (threadIdx.x >> 6)) << 9;
  const unsigned limb = a.limb0 + blockIdx.z;
  const PrimeK k = a.tabs[limb].k;
  const size_t toff = (size_t)limb << a.logn;
  const TW *__restrict__ wf = TT::table(a, false) + toff, *__restrict__ wi = TT::table(a, true) + toff;
  const unsigned p0 = first + 2 * blockIdx.y;  // TWO = false: the odd last polynomial of a batch (a single ciphertext is the reference's calling pattern) alone,
  constexpr bool two = TWO;                    // in its own instantiation -- under a run-time branch the pair form needs 168 registers instead of 116
  const size_t off0 = (size_t)p0 * a.poly_stride + ((size_t)blockIdx.z << a.logn) + wave0;
  const size_t off1 = two ? off0 + a.poly_stride : off0;
  const size_t koff = ((size_t)blockIdx.z << a.logn) + wave0;
  uint64_t x0[8], x1[8], e0[8], e1[8];
  Tw8<TW, LOW> tw;
  ln.load_h(x0, a.src[0] + off0);
  tw.load_h(ln, wave0, a.logn, wf);
  if constexpr (two) ln.load_h(x1, a.src[0] + off1);
  L8::ct_h(x0, tw.t, k);
  if constexpr (two) L8::ct_h(x1, tw.t, k);
  tw.load_m(ln, wave0, a.logn, wf);
  ln.h_to_m(x0); L8::ct_m(x0, tw.t, k);
  if constexpr (two) { ln.h_to_m(x1); L8::ct_m(x1, tw.t, k); }
  tw.load_l(ln, wave0, a.logn, wf);
  ln.m_to_l(x0); L8::ct_l(x0, tw.u, k);
  if constexpr (two) { ln.m_to_l(x1); L8::ct_l(x1, tw.u, k); }
  ln.load_l(e0, ka.evk0 + koff);
  ln.load_l(e1, ka.evk1 + koff);
  tw.load_l(ln, wave0, a.logn, wi);
  uint64_t y0[8], y1[8];                        // x0 * evk1, x1 * evk1;  x0, x1 become x0 * evk0, x1 * evk0
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint64_t u0 = TT::right(x0[e], k);
    x0[e] = TT::inv_from4(mulmod_lazy(u0, e0[e], k), k);
    y0[e] = TT::inv_from4(mulmod_lazy(u0, e1[e], k), k);
  }
  if constexpr (two) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint64_t u1 = TT::right(x1[e], k);
      x1[e] = TT::inv_from4(mulmod_lazy(u1, e0[e], k), k);
      y1[e] = TT::inv_from4(mulmod_lazy(u1, e1[e], k), k);
    }
  }
  L8::gs_l(x0, tw.u, k); ln.l_to_m(x0);
  L8::gs_l(y0, tw.u, k); ln.l_to_m(y0);
  if constexpr (two) {
    L8::gs_l(x1, tw.u, k); ln.l_to_m(x1);
    L8::gs_l(y1, tw.u, k); ln.l_to_m(y1);
  }
  tw.load_m(ln, wave0, a.logn, wi);
  L8::gs_hm(x0, tw.t, k); ln.m_to_h(x0);
  L8::gs_hm(y0, tw.t, k); ln.m_to_h(y0);
  if constexpr (two) {
    L8::gs_hm(x1, tw.t, k); ln.m_to_h(x1);
    L8::gs_hm(y1, tw.t, k); ln.m_to_h(y1);
  }
  tw.load_h(ln, wave0, a.logn, wi);
  L8::gs_hm(x0, tw.t, k);
  ln.store_h(a.dst[0] + off0, x0);
  L8::gs_hm(y0, tw.t, k);
  ln.store_h(a.dst[1] + off0, y0);
  if constexpr (two) {
    L8::gs_hm(x1, tw.t, k);
    ln.store_h(a.dst[0] + off1, x1);
    L8::gs_hm(y1, tw.t, k);
    ln.store_h(a.dst[1] + off1, y1);
  }
}

// contig_pass in the 8-per-lane geometry (n = 2^17): CONTIG8_POLYS polynomials of the same limb and tile share each twiddle group.
constexpr int CONTIG8_POLYS = 2;
template <bool INV, typename TW, int LOW, bool NT = false>
__global__ __launch_bounds__(CONTIG_WAVES * 64, 4) void contig_pass8(PassArgs a, unsigned polys) {
  using TT = TwTraits<TW>;
  using L8 = Lane8N<LOW, NT>;
  __shared__ uint64_t lds[CONTIG_WAVES * LANE8_LDS_PER_WAVE];
  L8 ln(lds + (threadIdx.x >> 6) * LANE8_LDS_PER_WAVE);
  const unsigned wave0 = (blockIdx.x * CONTIG_WAVES + (threadIdx.x >> 6)) << 9;
  const unsigned limb = a.limb0 + blockIdx.z;
  const PrimeK k = a.tabs[limb].k;
  const unsigned p0 = blockIdx.y * CONTIG8_POLYS;
  const unsigned cnt = polys - p0 < CONTIG8_POLYS ? polys - p0 : CONTIG8_POLYS;
  const size_t off = (size_t)p0 * a.poly_stride + ((size_t)blockIdx.z << a.logn) + wave0;
  const uint64_t *__restrict__ src = a.src[0] + off;
  uint64_t *__restrict__ dst = a.dst[0] + off;
  const TW *__restrict__ wt = TT::table(a, INV) + ((size_t)limb << a.logn);
  uint64_t x[CONTIG8_POLYS][8];
  Tw8<TW, LOW> tw;
#pragma unroll
  for (int j = 0; j < CONTIG8_POLYS; ++j)
    if (j < (int)cnt) ln.load_h(x[j], src + (size_t)j * a.poly_stride);
  if (!INV) {
    tw.load_h(ln, wave0, a.logn, wt);
#pragma unroll
    for (int j = 0; j < CONTIG8_POLYS; ++j) if (j < (int)cnt) L8::ct_h(x[j], tw.t, k);
    tw.load_m(ln, wave0, a.logn, wt);
#pragma unroll
    for (int j = 0; j < CONTIG8_POLYS; ++j) if (j < (int)cnt) { ln.h_to_m(x[j]); L8::ct_m(x[j], tw.t, k); }
    tw.load_l(ln, wave0, a.logn, wt);
#pragma unroll
    for (int j = 0; j < CONTIG8_POLYS; ++j)
      if (j < (int)cnt) {
        ln.m_to_l(x[j]);
        L8::ct_l(x[j], tw.u, k);
        uint32_t zm = ~0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[j][e] = TT::canon_fwd(x[j][e], k); zero_watch(zm, x[j][e]); }
        zero_note(a, zm, p0 + j, blockIdx.z);
        ln.l_to_h(x[j]);
        ln.store_h(dst + (size_t)j * a.poly_stride, x[j]);
      }
  } else {
    tw.load_l(ln, wave0, a.logn, wt);
#pragma unroll
    for (int j = 0; j < CONTIG8_POLYS; ++j) if (j < (int)cnt) { ln.h_to_l(x[j]); L8::gs_l_canon(x[j], tw.u, k); }
    tw.load_m(ln, wave0, a.logn, wt);
#pragma unroll
    for (int j = 0; j < CONTIG8_POLYS; ++j) if (j < (int)cnt) { ln.l_to_m(x[j]); L8::gs_hm(x[j], tw.t, k); }
    tw.load_h(ln, wave0, a.logn, wt);
#pragma unroll
    for (int j = 0; j < CONTIG8_POLYS; ++j)
      if (j < (int)cnt) {
        ln.m_to_h(x[j]);
        L8::gs_hm(x[j], tw.t, k);
        ln.store_h(dst + (size_t)j * a.poly_stride, x[j]);
      }
  }
}

// ---------------------------------------------------------------------------
// Pointwise limb ops on whole slabs: poly_rns_mul / poly_rns_add,
// src/poly.c:71-82.  16 bytes per lane, grid-stride free (exact grid).
// ---------------------------------------------------------------------------
template <bool MUL>
__global__ __launch_bounds__(256) void pointwise(PassArgs a) {
  const LimbTab &tab = a.tabs[a.limb0 + blockIdx.z];
  const PrimeK k = tab.k;
  const unsigned i2 = 2 * (blockIdx.x * 256 + threadIdx.x);
  if (i2 >= (1u << a.logn)) return;
  const size_t off = (size_t)blockIdx.y * a.poly_stride + ((size_t)blockIdx.z << a.logn) + i2;
  const ulonglong2 u = *reinterpret_cast<const ulonglong2 *>(a.src[0] + off);
  const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(a.src[1] + off);
  ulonglong2 r;
  if (MUL) { r.x = mulmod_canon(u.x, v.x, k); r.y = mulmod_canon(u.y, v.y, k); }
  else     { r.x = addmod_canon(u.x, v.x, k); r.y = addmod_canon(u.y, v.y, k); }
  *reinterpret_cast<ulonglong2 *>(a.dst[0] + off) = r;
}

// ---------------------------------------------------------------------------
// Small rings (n <= 2^SMALL_MAX_LOGN): one workgroup per limb, the limb lives
// in LDS, radix-2 stages exactly as src/ntt.c walks them.  Also the in-device
// cross-check for the two-pass kernels in the tests.
// ---------------------------------------------------------------------------
constexpr int SMALL_MAX_LOGN = 12;

template <bool INV, typename TW>
__global__ __launch_bounds__(256) void small_ntt(PassArgs a) {
  using TT = TwTraits<TW>;
  __shared__ uint64_t s[1 << SMALL_MAX_LOGN];
  const LimbTab &tab = a.tabs[a.limb0 + blockIdx.z];
  const PrimeK k = tab.k;
  const LastK<TW> last(tab);
  const unsigned n = 1u << a.logn;
  const TW *__restrict__ wt = TT::table(a, INV) + ((size_t)(a.limb0 + blockIdx.z) << a.logn);
  const unsigned slab = blockIdx.y % a.nslab, poly = blockIdx.y / a.nslab;
  const size_t off = (size_t)poly * a.poly_stride + ((size_t)blockIdx.z << a.logn);
  const uint64_t *__restrict__ src = a.src[slab] + off;
  uint64_t *__restrict__ dst = a.dst[slab] + off;
  for (unsigned i = threadIdx.x; i < n; i += 256) s[i] = src[i];
  __syncthreads();
  if (!INV) {
    for (unsigned len = n >> 1; len >= 1; len >>= 1) {
      for (unsigned b = threadIdx.x; b < (n >> 1); b += 256) {
        const unsigned blk = b / len, j = blk * 2 * len + (b % len);
        uint64_t x = s[j], y = s[j + len];
        ct_bfly(x, y, wt[n / (2 * len) + blk], k);
        s[j] = TT::canon_fwd(x, k); s[j + len] = TT::canon_fwd(y, k);
      }
      __syncthreads();
    }
  } else {
    for (unsigned len = 1; len < (n >> 1); len <<= 1) {
      for (unsigned b = threadIdx.x; b < (n >> 1); b += 256) {
        const unsigned blk = b / len, j = blk * 2 * len + (b % len);
        uint64_t x = s[j], y = s[j + len];
        gs_bfly(x, y, wt[n / (2 * len) + blk], k);
        s[j] = TT::canon_inv(x, k); s[j + len] = TT::canon_inv(y, k);
      }
      __syncthreads();
    }
    if (n >= 2) {
      for (unsigned b = threadIdx.x; b < (n >> 1); b += 256) {
        uint64_t x = s[b], y = s[b + (n >> 1)];
        gs_last(x, y, last, k);
        s[b] = x; s[b + (n >> 1)] = y;
      }
      __syncthreads();
    }
  }
  uint32_t zm = ~0u;
  for (unsigned i = threadIdx.x; i < n; i += 256) { dst[i] = s[i]; zero_watch(zm, s[i]); }
  if (!INV && a.nslab == 1) zero_note(a, zm, blockIdx.y, blockIdx.z);
}

// ---------------------------------------------------------------------------
// The reference's representation of zero.  src/ntt.c:45-48 keeps its data in [0, p], not [0, p): the sum leg
//   a[j] = (a[j] <= q - t) ? a[j] + t : a[j] + t - q
// stores q itself when a[j] + t == q, and such a q survives a later stage whenever its partner's product t is 0.  A
// finished forward transform therefore holds p at some of the positions whose residue is 0 (every other residue is
// canonical: the product leg fqmul is).  The lazy kernels above cannot tell which; instead the forward kernels that hand
// canonical output to the caller flag every (polynomial, limb) whose output contains a 0 (zero_watch) and this kernel
// redoes the flagged limbs with the reference's own arithmetic, literally: src/ntt.c:54-73 on the canonical output gives the
// canonical input back (the input domain of gpq_ntt), src/ntt.c:37-52 on that gives the reference's output.
// fqmul(a, zeta) of src/ntt.c:32-35 is a * zeta_std mod p in [0, p) for every 64-bit a (montgomery_reduce, src/reduce.c:59-66:
// hi < p, t < p), which is what ref_fqmul computes from the standard-form tables; everything else is the reference's
// unsigned 64-bit arithmetic as written, wrap-around included, so mode 1 / 2 reproduce src/ntt.c for ANY input words.
// One workgroup per limb, stages through global memory (L2-resident: a limb is at most 1 MiB), a barrier per stage.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ref_fqmul(uint64_t a, uint64_t w, const PrimeK &k) {
  if (a >= 2 * k.p4) a %= k.p;                // only words outside the reference's own [0, p] domain get here
  return mulmod_canon(a, w, k);
}
template <bool INV>
__device__ void ref_transform_limb(uint64_t *__restrict__ v, unsigned logn, const uint64_t *__restrict__ w, const LimbTab &tab) {
  const PrimeK k = tab.k;
  const uint64_t q = k.p;
  const unsigned n = 1u << logn, half = n >> 1, T = blockDim.x;
  if (!INV) {
    for (unsigned ls = logn; ls-- > 0;) {                          // len = 2^ls = n/2 .. 1          src/ntt.c:42
      const unsigned len = 1u << ls;
      for (unsigned b = threadIdx.x; b < half; b += T) {
        const unsigned blk = b >> ls, j = (blk << (ls + 1)) + (b & (len - 1));
        const uint64_t zeta = w[(n >> (ls + 1)) + blk];            // zetas[k++], k = n/(2 len) + block index
        const uint64_t x = v[j];
        const uint64_t t = ref_fqmul(v[j + len], zeta, k);
        v[j + len] = (x >= t) ? x - t : x - t + q;                 // src/ntt.c:46
        v[j] = (x <= q - t) ? x + t : x + t - q;                   // src/ntt.c:47
      }
      __syncthreads();
    }
  } else {
    for (unsigned ls = 0; ls < logn; ++ls) {                       // len = 1 .. n/2                  src/ntt.c:60
      const unsigned len = 1u << ls;
      for (unsigned b = threadIdx.x; b < half; b += T) {
        const unsigned blk = b >> ls, j = (blk << (ls + 1)) + (b & (len - 1));
        const uint64_t zeta_inv = w[(n >> (ls + 1)) + blk];
        const uint64_t t = v[j], y = v[j + len];
        v[j] = (y <= q - t) ? t + y : t + y - q;                   // src/ntt.c:65
        const uint64_t d = (y <= t) ? t - y : t - y + q;           // src/ntt.c:66
        v[j + len] = ref_fqmul(d, zeta_inv, k);                    // src/ntt.c:67
      }
      __syncthreads();
    }
    for (unsigned i = threadIdx.x; i < n; i += T) v[i] = ref_fqmul(v[i], tab.ninv, k);   // src/ntt.c:71-72
    __syncthreads();
  }
}

// mode 0: redo the flagged limbs of a finished gpq_ntt (and clear their flags); 1: src/ntt.c:37-52 on every limb;
// 2: src/ntt.c:54-73 on every limb.  a.dst[0] = slab, grid.x workgroups walk the polys * limbs limbs of the launch.
__global__ __launch_bounds__(1024) void ref_zero_redo(PassArgs a, unsigned polys, unsigned limbs, unsigned mode) {
  const unsigned total = polys * limbs;
  for (unsigned idx = blockIdx.x; idx < total; idx += gridDim.x) {
    const unsigned poly = idx / limbs, z = idx % limbs;
    unsigned *flag = a.zflag + (size_t)poly * a.zstride + z;
    if (mode == 0 && *flag == 0) continue;                         // uniform per workgroup
    const unsigned limb = a.limb0 + z;
    const LimbTab &tab = a.tabs[limb];
    uint64_t *v = a.dst[0] + (size_t)poly * a.poly_stride + ((size_t)z << a.logn);
    const size_t toff = (size_t)limb << a.logn;
    if (mode != 1) ref_transform_limb<true>(v, a.logn, a.winv + toff, tab);
    if (mode != 2) ref_transform_limb<false>(v, a.logn, a.w + toff, tab);
    if (mode == 0) { __syncthreads(); if (threadIdx.x == 0) *flag = 0; }
  }
}

}  // namespace gpq
