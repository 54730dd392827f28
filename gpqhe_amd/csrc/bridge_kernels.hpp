// bridge_kernels.hpp -- the MPI <-> RNS bridge on the device.
//
// The reference keeps ciphertext coefficients as libgcrypt big integers and
// crosses into RNS form around every multiplication:
//   rns_decompose    src/rns.c:37-48     ahat[i] = a[i] mod p        (floor mod)
//   rns_reconstruct  src/rns.c:60-75     a = sum_d ahat_d * (phat_d * phat_invmp_d mod P) mod P
//   poly_rns2mpi     src/poly.c:109-120  reconstruct, mpi_smod(P, P/2), mpi_smod(q, q/2)
//   mpi_smod         src/types.c:108-113 r mod q, minus q when r >= floor(q/2)
//   mpi_rdiv         src/types.c:115-128 floor(a/m), plus one when the remainder > floor(m/2)
//   he_rs            src/he-rescale.c:33-54
// Here a polynomial of big integers is a "big slab": uint64_t[W][n], word j of
// coefficient i at j*n + i (word-major, so a wave reads 64 consecutive words),
// little-endian words, two's complement over 64*W bits.
#pragma once
#include "modarith.hpp"
#include "tables.hpp"

namespace gpq {

// A batch whose polynomials live in two places: polynomials [0, split) at a, the rest at b (a null place: those polynomials
// have none -- an absent addend).  The relinearisation tail runs c0 and c1 of a launch group as ONE batch of 2 x polys
// polynomials: their inputs are adjacent in the workspace, their outputs (and he_swk's single addend) are the caller's separate slabs.
template <typename T>
struct Two {
  T *a, *b;
  unsigned split;
  __device__ __forceinline__ T *at(unsigned poly, size_t stride) const {
    return poly < split ? (a ? a + (size_t)poly * stride : nullptr) : (b ? b + (size_t)(poly - split) * stride : nullptr);
  }
};
template <typename T> inline Two<T> one_place(T *a) { return Two<T>{a, nullptr, ~0u}; }

// What a fast kernel of bridge_stream.hpp leaves for the masked kernels behind it: one word per wave of ITS launch, non-zero when that
// wave flagged a coefficient; wave w of the producer owned the groups of 64 coefficients w, w + waves, w + 2 waves, ...  A masked kernel
// given a scope is launched as ceil(waves / (its threads / 64)) workgroups: its wave w returns at once when word w is 0 (the common
// case: 2^-38 of the coefficients are ever flagged) and otherwise walks the producer wave's groups, still checking its per-coefficient mask.
struct FlagScope {
  const unsigned *wave_any;
  unsigned waves, total_groups;
};
constexpr FlagScope kNoScope{nullptr, 0u, 0u};

// body(poly, i) for the coefficients of a masked kernel: the scope's groups, or the classic grid (n / threads, polys)
template <class Body>
__device__ __forceinline__ void scoped_or_grid(const FlagScope &sc, unsigned logn, unsigned threads, Body body) {
  if (sc.wave_any) {
    const unsigned wv = blockIdx.x * (threads >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wv >= sc.waves || !sc.wave_any[wv]) return;
    const unsigned lg = logn - 6;
    for (unsigned g = wv; g < sc.total_groups; g += sc.waves) body(g >> lg, ((g & ((1u << lg) - 1)) << 6) + lane);
  } else {
    const unsigned i = blockIdx.x * threads + threadIdx.x;
    if (i < (1u << logn)) body(blockIdx.y, i);
  }
}


constexpr uint64_t M59 = (1ull << 59) - 1;

// r*2^59 + d (mod p), lazily: r < 4p, d < 2^59 -> (0, 4p).
//   2^59 == -c:  r*2^59 + d == d - c*r ; t = c*r < 2^89.3, t = th*2^59 + tl == tl - c*th
//   => d - tl + c*th == d + (2^59-1-tl) + c*th + (c+1)  (mod p), every term non-negative.
__device__ __forceinline__ uint64_t horner59(uint64_t r, uint64_t d, const PrimeK &k) {
  const uint64_t t0 = mad_u64(k.c, (uint32_t)r, 0);
  const uint64_t t1 = mad_u64(k.c, (uint32_t)(r >> 32), (uint32_t)(t0 >> 32));  // t = t1 : lo32(t0)
  const uint32_t t1lo = (uint32_t)t1, t1hi = (uint32_t)(t1 >> 32);
  const uint32_t th = __builtin_amdgcn_alignbit(t1hi, t1lo, 27);
  const uint64_t ntl = pack64(~(uint32_t)t0, ~t1lo & 0x7ffffffu);
  return mad_u64(k.c, th, d + k.c1) + ntl;
}

// ---------------------------------------------------------------------------
// rns_decompose: big slab -> limb-major slab, every limb of one coefficient by
// the same thread (the words are read once, the limb loop is uniform).
// MAXW = words kept in registers (the value is sign-extended to 64*MAXW bits).
// ---------------------------------------------------------------------------
// Up to four big slabs decomposed by one launch (the four polynomials of he_mul's two ciphertexts are separate slabs of the
// caller): polynomial k of the launch is polynomial k % per of slab k / per.
struct BigSources {
  const uint64_t *src[4];
  unsigned per;             // polynomials per source slab
  __device__ __forceinline__ const uint64_t *at(unsigned poly, size_t stride) const { return src[poly / per] + (size_t)(poly % per) * stride; }
};
inline BigSources one_source(const uint64_t *big) { return BigSources{{big, nullptr, nullptr, nullptr}, ~0u}; }

struct DecomposeArgs {
  const LimbTab *tabs;
  BigSources big;           // [polys][W][n]
  uint64_t *slab;           // [polys][dim][n]: limb d of the output is prime limb0 + d
  unsigned W, dim, logn, limb0;
  const unsigned char *only;   // optional [polys][n]: when given, coefficients with 0 are skipped
  FlagScope scope;
};

template <int MAXW>
__device__ __forceinline__ void decompose_body(const DecomposeArgs &a, unsigned poly, unsigned i) {
  constexpr int ND = (64 * MAXW + 58) / 59;             // 59-bit digits, the top one signed
  constexpr int TOPBITS = 64 * MAXW - 59 * (ND - 1);
  if (a.only && !a.only[((size_t)poly << a.logn) + i]) return;
  const uint64_t *__restrict__ src = a.big.at(poly, (size_t)a.W << a.logn) + i;
  uint64_t w[MAXW + 1];
#pragma unroll
  for (int j = 0; j < MAXW; ++j) w[j] = j < (int)a.W ? src[(size_t)j << a.logn] : 0;
  // sign of the value = top bit of its last real word; found without a runtime register index
  uint64_t sx = 0;
#pragma unroll
  for (int j = 0; j < MAXW; ++j) if (j == (int)a.W - 1) sx = (uint64_t)((int64_t)w[j] >> 63);
#pragma unroll
  for (int j = 0; j < MAXW; ++j) if (j >= (int)a.W) w[j] = sx;
  w[MAXW] = sx;
  uint64_t dg[ND];
#pragma unroll
  for (int t = 0; t < ND; ++t) {
    const int bit = 59 * t, wi = bit >> 6, sh = bit & 63;
    const uint64_t lo = w[wi] >> sh;
    const uint64_t hi = sh ? (w[wi + 1] << (64 - sh)) : 0;
    dg[t] = (lo | hi) & M59;
  }
  // top digit as a signed number of TOPBITS bits
  const int64_t top = ((int64_t)(dg[ND - 1] << (64 - TOPBITS))) >> (64 - TOPBITS);
  uint64_t *__restrict__ dst = a.slab + ((size_t)poly * a.dim << a.logn) + i;
  for (unsigned d = 0; d < a.dim; ++d) {
    const PrimeK k = a.tabs[a.limb0 + d].k;
    uint64_t r = top < 0 ? (uint64_t)top + k.p : (uint64_t)top;     // [0, p + 2^58)
#pragma unroll
    for (int t = ND - 2; t >= 0; --t) r = horner59(r, dg[t], k);
    dst[(size_t)d << a.logn] = canon4(r, k);
  }
}
template <int MAXW>
__global__ __launch_bounds__(256) void bridge_decompose(DecomposeArgs a) {
  scoped_or_grid(a.scope, a.logn, 256, [&](unsigned poly, unsigned i) { decompose_body<MAXW>(a, poly, i); });
}

// ---------------------------------------------------------------------------
// poly_rns2mpi: limb-major slab -> big slab.
//   y_d = ahat_d * phat_invmp_d mod p_d          (one modular multiply per limb)
//   S   = sum_d y_d * phat_d      < dim * P      (multiword, WP+1 words in registers)
//   S mod P by conditional subtraction of 32P, 16P, ..., P (dim <= 63)
//   centre mod P (>= floor(P/2) -> minus P), then centre mod q = 2^logq.
// Equal to the reference's sum of (ahat_d * C_d mod P) mod P with C_d = phat_d*phat_invmp_d mod P.
// ---------------------------------------------------------------------------
struct ReconstructArgs {
  const LimbTab *tabs;
  const uint64_t *slab;        // [polys][dim][n]
  Two<uint64_t> big;           // [polys][Wout][n]
  const uint64_t *phat;        // [dim][WP]       P / p_d
  const uint64_t *phat_inv;    // [dim]           (P/p_d)^-1 mod p_d
  const uint64_t *pmult;       // [6][WP+1]       32P, 16P, 8P, 4P, 2P, P
  const uint64_t *phalf;       // [WP+1]          floor(P/2)
  unsigned char *tie;          // optional [polys][n]: 1 where (S mod P) == floor(P/2) exactly
  const unsigned char *only;   // optional [polys][n]: when given, coefficients with 0 are skipped
  const uint64_t *inv128;      // [dim][2]  floor(2^128 / p_d)        (fast path only)
  unsigned dim, logn, Wout, logq;   // logq = 0: no reduction mod q (Wout >= WP+1 then)
  unsigned limb0;              // the basis is primes limb0 .. limb0+dim-1
  unsigned slab_dim, slab_first;    // the slab has slab_dim limbs per polynomial; read limbs slab_first ..
  unsigned centre;             // 0: leave the result in [0, P)
  unsigned prescaled;          // the slab already holds y_d (bridge_relin_front_mfma writes Q's residues so)
  FlagScope scope;             // with `only`: the launch that wrote it (bridge_stream.hpp); bridge_reconstruct alone reads it
};

template <int WP>
__device__ __forceinline__ void reconstruct_body(const ReconstructArgs &a, unsigned poly, unsigned i) {
  if (a.only && !a.only[((size_t)poly << a.logn) + i]) return;
  const uint64_t *__restrict__ src = a.slab + (((size_t)poly * a.slab_dim + a.slab_first) << a.logn) + i;
  uint64_t S[WP + 1];
#pragma unroll
  for (int j = 0; j <= WP; ++j) S[j] = 0;
  for (unsigned d = 0; d < a.dim; ++d) {
    const PrimeK k = a.tabs[a.limb0 + d].k;
    const uint64_t xd = src[(size_t)d << a.logn];
    const uint64_t y = a.prescaled ? xd : mulmod_canon(xd, a.phat_inv[d], k);
    const uint64_t *__restrict__ ph = a.phat + (size_t)d * WP;
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const u128 t = (u128)y * ph[j] + S[j] + carry;
      S[j] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    S[WP] += carry;
  }
  // S mod P
#pragma unroll 1
  for (int m = 0; m < 6; ++m) {
    const uint64_t *__restrict__ mp = a.pmult + (size_t)m * (WP + 1);
    uint64_t T[WP + 1];
    uint64_t borrow = 0;
#pragma unroll
    for (int j = 0; j <= WP; ++j) {
      const u128 t = (u128)S[j] - mp[j] - borrow;
      T[j] = (uint64_t)t;
      borrow = (uint64_t)(t >> 64) & 1;
    }
    if (!borrow) {
#pragma unroll
      for (int j = 0; j <= WP; ++j) S[j] = T[j];
    }
  }
  // mpi_smod(., P, P/2): r >= floor(P/2) -> r - P
  if (a.centre) {
    uint64_t borrow = 0, any = 0;
#pragma unroll
    for (int j = 0; j <= WP; ++j) {
      const u128 t = (u128)S[j] - a.phalf[j] - borrow;
      any |= (uint64_t)t;
      borrow = (uint64_t)(t >> 64) & 1;
    }
    if (a.tie) a.tie[((size_t)poly << a.logn) + i] = !borrow && !any;
    if (!borrow) {
      const uint64_t *__restrict__ mp = a.pmult + (size_t)5 * (WP + 1);
      uint64_t b2 = 0;
#pragma unroll
      for (int j = 0; j <= WP; ++j) {
        const u128 t = (u128)S[j] - mp[j] - b2;
        S[j] = (uint64_t)t;
        b2 = (uint64_t)(t >> 64) & 1;
      }
    }
  }
  // mpi_smod(., 2^logq, 2^(logq-1)): keep logq bits, sign-extend from bit logq-1
  uint64_t *__restrict__ dst = a.big.at(poly, (size_t)a.Wout << a.logn) + i;
  const uint64_t sext = (uint64_t)((int64_t)S[WP] >> 63);
  uint64_t qsign = 0;
  if (a.logq) {
    const unsigned sb = a.logq - 1;
#pragma unroll
    for (int j = 0; j <= WP; ++j) if (j == (int)(sb >> 6)) qsign = 0 - ((S[j] >> (sb & 63)) & 1);
  }
#pragma unroll
  for (int j = 0; j <= WP; ++j) {
    if (j < (int)a.Wout) {
      uint64_t v = S[j];
      if (a.logq) {
        const int lo = 64 * j;
        if (lo >= (int)a.logq) v = qsign;
        else if (lo + 64 > (int)a.logq) {
          const uint64_t mask = (1ull << (a.logq - lo)) - 1;
          v = (v & mask) | (qsign & ~mask);
        }
      }
      dst[(size_t)j << a.logn] = v;
    }
  }
  const uint64_t fill = a.logq ? qsign : sext;
  for (unsigned j = WP + 1; j < a.Wout; ++j) dst[(size_t)j << a.logn] = fill;
}
template <int WP>
__global__ __launch_bounds__(128) void bridge_reconstruct(ReconstructArgs a) {
  scoped_or_grid(a.scope, a.logn, 128, [&](unsigned poly, unsigned i) { reconstruct_body<WP>(a, poly, i); });
}

// ---------------------------------------------------------------------------
// poly_rns2mpi, fast path for q = 2^logq: the result only needs its low WL >= ceil(logq/64)
// words, so the CRT sum is accumulated modulo 2^(64 WL); the multiple of P to take off and the
// centring decision come from a fixed-point estimate of S/P = sum_d y_d / p_d:
//   F = sum_d y_d * floor(2^128/p_d)   underestimates 2^128 * S/P by less than dim * 2^60 <= 2^66
//   k = F >> 128,  centred  <=>  frac >= 1/2,  multiple of P to take off = k + centred.
// If the true fraction crosses 1 while F still reads 1 - eps, the multiple is k+1 either way (centred
// from below, or not centred with the next k), so only the 1/2 boundary is a real discontinuity:
// coefficients with frac(F) in [1/2 - 2^-61, 1/2) are flagged and redone by the exact kernel above,
// which also handles the exact tie S mod P == floor(P/2).  (he_mul's products sit far from +-P/2.)
// ---------------------------------------------------------------------------
template <int WL>
__global__ __launch_bounds__(256) void bridge_reconstruct_low(ReconstructArgs a, unsigned WPstride, unsigned char *redo) {
  const unsigned n = 1u << a.logn;
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t *__restrict__ src = a.slab + (((size_t)blockIdx.y * a.slab_dim + a.slab_first) << a.logn) + i;
  uint64_t S[WL];
#pragma unroll
  for (int j = 0; j < WL; ++j) S[j] = 0;
  uint64_t f0 = 0, f1 = 0, f2 = 0;
  for (unsigned d = 0; d < a.dim; ++d) {
    const PrimeK k = a.tabs[a.limb0 + d].k;
    const uint64_t xd = src[(size_t)d << a.logn];
    const uint64_t y = a.prescaled ? xd : mulmod_canon(xd, a.phat_inv[d], k);
    const uint64_t *__restrict__ ph = a.phat + (size_t)d * WPstride;
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < WL; ++j) {
      const u128 t = (u128)y * ph[j] + S[j] + carry;
      S[j] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    const u128 g0 = (u128)y * a.inv128[2 * d] + f0;
    const u128 g1 = (u128)y * a.inv128[2 * d + 1] + f1 + (uint64_t)(g0 >> 64);
    f0 = (uint64_t)g0; f1 = (uint64_t)g1; f2 += (uint64_t)(g1 >> 64);
  }
  const size_t flag_at = ((size_t)blockIdx.y << a.logn) + i;
  // frac = f1:f0 / 2^128 ; slack 2^67 on both decisions
  const bool ambiguous = (f1 >> 3) == ((1ull << 60) - 1);               // frac in [1/2 - 2^-61, 1/2)
  redo[flag_at] = ambiguous;
  if (a.tie) a.tie[flag_at] = 0;
  if (ambiguous) return;                                                // the exact kernel writes this coefficient
  const uint64_t mult = f2 + (f1 >> 63);                                // k, plus one when centring takes P off once more
  const uint64_t *__restrict__ P = a.pmult + (size_t)5 * (WPstride + 1);
  uint64_t borrow = 0, mcarry = 0;
#pragma unroll
  for (int j = 0; j < WL; ++j) {
    const u128 kp = (u128)mult * P[j] + mcarry;
    mcarry = (uint64_t)(kp >> 64);
    const u128 t = (u128)S[j] - (uint64_t)kp - borrow;
    S[j] = (uint64_t)t;
    borrow = (uint64_t)(t >> 64) & 1;
  }
  uint64_t *__restrict__ dst = a.big.at(blockIdx.y, (size_t)a.Wout << a.logn) + i;
  const unsigned sb = a.logq - 1;
  uint64_t qsign = 0;
#pragma unroll
  for (int j = 0; j < WL; ++j) if (j == (int)(sb >> 6)) qsign = 0 - ((S[j] >> (sb & 63)) & 1);
#pragma unroll
  for (int j = 0; j < WL; ++j) {
    if (j < (int)a.Wout) {
      uint64_t v = S[j];
      const int lo = 64 * j;
      if (lo >= (int)a.logq) v = qsign;
      else if (lo + 64 > (int)a.logq) {
        const uint64_t mask = (1ull << (a.logq - lo)) - 1;
        v = (v & mask) | (qsign & ~mask);
      }
      dst[(size_t)j << a.logn] = v;
    }
  }
  for (unsigned j = WL; j < a.Wout; ++j) dst[(size_t)j << a.logn] = qsign;
}

// ---------------------------------------------------------------------------
// Tail of he_relin / he_swk (src/he-mult.c:67-77, src/he-automorphism.c:68-76) for q_l = 2^k:
//   c  = poly_rns2mpi(chat, P*q_l)         x = CRT(chat) centred mod Pi_B, c == x (mod P*2^k)
//   c  = mpi_rdiv(c, P)                    floor(c/P) + [c mod P > floor(P/2)]
//   c  = mpi_smod(mpi_addm(c, d, q_l))     only c mod 2^k matters
// c mod P == x mod P and floor(c/P) == floor(x/P) (mod 2^k), so with r = x mod P (CRT over
// the first dimP limbs, where P = p_0..p_{dimP-1}) the quotient Q = (x - r)/P is an exact
// division, done limb-wise on the remaining limbs:  Qhat_d = (chat_d - r) * P^-1 mod p_d.
// ---------------------------------------------------------------------------
struct ExactDivArgs {
  const LimbTab *tabs;
  const uint64_t *chat;        // [polys][dimB][n]
  const uint64_t *rhat;        // [polys][cnt][n]   r mod p_d, d = dimP ..
  uint64_t *qhat;              // [polys][cnt][n]
  const uint64_t *pinv;        // [cnt]             P^-1 mod p_d
  unsigned dimB, dimP, cnt, logn;
};

__global__ __launch_bounds__(256) void bridge_exactdiv(ExactDivArgs a) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (1u << a.logn)) return;
  const unsigned d = blockIdx.z;
  const PrimeK k = a.tabs[a.dimP + d].k;
  const uint64_t x = a.chat[(((size_t)blockIdx.y * a.dimB + a.dimP + d) << a.logn) + i];
  const size_t o = (((size_t)blockIdx.y * a.cnt + d) << a.logn) + i;
  a.qhat[o] = mulmod_canon(x + k.p - a.rhat[o], a.pinv[d], k);   // (x - r) in (0, 2p)
}

// out = smod(Qc + [r > floor(P/2)] + d, 2^k), W words.  Qc = centred CRT of Qhat already reduced
// smod 2^k; `tie` marks the coefficients whose quotient sat exactly on floor(Pi'/2), where the
// centring of x (not of Q) decides the wrap: add Pi' back when r < floor(P/2)  (see DESIGN.md).
struct AddRoundArgs {
  Two<uint64_t> out;           // [polys][W][n]
  Two<const uint64_t> qc;      // [polys][W][n]
  const uint64_t *r;           // [polys][Wr][n]   in [0, P)
  Two<const uint64_t> d;       // [polys][W][n] or null
  const uint64_t *phalf;       // [Wr]  floor(P/2)
  const uint64_t *piq;         // [>= W words of Pi' (low words)]
  const unsigned char *tie;    // [polys][n]
  unsigned W, Wr, logn, logql;
  const unsigned char *only;   // optional [polys][n]: when given, coefficients with 0 are skipped
  const unsigned char *rflags; // optional [polys][n]: r against floor(P/2) as RF_GT / RF_LT bits instead of the words of r
  FlagScope scope;             // with `only`: the launch that wrote it
};

__device__ __forceinline__ void addround_body(const AddRoundArgs &a, unsigned poly, unsigned i) {
  if (a.only && !a.only[((size_t)poly << a.logn) + i]) return;
  int cmp = 0;
  if (a.rflags) {
    const unsigned char f = a.rflags[((size_t)poly << a.logn) + i];
    cmp = (f & 1) ? 1 : ((f & 2) ? -1 : 0);
  } else {
    const uint64_t *__restrict__ r = a.r + ((size_t)poly * a.Wr << a.logn) + i;
    // compare r with floor(P/2), most significant word first
    for (int j = (int)a.Wr - 1; j >= 0 && cmp == 0; --j) {
      const uint64_t rv = r[(size_t)j << a.logn], hv = a.phalf[j];
      cmp = rv > hv ? 1 : (rv < hv ? -1 : 0);
    }
  }
  uint64_t carry = cmp > 0;                               // mpi_rdiv: round up when r > floor(P/2)
  const bool fix = a.tie[((size_t)poly << a.logn) + i] && cmp < 0;
  const size_t stride = (size_t)a.W << a.logn;
  uint64_t *outp = a.out.at(poly, stride) + i;
  const uint64_t *qcp = a.qc.at(poly, stride) + i;
  const uint64_t *dbase = a.d.at(poly, stride);
  const uint64_t *dp = dbase ? dbase + i : nullptr;
  const unsigned sb = a.logql - 1;
  uint64_t qsign = 0;
  for (unsigned j = 0; j < a.W; ++j) {
    const size_t o = (size_t)j << a.logn;
    const u128 t = (u128)qcp[o] + carry + (dp ? dp[o] : 0) + (fix ? a.piq[j] : 0);
    uint64_t v = (uint64_t)t;
    carry = (uint64_t)(t >> 64);                          // 0..2
    const unsigned lo = 64 * j;
    if (lo + 64 > sb && lo <= sb) qsign = 0 - ((v >> (sb - lo)) & 1);
    if (lo >= a.logql) v = qsign;
    else if (lo + 64 > a.logql) {
      const uint64_t mask = (1ull << (a.logql - lo)) - 1;
      v = (v & mask) | (qsign & ~mask);
    }
    outp[o] = v;
  }
}
__global__ __launch_bounds__(256) void bridge_addround(AddRoundArgs a) {
  scoped_or_grid(a.scope, a.logn, 256, [&](unsigned poly, unsigned i) { addround_body(a, poly, i); });
}

// Round bits of the coefficients bridge_relin_front_mfma could not decide (RF_AMB): r, made exactly by
// bridge_reconstruct for these coefficients only, against floor(P/2).
struct RoundFixArgs {
  const uint64_t *r;           // [polys][Wr][n]   in [0, P), valid where amb != 0
  const uint64_t *phalf;       // [Wr]
  const unsigned char *amb;    // [polys][n]
  unsigned char *flags;        // [polys][n]
  unsigned Wr, logn;
  FlagScope scope;             // the launch whose flagged groups `amb` lies in
};
__device__ __forceinline__ void roundfix_body(const RoundFixArgs &a, unsigned poly, unsigned i) {
  const size_t at = ((size_t)poly << a.logn) + i;
  if (!a.amb[at]) return;
  const uint64_t *__restrict__ r = a.r + ((size_t)poly * a.Wr << a.logn) + i;
  int cmp = 0;
  for (int j = (int)a.Wr - 1; j >= 0 && cmp == 0; --j) {
    const uint64_t rv = r[(size_t)j << a.logn], hv = a.phalf[j];
    cmp = rv > hv ? 1 : (rv < hv ? -1 : 0);
  }
  a.flags[at] = cmp > 0 ? 1 : (cmp < 0 ? 2 : 0);
}
__global__ __launch_bounds__(256) void bridge_roundfix(RoundFixArgs a) {
  scoped_or_grid(a.scope, a.logn, 256, [&](unsigned poly, unsigned i) { roundfix_body(a, poly, i); });
}

// The exact kernels behind a streaming kernel (bridge_stream.hpp), fused where their hand-overs stay inside one thread: every one of them works
// on a coefficient (or, bridge_limb_scale, on a wave's group of 64) by itself, so a chain of them is one launch in which a thread writes what it
// reads next.  In the common case -- nothing flagged -- a launch is a few hundred workgroups that read one word each and return; nine of them
// per launch group were 0.11 ms per 64 ciphertexts, a fifth of what the fast kernels leave to gain.  Scope launches only.
template <int WP, int MAXW>
__global__ __launch_bounds__(128) void bridge_fallback_crt_decompose(ReconstructArgs ra, DecomposeArgs da) {
  scoped_or_grid(ra.scope, ra.logn, 128, [&](unsigned poly, unsigned i) {
    reconstruct_body<WP>(ra, poly, i);                    // d2 of the flagged coefficients into the scratch words ...
    decompose_body<MAXW>(da, poly, i);                    // ... and from there over the key switch's limbs (same thread, same addresses)
  });
}
template <int WPP, int WPQ>
__global__ __launch_bounds__(128) void bridge_fallback_tail_post(ReconstructArgs rr, RoundFixArgs rf, ReconstructArgs qq, AddRoundArgs ar) {
  scoped_or_grid(rr.scope, rr.logn, 128, [&](unsigned poly, unsigned i) {
    reconstruct_body<WPP>(rr, poly, i);                   // r = x mod P exactly, where the front could not decide the rounding (amb)
    roundfix_body(rf, poly, i);                           // its round bit
    reconstruct_body<WPQ>(qq, poly, i);                   // Q exactly for the flagged coefficients (redo)
    addround_body(ar, poly, i);                           // + round + d, mod 2^logq
  });
}

// Big slabs between the layout the kernels use -- word j of coefficient i at j*n + i -- and rows of W words per coefficient (i*W + j),
// the layout a host thread fills or reads as ONE sequential stream (the MPI-typed calls stage through it: gathering 65536 scattered libgcrypt
// integers into 14 strided streams per thread was most of a call's time).  A 64 x W tile goes through LDS so that both sides move
// whole cache lines.  to_rows = 0: rows -> words (after an upload), 1: words -> rows (before a download).
struct BigTransposeArgs { const uint64_t *src; uint64_t *dst; unsigned W, logn, to_rows; };
__global__ __launch_bounds__(64) void bridge_big_transpose(BigTransposeArgs a) {
  __shared__ uint64_t tile[64 * 65];                     // up to 64 words per coefficient, padded
  const unsigned n = 1u << a.logn, i0 = blockIdx.x * 64, lane = threadIdx.x;
  const size_t poly = (size_t)blockIdx.y * a.W << a.logn;
  const unsigned total = 64 * a.W;                        // words of the tile
  if (!a.to_rows) {
    const uint64_t *__restrict__ rows = a.src + poly + (size_t)i0 * a.W;      // 64 consecutive rows: total contiguous words
    for (unsigned k = lane; k < total; k += 64) tile[(k / a.W) * 65 + (k % a.W)] = rows[k];
    __syncthreads();
    uint64_t *__restrict__ words = a.dst + poly + i0 + lane;
    for (unsigned j = 0; j < a.W; ++j) words[(size_t)j << a.logn] = tile[lane * 65 + j];
  } else {
    const uint64_t *__restrict__ words = a.src + poly + i0 + lane;
    for (unsigned j = 0; j < a.W; ++j) tile[lane * 65 + j] = words[(size_t)j << a.logn];
    __syncthreads();
    uint64_t *__restrict__ rows = a.dst + poly + (size_t)i0 * a.W;
    for (unsigned k = lane; k < total; k += 64) rows[k] = tile[(k / a.W) * 65 + (k % a.W)];
  }
  (void)n;
}

// out = a + b (mode 0), a - b (mode 1), -a (mode 2) on big slabs: W-word two's complement per coefficient, word-major, wrapping at 2^(64 W).
// The additive calls of the reference (he_add / he_sub / he_addpt / he_subpt / he_neg, src/he-add.c:32-140) are this followed by the
// centring mpi_smod the rescale kernels already do; one thread per coefficient carries through its W words, every word plane of a
// wavefront is one contiguous run of memory.  out may be a or b.
struct BigAddSubArgs { uint64_t *out; const uint64_t *a, *b; unsigned W, logn, mode; };
__global__ __launch_bounds__(256) void bridge_big_addsub(BigAddSubArgs k) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (1u << k.logn)) return;
  const size_t base = ((size_t)blockIdx.y * k.W << k.logn) + i;
  unsigned carry = k.mode ? 1u : 0u;                       // a - b = a + ~b + 1,  -a = 0 + ~a + 1
  for (unsigned j = 0; j < k.W; ++j) {
    const size_t at = base + ((size_t)j << k.logn);
    const uint64_t x = k.mode == 2 ? 0ull : k.a[at];
    uint64_t y = k.mode == 2 ? k.a[at] : k.b[at];
    if (k.mode) y = ~y;
    const uint64_t s1 = x + y, s2 = s1 + carry;
    carry = (s1 < x) | (s2 < s1);
    k.out[at] = s2;
  }
}

// chat[poly][d][i] <- chat[poly][d][i] * scale[d] mod p_d, for every coefficient (only == nullptr) or for the groups of 64 coefficients that
// hold a non-zero entry of `only`: puts the CRT weights of the one-product tail on a raw slab (gpq_relin_tail_overwriting), or takes them
// off again for the groups its exact fallback re-runs with the kernels that read raw residues.
struct LimbScaleArgs { const LimbTab *tabs; uint64_t *chat; const uint64_t *scale; const unsigned char *only; unsigned dim, logn; FlagScope scope; };
__device__ __forceinline__ void limb_scale_body(const LimbScaleArgs &a, unsigned poly, unsigned i) {     // a wave = one group of 64
  if (a.only && !__builtin_amdgcn_ballot_w64(a.only[((size_t)poly << a.logn) + i] != 0)) return;
  uint64_t *__restrict__ p = a.chat + ((size_t)poly * a.dim << a.logn) + i;
  for (unsigned d = 0; d < a.dim; ++d) {
    const PrimeK k = a.tabs[d].k;
    p[(size_t)d << a.logn] = mulmod_canon(p[(size_t)d << a.logn], a.scale[d], k);
  }
}
__global__ __launch_bounds__(256) void bridge_limb_scale(LimbScaleArgs a) {
  scoped_or_grid(a.scope, a.logn, 256, [&](unsigned poly, unsigned i) { limb_scale_body(a, poly, i); });
}
template <int WP>
__global__ __launch_bounds__(128) void bridge_fallback_tail_pre(ReconstructArgs dd, LimbScaleArgs un) {
  scoped_or_grid(dd.scope, dd.logn, 128, [&](unsigned poly, unsigned i) {
    reconstruct_body<WP>(dd, poly, i);                    // the addend d of the flagged coefficients as words (exact CRT of its limbs)
    limb_scale_body(un, poly, i);                         // the CRT weights off the key switch's limbs of the flagged groups, for the exact tail
  });
}

// ---------------------------------------------------------------------------
// he_genswk's combination (src/he-kem.c:86-99) before the reduction mod P q_L:  x = -t + e + P * sp  per coefficient, with
// t = swk.p1 * sk (already mod P q_L), e the error, sp the polynomial the key hides (s^2, or the rotated / conjugated s).
// Full width (WF = W + WPw + 1 words, two's complement); mpi_smod by P q_L follows (bridge_smod_general).
// Key-generation path: runtime word loops, per-thread arrays in scratch.
// ---------------------------------------------------------------------------
constexpr int GENSWK_MAXW = 48, GENSWK_MAXP = 32;

struct GenswkArgs {
  const uint64_t *t, *e, *sp;  // [W][n] each, two's complement
  const uint64_t *P;           // [WPw]
  uint64_t *x;                 // [WF][n]
  unsigned W, WPw, WF, logn;
};

__global__ __launch_bounds__(64) void bridge_genswk_combine(GenswkArgs a) {
  const unsigned i = blockIdx.x * 64 + threadIdx.x;
  if (i >= (1u << a.logn)) return;
  uint64_t acc[GENSWK_MAXW + GENSWK_MAXP + 2], m[GENSWK_MAXW];
  const unsigned W = a.W, WF = a.WF;
  // acc = e - t, sign-extended to WF words
  uint64_t borrow = 0;
  for (unsigned j = 0; j < W; ++j) {
    const u128 d = (u128)a.e[((size_t)j << a.logn) + i] - a.t[((size_t)j << a.logn) + i] - borrow;
    acc[j] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
  }
  const uint64_t es = (uint64_t)((int64_t)a.e[((size_t)(W - 1) << a.logn) + i] >> 63);
  const uint64_t ts = (uint64_t)((int64_t)a.t[((size_t)(W - 1) << a.logn) + i] >> 63);
  for (unsigned j = W; j < WF; ++j) {
    const u128 d = (u128)es - ts - borrow;
    acc[j] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
  }
  // |sp| and its sign
  const bool neg = a.sp[((size_t)(W - 1) << a.logn) + i] >> 63;
  uint64_t carry = neg;
  for (unsigned j = 0; j < W; ++j) {
    uint64_t w = a.sp[((size_t)j << a.logn) + i];
    if (neg) { w = ~w + carry; carry = carry && w == 0; }
    m[j] = w;
  }
  // acc +/-= |sp| * P, row by row
  for (unsigned j = 0; j < W; ++j) {
    if (!m[j]) continue;
    uint64_t c = 0, bw = 0;
    for (unsigned k = 0; j + k < WF; ++k) {
      const u128 pr = (u128)m[j] * (k < a.WPw ? a.P[k] : 0) + c;
      c = (uint64_t)(pr >> 64);
      if (neg) { const u128 d = (u128)acc[j + k] - (uint64_t)pr - bw; acc[j + k] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1; }
      else { const u128 d = (u128)acc[j + k] + (uint64_t)pr + bw; acc[j + k] = (uint64_t)d; bw = (uint64_t)(d >> 64); }
    }
  }
  for (unsigned j = 0; j < WF; ++j) a.x[((size_t)j << a.logn) + i] = acc[j];
}

// ---------------------------------------------------------------------------
// mpi_smod(x, M, floor(M/2)) for an arbitrary modulus M of L words (src/types.c:108-113): the general
// form of the second centring of poly_rns2mpi, needed when q is not a power of two (he_genswk calls
// poly_mul with q = P*q_L, src/he-kem.c:95).  Multiword Barrett (HAC 14.42, base 2^64) on |x| with
// mu = floor(2^(128 L) / M), then the sign and the centring.  Key-generation path: written for
// generality, not speed (runtime word loops, per-thread arrays in scratch).
// ---------------------------------------------------------------------------
constexpr int SMOD_MAXW = 96;     // x of at most 96 words, modulus of at most 48

struct SmodArgs {
  const uint64_t *x;       // [polys][Wx][n]  two's complement
  uint64_t *out;           // [polys][Wout][n]
  const uint64_t *M;       // [L]
  const uint64_t *mu;      // [L+1]
  const uint64_t *Mhalf;   // [L]  floor(M/2)
  unsigned Wx, Wout, L, logn;
};

__global__ __launch_bounds__(64) void bridge_smod_general(SmodArgs a) {
  const unsigned i = blockIdx.x * 64 + threadIdx.x;
  if (i >= (1u << a.logn)) return;
  const unsigned L = a.L, Wx = a.Wx;
  const uint64_t *__restrict__ xs = a.x + ((size_t)blockIdx.y * Wx << a.logn) + i;
  uint64_t ax[SMOD_MAXW + 2], v[SMOD_MAXW + 2], q2[SMOD_MAXW + 4], r[SMOD_MAXW / 2 + 3];
  // |x|
  const bool neg = xs[(size_t)(Wx - 1) << a.logn] >> 63;
  uint64_t carry = neg;
  for (unsigned j = 0; j < Wx; ++j) {
    uint64_t w = xs[(size_t)j << a.logn];
    if (neg) { w = ~w + carry; carry = carry && w == 0; }
    ax[j] = w;
  }
  // Horner in base b^L from the top block down: r <- (r * b^L + block) mod M, each step one Barrett
  // reduction of a 2L-word value (r < M < b^L, so the value is below b^(2L) as HAC 14.42 needs)
  for (unsigned j = 0; j <= L; ++j) r[j] = 0;
  const unsigned nblk = (Wx + L - 1) / L;
  for (int blk = (int)nblk - 1; blk >= 0; --blk) {
    for (unsigned j = 0; j < L; ++j) { const unsigned w = blk * L + j; v[j] = w < Wx ? ax[w] : 0; v[L + j] = r[j]; }
    v[2 * L] = 0;
    // q3 = floor( floor(v / b^(L-1)) * mu / b^(L+1) ),  q1 has L+1 words, mu has L+1 words
    for (unsigned j = 0; j < 2 * L + 3; ++j) q2[j] = 0;
    for (unsigned s = 0; s <= L; ++s) {
      const uint64_t q1 = v[L - 1 + s];
      uint64_t c = 0;
      for (unsigned t = 0; t <= L; ++t) {
        const u128 p = (u128)q1 * a.mu[t] + q2[s + t] + c;
        q2[s + t] = (uint64_t)p;
        c = (uint64_t)(p >> 64);
      }
      q2[s + L + 1] += c;
    }
    // r = (v - q3*M) mod b^(L+1);  q3 = q2[L+1 ..]
    for (unsigned j = 0; j <= L; ++j) r[j] = v[j];
    for (unsigned s = 0; s <= L; ++s) {
      const uint64_t q3 = q2[L + 1 + s];
      uint64_t c = 0, bor = 0;
      for (unsigned t = 0; s + t <= L; ++t) {
        const u128 p = (u128)q3 * (t < L ? a.M[t] : 0) + c;
        c = (uint64_t)(p >> 64);
        const u128 d = (u128)r[s + t] - (uint64_t)p - bor;
        r[s + t] = (uint64_t)d;
        bor = (uint64_t)(d >> 64) & 1;
      }
    }
    // at most two corrective subtractions (HAC 14.42 step 4)
    for (int rep = 0; rep < 3; ++rep) {
      uint64_t bor = 0;
      uint64_t t_[SMOD_MAXW / 2 + 3];
      for (unsigned j = 0; j <= L; ++j) {
        const u128 d = (u128)r[j] - (j < L ? a.M[j] : 0) - bor;
        t_[j] = (uint64_t)d;
        bor = (uint64_t)(d >> 64) & 1;
      }
      if (!bor) for (unsigned j = 0; j <= L; ++j) r[j] = t_[j];
    }
  }
  // sign: (-|x|) mod M = M - r unless r == 0
  if (neg) {
    uint64_t any = 0;
    for (unsigned j = 0; j <= L; ++j) any |= r[j];
    if (any) {
      uint64_t bor = 0;
      for (unsigned j = 0; j <= L; ++j) {
        const u128 d = (u128)(j < L ? a.M[j] : 0) - r[j] - bor;
        r[j] = (uint64_t)d;
        bor = (uint64_t)(d >> 64) & 1;
      }
    }
  }
  // centre: r >= floor(M/2) -> r - M
  {
    uint64_t bor = 0;
    for (unsigned j = 0; j <= L; ++j) {
      const u128 d = (u128)r[j] - (j < L ? a.Mhalf[j] : 0) - bor;
      bor = (uint64_t)(d >> 64) & 1;
    }
    if (!bor) {
      uint64_t b2 = 0;
      for (unsigned j = 0; j <= L; ++j) {
        const u128 d = (u128)r[j] - (j < L ? a.M[j] : 0) - b2;
        r[j] = (uint64_t)d;
        b2 = (uint64_t)(d >> 64) & 1;
      }
    }
  }
  uint64_t *__restrict__ dst = a.out + ((size_t)blockIdx.y * a.Wout << a.logn) + i;
  const uint64_t fill = (uint64_t)((int64_t)r[L] >> 63);
  for (unsigned j = 0; j < a.Wout; ++j) dst[(size_t)j << a.logn] = j <= L ? r[j] : fill;
}

// ---------------------------------------------------------------------------
// poly_rot / poly_conj (src/poly.c:263-283) on big slabs: signed permutations of the coefficients.
//   rot : r[k] = a[i] (k < n) or r[k-n] = -a[i],  k = i * 5^rot mod 2n
//   conj: r[0] = a[0], r[i] = -a[n-i]
// One thread per source coefficient; negation is two's complement over the W words.
// ---------------------------------------------------------------------------
struct PermuteArgs { const uint64_t *a; uint64_t *r; unsigned W, logn; unsigned long long power; int conj; };

__global__ __launch_bounds__(256) void bridge_permute(PermuteArgs p) {
  const unsigned n = 1u << p.logn;
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned dst;
  bool neg;
  if (p.conj) { dst = i ? n - i : 0; neg = i != 0; }
  else {
    const unsigned k = (unsigned)(((unsigned long long)i * p.power) & (2ull * n - 1));
    neg = k >= n;
    dst = neg ? k - n : k;
  }
  const size_t base = (size_t)blockIdx.y * p.W << p.logn;
  uint64_t carry = neg;
  for (unsigned j = 0; j < p.W; ++j) {
    uint64_t w = p.a[base + ((size_t)j << p.logn) + i];
    if (neg) { w = ~w + carry; carry = carry && w == 0; }
    p.r[base + ((size_t)j << p.logn) + dst] = w;
  }
}

// ---------------------------------------------------------------------------
// mpi_rdiv(x, m) for a one-word divisor m (hectx.p = Delta is a uint64_t, src/gpqhe.h:100, src/precomp.c:448):
// floor(x/m), plus one when the remainder is strictly above floor(m/2) (src/types.c:115-128).  Long division of
// |x| by words from the top; floor semantics for negative x restored afterwards.  In place allowed.
// ---------------------------------------------------------------------------
struct RdivWordArgs { const uint64_t *x; uint64_t *out; unsigned W, logn; unsigned long long m; };

__global__ __launch_bounds__(256) void bridge_rdiv_word(RdivWordArgs a) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (1u << a.logn)) return;
  const size_t base = ((size_t)blockIdx.y * a.W << a.logn) + i;
  const bool neg = a.x[base + ((size_t)(a.W - 1) << a.logn)] >> 63;
  // magnitude, most significant word first; the negation carry needs the low words first, so two sweeps
  uint64_t carry = neg;
  for (unsigned j = 0; j < a.W; ++j) {
    uint64_t w = a.x[base + ((size_t)j << a.logn)];
    if (neg) { w = ~w + carry; carry = carry && w == 0; }
    a.out[base + ((size_t)j << a.logn)] = w;
  }
  uint64_t rem = 0;
  for (int j = (int)a.W - 1; j >= 0; --j) {
    const u128 cur = ((u128)rem << 64) | a.out[base + ((size_t)j << a.logn)];
    a.out[base + ((size_t)j << a.logn)] = (uint64_t)(cur / a.m);
    rem = (uint64_t)(cur % a.m);
  }
  // x = s*(Q*m + rem): floor and remainder of the signed value, then the rounding rule
  uint64_t add = 0;                 // added to the magnitude Q before restoring the sign
  bool roundup;
  if (!neg) roundup = rem > a.m / 2;
  else if (rem == 0) roundup = false;
  else { add = 1; roundup = (a.m - rem) > a.m / 2; }           // floor = -(Q+1), remainder m - rem
  // result = neg ? -(Q + add) + roundup : Q + roundup
  uint64_t c1 = neg ? add : (roundup ? 1 : 0);
  for (unsigned j = 0; j < a.W && c1; ++j) {
    uint64_t &w = a.out[base + ((size_t)j << a.logn)];
    w += c1; c1 = w == 0;
  }
  if (neg) {
    uint64_t c = 1;
    for (unsigned j = 0; j < a.W; ++j) { uint64_t &w = a.out[base + ((size_t)j << a.logn)]; w = ~w + c; c = c && w == 0; }
    uint64_t c2 = roundup ? 1 : 0;
    for (unsigned j = 0; j < a.W && c2; ++j) { uint64_t &w = a.out[base + ((size_t)j << a.logn)]; w += c2; c2 = w == 0; }
  }
}

// General-modulus tail of he_relin: full = Q + [r > floor(P/2)] + d (+ Pi' on the wrap corner), WF words, signed.
struct AddRoundFullArgs {
  uint64_t *full;              // [polys][WF][n]
  const uint64_t *q;           // [polys][WQ][n]   centred quotient, full width
  const uint64_t *r;           // [polys][Wr][n]
  const uint64_t *d;           // [polys][W][n] or null
  const uint64_t *phalf;       // [Wr]
  const uint64_t *piq;         // [WQ]  Pi'
  const unsigned char *tie;
  unsigned WF, WQ, Wr, W, logn;
};

__global__ __launch_bounds__(256) void bridge_addround_full(AddRoundFullArgs a) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (1u << a.logn)) return;
  const uint64_t *__restrict__ r = a.r + ((size_t)blockIdx.y * a.Wr << a.logn) + i;
  int cmp = 0;
  for (int j = (int)a.Wr - 1; j >= 0 && cmp == 0; --j) {
    const uint64_t rv = r[(size_t)j << a.logn], hv = a.phalf[j];
    cmp = rv > hv ? 1 : (rv < hv ? -1 : 0);
  }
  const bool fix = a.tie[((size_t)blockIdx.y << a.logn) + i] && cmp < 0;
  const uint64_t *__restrict__ q = a.q + ((size_t)blockIdx.y * a.WQ << a.logn) + i;
  const uint64_t *__restrict__ d = a.d ? a.d + ((size_t)blockIdx.y * a.W << a.logn) + i : nullptr;
  const uint64_t qs = (uint64_t)((int64_t)q[(size_t)(a.WQ - 1) << a.logn] >> 63);
  const uint64_t ds = d ? (uint64_t)((int64_t)d[(size_t)(a.W - 1) << a.logn] >> 63) : 0;
  uint64_t carry = cmp > 0;
  for (unsigned j = 0; j < a.WF; ++j) {
    const uint64_t qv = j < a.WQ ? q[(size_t)j << a.logn] : qs;
    const uint64_t dv = d ? (j < a.W ? d[(size_t)j << a.logn] : ds) : 0;
    const uint64_t pv = fix && j < a.WQ ? a.piq[j] : 0;
    const u128 t = (u128)qv + dv + pv + carry;
    a.full[(((size_t)blockIdx.y * a.WF + j) << a.logn) + i] = (uint64_t)t;
    carry = (uint64_t)(t >> 64);
  }
}

// ---------------------------------------------------------------------------
// he_add / he_sub / he_neg on big slabs (src/he-add.c:32-142): mpi_addm / mpi_subm then mpi_smod, q_l = 2^k.
// mode 0: a + b, 1: a - b, 2: -a.  Not on the NTT path; here so that a ciphertext can stay in HBM between
// multiplications.
// ---------------------------------------------------------------------------
struct AddSubArgs { uint64_t *r; const uint64_t *a; const uint64_t *b; unsigned W, logn, logql, mode; };

__global__ __launch_bounds__(256) void bridge_addsub(AddSubArgs p) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (1u << p.logn)) return;
  const size_t base = ((size_t)blockIdx.y * p.W << p.logn) + i;
  const unsigned sb = p.logql - 1;
  uint64_t carry = p.mode ? 1 : 0, qsign = 0;
  for (unsigned j = 0; j < p.W; ++j) {
    const size_t o = base + ((size_t)j << p.logn);
    const uint64_t x = p.mode == 2 ? 0 : p.a[o];
    const uint64_t y = p.mode == 0 ? p.b[o] : ~(p.mode == 2 ? p.a[o] : p.b[o]);   // x - y = x + ~y + 1
    const u128 t = (u128)x + y + carry;
    uint64_t v = (uint64_t)t;
    carry = (uint64_t)(t >> 64);
    const unsigned lo = 64 * j;
    if (lo + 64 > sb && lo <= sb) qsign = 0 - ((v >> (sb - lo)) & 1);
    if (lo >= p.logql) v = qsign;
    else if (lo + 64 > p.logql) {
      const uint64_t mask = (1ull << (p.logql - lo)) - 1;
      v = (v & mask) | (qsign & ~mask);
    }
    p.r[o] = v;
  }
}

// ---------------------------------------------------------------------------
// he_rs on one big slab, Delta = 2^s and q_l = 2^logql (the reference's test
// parameters, tests/gpqhe.c:1349-1352): c <- smod(rdiv(c, Delta), q_l), in place.
//   rdiv: floor(c / 2^s) = arithmetic shift; plus one when (c mod 2^s) > 2^(s-1)
//   smod: keep logql bits, sign-extend from bit logql-1
// One thread per coefficient, words walked in ascending order (reads run ahead
// of writes, so in place is safe).
// ---------------------------------------------------------------------------
struct RescaleArgs { uint64_t *big; unsigned W, logn, s, logql; };

// one coefficient in place: c = word 0 of the coefficient inside its big slab [W][n]
__device__ __forceinline__ void rescale_coefficient(uint64_t *__restrict__ c, unsigned W, unsigned logn, unsigned s, unsigned logql) {
  const unsigned ws = s >> 6, bs = s & 63;
  auto word = [&](unsigned j) -> uint64_t {               // sign-extended read
    return j < W ? c[(size_t)j << logn] : (uint64_t)((int64_t)c[(size_t)(W - 1) << logn] >> 63);
  };
  // remainder r = c mod 2^s  >  2^(s-1)  <=>  bit s-1 set and some lower bit set
  uint64_t carry = 0;
  if (s) {
    const unsigned hb = s - 1, hw = hb >> 6, hbit = hb & 63;
    const uint64_t wh = word(hw);
    uint64_t lower = wh & ((1ull << hbit) - 1);
    for (unsigned j = 0; j < hw; ++j) lower |= word(j);
    carry = ((wh >> hbit) & 1) && lower;
  }
  const unsigned sb = logql - 1;
  uint64_t qsign = 0;
  for (unsigned j = 0; j < W; ++j) {
    const uint64_t lo = word(j + ws), hi = word(j + ws + 1);
    uint64_t v = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
    const uint64_t v1 = v + carry;
    carry = v1 < v;
    v = v1;
    const unsigned base = 64 * j;
    if (base + 64 > sb && base <= sb) qsign = 0 - ((v >> (sb - base)) & 1);
    if (base >= logql) v = qsign;
    else if (base + 64 > logql) {
      const uint64_t mask = (1ull << (logql - base)) - 1;
      v = (v & mask) | (qsign & ~mask);
    }
    c[(size_t)j << logn] = v;
  }
}

__global__ __launch_bounds__(256) void bridge_rescale(RescaleArgs a) {
  const unsigned n = 1u << a.logn;
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  rescale_coefficient(a.big + ((size_t)blockIdx.y * a.W << a.logn) + i, a.W, a.logn, a.s, a.logql);
}

// The same for the coefficients a streaming tail flagged (gpq_he_mul_rs: the tail kernel rescales what it decides itself in registers; the exact
// kernels behind it write the flagged coefficients unrescaled, and this finishes them).
struct RescaleMaskedArgs { Two<uint64_t> out; const unsigned char *only; unsigned W, logn, s, logql; FlagScope scope; };
__global__ __launch_bounds__(256) void bridge_rescale_masked(RescaleMaskedArgs a) {
  scoped_or_grid(a.scope, a.logn, 256, [&](unsigned poly, unsigned i) {
    if (!a.only[((size_t)poly << a.logn) + i]) return;
    rescale_coefficient(a.out.at(poly, (size_t)a.W << a.logn) + i, a.W, a.logn, a.s, a.logql);
  });
}

}  // namespace gpq
