// bridge_mfma.hpp -- the two GEMM-shaped steps of the MPI <-> RNS bridge on the matrix cores.
//
// rns_decompose (src/rns.c:37-48) and the CRT sum of rns_reconstruct (src/rns.c:60-75) are products of
// a per-coefficient digit vector with a FIXED matrix of constants:
//   decompose    x mod p_j = sum_k byte_k(x) * (256^k mod p_j)                     (k < 8W, j < dim)
//   reconstruct  S mod 2^(64 WL) = sum_d y_d * (P/p_d mod 2^(64 WL))               (d < dim)
// i.e. dense integer contractions over 10^6 coefficients per launch -- the one place on this path where
// the work IS a GEMM (the NTT is not: north_star).  Both sides are cut into signed bytes and multiplied
// with v_mfma_i32_32x32x32_i8 (signed x signed, exact i32 accumulation; lane maps checked with exact
// data by tools/mfma_i8_probe.hip: lane l = (r = l&31, h = l>>5) holds A[r][16h+t] / B[16h+t][r],
// C/D col = l&31, row = (g&3) + 8(g>>2) + 4h).  The VALU keeps only the recombination of the byte
// columns and the final modular reduction.  Results are bit-identical with the VALU kernels of
// bridge_kernels.hpp (tests/test_bridge_gpu.py runs both).
//
// Signed bytes.  A variable byte u in 0..255 enters as s = u - 128 = (int8)(u ^ 0x80); the +128 of
// every byte position is a per-prime constant added after the contraction.  The constant side is
// recoded on the host into balanced digits in [-128, 127].
#pragma once
#include "modarith.hpp"
#include "tables.hpp"
#include "bridge_kernels.hpp"

#ifndef GPQ_DECOMP_PIPE
#define GPQ_DECOMP_PIPE 0      /* bridge_decompose_mfma: row tiles software-pipelined by one -- measured 9 % SLOWER at three waves per SIMD (profiles/r04/v3_decompose_pipe_ab.txt) */
#endif
#ifndef GPQ_FRONT_XG
#define GPQ_FRONT_XG 8   /* k steps from which the relinearisation front stops fetching the next group early */
#endif
namespace gpq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Orientation.  The CONSTANT matrix is the A operand (rows = output byte columns) and the coefficients are the
// B operand (columns = coefficients): D[row][col] puts a coefficient on the lane (col = l & 31) and the byte
// columns in the registers, rows (g&3) + 8(g>>2) + 4h.  Registers 4w..4w+3 of lane half h are byte columns
// 8w+4h .. 8w+4h+3: the low half of 64-bit word w (or of prime w's eight digits) sits in the lower 32 lanes,
// the high half in the upper 32.  With two coefficient tiles per wave one v_permlane32_swap per register pair
// hands every lane both halves of ITS coefficient (lower lanes finish tile 0, upper lanes tile 1): the
// recombination needs no LDS and every lane stays busy.  LDS only holds the constant fragments.

// four signed byte-column sums -> their value sum_b c_b 256^b (46 bits)
__device__ __forceinline__ int64_t horner4(int c0, int c1, int c2, int c3) {
  return (int64_t)(c3 * 256 + c2) * 65536 + (c1 * 256 + c0);
}

// Lane half h holds p0 = half-word of tile 0, p1 = of tile 1 (low halves in lanes 0-31, high halves in 32-63).
// Afterwards every lane has L and H of the coefficient it finishes (tile = its half).
__device__ __forceinline__ void swap_halves(int64_t p0, int64_t p1, int64_t &L, int64_t &H) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)(uint64_t)p0, (unsigned)(uint64_t)p1, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)((uint64_t)p0 >> 32), (unsigned)((uint64_t)p1 >> 32), false, false);
  L = (int64_t)(((uint64_t)hi[0] << 32) | lo[0]);
  H = (int64_t)(((uint64_t)hi[1] << 32) | lo[1]);
}

// L + H 2^32 + carry in -> one 64-bit word + signed carry out
__device__ __forceinline__ uint64_t fold_word(int64_t L, int64_t H, int64_t &carry) {
  const int64_t T = carry + L;
  const uint64_t lo = (uint64_t)T + ((uint64_t)H << 32);
  carry = (H >> 32) + (T >> 63) + (lo < (uint64_t)T ? 1 : 0);
  return lo;
}

// ---------------------------------------------------------------------------
// rns_decompose on the matrix cores.
//   constants  T[8j + b][k] = balanced digit b of 256^k mod p_j            (A operand, rows = prime j's eight digits)
//   data       X[k][m]      = byte k of coefficient m, as s = u - 128 (top byte of the top word: signed as it is)
//   D[8j + b][m] = sum_k T X        |D| <= 8W * 2^14
//   x mod p_j  = sum_b D[8j+b] 256^b + K_j,     K_j = 128 * sum_{k < 8W-1} 256^k mod p_j
// One wave owns 64 consecutive coefficients (two 32-column tiles); the constant fragments sit in LDS.
// Epilogue per (coefficient, prime): L = D3..D0 and H = D7..D4 by Horner (46 bits each), exchanged between the lane
// halves,  V = H 2^32 + L,  H = Hh 2^27 + Hl  =>  V == Hl 2^32 + L - c Hh  (mod p),  + Kq_j, in (0, 3p) -> canonical.
// ---------------------------------------------------------------------------
struct DecomposeMfmaArgs {
  BigSources big;            // [polys][W][n]
  uint64_t *slab;            // [polys][dim][n]
  const v4i *bfrag;          // [NT][KS][64]: the constant (A) fragment of lane l for (row tile, k step)
  const uint64_t *pk;        // [4 NT][3]: p_j, Kq_j = 2^50 + ((K_j - 2^50) mod p_j), c_j   (zeros for padding primes)
  unsigned W, dim, logn, NT;               // NT: row tiles of 4 primes
  unsigned groups_per_poly, total_groups;  // groups of 64 coefficients
  unsigned lazy;                           // residues out in (0, 3p) -- what the forward transform behind gpq_he_mul's decompositions accepts -- not canonical
};

template <int KS>
__global__ __launch_bounds__(256, 2) void bridge_decompose_mfma(DecomposeMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4i *Bl = reinterpret_cast<v4i *>(smem);
  const unsigned nB = a.NT * KS * 64;
  uint64_t *pkl = reinterpret_cast<uint64_t *>(smem + (size_t)nB * 16);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (unsigned i = threadIdx.x; i < nB; i += 256) Bl[i] = a.bfrag[i];
  for (unsigned i = threadIdx.x; i < a.NT * 12; i += 256) pkl[i] = a.pk[i];
  __syncthreads();
  const unsigned r = lane & 31, h = lane >> 5;
  v4i X[2][KS];
  // the data fragments of a group of 64 coefficients: lane (r, h) holds words 4s+2h, 4s+2h+1 of coefficients r and 32+r
  auto load_X = [&](unsigned g) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    const uint64_t *__restrict__ src = a.big.at(poly, (size_t)a.W << a.logn) + coef0 + r;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        uint64_t x[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          // unconditional load from a clamped word index, zero selected afterwards: a load under a branch
          // is followed by its own vmcnt(0) and the 16 loads of a group would run one after the other
          const unsigned w = 4 * s + 2 * h + e, wc = w < a.W ? w : a.W - 1;
          const uint64_t m = w + 1 < a.W ? 0x8080808080808080ull : 0x0080808080808080ull;
          const uint64_t v = __builtin_nontemporal_load(&src[((size_t)wc << a.logn) + 32 * t]);
          x[e] = (v ^ m) & (w < a.W ? ~0ull : 0ull);
        }
        X[t][s] = v4i{(int)(uint32_t)x[0], (int)(uint32_t)(x[0] >> 32), (int)(uint32_t)x[1], (int)(uint32_t)(x[1] >> 32)};
      }
  };
  const unsigned g0 = blockIdx.x * 4 + wave, gstep = gridDim.x * 4;
  if (g0 < a.total_groups) load_X(g0);
  for (unsigned g = g0; g < a.total_groups; g += gstep) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    uint64_t *__restrict__ dst = a.slab + ((size_t)poly * a.dim << a.logn) + coef0 + lane;   // this lane finishes coefficient coef0 + lane
    if (KS >= 8 && g != g0) load_X(g);                 // 32-word inputs: no registers left for the early fetch
    // (GPQ_DECOMP_PIPE = 1: the MFMAs of tile q + 1 issued before the integer epilogue of tile q.  With three waves per SIMD the other waves
    // already cover a wave's wait for its own products; the second accumulator pair and the barriers cost more than they return.)
    auto product = [&](unsigned q, v16i &acc0, v16i &acc1) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const v4i cf = Bl[(q * KS + s) * 64 + lane];
        if (s == 0) {
          v16i z;
#pragma unroll
          for (int e = 0; e < 16; ++e) z[e] = 0;
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[0][s], z, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[1][s], z, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[0][s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[1][s], acc1, 0, 0, 0);
        }
      }
      // the fragments are dead once the last row tile has been issued: fetch the next group's under the remaining epilogues
      if (KS < 8 && q + 1 == a.NT && g + gstep < a.total_groups) load_X(g + gstep);
    };
    auto finish = [&](unsigned q, const v16i &acc0, const v16i &acc1) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {                    // prime j = 4q + w: digits 0-3 in the lower lanes, 4-7 in the upper
        int64_t L, H;
        swap_halves(horner4(acc0[4 * w], acc0[4 * w + 1], acc0[4 * w + 2], acc0[4 * w + 3]),
                    horner4(acc1[4 * w], acc1[4 * w + 1], acc1[4 * w + 2], acc1[4 * w + 3]), L, H);
        const unsigned j = 4 * q + w;
        const uint64_t p = pkl[3 * j], kq = pkl[3 * j + 1];
        const int c = (int)(uint32_t)pkl[3 * j + 2];
        const int Hh = (int)(H >> 27);
        const uint64_t Hl = (uint64_t)H & 0x7ffffffu;
        uint64_t v = (Hl << 32) + (uint64_t)L + kq;
        v = (uint64_t)((int64_t)(-c) * Hh + (int64_t)v);               // in (0, 3p)
        if (j < a.dim) __builtin_nontemporal_store(a.lazy ? v : canon_fold(v, p, (uint32_t)c), &dst[(size_t)j << a.logn]);
      }
    };
#if GPQ_DECOMP_PIPE
    v16i pa0, pa1, pb0, pb1;
    product(0, pa0, pa1);
    for (unsigned q = 0; q < a.NT; q += 2) {
      if (q + 1 < a.NT) product(q + 1, pb0, pb1);
      __builtin_amdgcn_sched_barrier(0);
      finish(q, pa0, pa1);
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < a.NT) {
        if (q + 2 < a.NT) product(q + 2, pa0, pa1);
        __builtin_amdgcn_sched_barrier(0);
        finish(q + 1, pb0, pb1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#else
    for (unsigned q = 0; q < a.NT; ++q) {
      v16i acc0, acc1;
      product(q, acc0, acc1);
      finish(q, acc0, acc1);
    }
#endif
  }
}

// ---------------------------------------------------------------------------
// poly_rns2mpi fast path (q = 2^logq, centred; see bridge_reconstruct_low) on the matrix cores.
//   y_d = ahat_d * phat_invmp_d mod p_d                       VALU, by the lane that owns (row, limb) in the A layout
//   data       X[8d + i][m] = byte i of y_d of coefficient m, as s = u - 128
//   constants  T[n][8d + i] = balanced digit n - i of phat_d mod 2^(64 WL)         rows 0 .. 8WL-1   (S)
//                             balanced digit m - i of floor(2^104 / p_d)           rows 8WL + m, m < 14   (F)
//   S = sum_n D[n] 256^n + Kc  (mod 2^(64 WL)),   F = sum_m D[8WL+m] 256^m + Kf
//   F underestimates 2^104 S/P by less than dim 2^60 < 2^66:  k = F >> 104, centred <=> bit 103,
//   coefficients with frac(F) in [1/2 - 2^-38, 1/2) are flagged for the exact kernel (as bridge_reconstruct_low does
//   with its 2^-61 window); result = S - (k + centred) P, masked and sign-extended at logq.
// K-outer loop: the accumulators of all row tiles stay in registers (2 coefficient tiles x NT x 16), the data
// fragments of one k step (4 limbs) are made on the fly.  Epilogue: the half-words are exchanged between the
// lane halves (swap_halves), then every lane folds the words of its coefficient with a running signed carry.
// ---------------------------------------------------------------------------
struct ReconMfmaArgs {
  const uint64_t *slab;      // [polys][slab_dim][n]
  Two<uint64_t> big;         // [polys][Wout][n]
  const v4i *bfrag;          // [KS][NT][64]
  const uint64_t *lk;        // [4 KS][2]: p_d, phat_invmp_d   (p = 0: padding limb)
  const uint64_t *kc;        // [WL + 2]: (unused: Kc lives in pm), then Kf (2 words)
  const uint64_t *pm;        // [65][WL]: (m * P - Kc) mod 2^(64 WL), Kc = the offset of the signed bytes of S
  unsigned char *redo;       // [polys][n]
  unsigned char *tie;        // optional, cleared
  unsigned dim, KS, logn, Wout, logq, slab_dim, slab_first;
  unsigned groups_per_poly, total_groups;
  // tail of he_relin / he_swk fused in (bridge_addround's work for the coefficients decided here):
  Two<const uint64_t> addend;    // optional [polys][Wout][n]: d of src/he-mult.c:72-76
  const unsigned char *rflags;   // optional [polys][n]: RF_GT = round the quotient up (mpi_rdiv)
  unsigned prescaled;            // the slab already holds y_d = ahat_d * phat_invmp_d (bridge_relin_front_mfma writes it so)
  // The whole relinearisation tail as ONE product (WL = 16 only; bridge.hip: tail_direct): with frac_bits = 104 the constant matrix holds
  // floor(Pi' 2^104 / p_d) for ALL limbs of the key switch's basis, so the columns are 2^104 x / P as a fixed-point number -- the low
  // 104 bits its fraction (x mod P) / P, the rest floor(x / P): the quotient and the rounding decision of mpi_rdiv come out of the same
  // contraction that centres x, without the residues of r = x mod P, without the exact division, without a second CRT.
  unsigned frac_bits;
  unsigned char *amb_clear;      // optional [polys][n]: cleared (the exact fallback's `amb` array must be 0 outside the groups it re-runs)
};

// per-coefficient flags of the relinearisation tail: r = x mod P against floor(P/2)
constexpr unsigned char RF_GT = 1, RF_LT = 2, RF_AMB = 4;

// k steps of residues a wave keeps in flight.  The kernel is bound by bytes in flight, not by arithmetic (two waves per SIMD at its register
// count: 8 waves x AHEAD x 2 KB per CU against latency x bandwidth): every further step is 8 VGPRs.
#ifndef GPQ_RECON_AHEAD
#define GPQ_RECON_AHEAD 2        /* WL <= 14 */
#endif
#ifndef GPQ_RECON_AHEAD16
#define GPQ_RECON_AHEAD16 2      /* WL = 16 (the one-product tail) */
#endif
template <int WL>
__global__ __launch_bounds__(512) void bridge_reconstruct_low_mfma(ReconMfmaArgs a) {
  constexpr int NT = (8 * WL + 14 + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4i *Bl = reinterpret_cast<v4i *>(smem);
  const unsigned nB = a.KS * NT * 64;
  uint64_t *lkl = reinterpret_cast<uint64_t *>(smem + (size_t)nB * 16);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (unsigned i = threadIdx.x; i < nB; i += 512) Bl[i] = a.bfrag[i];
  for (unsigned i = threadIdx.x; i < a.KS * 8; i += 512) lkl[i] = a.lk[i];
  __syncthreads();
  const unsigned r = lane & 31, h = lane >> 5;
  const unsigned g0 = blockIdx.x * 8 + wave, gstep = gridDim.x * 8;
  // residues of k step s of group g: limbs 4s+2h, 4s+2h+1 of coefficients r (tile 0) and 32+r (tile 1); padding limbs and
  // steps past the end read the last limb.  AH steps are kept in flight (xq), across the group boundary too where the registers allow (CROSS).
  constexpr int AH = WL <= 14 ? GPQ_RECON_AHEAD : GPQ_RECON_AHEAD16;
  uint64_t xq[AH][4];
  auto fetch = [&](unsigned g, unsigned s, uint64_t (&xn)[4]) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    const uint64_t *__restrict__ src = a.slab + (((size_t)poly * a.slab_dim + a.slab_first) << a.logn) + coef0 + r;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned d = 4 * s + 2 * h + e, dc = d < a.dim ? d : a.dim - 1;
      xn[e] = src[(size_t)dc << a.logn];
      xn[2 + e] = src[((size_t)dc << a.logn) + 32];
    }
  };
  constexpr bool CROSS = WL <= 10;      // fetch the next group's first steps under this group's epilogue (registers allowing)
  if (CROSS && g0 < a.total_groups) {
#pragma unroll
    for (int j = 0; j < AH; ++j) fetch(g0, j, xq[j]);
  }
  for (unsigned g = g0; g < a.total_groups; g += gstep) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    const unsigned gn = g + gstep < a.total_groups ? g + gstep : g;          // next group (or this one again: harmless reads)
    if (!CROSS) {
#pragma unroll
      for (int j = 0; j < AH; ++j) fetch(g, j, xq[j]);
    }
    v16i acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][q][e] = 0;
    auto step = [&](unsigned s, const uint64_t (&x)[4]) {
      uint64_t y[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned d = 4 * s + 2 * h + e;
        const uint64_t p = lkl[2 * d], w = lkl[2 * d + 1];
        PrimeK k;
        k.p = p; k.p2 = p << 1; k.c = (uint32_t)p; k.c1 = k.c + 1;   // p = 2^59 + c: c is the low word
        const uint64_t m = p ? ~0ull : 0ull;
        y[e] = ((a.prescaled ? x[e] : canon_fold(mulmod_lazy(x[e], w, k), p, k.c)) ^ 0x8080808080808080ull) & m;
        y[2 + e] = ((a.prescaled ? x[2 + e] : canon_fold(mulmod_lazy(x[2 + e], w, k), p, k.c)) ^ 0x8080808080808080ull) & m;
      }
      const v4i A0 = v4i{(int)(uint32_t)y[0], (int)(uint32_t)(y[0] >> 32), (int)(uint32_t)y[1], (int)(uint32_t)(y[1] >> 32)};
      const v4i A1 = v4i{(int)(uint32_t)y[2], (int)(uint32_t)(y[2] >> 32), (int)(uint32_t)y[3], (int)(uint32_t)(y[3] >> 32)};
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const v4i b = Bl[(s * NT + q) * 64 + lane];
        acc[0][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, A0, acc[0][q], 0, 0, 0);
        acc[1][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, A1, acc[1][q], 0, 0, 0);
      }
    };
    for (unsigned s = 0; s < a.KS; s += AH) {
#pragma unroll
      for (int j = 0; j < AH; ++j) {
        if (s + j < a.KS) {
          uint64_t x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = xq[j][e];
          if (s + j + AH < a.KS) fetch(g, s + j + AH, xq[j]); else if (CROSS) fetch(gn, j, xq[j]);     // (slot j opens the next group with its step j)
          step(s + j, x);
        }
      }
    }
    // epilogue: this lane finishes coefficient coef0 + lane (tile h, column r)
    constexpr bool EARLY_D = WL <= 14;                     // (no registers left for it at WL = 16)
    uint64_t dd[WL];                                       // d of the fused tail, fetched under the column folding
    const uint64_t *addend = a.rflags ? a.addend.at(poly, (size_t)a.Wout << a.logn) : nullptr;   // uniform per group
    if (EARLY_D && addend) {
      const uint64_t *__restrict__ dp = addend + coef0 + lane;
#pragma unroll
      for (int j = 0; j < WL; ++j) dd[j] = dp[(size_t)(j < (int)a.Wout ? j : 0) << a.logn];
    }
    uint64_t V[4 * NT];
    int64_t carry = 0;
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        int64_t L, H;
        swap_halves(horner4(acc[0][q][4 * w], acc[0][q][4 * w + 1], acc[0][q][4 * w + 2], acc[0][q][4 * w + 3]),
                    horner4(acc[1][q][4 * w], acc[1][q][4 * w + 1], acc[1][q][4 * w + 2], acc[1][q][4 * w + 3]), L, H);
        if (4 * q + w == WL) carry = 0;                  // the F columns start their own carry chain
        V[4 * q + w] = fold_word(L, H, carry);
      }
    // (the offset of the signed bytes of S is folded into the table of multiples of P; F's is added here)
    const u128 F = (((u128)V[WL + 1] << 64) | V[WL]) + (((u128)a.kc[WL + 1] << 64) | a.kc[WL]);
    const uint64_t f1 = (uint64_t)(F >> 64);
    const size_t flag_at = ((size_t)poly << a.logn) + coef0 + lane;
    bool ambiguous = ((f1 >> 2) & ((1ull << 38) - 1)) == ((1ull << 37) - 1);   // frac in [1/2 - 2^-38, 1/2)
    if (a.tie) a.tie[flag_at] = 0;
    if (a.amb_clear) a.amb_clear[flag_at] = 0;
    const unsigned mult = (unsigned)(f1 >> 40) + (unsigned)((f1 >> 39) & 1);       // k, plus one when centring takes P off once more
    if (!ambiguous) {
      const uint64_t *__restrict__ P = a.pm + (size_t)mult * WL;
      uint64_t borrow = 0;
#pragma unroll
      for (int j = 0; j < WL; ++j) {
        const u128 t = (u128)V[j] - P[j] - borrow;
        V[j] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
      }
    }
    if constexpr (WL == 16) {
      if (a.frac_bits && !ambiguous) {
        // V = 2^104 x / P (two's complement, 16 words), underestimated by less than 2^-40 (the constants of the limbs of P are floors, y_d < 2^60):
        //   fraction f = bits 0..103 = (x mod P) / P;  f >= 1 - 2^-38: the estimate may have borrowed from the integer part -> exact path;
        //   mpi_rdiv rounds up iff (x mod P) > floor(P/2) iff the true fraction > 1/2 (P is odd):  f >= 1/2 -> up;  f < 1/2 - 2^-38 -> down;  between -> exact path
        const uint64_t f40 = V[1] & ((1ull << 40) - 1), top38 = f40 >> 2;
        ambiguous = top38 == ((1ull << 38) - 1) || top38 == ((1ull << 37) - 1);
        if (!ambiguous) {
          const uint64_t up_bit = f40 >> 39;
          uint64_t Q[14];
#pragma unroll
          for (int j = 0; j < 14; ++j) Q[j] = (V[j + 1] >> 40) | (V[j + 2] << 24);     // floor(x / P) mod 2^896
          const uint64_t *addend = a.addend.at(poly, (size_t)a.Wout << a.logn);
          uint64_t cr = up_bit;
#pragma unroll
          for (int j = 0; j < 14; ++j) {
            const uint64_t dj = (addend && j < (int)a.Wout) ? (addend + coef0 + lane)[(size_t)j << a.logn] : 0;
            const u128 t = (u128)Q[j] + cr + dj;
            Q[j] = (uint64_t)t; cr = (uint64_t)(t >> 64);
          }
          uint64_t *__restrict__ dst = a.big.at(poly, (size_t)a.Wout << a.logn) + coef0 + lane;
          const int sw = (int)((a.logq - 1) >> 6);
          const unsigned up = 63 - ((a.logq - 1) & 63);
          uint64_t ext = 0;
#pragma unroll
          for (int j = 0; j < 14; ++j) if (j == sw) ext = (uint64_t)((int64_t)(Q[j] << up) >> up);
          const uint64_t qsign = (uint64_t)((int64_t)ext >> 63);
#pragma unroll
          for (int j = 0; j < 14; ++j)
            if (j < (int)a.Wout) dst[(size_t)j << a.logn] = j < sw ? Q[j] : (j == sw ? ext : qsign);
          for (unsigned j = 14; j < a.Wout; ++j) dst[(size_t)j << a.logn] = qsign;
        }
      }
    }
    a.redo[flag_at] = ambiguous;
    if (!ambiguous && !(WL == 16 && a.frac_bits)) {
      if (a.rflags) {                                    // + [r > floor(P/2)] + d   (mod 2^logq: only the low words matter)
        if (!EARLY_D && addend) {
          const uint64_t *__restrict__ dp = addend + coef0 + lane;
#pragma unroll
          for (int j = 0; j < WL; ++j) dd[j] = dp[(size_t)(j < (int)a.Wout ? j : 0) << a.logn];
        }
        uint64_t cr = a.rflags[flag_at] & RF_GT;
#pragma unroll
        for (int j = 0; j < WL; ++j) {
          const u128 t = (u128)V[j] + cr + ((addend && j < (int)a.Wout) ? dd[j] : 0);
          V[j] = (uint64_t)t; cr = (uint64_t)(t >> 64);
        }
      }
      uint64_t *__restrict__ dst = a.big.at(poly, (size_t)a.Wout << a.logn) + coef0 + lane;
      // mpi_smod by 2^logq: sign-extend from bit logq-1 (word sw, bit sbit); words above it are the sign
      const int sw = (int)((a.logq - 1) >> 6);
      const unsigned up = 63 - ((a.logq - 1) & 63);
      uint64_t ext = 0;
#pragma unroll
      for (int j = 0; j < WL; ++j) if (j == sw) ext = (uint64_t)((int64_t)(V[j] << up) >> up);
      const uint64_t qsign = (uint64_t)((int64_t)ext >> 63);
#pragma unroll
      for (int j = 0; j < WL; ++j)
        if (j < (int)a.Wout) dst[(size_t)j << a.logn] = j < sw ? V[j] : (j == sw ? ext : qsign);
      for (unsigned j = WL; j < a.Wout; ++j) dst[(size_t)j << a.logn] = qsign;
    }
  }
}

// ---------------------------------------------------------------------------
// Front of the relinearisation tail (src/he-mult.c:67-77, src/he-automorphism.c:68-76; see bridge_exactdiv
// for the exact-division argument).  x = CRT(chat) over dimB limbs, P = p_0..p_{dimP-1}, r = x mod P,
// Q = (x - r)/P limb-wise on the limbs j >= dimP.  With y_d = chat_d * phat_invmp_d (d < dimP) and
// S = sum_d y_d * (P/p_d) = r + k P:
//   S mod p_j        by the matrix cores: bytes(y) x balanced digits of ((P/p_d) 256^i mod p_j)   (like rns_decompose)
//   k, [r > P/2]     from F = sum_d y_d floor(2^104/p_d) (14 more columns): k = F >> 104, round bit = bit 103
//   r mod p_j        = S mod p_j - k (P mod p_j)
//   yq_j             = (chat_j - r) * (P^-1 * phat'_invmp_j) mod p_j : Q's residue, already scaled for the CRT over the j limbs
// F underestimates by < 2^66.  If that makes k one too small, r comes out as r + P and rounds UP, Q one less: the sum
// Q + round is the same -- only the 1/2 boundary matters.  Coefficients with frac(F) in [1/2 - 2^-38, 1/2) are marked
// RF_AMB; the exact kernel settles their round bits afterwards (k stays consistent either way).
// No big r, no decompose of it, no separate exact-division pass.
// ---------------------------------------------------------------------------
struct RelinFrontArgs {
  const uint64_t *chat;      // [polys][dimB][n]
  uint64_t *yq;              // [polys][cnt][n]
  const v4i *bfrag;          // [NT][KS][64]; column tile NT-1 holds the 14 F columns
  const uint64_t *lk;        // [4 KS][2]: p_d, phat_invmp_d (d < dimP; p = 0 padding)
  const uint64_t *pk;        // [4 (NT-1)][3]: p_j, Kq_j, w_j = P^-1 phat'_invmp_j mod p_j
  const uint64_t *tkp;       // [cnt][64]: (p_j - (k P mod p_j)) mod p_j
  const uint64_t *kf;        // [2]
  unsigned char *flags;      // [polys][n]  RF_*
  unsigned char *amb;        // [polys][n]  1 where RF_AMB
  unsigned dimB, dimP, cnt, logn, NT, groups_per_poly, total_groups;
  // optional [polys][n]: when given, only the groups of 64 coefficients that hold a non-zero entry are processed and flags / amb
  // are left alone -- the re-run behind bridge_relin_tail_mfma, which needs yq in memory for the few coefficients it could not finish
  const unsigned char *only;
  unsigned only_writes_flags;   // with `only`: write flags / amb for the groups it re-runs after all (behind the one-product tail nothing else has)
  unsigned prescaled;        // the limbs below dimP already hold y_d = chat_d * phat_invmp_d (the key switch's inverse pass scaled them: ScaledInverse)
  unsigned wscaled;          // ... and the limbs above hold chat_j * w_j, bfrag / pk / tkp are the w-scaled tables: yq_j = x'_j - (r w_j mod p_j)
  FlagScope scope;           // with `only`: the launch that wrote it (bridge_stream.hpp); the grid is then scope.waves / 4 workgroups, so that
                             // wave w of this launch walks the groups of the producer's wave w
};

#ifndef GPQ_FRONT_OCC
#define GPQ_FRONT_OCC 3   /* waves per SIMD asked of the compiler for P of up to 16 limbs (measured +3 % on the whole he_mul against 2, 28 B of scratch) */
#endif
template <int KS>
__global__ __launch_bounds__(256, (KS <= 4 ? GPQ_FRONT_OCC : 2)) void bridge_relin_front_mfma(RelinFrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.scope.wave_any) {                                  // nothing flagged by the four producer waves this workgroup stands for: done (uniform)
    unsigned flagged = 0;
    for (unsigned k = 0; k < 4; ++k) if (blockIdx.x * 4 + k < a.scope.waves) flagged |= a.scope.wave_any[blockIdx.x * 4 + k];
    if (!flagged) return;
  }
  v4i *Bl = reinterpret_cast<v4i *>(smem);
  const unsigned nB = a.NT * KS * 64;
  uint64_t *lkl = reinterpret_cast<uint64_t *>(smem + (size_t)nB * 16);                 // 8 KS words
  uint64_t *pkl = lkl + 8 * KS;                                                          // 12 (NT-1) words
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (unsigned i = threadIdx.x; i < nB; i += 256) Bl[i] = a.bfrag[i];
  for (unsigned i = threadIdx.x; i < 8u * KS; i += 256) lkl[i] = a.lk[i];
  for (unsigned i = threadIdx.x; i < 12 * (a.NT - 1); i += 256) pkl[i] = a.pk[i];
  __syncthreads();
  const unsigned r = lane & 31, h = lane >> 5;
  const uint64_t kf0 = a.kf[0], kf1 = a.kf[1];
  // raw residues of the limbs below dimP for the data fragments: lane (r, h) holds limbs 4s+2h, 4s+2h+1 of
  // coefficients r and 32+r; the next group's are fetched under the current group's work
  uint64_t raw[KS][4];
  auto load_raw = [&](unsigned g) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    const uint64_t *__restrict__ src = a.chat + ((size_t)poly * a.dimB << a.logn) + coef0 + r;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned d = 4 * s + 2 * h + e, dc = d < a.dimP ? d : a.dimP - 1;
        raw[s][e] = src[(size_t)dc << a.logn];
        raw[s][2 + e] = src[((size_t)dc << a.logn) + 32];
      }
  };
  const unsigned g0 = blockIdx.x * 4 + wave, gstep = gridDim.x * 4;
  const bool masked = a.only != nullptr;
  if (!masked && g0 < a.total_groups) load_raw(g0);
  const bool idle = a.scope.wave_any && (g0 >= a.scope.waves || !a.scope.wave_any[g0]);      // the producer's wave g0 flagged nothing
  for (unsigned g = idle ? a.total_groups : g0; g < a.total_groups; g += gstep) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    if (masked) {
      if (!__builtin_amdgcn_ballot_w64(a.only[((size_t)poly << a.logn) + coef0 + lane] != 0)) continue;
      load_raw(g);
    }
    // from here on this lane finishes coefficient coef0 + lane
    const uint64_t *__restrict__ src = a.chat + (((size_t)poly * a.dimB + a.dimP) << a.logn) + coef0 + lane;
    uint64_t *__restrict__ dst = a.yq + ((size_t)poly * a.cnt << a.logn) + coef0 + lane;
    if (!masked && KS >= GPQ_FRONT_XG && g != g0) load_raw(g);     // no registers for the early fetch with 32 limbs in P
    v4i X[2][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint64_t y[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned d = 4 * s + 2 * h + e;
        const uint64_t p = lkl[2 * d], w = lkl[2 * d + 1];
        PrimeK k;
        k.p = p; k.p2 = p << 1; k.c = (uint32_t)p; k.c1 = k.c + 1;
        const uint64_t m = p ? ~0ull : 0ull;
        y[e] = ((a.prescaled ? raw[s][e] : canon_fold(mulmod_lazy(raw[s][e], w, k), p, k.c)) ^ 0x8080808080808080ull) & m;
        y[2 + e] = ((a.prescaled ? raw[s][2 + e] : canon_fold(mulmod_lazy(raw[s][2 + e], w, k), p, k.c)) ^ 0x8080808080808080ull) & m;
      }
      X[0][s] = v4i{(int)(uint32_t)y[0], (int)(uint32_t)(y[0] >> 32), (int)(uint32_t)y[1], (int)(uint32_t)(y[1] >> 32)};
      X[1][s] = v4i{(int)(uint32_t)y[2], (int)(uint32_t)(y[2] >> 32), (int)(uint32_t)y[3], (int)(uint32_t)(y[3] >> 32)};
      __builtin_amdgcn_sched_barrier(0);   // four multiplies at a time: interleaving all 4 KS of them spills
    }
    if (!masked && KS < GPQ_FRONT_XG && g + gstep < a.total_groups) load_raw(g + gstep);
    // residues chat_j of this lane's coefficient, one row tile (4 limbs) ahead
    uint64_t xn[4];
    auto fetch_x = [&](unsigned q) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned j = 4 * q + w, jc = j < a.cnt ? j : a.cnt - 1;
        xn[w] = src[(size_t)jc << a.logn];
      }
    };
    fetch_x(0);
    // row tile NT-1 first: F -> k and the round bit
    unsigned kk;
    {
      v16i f0, f1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { f0[e] = 0; f1[e] = 0; }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const v4i cf = Bl[((a.NT - 1) * KS + s) * 64 + lane];
        f0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[0][s], f0, 0, 0, 0);
        f1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[1][s], f1, 0, 0, 0);
      }
      int64_t L0, H0, L1, H1, carry = 0;
      swap_halves(horner4(f0[0], f0[1], f0[2], f0[3]), horner4(f1[0], f1[1], f1[2], f1[3]), L0, H0);
      swap_halves(horner4(f0[4], f0[5], f0[6], f0[7]), horner4(f1[4], f1[5], f1[6], f1[7]), L1, H1);
      const uint64_t F0 = fold_word(L0, H0, carry), F1 = fold_word(L1, H1, carry);
      const u128 F = (((u128)F1 << 64) | F0) + (((u128)kf1 << 64) | kf0);
      const uint64_t f_hi = (uint64_t)(F >> 64);
      const bool ambiguous = ((f_hi >> 2) & ((1ull << 38) - 1)) == ((1ull << 37) - 1);
      const unsigned gt = (unsigned)(f_hi >> 39) & 1;
      const size_t flag_at = ((size_t)poly << a.logn) + coef0 + lane;
      if (!masked || a.only_writes_flags) {
        a.flags[flag_at] = (unsigned char)(ambiguous ? RF_AMB : (gt ? RF_GT : RF_LT));
        a.amb[flag_at] = ambiguous;
      }
      kk = (unsigned)(f_hi >> 40);
    }
    for (unsigned q = 0; q + 1 < a.NT; ++q) {
      uint64_t xc[4], tk[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned j = 4 * q + w, jc = j < a.cnt ? j : a.cnt - 1;
        xc[w] = xn[w];
        tk[w] = a.tkp[(size_t)jc * 64 + kk];
      }
      if (q + 2 < a.NT) fetch_x(q + 1);
      v16i acc0, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[e] = 0; acc1[e] = 0; }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const v4i cf = Bl[(q * KS + s) * 64 + lane];
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[0][s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[1][s], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int w = 0; w < 4; ++w) {                                        // limb dimP + j, j = 4q + w
        int64_t L, H;
        swap_halves(horner4(acc0[4 * w], acc0[4 * w + 1], acc0[4 * w + 2], acc0[4 * w + 3]),
                    horner4(acc1[4 * w], acc1[4 * w + 1], acc1[4 * w + 2], acc1[4 * w + 3]), L, H);
        const unsigned j = 4 * q + w;
        const uint64_t p = pkl[3 * j], kq = pkl[3 * j + 1], wj = pkl[3 * j + 2];
        PrimeK k;
        k.p = p; k.p2 = p << 1; k.c = (uint32_t)p; k.c1 = k.c + 1;
        const int Hh = (int)(H >> 27);
        const uint64_t Hl = (uint64_t)H & 0x7ffffffu;
        uint64_t v = (Hl << 32) + (uint64_t)L + kq;
        v = (uint64_t)((int64_t)(-(int)k.c) * Hh + (int64_t)v) + tk[w];     // r mod p_j, lazily: in (0, 4p)
        const uint64_t yq = a.wscaled ? canon_fold(xc[w] + (p << 2) - v, p, k.c)                          // x'_j - r w_j, in (0, 5p)
                                      : canon_fold(mulmod_lazy(xc[w] + (p << 2) - v, wj, k), p, k.c);  // (x - r) in (0, 5p), times w_j
        if (j < a.cnt) dst[(size_t)j << a.logn] = yq;
#ifdef GPQ_FRONT_ITEM_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
  }
}

// ---------------------------------------------------------------------------
// The relinearisation tail in ONE pass over a coefficient: bridge_relin_front_mfma's work and the CRT of Q that
// bridge_reconstruct_low_mfma does on its output, without the round trip of Q's residues (cnt limbs written and read again:
// 60 of the 133 words the two kernels move per coefficient at the headline shape).  Row tile q of the front yields Q's scaled
// residues of limbs 4q .. 4q+3 for the coefficient a lane finishes -- exactly k step q of the CRT product, whose data fragment
// wants limbs 4q+2h, 4q+2h+1 of coefficients r (tile 0) and 32+r (tile 1) on lane (r, h): two v_permlane32_swap per pair of
// limbs put them there (lower lanes keep limbs 4q, 4q+1 of their own coefficient and receive those of lane 32+r's; upper lanes
// keep 4q+2, 4q+3 and receive lane r's), and the CRT accumulators take the step while the front works on its next row tile.
// Epilogue = bridge_reconstruct_low_mfma's with the round bit taken from this kernel's own F columns.
// What it cannot finish -- r / P within 2^-38 below 1/2 (RF_AMB) or Q / Pi' likewise (the CRT's own window) -- is flagged in
// `redo` and finished as before: bridge_roundfix settles the round bits, bridge_relin_front_mfma re-runs on the flagged groups
// only (a.only) to put Q's residues in memory, the exact CRT kernel and bridge_addround complete those coefficients.
// ---------------------------------------------------------------------------
struct RelinTailArgs {
  RelinFrontArgs f;          // chat, tables of the front, flags / amb (written), yq unused, only = nullptr
  const v4i *rfrag;          // [KSr][NTR][64]  constant fragments of the CRT over the cnt limbs (get_recon_mfma of that basis)
  const uint64_t *rkc;       // [WL + 2]
  const uint64_t *rpm;       // [65][WL]
  unsigned char *redo;       // [polys][n]  1 = not finished here
  unsigned char *tie;        // [polys][n]  cleared
  Two<uint64_t> out;         // [polys][W][n]
  Two<const uint64_t> addend;
  unsigned W, logq, KSr;
};

// One wave per SIMD (4 per CU) with the whole register file: the CRT accumulators, the front's fragments and every global load
// of the NEXT group (the limbs of P for the fragments, the cnt residues and the addend of the lane's own coefficient: ~120
// registers) issued before the current group's arithmetic -- at two waves per SIMD the same code spills 80 registers and runs
// 25 % slower than the two kernels it replaces (profiles/r03).
#ifndef GPQ_TAIL_WAVES
#define GPQ_TAIL_WAVES 1   /* waves per SIMD of bridge_relin_tail_mfma: 1 = the whole next group's residues in registers (no spill), 2 = residues one row tile ahead (73 spilled registers: slower) */
#endif
constexpr int RELIN_TAIL_MAXTILES = 8;     // row tiles of the front = k steps of the CRT: cnt <= 32 limbs above P

template <int KS, int WL>
__global__ __launch_bounds__(256, GPQ_TAIL_WAVES) void bridge_relin_tail_mfma(RelinTailArgs t) {
  constexpr int NTR = (8 * WL + 14 + 31) / 32;
  constexpr int MT = RELIN_TAIL_MAXTILES;
  const RelinFrontArgs &a = t.f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4i *Bl = reinterpret_cast<v4i *>(smem);
  const unsigned nB = a.NT * KS * 64, nR = t.KSr * NTR * 64;
  v4i *Rl = Bl + nB;
  uint64_t *lkl = reinterpret_cast<uint64_t *>(smem + (size_t)(nB + nR) * 16);          // 8 KS words
  uint64_t *pkl = lkl + 8 * KS;                                                          // 12 (NT-1) words
  // (the wave index is uniform, but the compiler only knows it after readfirstlane: with it every group / polynomial / limb offset
  // below is scalar and the loads take an SGPR base + a 32-bit lane offset instead of thirty 64-bit VGPR addresses)
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (unsigned i = threadIdx.x; i < nB; i += 256) Bl[i] = a.bfrag[i];
  for (unsigned i = threadIdx.x; i < nR; i += 256) Rl[i] = t.rfrag[i];
  for (unsigned i = threadIdx.x; i < 8u * KS; i += 256) lkl[i] = a.lk[i];
  for (unsigned i = threadIdx.x; i < 12 * (a.NT - 1); i += 256) pkl[i] = a.pk[i];
  __syncthreads();
  const unsigned r = lane & 31, h = lane >> 5;
  const unsigned lg = a.logn - 6;                                                        // groups per polynomial = n / 64: shifts, not a (vector) division
  const unsigned ntp = a.NT - 1;                                                         // row tiles in use (<= MT)
  const uint64_t kf0 = a.kf[0], kf1 = a.kf[1];
  const unsigned g0 = blockIdx.x * 4 + wave, gstep = gridDim.x * 4;
  // Everything a group reads from global memory is fetched one group ahead INTO THE SAME REGISTERS: a value is replaced by the
  // next group's as soon as the current group has consumed it (the residues of limbs 4q .. 4q+3 after row tile q; the limbs of P
  // for the next fragments and this group's addend at the start of the epilogue), so the loads fly under the group's arithmetic.
#if GPQ_TAIL_WAVES >= 2
  uint64_t raw[KS][4], xn[4], dd[WL];                                    // two waves per SIMD: the residues one row tile ahead only
#else
  uint64_t raw[KS][4], xj[4 * MT], dd[WL];
#endif
  // uniform bases (SGPRs): the limbs of P of a group, the limbs above P, the addend; lane offsets apart
  const unsigned roff = ((2 * h) << a.logn) + r;                         // fragment lanes: limb 4s + 2h + e of coefficient r (+ 32 for tile 1)
  auto src_p = [&](unsigned g) {
    const unsigned poly = g >> lg, coef0 = (g & ((1u << lg) - 1)) << 6;
    return a.chat + ((size_t)poly * a.dimB << a.logn) + coef0;
  };
  auto src_j = [&](unsigned g) {
    const unsigned poly = g >> lg, coef0 = (g & ((1u << lg) - 1)) << 6;
    return a.chat + (((size_t)poly * a.dimB + a.dimP) << a.logn) + coef0;
  };
  auto src_d = [&](unsigned g) -> const uint64_t * {
    const unsigned poly = g >> lg, coef0 = (g & ((1u << lg) - 1)) << 6;
    const uint64_t *base = t.addend.at(poly, (size_t)t.W << a.logn);                     // uniform per group
    return base ? base + coef0 : nullptr;
  };
  // (limbs 4s + 2h + e up to 4 KS - 1 <= dimP + 3: the padding limbs read real limbs above P -- in bounds, cnt >= 4 -- and are masked)
  auto load_raw = [&](const uint64_t *__restrict__ srcp) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint64_t *__restrict__ ub = srcp + ((size_t)(4 * s + e) << a.logn);
        raw[s][e] = ub[roff];
        raw[s][2 + e] = ub[roff + 32];
      }
  };
  auto load_tile = [&](const uint64_t *__restrict__ src, int q) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned j = 4 * q + w, jc = __builtin_amdgcn_readfirstlane(j < a.cnt ? j : a.cnt - 1);   // scalar: left to the compiler the clamp and the
#if GPQ_TAIL_WAVES >= 2
      xn[w] = (src + ((size_t)jc << a.logn))[lane];
#else
      xj[4 * q + w] = (src + ((size_t)jc << a.logn))[lane];                                           // thirty 64-bit offsets end up in (hoisted) VGPRs
#endif
    }
  };
  auto load_dd = [&](const uint64_t *__restrict__ dp) {
    if (dp) {
#pragma unroll
      for (int j = 0; j < WL; ++j) dd[j] = (dp + ((size_t)(j < (int)t.W ? j : 0) << a.logn))[lane];
    }
  };
  if (g0 < a.total_groups) {
    load_raw(src_p(g0));
    const uint64_t *sj = src_j(g0);
#if GPQ_TAIL_WAVES >= 2
    load_tile(sj, 0);
#else
#pragma unroll
    for (int q = 0; q < MT; ++q) load_tile(sj, q);
#endif
  }
  for (unsigned g = g0; g < a.total_groups; g += gstep) {
    const unsigned poly = g >> lg, coef0 = (g & ((1u << lg) - 1)) << 6;
    const size_t flag_at = ((size_t)poly << a.logn) + coef0 + lane;
    const bool has_addend = t.addend.at(poly, 0) != nullptr;
    const unsigned gn = g + gstep < a.total_groups ? g + gstep : g;      // next group (or this one again: harmless reads)
    // ---- the front's data fragments: y_d = chat_d * phat_invmp_d for the limbs of P
    v4i X[2][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint64_t y[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned d = 4 * s + 2 * h + e;
        const uint64_t p = lkl[2 * d], w = lkl[2 * d + 1];
        PrimeK k;
        k.p = p; k.p2 = p << 1; k.c = (uint32_t)p; k.c1 = k.c + 1;
        const uint64_t m = p ? ~0ull : 0ull;
        y[e] = ((a.prescaled ? raw[s][e] : canon_fold(mulmod_lazy(raw[s][e], w, k), p, k.c)) ^ 0x8080808080808080ull) & m;
        y[2 + e] = ((a.prescaled ? raw[s][2 + e] : canon_fold(mulmod_lazy(raw[s][2 + e], w, k), p, k.c)) ^ 0x8080808080808080ull) & m;
      }
      X[0][s] = v4i{(int)(uint32_t)y[0], (int)(uint32_t)(y[0] >> 32), (int)(uint32_t)y[1], (int)(uint32_t)(y[1] >> 32)};
      X[1][s] = v4i{(int)(uint32_t)y[2], (int)(uint32_t)(y[2] >> 32), (int)(uint32_t)y[3], (int)(uint32_t)(y[3] >> 32)};
      __builtin_amdgcn_sched_barrier(0);
    }
    const uint64_t *__restrict__ sjn = src_j(gn);
#if GPQ_TAIL_WAVES >= 2
    const uint64_t *__restrict__ sjc = src_j(g);
#endif
    // ---- row tile NT-1 of the front first: F -> k and the round bit
    unsigned kk, gt;
    bool amb_front;
    {
      v16i f0, f1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { f0[e] = 0; f1[e] = 0; }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const v4i cf = Bl[((a.NT - 1) * KS + s) * 64 + lane];
        f0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[0][s], f0, 0, 0, 0);
        f1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[1][s], f1, 0, 0, 0);
      }
      int64_t L0, H0, L1, H1, carry = 0;
      swap_halves(horner4(f0[0], f0[1], f0[2], f0[3]), horner4(f1[0], f1[1], f1[2], f1[3]), L0, H0);
      swap_halves(horner4(f0[4], f0[5], f0[6], f0[7]), horner4(f1[4], f1[5], f1[6], f1[7]), L1, H1);
      const uint64_t F0 = fold_word(L0, H0, carry), F1 = fold_word(L1, H1, carry);
      const u128 F = (((u128)F1 << 64) | F0) + (((u128)kf1 << 64) | kf0);
      const uint64_t f_hi = (uint64_t)(F >> 64);
      amb_front = ((f_hi >> 2) & ((1ull << 38) - 1)) == ((1ull << 37) - 1);
      gt = (unsigned)(f_hi >> 39) & 1;
      a.flags[flag_at] = (unsigned char)(amb_front ? RF_AMB : (gt ? RF_GT : RF_LT));
      a.amb[flag_at] = amb_front;
      kk = (unsigned)(f_hi >> 40);
    }
    // ---- row tiles of the front, each feeding one k step of the CRT of Q
    v16i acc[2][NTR];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int qq = 0; qq < NTR; ++qq)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tt][qq][e] = 0;
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      if ((unsigned)q < ntp) {                                           // uniform
        uint64_t tk[4];
#if GPQ_TAIL_WAVES >= 2
        uint64_t xc[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) xc[w] = xn[w];
        if ((unsigned)q + 1 < ntp) load_tile(sjc, q + 1); else load_tile(sjn, 0);   // next row tile, or the next group's first
#endif
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned j = 4 * q + w, jc = __builtin_amdgcn_readfirstlane(j < a.cnt ? j : a.cnt - 1);
          tk[w] = (a.tkp + (size_t)jc * 64)[kk];
        }
        v16i acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0; acc1[e] = 0; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const v4i cf = Bl[(q * KS + s) * 64 + lane];
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[0][s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cf, X[1][s], acc1, 0, 0, 0);
        }
        uint64_t yb[4];                                                  // Q's scaled residues of limbs 4q .. 4q+3 as signed bytes (0 for padding limbs)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          int64_t L, H;
          swap_halves(horner4(acc0[4 * w], acc0[4 * w + 1], acc0[4 * w + 2], acc0[4 * w + 3]),
                      horner4(acc1[4 * w], acc1[4 * w + 1], acc1[4 * w + 2], acc1[4 * w + 3]), L, H);
          const unsigned j = 4 * q + w;
          const uint64_t p = pkl[3 * j], kq = pkl[3 * j + 1], wj = pkl[3 * j + 2];
          PrimeK k;
          k.p = p; k.p2 = p << 1; k.c = (uint32_t)p; k.c1 = k.c + 1;
          const int Hh = (int)(H >> 27);
          const uint64_t Hl = (uint64_t)H & 0x7ffffffu;
          uint64_t v = (Hl << 32) + (uint64_t)L + kq;
          v = (uint64_t)((int64_t)(-(int)k.c) * Hh + (int64_t)v) + tk[w];   // r mod p_j, lazily: in (0, 4p)
#if GPQ_TAIL_WAVES >= 2
          const uint64_t yq = a.wscaled ? canon_fold(xc[w] + (p << 2) - v, p, k.c) : canon_fold(mulmod_lazy(xc[w] + (p << 2) - v, wj, k), p, k.c);
#else
          const uint64_t yq = a.wscaled ? canon_fold(xj[4 * q + w] + (p << 2) - v, p, k.c) : canon_fold(mulmod_lazy(xj[4 * q + w] + (p << 2) - v, wj, k), p, k.c);
#endif
          yb[w] = (yq ^ 0x8080808080808080ull) & (p ? ~0ull : 0ull);
        }
#if GPQ_TAIL_WAVES < 2
        load_tile(sjn, q);                                               // these four registers now wait for the next group
#endif
        // limbs (4q, 4q+2) and (4q+1, 4q+3) change lane halves: [0] = tile 0's fragment words, [1] = tile 1's
        const auto e0l = __builtin_amdgcn_permlane32_swap((unsigned)yb[0], (unsigned)yb[2], false, false);
        const auto e0h = __builtin_amdgcn_permlane32_swap((unsigned)(yb[0] >> 32), (unsigned)(yb[2] >> 32), false, false);
        const auto e1l = __builtin_amdgcn_permlane32_swap((unsigned)yb[1], (unsigned)yb[3], false, false);
        const auto e1h = __builtin_amdgcn_permlane32_swap((unsigned)(yb[1] >> 32), (unsigned)(yb[3] >> 32), false, false);
        const v4i A0 = v4i{(int)e0l[0], (int)e0h[0], (int)e1l[0], (int)e1h[0]};
        const v4i A1 = v4i{(int)e0l[1], (int)e0h[1], (int)e1l[1], (int)e1h[1]};
#pragma unroll
        for (int qq = 0; qq < NTR; ++qq) {
          const v4i b = Rl[(q * NTR + qq) * 64 + lane];
          acc[0][qq] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, A0, acc[0][qq], 0, 0, 0);
          acc[1][qq] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, A1, acc[1][qq], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                               // one row tile at a time: hoisting the tiles' loads over each other spills
      }
    }
    // ---- CRT epilogue (bridge_reconstruct_low_mfma's) + finish.  The front's fragments are dead: this group's addend and the
    // next group's limbs of P are fetched under the column folding (earlier they would not fit the registers).
    load_dd(src_d(g));
    load_raw(src_p(gn));
    uint64_t V[4 * NTR];
    int64_t carry = 0;
#pragma unroll
    for (int qq = 0; qq < NTR; ++qq)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        int64_t L, H;
        swap_halves(horner4(acc[0][qq][4 * w], acc[0][qq][4 * w + 1], acc[0][qq][4 * w + 2], acc[0][qq][4 * w + 3]),
                    horner4(acc[1][qq][4 * w], acc[1][qq][4 * w + 1], acc[1][qq][4 * w + 2], acc[1][qq][4 * w + 3]), L, H);
        if (4 * qq + w == WL) carry = 0;
        V[4 * qq + w] = fold_word(L, H, carry);
        if (w == 3) __builtin_amdgcn_sched_barrier(0);   // a row tile of accumulators at a time: read out all at once they spill
      }
    const u128 F = (((u128)V[WL + 1] << 64) | V[WL]) + (((u128)t.rkc[WL + 1] << 64) | t.rkc[WL]);
    const uint64_t f1 = (uint64_t)(F >> 64);
    const bool unfinished = amb_front || ((f1 >> 2) & ((1ull << 38) - 1)) == ((1ull << 37) - 1);
    t.redo[flag_at] = unfinished;
    t.tie[flag_at] = 0;
    if (!unfinished) {
      const unsigned mult = (unsigned)(f1 >> 40) + (unsigned)((f1 >> 39) & 1);
      const uint64_t *__restrict__ P = t.rpm + (size_t)mult * WL;
      uint64_t borrow = 0;
#pragma unroll
      for (int j = 0; j < WL; ++j) {
        const u128 d2 = (u128)V[j] - P[j] - borrow;
        V[j] = (uint64_t)d2;
        borrow = (uint64_t)(d2 >> 64) & 1;
      }
      uint64_t cr = gt;                                    // + [r > floor(P/2)] + d   (mod 2^logq: only the low words matter)
#pragma unroll
      for (int j = 0; j < WL; ++j) {
        const u128 s2 = (u128)V[j] + cr + ((has_addend && j < (int)t.W) ? dd[j] : 0);
        V[j] = (uint64_t)s2; cr = (uint64_t)(s2 >> 64);
      }
      uint64_t *__restrict__ dst = t.out.at(poly, (size_t)t.W << a.logn) + coef0;        // uniform; + lane below
      const int sw = (int)((t.logq - 1) >> 6);
      const unsigned up = 63 - ((t.logq - 1) & 63);
      uint64_t ext = 0;
#pragma unroll
      for (int j = 0; j < WL; ++j) if (j == sw) ext = (uint64_t)((int64_t)(V[j] << up) >> up);
      const uint64_t qsign = (uint64_t)((int64_t)ext >> 63);
#pragma unroll
      for (int j = 0; j < WL; ++j)
        if (j < (int)t.W) (dst + ((size_t)j << a.logn))[lane] = j < sw ? V[j] : (j == sw ? ext : qsign);
      for (unsigned j = WL; j < t.W; ++j) (dst + ((size_t)j << a.logn))[lane] = qsign;
    }
  }
}

}  // namespace gpq
