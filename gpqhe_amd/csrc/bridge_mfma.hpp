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

namespace gpq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MFMA_TILE_RS = 36;                        // words per row of a 32x32 i32 tile in LDS (16-B aligned rows)
constexpr int MFMA_TILE_WORDS = 32 * MFMA_TILE_RS;

// A wave's LDS instructions execute in issue order, so its reads see its earlier writes; this only stops
// the compiler from moving LDS accesses across the exchange.  (A wavefront-scope fence also waits for the
// outstanding GLOBAL stores -- vmcnt(0) -- at every tile: measured 8x slower here.)
__device__ __forceinline__ void mfma_wave_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// one 32x32 accumulator tile -> the wave's LDS tile, [row][col]
__device__ __forceinline__ void tile_store(int *tile, const v16i &acc, unsigned lane) {
  const unsigned c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 16; ++g) tile[((g & 3) + 8 * (g >> 2) + 4 * h) * MFMA_TILE_RS + c] = acc[g];
}

// ---------------------------------------------------------------------------
// rns_decompose on the matrix cores.
//   A[m][k]        = byte k of coefficient m, as s = u - 128 (top byte of the top word: signed as it is)
//   B[k][8j + b]   = balanced digit b of T_jk = 256^k mod p_j
//   C[m][8j + b]   = sum_k A B        |C| <= 8W * 2^14
//   x mod p_j      = sum_b C[8j+b] 256^b + K_j,     K_j = 128 * sum_{k < 8W-1} 256^k mod p_j
// One wave owns 64 consecutive coefficients (two 32-row tiles); the constant matrix sits in LDS.
// Epilogue per (coefficient, prime): H = C7..C4, L = C3..C0 by Horner (46 bits each),
//   V = H 2^32 + L,  H = Hh 2^27 + Hl  =>  V == Hl 2^32 + L - c Hh  (mod p),  + Kq_j, in (0, 3p) -> canonical.
// ---------------------------------------------------------------------------
struct DecomposeMfmaArgs {
  const uint64_t *big;       // [polys][W][n]
  uint64_t *slab;            // [polys][dim][n]
  const v4i *bfrag;          // [NT][KS][64]: the B fragment of lane l for (column tile, k step)
  const uint64_t *pk;        // [4 NT][3]: p_j, Kq_j = 2^50 + ((K_j - 2^50) mod p_j), c_j   (zeros for padding primes)
  unsigned W, dim, logn, NT;               // NT: column tiles of 4 primes, a multiple of 4
  unsigned groups_per_poly, total_groups;  // groups of 64 coefficients
};

constexpr int NTG = 1;   // column tiles (of 4 primes) per accumulator group: 2 tiles x 2 row tiles x 16 = 64 accumulator registers

template <int KS>
__global__ __launch_bounds__(256, 2) void bridge_decompose_mfma(DecomposeMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4i *Bl = reinterpret_cast<v4i *>(smem);
  const unsigned nB = a.NT * KS * 64;
  uint64_t *pkl = reinterpret_cast<uint64_t *>(smem + (size_t)nB * 16);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int *tile = reinterpret_cast<int *>(smem + (size_t)nB * 16 + (size_t)a.NT * 4 * 24) + wave * MFMA_TILE_WORDS;
  for (unsigned i = threadIdx.x; i < nB; i += 256) Bl[i] = a.bfrag[i];
  for (unsigned i = threadIdx.x; i < a.NT * 12; i += 256) pkl[i] = a.pk[i];
  __syncthreads();
  const unsigned r = lane & 31, h = lane >> 5;
  v4i A[2][KS];
  // the A fragments of a group of 64 coefficients: lane (r, h) holds words 4s+2h, 4s+2h+1 of row r of each tile
  auto load_A = [&](unsigned g) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    const uint64_t *__restrict__ src = a.big + ((size_t)poly * a.W << a.logn) + coef0 + r;
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
          const uint64_t v = src[((size_t)wc << a.logn) + 32 * t];
          x[e] = (v ^ m) & (w < a.W ? ~0ull : 0ull);
        }
        A[t][s] = v4i{(int)(uint32_t)x[0], (int)(uint32_t)(x[0] >> 32), (int)(uint32_t)x[1], (int)(uint32_t)(x[1] >> 32)};
      }
  };
  const unsigned g0 = blockIdx.x * 4 + wave, gstep = gridDim.x * 4;
  if (g0 < a.total_groups) load_A(g0);
  for (unsigned g = g0; g < a.total_groups; g += gstep) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    uint64_t *__restrict__ dst = a.slab + ((size_t)poly * a.dim << a.logn) + coef0;
    if (KS >= 8 && g != g0) load_A(g);                 // 32-word inputs: no registers left for the early fetch
    for (unsigned ng = 0; ng < a.NT; ng += NTG) {
      v16i acc[2][NTG];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < NTG; ++q)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[t][q][e] = 0;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int q = 0; q < NTG; ++q) {
          const v4i b = Bl[((ng + q) * KS + s) * 64 + lane];
          acc[0][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0][s], b, acc[0][q], 0, 0, 0);
          acc[1][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1][s], b, acc[1][q], 0, 0, 0);
        }
      // the fragments are dead after the last column group: fetch the next group's under this one's epilogue
      if (KS < 8 && ng + NTG >= a.NT && g + gstep < a.total_groups) load_A(g + gstep);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < NTG; ++q) {
          tile_store(tile, acc[t][q], lane);
          mfma_wave_sync();
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const unsigned pq = h + 2 * u, j = 4 * (ng + q) + pq;          // item = (row r, prime j)
            const v4i lo = *reinterpret_cast<const v4i *>(tile + r * MFMA_TILE_RS + 8 * pq);
            const v4i hi = *reinterpret_cast<const v4i *>(tile + r * MFMA_TILE_RS + 8 * pq + 4);
            if (j < a.dim) {
              const uint64_t p = pkl[3 * j], kq = pkl[3 * j + 1];
              const int c = (int)(uint32_t)pkl[3 * j + 2];
              const int64_t H = (int64_t)(hi[3] * 256 + hi[2]) * 65536 + (hi[1] * 256 + hi[0]);
              const int64_t L = (int64_t)(lo[3] * 256 + lo[2]) * 65536 + (lo[1] * 256 + lo[0]);
              const int Hh = (int)(H >> 27);
              const uint64_t Hl = (uint64_t)H & 0x7ffffffu;
              uint64_t v = (Hl << 32) + (uint64_t)L + kq;
              v = (uint64_t)((int64_t)(-c) * Hh + (int64_t)v);             // in (0, 3p)
              v = csub(csub(v, p << 1), p);
              dst[((size_t)j << a.logn) + 32 * t + r] = v;
            }
          }
          mfma_wave_sync();
        }
    }
  }
}

// ---------------------------------------------------------------------------
// poly_rns2mpi fast path (q = 2^logq, centred; see bridge_reconstruct_low) on the matrix cores.
//   y_d = ahat_d * phat_invmp_d mod p_d                       VALU, by the lane that owns (row, limb) in the A layout
//   A[m][8d + i]   = byte i of y_d, as s = u - 128
//   B[8d + i][n]   = balanced digit n - i of phat_d mod 2^(64 WL)         columns 0 .. 8WL-1   (S)
//                    balanced digit m - i of floor(2^104 / p_d)           columns 8WL + m, m < 14   (F)
//   S = sum_n C[n] 256^n + Kc  (mod 2^(64 WL)),   F = sum_m C[8WL+m] 256^m + Kf
//   F underestimates 2^104 S/P by less than dim 2^60 < 2^66:  k = F >> 104, centred <=> bit 103,
//   coefficients with frac(F) in [1/2 - 2^-38, 1/2) are flagged for the exact kernel (as bridge_reconstruct_low does
//   with its 2^-61 window); result = S - (k + centred) P, masked and sign-extended at logq.
// K-outer loop: the accumulators of all column tiles stay in registers (2 row tiles x NT x 16), the A
// fragments of one k step (4 limbs) are made on the fly.  Epilogue: lane = coefficient, the byte columns
// are folded 8 at a time into words with a running signed carry.
// ---------------------------------------------------------------------------
struct ReconMfmaArgs {
  const uint64_t *slab;      // [polys][slab_dim][n]
  uint64_t *big;             // [polys][Wout][n]
  const v4i *bfrag;          // [KS][NT][64]
  const uint64_t *lk;        // [4 KS][2]: p_d, phat_invmp_d   (p = 0: padding limb)
  const uint64_t *kc;        // [WL + 2]: Kc words, then Kf (2 words)
  const uint64_t *pm;        // [65][WL]: m * P mod 2^(64 WL)
  unsigned char *redo;       // [polys][n]
  unsigned char *tie;        // optional, cleared
  unsigned dim, KS, logn, Wout, logq, slab_dim, slab_first;
  unsigned groups_per_poly, total_groups;
  // tail of he_relin / he_swk fused in (bridge_addround's work for the coefficients decided here):
  const uint64_t *addend;        // optional [polys][Wout][n]: d of src/he-mult.c:72-76
  const unsigned char *rflags;   // optional [polys][n]: RF_GT = round the quotient up (mpi_rdiv)
  unsigned prescaled;            // the slab already holds y_d = ahat_d * phat_invmp_d (bridge_relin_front_mfma writes it so)
};

// per-coefficient flags of the relinearisation tail: r = x mod P against floor(P/2)
constexpr unsigned char RF_GT = 1, RF_LT = 2, RF_AMB = 4;

// 8 byte columns (signed 32-bit sums) + carry in -> one 64-bit word + carry out
__device__ __forceinline__ uint64_t fold8(const int (&c)[32], int at, int64_t &carry) {
  const int64_t H = (int64_t)(c[at + 7] * 256 + c[at + 6]) * 65536 + (c[at + 5] * 256 + c[at + 4]);
  const int64_t L = (int64_t)(c[at + 3] * 256 + c[at + 2]) * 65536 + (c[at + 1] * 256 + c[at + 0]);
  const int64_t T = carry + L;
  const uint64_t lo = (uint64_t)T + ((uint64_t)H << 32);
  carry = (H >> 32) + (T >> 63) + (lo < (uint64_t)T ? 1 : 0);
  return lo;
}

template <int WL>
__global__ __launch_bounds__(512) void bridge_reconstruct_low_mfma(ReconMfmaArgs a) {
  constexpr int NT = (8 * WL + 14 + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4i *Bl = reinterpret_cast<v4i *>(smem);
  const unsigned nB = a.KS * NT * 64;
  uint64_t *lkl = reinterpret_cast<uint64_t *>(smem + (size_t)nB * 16);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int *tile = reinterpret_cast<int *>(smem + (size_t)nB * 16 + (size_t)a.KS * 64) + wave * 2 * MFMA_TILE_WORDS;
  for (unsigned i = threadIdx.x; i < nB; i += 512) Bl[i] = a.bfrag[i];
  for (unsigned i = threadIdx.x; i < a.KS * 8; i += 512) lkl[i] = a.lk[i];
  __syncthreads();
  const unsigned r = lane & 31, h = lane >> 5, n = 1u << a.logn;
  const unsigned gstep = gridDim.x * 8;
  for (unsigned g = blockIdx.x * 8 + wave; g < a.total_groups; g += gstep) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    const uint64_t *__restrict__ src = a.slab + (((size_t)poly * a.slab_dim + a.slab_first) << a.logn) + coef0 + r;
    v16i acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][q][e] = 0;
    // residues of k step s: limbs 4s+2h, 4s+2h+1 of rows r (tile 0) and 32+r (tile 1); padding limbs read limb dim-1
    uint64_t xn[4];
    auto fetch = [&](unsigned s) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned d = 4 * s + 2 * h + e, dc = d < a.dim ? d : a.dim - 1;
        xn[e] = src[(size_t)dc << a.logn];
        xn[2 + e] = src[((size_t)dc << a.logn) + 32];
      }
    };
    fetch(0);
    for (unsigned s = 0; s < a.KS; ++s) {
      uint64_t x[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = xn[e];
      if (s + 1 < a.KS) fetch(s + 1);
      uint64_t y[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned d = 4 * s + 2 * h + e;
        const uint64_t p = lkl[2 * d], w = lkl[2 * d + 1];
        PrimeK k;
        k.p = p; k.p2 = p << 1; k.c = (uint32_t)p; k.c1 = k.c + 1;   // p = 2^59 + c: c is the low word
        const uint64_t m = p ? ~0ull : 0ull;
        y[e] = ((a.prescaled ? x[e] : mulmod_canon(x[e], w, k)) ^ 0x8080808080808080ull) & m;
        y[2 + e] = ((a.prescaled ? x[2 + e] : mulmod_canon(x[2 + e], w, k)) ^ 0x8080808080808080ull) & m;
      }
      const v4i A0 = v4i{(int)(uint32_t)y[0], (int)(uint32_t)(y[0] >> 32), (int)(uint32_t)y[1], (int)(uint32_t)(y[1] >> 32)};
      const v4i A1 = v4i{(int)(uint32_t)y[2], (int)(uint32_t)(y[2] >> 32), (int)(uint32_t)y[3], (int)(uint32_t)(y[3] >> 32)};
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const v4i b = Bl[(s * NT + q) * 64 + lane];
        acc[0][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, b, acc[0][q], 0, 0, 0);
        acc[1][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, b, acc[1][q], 0, 0, 0);
      }
    }
    // epilogue: lane = coefficient coef0 + lane (row r of tile h)
    constexpr bool EARLY_D = WL <= 14;                     // (no registers left for it at WL = 16)
    uint64_t dd[WL];                                       // d of the fused tail, fetched under the column folding
    if (EARLY_D && a.rflags && a.addend) {
      const uint64_t *__restrict__ dp = a.addend + ((size_t)poly * a.Wout << a.logn) + coef0 + lane;
#pragma unroll
      for (int j = 0; j < WL; ++j) dd[j] = dp[(size_t)(j < (int)a.Wout ? j : 0) << a.logn];
    }
    uint64_t V[4 * NT];
    int64_t carry = 0;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      tile_store(tile, acc[0][q], lane);
      tile_store(tile + MFMA_TILE_WORDS, acc[1][q], lane);
      mfma_wave_sync();
      int c[32];
      const int *row = tile + h * MFMA_TILE_WORDS + r * MFMA_TILE_RS;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const v4i v = *reinterpret_cast<const v4i *>(row + 4 * e);
        c[4 * e] = v[0]; c[4 * e + 1] = v[1]; c[4 * e + 2] = v[2]; c[4 * e + 3] = v[3];
      }
      mfma_wave_sync();
#pragma unroll
      for (int wq = 0; wq < 4; ++wq) {
        if (4 * q + wq == WL) carry = 0;                 // the F columns start their own carry chain
        V[4 * q + wq] = fold8(c, 8 * wq, carry);
      }
    }
    // constants of the signed-byte offsets
    uint64_t cy = 0;
#pragma unroll
    for (int j = 0; j < WL; ++j) {
      const u128 t = (u128)V[j] + a.kc[j] + cy;
      V[j] = (uint64_t)t; cy = (uint64_t)(t >> 64);
    }
    const u128 F = (((u128)V[WL + 1] << 64) | V[WL]) + (((u128)a.kc[WL + 1] << 64) | a.kc[WL]);
    const uint64_t f1 = (uint64_t)(F >> 64);
    const size_t flag_at = ((size_t)poly << a.logn) + coef0 + lane;
    const bool ambiguous = ((f1 >> 2) & ((1ull << 38) - 1)) == ((1ull << 37) - 1);   // frac in [1/2 - 2^-38, 1/2)
    a.redo[flag_at] = ambiguous;
    if (a.tie) a.tie[flag_at] = 0;
    if (!ambiguous) {
      const unsigned mult = (unsigned)(f1 >> 40) + (unsigned)((f1 >> 39) & 1);       // k, plus one when centring takes P off once more
      const uint64_t *__restrict__ P = a.pm + (size_t)mult * WL;
      uint64_t borrow = 0;
#pragma unroll
      for (int j = 0; j < WL; ++j) {
        const u128 t = (u128)V[j] - P[j] - borrow;
        V[j] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
      }
      if (a.rflags) {                                    // + [r > floor(P/2)] + d   (mod 2^logq: only the low words matter)
        if (!EARLY_D && a.addend) {
          const uint64_t *__restrict__ dp = a.addend + ((size_t)poly * a.Wout << a.logn) + coef0 + lane;
#pragma unroll
          for (int j = 0; j < WL; ++j) dd[j] = dp[(size_t)(j < (int)a.Wout ? j : 0) << a.logn];
        }
        uint64_t cr = a.rflags[flag_at] & RF_GT;
#pragma unroll
        for (int j = 0; j < WL; ++j) {
          const u128 t = (u128)V[j] + cr + ((a.addend && j < (int)a.Wout) ? dd[j] : 0);
          V[j] = (uint64_t)t; cr = (uint64_t)(t >> 64);
        }
      }
      uint64_t *__restrict__ dst = a.big + ((size_t)poly * a.Wout << a.logn) + coef0 + lane;
      const unsigned sb = a.logq - 1;
      uint64_t qsign = 0;
#pragma unroll
      for (int j = 0; j < WL; ++j) if (j == (int)(sb >> 6)) qsign = 0 - ((V[j] >> (sb & 63)) & 1);
#pragma unroll
      for (int j = 0; j < WL; ++j) {
        if (j < (int)a.Wout) {
          uint64_t v = V[j];
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
    (void)n;
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
};

template <int KS>
__global__ __launch_bounds__(256, 2) void bridge_relin_front_mfma(RelinFrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4i *Bl = reinterpret_cast<v4i *>(smem);
  const unsigned nB = a.NT * KS * 64;
  uint64_t *lkl = reinterpret_cast<uint64_t *>(smem + (size_t)nB * 16);                 // 8 KS words
  uint64_t *pkl = lkl + 8 * KS;                                                          // 12 (NT-1) words
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char *wbase = smem + (size_t)nB * 16 + (size_t)(8 * KS + 12 * (a.NT - 1)) * 8;
  int *tile = reinterpret_cast<int *>(wbase) + wave * (2 * MFMA_TILE_WORDS + 64);
  int *kbuf = tile + 2 * MFMA_TILE_WORDS;
  for (unsigned i = threadIdx.x; i < nB; i += 256) Bl[i] = a.bfrag[i];
  for (unsigned i = threadIdx.x; i < 8u * KS; i += 256) lkl[i] = a.lk[i];
  for (unsigned i = threadIdx.x; i < 12 * (a.NT - 1); i += 256) pkl[i] = a.pk[i];
  __syncthreads();
  const unsigned r = lane & 31, h = lane >> 5;
  const uint64_t kf0 = a.kf[0], kf1 = a.kf[1];
  for (unsigned g = blockIdx.x * 4 + wave; g < a.total_groups; g += gridDim.x * 4) {
    const unsigned poly = g / a.groups_per_poly, coef0 = (g % a.groups_per_poly) << 6;
    const uint64_t *__restrict__ src = a.chat + ((size_t)poly * a.dimB << a.logn) + coef0;
    uint64_t *__restrict__ dst = a.yq + ((size_t)poly * a.cnt << a.logn) + coef0;
    // A fragments: y of limbs 4s+2h, 4s+2h+1 for rows r (tile 0) and 32+r (tile 1)
    v4i A[2][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint64_t y[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned d = 4 * s + 2 * h + e, dc = d < a.dimP ? d : a.dimP - 1;
        const uint64_t p = lkl[2 * d], w = lkl[2 * d + 1];
        PrimeK k;
        k.p = p; k.p2 = p << 1; k.c = (uint32_t)p; k.c1 = k.c + 1;
        const uint64_t m = p ? ~0ull : 0ull;
        y[e] = (mulmod_canon(src[((size_t)dc << a.logn) + r], w, k) ^ 0x8080808080808080ull) & m;
        y[2 + e] = (mulmod_canon(src[((size_t)dc << a.logn) + 32 + r], w, k) ^ 0x8080808080808080ull) & m;
      }
      A[0][s] = v4i{(int)(uint32_t)y[0], (int)(uint32_t)(y[0] >> 32), (int)(uint32_t)y[1], (int)(uint32_t)(y[1] >> 32)};
      A[1][s] = v4i{(int)(uint32_t)y[2], (int)(uint32_t)(y[2] >> 32), (int)(uint32_t)y[3], (int)(uint32_t)(y[3] >> 32)};
    }
    // residues chat_j of the items of one column tile: (tile t, u) -> limb dimP + 4q + h + 2u, row 32t + r; fetched one
    // column tile ahead so that the loads are under the previous tile's work
    uint64_t xn[4];
    auto fetch_x = [&](unsigned q) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned j = 4 * q + h + 2 * u, jc = j < a.cnt ? j : a.cnt - 1;
          xn[2 * t + u] = src[((size_t)(a.dimP + jc) << a.logn) + 32 * t + r];
        }
    };
    fetch_x(0);
    // column tile NT-1 first: F -> k and the round bit of every coefficient (lane = coefficient)
    {
      v16i f0, f1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { f0[e] = 0; f1[e] = 0; }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const v4i b = Bl[((a.NT - 1) * KS + s) * 64 + lane];
        f0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0][s], b, f0, 0, 0, 0);
        f1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1][s], b, f1, 0, 0, 0);
      }
      tile_store(tile, f0, lane);
      tile_store(tile + MFMA_TILE_WORDS, f1, lane);
      mfma_wave_sync();
      int c[32];
      const int *row = tile + h * MFMA_TILE_WORDS + r * MFMA_TILE_RS;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const v4i v = *reinterpret_cast<const v4i *>(row + 4 * e);
        c[4 * e] = v[0]; c[4 * e + 1] = v[1]; c[4 * e + 2] = v[2]; c[4 * e + 3] = v[3];
      }
#pragma unroll
      for (int e = 16; e < 32; ++e) c[e] = 0;
      int64_t carry = 0;
      const uint64_t F0 = fold8(c, 0, carry), F1 = fold8(c, 8, carry);
      const u128 F = (((u128)F1 << 64) | F0) + (((u128)kf1 << 64) | kf0);
      const uint64_t f_hi = (uint64_t)(F >> 64);
      const bool ambiguous = ((f_hi >> 2) & ((1ull << 38) - 1)) == ((1ull << 37) - 1);
      const unsigned gt = (unsigned)(f_hi >> 39) & 1;
      const size_t flag_at = ((size_t)poly << a.logn) + coef0 + lane;
      a.flags[flag_at] = (unsigned char)(ambiguous ? RF_AMB : (gt ? RF_GT : RF_LT));
      a.amb[flag_at] = ambiguous;
      kbuf[lane] = (int)(f_hi >> 40);
      mfma_wave_sync();
    }
    for (unsigned q = 0; q + 1 < a.NT; ++q) {
      uint64_t xc[4], tk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) xc[e] = xn[e];
      if (q + 2 < a.NT) fetch_x(q + 1);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned j = 4 * q + h + 2 * u, jc = j < a.cnt ? j : a.cnt - 1;
          tk[2 * t + u] = a.tkp[(size_t)jc * 64 + (unsigned)kbuf[32 * t + r]];
        }
      v16i acc0, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[e] = 0; acc1[e] = 0; }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const v4i b = Bl[(q * KS + s) * 64 + lane];
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0][s], b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1][s], b, acc1, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        tile_store(tile, t ? acc1 : acc0, lane);
        mfma_wave_sync();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned pq = h + 2 * u, j = 4 * q + pq;                   // item = (row r, limb dimP + j)
          const v4i lo = *reinterpret_cast<const v4i *>(tile + r * MFMA_TILE_RS + 8 * pq);
          const v4i hi = *reinterpret_cast<const v4i *>(tile + r * MFMA_TILE_RS + 8 * pq + 4);
          const uint64_t p = pkl[3 * j], kq = pkl[3 * j + 1], w = pkl[3 * j + 2];
          PrimeK k;
          k.p = p; k.p2 = p << 1; k.c = (uint32_t)p; k.c1 = k.c + 1;
          const int64_t H = (int64_t)(hi[3] * 256 + hi[2]) * 65536 + (hi[1] * 256 + hi[0]);
          const int64_t L = (int64_t)(lo[3] * 256 + lo[2]) * 65536 + (lo[1] * 256 + lo[0]);
          const int Hh = (int)(H >> 27);
          const uint64_t Hl = (uint64_t)H & 0x7ffffffu;
          uint64_t v = (Hl << 32) + (uint64_t)L + kq;
          v = (uint64_t)((int64_t)(-(int)k.c) * Hh + (int64_t)v) + tk[2 * t + u];   // r mod p_j, lazily: in (0, 4p)
          const uint64_t yq = mulmod_canon_lazy(xc[2 * t + u] + (p << 2) - v, w, k);  // (x - r) in (0, 5p)
          if (j < a.cnt) dst[((size_t)j << a.logn) + 32 * t + r] = yq;
        }
        mfma_wave_sync();
      }
    }
  }
}

}  // namespace gpq
