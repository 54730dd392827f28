// bridge_stream.hpp -- the matrix-core bridge kernels of gpq_he_mul / gpq_he_swk as STREAMS (round 4).
//
// bridge_mfma.hpp's kernels are bound by exposed memory latency: their residue prefetch rotates through register copies, and the
// compiler waits for every outstanding load at the bottom of each two-step loop iteration (vmcnt(0) in the ISA), at two waves per SIMD.
// The kernels here do the same arithmetic -- the same constant matrices, the same column folding, the same windows -- with
//   * 16 bytes per lane on every global access: lane (r, h) reads coefficients 2r, 2r+1 of limb 4s+2h (+1), so the two coefficient
//     tiles of a wave are the EVEN and the ODD coefficients of its group of 64 (tile t, column r = coefficient 2r + t), a wave-level
//     load covers two whole 512-byte limb segments, and the B fragments of both tiles come out of the two loads without a shuffle;
//     results go out the same way after one v_permlane32_swap per register pair (word j / j+1 or limb j / j+1 trade lane halves);
//   * a RING of R k-steps of residues in flight per wave, statically unrolled (no register rotation: a slot is refilled for the step
//     R positions further down the stream -- across phases and across groups -- as soon as the step in it has been turned into
//     fragments), so the loads of the next group fly under the epilogue of this one;
//   * no masks on padding limbs: their rows of the constant matrix are zero;
//   * fusions that keep big integers out of memory (src/he-mult.c:139-141 feeding :59 and :72-76):
//       bridge_crt_decompose   CRT(d2hat) mod q_l -> rns_decompose over the key switch's limbs: the W words of d2 stay in registers
//       bridge_tail_stream     the one-product relinearisation tail + CRT(d0hat | d1hat) + the addition: d0, d1 never exist as words
// What a kernel cannot decide (the 2^-38 windows of bridge_mfma.hpp) is flagged per coefficient in `redo`, and every wave leaves ONE
// word saying whether it flagged anything: the masked exact kernels behind it (FlagScope) are launched over the same wave -> group
// assignment and return at once when their word is 0 -- the common case costs a launch of 256 workgroups that read one word each.
#pragma once
#include "bridge_mfma.hpp"

#ifndef GPQ_STREAM_LOAD_AUX
#define GPQ_STREAM_LOAD_AUX 2  /* cache policy of the streamed loads: 0 default, 2 = nt (every residue / word is read once): measured -4 % on the tail kernel */
#endif
#ifndef GPQ_STREAM_STORE_AUX
#define GPQ_STREAM_STORE_AUX 2 /* ... and of the streamed stores (results the next kernel reads from HBM anyway: the slabs are far larger than the caches): another -8 % */
#endif
#ifndef GPQ_STREAM_PROBE
#define GPQ_STREAM_PROBE 0     /* 1: no MFMA; 2: also no column folding (timing probes of where the streaming kernels' time goes; wrong results) */
#endif

namespace gpq {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// one k step as fetched: limbs (words) 4s+2h and 4s+2h+1 of coefficients 2r, 2r+1
struct StepRegs { v4u lo, hi; };

__device__ __forceinline__ v4i frag_even(const StepRegs &x, unsigned m) { return v4i{(int)(x.lo[0] ^ m), (int)(x.lo[1] ^ m), (int)(x.hi[0] ^ m), (int)(x.hi[1] ^ m)}; }
__device__ __forceinline__ v4i frag_odd(const StepRegs &x, unsigned m) { return v4i{(int)(x.lo[2] ^ m), (int)(x.lo[3] ^ m), (int)(x.hi[2] ^ m), (int)(x.hi[3] ^ m)}; }

// Per group of 64 coefficients the kernels pass the lane's LDS offset through an empty asm: the first ring round of a group reads its
// constant fragments from LDS at fixed addresses, and the compiler would keep them in registers across the group loop (80 VGPRs).
__device__ __forceinline__ unsigned opaque_v(unsigned x) { asm volatile("" : "+v"(x)); return x; }

// Constant tables into LDS at kernel start, 512 threads, four loads in flight per thread (one at a time, the ~100 KB of a workgroup cost a
// dozen dependent round trips: 15-20 us of a 600 us launch).
template <typename T>
__device__ __forceinline__ void stage_lds(T *dst, const T *__restrict__ src, unsigned count) {
  unsigned i = threadIdx.x;
  for (; i + 3 * 512 < count; i += 4 * 512) {
    const T v0 = src[i], v1 = src[i + 512], v2 = src[i + 1024], v3 = src[i + 1536];
    dst[i] = v0; dst[i + 512] = v1; dst[i + 1024] = v2; dst[i + 1536] = v3;
  }
  for (; i < count; i += 512) dst[i] = src[i];
}

// The table of multiples [65][WL] into LDS with rows of WL + 1 words.  Every lane reads ITS row (the multiple of P its coefficient takes off): with
// WL = 16 a row is 128 bytes, rows then differ by 32 banks and the 64 lanes of a ds_read_b64 share two bank pairs -- 11.2 M conflict cycles per launch of
// bridge_tail_stream, the only kernel of the library with any (profiles/r04/v17_mpi_pmc.txt).  At 136 bytes the rows of 32 consecutive multiples
// start in 32 different bank pairs.
#ifndef GPQ_TAIL_ROW
#define GPQ_TAIL_ROW 17   /* 16 = unpadded (round 4), for A/B builds */
#endif
constexpr int kTailRow = GPQ_TAIL_ROW;          // words per row of the tail's tables of multiples in LDS (WL = 16, + 1)
template <int WL>
__device__ __forceinline__ void stage_multiples(uint64_t *dst, const uint64_t *__restrict__ src) {
  for (unsigned i = threadIdx.x; i < 65u * WL; i += 512) dst[(i / WL) * kTailRow + i % WL] = src[i];
}

// A slab as a buffer resource (stride 0, range-checked: an access past the end reads zeros / is dropped).  Built from kernel arguments only.
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc slab_rsrc(const void *p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(unsigned)(bytes < 0xfffff000ull ? bytes : 0xfffff000ull), 0x00020000);
}

__device__ __forceinline__ BufRsrc window_rsrc(const void *p) { return slab_rsrc(p, 0xfffff000ull); }   // no range: every access through it is guarded

// Fetch of a k step: limbs 4 sidx + 2h and + 1, coefficients 2r, 2r+1.  `soff` (uniform) = byte offset of limb 4 sidx, coefficient coef0
// of the polynomial inside the slab; `lane_off` = ((2h) << sh) + 16 r for EVERY step: one VGPR of addressing per kernel, the rest is SALU.
// Limbs past the polynomial's last read whatever follows it in the slab (the next polynomial, or zeros past the end): their rows of
// the constant matrix are zero.
__device__ __forceinline__ void fetch_step(StepRegs &x, BufRsrc rs, unsigned soff, unsigned sh, unsigned lane_off) {
  x.lo = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, soff, GPQ_STREAM_LOAD_AUX);
  x.hi = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, soff + (1u << sh), GPQ_STREAM_LOAD_AUX);
}

// The k steps of a product, statically unrolled over a ring of R slots: step s lives in slot (S0 + s) % R and the slot is refilled --
// fetch(slot, S0 + s + R): the stream position R further on, in this product, the next one or the next group -- as soon as the step has
// been turned into fragments.  The constant fragments of step s + 1 are read from LDS while step s multiplies (two fragment sets: a
// ds_read followed at once by its s_waitcnt and two MFMAs leaves the matrix pipe idle for the LDS latency four times per step).  A
// scheduling barrier per step keeps the compiler from lifting a whole group's fragment reads and offset arithmetic to the top (spills);
// a runtime loop over rounds would bring back register copies of the ring behind a vmcnt wait.
template <int NT, int R, int KS, int S0, class Fetch>
__device__ __forceinline__ void crt_steps(v16i (&acc)[2][NT], StepRegs (&ring)[R], const v4i *bl /* LDS: the product's fragments, + lane */, Fetch fetch) {
  v4i bf[2][NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) bf[0][q] = bl[q * 64];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const StepRegs x = ring[(S0 + s) % R];
    fetch(ring[(S0 + s) % R], S0 + s + R);
    if (s + 1 < KS) {
#pragma unroll
      for (int q = 0; q < NT; ++q) bf[(s + 1) & 1][q] = bl[((size_t)(s + 1) * NT + q) * 64];
    }
    const v4i A0 = frag_even(x, 0x80808080u), A1 = frag_odd(x, 0x80808080u);
#if GPQ_STREAM_PROBE >= 1    /* timing probe (wrong results): no matrix-core work, the fragments are folded into one accumulator register */
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      if (s == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[0][q][e] = 0; acc[1][q][e] = 0; }
      }
      acc[0][q][0] ^= A0[0] ^ A0[1] ^ A0[2] ^ A0[3] ^ bf[s & 1][q][0];
      acc[1][q][0] ^= A1[0] ^ A1[1] ^ A1[2] ^ A1[3];
    }
    __builtin_amdgcn_sched_barrier(0);
    continue;
#endif
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const v4i b = bf[s & 1][q];
      if (s == 0) {
        v16i z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0;
        acc[0][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, A0, z, 0, 0, 0);
        acc[1][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, A1, z, 0, 0, 0);
      } else {
        acc[0][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, A0, acc[0][q], 0, 0, 0);
        acc[1][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, A1, acc[1][q], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// byte columns -> words: lane (r, h) ends up with the 4 NT words of coefficient 2r + h; the columns from WL on start their own carry chain (F)
template <int NT, int WL>
__device__ __forceinline__ void fold_columns(const v16i (&acc)[2][NT], uint64_t (&V)[4 * NT]) {
  int64_t carry = 0;
#pragma unroll
  for (int q = 0; q < NT; ++q)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int64_t L, H;
      swap_halves(horner4(acc[0][q][4 * w], acc[0][q][4 * w + 1], acc[0][q][4 * w + 2], acc[0][q][4 * w + 3]),
                  horner4(acc[1][q][4 * w], acc[1][q][4 * w + 1], acc[1][q][4 * w + 2], acc[1][q][4 * w + 3]), L, H);
      if (4 * q + w == WL) carry = 0;
      V[4 * q + w] = fold_word(L, H, carry);
      if (w == 3) __builtin_amdgcn_sched_barrier(0);      // a row tile of accumulators at a time: read out all at once they spill
    }
}

// columns of word idx alone (the streaming epilogues below fold one word at a time so that the words never all exist next to the accumulators)
template <int NT>
__device__ __forceinline__ void halves_at(const v16i (&acc)[2][NT], int idx, int64_t &L, int64_t &H) {
  const int q = idx / 4, w = idx % 4;
  swap_halves(horner4(acc[0][q][4 * w], acc[0][q][4 * w + 1], acc[0][q][4 * w + 2], acc[0][q][4 * w + 3]),
              horner4(acc[1][q][4 * w], acc[1][q][4 * w + 1], acc[1][q][4 * w + 2], acc[1][q][4 * w + 3]), L, H);
}
// the F columns of a CRT product: the multiple of P to take off (k, plus one when centring takes P off once more), and whether the
// fraction sits in the window [1/2 - 2^-38, 1/2) that the estimate cannot decide   (bridge_reconstruct_low_mfma's rule, unchanged)
__device__ __forceinline__ unsigned crt_multiple(uint64_t F0, uint64_t F1, uint64_t kf0, uint64_t kf1, bool &ambiguous) {
  const u128 F = (((u128)F1 << 64) | F0) + (((u128)kf1 << 64) | kf0);
  const uint64_t f1 = (uint64_t)(F >> 64);
  ambiguous = ((f1 >> 2) & ((1ull << 38) - 1)) == ((1ull << 37) - 1);
  const unsigned mult = (unsigned)(f1 >> 40) + (unsigned)((f1 >> 39) & 1);
  return mult < 64 ? mult : 64;
}

// F = words WL, WL + 1 (their own carry chain) + Kf -> the multiple of P and the window flag
template <int NT, int WL>
__device__ __forceinline__ unsigned multiple_of(const v16i (&acc)[2][NT], uint64_t kf0, uint64_t kf1, bool &ambiguous) {
  int64_t L, H, fc = 0;
  halves_at<NT>(acc, WL, L, H);
  const uint64_t F0 = fold_word(L, H, fc);
  halves_at<NT>(acc, WL + 1, L, H);
  const uint64_t F1 = fold_word(L, H, fc);
  return crt_multiple(F0, F1, kf0, kf1, ambiguous);
}

// V -= row `mult` of the table of multiples (LDS)
template <int WL, int NV>
__device__ __forceinline__ void take_multiple(uint64_t (&V)[NV], const uint64_t *pml, unsigned mult) {
  const uint64_t *__restrict__ P = pml + (size_t)mult * WL;
  uint64_t borrow = 0;
#pragma unroll
  for (int j = 0; j < WL; ++j) {
    const u128 t = (u128)V[j] - P[j] - borrow;
    V[j] = (uint64_t)t;
    borrow = (uint64_t)(t >> 64) & 1;
  }
}

// mpi_smod by 2^logq on the low WL words: sign-extend from bit logq - 1
template <int WL, int NV>
__device__ __forceinline__ void sign_extend(uint64_t (&V)[NV], unsigned logq) {
  const int sw = (int)((logq - 1) >> 6);
  const unsigned up = 63 - ((logq - 1) & 63);
  uint64_t ext = 0;
#pragma unroll
  for (int j = 0; j < WL; ++j) if (j == sw) ext = (uint64_t)((int64_t)(V[j] << up) >> up);
  const uint64_t qsign = (uint64_t)((int64_t)ext >> 63);
#pragma unroll
  for (int j = 0; j < WL; ++j) V[j] = j < sw ? V[j] : (j == sw ? ext : qsign);
}

// Words j, j+1 of the coefficient of every lane (2r + h) -> lane (r, 0): word j of coefficients 2r, 2r+1; lane (r, 1): word j+1 of both.
__device__ __forceinline__ v4u pair_for_store(uint64_t a, uint64_t b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(a >> 32), (unsigned)(b >> 32), false, false);
  return v4u{lo[0], hi[0], lo[1], hi[1]};
}
// the inverse: a 16-byte load of word j + h, coefficients 2r, 2r+1 -> words j, j+1 of the lane's own coefficient
__device__ __forceinline__ void pair_from_load(v4u v, uint64_t &a, uint64_t &b) {
  const auto lo = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
  a = ((uint64_t)hi[0] << 32) | lo[0];
  b = ((uint64_t)hi[1] << 32) | lo[1];
}

// One row tile (4 primes) of the rns_decompose product over the fragments X of a group (bridge_decompose_mfma's arithmetic), split
// in two so that callers can issue tile q + 1's MFMAs before the integer epilogue of tile q (the matrix pipe then works under the VALU).
// The per-prime scalars of the tile (p_j, Kq_j, c_j: uniform, scalar loads) are fetched with the product, a tile ahead of their use.
// They are read through the CONSTANT address space: next to the kernel's buffer stores the compiler cannot prove a global table
// unclobbered, reads it with vector loads -- and waits for them with vmcnt(0), i.e. for every store and prefetch in flight.
struct PkTile { uint64_t p[4], kq[4]; unsigned c[4]; };
typedef const __attribute__((address_space(4))) uint64_t *ConstTable;
template <int KSD>
__device__ __forceinline__ void decompose_tile_product(const v4i *dl /* LDS + lane */, const uint64_t *__restrict__ pk, unsigned q, const v4i (&X)[2][KSD],
                                                       v16i &acc0, v16i &acc1, PkTile &t) {
  ConstTable pkc = (ConstTable)pk;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const unsigned j = 4 * q + w;
    t.p[w] = pkc[3 * j]; t.kq[w] = pkc[3 * j + 1]; t.c[w] = (unsigned)pkc[3 * j + 2];
  }
#pragma unroll
  for (int s = 0; s < KSD; ++s) {
    const v4i cf = dl[((size_t)q * KSD + s) * 64];
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
}
// ... and its epilogue: x mod p_j for the primes j = 4q .. 4q+3, stored as limbs j, j+1 pairs (lane half h stores limb j + h for
// coefficients 2r, 2r+1).  out_off = (h << sh) + 16 r.
// lazy: leave the residue in (0, 3p) -- what the forward transform that reads it accepts (ntt_kernels.hpp: its first stage takes x, y < 4p in
// every butterfly class) -- instead of canonicalising it (7 of the ~39 VALU instructions per residue).
__device__ __forceinline__ void decompose_tile_finish(const PkTile &t, unsigned q, const v16i &acc0, const v16i &acc1, BufRsrc rs_out,
                                                      unsigned out_off, unsigned sh, unsigned h, unsigned dim, bool lazy) {
  uint64_t res[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    int64_t L, H;
    swap_halves(horner4(acc0[4 * w], acc0[4 * w + 1], acc0[4 * w + 2], acc0[4 * w + 3]),
                horner4(acc1[4 * w], acc1[4 * w + 1], acc1[4 * w + 2], acc1[4 * w + 3]), L, H);
    const int c = (int)t.c[w];
    const int Hh = (int)(H >> 27);
    const uint64_t Hl = (uint64_t)H & 0x7ffffffu;
    uint64_t v = (Hl << 32) + (uint64_t)L + t.kq[w];
    v = (uint64_t)((int64_t)(-c) * Hh + (int64_t)v);               // in (0, 3p)
    res[w] = lazy ? v : canon_fold(v, t.p[w], (uint32_t)c);
  }
#pragma unroll
  for (int w = 0; w < 4; w += 2) {
    const unsigned j = 4 * q + w;
    const v4u o = pair_for_store(res[w], res[w + 1]);
    if (j + h < dim) __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, out_off, j << sh, GPQ_STREAM_STORE_AUX);
  }
}
// all NT row tiles of a group, software-pipelined by one
template <int KSD>
__device__ __forceinline__ void decompose_tiles(const v4i *dl, const uint64_t *__restrict__ pk, unsigned NT, const v4i (&X)[2][KSD], BufRsrc rs_out,
                                                unsigned out_off, unsigned sh, unsigned h, unsigned dim, bool lazy) {
  v16i pa0, pa1, pb0, pb1;
  PkTile ta, tb;
  decompose_tile_product<KSD>(dl, pk, 0, X, pa0, pa1, ta);
  for (unsigned q = 0; q < NT; q += 2) {
    if (q + 1 < NT) decompose_tile_product<KSD>(dl, pk, q + 1, X, pb0, pb1, tb);
    __builtin_amdgcn_sched_barrier(0);
    decompose_tile_finish(ta, q, pa0, pa1, rs_out, out_off, sh, h, dim, lazy);
    __builtin_amdgcn_sched_barrier(0);
    if (q + 1 < NT) {
      if (q + 2 < NT) decompose_tile_product<KSD>(dl, pk, q + 2, X, pa0, pa1, ta);
      __builtin_amdgcn_sched_barrier(0);
      decompose_tile_finish(tb, q + 1, pb0, pb1, rs_out, out_off, sh, h, dim, lazy);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---------------------------------------------------------------------------
// CRT(d2hat) -> rns_decompose  (src/he-mult.c:140 feeding :59): poly_rns2mpi's fast path on the limbs of the tensor stage, then the W
// words of every coefficient -- still in the registers of the lane that finished it -- become the B fragments of bridge_decompose_mfma's
// product over the key switch's limbs.  Constants: get_recon_mfma's matrix padded to KS steps, get_decomp_mfma's at KSD steps.
// ---------------------------------------------------------------------------
struct CrtDecomposeArgs {
  const uint64_t *slab;      // [polys][dimA][n]   y_d = d2hat_d (P_A/p_d)^-1 (the tensor stage's inverse pass scaled them)
  size_t slab_bytes;         // (< 4 GB: the host checks)
  uint64_t *out;             // [polys][dimB][n]
  const v4i *cfrag;          // [KS][NT][64]
  const uint64_t *kc;        // [WL + 2]: Kf in the last two
  const uint64_t *pm;        // [65][WL]
  const v4i *dfrag;          // [NTD][KSD][64]
  const uint64_t *pk;        // [4 NTD][3]: p_j, Kq_j, c_j
  unsigned char *redo;       // [polys][n]
  unsigned *wave_any;        // [waves of the launch]
  unsigned dimA, dimB, NTD, logn, logq, W, total_groups;
  unsigned dimS;             // limbs per polynomial of `out` (its stride); = dimB unless a launch writes a sub-range of the limbs (A/B builds: GPQ_CRT_SPLIT)
  unsigned force;            // tests: also flag every coefficient whose index is a multiple of it (the exact kernels must then give the same words)
  unsigned lazy;             // residues out in (0, 3p) for the forward transform that follows (decompose_tile_finish)
};

template <int WL, int KS, int KSD, int R>
__global__ __launch_bounds__(512) void bridge_crt_decompose(CrtDecomposeArgs a) {
  constexpr int NT = (8 * WL + 14 + 31) / 32;
  static_assert(KS % R == 0, "the ring must divide the steps of a group");
  static_assert(4 * KSD <= 4 * NT, "the words of the value come out of the folded columns");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4i *Cl = reinterpret_cast<v4i *>(smem);
  v4i *Dl = Cl + KS * NT * 64;
  const unsigned nD = a.NTD * KSD * 64;
  uint64_t *pml = reinterpret_cast<uint64_t *>(Dl + nD);
  stage_lds(Cl, a.cfrag, (unsigned)(KS * NT * 64));
  stage_lds(Dl, a.dfrag, nD);
  stage_lds(pml, a.pm, 65u * WL);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned r = lane & 31, h = lane >> 5, lane16 = r * 16;
  const unsigned lg = a.logn - 6, sh = a.logn + 3;
  const unsigned nwaves = gridDim.x * 8, pw = blockIdx.x * 8 + wave;
  const uint64_t kf0 = a.kc[WL], kf1 = a.kc[WL + 1];
  const unsigned lane_off = ((2 * h) << sh) + lane16;
  const BufRsrc rs_in = slab_rsrc(a.slab, a.slab_bytes);
  auto src_of = [&](unsigned g) {                          // byte offset of limb 0 of the group's polynomial at its first coefficient
    const unsigned poly = g >> lg, coef0 = (g & ((1u << lg) - 1)) << 6;
    return ((poly * a.dimA) << sh) + coef0 * 8;
  };
  StepRegs ring[R];
  if (pw < a.total_groups) {
    const unsigned b0 = src_of(pw);
#pragma unroll
    for (int j = 0; j < R; ++j) fetch_step(ring[j], rs_in, b0 + ((4u * j) << sh), sh, lane_off);
  }
  unsigned any = 0;
  for (unsigned g = pw; g < a.total_groups; g += nwaves) {
    const unsigned poly = g >> lg, coef0 = (g & ((1u << lg) - 1)) << 6;
    const unsigned bn = src_of(g + nwaves < a.total_groups ? g + nwaves : g);           // next group (or this one again: harmless reads)
    const unsigned bc = src_of(g);
    const unsigned lo = opaque_v(lane);
    const v4i *cl = Cl + lo, *dl = Dl + lo;
    v16i acc[2][NT];
    // stream position p (0 .. KS-1: this group, from KS on: the next one) -> its slot's refill
    auto refill = [&](StepRegs &x, int p) { fetch_step(x, rs_in, (p < KS ? bc + ((4u * p) << sh) : bn + ((4u * (p - KS)) << sh)), sh, lane_off); };
    crt_steps<NT, R, KS, 0>(acc, ring, cl, refill);
    uint64_t V[4 * NT];
    fold_columns<NT, WL>(acc, V);
    bool ambiguous;
    const unsigned mult = crt_multiple(V[WL], V[WL + 1], kf0, kf1, ambiguous);
    take_multiple<WL>(V, pml, mult);
    sign_extend<WL>(V, a.logq);
    if (a.force) ambiguous = ambiguous || (coef0 + 2 * r + h) % a.force == 0;
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)ambiguous, window_rsrc(a.redo + ((size_t)poly << a.logn) + coef0), 2 * r + h, 0, 0);
    any |= __builtin_amdgcn_ballot_w64(ambiguous) != 0;
    // the words as signed bytes (the top byte of the top word is signed as it is; words past W do not exist), then the lane halves trade
    // words 4s+2, 4s+3 of the even coefficient against words 4s, 4s+1 of the odd one: the B fragments of both tiles
    v4i X[2][KSD];
#pragma unroll
    for (int s = 0; s < KSD; ++s) {
      unsigned wlo[4], whi[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned j = 4 * s + e;
        const uint64_t v = V[j];
        const unsigned mlo = j < a.W ? 0x80808080u : 0u, mhi = j + 1 < a.W ? 0x80808080u : (j < a.W ? 0x00808080u : 0u);
        wlo[e] = j < a.W ? (unsigned)v ^ mlo : 0u;
        whi[e] = j < a.W ? (unsigned)(v >> 32) ^ mhi : 0u;
      }
      const auto p0l = __builtin_amdgcn_permlane32_swap(wlo[0], wlo[2], false, false);
      const auto p0h = __builtin_amdgcn_permlane32_swap(whi[0], whi[2], false, false);
      const auto p1l = __builtin_amdgcn_permlane32_swap(wlo[1], wlo[3], false, false);
      const auto p1h = __builtin_amdgcn_permlane32_swap(whi[1], whi[3], false, false);
      X[0][s] = v4i{(int)p0l[0], (int)p0h[0], (int)p1l[0], (int)p1h[0]};
      X[1][s] = v4i{(int)p0l[1], (int)p0h[1], (int)p1l[1], (int)p1h[1]};
    }
    const BufRsrc rs_out = window_rsrc(a.out + ((size_t)poly * a.dimS << a.logn) + coef0);       // limb 0 of the polynomial at the group's first coefficient
    const unsigned out_off = (h << sh) + lane16;           // + limb j (uniform): lane half h stores limb j + h for coefficients 2r, 2r+1
    decompose_tiles<KSD>(dl, a.pk, a.NTD, X, rs_out, out_off, sh, h, a.dimB, a.lazy != 0);
  }
  if (lane == 0) a.wave_any[pw] = any;
}

// ---------------------------------------------------------------------------
// The relinearisation tail (src/he-mult.c:67-77, src/he-automorphism.c:68-76) as ONE product (bridge_mfma.hpp, frac_bits = 104) with its
// addend inside: the 16-word columns are the fixed-point number 2^104 x / P, and
//   DCRT   the addend is poly_rns2mpi(d0hat | d1hat) (src/he-mult.c:139,141): its limbs are MORE ROWS OF THE SAME PRODUCT, with the
//          constants (P_A / p_d) 2^104 mod 2^1024 -- the columns then hold 2^104 (x / P + d): the fraction bits (x mod P) / P that mpi_rdiv
//          rounds on are untouched (d's part is a multiple of 2^104), the bits above are floor(x / P) + d.  One accumulator set, one
//          column folding, no quotient parked in registers while a second product runs; d0, d1 never reach memory as words.  The
//          multiple of P_A that centring d takes off comes from 14 more fixed-point columns (144 .. 157, next to the tail's 128 .. 141);
//   !DCRT  he_swk's addend, a big slab (or none), read 16 bytes per lane.
//   out = smod(floor(x/P) + round + d, 2^logq)
// ---------------------------------------------------------------------------
struct TailStreamArgs {
  const uint64_t *chat;      // [polys][dimB][n]   limbs weighted for the CRT over all dimB limbs (ScaledInverse, get_tail_direct)
  const uint64_t *dhat;      // DCRT: [polys][dimA][n] limbs of d weighted for the CRT over dimA limbs
  size_t chat_bytes, dhat_bytes;   // (< 4 GB each: the host checks)
  Two<const uint64_t> addend;// !DCRT: [polys][W][n] or null places
  Two<uint64_t> out;         // [polys][W][n]
  const v4i *tfrag;          // [KST][5][64]
  const uint64_t *tkc;       // [18]: Kf in the last two
  const uint64_t *tpm;       // [65][16]
  const v4i *dfrag;          // DCRT: [KSD][5][64]   (get_addend_rows)
  const uint64_t *dkc;       // [18]
  const uint64_t *dpm;       // [65][16]
  unsigned char *redo, *tie, *amb_clear;   // [polys][n]
  unsigned *wave_any;
  unsigned dimB, dimA, logn, W, logq, total_groups;
  unsigned force;            // tests: also flag every coefficient whose index is a multiple of it
  unsigned rs_s, rs_logq;    // gpq_he_mul_rs: he_rs (src/he-rescale.c:45-48) on the way out -- rounding division by 2^rs_s (1 .. 63), centred mod 2^rs_logq; 0 = off
};

// V = floor(V / 2^s) + [V mod 2^s > 2^(s-1)] on a sign-extended value of NV words, 1 <= s <= 63 (mpi_rdiv, src/types.c:115-128: floor for
// negatives, ties down) -- bridge_kernels.hpp's rescale_coefficient on registers
template <int NV>
__device__ __forceinline__ void rescale_words(uint64_t (&V)[NV], unsigned s) {
  uint64_t carry = ((V[0] >> (s - 1)) & 1) & (uint64_t)((V[0] & ((1ull << (s - 1)) - 1)) != 0);
  const uint64_t sign = (uint64_t)((int64_t)V[NV - 1] >> 63);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const uint64_t hi = j + 1 < NV ? V[j + 1] : sign;
    const uint64_t v = (V[j] >> s) | (hi << (64 - s));
    const uint64_t v1 = v + carry;
    carry = v1 < v;
    V[j] = v1;
  }
}

constexpr int TAIL_DF_WORD = 18;     // the addend's fixed-point columns are words 18, 19 of the folded columns (the tail's own: 16, 17)

template <int KST, int KSD, bool DCRT, int R>
__global__ __launch_bounds__(512) void bridge_tail_stream(TailStreamArgs a) {
  constexpr int WL = 16, NT = 5;
  constexpr int NSTEP = KST + (DCRT ? KSD : 0);
  static_assert(NSTEP % R == 0 && R <= KST, "the ring must divide the steps of a group");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4i *Tl = reinterpret_cast<v4i *>(smem);                                   // [NSTEP][5][64]: the tail's rows, then the addend's
  uint64_t *tpml = reinterpret_cast<uint64_t *>(Tl + NSTEP * NT * 64);
  uint64_t *dpml = tpml + 65 * kTailRow;
  stage_lds(Tl, a.tfrag, (unsigned)(KST * NT * 64));
  stage_multiples<WL>(tpml, a.tpm);
  if (DCRT) {
    stage_lds(Tl + KST * NT * 64, a.dfrag, (unsigned)(KSD * NT * 64));
    stage_multiples<WL>(dpml, a.dpm);
  }
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned r = lane & 31, h = lane >> 5, lane16 = r * 16;
#if GPQ_STREAM_PROBE >= 3     /* timing probe: read the slabs as if every group's limbs were contiguous (512 bytes per limb, group after group) */
  const unsigned lg = a.logn - 6, sh = 9, sh_out = a.logn + 3;
#else
  const unsigned lg = a.logn - 6, sh = a.logn + 3, sh_out = sh;
#endif
  const unsigned nwaves = gridDim.x * 8, pw = blockIdx.x * 8 + wave;
  const uint64_t tkf0 = a.tkc[WL], tkf1 = a.tkc[WL + 1];
  const uint64_t dkf0 = DCRT ? a.dkc[WL] : 0, dkf1 = DCRT ? a.dkc[WL + 1] : 0;
  const unsigned lane_off = ((2 * h) << sh) + lane16;
  const BufRsrc rs_c = slab_rsrc(a.chat, a.chat_bytes), rs_d = slab_rsrc(DCRT ? a.dhat : a.chat, DCRT ? a.dhat_bytes : 0);
  auto chat_of = [&](unsigned g) {                         // byte offsets inside the slabs (< 4 GB: the host checks)
    const unsigned poly = g >> lg, coef0 = (g & ((1u << lg) - 1)) << 6;
#if GPQ_STREAM_PROBE >= 3
    return (g * a.dimB) << 9;
#endif
    return ((poly * a.dimB) << sh) + coef0 * 8;
  };
  auto dhat_of = [&](unsigned g) {
    const unsigned poly = g >> lg, coef0 = (g & ((1u << lg) - 1)) << 6;
#if GPQ_STREAM_PROBE >= 3
    return (g * a.dimA) << 9;
#endif
    return ((poly * a.dimA) << sh) + coef0 * 8;
  };
  StepRegs ring[R];
  if (pw < a.total_groups) {
    const unsigned c0 = chat_of(pw);
#pragma unroll
    for (int j = 0; j < R; ++j) fetch_step(ring[j], rs_c, c0 + ((4u * j) << sh), sh, lane_off);       // (R <= KST)
  }
  unsigned any = 0;
  for (unsigned g = pw; g < a.total_groups; g += nwaves) {
    const unsigned poly = g >> lg, coef0 = (g & ((1u << lg) - 1)) << 6;
    const unsigned gn = g + nwaves < a.total_groups ? g + nwaves : g;
    const unsigned cc = chat_of(g), cn = chat_of(gn), dc = dhat_of(g);
    const v4i *tl = Tl + opaque_v(lane);
    // stream position p of a group: the tail's KST steps, then the addend's KSD; from NSTEP on: the next group's
    auto refill = [&](StepRegs &x, int p) {
      const bool nxt = p >= NSTEP;
      const int q = nxt ? p - NSTEP : p;
      if (!DCRT || q < KST) fetch_step(x, rs_c, (nxt ? cn : cc) + ((4u * q) << sh), sh, lane_off);
      else fetch_step(x, rs_d, dc + ((4u * (q - KST)) << sh), sh, lane_off);
    };
    v16i acc[2][NT];
    crt_steps<NT, R, NSTEP, 0>(acc, ring, tl, refill);
    // The fixed-point columns first (k, the centring, the windows), then the 16 words one at a time: V = 2^104 (x / P [+ d]) in two's
    // complement, x / P underestimated by less than 2^-40.  Its fraction (low 104 bits) decides mpi_rdiv's rounding: within 2^-38 below 1
    // (the estimate may have borrowed from the integer part) or below 1/2 -> the exact path; the bits above are floor(x / P) [+ d].
    bool ambiguous, amb_d = false;
    const unsigned mult = multiple_of<NT, WL>(acc, tkf0, tkf1, ambiguous);
    const uint64_t *__restrict__ P = tpml + (size_t)mult * kTailRow;
    const uint64_t *__restrict__ PD = dpml;
    if (DCRT) PD = dpml + (size_t)multiple_of<NT, TAIL_DF_WORD>(acc, dkf0, dkf1, amb_d) * kTailRow;
    ambiguous = ambiguous || amb_d;
    uint64_t Q[14];
    int64_t carry = 0;
    uint64_t borrow = 0, prev = 0, cr = 0;
#if GPQ_STREAM_PROBE >= 2
#pragma unroll
    for (int j = 0; j < 14; ++j) Q[j] = (uint64_t)(unsigned)acc[j & 1][j % NT][0] + P[j] + (DCRT ? PD[j] : 0);
#else
#pragma unroll
    for (int j = 0; j < WL; ++j) {
      int64_t L, H;
      halves_at<NT>(acc, j, L, H);
      u128 t = (u128)fold_word(L, H, carry) - P[j] - borrow;
      if (DCRT) t -= PD[j];
      const uint64_t v = (uint64_t)t;
      borrow = (uint64_t)(0 - (uint64_t)(t >> 64));                                   // 0, 1 or 2
      if (j == 1) {
        const uint64_t f40 = v & ((1ull << 40) - 1), top38 = f40 >> 2;
        ambiguous = ambiguous || top38 == ((1ull << 38) - 1) || top38 == ((1ull << 37) - 1);
        cr = f40 >> 39;
      }
      if (j >= 2) Q[j - 2] = (prev >> 40) | (v << 24);                                // floor(x / P) [+ d] mod 2^896
      prev = v;
      if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);                              // a row tile of accumulators at a time
    }
#endif
    if (DCRT) {
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const u128 t = (u128)Q[j] + cr;
        Q[j] = (uint64_t)t; cr = (uint64_t)(t >> 64);
      }
    } else {
      const uint64_t *ad = a.addend.at(poly, (size_t)a.W << a.logn);                 // uniform per group
      const BufRsrc rs_ad = window_rsrc(ad ? ad + coef0 : a.chat);
      uint64_t dd[14];
#pragma unroll
      for (int j = 0; j < 14; j += 2) {
        v4u v = v4u{0, 0, 0, 0};
        if (ad && j + h < a.W) v = __builtin_amdgcn_raw_buffer_load_b128(rs_ad, (h << sh_out) + lane16, (unsigned)j << sh_out, 0);
        pair_from_load(v, dd[j], dd[j + 1]);
      }
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const u128 t = (u128)Q[j] + cr + dd[j];
        Q[j] = (uint64_t)t; cr = (uint64_t)(t >> 64);
      }
    }
    sign_extend<14>(Q, a.logq);
    if (a.rs_s) {                                                                     // (uniform) he_rs on the way out
      rescale_words<14>(Q, a.rs_s);
      sign_extend<14>(Q, a.rs_logq);
    }
    const size_t flag_at = ((size_t)poly << a.logn) + coef0;
    if (a.force) ambiguous = ambiguous || (coef0 + 2 * r + h) % a.force == 0;
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)ambiguous, window_rsrc(a.redo + flag_at), 2 * r + h, 0, 0);
#if GPQ_STREAM_PROBE != 4 && GPQ_STREAM_PROBE != 5
    if (a.tie) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)0, window_rsrc(a.tie + flag_at), 2 * r + h, 0, 0);
    if (a.amb_clear) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)0, window_rsrc(a.amb_clear + flag_at), 2 * r + h, 0, 0);
#endif
    any |= __builtin_amdgcn_ballot_w64(ambiguous) != 0;
    const BufRsrc rs_out = window_rsrc(a.out.at(poly, (size_t)a.W << a.logn) + coef0);
#pragma unroll
    for (int j = 0; j < 14; j += 2) {
      const v4u o = pair_for_store(Q[j], Q[j + 1]);
#if GPQ_STREAM_PROBE == 5
      if (o[0] == 0x12345678u && j + h < a.W)
#else
      if (j + h < a.W)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, (h << sh_out) + lane16, (unsigned)j << sh_out, GPQ_STREAM_STORE_AUX);        // (W <= 14: the host checks)
    }
  }
  if (lane == 0) a.wave_any[pw] = any;
}

}  // namespace gpq
