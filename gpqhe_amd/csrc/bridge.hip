// bridge.hip -- host side of the MPI <-> RNS bridge: CRT constants per prefix of
// the prime chain (rns_init, src/precomp.c:266-293) and the C ABI entry points
// gpq_rns_decompose / gpq_rns_reconstruct / gpq_poly_mul / gpq_he_rs.
#include "../../include/gpqhe_hip.h"
#include "engine_internal.hpp"
#include "bridge_kernels.hpp"
#include "bridge_mfma.hpp"
#include "bridge_stream.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

using namespace gpq;

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return gpq_fail(GPQ_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace {

typedef unsigned __int128 u128h;
typedef std::vector<uint64_t> Big;  // little-endian words, unsigned

// Dynamic LDS above 64 KB needs the function attribute, once per (kernel, device).
struct LdsRaised {
  bool dev[64] = {};
  int raise(const void *fn, int bytes) {
    int d = 0;
    HIP_TRY(hipGetDevice(&d));
    if (d < 0 || d >= 64 || !dev[d]) {
      HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      if (d >= 0 && d < 64) dev[d] = true;
    }
    return GPQ_OK;
  }
};


void mul_small(Big &a, uint64_t m) {
  uint64_t carry = 0;
  for (auto &w : a) { u128h t = (u128h)w * m + carry; w = (uint64_t)t; carry = (uint64_t)(t >> 64); }
  if (carry) a.push_back(carry);
}
uint64_t divmod_small(Big &a, uint64_t m) {  // a <- floor(a/m), returns a mod m
  uint64_t rem = 0;
  for (size_t i = a.size(); i-- > 0;) { u128h t = ((u128h)rem << 64) | a[i]; a[i] = (uint64_t)(t / m); rem = (uint64_t)(t % m); }
  while (a.size() > 1 && a.back() == 0) a.pop_back();
  return rem;
}
uint64_t mod_small(const Big &a, uint64_t m) { Big t = a; return divmod_small(t, m); }
void shr1(Big &a) {
  for (size_t i = 0; i < a.size(); ++i) a[i] = (a[i] >> 1) | (i + 1 < a.size() ? a[i + 1] << 63 : 0);
}
uint64_t powm(uint64_t b, uint64_t e, uint64_t m) {
  uint64_t r = 1;
  while (e) { if (e & 1) r = (uint64_t)((u128h)r * b % m); b = (uint64_t)((u128h)b * b % m); e >>= 1; }
  return r;
}
void put(std::vector<uint64_t> &dst, size_t off, const Big &v, size_t words) {
  for (size_t j = 0; j < words; ++j) dst[off + j] = j < v.size() ? v[j] : 0;
}

const int kWP[] = {8, 16, 32, 48, 56};

// ---- host big integers for the general-modulus path (sizes of a few thousand bits) ----
int cmp_big(const Big &a, const Big &b) {
  size_t na = a.size(), nb = b.size();
  while (na > 1 && a[na - 1] == 0) --na;
  while (nb > 1 && b[nb - 1] == 0) --nb;
  if (na != nb) return na < nb ? -1 : 1;
  for (size_t i = na; i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
void sub_big(Big &a, const Big &b) {  // a -= b, a >= b
  uint64_t bor = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    const u128h d = (u128h)a[i] - (i < b.size() ? b[i] : 0) - bor;
    a[i] = (uint64_t)d; bor = (uint64_t)(d >> 64) & 1;
  }
}
Big floor_pow2_div(unsigned bits, const Big &m) {  // floor(2^bits / m), restoring division bit by bit
  Big q((bits + 64) / 64, 0), rem(m.size() + 1, 0);
  for (int b = (int)bits; b >= 0; --b) {
    uint64_t c = b == (int)bits ? 1 : 0;           // shift rem left by one, bring in the next dividend bit
    for (size_t i = 0; i < rem.size(); ++i) { const uint64_t n = rem[i] >> 63; rem[i] = (rem[i] << 1) | c; c = n; }
    if (cmp_big(rem, m) >= 0) { sub_big(rem, m); q[b >> 6] |= 1ull << (b & 63); }
  }
  return q;
}

// CRT constants of primes first .. first+dim-1: what struct rns_ctx node dim-1 holds for first = 0
// (src/poly.h:35-38); sub-ranges serve the exact division of he_relin.
int get_basis(gpq_ctx *c, unsigned first, unsigned dim, gpq_bridge_basis **out) {
  const auto key = std::make_pair(first, dim);
  auto it = c->cache->bases.find(key);
  if (it != c->cache->bases.end()) { *out = &it->second; return GPQ_OK; }
  if (dim < 1 || first + dim > c->nprimes || dim > 63)
    return gpq_fail(GPQ_ERR_INVALID, "bridge: limbs %u..%u outside the chain of %u (at most 63 per basis)", first, first + dim, c->nprimes);
  Big P{1};
  for (unsigned d = 0; d < dim; ++d) mul_small(P, c->p[first + d]);  // src/precomp.c:274-277
  int WP = 0;
  for (int w : kWP) if ((size_t)w >= P.size()) { WP = w; break; }
  if (!WP) return gpq_fail(GPQ_ERR_UNSUPPORTED, "bridge: P of %u limbs needs %zu words", dim, P.size());
  gpq_bridge_basis b;
  b.first = first; b.dim = dim; b.WP = WP; b.pbits = 64 * (unsigned)(P.size() - 1) + (64 - __builtin_clzll(P.back()));
  std::vector<uint64_t> phat((size_t)dim * WP), pinv(dim), pmult((size_t)6 * (WP + 1)), phalf(WP + 1);
  for (unsigned d = 0; d < dim; ++d) {
    const uint64_t pd = c->p[first + d];
    Big q = P;
    divmod_small(q, pd);                                             // phat_d = P / p_d   :287
    put(phat, (size_t)d * WP, q, WP);
    pinv[d] = powm(mod_small(q, pd), pd - 2, pd);                    // :288-289
  }
  Big h = P; shr1(h);                                                // P_2 = floor(P/2)   :278
  put(phalf, 0, h, WP + 1);
  Big m = P;
  for (int k = 5; k >= 0; --k) { put(pmult, (size_t)k * (WP + 1), m, WP + 1); mul_small(m, 2); }  // P,2P,..,32P at rows 5..0
  DeviceScope on_device(c->device);
  HIP_TRY(gpq_table_malloc(c, (void **)&b.d_phat, phat.size() * 8));
  HIP_TRY(gpq_table_malloc(c, (void **)&b.d_phat_inv, pinv.size() * 8));
  HIP_TRY(gpq_table_malloc(c, (void **)&b.d_pmult, pmult.size() * 8));
  HIP_TRY(gpq_table_malloc(c, (void **)&b.d_phalf, phalf.size() * 8));
  HIP_TRY(hipMemcpy(b.d_phat, phat.data(), phat.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.d_phat_inv, pinv.data(), pinv.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.d_pmult, pmult.data(), pmult.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.d_phalf, phalf.data(), phalf.size() * 8, hipMemcpyHostToDevice));
  std::vector<uint64_t> inv128(2 * (size_t)dim);
  for (unsigned d = 0; d < dim; ++d) {
    const u128h q = ~(u128h)0 / c->p[first + d];                     // floor(2^128 / p_d): p_d does not divide 2^128
    inv128[2 * d] = (uint64_t)q; inv128[2 * d + 1] = (uint64_t)(q >> 64);
  }
  HIP_TRY(gpq_table_malloc(c, (void **)&b.d_inv128, inv128.size() * 8));
  HIP_TRY(hipMemcpy(b.d_inv128, inv128.data(), inv128.size() * 8, hipMemcpyHostToDevice));
  b.h_phat_inv = pinv;
  b.h_P = P;
  b.h_phat = phat;
  *out = &(c->cache->bases[key] = b);
  return GPQ_OK;
}

int get_relin(gpq_ctx *c, unsigned dimP, unsigned dimB, gpq_relin_tables **out) {
  const auto key = std::make_pair(dimP, dimB);
  auto it = c->cache->relins.find(key);
  if (it != c->cache->relins.end()) { *out = &it->second; return GPQ_OK; }
  gpq_bridge_basis *bp;
  int rc = get_basis(c, 0, dimP, &bp);
  if (rc) return rc;
  std::vector<uint64_t> pinv(dimB - dimP);
  for (unsigned d = dimP; d < dimB; ++d) pinv[d - dimP] = powm(mod_small(bp->h_P, c->p[d]), c->p[d] - 2, c->p[d]);
  gpq_relin_tables t;
  HIP_TRY(gpq_table_malloc(c, (void **)&t.d_pinv, pinv.size() * 8));
  HIP_TRY(hipMemcpy(t.d_pinv, pinv.data(), pinv.size() * 8, hipMemcpyHostToDevice));
  *out = &(c->cache->relins[key] = t);
  return GPQ_OK;
}

// The context's per-limb table with the constants of the LAST inverse stage -- n^-1 and winv[1] n^-1 (src/ntt.c:71-72 folded into the
// stage, ntt_kernels.hpp gs_last) -- multiplied by (P/p_d)^-1 mod p_d for the limbs d of basis b: an inverse transform that reads it
// hands out y_d = ahat_d * phat_invmp_d, the first product of rns_reconstruct (src/rns.c:66-68), for free -- one modular multiply
// and a canonicalisation per (coefficient, limb) less in the CRT kernels that follow (`prescaled`).  Limbs outside the basis keep n^-1.
int get_scaled_tabs(gpq_ctx *c, gpq_bridge_basis *b, const LimbTab **out) {
  if (!b->d_tabs_scaled) {
    std::vector<LimbTab> t = c->h_tabs;
    auto pair_of = [](uint64_t w, uint64_t p) { return TwS{p - w, p - (uint64_t)(((u128h)w << 31) % p)}; };
    for (unsigned d = 0; d < b->dim; ++d) {
      LimbTab &e = t[b->first + d];
      const uint64_t p = e.k.p, s = b->h_phat_inv[d];
      e.ninv = (uint64_t)((u128h)e.ninv * s % p);
      e.winv1_ninv = (uint64_t)((u128h)e.winv1_ninv * s % p);
      if (b->first + d < c->nsplit_tables) { e.ninv_s = pair_of(e.ninv, p); e.winv1_ninv_s = pair_of(e.winv1_ninv, p); }
    }
    DeviceScope on_device(c->device);
    HIP_TRY(gpq_table_malloc(c, (void **)&b->d_tabs_scaled, t.size() * sizeof(LimbTab)));
    HIP_TRY(hipMemcpy(b->d_tabs_scaled, t.data(), t.size() * sizeof(LimbTab), hipMemcpyHostToDevice));
  }
  *out = b->d_tabs_scaled;
  return GPQ_OK;
}
// for the duration of one gpq_he_mul_tensor / gpq_keyswitch call
struct ScaledInverse {
  gpq_ctx *c;
  ScaledInverse(gpq_ctx *ctx, const LimbTab *tabs) : c(ctx) { c->inv_tabs_override = tabs; }
  ~ScaledInverse() { c->inv_tabs_override = nullptr; }
};
inline bool can_prescale(const gpq_ctx *c) { return c->prescale && c->logn > 12 && !c->h_tabs.empty(); }   // the two-pass transforms only (small rings: gpq_invntt)

// grid of a masked kernel of `threads` threads: over the waves of the launch that wrote its mask (FlagScope), or over (n, polys)
inline dim3 masked_grid(const FlagScope &sc, unsigned threads, unsigned n, unsigned polys) {
  if (sc.wave_any) return dim3((sc.waves + threads / 64 - 1) / (threads / 64));
  return dim3((n + threads - 1) / threads, polys);
}
template <int WP>
void launch_exact(const ReconstructArgs &a, unsigned n, unsigned batch, hipStream_t s) {
  hipLaunchKernelGGL((bridge_reconstruct<WP>), masked_grid(a.scope, 128, n, batch), dim3(128), 0, s, a);
}
template <int WL>
void launch_low(const ReconstructArgs &a, unsigned WPstride, unsigned char *redo, unsigned n, unsigned batch, hipStream_t s) {
  hipLaunchKernelGGL((bridge_reconstruct_low<WL>), dim3((n + 255) / 256, batch), dim3(256), 0, s, a, WPstride, redo);
}

// balanced base-256 digits of a little-endian multiword value, `nd` digits (the final carry is dropped: mod 256^nd)
void balanced_digits(const uint64_t *words, size_t nwords, int8_t *out, size_t nd) {
  unsigned carry = 0;
  for (size_t i = 0; i < nd; ++i) {
    const unsigned byte = i / 8 < nwords ? (unsigned)((words[i / 8] >> (8 * (i % 8))) & 0xff) : 0;
    const unsigned t = byte + carry;
    if (t >= 128) { out[i] = (int8_t)((int)t - 256); carry = 1; } else { out[i] = (int8_t)t; carry = 0; }
  }
}

// Constant matrix, offsets and multiples of `Pw` for bridge_reconstruct_low_mfma<WL>: the contraction sum_d y_d * weight_d mod 2^(64 WL)
// plus the fixed-point columns F = sum_d y_d floor(2^104 / p_d).  For a CRT, weight_d = P/p_d and Pw = P (get_recon_mfma); for the
// one-product relinearisation tail, weight_d = floor(Pi' 2^104 / p_d) and Pw = Pi' 2^104 (get_tail_direct).
int build_recon_mfma(gpq_ctx *c, const std::vector<uint64_t> &primes, const std::vector<uint64_t> &weight_inv, const std::vector<Big> &weight,
                     const Big &Pw, int WL, gpq_recon_mfma *tp, unsigned KSpad = 0, unsigned fcol0 = 0) {
  gpq_recon_mfma &t = *tp;
  const unsigned dim = (unsigned)primes.size(), NT = (8 * WL + 14 + 31) / 32, ncol = 32 * NT;
  t.KS = std::max((dim + 3) / 4, KSpad);                 // (KSpad: zero rows up to the k steps a bridge_stream.hpp instantiation runs)
  if (!fcol0) fcol0 = 8u * WL;                           // first of the 14 fixed-point columns (the addend's rows inside the tail's product: 144)
  t.lds_bytes = (size_t)t.KS * NT * 1024 + (size_t)t.KS * 64;
  if (t.lds_bytes <= 156 * 1024) {
    std::vector<int8_t> bf((size_t)t.KS * NT * 1024, 0);
    std::vector<uint64_t> lk((size_t)t.KS * 8, 0), kc(WL + 2, 0), pm((size_t)65 * WL, 0);
    std::vector<int8_t> beta(8 * (size_t)WL), phi(8);
    Big sum_phat(WL, 0);
    u128h sum_inv = 0;
    for (unsigned d = 0; d < dim; ++d) {
      const uint64_t pd = primes[d];
      lk[2 * (size_t)d] = pd;
      lk[2 * (size_t)d + 1] = weight_inv[d];
      const Big &ph = weight[d];
      balanced_digits(ph.data(), ph.size() < (size_t)WL ? ph.size() : (size_t)WL, beta.data(), beta.size());
      const uint64_t inv = (uint64_t)((((u128h)1) << 104) / pd);                       // < 2^46
      balanced_digits(&inv, 1, phi.data(), 8);
      uint64_t cy = 0;                                                                   // sum_phat += weight_d mod 2^(64 WL)
      for (int j = 0; j < WL; ++j) {
        const u128h s2 = (u128h)sum_phat[j] + ((size_t)j < ph.size() ? ph[j] : 0) + cy;
        sum_phat[j] = (uint64_t)s2; cy = (uint64_t)(s2 >> 64);
      }
      sum_inv += inv;
      for (unsigned i = 0; i < 8; ++i) {
        const unsigned k = 8 * d + i, s = k / 32, h = (k % 32) / 16, tt = k % 16;
        for (unsigned col = 0; col < ncol; ++col) {
          int8_t v = 0;
          if (col < 8u * WL) { if (col >= i) v = beta[col - i]; }
          else if (col >= fcol0) { const unsigned m = col - fcol0; if (m >= i && m - i < 8 && m < 14) v = phi[m - i]; }
          if (!v) continue;
          const unsigned nt = col / 32, lane = 32 * h + col % 32;
          bf[(((size_t)s * NT + nt) * 64 + lane) * 16 + tt] = v;
        }
      }
    }
    // offsets of the signed bytes: 0x8080..80 * sum weight_d (mod 2^(64 WL)) and 0x8080..80 * sum inv_d
    Big kcS = sum_phat;
    mul_small(kcS, 0x8080808080808080ull);
    for (int j = 0; j < WL; ++j) kc[j] = (size_t)j < kcS.size() ? kcS[j] : 0;
    const u128h lo = (u128h)(uint64_t)sum_inv * 0x8080808080808080ull;
    const u128h hi = (u128h)(uint64_t)(sum_inv >> 64) * 0x8080808080808080ull;
    const u128h kf = lo + (hi << 64);
    kc[WL] = (uint64_t)kf; kc[WL + 1] = (uint64_t)(kf >> 64);
    Big mP{0};
    for (unsigned m = 0; m <= 64; ++m) {
      uint64_t bw = 0;                                                                   // (m Pw - Kc) mod 2^(64 WL)
      for (int j = 0; j < WL; ++j) {
        const u128h d2 = (u128h)((size_t)j < mP.size() ? mP[j] : 0) - kc[j] - bw;
        pm[(size_t)m * WL + j] = (uint64_t)d2; bw = (uint64_t)(d2 >> 64) & 1;
      }
      Big nxt(std::max(mP.size(), Pw.size()) + 1, 0);                                    // mP += Pw
      uint64_t cy = 0;
      for (size_t j = 0; j < nxt.size(); ++j) {
        const u128h s2 = (u128h)(j < mP.size() ? mP[j] : 0) + (j < Pw.size() ? Pw[j] : 0) + cy;
        nxt[j] = (uint64_t)s2; cy = (uint64_t)(s2 >> 64);
      }
      mP = nxt;
    }
    DeviceScope on_device(c->device);
    HIP_TRY(gpq_table_malloc(c, (void **)&t.d_bfrag, bf.size()));
    HIP_TRY(gpq_table_malloc(c, (void **)&t.d_lk, lk.size() * 8));
    HIP_TRY(gpq_table_malloc(c, (void **)&t.d_kc, kc.size() * 8));
    HIP_TRY(gpq_table_malloc(c, (void **)&t.d_pm, pm.size() * 8));
    HIP_TRY(hipMemcpy(t.d_bfrag, bf.data(), bf.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.d_lk, lk.data(), lk.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.d_kc, kc.data(), kc.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.d_pm, pm.data(), pm.size() * 8, hipMemcpyHostToDevice));
  }
  return GPQ_OK;
}

// ... for the CRT over basis b
int get_recon_mfma(gpq_ctx *c, gpq_bridge_basis *b, int WL, gpq_recon_mfma **out, unsigned KSpad = 0) {
  if (KSpad <= (b->dim + 3) / 4) KSpad = 0;
  const int key = WL + 1000 * (int)KSpad;
  auto it = b->mfma.find(key);
  if (it != b->mfma.end()) { *out = &it->second; return GPQ_OK; }
  gpq_recon_mfma t;
  std::vector<uint64_t> primes(b->dim);
  std::vector<Big> weight(b->dim);
  for (unsigned d = 0; d < b->dim; ++d) {
    primes[d] = c->p[b->first + d];
    weight[d].assign(b->h_phat.begin() + (size_t)d * b->WP, b->h_phat.begin() + (size_t)(d + 1) * b->WP);
  }
  if (int rc = build_recon_mfma(c, primes, b->h_phat_inv, weight, b->h_P, WL, &t, KSpad)) return rc;
  *out = &(b->mfma[key] = t);
  return GPQ_OK;
}

// Rows that put poly_rns2mpi(dhat) INSIDE the one-product relinearisation tail (bridge_stream.hpp, DCRT): the CRT over basis b with every
// weight shifted up by the tail's 104 fraction bits -- weight_d = (P/p_d) 2^104, P_w = P 2^104, all modulo 2^1024 -- and the fixed-point
// columns for its own multiple of P at 144 .. 157 (the tail's are 128 .. 141).  Padded to KSpad k steps.
int get_addend_rows(gpq_ctx *c, gpq_bridge_basis *b, unsigned KSpad, gpq_recon_mfma **out) {
  const int key = 100016 + 1000 * (int)KSpad;
  auto it = b->mfma.find(key);
  if (it != b->mfma.end()) { *out = &it->second; return GPQ_OK; }
  auto shifted = [](const Big &v) {                      // v * 2^104
    Big r = v;
    r.insert(r.begin(), 0);
    mul_small(r, 1ull << 40);
    return r;
  };
  std::vector<uint64_t> primes(b->dim);
  std::vector<Big> weight(b->dim);
  for (unsigned d = 0; d < b->dim; ++d) {
    primes[d] = c->p[b->first + d];
    weight[d] = shifted(Big(b->h_phat.begin() + (size_t)d * b->WP, b->h_phat.begin() + (size_t)(d + 1) * b->WP));
    weight[d].resize(weight[d].size() < 16 ? 16 : weight[d].size(), 0);
  }
  gpq_recon_mfma t;
  if (int rc = build_recon_mfma(c, primes, b->h_phat_inv, weight, shifted(b->h_P), 16, &t, KSpad, 144)) return rc;
  *out = &(b->mfma[key] = t);
  return GPQ_OK;
}

template <int WL>
int launch_low_mfma(const ReconMfmaArgs &a, size_t lds, hipStream_t s);
int launch_low_mfma16(const ReconMfmaArgs &a, size_t lds, hipStream_t s) { return launch_low_mfma<16>(a, lds, s); }
template <int WL>
int launch_low_mfma(const ReconMfmaArgs &a, size_t lds, hipStream_t s) {
  static LdsRaised raised;
  if (int rc = raised.raise(reinterpret_cast<const void *>(&bridge_reconstruct_low_mfma<WL>), 156 * 1024)) return rc;
  unsigned blocks = 256;                                   // one 8-wave workgroup per CU (LDS), persistent over the groups
  if (blocks > (a.total_groups + 7) / 8) blocks = (a.total_groups + 7) / 8;
  hipLaunchKernelGGL((bridge_reconstruct_low_mfma<WL>), dim3(blocks), dim3(512), lds, s, a);
  return GPQ_OK;
}

// Options of the relinearisation tail: `only` restricts the exact kernel to flagged coefficients; `prescaled` says the slab
// already holds y_d; `addend`/`rflags` ask the matrix-core fast path to finish the tail itself (then *fused is set and the
// coefficients it handed to the exact kernel -- c->d_redo -- still need bridge_addround).
struct ReconExtra {
  const unsigned char *only = nullptr;
  bool prescaled = false;
  Two<const uint64_t> addend{nullptr, nullptr, ~0u};
  const unsigned char *rflags = nullptr;
  bool *fused = nullptr;
  uint64_t *big_b = nullptr;      // the polynomials from `split` on are written here instead (Two<>, bridge_kernels.hpp)
  unsigned split = ~0u;
  bool exact_only = false;        // skip the fast paths: the exact kernel alone (restricted by `only`)
  FlagScope scope = kNoScope;     // with `only` and exact_only: the bridge_stream.hpp launch that wrote the mask
};

// per-coefficient "redo exactly" flags of the fast CRT paths
int ensure_redo(gpq_ctx *c, size_t flags, hipStream_t s) {
  if (flags <= c->redo_cap) return GPQ_OK;
  // Growing inside a stream capture would put hipMalloc into the graph; and a graph captured earlier keeps the old
  // pointer, so outgrown buffers are retired (freed with the context), never freed here.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return gpq_fail(GPQ_ERR_INVALID, "the first call at a new batch size allocates scratch: run it once outside stream capture");
  DeviceScope on_device(c->device);
  unsigned char *grown = nullptr;
  HIP_TRY(hipMalloc((void **)&grown, flags));
  if (c->d_redo) c->retired.push_back(c->d_redo);
  c->d_redo = grown;
  c->redo_cap = flags;
  return GPQ_OK;
}

// arguments of the exact kernel bridge_reconstruct<b->WP> for one call (launch_reconstruct, and the fused fallback kernels behind bridge_stream.hpp)
ReconstructArgs exact_args(const gpq_ctx *c, const gpq_bridge_basis *b, uint64_t *big, unsigned Wout, const uint64_t *slab, unsigned slab_dim,
                           unsigned slab_first, unsigned logq, bool centre, unsigned char *tie, unsigned logn, const ReconExtra &x) {
  return ReconstructArgs{c->d_tabs, slab, Two<uint64_t>{big, x.big_b, x.split}, b->d_phat, b->d_phat_inv, b->d_pmult, b->d_phalf, tie, x.only, b->d_inv128,
                         b->dim, logn, Wout, logq, b->first, slab_dim, slab_first, centre ? 1u : 0u, x.prescaled ? 1u : 0u, x.scope};
}

int launch_reconstruct(gpq_ctx *c, const gpq_bridge_basis *b, uint64_t *big, unsigned Wout, const uint64_t *slab, unsigned slab_dim,
                       unsigned slab_first, unsigned batch, unsigned logq, bool centre, unsigned char *tie, hipStream_t s, int logn_override = -1,
                       const ReconExtra &x = ReconExtra()) {
  const unsigned logn = logn_override < 0 ? c->logn : (unsigned)logn_override, n = 1u << logn;
  const Two<uint64_t> bigs{big, x.big_b, x.split};
  ReconstructArgs a = exact_args(c, b, big, Wout, slab, slab_dim, slab_first, logq, centre, tie, logn, x);
  if (x.fused) *x.fused = false;
  // fast path: centred result modulo a power of two that needs fewer words than P has
  const unsigned need = (logq + 63) / 64;
  // (the centring threshold floor(P/2)/P differs from 1/2 by 1/(2P): negligible against the 2^-61 slack only for large P)
  const bool fast = logq && centre && !c->exact_crt && need + 1 < (unsigned)b->WP && need <= 16 && b->pbits >= 160;
  if (fast && !x.exact_only) {
    if (int rc = ensure_redo(c, (size_t)batch << logn, s)) return rc;
    ProfScope prof(c, GPQ_K_RECONSTRUCT, s);
    bool done = false;
    if (c->bridge_mfma && logn >= 6 && b->dim >= 4) {      // CRT sum as bytes x constant matrix on the matrix cores
      const int WL = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 7 ? 7 : need <= 10 ? 10 : need <= 14 ? 14 : 16;
      gpq_recon_mfma *t;
      int rc = get_recon_mfma(c, const_cast<gpq_bridge_basis *>(b), WL, &t);
      if (rc) return rc;
      if (t->d_bfrag) {
        const unsigned gpp = n >> 6;
        ReconMfmaArgs m{slab, bigs, (const v4i *)t->d_bfrag, t->d_lk, t->d_kc, t->d_pm, c->d_redo, tie, b->dim, t->KS, logn, Wout, logq,
                        slab_dim, slab_first, gpp, gpp * batch, x.addend, x.rflags, x.prescaled ? 1u : 0u, 0u, nullptr};
        if (x.fused && x.rflags) *x.fused = true;
        switch (WL) {
          case 1: rc = launch_low_mfma<1>(m, t->lds_bytes, s); break;
          case 2: rc = launch_low_mfma<2>(m, t->lds_bytes, s); break;
          case 4: rc = launch_low_mfma<4>(m, t->lds_bytes, s); break;
          case 7: rc = launch_low_mfma<7>(m, t->lds_bytes, s); break;
          case 10: rc = launch_low_mfma<10>(m, t->lds_bytes, s); break;
          case 14: rc = launch_low_mfma<14>(m, t->lds_bytes, s); break;
          default: rc = launch_low_mfma<16>(m, t->lds_bytes, s); break;
        }
        if (rc) return rc;
        done = true;
      }
    }
    if (done) {}
    else if (need <= 1) launch_low<1>(a, b->WP, c->d_redo, n, batch, s);
    else if (need <= 2) launch_low<2>(a, b->WP, c->d_redo, n, batch, s);
    else if (need <= 4) launch_low<4>(a, b->WP, c->d_redo, n, batch, s);
    else if (need <= 7) launch_low<7>(a, b->WP, c->d_redo, n, batch, s);     // q up to 2^448 (reference default 2^438)
    else if (need <= 10) launch_low<10>(a, b->WP, c->d_redo, n, batch, s);
    else if (need <= 14) launch_low<14>(a, b->WP, c->d_redo, n, batch, s);   // q up to 2^896 (headline 2^850)
    else launch_low<16>(a, b->WP, c->d_redo, n, batch, s);
    a.only = c->d_redo;   // exact kernel below redoes only the flagged coefficients
    a.scope = kNoScope;   // (the fast kernels above leave no per-wave words)
  }
  ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s);
  switch (b->WP) {
    case 8: launch_exact<8>(a, n, batch, s); break;
    case 16: launch_exact<16>(a, n, batch, s); break;
    case 32: launch_exact<32>(a, n, batch, s); break;
    case 48: launch_exact<48>(a, n, batch, s); break;
    case 56: launch_exact<56>(a, n, batch, s); break;
    default: return gpq_fail(GPQ_ERR_UNSUPPORTED, "reconstruct: WP=%d", b->WP);
  }
  return GPQ_OK;
}

// balanced base-256 digits of v < 2^63: v = sum_b d_b 256^b, d_b in [-128, 127] (top digit small and positive)
void balanced8(uint64_t v, int8_t out[8]) {
  const uint64_t t = v + 0x0080808080808080ull;
  for (int b = 0; b < 7; ++b) out[b] = (int8_t)(((t >> (8 * b)) & 0xff) ^ 0x80);
  out[7] = (int8_t)(t >> 56);
}

constexpr size_t kMfmaLdsMax = 96 * 1024;   // of the CU's 160 KB: one workgroup always fits, two when the tables are small

// constant matrix of bridge_decompose_mfma for the primes limb0 .. limb0+dim-1 and W-word inputs
int get_decomp_mfma(gpq_ctx *c, unsigned limb0, unsigned dim, unsigned W, gpq_decomp_mfma **out, unsigned KSforce = 0) {
  const unsigned KSnat = W <= 4 ? 1 : W <= 8 ? 2 : W <= 16 ? 4 : 8;
  if (KSforce <= KSnat) KSforce = 0;
  const auto key = std::make_pair(std::make_pair(limb0, dim), W + 1000 * KSforce);
  auto it = c->cache->decomps.find(key);
  if (it != c->cache->decomps.end()) { *out = &it->second; return GPQ_OK; }
  gpq_decomp_mfma t;
  const unsigned KB = 8 * W;
  t.KS = KSforce ? KSforce : KSnat;                      // (KSforce: zero columns up to the k steps a bridge_stream.hpp instantiation runs)
  t.NT = (dim + 3) / 4;
  t.lds_bytes = (size_t)t.NT * t.KS * 1024 + (size_t)t.NT * 96;
  if (t.lds_bytes <= kMfmaLdsMax) {
    std::vector<int8_t> bf((size_t)t.NT * t.KS * 1024, 0);
    std::vector<uint64_t> pk((size_t)t.NT * 12, 0);
    std::vector<int8_t> dig((size_t)KB * 8);
    for (unsigned j = 0; j < dim; ++j) {
      const uint64_t p = c->p[limb0 + j];
      uint64_t T = 1, K = 0;                               // 256^k mod p ; sum_{k < KB-1} 256^k mod p
      for (unsigned k = 0; k < KB; ++k) {
        balanced8(T, &dig[(size_t)k * 8]);
        if (k + 1 < KB) K = (K + T) % p;
        T = (uint64_t)(((u128h)T << 8) % p);
      }
      K = (uint64_t)(((u128h)K << 7) % p);                 // 128 * sum
      const uint64_t off = 1ull << 50;
      pk[3 * (size_t)j] = p;
      pk[3 * (size_t)j + 1] = off + (K + p - off % p) % p;
      pk[3 * (size_t)j + 2] = p - (1ull << 59);
      const unsigned nt = j / 4, pq = j % 4;
      for (unsigned k = 0; k < KB; ++k) {
        const unsigned s = k / 32, h = (k % 32) / 16, tt = k % 16;
        for (unsigned b = 0; b < 8; ++b) {
          const unsigned lane = 32 * h + 8 * pq + b;       // B[k][col]: lane = (col, h), byte tt
          bf[(((size_t)nt * t.KS + s) * 64 + lane) * 16 + tt] = dig[(size_t)k * 8 + b];
        }
      }
    }
    DeviceScope on_device(c->device);
    HIP_TRY(gpq_table_malloc(c, (void **)&t.d_bfrag, bf.size()));
    HIP_TRY(gpq_table_malloc(c, (void **)&t.d_pk, pk.size() * 8));
    HIP_TRY(hipMemcpy(t.d_bfrag, bf.data(), bf.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.d_pk, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
  }
  *out = &(c->cache->decomps[key] = t);
  return GPQ_OK;
}

template <int KS>
int launch_decompose_mfma_t(const DecomposeMfmaArgs &a, size_t lds, hipStream_t s) {
  static LdsRaised raised;
  if (int rc = raised.raise(reinterpret_cast<const void *>(&bridge_decompose_mfma<KS>), (int)kMfmaLdsMax)) return rc;
  unsigned per_cu = (unsigned)((160 * 1024) / lds);     // workgroups a CU's LDS holds; the registers allow 3
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  unsigned blocks = 256 * per_cu;
  if (blocks > (a.total_groups + 3) / 4) blocks = (a.total_groups + 3) / 4;
  hipLaunchKernelGGL((bridge_decompose_mfma<KS>), dim3(blocks), dim3(256), lds, s, a);
  return GPQ_OK;
}

int launch_decompose(gpq_ctx *c, uint64_t *slab, const BigSources &big, unsigned W, unsigned limb0, unsigned dim, unsigned batch, hipStream_t s, bool lazy = false);
int launch_decompose(gpq_ctx *c, uint64_t *slab, const uint64_t *big, unsigned W, unsigned limb0, unsigned dim, unsigned batch, hipStream_t s, bool lazy = false) {
  return launch_decompose(c, slab, one_source(big), W, limb0, dim, batch, s, lazy);
}
// `batch` polynomials in all, `big.per` from each source slab in turn, written one after another to `slab`.  lazy: the residues may stay in
// (0, 3p) (matrix-core kernel only; for slabs that go straight into a two-pass forward transform: gpq_he_mul's own decompositions)
int launch_decompose(gpq_ctx *c, uint64_t *slab, const BigSources &big, unsigned W, unsigned limb0, unsigned dim, unsigned batch, hipStream_t s, bool lazy) {
  ProfScope prof(c, GPQ_K_DECOMPOSE, s);
  if (c->bridge_mfma && c->logn >= 6 && W <= 32 && dim >= 4) {
    gpq_decomp_mfma *t;
    int rc = get_decomp_mfma(c, limb0, dim, W, &t);
    if (rc) return rc;
    if (t->d_bfrag) {
      const unsigned gpp = c->n >> 6;
      DecomposeMfmaArgs m{big, slab, (const v4i *)t->d_bfrag, t->d_pk, W, dim, c->logn, t->NT, gpp, gpp * batch, lazy ? 1u : 0u};
      switch (t->KS) {
        case 1: return launch_decompose_mfma_t<1>(m, t->lds_bytes, s);
        case 2: return launch_decompose_mfma_t<2>(m, t->lds_bytes, s);
        case 4: return launch_decompose_mfma_t<4>(m, t->lds_bytes, s);
        default: return launch_decompose_mfma_t<8>(m, t->lds_bytes, s);
      }
    }
  }
  DecomposeArgs a{c->d_tabs, big, slab, W, dim, c->logn, limb0};
  const dim3 grid((c->n + 255) / 256, batch), block(256);
  if (W <= 4) hipLaunchKernelGGL((bridge_decompose<4>), grid, block, 0, s, a);
  else if (W <= 7) hipLaunchKernelGGL((bridge_decompose<7>), grid, block, 0, s, a);
  else if (W <= 14) hipLaunchKernelGGL((bridge_decompose<14>), grid, block, 0, s, a);
  else if (W <= 16) hipLaunchKernelGGL((bridge_decompose<16>), grid, block, 0, s, a);
  else if (W <= 32) hipLaunchKernelGGL((bridge_decompose<32>), grid, block, 0, s, a);
  else return gpq_fail(GPQ_ERR_UNSUPPORTED, "decompose: W=%u words (max 32)", W);
  return GPQ_OK;
}

// rns_decompose of the coefficients marked in `only` alone (integer-VALU kernel; the exact fallback behind bridge_crt_decompose)
int launch_decompose_masked(gpq_ctx *c, uint64_t *slab, const uint64_t *big, unsigned W, unsigned limb0, unsigned dim, unsigned batch,
                            const unsigned char *only, const FlagScope &scope, hipStream_t s) {
  ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s);
  DecomposeArgs a{c->d_tabs, one_source(big), slab, W, dim, c->logn, limb0, only, scope};
  const dim3 grid = masked_grid(scope, 256, c->n, batch), block(256);
  if (W <= 4) hipLaunchKernelGGL((bridge_decompose<4>), grid, block, 0, s, a);
  else if (W <= 7) hipLaunchKernelGGL((bridge_decompose<7>), grid, block, 0, s, a);
  else if (W <= 14) hipLaunchKernelGGL((bridge_decompose<14>), grid, block, 0, s, a);
  else if (W <= 16) hipLaunchKernelGGL((bridge_decompose<16>), grid, block, 0, s, a);
  else if (W <= 32) hipLaunchKernelGGL((bridge_decompose<32>), grid, block, 0, s, a);
  else return gpq_fail(GPQ_ERR_UNSUPPORTED, "decompose: W=%u words (max 32)", W);
  return GPQ_OK;
}

// ---- bridge_stream.hpp: launchers ----
constexpr unsigned kStreamBlocks = 256, kStreamWaves = 8;          // one 8-wave workgroup per CU, persistent over the groups
constexpr size_t kStreamLdsMax = 156 * 1024;

int ensure_wave_any(gpq_ctx *c, hipStream_t s) {
  if (c->d_wave_any) return GPQ_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return gpq_fail(GPQ_ERR_INVALID, "the first call allocates scratch: run it once outside stream capture");
  DeviceScope on_device(c->device);
  HIP_TRY(hipMalloc((void **)&c->d_wave_any, kStreamBlocks * kStreamWaves * sizeof(unsigned)));
  // on the LAUNCH stream: a plain hipMemset goes to the null stream, which a non-blocking stream (torch's side streams, the peer lane's) does not
  // wait for -- the producer kernel's flag words could be zeroed after it wrote them (tests/test_stream_bridge_gpu.py: a fresh peer lane)
  HIP_TRY(hipMemsetAsync(c->d_wave_any, 0, kStreamBlocks * kStreamWaves * sizeof(unsigned), s));
  return GPQ_OK;
}
inline unsigned stream_blocks(unsigned total_groups) {
  const unsigned need = (total_groups + kStreamWaves - 1) / kStreamWaves;
  return need < kStreamBlocks ? need : kStreamBlocks;
}
inline bool stream_fast_ok(const gpq_ctx *c, const gpq_bridge_basis *b, unsigned logq) {   // launch_reconstruct's conditions for the fast CRT path
  const unsigned need = (logq + 63) / 64;
  return logq && !c->exact_crt && c->bridge_mfma && c->logn >= 6 && b->dim >= 4 && need + 1 < (unsigned)b->WP && b->pbits >= 160;
}

template <int WL, int KS, int KSD, int R>
int launch_crt_decompose_t(const CrtDecomposeArgs &a, size_t lds, unsigned blocks, hipStream_t s) {
  static LdsRaised raised;
  if (int rc = raised.raise(reinterpret_cast<const void *>(&bridge_crt_decompose<WL, KS, KSD, R>), (int)kStreamLdsMax)) return rc;
  hipLaunchKernelGGL((bridge_crt_decompose<WL, KS, KSD, R>), dim3(blocks), dim3(512), lds, s, a);
  return GPQ_OK;
}

// src/he-mult.c:140 + :59 -- poly_rns2mpi(d2hat) mod 2^logq and rns_decompose of it over dimB limbs in one kernel; `scratch` = W words per
// coefficient for the coefficients the exact kernels redo.  *done = false: shape or settings outside the instantiations (caller runs the two kernels).
int crt_decompose_stream(gpq_ctx *c, gpq_bridge_basis *bA, uint64_t *out, const uint64_t *slab, uint64_t *scratch, unsigned W, unsigned dimA,
                         unsigned dimB, unsigned logq, unsigned polys, hipStream_t s, bool *done) {
  *done = false;
  const unsigned need = (logq + 63) / 64;
  if (!c->stream_bridge || !stream_fast_ok(c, bA, logq) || W < need || dimB < 4) return GPQ_OK;
  int WL, KS, KSD;
  if (W <= 7 && dimA <= 16) { WL = 7; KS = 4; KSD = 2; }
  else if (W <= 14 && dimA <= 32) { WL = 14; KS = 8; KSD = 4; }
  else return GPQ_OK;
  gpq_recon_mfma *tr;
  gpq_decomp_mfma *td;
  int rc;
  if ((rc = get_recon_mfma(c, bA, WL, &tr, KS)) || (rc = get_decomp_mfma(c, 0, dimB, W, &td, KSD))) return rc;
  if (!tr->d_bfrag || !td->d_bfrag || tr->KS != (unsigned)KS || td->KS != (unsigned)KSD) return GPQ_OK;
  const unsigned NT = (8 * WL + 14 + 31) / 32;
  const size_t lds = (size_t)KS * NT * 1024 + (size_t)td->NT * KSD * 1024 + (size_t)65 * WL * 8;
  if (lds > kStreamLdsMax) return GPQ_OK;
  if ((rc = ensure_redo(c, (size_t)polys << c->logn, s)) || (rc = ensure_wave_any(c, s))) return rc;
  const unsigned groups = (c->n >> 6) * polys, blocks = stream_blocks(groups);
  const size_t slab_bytes = ((size_t)polys * dimA << c->logn) * 8;
  if (slab_bytes >= 0xfffff000ull) return GPQ_OK;
  CrtDecomposeArgs a{slab, slab_bytes, out, (const v4i *)tr->d_bfrag, tr->d_kc, tr->d_pm, (const v4i *)td->d_bfrag, td->d_pk, c->d_redo, c->d_wave_any,
                     dimA, dimB, td->NT, c->logn, logq, W, groups, dimB, c->debug_force_redo, (c->lazy_decompose && c->logn > 12) ? 1u : 0u};
  {
    ProfScope prof(c, GPQ_K_CRT_DECOMPOSE, s);
#if defined(GPQ_CRT_SPLIT) && GPQ_CRT_SPLIT
    // A/B build (HISTORY.md R6.5): the output limbs over TWO launches -- the first td->NT / 2 row tiles, then the rest -- each
    // repeating the CRT of d2hat (second read of the slab, second product); constants, flags and per-wave words are the same in both.
    const unsigned nt1 = (td->NT + 1) / 2, l1 = 4 * nt1 < dimB ? 4 * nt1 : dimB;
    CrtDecomposeArgs a1 = a, a2 = a;
    a1.NTD = nt1; a1.dimB = l1;
    a2.NTD = td->NT - nt1; a2.dimB = dimB - l1;
    a2.out = out + ((size_t)l1 << c->logn); a2.dfrag = a.dfrag + (size_t)nt1 * KSD * 64; a2.pk = a.pk + (size_t)3 * l1;
    const size_t lds1 = (size_t)KS * NT * 1024 + (size_t)a1.NTD * KSD * 1024 + (size_t)65 * WL * 8, lds2 = (size_t)KS * NT * 1024 + (size_t)a2.NTD * KSD * 1024 + (size_t)65 * WL * 8;
    if (WL == 7) rc = launch_crt_decompose_t<7, 4, 2, 4>(a1, lds1, blocks, s);
    else rc = launch_crt_decompose_t<14, 8, 4, 4>(a1, lds1, blocks, s);
    if (!rc && a2.NTD) {
      if (WL == 7) rc = launch_crt_decompose_t<7, 4, 2, 4>(a2, lds2, blocks, s);
      else rc = launch_crt_decompose_t<14, 8, 4, 4>(a2, lds2, blocks, s);
    }
#else
    if (WL == 7) rc = launch_crt_decompose_t<7, 4, 2, 4>(a, lds, blocks, s);
    else rc = launch_crt_decompose_t<14, 8, 4, 4>(a, lds, blocks, s);
#endif
    if (rc) return rc;
  }
  // the coefficients in the CRT's window: exact CRT into the scratch words, integer-VALU decompose of those
  const FlagScope scope{c->d_wave_any, blocks * kStreamWaves, groups};
  ReconExtra ex;
  ex.prescaled = true; ex.exact_only = true; ex.only = c->d_redo; ex.scope = scope;
  const bool f32 = bA->WP == 32 && W > 7 && W <= 14, f16 = bA->WP == 16 && W <= 7;
  if (f32 || f16) {                                        // one launch: a thread decomposes the words it has just reconstructed
    ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s);
    const ReconstructArgs ra = exact_args(c, bA, scratch, W, slab, dimA, 0, logq, true, nullptr, c->logn, ex);
    const DecomposeArgs da{c->d_tabs, one_source(scratch), out, W, dimB, c->logn, 0, c->d_redo, scope};
    const dim3 grid = masked_grid(scope, 128, c->n, polys);
    if (f32) hipLaunchKernelGGL((bridge_fallback_crt_decompose<32, 14>), grid, dim3(128), 0, s, ra, da);
    else hipLaunchKernelGGL((bridge_fallback_crt_decompose<16, 7>), grid, dim3(128), 0, s, ra, da);
  } else {
    if ((rc = launch_reconstruct(c, bA, scratch, W, slab, dimA, 0, polys, logq, true, nullptr, s, -1, ex))) return rc;
    if ((rc = launch_decompose_masked(c, out, scratch, W, 0, dimB, polys, c->d_redo, scope, s))) return rc;
  }
  *done = true;
  return GPQ_OK;
}

int check(const gpq_ctx *c, unsigned dim, unsigned batch, const char *who) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "%s: null context", who);
  if (dim < 1 || dim > c->nprimes) return gpq_fail(GPQ_ERR_INVALID, "%s: dim=%u outside 1..%u", who, dim, c->nprimes);
  if (batch < 1) return gpq_fail(GPQ_ERR_INVALID, "%s: empty batch", who);
  // kernels launch on the calling thread's current device: it must be the one the context (its tables, the caller's slabs) lives on
  int dev = -1;
  if (hipGetDevice(&dev) == hipSuccess && dev != c->device)
    return gpq_fail(GPQ_ERR_INVALID, "%s: the context lives on device %d but the calling thread's current device is %d (gpq_set_device(gpq_ctx_device(ctx)) first)", who, c->device, dev);
  return GPQ_OK;
}
int launched(const char *who) {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GPQ_OK : gpq_fail(GPQ_ERR_HIP, "%s: launch failed: %s", who, hipGetErrorString(e));
}

}  // namespace

void gpq_bridge_release(gpq_ctx *c) {
  // the context's own mutable words
  if (c->d_redo) (void)hipFree(c->d_redo);
  c->d_redo = nullptr; c->redo_cap = 0;
  if (c->d_wave_any) (void)hipFree(c->d_wave_any);
  c->d_wave_any = nullptr;
  for (void *old : c->retired) (void)hipFree(old);
  c->retired.clear();
  // the constant cache: the owner's to free (a peer lane borrows its parent's, engine_internal.hpp)
  if (c->tables_of || !c->cache) { c->cache = nullptr; return; }
  for (auto &kv : c->cache->bases) {
    (void)hipFree(kv.second.d_phat); (void)hipFree(kv.second.d_phat_inv);
    (void)hipFree(kv.second.d_pmult); (void)hipFree(kv.second.d_phalf); (void)hipFree(kv.second.d_inv128);
    if (kv.second.d_tabs_scaled) (void)hipFree(kv.second.d_tabs_scaled);
    for (auto &m : kv.second.mfma) {
      if (m.second.d_bfrag) (void)hipFree(m.second.d_bfrag);
      if (m.second.d_lk) (void)hipFree(m.second.d_lk);
      if (m.second.d_kc) (void)hipFree(m.second.d_kc);
      if (m.second.d_pm) (void)hipFree(m.second.d_pm);
    }
  }
  for (auto &kv : c->cache->relins) {
    (void)hipFree(kv.second.d_pinv);
    for (void *q : {kv.second.d_bfrag, (void *)kv.second.d_lk, (void *)kv.second.d_pk, (void *)kv.second.d_tkp, (void *)kv.second.d_kf,
                    kv.second.d_bfrag_w, (void *)kv.second.d_pk_w, (void *)kv.second.d_tkp_w, (void *)kv.second.d_tabs_w,
                    kv.second.direct.d_bfrag, (void *)kv.second.direct.d_lk, (void *)kv.second.direct.d_kc, (void *)kv.second.direct.d_pm,
                    (void *)kv.second.d_scale, (void *)kv.second.d_unscale})
      if (q) (void)hipFree(q);
    for (auto &m : kv.second.direct_padded)
      for (void *q : {m.second.d_bfrag, (void *)m.second.d_lk, (void *)m.second.d_kc, (void *)m.second.d_pm})
        if (q) (void)hipFree(q);
  }
  for (auto &kv : c->cache->decomps) { if (kv.second.d_bfrag) (void)hipFree(kv.second.d_bfrag); if (kv.second.d_pk) (void)hipFree(kv.second.d_pk); }
  delete c->cache;
  c->cache = nullptr;
}

extern "C" int gpq_set_bridge_mfma(gpq_ctx *c, int on) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_bridge_mfma: null context");
  c->bridge_mfma = on != 0;
  return GPQ_OK;
}

extern "C" unsigned gpq_big_words(unsigned bits) { return (bits + 63) / 64; }

extern "C" uint64_t gpq_ctx_phat_invmp(gpq_ctx *c, unsigned dim, unsigned d) {
  gpq_bridge_basis *b;
  if (get_basis(c, 0, dim, &b) != GPQ_OK || d >= dim) return 0;
  return b->h_phat_inv[d];
}
extern "C" unsigned gpq_ctx_pbits(gpq_ctx *c, unsigned dim) {
  gpq_bridge_basis *b;
  return get_basis(c, 0, dim, &b) == GPQ_OK ? b->pbits : 0;
}

extern "C" int gpq_rns_decompose(gpq_ctx *c, uint64_t *slab, const uint64_t *big, unsigned W, unsigned dim, unsigned batch, void *stream) {
  int rc = check(c, dim, batch, "gpq_rns_decompose");
  if (rc) return rc;
  if (!slab || !big || W < 1) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_decompose: bad arguments");
  if ((rc = launch_decompose(c, slab, big, W, 0, dim, batch, (hipStream_t)stream))) return rc;
  return launched("gpq_rns_decompose");
}

extern "C" int gpq_rns_reconstruct(gpq_ctx *c, uint64_t *big, unsigned Wout, const uint64_t *slab, unsigned dim, unsigned batch,
                                   unsigned logq, void *stream) {
  int rc = check(c, dim, batch, "gpq_rns_reconstruct");
  if (rc) return rc;
  if (!slab || !big || Wout < 1) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct: bad arguments");
  gpq_bridge_basis *b;
  if ((rc = get_basis(c, 0, dim, &b))) return rc;
  if (logq && Wout < (logq + 63) / 64) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct: %u words cannot hold a value mod 2^%u", Wout, logq);
  if (!logq && Wout * 64 < b->pbits + 1) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct: %u words cannot hold a value mod P (%u bits)", Wout, b->pbits);
  if ((rc = launch_reconstruct(c, b, big, Wout, slab, dim, 0, batch, logq, true, nullptr, (hipStream_t)stream))) return rc;
  return launched("gpq_rns_reconstruct");
}

// rns_reconstruct for ONE coefficient (src/rns.c:60-75 is per coefficient): host residues in, host words out, value in [0, P).
extern "C" int gpq_rns_reconstruct_one(gpq_ctx *c, uint64_t *words, unsigned Wout, const uint64_t *residues, unsigned dim) {
  int rc = check(c, dim, 1, "gpq_rns_reconstruct_one");
  if (rc) return rc;
  if (!words || !residues) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct_one: null argument");
  gpq_bridge_basis *b;
  if ((rc = get_basis(c, 0, dim, &b))) return rc;
  if (Wout * 64 < b->pbits + 1) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct_one: %u words cannot hold a value mod P (%u bits)", Wout, b->pbits);
  for (unsigned d = 0; d < dim; ++d)
    if (residues[d] >= c->p[d]) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct_one: residue %u is not reduced", d);
  DeviceScope on_device(c->device);
  uint64_t *dev = nullptr;
  HIP_TRY(hipMalloc((void **)&dev, (size_t)(dim + Wout) * 8));
  hipError_t e = hipMemcpy(dev, residues, (size_t)dim * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = launch_reconstruct(c, b, dev + dim, Wout, dev, dim, 0, 1, 0, false, nullptr, nullptr, 0);
    if (rc == GPQ_OK) rc = launched("gpq_rns_reconstruct_one");
    if (rc == GPQ_OK) e = hipMemcpy(words, dev + dim, (size_t)Wout * 8, hipMemcpyDeviceToHost);
  }
  (void)hipFree(dev);
  if (e != hipSuccess) return gpq_fail(GPQ_ERR_HIP, "gpq_rns_reconstruct_one: %s", hipGetErrorString(e));
  return rc;
}

// rns_decompose for the limbs first .. first+count-1 only (src/rns.c:37-48 is per limb): slab[k][d][i], d < count.
extern "C" int gpq_rns_decompose_limbs(gpq_ctx *c, uint64_t *slab, const uint64_t *big, unsigned W, unsigned first, unsigned count,
                                       unsigned batch, void *stream) {
  int rc = check(c, count, batch, "gpq_rns_decompose_limbs");
  if (rc) return rc;
  if (!slab || !big || W < 1 || first + count > c->nprimes) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_decompose_limbs: bad arguments");
  if ((rc = launch_decompose(c, slab, big, W, first, count, batch, (hipStream_t)stream))) return rc;
  return launched("gpq_rns_decompose_limbs");
}

extern "C" size_t gpq_poly_mul_workspace_bytes(const gpq_ctx *c, unsigned dim, unsigned batch) {
  return 3ull * batch * ((size_t)dim << c->logn) * 8;
}

// poly_mul, src/poly.c:84-107, for q = 2^logq: decompose a and b to `dim` limbs, ntt, pointwise
// multiply, invntt, poly_rns2mpi.  r, a, b are big slabs of W words.
extern "C" int gpq_poly_mul(gpq_ctx *c, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned W, unsigned dim, unsigned logq,
                            unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dim, batch, "gpq_poly_mul");
  if (rc) return rc;
  if (!r || !a || !b || !workspace || !logq) return gpq_fail(GPQ_ERR_INVALID, "gpq_poly_mul: bad arguments (q must be 2^logq, logq > 0)");
  const size_t poly = (size_t)dim << c->logn;
  uint64_t *sa = (uint64_t *)workspace, *sb = sa + batch * poly, *sr = sb + batch * poly;
  if ((rc = gpq_rns_decompose(c, sa, a, W, dim, batch, stream))) return rc;
  if ((rc = gpq_rns_decompose(c, sb, b, W, dim, batch, stream))) return rc;
  if ((rc = gpq_poly_mul_rns(c, sr, sa, sb, dim, batch, stream))) return rc;
  return gpq_rns_reconstruct(c, r, W, sr, dim, batch, logq, stream);
}

// he_rs, src/he-rescale.c:33-54, for Delta = 2^logDelta and q_l = 2^logql: both polynomials of
// `batch` ciphertexts in place.  The level/nu/B bookkeeping of :36-38 stays with the caller.
extern "C" int gpq_he_rs(gpq_ctx *c, uint64_t *c0, uint64_t *c1, unsigned W, unsigned logDelta, unsigned logql, unsigned batch, void *stream) {
  if (!c || !c0 || !c1 || W < 1 || batch < 1 || logql < 1 || logql > 64 * W)
    return gpq_fail(GPQ_ERR_INVALID, "gpq_he_rs: bad arguments");
  const dim3 grid((c->n + 255) / 256, batch), block(256);
  for (uint64_t *p : {c0, c1}) {
    RescaleArgs a{p, W, c->logn, logDelta, logql};
    ProfScope prof(c, GPQ_K_RESCALE, (hipStream_t)stream);
    hipLaunchKernelGGL(bridge_rescale, grid, block, 0, (hipStream_t)stream, a);
  }
  return launched("gpq_he_rs");
}
extern "C" int gpq_he_rescale(gpq_ctx *c, uint64_t *c0, uint64_t *c1, unsigned W, unsigned logDelta, unsigned logql, unsigned batch, void *stream) {
  return gpq_he_rs(c, c0, c1, W, logDelta, logql, batch, stream);
}

// ---------------------------------------------------------------------------
// he_mul / he_swk at the big-slab level (q_l = 2^logql)
// ---------------------------------------------------------------------------
namespace {

struct TailPlan { unsigned Wr, cnt; size_t words, bytes; };

int tail_plan(gpq_ctx *c, unsigned W, unsigned dimP, unsigned dimB, unsigned polys, TailPlan *p) {
  gpq_bridge_basis *bp;
  int rc = get_basis(c, 0, dimP, &bp);
  if (rc) return rc;
  if (dimB <= dimP) return gpq_fail(GPQ_ERR_INVALID, "relin: dimB=%u must exceed dimP=%u", dimB, dimP);
  p->Wr = bp->pbits / 64 + 1;
  p->cnt = dimB - dimP;
  p->words = (size_t)polys * ((size_t)(p->Wr + 2 * p->cnt + W) << c->logn);
  p->bytes = p->words * 8 + ((size_t)polys << c->logn);
  return GPQ_OK;
}

// tables of bridge_relin_front_mfma for P = p_0..p_{dimP-1} and the limbs dimP..dimB-1
int get_relin_front(gpq_ctx *c, unsigned dimP, unsigned dimB, gpq_relin_tables *rt, gpq_bridge_basis *bp, gpq_bridge_basis *bq) {
  if (rt->front_tried) return GPQ_OK;
  rt->front_tried = true;
  const unsigned cnt = dimB - dimP;
  if (dimP < 4 || dimP > 32 || cnt < 4 || bp->pbits < 160) return GPQ_OK;
  const unsigned KS = dimP <= 8 ? 2 : dimP <= 16 ? 4 : 8, NTp = (cnt + 3) / 4, NT = NTp + 1;
  const size_t lds = (size_t)NT * KS * 1024 + (size_t)(8 * KS + 12 * NTp) * 8;
  if (lds > kMfmaLdsMax) return GPQ_OK;
  std::vector<int8_t> bf((size_t)NT * KS * 1024, 0);
  std::vector<uint64_t> lk((size_t)8 * KS, 0), pk((size_t)12 * NTp, 0), tkp((size_t)cnt * 64, 0), kf(2, 0);
  // The same tables with every constant of limb j multiplied by w_j = P^-1 (Pi'/p_j)^-1 mod p_j: with the limbs above P arriving
  // already multiplied by w_j (ScaledInverse on the key switch's inverse pass) Q's scaled residue is x'_j - (r w_j mod p_j), a
  // subtraction where the plain tables need a modular multiplication per (coefficient, limb).
  std::vector<int8_t> bfw;
  std::vector<uint64_t> pkw((size_t)12 * NTp, 0), tkpw((size_t)cnt * 64, 0), wscale(cnt, 1);
  u128h sum_inv = 0;
  for (unsigned d = 0; d < dimP; ++d) {
    const uint64_t pd = c->p[d];
    lk[2 * (size_t)d] = pd;
    lk[2 * (size_t)d + 1] = bp->h_phat_inv[d];
    const uint64_t inv = (uint64_t)((((u128h)1) << 104) / pd);
    sum_inv += inv;
    int8_t phi[8];
    balanced_digits(&inv, 1, phi, 8);
    for (unsigned i = 0; i < 8; ++i) {
      const unsigned k = 8 * d + i, s = k / 32, h = (k % 32) / 16, tt = k % 16;
      for (unsigned m = i; m < 14 && m - i < 8; ++m)
        bf[(((size_t)(NT - 1) * KS + s) * 64 + 32 * h + m) * 16 + tt] = phi[m - i];
    }
  }
  const u128h kfv = (u128h)(uint64_t)sum_inv * 0x8080808080808080ull;
  kf[0] = (uint64_t)kfv; kf[1] = (uint64_t)(kfv >> 64);
  bfw = bf;                                                                // the F columns (row tile NT-1) are the same
  for (int scaled = 0; scaled < 2; ++scaled) {
    std::vector<int8_t> &B = scaled ? bfw : bf;
    std::vector<uint64_t> &PK = scaled ? pkw : pk, &TK = scaled ? tkpw : tkp;
    for (unsigned j = 0; j < cnt; ++j) {
      const uint64_t pj = c->p[dimP + j];
      const uint64_t Pm0 = mod_small(bp->h_P, pj);
      const uint64_t Pinv = powm(Pm0, pj - 2, pj);
      const uint64_t wj = (uint64_t)((u128h)Pinv * bq->h_phat_inv[j] % pj);
      const uint64_t mulw = scaled ? wj : 1;
      const uint64_t Pm = (uint64_t)((u128h)Pm0 * mulw % pj);
      wscale[j] = wj;
      uint64_t sum_ph = 0;
      const unsigned nt = j / 4, pq = j % 4;
      for (unsigned d = 0; d < dimP; ++d) {
        Big ph(bp->h_phat.begin() + (size_t)d * bp->WP, bp->h_phat.begin() + (size_t)(d + 1) * bp->WP);
        uint64_t T = (uint64_t)((u128h)mod_small(ph, pj) * mulw % pj);   // (P/p_d) [w_j] mod p_j
        sum_ph = (uint64_t)(((u128h)sum_ph + T) % pj);
        for (unsigned i = 0; i < 8; ++i) {
          int8_t dig[8];
          balanced8(T, dig);
          const unsigned k = 8 * d + i, s = k / 32, h = (k % 32) / 16, tt = k % 16;
          for (unsigned b = 0; b < 8; ++b) B[(((size_t)nt * KS + s) * 64 + 32 * h + 8 * pq + b) * 16 + tt] = dig[b];
          T = (uint64_t)(((u128h)T << 8) % pj);
        }
      }
      const uint64_t K = (uint64_t)((u128h)(0x8080808080808080ull % pj) * sum_ph % pj);
      const uint64_t off = 1ull << 50;
      PK[3 * (size_t)j] = pj;
      PK[3 * (size_t)j + 1] = off + (K + pj - off % pj) % pj;
      PK[3 * (size_t)j + 2] = wj;
      for (unsigned k = 0; k < 64; ++k) TK[(size_t)j * 64 + k] = (pj - (uint64_t)((u128h)k * Pm % pj)) % pj;
    }
  }
  // the context's per-limb table for the key switch's inverse pass: (P/p_d)^-1 on the limbs of P, w_j above
  std::vector<LimbTab> tw = c->h_tabs;
  if (!tw.empty()) {
    auto pair_of = [](uint64_t w, uint64_t p) { return TwS{p - w, p - (uint64_t)(((u128h)w << 31) % p)}; };
    for (unsigned d = 0; d < dimB; ++d) {
      LimbTab &e = tw[d];
      const uint64_t p = e.k.p, sc = d < dimP ? bp->h_phat_inv[d] : wscale[d - dimP];
      e.ninv = (uint64_t)((u128h)e.ninv * sc % p);
      e.winv1_ninv = (uint64_t)((u128h)e.winv1_ninv * sc % p);
      if (d < c->nsplit_tables) { e.ninv_s = pair_of(e.ninv, p); e.winv1_ninv_s = pair_of(e.winv1_ninv, p); }
    }
  }
  DeviceScope on_device(c->device);
  HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_bfrag, bf.size()));
  HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_lk, lk.size() * 8));
  HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_pk, pk.size() * 8));
  HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_tkp, tkp.size() * 8));
  HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_kf, kf.size() * 8));
  HIP_TRY(hipMemcpy(rt->d_bfrag, bf.data(), bf.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(rt->d_lk, lk.data(), lk.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(rt->d_pk, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(rt->d_tkp, tkp.data(), tkp.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(rt->d_kf, kf.data(), kf.size() * 8, hipMemcpyHostToDevice));
  if (!tw.empty()) {
    HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_bfrag_w, bfw.size()));
    HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_pk_w, pkw.size() * 8));
    HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_tkp_w, tkpw.size() * 8));
    HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_tabs_w, tw.size() * sizeof(LimbTab)));
    HIP_TRY(hipMemcpy(rt->d_bfrag_w, bfw.data(), bfw.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(rt->d_pk_w, pkw.data(), pkw.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(rt->d_tkp_w, tkpw.data(), tkpw.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(rt->d_tabs_w, tw.data(), tw.size() * sizeof(LimbTab), hipMemcpyHostToDevice));
  }
  rt->NT = NT; rt->KS = KS; rt->lds_bytes = lds;
  return GPQ_OK;
}

// Tables of the ONE-PRODUCT relinearisation tail (bridge_reconstruct_low_mfma<16> with frac_bits = 104, bridge_mfma.hpp).  With y_d the
// residues scaled for the CRT over ALL dimB limbs (Pi_B = P Pi'), x = sum_d y_d Pi_B/p_d - kappa Pi_B and
//     2^104 x / P = sum_d y_d (Pi' 2^104 / p_d) - kappa Pi' 2^104 :
// exact integers for the limbs above P (p_j divides Pi'), floors for the limbs of P -- an underestimate by less than dimP 2^60 units of
// 2^-104.  The low 104 bits of the 16-word sum are the fraction (x mod P)/P that mpi_rdiv rounds on, the bits above floor(x/P); kappa
// (the multiples of Pi_B the centring of x takes off) comes from the same F columns as in any CRT.  Also: the per-limb table that makes
// the key switch's inverse pass deliver y_d (ScaledInverse), and the weights Pi_B/p_d mod p_d that take the scaling off again
// (bridge_limb_scale) for the few groups the exact kernels re-run.
int get_tail_direct(gpq_ctx *c, unsigned dimP, unsigned dimB, gpq_relin_tables *rt) {
  if (rt->direct_tried) return GPQ_OK;
  rt->direct_tried = true;
  gpq_bridge_basis *bB, *bq;
  int rc;
  if (dimB > 60 || dimB - dimP < 4) return GPQ_OK;
  if ((rc = get_basis(c, 0, dimB, &bB)) || (rc = get_basis(c, dimP, dimB - dimP, &bq))) return rc;
  if (c->h_tabs.empty()) return GPQ_OK;
  constexpr int WL = 16;
  Big num = bq->h_P;                                        // Pi' 2^104
  num.insert(num.begin(), 0);                               // << 64
  mul_small(num, 1ull << 40);                               // << 40
  std::vector<uint64_t> primes(dimB), scale(dimB), unscale(dimB);
  std::vector<Big> weight(dimB);
  for (unsigned d = 0; d < dimB; ++d) {
    primes[d] = c->p[d];
    Big q = num;
    (void)divmod_small(q, primes[d]);                       // floor(Pi' 2^104 / p_d): exact for d >= dimP
    q.resize(WL < (int)q.size() ? q.size() : WL, 0);
    weight[d] = q;
    scale[d] = bB->h_phat_inv[d];                           // (Pi_B/p_d)^-1 mod p_d
    Big ph(bB->h_phat.begin() + (size_t)d * bB->WP, bB->h_phat.begin() + (size_t)(d + 1) * bB->WP);
    unscale[d] = mod_small(ph, primes[d]);                  // Pi_B/p_d mod p_d
  }
  if ((rc = build_recon_mfma(c, primes, scale, weight, num, WL, &rt->direct))) return rc;
  if (!rt->direct.d_bfrag) return GPQ_OK;
  const LimbTab *tabs;
  if ((rc = get_scaled_tabs(c, bB, &tabs))) return rc;
  rt->d_tabs_direct = tabs;
  DeviceScope on_device(c->device);
  HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_scale, dimB * 8));
  HIP_TRY(gpq_table_malloc(c, (void **)&rt->d_unscale, dimB * 8));
  HIP_TRY(hipMemcpy(rt->d_scale, scale.data(), dimB * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(rt->d_unscale, unscale.data(), dimB * 8, hipMemcpyHostToDevice));
  return GPQ_OK;
}

// How the key switch of gpq_he_mul / gpq_he_swk may scale its output for the tail that follows: 0 = not at all, 1 = the limbs of P by
// (P/p_d)^-1 (any tail), 2 = also the limbs above P by w_j (the matrix-core front only: its exact fallbacks read those limbs through
// the same tables), 3 = every limb by (Pi_B/p_d)^-1 for the one-product tail.  *tabs = the per-limb table the inverse pass reads.
int tail_prescale_mode(gpq_ctx *c, unsigned dimP, unsigned dimB, const LimbTab **tabs, int *mode) {
  *tabs = nullptr; *mode = 0;
  if (!can_prescale(c)) return GPQ_OK;
  gpq_bridge_basis *bp, *bq;
  gpq_relin_tables *rt;
  int rc;
  if ((rc = get_basis(c, 0, dimP, &bp)) || (rc = get_basis(c, dimP, dimB - dimP, &bq)) || (rc = get_relin(c, dimP, dimB, &rt))) return rc;
  if (c->bridge_mfma && c->logn >= 6 && (rc = get_relin_front(c, dimP, dimB, rt, bp, bq))) return rc;
  if (c->bridge_mfma && c->logn >= 6 && rt->d_bfrag && c->tail_direct && !c->exact_crt && !c->fuse_tail) {
    if ((rc = get_tail_direct(c, dimP, dimB, rt))) return rc;
    if (rt->direct.d_bfrag && rt->d_tabs_direct) { *tabs = rt->d_tabs_direct; *mode = 3; return GPQ_OK; }
  }
  if (c->bridge_mfma && c->logn >= 6 && rt->d_bfrag && rt->d_tabs_w && c->prescale_upper) { *tabs = rt->d_tabs_w; *mode = 2; return GPQ_OK; }
  *mode = 1;
  return get_scaled_tabs(c, bp, tabs);
}

template <int KS, int WL>
int launch_relin_tail_t(const RelinTailArgs &a, size_t lds, hipStream_t s) {
  static LdsRaised raised;
  if (int rc = raised.raise(reinterpret_cast<const void *>(&bridge_relin_tail_mfma<KS, WL>), (int)kMfmaLdsMax)) return rc;
  unsigned blocks = 256 * GPQ_TAIL_WAVES;                  // 4-wave workgroups, GPQ_TAIL_WAVES per CU (registers; their tables fit the LDS twice), persistent over the groups
  if (blocks > (a.f.total_groups + 3) / 4) blocks = (a.f.total_groups + 3) / 4;
  hipLaunchKernelGGL((bridge_relin_tail_mfma<KS, WL>), dim3(blocks), dim3(256), lds, s, a);
  return GPQ_OK;
}

template <int KS>
int launch_relin_front_t(const RelinFrontArgs &a, size_t lds, hipStream_t s) {
  static LdsRaised raised;
  if (int rc = raised.raise(reinterpret_cast<const void *>(&bridge_relin_front_mfma<KS>), (int)kMfmaLdsMax)) return rc;
  unsigned per_cu = (unsigned)((160 * 1024) / lds);
  if (per_cu > 3) per_cu = 3;
  if (per_cu < 1) per_cu = 1;
  unsigned blocks = 256 * per_cu;
  if (blocks > (a.total_groups + 3) / 4) blocks = (a.total_groups + 3) / 4;
  if (a.scope.wave_any) blocks = (a.scope.waves + 3) / 4;     // wave w of this launch = wave w of the launch that wrote the mask
  hipLaunchKernelGGL((bridge_relin_front_mfma<KS>), dim3(blocks), dim3(256), lds, s, a);
  return GPQ_OK;
}
int launch_relin_front(const gpq_ctx *c, unsigned KS, const RelinFrontArgs &f, size_t lds, hipStream_t s) {
  ProfScope prof(c, f.only ? GPQ_K_BRIDGE_EXACT : GPQ_K_RELIN_FRONT, s);
  switch (KS) {
    case 2: return launch_relin_front_t<2>(f, lds, s);
    case 4: return launch_relin_front_t<4>(f, lds, s);
    default: return launch_relin_front_t<8>(f, lds, s);
  }
}

// The addend of the relinearisation tail as LIMBS (he_mul: d0hat | d1hat, src/he-mult.c:139,141 never leave RNS form before the tail):
// [polys][dimA][n] weighted for the CRT over bA; scratch = [polys][W][n] words for the coefficients the exact kernels redo (or for all of
// them when the streaming kernel does not cover the shape).
struct TailD { const uint64_t *hat; gpq_bridge_basis *bA; unsigned dimA; uint64_t *scratch; };

template <int KST, int KSD, bool DCRT, int R>
int launch_tail_stream_t(const TailStreamArgs &a, size_t lds, unsigned blocks, hipStream_t s) {
  static LdsRaised raised;
  if (int rc = raised.raise(reinterpret_cast<const void *>(&bridge_tail_stream<KST, KSD, DCRT, R>), (int)kStreamLdsMax)) return rc;
  hipLaunchKernelGGL((bridge_tail_stream<KST, KSD, DCRT, R>), dim3(blocks), dim3(512), lds, s, a);
  return GPQ_OK;
}

// get_tail_direct's matrix zero-padded to KST k steps
int get_tail_direct_padded(gpq_ctx *c, unsigned dimP, unsigned dimB, gpq_relin_tables *rt, unsigned KST, gpq_recon_mfma **out) {
  if (KST <= (dimB + 3) / 4) { *out = &rt->direct; return GPQ_OK; }
  auto it = rt->direct_padded.find(KST);
  if (it != rt->direct_padded.end()) { *out = &it->second; return GPQ_OK; }
  gpq_bridge_basis *bB, *bq;
  int rc;
  if ((rc = get_basis(c, 0, dimB, &bB)) || (rc = get_basis(c, dimP, dimB - dimP, &bq))) return rc;
  Big num = bq->h_P;                                        // Pi' 2^104 (as get_tail_direct)
  num.insert(num.begin(), 0);
  mul_small(num, 1ull << 40);
  std::vector<uint64_t> primes(dimB), scale(dimB);
  std::vector<Big> weight(dimB);
  for (unsigned d = 0; d < dimB; ++d) {
    primes[d] = c->p[d];
    Big q = num;
    (void)divmod_small(q, primes[d]);
    q.resize(16 < q.size() ? q.size() : 16, 0);
    weight[d] = q;
    scale[d] = bB->h_phat_inv[d];
  }
  gpq_recon_mfma t;
  if ((rc = build_recon_mfma(c, primes, scale, weight, num, 16, &t, KST))) return rc;
  *out = &(rt->direct_padded[KST] = t);
  return GPQ_OK;
}

// The one-product tail as a stream (bridge_stream.hpp), with the addend's CRT in the same kernel when it comes as limbs.  *scope = the
// launch's per-wave words for the masked kernels behind it; *done = false: not covered (shape, settings).
int tail_stream(gpq_ctx *c, gpq_relin_tables *rt, Two<uint64_t> out, const uint64_t *chat, Two<const uint64_t> dbig, const TailD *dh, unsigned W,
                unsigned dimP, unsigned dimB, unsigned logql, unsigned polys, unsigned char *tie, unsigned char *amb, hipStream_t s, FlagScope *scope, bool *done,
                unsigned rs = 0) {
  *done = false;
  const unsigned need = (logql + 63) / 64;
  if (!c->stream_bridge || W > 14 || W < need || logql > 896) return GPQ_OK;
  int KST, KSD = 0;
  if (dh) {
    if (!stream_fast_ok(c, dh->bA, logql)) return GPQ_OK;
    if (dimB <= 32 && dh->dimA <= 16) { KST = 8; KSD = 4; }
    else if (dimB <= 48 && dh->dimA <= 32) { KST = 12; KSD = 8; }
    else return GPQ_OK;
  } else {
    if (dimB <= 24) KST = 6; else if (dimB <= 48) KST = 12; else return GPQ_OK;
  }
  gpq_recon_mfma *tt, *td = nullptr;
  int rc;
  if ((rc = get_tail_direct_padded(c, dimP, dimB, rt, KST, &tt))) return rc;
  if (dh && (rc = get_addend_rows(c, dh->bA, KSD, &td))) return rc;
  if (!tt->d_bfrag || tt->KS != (unsigned)KST || (dh && (!td->d_bfrag || td->KS != (unsigned)KSD))) return GPQ_OK;
  const size_t lds = (size_t)(KST + KSD) * 5 * 1024 + (size_t)2 * 65 * kTailRow * 8;
  if (lds > kStreamLdsMax) return GPQ_OK;
  if ((rc = ensure_redo(c, (size_t)polys << c->logn, s)) || (rc = ensure_wave_any(c, s))) return rc;
  const unsigned groups = (c->n >> 6) * polys, blocks = stream_blocks(groups);
  const size_t chat_bytes = ((size_t)polys * dimB << c->logn) * 8, dhat_bytes = dh ? ((size_t)polys * dh->dimA << c->logn) * 8 : 0;
  if (chat_bytes >= 0xfffff000ull || dhat_bytes >= 0xfffff000ull) return GPQ_OK;
  TailStreamArgs a{chat, dh ? dh->hat : nullptr, chat_bytes, dhat_bytes, dbig, out, (const v4i *)tt->d_bfrag, tt->d_kc, tt->d_pm,
                   dh ? (const v4i *)td->d_bfrag : nullptr, dh ? td->d_kc : nullptr, dh ? td->d_pm : nullptr,
                   c->d_redo, tie, amb, c->d_wave_any, dimB, dh ? dh->dimA : 0u, c->logn, W, logql, groups, c->debug_force_redo,
                   rs, rs ? logql - rs : 0u};
  ProfScope prof(c, GPQ_K_TAIL_STREAM, s);
  if (dh && KST == 8) rc = launch_tail_stream_t<8, 4, true, 6>(a, lds, blocks, s);
  else if (dh) rc = launch_tail_stream_t<12, 8, true, 5>(a, lds, blocks, s);
  else if (KST == 6) rc = launch_tail_stream_t<6, 0, false, 3>(a, lds, blocks, s);
  else rc = launch_tail_stream_t<12, 0, false, 4>(a, lds, blocks, s);
  if (rc) return rc;
  *scope = FlagScope{c->d_wave_any, blocks * kStreamWaves, groups};
  *done = true;
  return GPQ_OK;
}

// src/he-mult.c:67-77 (d != null: c = rdiv(c,P) + d) and src/he-automorphism.c:68-76, for q_l = 2^logql.
// `polys` polynomials of chat; the first `split` of them go to out.a (+ addend d.a), the rest to out.b (+ d.b): c0 and c1 of a
// launch group are one batch (their chat slabs are adjacent in the workspace), half the launches and twice their size.
// chat_prescaled: the limbs below dimP already hold chat_d * (P/p_d)^-1 (ScaledInverse on the key switch's inverse pass).
int relin_tail(gpq_ctx *c, Two<uint64_t> out, const uint64_t *chat, Two<const uint64_t> dbig, unsigned W, unsigned dimP, unsigned dimB,
               unsigned logql, unsigned polys, void *ws, hipStream_t s, int chat_prescaled = 0, const TailD *dh = nullptr,
               unsigned rs = 0, bool *rs_done = nullptr) {
  // rs / rs_done (gpq_he_mul_rs): log2(Delta) of the he_rs the tail applies on the way out where its streaming kernel runs (1 <= rs <= 63), and
  // whether it did; explicit parameters -- gpq_he_swk and gpq_relin_tail share this function and pass neither
  if (rs_done) *rs_done = false;
  TailPlan tp;
  int rc = tail_plan(c, W, dimP, dimB, polys, &tp);
  if (rc) return rc;
  gpq_bridge_basis *bp, *bq;
  gpq_relin_tables *rt;
  if ((rc = get_basis(c, 0, dimP, &bp)) || (rc = get_basis(c, dimP, tp.cnt, &bq)) || (rc = get_relin(c, dimP, dimB, &rt))) return rc;
  if (W > (unsigned)bq->WP + 1) return gpq_fail(GPQ_ERR_UNSUPPORTED, "relin: W=%u words exceed the quotient basis", W);
  const size_t n = c->n;
  uint64_t *r = (uint64_t *)ws, *rhat = r + (size_t)polys * tp.Wr * n, *qhat = rhat + (size_t)polys * tp.cnt * n,
           *qc = qhat + (size_t)polys * tp.cnt * n;
  unsigned char *tie = (unsigned char *)(qc + (size_t)polys * W * n);
  const dim3 cgrid((c->n + 255) / 256, polys), cblock(256);
  const uint64_t *piq = bq->d_pmult + (size_t)5 * (bq->WP + 1);
  const Two<const uint64_t> qc_one{qc, nullptr, ~0u};
  // in place (out IS d) the exact kernel must not park Q over d: Q goes through qc
  const bool in_place = (dbig.a && dbig.a == out.a) || (dbig.b && dbig.b == out.b);

  if (chat_prescaled == 3) {
    // The limbs carry the CRT weights of the whole basis (ScaledInverse with get_tail_direct's table): ONE product gives floor(x/P), the
    // rounding decision and the centring of x.  `chat` is the caller's scratch here (c0hat | c1hat of the workspace): the groups of
    // 64 coefficients the product cannot decide get their weights taken off in place and go through the exact sequence below.
    uint64_t *chat_rw = const_cast<uint64_t *>(chat);
    unsigned char *flags = (unsigned char *)rhat, *amb = flags + ((size_t)polys << c->logn);
    const bool direct_ok = !in_place && rt->direct.d_bfrag && W <= 14 && logql <= 896 && c->bridge_mfma;
    // the flag bytes must have their final size BEFORE any argument block below copies c->d_redo: a later growth (the addend's CRT, the
    // product) would leave `un` with the outgrown buffer -- stale flags, and reads past its end (found by tools/soak_bridge.py, seed 23)
    if (direct_ok && (rc = ensure_redo(c, (size_t)polys << c->logn, s))) return rc;
    FlagScope scope = kNoScope;
    bool streamed = false;
    if (direct_ok && (rc = tail_stream(c, rt, out, chat, dbig, dh, W, dimP, dimB, logql, polys, tie, amb, s, &scope, &streamed, rs_done ? rs : 0u))) return rc;
    // Behind the streaming kernel the exact kernels only see flagged coefficients, and the chains whose hand-overs stay inside a thread are
    // ONE launch each (bridge_kernels.hpp: bridge_fallback_tail_pre / _post): pre = the addend's exact CRT + the weights off the flagged
    // groups; the front re-run; post = r, its round bit, Q's exact CRT, the finish.  Three launches where there were seven.
    // gpq_he_mul_rs: the streaming kernel rescaled what it decided; the coefficients it flagged are written unrescaled by the exact kernels
    // below and finished here (same flags, same scope)
    auto finish_rs = [&]() {
      if (!rs || !rs_done || !streamed) return;
      ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s);
      RescaleMaskedArgs ra{out, c->d_redo, W, c->logn, rs, logql - rs, scope};
      hipLaunchKernelGGL(bridge_rescale_masked, masked_grid(scope, 256, c->n, polys), dim3(256), 0, s, ra);
      *rs_done = true;
    };
    const bool fuse_pre = streamed && dh && (dh->bA->WP == 32 || dh->bA->WP == 16);
    const bool fuse_post = streamed && ((bp->WP == 16 && bq->WP == 32) || (bp->WP == 8 && bq->WP == 16));
    LimbScaleArgs un{c->d_tabs, chat_rw, rt->d_unscale, direct_ok ? c->d_redo : nullptr, dimB, c->logn, scope};
    if (dh) {
      // the addend as words: for the flagged coefficients only behind the streaming kernel (exact CRT), for all of them otherwise
      ReconExtra dx;
      dx.prescaled = true;
      if (streamed) { dx.exact_only = true; dx.only = c->d_redo; dx.scope = scope; }
      if (fuse_pre) {
        ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s);
        const ReconstructArgs dd = exact_args(c, dh->bA, dh->scratch, W, dh->hat, dh->dimA, 0, logql, true, nullptr, c->logn, dx);
        const dim3 grid = masked_grid(scope, 128, c->n, polys);
        if (dh->bA->WP == 32) hipLaunchKernelGGL((bridge_fallback_tail_pre<32>), grid, dim3(128), 0, s, dd, un);
        else hipLaunchKernelGGL((bridge_fallback_tail_pre<16>), grid, dim3(128), 0, s, dd, un);
      } else if ((rc = launch_reconstruct(c, dh->bA, dh->scratch, W, dh->hat, dh->dimA, 0, polys, logql, true, nullptr, s, -1, dx))) return rc;
      dbig = Two<const uint64_t>{dh->scratch, dh->scratch + (size_t)out.split * W * n, out.split};
    }
    if (direct_ok && !streamed) {
      if ((rc = ensure_redo(c, (size_t)polys << c->logn, s))) return rc;
      const unsigned gpp = c->n >> 6;
      ReconMfmaArgs m{chat, out, (const v4i *)rt->direct.d_bfrag, rt->direct.d_lk, rt->direct.d_kc, rt->direct.d_pm, c->d_redo, tie, dimB, rt->direct.KS,
                      c->logn, W, logql, dimB, 0, gpp, gpp * polys, dbig, nullptr, 1u, 104u, amb};
      {
        ProfScope prof(c, GPQ_K_RELIN_TAIL_DIRECT, s);
        if ((rc = launch_low_mfma16(m, rt->direct.lds_bytes, s))) return rc;
      }
    }
    // weights off: for the flagged groups, or -- no product possible (in place, no tables) -- for every coefficient
    if (!fuse_pre) { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_limb_scale, masked_grid(scope, 256, c->n, polys), cblock, 0, s, un); }
    if (!direct_ok) return relin_tail(c, out, chat, dbig, W, dimP, dimB, logql, polys, ws, s, 0);   // (no rs: the caller rescales)
    // the exact sequence on the flagged groups: front (re-run, writing its flags), r for its ambiguous ones, round bits, Q, finish
    const unsigned gpp = c->n >> 6;
    RelinFrontArgs f{chat, qhat, (const v4i *)rt->d_bfrag, rt->d_lk, rt->d_pk, rt->d_tkp, rt->d_kf, flags, amb,
                     dimB, dimP, tp.cnt, c->logn, rt->NT, gpp, gpp * polys, c->d_redo, 1u, 0u, 0u, scope};
    if ((rc = launch_relin_front(c, rt->KS, f, rt->lds_bytes, s))) return rc;
    ReconExtra only_amb;
    only_amb.only = amb; only_amb.scope = scope;
    RoundFixArgs rf{r, bp->d_phalf, amb, flags, tp.Wr, c->logn, scope};
    ReconExtra q;
    q.prescaled = true; q.exact_only = true; q.only = c->d_redo; q.big_b = out.b; q.split = out.split; q.scope = scope;
    AddRoundArgs ar{out, Two<const uint64_t>{out.a, out.b, out.split}, nullptr, dbig, bp->d_phalf, piq, tie, W, tp.Wr, c->logn, logql, c->d_redo, flags, scope};
    if (fuse_post) {
      ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s);
      const ReconstructArgs rr = exact_args(c, bp, r, tp.Wr, chat, dimB, 0, 0, false, nullptr, c->logn, only_amb);
      const ReconstructArgs qq = exact_args(c, bq, out.a, W, qhat, tp.cnt, 0, logql, true, tie, c->logn, q);
      const dim3 grid = masked_grid(scope, 128, c->n, polys);
      if (bp->WP == 16) hipLaunchKernelGGL((bridge_fallback_tail_post<16, 32>), grid, dim3(128), 0, s, rr, rf, qq, ar);
      else hipLaunchKernelGGL((bridge_fallback_tail_post<8, 16>), grid, dim3(128), 0, s, rr, rf, qq, ar);
      finish_rs();
      return launched("relin_tail");
    }
    if ((rc = launch_reconstruct(c, bp, r, tp.Wr, chat, dimB, 0, polys, 0, false, nullptr, s, -1, only_amb))) return rc;
    { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_roundfix, masked_grid(scope, 256, c->n, polys), cblock, 0, s, rf); }
    if ((rc = launch_reconstruct(c, bq, out.a, W, qhat, tp.cnt, 0, polys, logql, true, tie, s, -1, q))) return rc;
    { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_addround, masked_grid(scope, 256, c->n, polys), cblock, 0, s, ar); }
    finish_rs();
    return launched("relin_tail");
  }

  if (c->bridge_mfma && c->logn >= 6 && (rc = get_relin_front(c, dimP, dimB, rt, bp, bq))) return rc;
  if (c->bridge_mfma && c->logn >= 6 && rt->d_bfrag) {
    // Matrix-core front: Q's (pre-scaled) residues and the round bits straight from chat; the exact kernels only see the
    // coefficients whose rounding the fixed-point estimates cannot decide.
    unsigned char *flags = (unsigned char *)rhat, *amb = flags + ((size_t)polys << c->logn);   // rhat's place is free in this flow
    const unsigned gpp = c->n >> 6;
    const bool wsc = chat_prescaled == 2;                 // the limbs above P arrive multiplied by w_j: the w-scaled tables, no multiplication in the epilogue
    RelinFrontArgs f{chat, qhat, (const v4i *)(wsc ? rt->d_bfrag_w : rt->d_bfrag), rt->d_lk, wsc ? rt->d_pk_w : rt->d_pk, wsc ? rt->d_tkp_w : rt->d_tkp, rt->d_kf, flags, amb,
                     dimB, dimP, tp.cnt, c->logn, rt->NT, gpp, gpp * polys, nullptr, 0u, chat_prescaled ? 1u : 0u, wsc ? 1u : 0u};
    // One pass per coefficient (bridge_relin_tail_mfma): the front and the CRT of Q without the round trip of Q's residues.
    const unsigned need = (logql + 63) / 64;
    const int WLf = need <= 7 ? 7 : 14;
    gpq_recon_mfma *tq = nullptr;
    const bool can_fuse = c->fuse_tail && !in_place && !c->exact_crt && need <= 14 && need + 1 < (unsigned)bq->WP && bq->pbits >= 160 && tp.cnt >= 4 && tp.cnt <= 4 * RELIN_TAIL_MAXTILES &&
                          (rt->KS == 2 || rt->KS == 4);
    if (can_fuse && (rc = get_recon_mfma(c, bq, WLf, &tq))) return rc;
    if (can_fuse && tq->d_bfrag && tq->KS + 1 == rt->NT && rt->lds_bytes + (size_t)tq->KS * ((8 * WLf + 14 + 31) / 32) * 1024 <= kMfmaLdsMax) {
      if ((rc = ensure_redo(c, (size_t)polys << c->logn, s))) return rc;
      const size_t lds = rt->lds_bytes + (size_t)tq->KS * ((8 * WLf + 14 + 31) / 32) * 1024;
      RelinTailArgs ft{f, (const v4i *)tq->d_bfrag, tq->d_kc, tq->d_pm, c->d_redo, tie, out, dbig, W, logql, tq->KS};
      {
        ProfScope prof(c, GPQ_K_RELIN_TAIL_FUSED, s);
        if (rt->KS == 2 && WLf == 7) rc = launch_relin_tail_t<2, 7>(ft, lds, s);
        else if (rt->KS == 2) rc = launch_relin_tail_t<2, 14>(ft, lds, s);
        else if (WLf == 7) rc = launch_relin_tail_t<4, 7>(ft, lds, s);
        else rc = launch_relin_tail_t<4, 14>(ft, lds, s);
      }
      if (rc) return rc;
      // the few coefficients it flagged: round bits settled exactly, Q's residues made for their groups, exact CRT, finish
      ReconExtra only_amb;
      only_amb.only = amb; only_amb.prescaled = chat_prescaled != 0;
      if ((rc = launch_reconstruct(c, bp, r, tp.Wr, chat, dimB, 0, polys, 0, false, nullptr, s, -1, only_amb))) return rc;
      RoundFixArgs rf{r, bp->d_phalf, amb, flags, tp.Wr, c->logn};
      { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_roundfix, cgrid, cblock, 0, s, rf); }
      f.only = c->d_redo;
      if ((rc = launch_relin_front(c, rt->KS, f, rt->lds_bytes, s))) return rc;
      ReconExtra q;
      q.prescaled = true; q.exact_only = true; q.only = c->d_redo; q.big_b = out.b; q.split = out.split;
      if ((rc = launch_reconstruct(c, bq, out.a, W, qhat, tp.cnt, 0, polys, logql, true, tie, s, -1, q))) return rc;
      AddRoundArgs ar{out, Two<const uint64_t>{out.a, out.b, out.split}, nullptr, dbig, bp->d_phalf, piq, tie, W, tp.Wr, c->logn, logql, c->d_redo, flags};
      { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_addround, cgrid, cblock, 0, s, ar); }
      return launched("relin_tail");
    }
    if ((rc = launch_relin_front(c, rt->KS, f, rt->lds_bytes, s))) return rc;
    ReconExtra only_amb;
    only_amb.only = amb; only_amb.prescaled = chat_prescaled != 0;
    if ((rc = launch_reconstruct(c, bp, r, tp.Wr, chat, dimB, 0, polys, 0, false, nullptr, s, -1, only_amb))) return rc;
    RoundFixArgs rf{r, bp->d_phalf, amb, flags, tp.Wr, c->logn};
    { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_roundfix, cgrid, cblock, 0, s, rf); }
    bool fused = false;
    ReconExtra q;
    q.prescaled = true; q.fused = &fused;
    // Fused finish straight into `out` (the exact kernel parks Q of its few coefficients there before bridge_addround
    // adds d, so `out` must not BE d); otherwise Q goes to qc and bridge_addround finishes every coefficient.
    const bool direct = !in_place;
    if (direct) { q.addend = dbig; q.rflags = flags; q.big_b = out.b; q.split = out.split; }
    uint64_t *target = direct ? out.a : qc;
    if ((rc = launch_reconstruct(c, bq, target, W, qhat, tp.cnt, 0, polys, logql, true, tie, s, -1, q))) return rc;
    AddRoundArgs ar{out, direct ? Two<const uint64_t>{out.a, out.b, out.split} : qc_one, nullptr, dbig, bp->d_phalf, piq, tie, W, tp.Wr, c->logn, logql,
                    fused ? c->d_redo : nullptr, flags};
    { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_addround, cgrid, cblock, 0, s, ar); }
    return launched("relin_tail");
  }

  // r = x mod P from the first dimP limbs, unsigned
  ReconExtra rx;
  rx.prescaled = chat_prescaled != 0;
  if (chat_prescaled == 2) return gpq_fail(GPQ_ERR_INVALID, "relin_tail: w-scaled limbs need the matrix-core front");
  if ((rc = launch_reconstruct(c, bp, r, tp.Wr, chat, dimB, 0, polys, 0, false, nullptr, s, -1, rx))) return rc;
  if ((rc = launch_decompose(c, rhat, r, tp.Wr, dimP, tp.cnt, polys, s))) return rc;
  ExactDivArgs e{c->d_tabs, chat, rhat, qhat, rt->d_pinv, dimB, dimP, tp.cnt, c->logn};
  { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_exactdiv, dim3((c->n + 255) / 256, polys, tp.cnt), dim3(256), 0, s, e); }
  // Q = (x - r)/P over the remaining limbs, centred, already reduced smod 2^logql
  if ((rc = launch_reconstruct(c, bq, qc, W, qhat, tp.cnt, 0, polys, logql, true, tie, s))) return rc;
  AddRoundArgs ar{out, qc_one, r, dbig, bp->d_phalf, piq, tie, W, tp.Wr, c->logn, logql, nullptr, nullptr};
  { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_addround, cgrid, cblock, 0, s, ar); }
  return launched("relin_tail");
}

inline size_t align64(size_t b) { return (b + 63) & ~(size_t)63; }

}  // namespace

// dims the reference derives from the modulus chain: hectx.dim src/precomp.c:401, he_mul's
// tensor dim src/he-mult.c:99, he_relin/he_swk's dim src/he-mult.c:51, dimevk src/precomp.c:407.
extern "C" int gpq_he_dims(gpq_ctx *c, unsigned logqL, unsigned logql, unsigned *dimP, unsigned *dimA, unsigned *dimB, unsigned *dimevk) {
  if (!c || !logqL || !logql || logql > logqL) return gpq_fail(GPQ_ERR_INVALID, "gpq_he_dims: bad arguments");
  const unsigned logn = c->logn;
  const unsigned dp = (logqL + 1 + logn) / 59 + 1;
  gpq_bridge_basis *bp;
  int rc = get_basis(c, 0, dp, &bp);
  if (rc) return rc;
  const unsigned nbPqL = bp->pbits + logqL;
  if (dimP) *dimP = dp;
  if (dimA) *dimA = (2 * (logql + 1) + logn) / 59 + 1;
  if (dimB) *dimB = ((logql + 1) + nbPqL + logn) / 59 + 1;
  if (dimevk) *dimevk = ((logqL + 1) + nbPqL + logn) / 59 + 1;
  return GPQ_OK;
}

extern "C" size_t gpq_he_mul_workspace_bytes(gpq_ctx *c, unsigned W, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch) {
  const unsigned m = batch < c->chunk ? batch : c->chunk;
  TailPlan tp;
  if (tail_plan(c, W, dimP, dimB, 2 * m, &tp) != GPQ_OK) return 0;      // c0 and c1 of a launch group go through the tail together
  const size_t n = c->n;
  size_t b = 0;
  b += align64((size_t)m * 7 * dimA * n * 8);                       // 4 decomposed inputs + d0hat,d1hat,d2hat
  b += align64(gpq_tensor_workspace_bytes(c, dimA, m));
  b += align64((size_t)m * 3 * dimB * n * 8);                       // decomposed d2 (or d1), c0hat, c1hat
  b += align64(gpq_keyswitch_workspace_bytes(c, dimB, m));
  b += align64((size_t)m * 3 * W * n * 8);                          // d0, d1, d2
  b += align64(tp.bytes);
  return b;
}

// he_mul, src/he-mult.c:88-156 (decl src/gpqhe.h:147), on big slabs of W words, q_l = 2^logql.
// ct = ct1 * ct2 relinearised with rlk (NTT-domain slabs of at least dimB limbs).  The l / nu / B
// bookkeeping of :92-95 stays with the caller.
// rs = 0: he_mul; rs = log2(Delta): he_mul followed by he_rs (src/he-rescale.c:33-54) with q_(l-1) = 2^(logql - rs), the rescale applied by the tail
// kernel on the way out where that kernel runs (1 <= rs <= 63), by the rescale kernel on the group's outputs otherwise.
static int he_mul_impl(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *ct1c0, const uint64_t *ct1c1,
                       const uint64_t *ct2c0, const uint64_t *ct2c1, const uint64_t *rlk0, const uint64_t *rlk1, unsigned W,
                       unsigned logql, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream, unsigned rs);
extern "C" int gpq_he_mul(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *ct1c0, const uint64_t *ct1c1,
                          const uint64_t *ct2c0, const uint64_t *ct2c1, const uint64_t *rlk0, const uint64_t *rlk1, unsigned W,
                          unsigned logql, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream) {
  return he_mul_impl(c, out_c0, out_c1, ct1c0, ct1c1, ct2c0, ct2c1, rlk0, rlk1, W, logql, dimA, dimB, dimP, batch, workspace, stream, 0);
}
extern "C" int gpq_he_mul_rs(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *ct1c0, const uint64_t *ct1c1,
                             const uint64_t *ct2c0, const uint64_t *ct2c1, const uint64_t *rlk0, const uint64_t *rlk1, unsigned W,
                             unsigned logql, unsigned dimA, unsigned dimB, unsigned dimP, unsigned logDelta, unsigned batch, void *workspace, void *stream) {
  if (!logDelta || logDelta >= logql) return gpq_fail(GPQ_ERR_INVALID, "gpq_he_mul_rs: 0 < logDelta < logql");
  return he_mul_impl(c, out_c0, out_c1, ct1c0, ct1c1, ct2c0, ct2c1, rlk0, rlk1, W, logql, dimA, dimB, dimP, batch, workspace, stream, logDelta);
}
static int he_mul_impl(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *ct1c0, const uint64_t *ct1c1,
                       const uint64_t *ct2c0, const uint64_t *ct2c1, const uint64_t *rlk0, const uint64_t *rlk1, unsigned W,
                       unsigned logql, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream, unsigned rs) {
  const char *who = rs ? "gpq_he_mul_rs" : "gpq_he_mul";          // errors and launch checks name the entry point the caller used
  int rc = check(c, dimA, batch, who);
  if (rc || (rc = check(c, dimB, batch, who))) return rc;
  if (!out_c0 || !out_c1 || !ct1c0 || !ct1c1 || !ct2c0 || !ct2c1 || !rlk0 || !rlk1 || !workspace || !logql || W < (logql + 63) / 64)
    return gpq_fail(GPQ_ERR_INVALID, "%s: bad arguments", who);
  hipStream_t s = (hipStream_t)stream;
  const size_t n = c->n, bigpoly = (size_t)W * n;
  const unsigned m = batch < c->chunk ? batch : c->chunk;
  TailPlan tp;
  if ((rc = tail_plan(c, W, dimP, dimB, 2 * m, &tp))) return rc;
  char *w = (char *)workspace;
  uint64_t *sA = (uint64_t *)w; w += align64((size_t)m * 7 * dimA * n * 8);
  void *wsT = w; w += align64(gpq_tensor_workspace_bytes(c, dimA, m));
  uint64_t *sB = (uint64_t *)w; w += align64((size_t)m * 3 * dimB * n * 8);
  void *wsK = w; w += align64(gpq_keyswitch_workspace_bytes(c, dimB, m));
  uint64_t *dbig = (uint64_t *)w; w += align64((size_t)m * 3 * W * n * 8);
  void *wsTail = w;
  gpq_bridge_basis *bA;
  if ((rc = get_basis(c, 0, dimA, &bA))) return rc;
  PeerLane lane;
  if (batch > m && (rc = gpq_peer_lane(c, s, gpq_lane_key(1, W, dimA, dimB, dimP, m ^ (logql << 8)), [&](gpq_ctx *q) { return gpq_he_mul_workspace_bytes(q, W, dimA, dimB, dimP, m); }, &lane))) return rc;
  for (unsigned k0 = 0; k0 < batch; k0 += m) {
    const unsigned polys = batch - k0 < m ? batch - k0 : m;
    if (lane.c && ((k0 / m) & 1)) {                              // every other launch group: the same call on the peer, for this group's slice
      const size_t o = k0 * bigpoly;
      if ((rc = he_mul_impl(lane.c, out_c0 + o, out_c1 + o, ct1c0 + o, ct1c1 + o, ct2c0 + o, ct2c1 + o, rlk0, rlk1, W, logql, dimA, dimB, dimP,
                            polys, lane.ws, lane.s, rs))) return rc;
      continue;
    }
    const size_t pa = (size_t)polys * dimA * n, pb = (size_t)polys * dimB * n;
    uint64_t *h[4] = {sA, sA + pa, sA + 2 * pa, sA + 3 * pa};
    uint64_t *d0h = sA + 4 * pa, *d1h = sA + 5 * pa, *d2h = sA + 6 * pa;
    const uint64_t *in[4] = {ct1c0, ct1c1, ct2c0, ct2c1};
    // he_mul(&ct, &ct, &ct, rlk) (src/he-algo.c:151, the squarings of he_exp / he_inv): both operands are the same slabs --
    // two decompositions and two forward transforms instead of four, same residues
    const bool square = ct1c0 == ct2c0 && ct1c1 == ct2c1;
    {
      StageRange stage(square ? "gpq_he_mul: rns_decompose x2 (squaring)" : "gpq_he_mul: rns_decompose x4");
      const unsigned nin = square ? 2 : 4;                                                   // :117-120, one launch: h[0..3] are adjacent
      BigSources src{{in[0] + k0 * bigpoly, in[1] + k0 * bigpoly, in[square ? 0 : 2] + k0 * bigpoly, in[square ? 1 : 3] + k0 * bigpoly}, polys};
      if ((rc = launch_decompose(c, h[0], src, W, 0, dimA, nin * polys, s, c->lazy_decompose && c->logn > 12))) return rc;
    }
    const bool pre = can_prescale(c);      // the inverse passes hand the CRT kernels limbs already multiplied by (P/p_d)^-1
    const LimbTab *tabsA = nullptr, *tabsP = nullptr;
    int tail_mode = 0;
    if (pre && ((rc = get_scaled_tabs(c, bA, &tabsA)) || (rc = tail_prescale_mode(c, dimP, dimB, &tabsP, &tail_mode)))) return rc;
    {
      ScaledInverse scaled(c, tabsA);
      if ((rc = gpq_he_mul_tensor(c, d0h, d1h, d2h, h[0], h[1], square ? h[0] : h[2], square ? h[1] : h[3], dimA, polys, wsT, stream))) return rc;  // :121-136
    }
    uint64_t *d0 = dbig, *d1 = dbig + polys * bigpoly, *d2 = dbig + 2 * polys * bigpoly;
    uint64_t *d2hat = sB, *c0hat = sB + pb, *c1hat = sB + 2 * pb;
    // With the one-product tail and bridge_stream.hpp, d0, d1, d2 (:139-141) never exist as words: d2hat goes CRT -> rns_decompose (:59) in
    // one kernel, d0hat | d1hat enter the tail as limbs.  Otherwise: poly_rns2mpi of the three adjacent slabs in one launch, rns_decompose of d2.
    const bool limbs_addend = pre && tail_mode == 3 && c->stream_bridge;
    bool fused = false;
    if (limbs_addend) {
      StageRange stage("gpq_he_mul: poly_rns2mpi d2 -> he_relin rns_decompose (one kernel)");
      if ((rc = crt_decompose_stream(c, bA, d2hat, d2h, d2, W, dimA, dimB, logql, polys, s, &fused))) return rc;
    }
    if (!fused) {
      {
        StageRange stage("gpq_he_mul: poly_rns2mpi d0,d1,d2 (CRT)");
        ReconExtra rx;
        rx.prescaled = pre;
        if (limbs_addend) rc = launch_reconstruct(c, bA, d2, W, d2h, dimA, 0, polys, logql, true, nullptr, s, -1, rx);
        else rc = launch_reconstruct(c, bA, d0, W, d0h, dimA, 0, 3 * polys, logql, true, nullptr, s, -1, rx);
        if (rc) return rc;
      }
      StageRange stage("gpq_he_mul: he_relin rns_decompose d2");
      if ((rc = launch_decompose(c, d2hat, d2, W, 0, dimB, polys, s, c->lazy_decompose && c->logn > 12))) return rc;          // :59
    }
    // he_relin, :40-85
    {
      ScaledInverse scaled(c, tabsP);
      if ((rc = gpq_keyswitch(c, c0hat, c1hat, d2hat, rlk0, rlk1, dimB, polys, wsK, stream))) return rc;           // :60-64
    }
    StageRange stage("gpq_he_mul: he_relin tail (CRT, exact division by P, + d)");
    // c0 and c1 as one batch of 2 x polys polynomials: c0hat | c1hat and d0 | d1 are adjacent, the outputs are the caller's two slabs
    const TailD dh{d0h, bA, dimA, d0};
    bool rescaled = false;
    rc = relin_tail(c, Two<uint64_t>{out_c0 + k0 * bigpoly, out_c1 + k0 * bigpoly, polys}, c0hat,
                    limbs_addend ? Two<const uint64_t>{nullptr, nullptr, polys} : Two<const uint64_t>{d0, d1, polys},
                    W, dimP, dimB, logql, 2 * polys, wsTail, s, tail_mode, limbs_addend ? &dh : nullptr,                      // :67-77
                    rs >= 1 && rs <= 63 ? rs : 0u, &rescaled);   // the tail kernel shifts inside one word; other Deltas take the rescale kernel below
    if (rc) return rc;
    if (rs && !rescaled && (rc = gpq_he_rs(c, out_c0 + k0 * bigpoly, out_c1 + k0 * bigpoly, W, rs, logql - rs, polys, stream))) return rc;   // src/he-rescale.c:33-54
  }
  const unsigned lanes_used = lane.c ? 2u : 1u;
  if ((rc = gpq_peer_join(c, s, lane))) return rc;
  c->last_lanes = lanes_used;   // (after the nested entry points of the groups, which record their own)
  return launched(who);
}

// he_swk, src/he-automorphism.c:40-85: key-switch d1 with swk, c0 += d0, on big slabs, q_l = 2^logql.
// (poly_rot / poly_conj, src/poly.c:263-283, are coefficient permutations done by the caller.)
extern "C" int gpq_he_swk(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *d0, const uint64_t *d1,
                          const uint64_t *swk0, const uint64_t *swk1, unsigned W, unsigned logql, unsigned dimB, unsigned dimP,
                          unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dimB, batch, "gpq_he_swk");
  if (rc) return rc;
  if (!out_c0 || !out_c1 || !d0 || !d1 || !swk0 || !swk1 || !workspace || !logql || W < (logql + 63) / 64)
    return gpq_fail(GPQ_ERR_INVALID, "gpq_he_swk: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = c->n, bigpoly = (size_t)W * n;
  const unsigned m = batch < c->chunk ? batch : c->chunk;
  TailPlan tp;
  if ((rc = tail_plan(c, W, dimP, dimB, 2 * m, &tp))) return rc;
  char *w = (char *)workspace;
  uint64_t *sB = (uint64_t *)w; w += align64((size_t)m * 3 * dimB * n * 8);
  void *wsK = w; w += align64(gpq_keyswitch_workspace_bytes(c, dimB, m));
  void *wsTail = w;
  PeerLane lane;
  if (batch > m && (rc = gpq_peer_lane(c, s, gpq_lane_key(2, W, 0, dimB, dimP, m ^ (logql << 8)), [&](gpq_ctx *q) { return gpq_he_swk_workspace_bytes(q, W, dimB, dimP, m); }, &lane))) return rc;
  for (unsigned k0 = 0; k0 < batch; k0 += m) {
    const unsigned polys = batch - k0 < m ? batch - k0 : m;
    if (lane.c && ((k0 / m) & 1)) {                              // (as in gpq_he_mul)
      const size_t o = k0 * bigpoly;
      if ((rc = gpq_he_swk(lane.c, out_c0 + o, out_c1 + o, d0 + o, d1 + o, swk0, swk1, W, logql, dimB, dimP, polys, lane.ws, lane.s))) return rc;
      continue;
    }
    const size_t pb = (size_t)polys * dimB * n;
    uint64_t *d1hat = sB, *c0hat = sB + pb, *c1hat = sB + 2 * pb;
    if ((rc = launch_decompose(c, d1hat, d1 + k0 * bigpoly, W, 0, dimB, polys, s, c->lazy_decompose && c->logn > 12))) return rc;   // :60
    const LimbTab *tabsP = nullptr;
    int tail_mode = 0;
    if ((rc = tail_prescale_mode(c, dimP, dimB, &tabsP, &tail_mode))) return rc;
    {
      ScaledInverse scaled(c, tabsP);
      if ((rc = gpq_keyswitch(c, c0hat, c1hat, d1hat, swk0, swk1, dimB, polys, wsK, stream))) return rc;           // :61-65
    }
    // c0 (+ d0) and c1 (no addend) as one batch of 2 x polys polynomials                                           // :68-75
    if ((rc = relin_tail(c, Two<uint64_t>{out_c0 + k0 * bigpoly, out_c1 + k0 * bigpoly, polys}, c0hat, Two<const uint64_t>{d0 + k0 * bigpoly, nullptr, polys},
                         W, dimP, dimB, logql, 2 * polys, wsTail, s, tail_mode))) return rc;
  }
  const unsigned lanes_used = lane.c ? 2u : 1u;
  if ((rc = gpq_peer_join(c, s, lane))) return rc;
  c->last_lanes = lanes_used;   // (after the nested entry points of the groups, which record their own)
  return launched("gpq_he_swk");
}
extern "C" size_t gpq_he_swk_workspace_bytes(gpq_ctx *c, unsigned W, unsigned dimB, unsigned dimP, unsigned batch) {
  const unsigned m = batch < c->chunk ? batch : c->chunk;
  TailPlan tp;
  if (tail_plan(c, W, dimP, dimB, 2 * m, &tp) != GPQ_OK) return 0;
  return align64((size_t)m * 3 * dimB * c->n * 8) + align64(gpq_keyswitch_workspace_bytes(c, dimB, m)) + align64(tp.bytes);
}

// Tail of he_relin / he_swk alone (src/he-mult.c:67-77): out = smod(rdiv(poly_rns2mpi(chat, P*q_l), P) + d, q_l).
extern "C" size_t gpq_relin_tail_workspace_bytes(gpq_ctx *c, unsigned W, unsigned dimB, unsigned dimP, unsigned batch) {
  TailPlan tp;
  return tail_plan(c, W, dimP, dimB, batch, &tp) == GPQ_OK ? tp.bytes + 64 : 0;
}
extern "C" int gpq_relin_tail(gpq_ctx *c, uint64_t *out, const uint64_t *chat, const uint64_t *d, unsigned W, unsigned logql,
                              unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dimB, batch, "gpq_relin_tail");
  if (rc) return rc;
  if (!out || !chat || !workspace || !logql || W < (logql + 63) / 64) return gpq_fail(GPQ_ERR_INVALID, "gpq_relin_tail: bad arguments");
  return relin_tail(c, one_place(out), chat, one_place(d), W, dimP, dimB, logql, batch, workspace, (hipStream_t)stream);
}

// Big slabs between the kernels' layout (word-major) and rows of W words per coefficient (bridge_big_transpose): what the MPI-typed
// calls stage host data through.  to_rows = 0: rows -> words, 1: words -> rows.  n >= 64.
extern "C" int gpq_big_transpose(gpq_ctx *c, uint64_t *dst, const uint64_t *src, unsigned W, unsigned batch, int to_rows, void *stream) {
  if (!c || !dst || !src || W < 1 || W > 64 || batch < 1 || dst == src) return gpq_fail(GPQ_ERR_INVALID, "gpq_big_transpose: bad arguments");
  if (c->logn < 6) return gpq_fail(GPQ_ERR_INVALID, "gpq_big_transpose: n >= 64");
  BigTransposeArgs a{src, dst, W, c->logn, to_rows ? 1u : 0u};
  hipLaunchKernelGGL(bridge_big_transpose, dim3(c->n >> 6, batch), dim3(64), 0, (hipStream_t)stream, a);
  return launched("gpq_big_transpose");
}

// out = a + b / a - b / -a on `polys` big slabs of W words (two's complement, wrapping): the arithmetic of src/he-add.c before its mpi_smod
extern "C" int gpq_big_addsub(gpq_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, unsigned W, unsigned polys, int mode, void *stream) {
  if (!c || !out || !a || (mode != 2 && !b) || W < 1 || W > 64 || polys < 1 || mode < 0 || mode > 2) return gpq_fail(GPQ_ERR_INVALID, "gpq_big_addsub: bad arguments");
  BigAddSubArgs k{out, a, b, W, c->logn, (unsigned)mode};
  hipLaunchKernelGGL(bridge_big_addsub, dim3((c->n + 255) / 256, polys), dim3(256), 0, (hipStream_t)stream, k);
  return launched("gpq_big_addsub");
}

// The same tail for a caller that gives `chat` up as scratch: the CRT weights of the whole basis are put on it in place and the
// one-product kernel (tail_direct) finishes -- the form gpq_he_mul / gpq_he_swk reach without the extra pass, because their key
// switch delivers the weighted limbs.  Falls back to gpq_relin_tail's kernels when the product is not available (shape, settings).
extern "C" int gpq_relin_tail_overwriting(gpq_ctx *c, uint64_t *out, uint64_t *chat, const uint64_t *d, unsigned W, unsigned logql,
                                          unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dimB, batch, "gpq_relin_tail_overwriting");
  if (rc) return rc;
  if (!out || !chat || !workspace || !logql || W < (logql + 63) / 64) return gpq_fail(GPQ_ERR_INVALID, "gpq_relin_tail_overwriting: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  gpq_bridge_basis *bp, *bq;
  gpq_relin_tables *rt;
  if (dimB <= dimP) return gpq_fail(GPQ_ERR_INVALID, "relin: dimB=%u must exceed dimP=%u", dimB, dimP);
  if ((rc = get_basis(c, 0, dimP, &bp)) || (rc = get_basis(c, dimP, dimB - dimP, &bq)) || (rc = get_relin(c, dimP, dimB, &rt))) return rc;
  int mode = 0;
  if (c->bridge_mfma && c->logn >= 6 && c->tail_direct && !c->exact_crt) {
    if ((rc = get_relin_front(c, dimP, dimB, rt, bp, bq)) || (rt->d_bfrag && (rc = get_tail_direct(c, dimP, dimB, rt)))) return rc;
    if (rt->d_bfrag && rt->direct.d_bfrag && rt->d_scale) {
      LimbScaleArgs sc{c->d_tabs, chat, rt->d_scale, nullptr, dimB, c->logn};
      hipLaunchKernelGGL(bridge_limb_scale, dim3((c->n + 255) / 256, batch), dim3(256), 0, s, sc);
      mode = 3;
    }
  }
  return relin_tail(c, one_place(out), chat, one_place(d), W, dimP, dimB, logql, batch, workspace, s, mode);
}

// The relinearisation tail as one pass per coefficient (default) or as front + CRT kernels with Q's residues in memory between them;
// bit-identical (tests run both).
extern "C" int gpq_set_fused_tail(gpq_ctx *c, int on) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_fused_tail: null context");
  c->fuse_tail = on != 0;
  return GPQ_OK;
}

// gpq_he_mul / gpq_he_swk let their inverse transforms hand the CRT kernels limbs already multiplied by (P/p_d)^-1 (default on); off = the
// CRT kernels do that multiplication themselves.  Same results (tests run both).
extern "C" int gpq_set_prescale(gpq_ctx *c, int on) {      // 0: off, 1: the CRT weights only, 2: also w_j on the limbs above P for the relinearisation front, 3 (default): the one-product tail
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_prescale: null context");
  c->prescale = on != 0;
  c->prescale_upper = on >= 2;
  c->tail_direct = on >= 3;
  return GPQ_OK;
}

// gpq_he_mul / gpq_he_swk: bridge_stream.hpp's fused streaming kernels (default) or round 3's separate CRT / decompose / tail kernels.
// Same words either way (tests run both).
extern "C" int gpq_set_stream_bridge(gpq_ctx *c, int on) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_stream_bridge: null context");
  c->stream_bridge = on != 0;
  return GPQ_OK;
}

// gpq_he_mul / gpq_he_swk over more than one launch group: alternate groups on two streams (default) or all on the caller's.  Same words.
extern "C" int gpq_set_overlap(gpq_ctx *c, int on) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_overlap: null context");
  c->overlap = on < 0 ? -1 : (on != 0);
  return GPQ_OK;
}
// tests: make the next creation of the peer lane fail the way an allocation would (the call must run on one lane and say so once)
extern "C" int gpq_debug_fail_peer(gpq_ctx *c, int on) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_debug_fail_peer: null context");
  c->debug_peer_fail = on == 3 ? 0 : on;                         // 3: stop failing allocations but keep what was declined so far
  if (!on) { c->peer_failed = false; c->peer_ws_declined = 0; }
  return GPQ_OK;
}
// lanes the last gpq_he_mul / gpq_he_swk / gpq_he_mul_tensor / gpq_keyswitch call on this context ran on (1 or 2)
extern "C" unsigned gpq_last_lanes(const gpq_ctx *c) { return c ? c->last_lanes : 0; }

// gpq_he_mul: its own rns_decompose launches (src/he-mult.c:117-120, :59) may leave residues in (0, 3p) for the forward transforms that read them
// (default on; canonical with 0).  Same results: the transforms reduce lazily anyway.
extern "C" int gpq_set_lazy_decompose(gpq_ctx *c, int on) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_lazy_decompose: null context");
  c->lazy_decompose = on != 0;
  return GPQ_OK;
}

// Tests: bridge_stream.hpp's kernels also hand every coefficient whose index is a multiple of `every` to the exact kernels behind them (0: off).
extern "C" int gpq_debug_force_redo(gpq_ctx *c, unsigned every) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_debug_force_redo: null context");
  c->debug_force_redo = every;
  return GPQ_OK;
}

// Tests: force the exact (full-width) CRT kernel instead of the low-word fast path.
extern "C" int gpq_set_exact_crt(gpq_ctx *c, int on) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_exact_crt: null context");
  c->exact_crt = on != 0;
  return GPQ_OK;
}

// Diagnostics: how many of the first `count` coefficients the last fast CRT pass flagged for the exact kernel.
extern "C" long gpq_debug_redo_count(gpq_ctx *c, size_t count) {
  if (!c || !c->d_redo || count > c->redo_cap) return -1;
  std::vector<unsigned char> h(count);
  if (hipMemcpy(h.data(), c->d_redo, count, hipMemcpyDeviceToHost) != hipSuccess) return -2;
  long k = 0;
  for (unsigned char v : h) k += v != 0;
  return k;
}

// ---------------------------------------------------------------------------
// general modulus: poly_rns2mpi / poly_mul for any q given as little-endian words
// ---------------------------------------------------------------------------
extern "C" size_t gpq_poly_mul_general_workspace_bytes(gpq_ctx *c, unsigned dim, unsigned batch) {
  gpq_bridge_basis *b;
  if (get_basis(c, 0, dim, &b) != GPQ_OK) return 0;
  return gpq_poly_mul_workspace_bytes(c, dim, batch) + (size_t)batch * ((size_t)(b->WP + 1) << c->logn) * 8 + 3 * 8 * 64 + 64;
}

constexpr size_t kModConstWords = 3 * 64;   // M, mu, floor(M/2) of a general modulus, device side

// mpi_smod(x, q, floor(q/2)) of a big slab for an arbitrary q (host words): uploads the Barrett constants into
// `dconst` (kModConstWords device words) and launches the kernel.  x and out may be the same slab.
// What every entry point with a caller-supplied modulus checks before it launches anything: Wout = words of the result.
int check_modulus(const uint64_t *q_words, unsigned Lq, unsigned Wout) {
  if (!q_words || Lq < 1 || Lq > (unsigned)SMOD_MAXW / 2) return gpq_fail(GPQ_ERR_INVALID, "general modulus: bad word count %u", Lq);
  unsigned L = Lq;
  while (L > 1 && q_words[L - 1] == 0) --L;
  if (L == 1 && q_words[0] == 0) return gpq_fail(GPQ_ERR_INVALID, "zero modulus");
  if (Wout < L) return gpq_fail(GPQ_ERR_INVALID, "general modulus: %u words cannot hold a value mod q (%u words)", Wout, L);
  return GPQ_OK;
}

int launch_smod_general(gpq_ctx *c, uint64_t *out, unsigned Wout, const uint64_t *x, unsigned Wx, const uint64_t *q_words, unsigned Lq,
                        unsigned batch, uint64_t *dconst, hipStream_t s) {
  if (int rc = check_modulus(q_words, Lq, Wout)) return rc;
  Big M(q_words, q_words + Lq);
  while (M.size() > 1 && M.back() == 0) M.pop_back();
  const unsigned L = (unsigned)M.size();
  if (Wx > (unsigned)SMOD_MAXW) return gpq_fail(GPQ_ERR_UNSUPPORTED, "general modulus: value of %u words", Wx);
  Big mu = floor_pow2_div(128 * L, M), half = M;
  shr1(half);
  std::vector<uint64_t> consts(kModConstWords, 0);
  put(consts, 0, M, L); put(consts, 64, mu, L + 1); put(consts, 128, half, L);
  HIP_TRY(hipMemcpyAsync(dconst, consts.data(), consts.size() * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));   // consts is a local: keep it alive until the copy has happened
  SmodArgs a{x, out, dconst, dconst + 64, dconst + 128, Wx, Wout, L, c->logn};
  hipLaunchKernelGGL(bridge_smod_general, dim3((c->n + 63) / 64, batch), dim3(64), 0, s, a);
  return GPQ_OK;
}

// poly_rns2mpi (src/poly.c:109-120) for an arbitrary q: centred CRT value, then mpi_smod(., q, floor(q/2)).
// q = q_words[0..Lq) little-endian; `scratch` holds batch*(WP+1)*n words + 3*64 words.
extern "C" int gpq_rns_reconstruct_general(gpq_ctx *c, uint64_t *big, unsigned Wout, const uint64_t *slab, unsigned dim, unsigned batch,
                                           const uint64_t *q_words, unsigned Lq, void *scratch, void *stream) {
  int rc = check(c, dim, batch, "gpq_rns_reconstruct_general");
  if (rc) return rc;
  if (!big || !slab || !q_words || !scratch) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct_general: bad arguments");
  if ((rc = check_modulus(q_words, Lq, Wout))) return rc;
  gpq_bridge_basis *b;
  if ((rc = get_basis(c, 0, dim, &b))) return rc;
  const unsigned Wx = b->WP + 1;
  hipStream_t s = (hipStream_t)stream;
  uint64_t *xfull = (uint64_t *)scratch, *dconst = xfull + (size_t)batch * ((size_t)Wx << c->logn);
  if ((rc = launch_reconstruct(c, b, xfull, Wx, slab, dim, 0, batch, 0, true, nullptr, s))) return rc;   // centred mod P, full width
  if ((rc = launch_smod_general(c, big, Wout, xfull, Wx, q_words, Lq, batch, dconst, s))) return rc;
  return launched("gpq_rns_reconstruct_general");
}

// poly_mul (src/poly.c:84-107) for an arbitrary modulus q.
extern "C" int gpq_poly_mul_general(gpq_ctx *c, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned W, unsigned dim,
                                    const uint64_t *q_words, unsigned Lq, unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dim, batch, "gpq_poly_mul_general");
  if (rc) return rc;
  if (!r || !a || !b || !workspace) return gpq_fail(GPQ_ERR_INVALID, "gpq_poly_mul_general: bad arguments");
  if ((rc = check_modulus(q_words, Lq, W))) return rc;
  const size_t poly = (size_t)dim << c->logn;
  uint64_t *sa = (uint64_t *)workspace, *sb = sa + batch * poly, *sr = sb + batch * poly, *scratch = sr + batch * poly;
  if ((rc = gpq_rns_decompose(c, sa, a, W, dim, batch, stream))) return rc;
  if ((rc = gpq_rns_decompose(c, sb, b, W, dim, batch, stream))) return rc;
  if ((rc = gpq_poly_mul_rns(c, sr, sa, sb, dim, batch, stream))) return rc;
  return gpq_rns_reconstruct_general(c, r, W, sr, dim, batch, q_words, Lq, scratch, stream);
}

// ---------------------------------------------------------------------------
// he_mulpt, poly_rot / poly_conj, he_rot / he_conj at the big-slab level
// ---------------------------------------------------------------------------
extern "C" size_t gpq_he_mulpt_workspace_bytes(const gpq_ctx *c, unsigned dim, unsigned batch) {
  return 3ull * batch * ((size_t)dim << c->logn) * 8;
}

// he_mulpt, src/he-mult.c:159-196 (decl src/gpqhe.h:148): (c0, c1) * m per ciphertext, q_l = 2^logql.
// dim is the caller's (it depends on log2(pt->nu), a host double, :169).  Bookkeeping (:162-164) stays with the caller.
extern "C" int gpq_he_mulpt(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *c0, const uint64_t *c1, const uint64_t *m,
                            unsigned W, unsigned logql, unsigned dim, unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dim, batch, "gpq_he_mulpt");
  if (rc) return rc;
  if (!out_c0 || !out_c1 || !c0 || !c1 || !m || !workspace || !logql) return gpq_fail(GPQ_ERR_INVALID, "gpq_he_mulpt: bad arguments");
  const size_t poly = (size_t)dim << c->logn;
  uint64_t *s0 = (uint64_t *)workspace, *s1 = s0 + batch * poly, *sm = s1 + batch * poly;
  if ((rc = gpq_rns_decompose(c, sm, m, W, dim, batch, stream))) return rc;       // :176
  if ((rc = gpq_rns_decompose(c, s0, c0, W, dim, batch, stream))) return rc;      // :177
  if ((rc = gpq_rns_decompose(c, s1, c1, W, dim, batch, stream))) return rc;      // :178
  if ((rc = gpq_mulpt_rns(c, s0, s1, sm, s0, s1, dim, batch, stream))) return rc;                                          // :179-185
  if ((rc = gpq_rns_reconstruct(c, out_c0, W, s0, dim, batch, logql, stream))) return rc;                                 // :188
  return gpq_rns_reconstruct(c, out_c1, W, s1, dim, batch, logql, stream);                                                // :189
}

static int permute(gpq_ctx *c, uint64_t *r, const uint64_t *a, unsigned W, unsigned batch, unsigned long long power, int conj, void *stream) {
  if (!c || !r || !a || r == a || W < 1 || batch < 1) return gpq_fail(GPQ_ERR_INVALID, "poly_rot/poly_conj: bad arguments (not in place)");
  PermuteArgs p{a, r, W, c->logn, power, conj};
  hipLaunchKernelGGL(bridge_permute, dim3((c->n + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, p);
  return launched("bridge_permute");
}
// poly_rot, src/poly.c:263-275 (5^rot as :266-268 computes it, modulo 2^64 -- harmless since 2n divides 2^64)
extern "C" int gpq_poly_rot(gpq_ctx *c, uint64_t *r, const uint64_t *a, unsigned W, unsigned rot, unsigned batch, void *stream) {
  unsigned long long power = 1;
  for (unsigned j = 0; j < rot; ++j) power *= 5;
  return permute(c, r, a, W, batch, power, 0, stream);
}
// poly_conj, src/poly.c:277-283
extern "C" int gpq_poly_conj(gpq_ctx *c, uint64_t *r, const uint64_t *a, unsigned W, unsigned batch, void *stream) {
  return permute(c, r, a, W, batch, 1, 1, stream);
}

// he_add / he_sub / he_neg on one polynomial (src/he-add.c), q_l = 2^logql; r may alias a or b.
static int addsub(gpq_ctx *c, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned W, unsigned logql, unsigned mode, unsigned batch, void *stream) {
  if (!c || !r || !a || (mode < 2 && !b) || W < 1 || batch < 1 || !logql || logql > 64 * W) return gpq_fail(GPQ_ERR_INVALID, "gpq_big_add/sub/neg: bad arguments");
  AddSubArgs p{r, a, b, W, c->logn, logql, mode};
  hipLaunchKernelGGL(bridge_addsub, dim3((c->n + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, p);
  return launched("bridge_addsub");
}
extern "C" int gpq_big_add(gpq_ctx *c, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned W, unsigned logql, unsigned batch, void *stream) {
  return addsub(c, r, a, b, W, logql, 0, batch, stream);
}
extern "C" int gpq_big_sub(gpq_ctx *c, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned W, unsigned logql, unsigned batch, void *stream) {
  return addsub(c, r, a, b, W, logql, 1, batch, stream);
}
extern "C" int gpq_big_neg(gpq_ctx *c, uint64_t *r, const uint64_t *a, unsigned W, unsigned logql, unsigned batch, void *stream) {
  return addsub(c, r, a, nullptr, W, logql, 2, batch, stream);
}

// Key slabs as he_genswk stores them (src/he-kem.c:103-110): rns_decompose + ntt of a centred big-slab polynomial
// over dimevk limbs.  evk = uint64_t[batch][dimevk][n].
extern "C" int gpq_evk_pack(gpq_ctx *c, uint64_t *evk, const uint64_t *big, unsigned W, unsigned dimevk, unsigned batch, void *stream) {
  int rc = gpq_rns_decompose(c, evk, big, W, dimevk, batch, stream);
  return rc ? rc : gpq_ntt(c, evk, dimevk, batch, stream);
}

// he_genswk, src/he-kem.c:74-118, from the polynomials the reference samples on the host (p1 uniform mod P q_L, the error e) and
// the one the key hides (sp: s^2 for he_genrlk, the rotated / conjugated secret for he_genrk / he_genck):
//   swk.p0 = smod(-p1 * sk + e + P * sp, P q_L),  swk.p1 = smod(p1, P q_L),  both stored as rns_decompose + ntt over dimevk limbs.
// q_L = 2^logqL.  Big slabs of W words (W > bits(P q_L) / 64); one key per call.
namespace {
struct GenswkPlan { unsigned WPw, WF, dimmul, Lq; std::vector<uint64_t> PqL; size_t words; };
int genswk_plan(gpq_ctx *c, unsigned W, unsigned dimP, unsigned logqL, GenswkPlan *p) {
  gpq_bridge_basis *bp;
  int rc = get_basis(c, 0, dimP, &bp);
  if (rc) return rc;
  Big q = bp->h_P;
  const unsigned wsh = logqL / 64, bsh = logqL % 64;          // P << logqL
  Big sh(q.size() + wsh + 1, 0);
  for (size_t j = 0; j < q.size(); ++j) {
    sh[j + wsh] |= q[j] << bsh;
    if (bsh) sh[j + wsh + 1] |= q[j] >> (64 - bsh);
  }
  while (sh.size() > 1 && sh.back() == 0) sh.pop_back();
  p->PqL = sh; p->Lq = (unsigned)sh.size();
  p->WPw = (unsigned)bp->h_P.size();
  p->WF = W + p->WPw + 1;
  const unsigned nb = 64 * (p->Lq - 1) + (64 - __builtin_clzll(sh.back()));
  p->dimmul = (nb + c->logn) / 59 + 1;                          // src/he-kem.c:83
  if (W * 64 <= nb) return gpq_fail(GPQ_ERR_INVALID, "gpq_he_genswk: %u words cannot hold values mod P q_L (%u bits)", W, nb);
  if (W > 32 || p->WPw > (unsigned)GENSWK_MAXP || p->dimmul > c->nprimes)
    return gpq_fail(GPQ_ERR_UNSUPPORTED, "gpq_he_genswk: W=%u, P of %u words, %u limbs", W, p->WPw, p->dimmul);
  p->words = ((size_t)(3 * W + p->WF) << c->logn) + (p->WPw + 7) / 8 * 8 + kModConstWords;
  return GPQ_OK;
}
}  // namespace

extern "C" size_t gpq_he_genswk_workspace_bytes(gpq_ctx *c, unsigned W, unsigned dimP, unsigned logqL) {
  GenswkPlan p;
  if (!c || genswk_plan(c, W, dimP, logqL, &p) != GPQ_OK) return 0;
  return p.words * 8 + gpq_poly_mul_general_workspace_bytes(c, p.dimmul, 1);
}

extern "C" int gpq_he_genswk(gpq_ctx *c, uint64_t *evk0, uint64_t *evk1, const uint64_t *p1, const uint64_t *sk, const uint64_t *e,
                             const uint64_t *sp, unsigned W, unsigned dimP, unsigned logqL, unsigned dimevk, void *workspace, void *stream) {
  if (!c || !evk0 || !evk1 || !p1 || !sk || !e || !sp || !workspace || !logqL) return gpq_fail(GPQ_ERR_INVALID, "gpq_he_genswk: bad arguments");
  int rc = check(c, dimevk, 1, "gpq_he_genswk");
  if (rc) return rc;
  GenswkPlan gp;
  if ((rc = genswk_plan(c, W, dimP, logqL, &gp))) return rc;
  gpq_bridge_basis *bp;
  if ((rc = get_basis(c, 0, dimP, &bp))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = c->n;
  uint64_t *t = (uint64_t *)workspace, *x = t + W * n, *p0 = x + gp.WF * n, *p1c = p0 + W * n, *dP = p1c + W * n,
           *dconst = dP + (gp.WPw + 7) / 8 * 8;
  void *wsmul = dconst + kModConstWords;
  if ((rc = gpq_poly_mul_general(c, t, p1, sk, W, gp.dimmul, gp.PqL.data(), gp.Lq, 1, wsmul, stream))) return rc;      // :95
  HIP_TRY(hipMemcpyAsync(dP, bp->h_P.data(), gp.WPw * 8, hipMemcpyHostToDevice, s));
  GenswkArgs ga{t, e, sp, dP, x, W, gp.WPw, gp.WF, c->logn};
  hipLaunchKernelGGL(bridge_genswk_combine, dim3((c->n + 63) / 64), dim3(64), 0, s, ga);                                 // :96-98
  if ((rc = launch_smod_general(c, p0, W, x, gp.WF, gp.PqL.data(), gp.Lq, 1, dconst, s))) return rc;                    // :99
  if ((rc = launch_smod_general(c, p1c, W, p1, W, gp.PqL.data(), gp.Lq, 1, dconst, s))) return rc;                      // :100
  if ((rc = gpq_evk_pack(c, evk0, p0, W, dimevk, 1, stream)) || (rc = gpq_evk_pack(c, evk1, p1c, W, dimevk, 1, stream))) return rc;   // :103-110
  return launched("gpq_he_genswk");
}

// ---------------------------------------------------------------------------
// general q_l (any modulus, little-endian words) and Delta (any uint64_t): the same operations through the
// Barrett kernel.  Slow-path quality (key generation / unusual parameter sets); results follow the same
// reference semantics bit for bit.
// ---------------------------------------------------------------------------
namespace {

struct GenPlan { unsigned Wr, cnt, WQ, WF; size_t words, bytes; };

int gen_plan(gpq_ctx *c, unsigned W, unsigned dimP, unsigned dimB, unsigned polys, GenPlan *p) {
  gpq_bridge_basis *bp, *bq;
  int rc;
  if (dimB <= dimP) return gpq_fail(GPQ_ERR_INVALID, "relin: dimB=%u must exceed dimP=%u", dimB, dimP);
  if ((rc = get_basis(c, 0, dimP, &bp)) || (rc = get_basis(c, dimP, dimB - dimP, &bq))) return rc;
  p->Wr = bp->pbits / 64 + 1;
  p->cnt = dimB - dimP;
  p->WQ = bq->WP + 1;
  p->WF = (p->WQ > W ? p->WQ : W) + 1;
  p->words = (size_t)polys * ((size_t)(p->Wr + 2 * p->cnt + p->WQ + p->WF) << c->logn) + kModConstWords;
  p->bytes = p->words * 8 + ((size_t)polys << c->logn) + 64;
  return GPQ_OK;
}

int relin_tail_general(gpq_ctx *c, uint64_t *out, const uint64_t *chat, const uint64_t *dbig, unsigned W, unsigned dimP, unsigned dimB,
                       const uint64_t *ql_words, unsigned Lq, unsigned polys, void *ws, hipStream_t s) {
  GenPlan gp;
  int rc = gen_plan(c, W, dimP, dimB, polys, &gp);
  if (rc) return rc;
  gpq_bridge_basis *bp, *bq;
  gpq_relin_tables *rt;
  if ((rc = get_basis(c, 0, dimP, &bp)) || (rc = get_basis(c, dimP, gp.cnt, &bq)) || (rc = get_relin(c, dimP, dimB, &rt))) return rc;
  const size_t n = c->n;
  uint64_t *r = (uint64_t *)ws, *rhat = r + (size_t)polys * gp.Wr * n, *qhat = rhat + (size_t)polys * gp.cnt * n,
           *qfull = qhat + (size_t)polys * gp.cnt * n, *full = qfull + (size_t)polys * gp.WQ * n, *dconst = full + (size_t)polys * gp.WF * n;
  unsigned char *tie = (unsigned char *)(dconst + kModConstWords);
  if ((rc = launch_reconstruct(c, bp, r, gp.Wr, chat, dimB, 0, polys, 0, false, nullptr, s))) return rc;
  if ((rc = launch_decompose(c, rhat, r, gp.Wr, dimP, gp.cnt, polys, s))) return rc;
  ExactDivArgs e{c->d_tabs, chat, rhat, qhat, rt->d_pinv, dimB, dimP, gp.cnt, c->logn};
  { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_exactdiv, dim3((c->n + 255) / 256, polys, gp.cnt), dim3(256), 0, s, e); }
  if ((rc = launch_reconstruct(c, bq, qfull, gp.WQ, qhat, gp.cnt, 0, polys, 0, true, tie, s))) return rc;   // floor-quotient, full width
  AddRoundFullArgs ar{full, qfull, r, dbig, bp->d_phalf, bq->d_pmult + (size_t)5 * (bq->WP + 1), tie, gp.WF, gp.WQ, gp.Wr, W, c->logn};
  { ProfScope prof(c, GPQ_K_BRIDGE_EXACT, s); hipLaunchKernelGGL(bridge_addround_full, dim3((c->n + 255) / 256, polys), dim3(256), 0, s, ar); }
  if ((rc = launch_smod_general(c, out, W, full, gp.WF, ql_words, Lq, polys, dconst, s))) return rc;        // addm + smod, src/he-mult.c:73-76
  return launched("relin_tail_general");
}

}  // namespace

extern "C" size_t gpq_he_general_workspace_bytes(gpq_ctx *c, unsigned W, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch) {
  const unsigned m = batch < c->chunk ? batch : c->chunk;
  GenPlan gp;
  gpq_bridge_basis *bA;
  if (gen_plan(c, W, dimP, dimB, m, &gp) != GPQ_OK || get_basis(c, 0, dimA ? dimA : 1, &bA) != GPQ_OK) return 0;
  const size_t n = c->n;
  size_t b = 0;
  b += align64((size_t)m * 7 * dimA * n * 8);
  b += align64(gpq_tensor_workspace_bytes(c, dimA ? dimA : 1, m));
  b += align64((size_t)m * 3 * dimB * n * 8);
  b += align64(gpq_keyswitch_workspace_bytes(c, dimB, m));
  b += align64((size_t)m * 3 * W * n * 8);
  b += align64((size_t)m * (bA->WP + 1) * n * 8 + kModConstWords * 8);   // full-width CRT value of d0/d1/d2
  b += align64(gp.bytes);
  return b;
}

// he_rs for any Delta (uint64_t, as hectx_init takes it) and any q_l; scratch = 3*64 words.
extern "C" int gpq_he_rs_general(gpq_ctx *c, uint64_t *c0, uint64_t *c1, unsigned W, unsigned long long delta, const uint64_t *ql_words,
                                 unsigned Lq, unsigned batch, void *scratch, void *stream) {
  if (!c || !c0 || !c1 || !scratch || W < 1 || batch < 1 || !delta) return gpq_fail(GPQ_ERR_INVALID, "gpq_he_rs_general: bad arguments");
  if (int rcm = check_modulus(ql_words, Lq, W)) return rcm;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  for (uint64_t *p : {c0, c1}) {
    RdivWordArgs a{p, p, W, c->logn, delta};                                                            // src/he-rescale.c:45-46
    hipLaunchKernelGGL(bridge_rdiv_word, dim3((c->n + 255) / 256, batch), dim3(256), 0, s, a);
    if ((rc = launch_smod_general(c, p, W, p, W, ql_words, Lq, batch, (uint64_t *)scratch, s))) return rc;  // :47-48
  }
  return launched("gpq_he_rs_general");
}

extern "C" int gpq_relin_tail_general(gpq_ctx *c, uint64_t *out, const uint64_t *chat, const uint64_t *d, unsigned W, const uint64_t *ql_words,
                                      unsigned Lq, unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dimB, batch, "gpq_relin_tail_general");
  if (rc) return rc;
  if (!out || !chat || !workspace || !ql_words) return gpq_fail(GPQ_ERR_INVALID, "gpq_relin_tail_general: bad arguments");
  if (int rcm = check_modulus(ql_words, Lq, W)) return rcm;
  return relin_tail_general(c, out, chat, d, W, dimP, dimB, ql_words, Lq, batch, workspace, (hipStream_t)stream);
}

// he_mul for any q_l (src/he-mult.c:88-156); workspace from gpq_he_general_workspace_bytes.
extern "C" int gpq_he_mul_general(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *ct1c0, const uint64_t *ct1c1,
                                  const uint64_t *ct2c0, const uint64_t *ct2c1, const uint64_t *rlk0, const uint64_t *rlk1, unsigned W,
                                  const uint64_t *ql_words, unsigned Lq, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch,
                                  void *workspace, void *stream) {
  int rc = check(c, dimA, batch, "gpq_he_mul_general");
  if (rc || (rc = check(c, dimB, batch, "gpq_he_mul_general"))) return rc;
  if (!out_c0 || !out_c1 || !ct1c0 || !ct1c1 || !ct2c0 || !ct2c1 || !rlk0 || !rlk1 || !workspace || !ql_words)
    return gpq_fail(GPQ_ERR_INVALID, "gpq_he_mul_general: bad arguments");
  if (int rcm = check_modulus(ql_words, Lq, W)) return rcm;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = c->n, bigpoly = (size_t)W * n;
  const unsigned m = batch < c->chunk ? batch : c->chunk;
  GenPlan gp;
  gpq_bridge_basis *bA;
  if ((rc = gen_plan(c, W, dimP, dimB, m, &gp)) || (rc = get_basis(c, 0, dimA, &bA))) return rc;
  char *w = (char *)workspace;
  uint64_t *sA = (uint64_t *)w; w += align64((size_t)m * 7 * dimA * n * 8);
  void *wsT = w; w += align64(gpq_tensor_workspace_bytes(c, dimA, m));
  uint64_t *sB = (uint64_t *)w; w += align64((size_t)m * 3 * dimB * n * 8);
  void *wsK = w; w += align64(gpq_keyswitch_workspace_bytes(c, dimB, m));
  uint64_t *dbig = (uint64_t *)w; w += align64((size_t)m * 3 * W * n * 8);
  uint64_t *xfull = (uint64_t *)w; w += align64((size_t)m * (bA->WP + 1) * n * 8 + kModConstWords * 8);
  void *wsTail = w;
  const unsigned Wx = bA->WP + 1;
  for (unsigned k0 = 0; k0 < batch; k0 += m) {
    const unsigned polys = batch - k0 < m ? batch - k0 : m;
    const size_t pa = (size_t)polys * dimA * n, pb = (size_t)polys * dimB * n;
    uint64_t *h[4] = {sA, sA + pa, sA + 2 * pa, sA + 3 * pa};
    uint64_t *dh[3] = {sA + 4 * pa, sA + 5 * pa, sA + 6 * pa};   // d0hat, d1hat, d2hat
    const uint64_t *in[4] = {ct1c0, ct1c1, ct2c0, ct2c1};
    const bool square = ct1c0 == ct2c0 && ct1c1 == ct2c1;       // he_mul(&ct, &ct, &ct, rlk): as in gpq_he_mul
    for (int i = 0; i < (square ? 2 : 4); ++i)
      if ((rc = launch_decompose(c, h[i], in[i] + k0 * bigpoly, W, 0, dimA, polys, s))) return rc;
    if ((rc = gpq_he_mul_tensor(c, dh[0], dh[1], dh[2], h[0], h[1], square ? h[0] : h[2], square ? h[1] : h[3], dimA, polys, wsT, stream))) return rc;
    uint64_t *dd[3] = {dbig, dbig + polys * bigpoly, dbig + 2 * polys * bigpoly};
    uint64_t *dconst = xfull + (size_t)polys * Wx * n;
    for (int i = 0; i < 3; ++i) {                                                                   // :139-141
      if ((rc = launch_reconstruct(c, bA, xfull, Wx, dh[i], dimA, 0, polys, 0, true, nullptr, s))) return rc;
      if ((rc = launch_smod_general(c, dd[i], W, xfull, Wx, ql_words, Lq, polys, dconst, s))) return rc;
    }
    uint64_t *d2hat = sB, *c0hat = sB + pb, *c1hat = sB + 2 * pb;
    if ((rc = launch_decompose(c, d2hat, dd[2], W, 0, dimB, polys, s))) return rc;
    if ((rc = gpq_keyswitch(c, c0hat, c1hat, d2hat, rlk0, rlk1, dimB, polys, wsK, stream))) return rc;
    if ((rc = relin_tail_general(c, out_c0 + k0 * bigpoly, c0hat, dd[0], W, dimP, dimB, ql_words, Lq, polys, wsTail, s))) return rc;
    if ((rc = relin_tail_general(c, out_c1 + k0 * bigpoly, c1hat, dd[1], W, dimP, dimB, ql_words, Lq, polys, wsTail, s))) return rc;
  }
  return launched("gpq_he_mul_general");
}

// he_swk for any q_l (src/he-automorphism.c:40-85); workspace from gpq_he_general_workspace_bytes with dimA = 0.
extern "C" int gpq_he_swk_general(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *d0, const uint64_t *d1,
                                  const uint64_t *swk0, const uint64_t *swk1, unsigned W, const uint64_t *ql_words, unsigned Lq,
                                  unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dimB, batch, "gpq_he_swk_general");
  if (rc) return rc;
  if (!out_c0 || !out_c1 || !d0 || !d1 || !swk0 || !swk1 || !workspace || !ql_words) return gpq_fail(GPQ_ERR_INVALID, "gpq_he_swk_general: bad arguments");
  if (int rcm = check_modulus(ql_words, Lq, W)) return rcm;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = c->n, bigpoly = (size_t)W * n;
  const unsigned m = batch < c->chunk ? batch : c->chunk;
  char *w = (char *)workspace;
  uint64_t *sB = (uint64_t *)w; w += align64((size_t)m * 3 * dimB * n * 8);
  void *wsK = w; w += align64(gpq_keyswitch_workspace_bytes(c, dimB, m));
  void *wsTail = w;
  for (unsigned k0 = 0; k0 < batch; k0 += m) {
    const unsigned polys = batch - k0 < m ? batch - k0 : m;
    const size_t pb = (size_t)polys * dimB * n;
    uint64_t *d1hat = sB, *c0hat = sB + pb, *c1hat = sB + 2 * pb;
    if ((rc = launch_decompose(c, d1hat, d1 + k0 * bigpoly, W, 0, dimB, polys, s))) return rc;
    if ((rc = gpq_keyswitch(c, c0hat, c1hat, d1hat, swk0, swk1, dimB, polys, wsK, stream))) return rc;
    if ((rc = relin_tail_general(c, out_c0 + k0 * bigpoly, c0hat, d0 + k0 * bigpoly, W, dimP, dimB, ql_words, Lq, polys, wsTail, s))) return rc;
    if ((rc = relin_tail_general(c, out_c1 + k0 * bigpoly, c1hat, nullptr, W, dimP, dimB, ql_words, Lq, polys, wsTail, s))) return rc;
  }
  return launched("gpq_he_swk_general");
}

// he_mulpt for any q_l; workspace = gpq_he_mulpt_workspace_bytes + gpq_poly_mul_general_workspace_bytes.
extern "C" int gpq_he_mulpt_general(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *c0, const uint64_t *c1, const uint64_t *m,
                                    unsigned W, const uint64_t *ql_words, unsigned Lq, unsigned dim, unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dim, batch, "gpq_he_mulpt_general");
  if (rc) return rc;
  if (!out_c0 || !out_c1 || !c0 || !c1 || !m || !workspace || !ql_words) return gpq_fail(GPQ_ERR_INVALID, "gpq_he_mulpt_general: bad arguments");
  if (int rcm = check_modulus(ql_words, Lq, W)) return rcm;
  const size_t poly = (size_t)dim << c->logn;
  uint64_t *s0 = (uint64_t *)workspace, *s1 = s0 + batch * poly, *sm = s1 + batch * poly, *scratch = sm + batch * poly;
  if ((rc = gpq_rns_decompose(c, sm, m, W, dim, batch, stream)) || (rc = gpq_rns_decompose(c, s0, c0, W, dim, batch, stream)) ||
      (rc = gpq_rns_decompose(c, s1, c1, W, dim, batch, stream))) return rc;
  if ((rc = gpq_mulpt_rns(c, s0, s1, sm, s0, s1, dim, batch, stream))) return rc;
  if ((rc = gpq_rns_reconstruct_general(c, out_c0, W, s0, dim, batch, ql_words, Lq, scratch, stream))) return rc;
  return gpq_rns_reconstruct_general(c, out_c1, W, s1, dim, batch, ql_words, Lq, scratch, stream);
}
