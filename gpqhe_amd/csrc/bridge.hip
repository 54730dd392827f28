// bridge.hip -- host side of the MPI <-> RNS bridge: CRT constants per prefix of
// the prime chain (rns_init, src/precomp.c:266-293) and the C ABI entry points
// gpq_rns_decompose / gpq_rns_reconstruct / gpq_poly_mul / gpq_he_rs.
#include "../../include/gpqhe_hip.h"
#include "engine_internal.hpp"
#include "bridge_kernels.hpp"

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
  auto it = c->bases.find(key);
  if (it != c->bases.end()) { *out = &it->second; return GPQ_OK; }
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
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMalloc((void **)&b.d_phat, phat.size() * 8));
  HIP_TRY(hipMalloc((void **)&b.d_phat_inv, pinv.size() * 8));
  HIP_TRY(hipMalloc((void **)&b.d_pmult, pmult.size() * 8));
  HIP_TRY(hipMalloc((void **)&b.d_phalf, phalf.size() * 8));
  HIP_TRY(hipMemcpy(b.d_phat, phat.data(), phat.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.d_phat_inv, pinv.data(), pinv.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.d_pmult, pmult.data(), pmult.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.d_phalf, phalf.data(), phalf.size() * 8, hipMemcpyHostToDevice));
  std::vector<uint64_t> inv128(2 * (size_t)dim);
  for (unsigned d = 0; d < dim; ++d) {
    const u128h q = ~(u128h)0 / c->p[first + d];                     // floor(2^128 / p_d): p_d does not divide 2^128
    inv128[2 * d] = (uint64_t)q; inv128[2 * d + 1] = (uint64_t)(q >> 64);
  }
  HIP_TRY(hipMalloc((void **)&b.d_inv128, inv128.size() * 8));
  HIP_TRY(hipMemcpy(b.d_inv128, inv128.data(), inv128.size() * 8, hipMemcpyHostToDevice));
  b.h_phat_inv = pinv;
  b.h_P = P;
  *out = &(c->bases[key] = b);
  return GPQ_OK;
}

int get_relin(gpq_ctx *c, unsigned dimP, unsigned dimB, gpq_relin_tables **out) {
  const auto key = std::make_pair(dimP, dimB);
  auto it = c->relins.find(key);
  if (it != c->relins.end()) { *out = &it->second; return GPQ_OK; }
  gpq_bridge_basis *bp;
  int rc = get_basis(c, 0, dimP, &bp);
  if (rc) return rc;
  std::vector<uint64_t> pinv(dimB - dimP);
  for (unsigned d = dimP; d < dimB; ++d) pinv[d - dimP] = powm(mod_small(bp->h_P, c->p[d]), c->p[d] - 2, c->p[d]);
  gpq_relin_tables t;
  HIP_TRY(hipMalloc((void **)&t.d_pinv, pinv.size() * 8));
  HIP_TRY(hipMemcpy(t.d_pinv, pinv.data(), pinv.size() * 8, hipMemcpyHostToDevice));
  *out = &(c->relins[key] = t);
  return GPQ_OK;
}

template <int WP>
void launch_exact(const ReconstructArgs &a, unsigned n, unsigned batch, hipStream_t s) {
  hipLaunchKernelGGL((bridge_reconstruct<WP>), dim3((n + 127) / 128, batch), dim3(128), 0, s, a);
}
template <int WL>
void launch_low(const ReconstructArgs &a, unsigned WPstride, unsigned char *redo, unsigned n, unsigned batch, hipStream_t s) {
  hipLaunchKernelGGL((bridge_reconstruct_low<WL>), dim3((n + 255) / 256, batch), dim3(256), 0, s, a, WPstride, redo);
}

int launch_reconstruct(gpq_ctx *c, const gpq_bridge_basis *b, uint64_t *big, unsigned Wout, const uint64_t *slab, unsigned slab_dim,
                       unsigned slab_first, unsigned batch, unsigned logq, bool centre, unsigned char *tie, hipStream_t s, int logn_override = -1) {
  const unsigned logn = logn_override < 0 ? c->logn : (unsigned)logn_override, n = 1u << logn;
  ReconstructArgs a{c->d_tabs, slab, big, b->d_phat, b->d_phat_inv, b->d_pmult, b->d_phalf, tie, nullptr, b->d_inv128,
                    b->dim, logn, Wout, logq, b->first, slab_dim, slab_first, centre ? 1u : 0u};
  // fast path: centred result modulo a power of two that needs fewer words than P has
  const unsigned need = (logq + 63) / 64;
  // (the centring threshold floor(P/2)/P differs from 1/2 by 1/(2P): negligible against the 2^-61 slack only for large P)
  const bool fast = logq && centre && !c->exact_crt && need + 1 < (unsigned)b->WP && need <= 16 && b->pbits >= 160;
  if (fast) {
    const size_t flags = (size_t)batch << logn;
    if (flags > c->redo_cap) {
      if (c->d_redo) HIP_TRY(hipFree(c->d_redo));
      HIP_TRY(hipMalloc((void **)&c->d_redo, flags));
      c->redo_cap = flags;
    }
    if (need <= 1) launch_low<1>(a, b->WP, c->d_redo, n, batch, s);
    else if (need <= 2) launch_low<2>(a, b->WP, c->d_redo, n, batch, s);
    else if (need <= 4) launch_low<4>(a, b->WP, c->d_redo, n, batch, s);
    else if (need <= 7) launch_low<7>(a, b->WP, c->d_redo, n, batch, s);     // q up to 2^448 (reference default 2^438)
    else if (need <= 10) launch_low<10>(a, b->WP, c->d_redo, n, batch, s);
    else if (need <= 14) launch_low<14>(a, b->WP, c->d_redo, n, batch, s);   // q up to 2^896 (headline 2^850)
    else launch_low<16>(a, b->WP, c->d_redo, n, batch, s);
    a.only = c->d_redo;   // exact kernel below redoes only the flagged coefficients
  }
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

int launch_decompose(gpq_ctx *c, uint64_t *slab, const uint64_t *big, unsigned W, unsigned limb0, unsigned dim, unsigned batch, hipStream_t s) {
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

int check(const gpq_ctx *c, unsigned dim, unsigned batch, const char *who) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "%s: null context", who);
  if (dim < 1 || dim > c->nprimes) return gpq_fail(GPQ_ERR_INVALID, "%s: dim=%u outside 1..%u", who, dim, c->nprimes);
  if (batch < 1) return gpq_fail(GPQ_ERR_INVALID, "%s: empty batch", who);
  return GPQ_OK;
}
int launched(const char *who) {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GPQ_OK : gpq_fail(GPQ_ERR_HIP, "%s: launch failed: %s", who, hipGetErrorString(e));
}

}  // namespace

void gpq_bridge_release(gpq_ctx *c) {
  for (auto &kv : c->bases) {
    (void)hipFree(kv.second.d_phat); (void)hipFree(kv.second.d_phat_inv);
    (void)hipFree(kv.second.d_pmult); (void)hipFree(kv.second.d_phalf); (void)hipFree(kv.second.d_inv128);
  }
  if (c->d_redo) (void)hipFree(c->d_redo);
  c->d_redo = nullptr; c->redo_cap = 0;
  c->bases.clear();
  for (auto &kv : c->relins) (void)hipFree(kv.second.d_pinv);
  c->relins.clear();
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
  HIP_TRY(hipSetDevice(c->device));
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

// src/he-mult.c:67-77 (d != null: c = rdiv(c,P) + d) and src/he-automorphism.c:68-76, for q_l = 2^logql.
int relin_tail(gpq_ctx *c, uint64_t *out, const uint64_t *chat, const uint64_t *dbig, unsigned W, unsigned dimP, unsigned dimB,
               unsigned logql, unsigned polys, void *ws, hipStream_t s) {
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
  // r = x mod P from the first dimP limbs, unsigned
  if ((rc = launch_reconstruct(c, bp, r, tp.Wr, chat, dimB, 0, polys, 0, false, nullptr, s))) return rc;
  if ((rc = launch_decompose(c, rhat, r, tp.Wr, dimP, tp.cnt, polys, s))) return rc;
  ExactDivArgs e{c->d_tabs, chat, rhat, qhat, rt->d_pinv, dimB, dimP, tp.cnt, c->logn};
  hipLaunchKernelGGL(bridge_exactdiv, dim3((c->n + 255) / 256, polys, tp.cnt), dim3(256), 0, s, e);
  // Q = (x - r)/P over the remaining limbs, centred, already reduced smod 2^logql
  if ((rc = launch_reconstruct(c, bq, qc, W, qhat, tp.cnt, 0, polys, logql, true, tie, s))) return rc;
  AddRoundArgs ar{out, qc, r, dbig, bp->d_phalf, bq->d_pmult + (size_t)5 * (bq->WP + 1), tie, W, tp.Wr, c->logn, logql};
  hipLaunchKernelGGL(bridge_addround, dim3((c->n + 255) / 256, polys), dim3(256), 0, s, ar);
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
  if (tail_plan(c, W, dimP, dimB, m, &tp) != GPQ_OK) return 0;
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
extern "C" int gpq_he_mul(gpq_ctx *c, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *ct1c0, const uint64_t *ct1c1,
                          const uint64_t *ct2c0, const uint64_t *ct2c1, const uint64_t *rlk0, const uint64_t *rlk1, unsigned W,
                          unsigned logql, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream) {
  int rc = check(c, dimA, batch, "gpq_he_mul");
  if (rc || (rc = check(c, dimB, batch, "gpq_he_mul"))) return rc;
  if (!out_c0 || !out_c1 || !ct1c0 || !ct1c1 || !ct2c0 || !ct2c1 || !rlk0 || !rlk1 || !workspace || !logql || W < (logql + 63) / 64)
    return gpq_fail(GPQ_ERR_INVALID, "gpq_he_mul: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = c->n, bigpoly = (size_t)W * n;
  const unsigned m = batch < c->chunk ? batch : c->chunk;
  TailPlan tp;
  if ((rc = tail_plan(c, W, dimP, dimB, m, &tp))) return rc;
  char *w = (char *)workspace;
  uint64_t *sA = (uint64_t *)w; w += align64((size_t)m * 7 * dimA * n * 8);
  void *wsT = w; w += align64(gpq_tensor_workspace_bytes(c, dimA, m));
  uint64_t *sB = (uint64_t *)w; w += align64((size_t)m * 3 * dimB * n * 8);
  void *wsK = w; w += align64(gpq_keyswitch_workspace_bytes(c, dimB, m));
  uint64_t *dbig = (uint64_t *)w; w += align64((size_t)m * 3 * W * n * 8);
  void *wsTail = w;
  gpq_bridge_basis *bA;
  if ((rc = get_basis(c, 0, dimA, &bA))) return rc;
  for (unsigned k0 = 0; k0 < batch; k0 += m) {
    const unsigned polys = batch - k0 < m ? batch - k0 : m;
    const size_t pa = (size_t)polys * dimA * n, pb = (size_t)polys * dimB * n;
    uint64_t *h[4] = {sA, sA + pa, sA + 2 * pa, sA + 3 * pa};
    uint64_t *d0h = sA + 4 * pa, *d1h = sA + 5 * pa, *d2h = sA + 6 * pa;
    const uint64_t *in[4] = {ct1c0, ct1c1, ct2c0, ct2c1};
    for (int i = 0; i < 4; ++i)                                                             // :117-120
      if ((rc = launch_decompose(c, h[i], in[i] + k0 * bigpoly, W, 0, dimA, polys, s))) return rc;
    if ((rc = gpq_he_mul_tensor(c, d0h, d1h, d2h, h[0], h[1], h[2], h[3], dimA, polys, wsT, stream))) return rc;  // :121-136
    uint64_t *d0 = dbig, *d1 = dbig + polys * bigpoly, *d2 = dbig + 2 * polys * bigpoly;
    if ((rc = launch_reconstruct(c, bA, d0, W, d0h, dimA, 0, polys, logql, true, nullptr, s))) return rc;          // :139
    if ((rc = launch_reconstruct(c, bA, d2, W, d2h, dimA, 0, polys, logql, true, nullptr, s))) return rc;          // :140
    if ((rc = launch_reconstruct(c, bA, d1, W, d1h, dimA, 0, polys, logql, true, nullptr, s))) return rc;          // :141
    // he_relin, :40-85
    uint64_t *d2hat = sB, *c0hat = sB + pb, *c1hat = sB + 2 * pb;
    if ((rc = launch_decompose(c, d2hat, d2, W, 0, dimB, polys, s))) return rc;                                    // :59
    if ((rc = gpq_keyswitch(c, c0hat, c1hat, d2hat, rlk0, rlk1, dimB, polys, wsK, stream))) return rc;             // :60-64
    if ((rc = relin_tail(c, out_c0 + k0 * bigpoly, c0hat, d0, W, dimP, dimB, logql, polys, wsTail, s))) return rc; // :67-77
    if ((rc = relin_tail(c, out_c1 + k0 * bigpoly, c1hat, d1, W, dimP, dimB, logql, polys, wsTail, s))) return rc;
  }
  return launched("gpq_he_mul");
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
  if ((rc = tail_plan(c, W, dimP, dimB, m, &tp))) return rc;
  char *w = (char *)workspace;
  uint64_t *sB = (uint64_t *)w; w += align64((size_t)m * 3 * dimB * n * 8);
  void *wsK = w; w += align64(gpq_keyswitch_workspace_bytes(c, dimB, m));
  void *wsTail = w;
  for (unsigned k0 = 0; k0 < batch; k0 += m) {
    const unsigned polys = batch - k0 < m ? batch - k0 : m;
    const size_t pb = (size_t)polys * dimB * n;
    uint64_t *d1hat = sB, *c0hat = sB + pb, *c1hat = sB + 2 * pb;
    if ((rc = launch_decompose(c, d1hat, d1 + k0 * bigpoly, W, 0, dimB, polys, s))) return rc;                     // :60
    if ((rc = gpq_keyswitch(c, c0hat, c1hat, d1hat, swk0, swk1, dimB, polys, wsK, stream))) return rc;             // :61-65
    if ((rc = relin_tail(c, out_c0 + k0 * bigpoly, c0hat, d0 + k0 * bigpoly, W, dimP, dimB, logql, polys, wsTail, s))) return rc;  // :68-75
    if ((rc = relin_tail(c, out_c1 + k0 * bigpoly, c1hat, nullptr, W, dimP, dimB, logql, polys, wsTail, s))) return rc;
  }
  return launched("gpq_he_swk");
}
extern "C" size_t gpq_he_swk_workspace_bytes(gpq_ctx *c, unsigned W, unsigned dimB, unsigned dimP, unsigned batch) {
  const unsigned m = batch < c->chunk ? batch : c->chunk;
  TailPlan tp;
  if (tail_plan(c, W, dimP, dimB, m, &tp) != GPQ_OK) return 0;
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
  return relin_tail(c, out, chat, d, W, dimP, dimB, logql, batch, workspace, (hipStream_t)stream);
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
int launch_smod_general(gpq_ctx *c, uint64_t *out, unsigned Wout, const uint64_t *x, unsigned Wx, const uint64_t *q_words, unsigned Lq,
                        unsigned batch, uint64_t *dconst, hipStream_t s) {
  if (!q_words || Lq < 1 || Lq > (unsigned)SMOD_MAXW / 2) return gpq_fail(GPQ_ERR_INVALID, "general modulus: bad word count %u", Lq);
  Big M(q_words, q_words + Lq);
  while (M.size() > 1 && M.back() == 0) M.pop_back();
  const unsigned L = (unsigned)M.size();
  if (L == 1 && M[0] == 0) return gpq_fail(GPQ_ERR_INVALID, "zero modulus");
  if (Wx > (unsigned)SMOD_MAXW) return gpq_fail(GPQ_ERR_UNSUPPORTED, "general modulus: value of %u words", Wx);
  if (Wout < L) return gpq_fail(GPQ_ERR_INVALID, "general modulus: %u words cannot hold a value mod q (%u words)", Wout, L);
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
  if ((rc = gpq_ntt(c, sm, dim, batch, stream)) || (rc = gpq_ntt(c, s0, dim, batch, stream)) || (rc = gpq_ntt(c, s1, dim, batch, stream))) return rc;  // :179-181
  if ((rc = gpq_rns_mul(c, s0, s0, sm, dim, batch, stream)) || (rc = gpq_invntt(c, s0, dim, batch, stream))) return rc;   // :182-183
  if ((rc = gpq_rns_mul(c, s1, s1, sm, dim, batch, stream)) || (rc = gpq_invntt(c, s1, dim, batch, stream))) return rc;   // :184-185
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
  hipLaunchKernelGGL(bridge_exactdiv, dim3((c->n + 255) / 256, polys, gp.cnt), dim3(256), 0, s, e);
  if ((rc = launch_reconstruct(c, bq, qfull, gp.WQ, qhat, gp.cnt, 0, polys, 0, true, tie, s))) return rc;   // floor-quotient, full width
  AddRoundFullArgs ar{full, qfull, r, dbig, bp->d_phalf, bq->d_pmult + (size_t)5 * (bq->WP + 1), tie, gp.WF, gp.WQ, gp.Wr, W, c->logn};
  hipLaunchKernelGGL(bridge_addround_full, dim3((c->n + 255) / 256, polys), dim3(256), 0, s, ar);
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
    for (int i = 0; i < 4; ++i)
      if ((rc = launch_decompose(c, h[i], in[i] + k0 * bigpoly, W, 0, dimA, polys, s))) return rc;
    if ((rc = gpq_he_mul_tensor(c, dh[0], dh[1], dh[2], h[0], h[1], h[2], h[3], dimA, polys, wsT, stream))) return rc;
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
  const size_t poly = (size_t)dim << c->logn;
  uint64_t *s0 = (uint64_t *)workspace, *s1 = s0 + batch * poly, *sm = s1 + batch * poly, *scratch = sm + batch * poly;
  if ((rc = gpq_rns_decompose(c, sm, m, W, dim, batch, stream)) || (rc = gpq_rns_decompose(c, s0, c0, W, dim, batch, stream)) ||
      (rc = gpq_rns_decompose(c, s1, c1, W, dim, batch, stream))) return rc;
  if ((rc = gpq_ntt(c, sm, dim, batch, stream)) || (rc = gpq_ntt(c, s0, dim, batch, stream)) || (rc = gpq_ntt(c, s1, dim, batch, stream))) return rc;
  if ((rc = gpq_rns_mul(c, s0, s0, sm, dim, batch, stream)) || (rc = gpq_invntt(c, s0, dim, batch, stream))) return rc;
  if ((rc = gpq_rns_mul(c, s1, s1, sm, dim, batch, stream)) || (rc = gpq_invntt(c, s1, dim, batch, stream))) return rc;
  if ((rc = gpq_rns_reconstruct_general(c, out_c0, W, s0, dim, batch, ql_words, Lq, scratch, stream))) return rc;
  return gpq_rns_reconstruct_general(c, out_c1, W, s1, dim, batch, ql_words, Lq, scratch, stream);
}
