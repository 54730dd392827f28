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

// CRT constants of the first `dim` primes: what struct rns_ctx node dim-1 holds (src/poly.h:35-38).
int get_basis(gpq_ctx *c, unsigned dim, gpq_bridge_basis **out) {
  auto it = c->bases.find(dim);
  if (it != c->bases.end()) { *out = &it->second; return GPQ_OK; }
  if (dim < 1 || dim > c->nprimes || dim > 63) return gpq_fail(GPQ_ERR_INVALID, "bridge: dim=%u outside 1..min(%u,63)", dim, c->nprimes);
  Big P{1};
  for (unsigned d = 0; d < dim; ++d) mul_small(P, c->p[d]);          // src/precomp.c:274-277
  int WP = 0;
  for (int w : kWP) if ((size_t)w >= P.size()) { WP = w; break; }
  if (!WP) return gpq_fail(GPQ_ERR_UNSUPPORTED, "bridge: P of %u limbs needs %zu words", dim, P.size());
  gpq_bridge_basis b;
  b.dim = dim; b.WP = WP; b.pbits = 64 * (unsigned)(P.size() - 1) + (64 - __builtin_clzll(P.back()));
  std::vector<uint64_t> phat((size_t)dim * WP), pinv(dim), pmult((size_t)6 * (WP + 1)), phalf(WP + 1);
  for (unsigned d = 0; d < dim; ++d) {
    Big q = P;
    divmod_small(q, c->p[d]);                                        // phat_d = P / p_d   :287
    put(phat, (size_t)d * WP, q, WP);
    pinv[d] = powm(mod_small(q, c->p[d]), c->p[d] - 2, c->p[d]);     // :288-289
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
  b.h_phat_inv = pinv;
  *out = &(c->bases[dim] = b);
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
    (void)hipFree(kv.second.d_pmult); (void)hipFree(kv.second.d_phalf);
  }
  c->bases.clear();
}

extern "C" unsigned gpq_big_words(unsigned bits) { return (bits + 63) / 64; }

extern "C" uint64_t gpq_ctx_phat_invmp(gpq_ctx *c, unsigned dim, unsigned d) {
  gpq_bridge_basis *b;
  if (get_basis(c, dim, &b) != GPQ_OK || d >= dim) return 0;
  return b->h_phat_inv[d];
}
extern "C" unsigned gpq_ctx_pbits(gpq_ctx *c, unsigned dim) {
  gpq_bridge_basis *b;
  return get_basis(c, dim, &b) == GPQ_OK ? b->pbits : 0;
}

extern "C" int gpq_rns_decompose(gpq_ctx *c, uint64_t *slab, const uint64_t *big, unsigned W, unsigned dim, unsigned batch, void *stream) {
  int rc = check(c, dim, batch, "gpq_rns_decompose");
  if (rc) return rc;
  if (!slab || !big || W < 1) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_decompose: bad arguments");
  DecomposeArgs a{c->d_tabs, big, slab, W, dim, c->logn};
  const dim3 grid((c->n + 255) / 256, batch), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (W <= 4) hipLaunchKernelGGL((bridge_decompose<4>), grid, block, 0, s, a);
  else if (W <= 16) hipLaunchKernelGGL((bridge_decompose<16>), grid, block, 0, s, a);
  else if (W <= 32) hipLaunchKernelGGL((bridge_decompose<32>), grid, block, 0, s, a);
  else return gpq_fail(GPQ_ERR_UNSUPPORTED, "gpq_rns_decompose: W=%u words (max 32)", W);
  return launched("gpq_rns_decompose");
}

extern "C" int gpq_rns_reconstruct(gpq_ctx *c, uint64_t *big, unsigned Wout, const uint64_t *slab, unsigned dim, unsigned batch,
                                   unsigned logq, void *stream) {
  int rc = check(c, dim, batch, "gpq_rns_reconstruct");
  if (rc) return rc;
  if (!slab || !big || Wout < 1) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct: bad arguments");
  gpq_bridge_basis *b;
  if ((rc = get_basis(c, dim, &b))) return rc;
  if (logq && Wout < (logq + 63) / 64) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct: %u words cannot hold a value mod 2^%u", Wout, logq);
  if (!logq && Wout * 64 < b->pbits + 1) return gpq_fail(GPQ_ERR_INVALID, "gpq_rns_reconstruct: %u words cannot hold a value mod P (%u bits)", Wout, b->pbits);
  ReconstructArgs a{c->d_tabs, slab, big, b->d_phat, b->d_phat_inv, b->d_pmult, b->d_phalf, dim, c->logn, Wout, logq};
  const dim3 grid((c->n + 127) / 128, batch), block(128);
  hipStream_t s = (hipStream_t)stream;
  switch (b->WP) {
    case 8: hipLaunchKernelGGL((bridge_reconstruct<8>), grid, block, 0, s, a); break;
    case 16: hipLaunchKernelGGL((bridge_reconstruct<16>), grid, block, 0, s, a); break;
    case 32: hipLaunchKernelGGL((bridge_reconstruct<32>), grid, block, 0, s, a); break;
    case 48: hipLaunchKernelGGL((bridge_reconstruct<48>), grid, block, 0, s, a); break;
    case 56: hipLaunchKernelGGL((bridge_reconstruct<56>), grid, block, 0, s, a); break;
    default: return gpq_fail(GPQ_ERR_UNSUPPORTED, "gpq_rns_reconstruct: WP=%d", b->WP);
  }
  return launched("gpq_rns_reconstruct");
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
