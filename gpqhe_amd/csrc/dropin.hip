// dropin.hip -- the reference-named link-time symbols (include/gpqhe_hip_compat.h).
//
// These keep the reference's calling convention exactly: one limb in host
// memory, `const struct rns_ctx *` for the prime, ring degree from the global
// `polyctx`, void return, abort() on misuse.  Each call ships the limb to the
// GPU, runs the same kernels as the slab API and ships the result back, so a
// GPQHE binary linked against this library produces the reference's bits
// while its own limb loops are still on the host.  The slab API
// (gpqhe_hip.h) is the fast path; this file is the compatibility path.
#include "../../include/gpqhe_hip.h"
#include "../../include/gpqhe_hip_compat.h"

#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

// The reference defines `polyctx` in src/precomp.c:41; a host program that is
// not GPQHE may not have it, hence weak.
extern "C" struct poly_ctx polyctx __attribute__((weak));

namespace {

unsigned g_logn_override = 0;
std::mutex g_mu;

struct Slot {
  const struct rns_ctx *key;
  uint64_t p;
  unsigned logn;
  const uint64_t *z, *zi;
  gpq_ctx *ctx;
  uint64_t *dev[3];
};
std::vector<Slot> g_slots;

[[noreturn]] void die(const char *what) {
  errno = EINVAL;  // the reference's error convention, src/reduce.c:95-100
  fprintf(stderr, "\033[1m\033[31merror:\033[0m \033[1m%s\033[0m. %s (%s)\n", strerror(errno), what, gpq_last_error());
  abort();
}

unsigned ring_logn() {
  if (&polyctx != nullptr && polyctx.n != 0) return polyctx.logn;
  if (g_logn_override) return g_logn_override;
  die("ring degree unknown: neither `polyctx` is linked nor gpq_dropin_set_logn() was called");
}

Slot &slot_for(const struct rns_ctx *rns) {
  const unsigned logn = ring_logn();
  for (Slot &s : g_slots)
    if (s.key == rns && s.p == rns->p && s.logn == logn && s.z == rns->zetas && s.zi == rns->zetas_inv) return s;
  Slot s;
  s.key = rns; s.p = rns->p; s.logn = logn; s.z = rns->zetas; s.zi = rns->zetas_inv; s.ctx = nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) die("no HIP device");
  const uint64_t *zs[1] = {rns->zetas}, *zis[1] = {rns->zetas_inv};
  if (gpq_ctx_create_from_tables(&s.ctx, logn, 1, &rns->p, zs, zis, dev) != GPQ_OK) die("cannot build device tables for this rns_ctx");
  for (auto &d : s.dev)
    if (gpq_malloc((void **)&d, sizeof(uint64_t) << logn) != GPQ_OK) die("device allocation failed");
  g_slots.push_back(s);
  return g_slots.back();
}

void transform(uint64_t a[], const struct rns_ctx *rns, bool inverse) {
  std::lock_guard<std::mutex> lock(g_mu);
  Slot &s = slot_for(rns);
  const size_t bytes = sizeof(uint64_t) << s.logn;
  // The slab kernels take canonical residues (what rns_decompose / poly_rns_mul hand the reference's ntt / invntt); a limb
  // holding any other word -- p itself out of an earlier ntt, or garbage -- runs src/ntt.c as written instead (gpq_ntt_reference).
  bool canonical = true;
  for (size_t i = 0; i < ((size_t)1 << s.logn); ++i) canonical &= a[i] < s.p;
  int rc = gpq_upload(s.dev[0], a, bytes, nullptr);
  if (rc == GPQ_OK && !canonical) rc = gpq_ntt_reference(s.ctx, s.dev[0], 1, 1, inverse ? 1 : 0, nullptr);
  else if (rc == GPQ_OK) rc = inverse ? gpq_invntt(s.ctx, s.dev[0], 1, 1, nullptr) : gpq_ntt(s.ctx, s.dev[0], 1, 1, nullptr);
  if (rc == GPQ_OK) rc = gpq_download(a, s.dev[0], bytes, nullptr);
  if (rc == GPQ_OK) rc = gpq_stream_sync(nullptr);
  if (rc != GPQ_OK) die(inverse ? "invntt failed" : "ntt failed");
}

void pointwise(uint64_t r[], const uint64_t a[], const uint64_t b[], const struct rns_ctx *rns, bool mul) {
  std::lock_guard<std::mutex> lock(g_mu);
  Slot &s = slot_for(rns);
  const size_t bytes = sizeof(uint64_t) << s.logn;
  int rc = gpq_upload(s.dev[0], a, bytes, nullptr);
  if (rc == GPQ_OK) rc = gpq_upload(s.dev[1], b, bytes, nullptr);
  if (rc == GPQ_OK)
    rc = mul ? gpq_rns_mul(s.ctx, s.dev[2], s.dev[0], s.dev[1], 1, 1, nullptr)
             : gpq_rns_add(s.ctx, s.dev[2], s.dev[0], s.dev[1], 1, 1, nullptr);
  if (rc == GPQ_OK) rc = gpq_download(r, s.dev[2], bytes, nullptr);
  if (rc == GPQ_OK) rc = gpq_stream_sync(nullptr);
  if (rc != GPQ_OK) die(mul ? "poly_rns_mul failed" : "poly_rns_add failed");
}

}  // namespace

extern "C" {

void ntt(uint64_t a[], const struct rns_ctx *rns) { transform(a, rns, false); }
void invntt(uint64_t a[], const struct rns_ctx *rns) { transform(a, rns, true); }
void poly_ntt(uint64_t a[], const struct rns_ctx *rns) { transform(a, rns, false); }
void poly_invntt(uint64_t a[], const struct rns_ctx *rns) { transform(a, rns, true); }

void poly_rns_add(uint64_t r[], const uint64_t a[], const uint64_t b[], const struct rns_ctx *rns) { pointwise(r, a, b, rns, false); }
void poly_rns_mul(uint64_t r[], const uint64_t a[], const uint64_t b[], const struct rns_ctx *rns) { pointwise(r, a, b, rns, true); }

void gpq_dropin_set_logn(unsigned int logn) { g_logn_override = logn; }

void gpq_dropin_reset(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (Slot &s : g_slots) {
    for (auto d : s.dev) (void)gpq_free(d);
    gpq_ctx_destroy(s.ctx);
  }
  g_slots.clear();
}

// ---- scalar host helpers of src/reduce.c (used by precomp.c at init) -------

// src/reduce.c:36-48: q^-1 mod 2^64 (Newton iteration; same value as the
// reference's 64-step product for odd q).
uint64_t montgomery_inv(uint64_t q) {
  uint64_t x = q;
  for (int i = 0; i < 6; ++i) x *= 2 - q * x;
  return x;
}

// src/reduce.c:59-66
uint64_t montgomery_reduce(gpq_u128 a, uint64_t q, int64_t qinv) {
  const uint64_t lo = (uint64_t)a, hi = (uint64_t)(a >> 64);
  const uint64_t m = lo * (uint64_t)qinv;
  const uint64_t t = (uint64_t)(((gpq_u128)m * q) >> 64);
  return hi - t + (hi < t ? q : 0);
}

// src/reduce.c:75-78
uint64_t barrett_inv(uint64_t q) {
  const unsigned bits = 64 - (unsigned)__builtin_clzll(q);
  return (uint64_t)(((gpq_u128)1 << (2 * bits)) / q);
}

// src/reduce.c:88-106 (aborts for moduli shorter than 32 bits, as there)
uint64_t barrett_reduce(gpq_u128 a, uint64_t q, uint64_t qinv) {
  const int shift = 2 * (64 - __builtin_clzll(q)) - 64;
  if (shift < 0) {
    errno = EINVAL;
    fprintf(stderr, "\033[1m\033[31merror:\033[0m \033[1m%s\033[0m. The number of bits of the modulus is too small.\n",
            strerror(errno));
    abort();
  }
  const uint64_t lo = (uint64_t)a, hi = (uint64_t)(a >> 64);
  const gpq_u128 est = ((((gpq_u128)lo * qinv) >> 64) + (gpq_u128)hi * qinv) >> shift;
  const uint64_t r = (uint64_t)(a - est * q);
  return r - (r >= q ? q : 0);
}

}  // extern "C"
