// mpi_shim.hip -- poly_mul / he_mul / he_rs / he_rescale / he_moddown with the reference's own
// MPI-typed signatures (src/poly.h:86-87, src/gpqhe.h:136-137,147), so that GPQHE objects which
// call them link against libgpqhe_hip.so unchanged.  The work itself is the device code behind
// gpq_poly_mul / gpq_he_mul / gpq_he_rs; this file only moves coefficients between libgcrypt MPIs
// and big slabs and mirrors the host-side bookkeeping (ct->l, nu, B).
//
// libgcrypt is reached through its runtime ABI only (dlsym on the library the host program has
// loaded, or libgcrypt.so.20): this image ships no <gcrypt.h>, and none is needed.
#include "../../include/gpqhe_hip.h"
#include "../../include/gpqhe_hip_compat.h"

#include <dlfcn.h>
#include <sys/random.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

extern "C" struct poly_ctx polyctx __attribute__((weak));   // src/precomp.c:41
extern "C" struct he_ctx hectx __attribute__((weak));       // src/precomp.c:47
// the reference's samplers (src/sample.c; externs at src/he-kem.c:33-34): key generation draws from them in the reference's order
extern "C" void sample_error(poly_mpi_t *r) __attribute__((weak));
extern "C" void sample_uniform(poly_mpi_t *r, const gpq_MPI q) __attribute__((weak));

#include "mpi_convert.hpp"

// Every copy, kernel and event of the MPI-typed calls goes to the legacy null stream (stream argument nullptr), from the calling thread and
// from the conversion threads alike: that ONE stream is what orders a range's upload before the transposes and kernels that read it, and a
// pooled buffer's reuse behind its last reader.  With per-thread default streams nullptr would mean a different stream in every thread.
#if defined(HIP_API_PER_THREAD_DEFAULT_STREAM)
#error "mpi_shim.hip relies on the legacy (process-wide) null stream: build without -fgpu-default-stream=per-thread"
#endif

#include "shim_staging.hpp"
#include "shim_keys.hpp"
#include "shim_polys.hpp"

namespace {

// The MPI-typed entry points share the staging buffers, the buffer pools, the key cache and the worker threads: one call at a
// time (the reference itself is single-threaded; a second host thread simply waits here).  Recursive: he_rescale -> he_rs etc.
std::recursive_mutex g_call_mu;
#define SHIM_CALL() std::lock_guard<std::recursive_mutex> shim_call_lock(g_call_mu)

}  // namespace

extern "C" {

// index of this node's prime in the chain: a node with `dim` limbs carries prime dim-1 (src/precomp.c:266-293)
static unsigned limb_of(gpq_ctx *c, const struct rns_ctx *rns, const char *who) {
  if (!rns || rns->dim < 1 || rns->dim > gpq_ctx_nprimes(c) || gpq_ctx_const(c, rns->dim - 1, 0) != rns->p) die(who);
  return rns->dim - 1;
}

// src/rns.c:37-48 -- one limb: ahat[i] = a[i] mod rns->p, non-negative
void rns_decompose(uint64_t ahat[], const gpq_MPI a[], const struct rns_ctx *rns) {
  SHIM_CALL();
  need_gcrypt();
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n, limb = limb_of(c, rns, "rns_decompose: the node is not one of the caller's prime chain");
  poly_mpi_t view{const_cast<gpq_MPI *>(a)};
  const unsigned W = max_bits(&view, n) / 64 + 1;
  if (W > 32) die("rns_decompose: coefficients wider than 2047 bits");
  std::vector<uint64_t> h((size_t)W * n);
  to_slab(h.data(), &view, n, W);
  DevBuf big(h.size() * 8), out((size_t)n * 8);
  up(big, h);
  if (gpq_rns_decompose_limbs(c, out.u64(), big.u64(), W, limb, 1, 1, nullptr) != GPQ_OK) die("rns_decompose failed");
  if (gpq_download(ahat, out.p, (size_t)n * 8, nullptr) != GPQ_OK || gpq_stream_sync(nullptr) != GPQ_OK) die("download failed");
}

static void set_from_words(MPI r, const uint64_t *w, unsigned W) {
  poly_mpi_t one{&r};
  from_slab(&one, w, 1, W);
}

// src/rns.c:60-75 -- coefficient i of a slab with rns->dim limbs, value in [0, P)
void rns_reconstruct(gpq_MPI a, const uint64_t ahat[], const unsigned int i, const struct rns_ctx *rns) {
  SHIM_CALL();
  need_gcrypt();
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n;
  (void)limb_of(c, rns, "rns_reconstruct: the node is not one of the caller's prime chain");
  if (i >= n) die("rns_reconstruct: coefficient index out of range");
  const unsigned dim = rns->dim, W = gpq_ctx_pbits(c, dim) / 64 + 2;
  std::vector<uint64_t> col(dim), w(W);
  for (unsigned d = 0; d < dim; ++d) col[d] = ahat[(size_t)d * n + i];
  if (gpq_rns_reconstruct_one(c, w.data(), W, col.data(), dim) != GPQ_OK) die("rns_reconstruct failed");
  set_from_words(a, w.data(), W);
}

// src/poly.c:109-120 -- every coefficient: reconstruct, centre mod P, centre mod q
void poly_rns2mpi(poly_mpi_t *r, const poly_rns_t *rhat, const struct rns_ctx *rns, const gpq_MPI q) {
  SHIM_CALL();
  need_gcrypt();
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n;
  (void)limb_of(c, rns, "poly_rns2mpi: the node is not one of the caller's prime chain");
  const unsigned dim = rns->dim;
  const std::vector<uint64_t> qw = words_of(q, "poly_rns2mpi: the modulus must be positive");
  const unsigned nbq = G.mpi_get_nbits(q), W = nbq / 64 + 1;
  if (qw.size() > 48) die("poly_rns2mpi: modulus wider than 48 words");
  DevBuf slab((size_t)dim * n * 8), big((size_t)W * n * 8), scratch(gpq_poly_mul_general_workspace_bytes(c, dim, 1));
  if (gpq_upload(slab.p, rhat->coeffs, (size_t)dim * n * 8, nullptr) != GPQ_OK) die("upload failed");
  const int rc = is_pow2(qw) ? gpq_rns_reconstruct(c, big.u64(), W, slab.u64(), dim, 1, nbq - 1, nullptr)
                             : gpq_rns_reconstruct_general(c, big.u64(), W, slab.u64(), dim, 1, qw.data(), (unsigned)qw.size(), scratch.p, nullptr);
  if (rc != GPQ_OK) die("poly_rns2mpi failed");
  std::vector<uint64_t> h((size_t)W * n);
  down(h, big);
  from_slab(r, h.data(), n, W);
}

// src/poly.c:84-107
void poly_mul(poly_mpi_t *r, const poly_mpi_t *a, const poly_mpi_t *b, const unsigned int dim, const gpq_MPI q) {
  SHIM_CALL();
  need_gcrypt();
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n;
  const std::vector<uint64_t> qw = words_of(q, "poly_mul: the modulus must be positive");
  const unsigned nbq = G.mpi_get_nbits(q);
  const poly_mpi_t *in[2] = {a, b};
  poly_mpi_t *out[1] = {r};
  const bool pow2 = is_pow2(qw), same = a->coeffs == b->coeffs;
  // one pass at W words per coefficient; with `kept`, from the resident copies of both operands (false: one of them has outgrown its copy)
  auto pass = [&](unsigned W, bool kept) -> bool {
    if (W > 32) die("poly_mul: coefficients wider than 2047 bits");
    const size_t big = (size_t)W * n;
    HostBuf s0(big * 8), s1(big * 8), t0s(big * 8);
    DevBuf da(big * 8), db(big * 8), dr(big * 8), ws(gpq_poly_mul_general_workspace_bytes(c, dim, 1));
    const DevBuf *dd[2] = {&da, &db}, *oo[1] = {&dr};
    const HostBuf *ss[2] = {&s0, &s1}, *ts[1] = {&t0s};
    Operands ops(same ? 1 : 2, in, dd, ss, n, W);
    ops.prepare(kept);
    auto device_work = [&]() {
      const uint64_t *xa = ops.x[0], *xb = same ? xa : ops.x[1];
      const int rc = pow2 ? gpq_poly_mul(c, dr.u64(), xa, xb, W, dim, nbq - 1, 1, ws.p, nullptr)
                          : gpq_poly_mul_general(c, dr.u64(), xa, xb, W, dim, qw.data(), (unsigned)qw.size(), 1, ws.p, nullptr);
      if (rc != GPQ_OK) die("poly_mul failed");   // q = P*q_L in he_genswk (src/he-kem.c:95) takes the general path
      download_issue(ts, oo, 1, n, W);
    };
    device_work();
    if (ops.resident && ops.recheck()) {
      if (ops.misfits) { (void)gpq_stream_sync(nullptr); return false; }   // the copies queued into / out of this pass's page-locked buffers must land before the buffers go back to the pool
      device_work();
    }
    std::vector<uint64_t> oprints((size_t)ops.nt, 0);
    download_convert(out, ts, 1, n, W, oprints.data());
    remember_results(out, oo, 1, n, W, oprints);
    return true;
  };
  bool done = false;
  if (poly_cache_on(n)) {                                   // he_dec multiplies a chained ciphertext's c1 with the same secret key every time
    const PolySlot *ka = resident_poly(a, n, 0), *kb = resident_poly(b, n, 0);
    if (ka && kb && ka->trusted && kb->trusted && ka->W == kb->W && ka->W >= nbq / 64 + 1) done = pass(ka->W, true);
  }
  if (!done) {
    unsigned bits = max_bits(a, n), bb = max_bits(b, n);
    if (bb > bits) bits = bb;
    if (nbq > bits) bits = nbq;
    pass(bits / 64 + 1, false);
  }
}

// src/he-encrypt.c:105-125: m = c1 * sk + c0, centred mod q_l.  The reference multiplies through poly_mul and then adds and centres with
// 2n libgcrypt calls on the host; here the product never leaves the device before the sum is centred, and a chained ciphertext (and the
// secret key after its first use) is resident.
void he_dec(struct he_pt *pt, const struct he_ct *ct, const poly_mpi_t *sk) {
  SHIM_CALL();
  need_gcrypt();
  if (&hectx == nullptr || !hectx.q) die("`hectx` is not initialised (hectx_init first)");
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n, l = ct->l;
  const std::vector<uint64_t> qw = words_of(hectx.q[l], "he_dec: q_l must be positive");
  const bool pow2 = is_pow2(qw);
  const unsigned nbq = G.mpi_get_nbits(hectx.q[l]), logql = nbq - 1, dim = nbq / 59 + 1;        // :113
  const poly_mpi_t *in[3] = {&ct->c1, sk, &ct->c0};
  poly_mpi_t *out[1] = {&pt->m};
  pt->nu = ct->nu;                                                                               // :108
  if (logql == 0) {                                         // q_l = 1: mpi_smod leaves -1 everywhere (see he_rs)
    for (unsigned i = 0; i < n; ++i) { G.mpi_set_ui(pt->m.coeffs[i], 1); G.mpi_neg(pt->m.coeffs[i], pt->m.coeffs[i]); }
    return;
  }
  const unsigned Wout = logql / 64 + 1;
  auto pass = [&](unsigned W, bool kept) -> bool {
    if (W > 32) die("he_dec: coefficients wider than 2047 bits");
    const size_t big = (size_t)W * n;
    HostBuf s0(big * 8), s1(big * 8), s2(big * 8), t0s(big * 8);
    DevBuf d0(big * 8), d1(big * 8), d2(big * 8), dr(big * 8), dz(big * 8), ws(gpq_poly_mul_general_workspace_bytes(c, dim, 1)), scratch(192 * 8);
    const DevBuf *dd[3] = {&d0, &d1, &d2}, *oo[1] = {&dr};
    const HostBuf *ss[3] = {&s0, &s1, &s2}, *ts[1] = {&t0s};
    Operands ops(3, in, dd, ss, n, W);
    ops.prepare(kept);
    const unsigned Wdown = Wout < W ? Wout : W;
    auto device_work = [&]() {
      int rc = pow2 ? gpq_poly_mul(c, dr.u64(), ops.x[0], ops.x[1], W, dim, logql, 1, ws.p, nullptr)                                   // :115
                    : gpq_poly_mul_general(c, dr.u64(), ops.x[0], ops.x[1], W, dim, qw.data(), (unsigned)qw.size(), 1, ws.p, nullptr);
      if (rc == GPQ_OK) rc = gpq_big_addsub(c, dr.u64(), dr.u64(), ops.x[2], W, 1, 0, nullptr);                                       // :117
      if (rc == GPQ_OK && hipMemsetAsync(dz.p, 0, big * 8, nullptr) != hipSuccess) die("device memset failed");                       // the centring kernels take a pair
      if (rc == GPQ_OK)                                                                                                                // :118
        rc = pow2 ? gpq_he_rs(c, dr.u64(), dz.u64(), W, 0, logql, 1, nullptr)
                  : gpq_he_rs_general(c, dr.u64(), dz.u64(), W, 1ull, qw.data(), (unsigned)qw.size(), 1, scratch.p, nullptr);
      if (rc != GPQ_OK) die("he_dec failed");
      download_issue(ts, oo, 1, n, Wdown);
    };
    device_work();
    if (ops.resident && ops.recheck()) {
      if (ops.misfits) { (void)gpq_stream_sync(nullptr); return false; }   // the copies queued into / out of this pass's page-locked buffers must land before the buffers go back to the pool
      device_work();
    }
    std::vector<uint64_t> oprints((size_t)ops.nt, 0);
    download_convert(out, ts, 1, n, Wdown, oprints.data());
    remember_results(out, oo, 1, n, Wdown, oprints);
    return true;
  };
  bool done = false;
  if (pow2 && poly_cache_on(n)) {
    unsigned W = 0;
    bool all = true;
    for (int i = 0; i < 3 && all; ++i) {
      const PolySlot *k = resident_poly(in[i], n, 0);
      if (!k || !k->trusted || (W && k->W != W)) all = false; else W = k->W;
    }
    if (all && W >= Wout) done = pass(W, true);
  }
  if (!done) {
    unsigned bits = nbq;
    for (int i = 0; i < 3; ++i) { const unsigned bi = max_bits(in[i], n); if (bi > bits) bits = bi; }
    pass((bits + 1) / 64 + 1, false);
  }
}

// src/he-mult.c:88-156
void he_mul(he_ct_t *ct, const he_ct_t *ct1, const he_ct_t *ct2, const he_evk_t *rlk) {
  SHIM_CALL();
  need_gcrypt();
  if (&hectx == nullptr || !hectx.q) die("`hectx` is not initialised (hectx_init first)");
  if (ct1->l != ct2->l) die("he_mul: operands at different levels");   // assert at src/he-mult.c:90
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n, l = ct1->l;
  const double nu = ct1->nu * ct2->nu;                                                        // :93
  const double B = ct1->nu * ct2->B + ct2->nu * ct1->B + ct1->B * ct2->B + hectx.bnd.Bmult[l];  // :94-95
  const std::vector<uint64_t> qw = words_of(hectx.q[l], "he_mul: q_l must be positive");
  const bool pow2 = is_pow2(qw);
  const unsigned nbq = G.mpi_get_nbits(hectx.q[l]), logql = nbq - 1, nbPqL = G.mpi_get_nbits(hectx.PqL);
  const unsigned dimA = (nbq * 2 + polyctx.logn) / 59 + 1;                                     // :99
  const unsigned dimB = (nbq + nbPqL + polyctx.logn) / 59 + 1;                                 // :51
  const unsigned dimP = hectx.dim;                                                             // hectx.P, src/precomp.c:401-404
  if (gpq_ctx_pbits(c, dimP) != G.mpi_get_nbits(hectx.P)) die("he_mul: hectx.P is not the product of the first hectx.dim primes");
  const unsigned W = logql / 64 + 1;
  const size_t big = (size_t)W * n;
  const poly_mpi_t *in[4] = {&ct1->c0, &ct1->c1, &ct2->c0, &ct2->c1};
  HostBuf s0(big * 8), s1(big * 8), s2(big * 8), s3(big * 8), t0s(big * 8), t1s(big * 8);
  DevBuf d0(big * 8), d1(big * 8), d2(big * 8), d3(big * 8), o0(big * 8), o1(big * 8),
      ws(pow2 ? gpq_he_mul_workspace_bytes(c, W, dimA, dimB, dimP, 1) : gpq_he_general_workspace_bytes(c, W, dimA, dimB, dimP, 1));
  const DevBuf *dd[4] = {&d0, &d1, &d2, &d3}, *oo[2] = {&o0, &o1};
  const HostBuf *ss[4] = {&s0, &s1, &s2, &s3}, *ts[2] = {&t0s, &t1s};   // results come back into rows of their own: ss may still be filling (recheck)
  // he_mul(&ct, &ct, &ct, rlk), src/he-algo.c:151: one ciphertext on both sides -- convert and upload it once, square on the device
  const bool square = ct1->c0.coeffs == ct2->c0.coeffs && ct1->c1.coeffs == ct2->c1.coeffs;
  const double t0 = wall_ms();
  KeyPrint kp(rlk, dimB, n);
  KeySlot *spec = resident_key(kp);                  // a resident copy: used at once, verified while the device works
  Operands ops(square ? 2 : 4, in, dd, ss, n, W);
  if (spec) ops.prepare(true); else ops.prepare(false, kp.parts, &kp.task);
  uint64_t *k0, *k1;
  if (spec) { k0 = (uint64_t *)spec->d0; k1 = (uint64_t *)spec->d1; } else key_on_device(kp, &k0, &k1);
  if (!g_tick[0]) { (void)hipEventCreate(&g_tick[0]); (void)hipEventCreate(&g_tick[1]); }
  (void)hipEventRecord(g_tick[0], nullptr);
  const double t1 = wall_ms();
  auto device_work = [&]() {
    const uint64_t *const a0 = ops.x[0], *const a1 = ops.x[1], *const e0 = square ? a0 : ops.x[2], *const e1 = square ? a1 : ops.x[3];
    const int rc = pow2 ? gpq_he_mul(c, o0.u64(), o1.u64(), a0, a1, e0, e1, k0, k1, W, logql, dimA, dimB, dimP, 1, ws.p, nullptr)
                        : gpq_he_mul_general(c, o0.u64(), o1.u64(), a0, a1, e0, e1, k0, k1, W, qw.data(),
                                             (unsigned)qw.size(), dimA, dimB, dimP, 1, ws.p, nullptr);
    if (rc != GPQ_OK) die("he_mul failed");
    (void)hipEventRecord(g_tick[1], nullptr);
    download_issue(ts, oo, 2, n, W);
  };
  device_work();
  const double t2 = wall_ms();
  bool again = false;
  if (ops.resident) {                                      // operands and key are checked against the caller's memory while the device works
    again = ops.recheck(kp.parts, &kp.task);
    if (ops.misfits) die("coefficient does not fit the big slab");
    if (!kp.matches(*spec)) { key_on_device(kp, &k0, &k1); again = true; }
  } else if (spec && !key_still_valid(spec, kp)) {         // edited in place since the upload: the reference reads its key on every call
    key_on_device(kp, &k0, &k1);
    again = true;
  }
  if (again) device_work();
  poly_mpi_t *out[2] = {&ct->c0, &ct->c1};
  std::vector<uint64_t> oprints((size_t)2 * ops.nt, 0);
  download_convert(out, ts, 2, n, W, oprints.data());
  remember_results(out, oo, 2, n, W, oprints);
  const double t3 = wall_ms();
  float dev_ms = 0;
  (void)hipEventElapsedTime(&dev_ms, g_tick[0], g_tick[1]);
  g_last_ms[0] = t1 - t0; g_last_ms[1] = dev_ms; g_last_ms[2] = t3 - t2; g_last_ms[3] = t3 - t0;
  ct->l = l; ct->nu = nu; ct->B = B;                                                           // :92-95
}

static void rescale_common(he_ct_t *ct, bool divide) {
  SHIM_CALL();
  need_gcrypt();
  if (&hectx == nullptr || !hectx.q) die("`hectx` is not initialised (hectx_init first)");
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n;
  if (ct->l == 0) die("he_rs / he_moddown: already at level 0");
  const unsigned lnew = ct->l - 1;                                                             // src/he-rescale.c:36, :59
  const std::vector<uint64_t> qw = words_of(hectx.q[lnew], "he_rs: q_l must be positive");
  const std::vector<uint64_t> dw = words_of(hectx.p, "he_rs: Delta must be positive");
  if (dw.size() != 1) die("he_rs: Delta wider than 64 bits");          // hectx_init takes a uint64_t, src/gpqhe.h:100
  const bool pow2 = is_pow2(qw) && (!divide || is_pow2(dw));
  const unsigned logql = G.mpi_get_nbits(hectx.q[lnew]) - 1;
  const unsigned s = divide ? G.mpi_get_nbits(hectx.p) - 1 : 0;
  const poly_mpi_t *in[2] = {&ct->c0, &ct->c1};
  poly_mpi_t *out[2] = {&ct->c0, &ct->c1};
  if (logql == 0) {
    // q_0 = 1 (logq a multiple of logDelta, e.g. 850 = 17 x 50): mpi_smod (src/types.c:108-113) takes r = x mod 1 = 0, finds 0 >= floor(1/2)
    // and subtracts q -- every coefficient of the reference's result is -1, whatever the ciphertext was.  No device work in that.
    for (int k = 0; k < 2; ++k)
      for (unsigned i = 0; i < n; ++i) { G.mpi_set_ui(out[k]->coeffs[i], 1); G.mpi_neg(out[k]->coeffs[i], out[k]->coeffs[i]); }
    ct->l = lnew;
    if (divide) { ct->nu /= hectx.Delta; ct->B = ct->B / hectx.Delta + hectx.bnd.Brs; }
    return;
  }
  const unsigned Wout = logql / 64 + 1;                    // the results are centred mod q_lnew: they fit the width he_mul uses at that level
  // One pass at W words per coefficient; with `kept`, from the resident copies of both polynomials (returns false if one of them
  // turned out wider than its copy: the caller measures the integers and runs again)
  auto pass = [&](unsigned W, bool kept) -> bool {
    const size_t big = (size_t)W * n;
    HostBuf s0(big * 8), s1(big * 8), t0s(big * 8), t1s(big * 8);
    DevBuf d0(big * 8), d1(big * 8), scratch(192 * 8);
    const DevBuf *dd[2] = {&d0, &d1};
    const HostBuf *ss[2] = {&s0, &s1}, *ts[2] = {&t0s, &t1s};
    Operands ops(2, in, dd, ss, n, W);
    ops.prepare(kept);
    const unsigned Wdown = Wout < W ? Wout : W;
    auto device_work = [&]() {
      for (int i = 0; i < 2; ++i)                            // gpq_he_rs works in place: a resident copy is copied, not lent
        if (ops.x[i] != dd[i]->u64() && hipMemcpyAsync(dd[i]->p, ops.x[i], big * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) die("device copy failed");
      const int rc = pow2 ? gpq_he_rs(c, d0.u64(), d1.u64(), W, s, logql, 1, nullptr)                                  // :45-48 / :64-65
                          : gpq_he_rs_general(c, d0.u64(), d1.u64(), W, divide ? dw[0] : 1ull, qw.data(), (unsigned)qw.size(), 1, scratch.p, nullptr);
      if (rc != GPQ_OK) die("he_rs failed");
      download_issue(ts, dd, 2, n, Wdown);                   // the low Wdown words of every coefficient (word-major: the first rows)
    };
    device_work();
    if (ops.resident && ops.recheck()) {
      if (ops.misfits) { (void)gpq_stream_sync(nullptr); return false; }   // the copies queued into / out of this pass's page-locked buffers must land before the buffers go back to the pool
      device_work();
    }
    std::vector<uint64_t> oprints((size_t)2 * ops.nt, 0);
    download_convert(out, ts, 2, n, Wdown, oprints.data());
    remember_results(out, dd, 2, n, Wdown, oprints);
    return true;
  };
  bool done = false;
  if (poly_cache_on(n)) {
    const PolySlot *k0 = resident_poly(in[0], n, 0), *k1 = resident_poly(in[1], n, 0);
    if (k0 && k1 && k0->trusted && k1->trusted && k0->W == k1->W && k0->W >= Wout) done = pass(k0->W, true);
  }
  if (!done) {
    unsigned bits = max_bits(&ct->c0, n), b1 = max_bits(&ct->c1, n);
    if (b1 > bits) bits = b1;
    if (logql + 1 > bits) bits = logql + 1;
    pass(bits / 64 + 1, false);
  }
  ct->l = lnew;
  if (divide) { ct->nu /= hectx.Delta; ct->B = ct->B / hectx.Delta + hectx.bnd.Brs; }             // :37-38
}

// src/he-mult.c:159-196
void he_mulpt(struct he_ct *dest, const struct he_ct *src, const struct he_pt *pt) {
  SHIM_CALL();
  need_gcrypt();
  if (&hectx == nullptr || !hectx.q) die("`hectx` is not initialised (hectx_init first)");
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n, l = src->l;
  const double nu = src->nu * pt->nu, B = src->B * pt->nu;                                      // :163-164
  const std::vector<uint64_t> qw = words_of(hectx.q[l], "he_mulpt: q_l must be positive");
  const bool pow2 = is_pow2(qw);
  const unsigned logql = G.mpi_get_nbits(hectx.q[l]) - 1;
  const unsigned dim = (unsigned)((logql + 1 + log2(pt->nu) + polyctx.logn) / 59u + 1);        // :169, evaluated in double as there
  unsigned bits = max_bits(&pt->m, n);
  if (logql + 1 > bits) bits = logql + 1;
  const unsigned W = bits / 64 + 1;
  const size_t big = (size_t)W * n;
  HostBuf s0(big * 8), s1(big * 8), s2(big * 8), t0s(big * 8), t1s(big * 8);
  DevBuf d0(big * 8), d1(big * 8), dm(big * 8), o0(big * 8), o1(big * 8),
      ws(gpq_he_mulpt_workspace_bytes(c, dim, 1) + gpq_poly_mul_general_workspace_bytes(c, dim, 1));
  const DevBuf *dd[3] = {&d0, &d1, &dm}, *oo[2] = {&o0, &o1};
  const HostBuf *ss[3] = {&s0, &s1, &s2}, *ts[2] = {&t0s, &t1s};
  const poly_mpi_t *in[3] = {&src->c0, &src->c1, &pt->m};
  Operands ops(3, in, dd, ss, n, W);
  ops.prepare(true);
  auto device_work = [&]() {
    const int rc = pow2 ? gpq_he_mulpt(c, o0.u64(), o1.u64(), ops.x[0], ops.x[1], ops.x[2], W, logql, dim, 1, ws.p, nullptr)
                        : gpq_he_mulpt_general(c, o0.u64(), o1.u64(), ops.x[0], ops.x[1], ops.x[2], W, qw.data(), (unsigned)qw.size(), dim, 1, ws.p, nullptr);
    if (rc != GPQ_OK) die("he_mulpt failed");
    download_issue(ts, oo, 2, n, W);
  };
  device_work();
  if (ops.resident && ops.recheck()) {
    if (ops.misfits) die("coefficient does not fit the big slab");
    device_work();
  }
  poly_mpi_t *out[2] = {&dest->c0, &dest->c1};
  std::vector<uint64_t> oprints((size_t)2 * ops.nt, 0);
  download_convert(out, ts, 2, n, W, oprints.data());
  remember_results(out, oo, 2, n, W, oprints);
  dest->l = l; dest->nu = nu; dest->B = B;                                                      // :162-164
}

#include "shim_additive.hpp"
#include "shim_keygen.hpp"

}  // extern "C"

#include "shim_chain.hpp"
