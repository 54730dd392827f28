// ctx_compat.hip -- libgpqhe_hip_ctx.so: context construction and polynomial storage with the reference's names, for hosts
// that are NOT GPQHE (SURVEY.md 8b lists them among libgpqhe.so's exports; tests/c/mpi_host.c is such a host).
//
//   polyctx_init / polyctx_exit  src/precomp.c:328-384, :463-487        hectx_init / hectx_exit  src/precomp.c:386-450, :489-503
//   poly_mpi_alloc/free, poly_rns_alloc/free                            src/poly.c:46-69
//   data symbols polyctx, hectx, GPQHE_TWO                              src/precomp.c:37-47
//
// Kept OUT of libgpqhe_hip.so on purpose: between two shared objects ld.so takes the first definition in search order and
// ignores weak/strong, so definitions of these names inside libgpqhe_hip.so would shadow GPQHE's own precomp.o / poly.o whenever
// the library came first on the link line.  libgpqhe_hip.so only references polyctx / hectx; a GPQHE build links it in any
// order and never links this file; a non-GPQHE host links -lgpqhe_hip -lgpqhe_hip_ctx.  The definitions stay weak so that an
// executable's own (strong) ones still win if it has any.
// kemctx / bootstrapctx belong to the KEM and the bootstrapping skeleton (out of scope) and are not defined.
#include "../../include/gpqhe_hip.h"
#include "../../include/gpqhe_hip_compat.h"
#include "../../include/gpqhe_hip_ctx.h"

#include <cmath>

#include "mpi_convert.hpp"

extern "C" gpq_ctx *gpq_mpi_shim_engine(void);

namespace {

// logqub of the homomorphic-encryption standard for 128-bit classical security (the reference's build: GPQHE_CQ 'C',
// GPQHE_SEC_LEVEL 128, src/params.h:39-46; table src/precomp.c:53-64); 0 outside 10..15
unsigned std_logqub(unsigned logn) {
  static const unsigned tab[6] = {27, 54, 109, 218, 438, 881};
  return (logn >= 10 && logn <= 15) ? tab[logn - 10] : 0;
}

bool g_own_polyctx = false, g_own_hectx = false;

}  // namespace

extern "C" {

__attribute__((weak)) struct poly_ctx polyctx;      // src/precomp.c:41
__attribute__((weak)) struct he_ctx hectx;          // src/precomp.c:47
__attribute__((weak)) gpq_MPI GPQHE_TWO;            // src/precomp.c:37

// src/precomp.c:328-384.  The chain is built by the engine (same primes, constants and tables: tests/test_ntt_gpu.py pins them
// to SURVEY.md 8c) and handed out in the reference's representation; the ring part (src/precomp.c:295-326) is the encoder's.
__attribute__((weak)) void polyctx_init(unsigned int logn, gpq_MPI q) {
  need_gcrypt();
  if (logn < 1 || logn > 17) die("polyctx_init: 1 <= logn <= 17");
  memset(&polyctx, 0, sizeof polyctx);
  GPQHE_TWO = G.mpi_set_ui(G.mpi_new(0), 2);
  polyctx.logn = logn; polyctx.n = 1u << logn; polyctx.m = 2 * polyctx.n;
  polyctx.logq = G.mpi_get_nbits(q) - 1;
  polyctx.logqub = std_logqub(logn);
  if (logn < 10 || logn > 15) polyctx.logqub = polyctx.logq;                     // :339-340
  if (polyctx.logq > polyctx.logqub) {                                           // :343-350
    errno = EINVAL;
    fprintf(stderr, "\033[1m\033[31merror:\033[0m \033[1m%s\033[0m. The input modulus q is too large. Must guarantee log(q)<=log(qub).\n", strerror(errno));
    abort();
  }
  polyctx.q = G.mpi_set(G.mpi_new(0), q);
  polyctx.logR = 64; polyctx.R = (gpq_u128)1 << 64; polyctx.Rsub1 = polyctx.R - 1;
  polyctx.dimub = (1 + logn + 4 * polyctx.logqub) / 59 + 1;                      // :357
  polyctx.rns = (struct rns_ctx *)calloc(polyctx.dimub, sizeof(struct rns_ctx));
  // engine() builds the context for (logn, dimub) -- it needs n and dimub, and compares primes only once rns is set
  struct rns_ctx *nodes = polyctx.rns;
  polyctx.rns = nullptr;
  gpq_ctx *c = gpq_mpi_shim_engine();
  if (gpq_fill_rns_chain(nodes, polyctx.dimub, c, 1) != GPQ_OK) die("polyctx_init: cannot build the prime chain");
  polyctx.rns = nodes;
  // ring_init, src/precomp.c:295-311: the rotation group 5^i mod m and the m-th roots of unity the encoder reads
  const unsigned nh = polyctx.n / 2, m = polyctx.m;
  polyctx.ring.cyc_group = (unsigned int *)malloc((nh ? nh : 1) * sizeof(unsigned int));
  polyctx.ring.cyc_group[0] = 1;
  for (unsigned i = 1; i < nh; ++i) polyctx.ring.cyc_group[i] = (unsigned)((5ull * polyctx.ring.cyc_group[i - 1]) % m);
  double *z = (double *)malloc((size_t)(m + 1) * 2 * sizeof(double));           // _Complex double = (re, im)
  for (unsigned i = 0; i < m; ++i) {
    const double theta = 2 * 3.141592653589793238462643383279502884 * i / m;
    z[2 * i] = cos(theta); z[2 * i + 1] = sin(theta);
  }
  z[2 * m] = z[0]; z[2 * m + 1] = z[1];
  polyctx.ring.zetas = (_Complex double *)z;
  g_own_polyctx = true;
}

__attribute__((weak)) void polyctx_exit(void) {                                  // src/precomp.c:463-487
  if (!g_own_polyctx) return;
  G.mpi_release(GPQHE_TWO);
  G.mpi_release(polyctx.q);
  gpq_release_rns_chain(polyctx.rns);
  free(polyctx.rns);
  free(polyctx.ring.cyc_group);
  free(polyctx.ring.zetas);
  memset(&polyctx, 0, sizeof polyctx);
  g_own_polyctx = false;
  gpq_mpi_shim_release();
}

// src/precomp.c:386-450: qtable_init (q[l] = floor(q[l+1] / Delta), P = first hectx.dim primes, P q_L, dimevk), bounds_init
// (the noise bounds of the CKKS paper, host doubles) and the argument checks.
__attribute__((weak)) void hectx_init(unsigned int logn, gpq_MPI q, unsigned int slots, uint64_t Delta) {
  polyctx_init(logn, q);
  if (slots & (slots - 1)) { errno = EINVAL; fprintf(stderr, "\033[1m\033[31merror:\033[0m \033[1m%s\033[0m. The slots must be the power of 2.\n", strerror(errno)); abort(); }
  if (slots > polyctx.n / 2) { errno = EINVAL; fprintf(stderr, "\033[1m\033[31merror:\033[0m \033[1m%s\033[0m. Must guarantee slots<=(n/2).\n", strerror(errno)); abort(); }
  if (Delta < 2) die("hectx_init: Delta must be at least 2");
  memset(&hectx, 0, sizeof hectx);
  hectx.slots = slots;
  hectx.Delta = (double)Delta;
  hectx.p = G.mpi_set_ui(G.mpi_new(0), Delta);
  const unsigned logq = polyctx.logq, logDelta = 63 - (unsigned)__builtin_clzll(Delta);
  hectx.L = logq / logDelta;                                                     // "ceil" of an integer quotient, :391
  hectx.q = (gpq_MPI *)G.xmalloc((hectx.L + 1) * sizeof(gpq_MPI));
  hectx.qh = (gpq_MPI *)G.xmalloc((hectx.L + 1) * sizeof(gpq_MPI));
  Words cur = words_of(q, "hectx_init: q must be positive");
  const Words qL = cur;
  Words q0;
  for (int l = (int)hectx.L; l >= 0; --l) {                                      // :394-400
    hectx.q[l] = mpi_of(cur);
    Words h = cur;
    shr1(h);
    hectx.qh[l] = mpi_of(h);
    if (l == 0) q0 = cur;
    (void)divmod_word(cur, Delta);
  }
  hectx.dim = (bits_of(qL) + logn) / 59 + 1;                                     // :401
  if (hectx.dim > polyctx.dimub) die("hectx_init: the chain is shorter than hectx.dim");
  Words P(1, 1);
  const struct rns_ctx *r = polyctx.rns;
  for (unsigned d = 0; d < hectx.dim; ++d, r = r->next) mul_word(P, r->p);
  hectx.P = mpi_of(P);                                                           // :402-404
  const Words PqL = mul_words(P, qL);
  hectx.PqL = mpi_of(PqL);                                                       // :405-406
  hectx.dimevk = (bits_of(qL) + bits_of(PqL) + logn) / 59 + 1;                   // :407
  // bounds_init, :411-432
  const double n = polyctx.n, h = 64 /* GPQHE_BLKSIZ */, sigma = 3.1915382432114616 /* GPQHE_SIGMA */;
  hectx.bnd.Bclean = 8 * sqrt(2) * sigma * n + 6 * sigma * sqrt(n) + 16 * sigma * sqrt(h * n);
  hectx.bnd.Brs = sqrt(n / 3.) * (3 + 8 * sqrt(h));
  hectx.bnd.Bks = 8 * sigma * n / sqrt(3);
  hectx.bnd.Bmult = (double *)malloc((hectx.L + 1) * sizeof(double));
  long double Pinv = 1;
  for (r = polyctx.rns; r; r = r->next) Pinv *= 1. / r->p;
  long double Pinvql = Pinv * (q0.empty() ? 0 : q0[0]);                          // mpi_to_u64(q[0])
  hectx.bnd.Bmult[0] = (double)(Pinvql * hectx.bnd.Bks + hectx.bnd.Brs);
  for (unsigned l = 1; l <= hectx.L; ++l) {
    Pinvql *= hectx.Delta;
    hectx.bnd.Bmult[l] = (double)(Pinvql * hectx.bnd.Bks + hectx.bnd.Brs);
  }
  if (!((double)Delta > polyctx.n + 2 * hectx.bnd.Bclean)) die("hectx_init: Delta <= n + 2 Bclean (assert at src/precomp.c:449)");
  g_own_hectx = true;
}

__attribute__((weak)) void hectx_exit(void) {                                    // src/precomp.c:489-503
  if (g_own_hectx) {
    for (unsigned l = 0; l <= hectx.L; ++l) { G.mpi_release(hectx.q[l]); G.mpi_release(hectx.qh[l]); }
    G.xfree(hectx.q); G.xfree(hectx.qh);
    G.mpi_release(hectx.p); G.mpi_release(hectx.P); G.mpi_release(hectx.PqL);
    free(hectx.bnd.Bmult);
    memset(&hectx, 0, sizeof hectx);
    g_own_hectx = false;
  }
  polyctx_exit();
}

// src/poly.c:46-69.  poly_rns_alloc: the reference clears only the first 8 bytes; contents are indeterminate
// until written either way, so this one clears nothing.
__attribute__((weak)) void poly_mpi_alloc(poly_mpi_t *a) {
  need_gcrypt();
  a->coeffs = (gpq_MPI *)G.xmalloc((size_t)polyctx.n * sizeof(gpq_MPI));
  for (unsigned i = 0; i < polyctx.n; ++i) a->coeffs[i] = G.mpi_new(0);
}
__attribute__((weak)) void poly_mpi_free(poly_mpi_t *a) {
  need_gcrypt();
  for (unsigned i = 0; i < polyctx.n; ++i) G.mpi_release(a->coeffs[i]);
  G.xfree(a->coeffs);
  a->coeffs = nullptr;
}
__attribute__((weak)) void poly_rns_alloc(poly_rns_t *a, const unsigned int dim) {
  a->coeffs = (uint64_t *)malloc((size_t)dim * polyctx.n * sizeof(uint64_t));
}
__attribute__((weak)) void poly_rns_free(poly_rns_t *a) {
  free(a->coeffs);
  a->coeffs = nullptr;
}

}  // extern "C"
