/*
 * mpi_host.c -- a C host with REAL libgcrypt MPIs driving the reference-named, MPI-typed
 * entry points of libgpqhe_hip.so (poly_mul, he_mul, he_rs, he_rescale, he_moddown), the way
 * GPQHE's own tests/polymul.c and tests/gpqhe.c drive the reference's.
 *
 *   mpi_host polymul                     the two products of tests/polymul.c:59-76 (n = 128, 5 limbs, q = 2^61)
 *   mpi_host crt                         rns_decompose per limb, rns_reconstruct per coefficient, poly_rns2mpi (tests/crt.c:76-109)
 *   mpi_host polymulmono <logn>          poly_mul of a dense polynomial by -3 x^5 at a size that takes the threaded conversions
 *   mpi_host keygen <logn> <logq>        he_genrlk / he_genck / he_genrk with deterministic stand-ins for the reference's samplers
 *   mpi_host residentfuzz <logn> <logq> <logDelta> <steps> <seed>   random walk over the calls, resident polynomials vs fresh uploads
 *   mpi_host hemultime <logn> <logq> [threads]   wall time of he_mul / he_rescale through the MPI-typed symbols (conversions and copies included)
 *   mpi_host hemul  <in.txt>             he_mul on ciphertexts read as hex, then he_rs, then he_moddown
 *   mpi_host ctxcheck <logn> <logq> <Delta>   every field hectx_init / polyctx_init fill, for comparison with the restated formulas
 *
 * `polyctx`, `hectx`, polyctx_init, hectx_init and the poly_*_alloc functions are the library's (weak) definitions of the
 * reference's symbols (src/poly.h:80-83,94-95, src/gpqhe.h:100-101), so this host restates nothing of src/precomp.c.
 * libgcrypt's public functions are declared here by hand: the image has the runtime library only.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpqhe_hip.h"
#include "gpqhe_hip_compat.h"
#include "gpqhe_hip_ctx.h"     /* polyctx_init / hectx_init / poly_*_alloc: this host is not GPQHE, it links -lgpqhe_hip_ctx for them */

typedef void *MPI;
MPI gcry_mpi_new(unsigned int nbits);
void gcry_mpi_release(MPI a);
MPI gcry_mpi_set_ui(MPI w, unsigned long u);
MPI gcry_mpi_copy(const MPI a);
void gcry_mpi_mul(MPI w, MPI u, MPI v);
void gcry_mpi_add(MPI w, MPI u, MPI v);
void gcry_mpi_add_ui(MPI w, MPI u, unsigned long v);
void gcry_mpi_addm(MPI w, MPI u, MPI v, MPI m);
void gcry_mpi_sub(MPI w, MPI u, MPI v);
MPI gcry_mpi_set(MPI w, const MPI u);
void gcry_mpi_div(MPI q, MPI r, MPI dividend, MPI divisor, int round);
void gcry_mpi_mul_ui(MPI w, MPI u, unsigned long v);
void gcry_mpi_lshift(MPI x, MPI a, unsigned int n);
void gcry_mpi_rshift(MPI x, MPI a, unsigned int n);
void gcry_mpi_neg(MPI w, MPI u);
int gcry_mpi_is_neg(MPI a);
unsigned int gcry_mpi_get_nbits(MPI a);
unsigned int gcry_mpi_scan(MPI *ret, int format, const void *buffer, size_t buflen, size_t *nscanned);
unsigned int gcry_mpi_aprint(int format, unsigned char **buffer, size_t *nwritten, const MPI a);
void gcry_free(void *p);
void gcry_mpi_mod(MPI r, MPI dividend, MPI divisor);
int gcry_mpi_cmp(const MPI u, const MPI v);
#define FMT_HEX 4

/* `polyctx`, `hectx`, polyctx_init, hectx_init, poly_mpi_alloc come from libgpqhe_hip.so (weak definitions, gpqhe_hip_compat.h):
 * this host is not GPQHE, so nothing here restates src/precomp.c. */

static void print_mpi(MPI a)
{
  unsigned char *s = NULL;
  gcry_mpi_aprint(FMT_HEX, &s, NULL, a);
  printf("%s\n", (char *)s);
  gcry_free(s);
}

static void poly_alloc(poly_mpi_t *a) { poly_mpi_alloc(a); }   /* src/poly.c:46-51 */

static MPI pow2(unsigned logq)
{
  MPI q = gcry_mpi_new(0);
  gcry_mpi_set_ui(q, 1);
  gcry_mpi_lshift(q, q, logq);
  return q;
}

static void ctx_init(unsigned logn, unsigned logq)         /* polyctx_init(logn, 2^logq), src/precomp.c:328-384 */
{
  MPI q = pow2(logq);
  polyctx_init(logn, q);
  gcry_mpi_release(q);
  /* dimub follows logqub: the security table for 10 <= logn <= 15, logq otherwise (src/precomp.c:338-340, :357) */
  if (polyctx.dimub != gpq_dimub(logn, polyctx.logqub) || !polyctx.rns || polyctx.rns->dim != 1 || !polyctx.rns->P || !polyctx.ring.cyc_group) {
    fprintf(stderr, "polyctx_init left an incomplete context\n");
    exit(1);
  }
}

static int polymul(int odd_modulus)
{
  ctx_init(7, 61);                                          /* tests/polymul.c:84-93 */
  MPI q = gcry_mpi_new(0);
  gcry_mpi_set_ui(q, odd_modulus ? 3 : 1);
  gcry_mpi_lshift(q, q, 61);
  if (odd_modulus) { MPI seven = gcry_mpi_new(0); gcry_mpi_set_ui(seven, 7); gcry_mpi_add(q, q, seven); }   /* q = 3*2^61 + 7 */
  poly_mpi_t a, b, r;
  poly_alloc(&a); poly_alloc(&b); poly_alloc(&r);
  for (int t = 0; t < 2; t++) {
    for (unsigned i = 0; i < polyctx.n; i++) {
      if (t == 0) { gcry_mpi_set_ui(a.coeffs[i], i + 2); gcry_mpi_set_ui(b.coeffs[i], i + 3); }             /* :60-63 */
      else { gcry_mpi_set_ui(a.coeffs[i], polyctx.rns->p - i - 1); gcry_mpi_set_ui(b.coeffs[i], polyctx.rns->next->p - i - 1); } /* :69-74 */
    }
    poly_mul(&r, &a, &b, polyctx.dimub, q);                 /* :64, :75 */
    for (unsigned i = 0; i < polyctx.n; i++) print_mpi(r.coeffs[i]);
  }
  return 0;
}

static int crt(void)
{
  ctx_init(7, 61);
  const unsigned dim = polyctx.dimub;                       /* 5 limbs */
  struct rns_ctx *rns = polyctx.rns;
  MPI P3 = gcry_mpi_new(0);
  gcry_mpi_set_ui(P3, 1);
  for (unsigned d = 0; d < 3; d++, rns = rns->next) gcry_mpi_mul_ui(P3, P3, rns->p);
  poly_mpi_t a, r;
  poly_alloc(&a); poly_alloc(&r);
  MPI t = gcry_mpi_new(0);
  for (unsigned i = 0; i < polyctx.n; i++) {                /* a[i] = P[2] - i - 1 (tests/crt.c:79-81), every odd one negated */
    gcry_mpi_set_ui(t, i + 1);
    gcry_mpi_neg(t, t);
    gcry_mpi_add(a.coeffs[i], P3, t);
    if (i & 1) gcry_mpi_neg(a.coeffs[i], a.coeffs[i]);
  }
  poly_rns_t ahat;
  ahat.coeffs = malloc((size_t)dim * polyctx.n * 8);
  rns = polyctx.rns;
  for (unsigned d = 0; d < dim; d++, rns = rns->next) rns_decompose(&ahat.coeffs[d * polyctx.n], a.coeffs, rns);   /* :96-99 */
  for (unsigned d = 0; d < dim; d++)
    for (unsigned i = 0; i < polyctx.n; i++) printf("%llu\n", (unsigned long long)ahat.coeffs[d * polyctx.n + i]);
  for (rns = polyctx.rns; rns->dim < dim; rns = rns->next);
  for (unsigned i = 0; i < polyctx.n; i++) { rns_reconstruct(r.coeffs[i], ahat.coeffs, i, rns); print_mpi(r.coeffs[i]); }  /* :107-109 */
  MPI q = gcry_mpi_new(0);
  gcry_mpi_set_ui(q, 1);
  gcry_mpi_lshift(q, q, 61);
  poly_rns2mpi(&r, &ahat, rns, q);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(r.coeffs[i]);
  gcry_mpi_set_ui(q, 1000003);                              /* an odd modulus takes the general second centring */
  gcry_mpi_mul_ui(q, q, 1000003);
  gcry_mpi_mul_ui(q, q, 1000003);
  poly_rns2mpi(&r, &ahat, rns, q);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(r.coeffs[i]);
  return 0;
}

static uint64_t splitmix64(uint64_t *s)
{
  uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

static void read_poly(FILE *f, poly_mpi_t *a)
{
  char *line = malloc(4096);
  for (unsigned i = 0; i < polyctx.n; i++) {
    if (!fgets(line, 4096, f)) exit(3);
    line[strcspn(line, "\n")] = 0;
    MPI t = NULL;
    const int neg = line[0] == '-';
    gcry_mpi_scan(&t, FMT_HEX, line + neg, 0, NULL);
    if (neg) gcry_mpi_neg(t, t);
    gcry_mpi_release(a->coeffs[i]);
    a->coeffs[i] = t;
  }
  free(line);
}

static void he_ctx_init(unsigned logn, MPI q, unsigned long long Delta)   /* hectx_init, src/precomp.c:434-450 */
{
  hectx_init(logn, q, 2, Delta);
  /* the noise bookkeeping of he_mul / he_rs reads these two; fixed values make the expected nu / B of the tests exact */
  hectx.bnd.Brs = 11.5;
  for (unsigned l = 0; l <= hectx.L; l++) hectx.bnd.Bmult[l] = 100.0 + l;
}

static int hemul(const char *path)
{
  FILE *f = fopen(path, "r");
  unsigned logn, level;
  unsigned long long Delta;
  char qhex[2048];
  if (!f || fscanf(f, "%u %2047s %llu %u\n", &logn, qhex, &Delta, &level) != 4) return 3;
  MPI q = NULL;
  gcry_mpi_scan(&q, FMT_HEX, qhex, 0, NULL);
  he_ctx_init(logn, q, Delta);
  printf("dims %u %u %u\n", hectx.dim, hectx.dimevk, hectx.L);

  he_ct_t ct1, ct2, ct;
  poly_alloc(&ct1.c0); poly_alloc(&ct1.c1); poly_alloc(&ct2.c0); poly_alloc(&ct2.c1); poly_alloc(&ct.c0); poly_alloc(&ct.c1);
  read_poly(f, &ct1.c0); read_poly(f, &ct1.c1); read_poly(f, &ct2.c0); read_poly(f, &ct2.c1);
  ct1.l = ct2.l = level; ct1.nu = 3.0; ct1.B = 5.0; ct2.nu = 7.0; ct2.B = 11.0;
  he_evk_t rlk;                                              /* he_alloc_evk, src/he-mem.c:42-46; synthetic NTT-domain key */
  rlk.p0.coeffs = malloc((size_t)hectx.dimevk * polyctx.n * 8); rlk.p1.coeffs = malloc((size_t)hectx.dimevk * polyctx.n * 8);
  uint64_t s0 = 3000, s1 = 3001;
  struct rns_ctx *r = polyctx.rns;
  for (unsigned d = 0; d < hectx.dimevk; d++, r = r->next)
    for (unsigned i = 0; i < polyctx.n; i++) {
      rlk.p0.coeffs[(size_t)d * polyctx.n + i] = splitmix64(&s0) % r->p;
      rlk.p1.coeffs[(size_t)d * polyctx.n + i] = splitmix64(&s1) % r->p;
    }
  he_mul(&ct, &ct1, &ct2, &rlk);                             /* tests/gpqhe.c:458 */
  printf("he_mul %u %.17g %.17g\n", ct.l, ct.nu, ct.B);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c1.coeffs[i]);
  he_rescale(&ct);                                           /* he_rs, tests/gpqhe.c:480 */
  printf("he_rs %u %.17g %.17g\n", ct.l, ct.nu, ct.B);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c1.coeffs[i]);
  he_moddown(&ct);                                           /* tests/gpqhe.c:498 */
  printf("he_moddown %u\n", ct.l);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c1.coeffs[i]);
  he_mul(&ct, &ct, &ct, &rlk);                               /* aliased operands, as src/he-algo.c:151 */
  printf("he_sq %u\n", ct.l);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c1.coeffs[i]);

  /* rotation / conjugation keys: rk[rot] and ck, synthetic NTT-domain slabs like rlk */
  he_evk_t rk[2], ck;
  he_evk_t *keys[2] = {&rk[1], &ck};
  uint64_t seeds[2][2] = {{5002, 5003}, {6000, 6001}};
  rk[0].p0.coeffs = rk[0].p1.coeffs = NULL;
  for (int t = 0; t < 2; t++) {
    keys[t]->p0.coeffs = malloc((size_t)hectx.dimevk * polyctx.n * 8); keys[t]->p1.coeffs = malloc((size_t)hectx.dimevk * polyctx.n * 8);
    r = polyctx.rns;
    for (unsigned d = 0; d < hectx.dimevk; d++, r = r->next)
      for (unsigned i = 0; i < polyctx.n; i++) {
        keys[t]->p0.coeffs[(size_t)d * polyctx.n + i] = splitmix64(&seeds[t][0]) % r->p;
        keys[t]->p1.coeffs[(size_t)d * polyctx.n + i] = splitmix64(&seeds[t][1]) % r->p;
      }
  }
  he_rot(&ct, 1, rk);                                        /* tests/gpqhe.c rot; rk[rot] */
  printf("he_rot %u\n", ct.l);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c1.coeffs[i]);
  he_conj(&ct, &ck);
  printf("he_conj %u\n", ct.l);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(ct.c1.coeffs[i]);
  he_pt_t pt;
  poly_alloc(&pt.m);
  pt.nu = 1024.0;
  for (unsigned i = 0; i < polyctx.n; i++) {                 /* a plaintext polynomial with signed ~2^30 coefficients */
    gcry_mpi_set_ui(pt.m.coeffs[i], ((unsigned long)(i + 1) << 20) + 12345u * i);
    if (i & 1) gcry_mpi_neg(pt.m.coeffs[i], pt.m.coeffs[i]);
  }
  he_ct_t prod;
  poly_alloc(&prod.c0); poly_alloc(&prod.c1);
  ct.nu = 2.0; ct.B = 3.0;
  he_mulpt(&prod, &ct, &pt);                                 /* tests/gpqhe.c:516 */
  printf("he_mulpt %u %.17g %.17g\n", prod.l, prod.nu, prod.B);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(prod.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(prod.c1.coeffs[i]);
  /* src/he-add.c: he_add, he_sub, he_addpt, he_subpt, he_neg on the same chain (he_inv's pattern, src/he-algo.c:146-155) */
  he_ct_t sum;
  poly_alloc(&sum.c0); poly_alloc(&sum.c1);
  ct.l = prod.l;
  he_add(&sum, &prod, &ct);
  printf("he_add %u %.17g %.17g\n", sum.l, sum.nu, sum.B);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c1.coeffs[i]);
  he_sub(&sum, &sum, &prod);                                 /* output aliases the first operand */
  printf("he_sub %u\n", sum.l);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c1.coeffs[i]);
  he_addpt(&sum, &prod, &pt);
  printf("he_addpt %u %.17g %.17g\n", sum.l, sum.nu, sum.B);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c1.coeffs[i]);
  he_subpt(&sum, &sum, &pt);
  printf("he_subpt %u\n", sum.l);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c1.coeffs[i]);
  he_neg(&sum);
  printf("he_neg %u\n", sum.l);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c0.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(sum.c1.coeffs[i]);
  he_ct_t cp;
  poly_alloc(&cp.c0); poly_alloc(&cp.c1);
  sum.nu = 6.5; sum.B = 7.25;
  he_copy_ct(&cp, &sum);                                     /* src/he-mem.c:88-97 */
  int same = cp.l == sum.l && cp.nu == 6.5 && cp.B == 7.25;
  for (unsigned i = 0; i < polyctx.n; i++) if (gcry_mpi_cmp(cp.c0.coeffs[i], sum.c0.coeffs[i]) || gcry_mpi_cmp(cp.c1.coeffs[i], sum.c1.coeffs[i])) same = 0;
  printf("he_copy_ct %s\n", same ? "identical" : "DIFFERS");
  poly_mpi_t sk;                                             /* a sparse secret key: 1 - x^5 + x^(n-1) */
  poly_alloc(&sk);
  gcry_mpi_set_ui(sk.coeffs[0], 1); gcry_mpi_set_ui(sk.coeffs[5], 1); gcry_mpi_neg(sk.coeffs[5], sk.coeffs[5]); gcry_mpi_set_ui(sk.coeffs[polyctx.n - 1], 1);
  he_pt_t dec;
  poly_alloc(&dec.m);
  he_dec(&dec, &cp, &sk);                                    /* src/he-encrypt.c:105-125 */
  printf("he_dec %.17g\n", dec.nu);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(dec.m.coeffs[i]);
  he_dec(&dec, &cp, &sk);                                    /* again: the ciphertext and the key are resident now */
  printf("he_dec again\n");
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(dec.m.coeffs[i]);
  uint64_t confirmed = 0, changed = 0;
  gpq_mpi_shim_poly_stats(&confirmed, &changed);             /* operands the chain took from the device copies of earlier results */
  printf("resident %llu %llu\n", (unsigned long long)confirmed, (unsigned long long)changed);
  return 0;
}

/* poly_mul at a size where the shim converts coefficients on several host threads: dense a (seeded, signed, ~100 bits) times the
 * monomial -3 x^5, so that the expected product is a signed shift of a (cheap to restate exactly) */
static int polymulmono(unsigned logn)
{
  ctx_init(logn, 109);
  MPI q = gcry_mpi_new(0);
  gcry_mpi_set_ui(q, 1);
  gcry_mpi_lshift(q, q, 109);
  poly_mpi_t a, b, r;
  poly_alloc(&a); poly_alloc(&b); poly_alloc(&r);
  uint64_t st = 4096;
  for (unsigned i = 0; i < polyctx.n; i++) {
    unsigned char buf[12];
    const uint64_t v0 = splitmix64(&st), v1 = splitmix64(&st);
    memcpy(buf, &v0, 8); memcpy(buf + 8, &v1, 4);
    MPI t = NULL;
    gcry_mpi_scan(&t, 5, buf, 12, NULL);                      /* 96-bit magnitude, big-endian bytes of the two words */
    if (v1 >> 63) gcry_mpi_neg(t, t);
    gcry_mpi_release(a.coeffs[i]); a.coeffs[i] = t;
    gcry_mpi_set_ui(b.coeffs[i], i == 5 ? 3 : 0);
  }
  gcry_mpi_neg(b.coeffs[5], b.coeffs[5]);
  poly_mul(&r, &a, &b, polyctx.dimub, q);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(a.coeffs[i]);
  for (unsigned i = 0; i < polyctx.n; i++) print_mpi(r.coeffs[i]);
  return 0;
}

/* The reference's samplers (src/sample.c) as the key generation of libgpqhe_hip.so finds them in the host program: deterministic
 * stand-ins here so that the test can restate the keys.  error: splitmix % 17 - 8; uniform: nbytes(q) + 8 splitmix bytes mod q. */
static uint64_t err_state = 111, uni_state = 222;
void sample_error(poly_mpi_t *r)
{
  for (unsigned i = 0; i < polyctx.n; i++) {
    const long v = (long)(splitmix64(&err_state) % 17) - 8;
    gcry_mpi_set_ui(r->coeffs[i], (unsigned long)(v < 0 ? -v : v));
    if (v < 0) gcry_mpi_neg(r->coeffs[i], r->coeffs[i]);
  }
}
void sample_uniform(poly_mpi_t *r, const MPI q)
{
  const unsigned nb = (gcry_mpi_get_nbits(q) + 7) / 8 + 8;
  unsigned char *buf = malloc(nb + 8);
  for (unsigned i = 0; i < polyctx.n; i++) {
    for (unsigned b = 0; b < nb; b += 8) { const uint64_t v = splitmix64(&uni_state); memcpy(buf + b, &v, 8); }
    MPI t = NULL;
    gcry_mpi_scan(&t, 5, buf, nb, NULL);
    gcry_mpi_mod(r->coeffs[i], t, q);
    gcry_mpi_release(t);
  }
  free(buf);
}

static uint64_t fnv1a(const uint64_t *p, size_t words)
{
  uint64_t h = 0xcbf29ce484222325ull;
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < words * 8; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
  return h;
}

/* he_genrlk, he_genck, he_genrk through the reference's signatures (src/gpqhe.h:131-133) with the samplers above */
static int keygen(unsigned logn, unsigned logq)
{
  MPI q = gcry_mpi_new(0);
  gcry_mpi_set_ui(q, 1);
  gcry_mpi_lshift(q, q, logq);
  he_ctx_init(logn, q, 1ull << 30);
  poly_mpi_t sk;
  poly_alloc(&sk);
  uint64_t st = 333;
  for (unsigned i = 0; i < polyctx.n; i++) {                 /* ternary secret */
    const unsigned v = (unsigned)(splitmix64(&st) % 3);
    gcry_mpi_set_ui(sk.coeffs[i], v == 2 ? 1 : v);
    if (v == 2) gcry_mpi_neg(sk.coeffs[i], sk.coeffs[i]);
  }
  he_evk_t keys[4];                                          /* rlk, ck, rk[0], rk[1] */
  const size_t words = (size_t)hectx.dimevk * polyctx.n;
  for (int k = 0; k < 4; k++) { keys[k].p0.coeffs = malloc(words * 8); keys[k].p1.coeffs = malloc(words * 8); }
  he_genrlk(&keys[0], &sk);
  he_genck(&keys[1], &sk);
  he_genrk(&keys[2], &sk);
  printf("dims %u %u\n", hectx.dim, hectx.dimevk);
  for (int k = 0; k < 4; k++)
    printf("key %d %016llx %016llx\n", k, (unsigned long long)fnv1a(keys[k].p0.coeffs, words), (unsigned long long)fnv1a(keys[k].p1.coeffs, words));
  return 0;
}

#include <time.h>
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

static int cmp_double(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }

/* wall time of the MPI-typed he_mul (host conversions and PCIe copies included) on random centred ciphertexts */
static int hemultime(unsigned logn, unsigned logq)
{
  MPI q = gcry_mpi_new(0);
  gcry_mpi_set_ui(q, 1);
  gcry_mpi_lshift(q, q, logq);
  he_ctx_init(logn, q, 1ull << 50);
  he_ct_t ct1, ct2, ct;
  poly_mpi_t *ps[6] = {&ct1.c0, &ct1.c1, &ct2.c0, &ct2.c1, &ct.c0, &ct.c1};
  for (int i = 0; i < 6; i++) poly_alloc(ps[i]);
  uint64_t st = 99;
  unsigned char *buf = malloc(logq / 8 + 8);
  for (int i = 0; i < 4; i++)
    for (unsigned k = 0; k < polyctx.n; k++) {
      const unsigned nb = (logq - 2) / 8;                     /* |coefficient| < 2^(logq-2) < q/2 */
      for (unsigned b = 0; b < nb; b += 8) { uint64_t v = splitmix64(&st); memcpy(buf + b, &v, 8); }
      MPI t = NULL;
      gcry_mpi_scan(&t, 5, buf, nb, NULL);
      if (splitmix64(&st) & 1) gcry_mpi_neg(t, t);
      gcry_mpi_release(ps[i]->coeffs[k]);
      ps[i]->coeffs[k] = t;
    }
  ct1.l = ct2.l = hectx.L; ct1.nu = ct2.nu = 1.0; ct1.B = ct2.B = 1.0;
  he_evk_t rlk;
  rlk.p0.coeffs = malloc((size_t)hectx.dimevk * polyctx.n * 8); rlk.p1.coeffs = malloc((size_t)hectx.dimevk * polyctx.n * 8);
  struct rns_ctx *r = polyctx.rns;
  for (unsigned d = 0; d < hectx.dimevk; d++, r = r->next)
    for (unsigned i = 0; i < polyctx.n; i++) {
      rlk.p0.coeffs[(size_t)d * polyctx.n + i] = splitmix64(&st) % r->p;
      rlk.p1.coeffs[(size_t)d * polyctx.n + i] = splitmix64(&st) % r->p;
    }
  he_mul(&ct, &ct1, &ct2, &rlk);                              /* first call builds the device tables */
  /* the key stays on the device between calls; a key rewritten in place must give the new key's result (fingerprint), and the
   * same after an explicit gpq_mpi_shim_forget_keys() */
  {
    he_ct_t a, b;
    poly_alloc(&a.c0); poly_alloc(&a.c1); poly_alloc(&b.c0); poly_alloc(&b.c1);
    r = polyctx.rns;
    for (unsigned d = 0; d < hectx.dimevk; d++, r = r->next)                                  /* (nearly) every word changes, residues stay canonical */
      for (unsigned i = 0; i < polyctx.n; i++) { uint64_t *w = &rlk.p0.coeffs[(size_t)d * polyctx.n + i]; if ((*w ^ 1) < r->p) *w ^= 1; }
    he_mul(&a, &ct1, &ct2, &rlk);
    gpq_mpi_shim_forget_keys();
    he_mul(&b, &ct1, &ct2, &rlk);
    int same_as_old = 1, consistent = 1;
    for (unsigned i = 0; i < polyctx.n; i++) {
      if (gcry_mpi_cmp(a.c0.coeffs[i], ct.c0.coeffs[i])) same_as_old = 0;
      if (gcry_mpi_cmp(a.c0.coeffs[i], b.c0.coeffs[i]) || gcry_mpi_cmp(a.c1.coeffs[i], b.c1.coeffs[i])) consistent = 0;
    }
    printf("key cache: rewritten key %s, cached vs fresh upload %s\n", same_as_old ? "IGNORED" : "seen", consistent ? "identical" : "DIFFER");
    /* ONE word changed, at an index no sampling scheme of ~1000 words looks at: the product must be the edited key's (the
     * reference reads its key on every call), i.e. differ from the product before the edit and equal a fresh upload's */
    const size_t words = (size_t)hectx.dimevk * polyctx.n, step = words / 509, at = 3 * step + step / 2 + 1;
    rlk.p1.coeffs[at] = rlk.p1.coeffs[at] > 5 ? rlk.p1.coeffs[at] - 5 : rlk.p1.coeffs[at] + 5;
    he_ct_t e;
    poly_alloc(&e.c0); poly_alloc(&e.c1);
    he_mul(&e, &ct1, &ct2, &rlk);
    gpq_mpi_shim_forget_keys();
    he_mul(&b, &ct1, &ct2, &rlk);
    int seen = 0, fresh = 1;
    for (unsigned i = 0; i < polyctx.n; i++) {
      if (gcry_mpi_cmp(e.c0.coeffs[i], a.c0.coeffs[i]) || gcry_mpi_cmp(e.c1.coeffs[i], a.c1.coeffs[i])) seen = 1;
      if (gcry_mpi_cmp(e.c0.coeffs[i], b.c0.coeffs[i]) || gcry_mpi_cmp(e.c1.coeffs[i], b.c1.coeffs[i])) fresh = 0;
    }
    printf("key cache: one unsampled word edited in place %s, cached vs fresh upload %s\n", seen ? "seen" : "IGNORED", fresh ? "identical" : "DIFFER");
    /* lowering the number of resident keys evicts at once */
    he_mul(&e, &ct1, &ct2, &rlk);
    const unsigned before = gpq_mpi_shim_resident_keys();
    gpq_mpi_shim_set_key_slots(1);
    const unsigned one = gpq_mpi_shim_resident_keys();
    gpq_mpi_shim_set_key_slots(64);
    printf("key cache: resident %u, after set_key_slots(1) %u\n", before, one);
  }
  {
    /* the conversions read / write libgcrypt's limbs in place (after the layout probe); through gcry_mpi_print / gcry_mpi_scan the
     * product must be the same integers */
    he_ct_t x, y;
    poly_alloc(&x.c0); poly_alloc(&x.c1); poly_alloc(&y.c0); poly_alloc(&y.c1);
    const int direct = gpq_mpi_shim_set_direct_mpi(1);
    he_mul(&x, &ct1, &ct2, &rlk);
    (void)gpq_mpi_shim_set_direct_mpi(0);
    const double t0 = now_ms();
    he_mul(&y, &ct1, &ct2, &rlk);
    const double slow = now_ms() - t0;
    (void)gpq_mpi_shim_set_direct_mpi(1);
    int same = 1;
    for (unsigned i = 0; i < polyctx.n; i++) if (gcry_mpi_cmp(x.c0.coeffs[i], y.c0.coeffs[i]) || gcry_mpi_cmp(x.c1.coeffs[i], y.c1.coeffs[i])) same = 0;
    printf("direct mpi access: %s, print/scan path %s (%.2f ms per call that way)\n", direct ? "in use" : "NOT in use", same ? "identical" : "DIFFERS", slow);
  }
  {
    /* resident polynomials: the device keeps the slabs of polynomials it has seen or produced and starts from them while the host
     * threads check them against the caller's integers -- whatever the caller does to its integers between two calls, the result
     * must be what a library without that memory computes (gpq_mpi_shim_set_poly_slots(0)) */
    he_ct_t u, v, w;
    poly_mpi_t *qs[6] = {&u.c0, &u.c1, &v.c0, &v.c1, &w.c0, &w.c1};
    for (int i = 0; i < 6; i++) poly_alloc(qs[i]);
    uint64_t ok0 = 0, stale0 = 0, ok1 = 0, stale1 = 0;
    int bad = 0;
#define SAME_CT(x, y) ({ int same_ = 1; for (unsigned i_ = 0; i_ < polyctx.n; i_++) if (gcry_mpi_cmp((x).c0.coeffs[i_], (y).c0.coeffs[i_]) || gcry_mpi_cmp((x).c1.coeffs[i_], (y).c1.coeffs[i_])) same_ = 0; same_; })
    he_mul(&u, &ct1, &ct2, &rlk);                             /* operands resident (the calls above), unchanged */
    gpq_mpi_shim_poly_stats(&ok0, &stale0);
    gpq_mpi_shim_set_poly_slots(0);
    he_mul(&v, &ct1, &ct2, &rlk);
    gpq_mpi_shim_set_poly_slots(32);
    if (!SAME_CT(u, v)) { bad = 1; printf("resident polynomials: unchanged operands DIFFER from a fresh upload\n"); }
    he_mul(&u, &ct1, &ct2, &rlk);                             /* slots were emptied: uploads, remembers */
    /* one coefficient changed by one / one sign flipped / one integer object replaced by an equal one / one overwritten with a small value */
    for (int edit = 0; edit < 4; edit++) {
      MPI *c = &ct1.c0.coeffs[(12345 + 977 * edit) % polyctx.n];
      if (edit == 0) gcry_mpi_add_ui(*c, *c, 1);
      if (edit == 1) { c = &ct2.c1.coeffs[4242 % polyctx.n]; gcry_mpi_neg(*c, *c); }
      if (edit == 2) { MPI t = gcry_mpi_copy(*c); gcry_mpi_release(*c); *c = t; }
      if (edit == 3) { c = &ct1.c1.coeffs[7 % polyctx.n]; gcry_mpi_set_ui(*c, 3); }
      gpq_mpi_shim_poly_stats(&ok0, &stale0);
      he_mul(&w, &ct1, &ct2, &rlk);
      gpq_mpi_shim_poly_stats(&ok1, &stale1);
      gpq_mpi_shim_set_poly_slots(0);
      he_mul(&v, &ct1, &ct2, &rlk);
      gpq_mpi_shim_set_poly_slots(32);
      const int changed = !SAME_CT(w, u), expect_change = edit != 2;
      if (!SAME_CT(w, v) || changed != expect_change || (stale1 - stale0) != (uint64_t)expect_change) {
        bad = 1;
        printf("resident polynomials: edit %d -- result %s a fresh upload, %s the product before the edit, %llu operand(s) found changed\n", edit,
               SAME_CT(w, v) ? "equals" : "DIFFERS from", changed ? "differs from" : "equals", (unsigned long long)(stale1 - stale0));
      }
      he_mul(&u, &ct1, &ct2, &rlk);
    }
    /* the chain of src/he-algo.c:140-160 on one ciphertext: square, rescale, square, rescale, rotate-free -- with and without the memory */
    for (int pass = 0; pass < 2; pass++) {
      he_ct_t *c = pass ? &w : &v;
      gpq_mpi_shim_set_poly_slots(pass ? 32 : 0);
      for (unsigned i = 0; i < polyctx.n; i++) { gcry_mpi_set(c->c0.coeffs[i], ct1.c0.coeffs[i]); gcry_mpi_set(c->c1.coeffs[i], ct1.c1.coeffs[i]); }
      c->l = hectx.L; c->nu = 1.0; c->B = 1.0;
      for (int step = 0; step < 3; step++) {
        he_mul(c, c, c, &rlk);
        if (step == 1) gcry_mpi_add_ui(c->c1.coeffs[99], c->c1.coeffs[99], 5);     /* the caller touches the result between two calls */
        he_rescale(c);
        if (step == 0) he_moddown(c);
      }
    }
    gpq_mpi_shim_set_poly_slots(32);
    if (!SAME_CT(v, w) || v.l != w.l) { bad = 1; printf("resident polynomials: the chained calls DIFFER from the same chain with fresh uploads\n"); }
    /* poly_mul as he_dec calls it (src/he-enc.c): a chained ciphertext's c1 times the same small polynomial, twice with the memory, once without */
    for (unsigned i = 0; i < polyctx.n; i++) gcry_mpi_set_ui(u.c0.coeffs[i], (unsigned long)(splitmix64(&st) % 3));
    poly_mul(&u.c1, &w.c1, &u.c0, hectx.dim, hectx.q[w.l]);
    poly_mul(&u.c1, &w.c1, &u.c0, hectx.dim, hectx.q[w.l]);
    gcry_mpi_add_ui(w.c1.coeffs[5], w.c1.coeffs[5], 1);
    poly_mul(&u.c1, &w.c1, &u.c0, hectx.dim, hectx.q[w.l]);
    gpq_mpi_shim_set_poly_slots(0);
    poly_mul(&v.c1, &w.c1, &u.c0, hectx.dim, hectx.q[w.l]);
    gpq_mpi_shim_set_poly_slots(32);
    for (unsigned i = 0; i < polyctx.n; i++) if (gcry_mpi_cmp(u.c1.coeffs[i], v.c1.coeffs[i])) { bad = 1; printf("resident polynomials: poly_mul DIFFERS from a fresh upload at %u\n", i); break; }
    {
      /* a polynomial the caller keeps changing behind the library's back (he_add and friends run on the host) is not guessed at again:
       * after the first wrong guess its calls convert and upload before the device starts, until one finds it unchanged */
      uint64_t s0 = 0, s1 = 0, c0 = 0, c1 = 0;
      he_mul(&u, &ct1, &ct2, &rlk);
      he_mul(&u, &ct1, &ct2, &rlk);
      gpq_mpi_shim_poly_stats(&c0, &s0);
      for (int k = 0; k < 3; k++) { gcry_mpi_add_ui(ct2.c0.coeffs[k], ct2.c0.coeffs[k], 1); he_mul(&u, &ct1, &ct2, &rlk); }
      gpq_mpi_shim_poly_stats(&c1, &s1);
      if (s1 - s0 != 1) { bad = 1; printf("resident polynomials: %llu wrong guesses for an operand changed three times in a row (one expected)\n", (unsigned long long)(s1 - s0)); }
      he_mul(&u, &ct1, &ct2, &rlk);                           /* unchanged this time: converted and uploaded, trusted again ... */
      gpq_mpi_shim_poly_stats(&c0, &s0);
      he_mul(&v, &ct1, &ct2, &rlk);                           /* ... and served from the device copies */
      gpq_mpi_shim_poly_stats(&c1, &s1);
      if (c1 - c0 != 4 || s1 != s0 || !SAME_CT(u, v)) { bad = 1; printf("resident polynomials: trust not regained (%llu confirmed, %llu changed)\n", (unsigned long long)(c1 - c0), (unsigned long long)(s1 - s0)); }
    }
    gpq_mpi_shim_poly_stats(&ok1, &stale1);
    printf("resident polynomials: %s (%u resident, %llu operands confirmed, %llu found changed and uploaded again)\n", bad ? "MISMATCH" : "edits and chains identical to fresh uploads",
           gpq_mpi_shim_resident_polys(), (unsigned long long)ok1, (unsigned long long)stale1);
  }
  enum { CALLS = 50 };
  double tm[CALLS], tsq[CALLS], trs[CALLS], part[8];
  /* (a) operands the library has never seen (gpq_mpi_shim_set_poly_slots(0): every call converts and uploads before the device starts) */
  gpq_mpi_shim_set_poly_slots(0);
  for (int i = 0; i < 3; i++) he_mul(&ct, &ct1, &ct2, &rlk);
  for (int i = 0; i < CALLS; i++) { const double t0 = now_ms(); he_mul(&ct, &ct1, &ct2, &rlk); tm[i] = now_ms() - t0; }
  gpq_mpi_shim_last_timing(part);
  he_mul(&ct, &ct1, &ct1, &rlk);                              /* a squaring, as he_mul(&bn, &bn, &bn, rlk) at src/he-algo.c:151 */
  for (int i = 0; i < CALLS; i++) { const double t0 = now_ms(); he_mul(&ct, &ct1, &ct1, &rlk); tsq[i] = now_ms() - t0; }
  gpq_mpi_shim_last_timing(part + 4);
  ct.l = hectx.L; he_rescale(&ct);                            /* first call at this shape allocates the staging buffers */
  for (int i = 0; i < CALLS; i++) {                           /* a full-size ciphertext every time (rescaling the same one again and again would time ever smaller integers) */
    for (unsigned k = 0; k < polyctx.n; k++) { gcry_mpi_set(ct.c0.coeffs[k], ct1.c0.coeffs[k]); gcry_mpi_set(ct.c1.coeffs[k], ct1.c1.coeffs[k]); }
    ct.l = hectx.L;
    const double t0 = now_ms(); he_rescale(&ct); trs[i] = now_ms() - t0;
  }
  qsort(tm, CALLS, sizeof *tm, cmp_double); qsort(tsq, CALLS, sizeof *tsq, cmp_double); qsort(trs, CALLS, sizeof *trs, cmp_double);
  printf("he_mul(MPI) n=2^%u logq=%u dims %u/%u: %.1f ms per call; he_rescale %.1f ms\n", logn, logq, hectx.dim, hectx.dimevk, tm[CALLS / 2], trs[CALLS / 2]);
  printf("  %d calls each: he_mul p50 %.2f p95 %.2f min %.2f max %.2f ms; squaring p50 %.2f p95 %.2f ms; he_rescale p50 %.2f p95 %.2f ms\n", CALLS,
         tm[CALLS / 2], tm[CALLS * 95 / 100], tm[0], tm[CALLS - 1], tsq[CALLS / 2], tsq[CALLS * 95 / 100], trs[CALLS / 2], trs[CALLS * 95 / 100]);
  printf("  last he_mul: convert+upload %.2f ms, kernels %.2f ms, download+convert %.2f ms, call %.2f ms\n", part[0], part[1], part[2], part[3]);
  printf("  squaring he_mul(&ct, &a, &a): %.1f ms per call (convert+upload %.2f ms, kernels %.2f ms, download+convert %.2f ms)\n", tsq[CALLS / 2], part[4], part[5], part[6]);
  /* (b) chained calls, GPQHE's own pattern (src/he-algo.c:140-160): the operands of a call are what an earlier call wrote or read */
  gpq_mpi_shim_set_poly_slots(32);
  for (int i = 0; i < 3; i++) he_mul(&ct, &ct1, &ct2, &rlk);
  for (int i = 0; i < CALLS; i++) { const double t0 = now_ms(); he_mul(&ct, &ct1, &ct2, &rlk); tm[i] = now_ms() - t0; }
  gpq_mpi_shim_last_timing(part);
  ct.l = hectx.L;
  he_mul(&ct, &ct, &ct, &rlk);
  for (int i = 0; i < CALLS; i++) { const double t0 = now_ms(); he_mul(&ct, &ct, &ct, &rlk); tsq[i] = now_ms() - t0; }     /* x -> x^2 -> x^4 ... on one ciphertext */
  gpq_mpi_shim_last_timing(part + 4);
  for (int i = 0; i < CALLS; i++) { ct.l = hectx.L; he_mul(&ct, &ct1, &ct2, &rlk); const double t0 = now_ms(); he_rescale(&ct); trs[i] = now_ms() - t0; }
  qsort(tm, CALLS, sizeof *tm, cmp_double); qsort(tsq, CALLS, sizeof *tsq, cmp_double); qsort(trs, CALLS, sizeof *trs, cmp_double);
  printf("  chained (operands resident, checked while the device works): he_mul p50 %.2f p95 %.2f ms; he_mul(&ct, &ct, &ct) p50 %.2f p95 %.2f ms; he_rescale of a product p50 %.2f p95 %.2f ms\n",
         tm[CALLS / 2], tm[CALLS * 95 / 100], tsq[CALLS / 2], tsq[CALLS * 95 / 100], trs[CALLS / 2], trs[CALLS * 95 / 100]);
  printf("  last chained he_mul: before the device starts %.2f ms, kernels %.2f ms, check+download+convert %.2f ms, call %.2f ms\n", part[0], part[1], part[2], part[3]);
  {
    /* the additive calls in a chain (he_inv's iteration: he_mul, he_rs, he_addpt, he_mul, he_rs, src/he-algo.c:151-155) */
    he_pt_t one;
    poly_alloc(&one.m);
    one.nu = 1.0;
    gcry_mpi_set_ui(one.m.coeffs[0], 1ul << 20);
    double tadd[CALLS], tapt[CALLS], tneg[CALLS];
    he_ct_t acc;
    poly_alloc(&acc.c0); poly_alloc(&acc.c1);
    ct.l = hectx.L; he_mul(&ct, &ct1, &ct2, &rlk);
    he_add(&acc, &ct, &ct); he_addpt(&acc, &ct, &one); he_neg(&acc);
    for (int i = 0; i < CALLS; i++) {
      he_mul(&ct, &ct1, &ct2, &rlk);
      double t0 = now_ms(); he_add(&acc, &acc, &ct); tadd[i] = now_ms() - t0;
      t0 = now_ms(); he_addpt(&acc, &acc, &one); tapt[i] = now_ms() - t0;
      t0 = now_ms(); he_neg(&acc); tneg[i] = now_ms() - t0;
    }
    qsort(tadd, CALLS, sizeof *tadd, cmp_double); qsort(tapt, CALLS, sizeof *tapt, cmp_double); qsort(tneg, CALLS, sizeof *tneg, cmp_double);
    {
      /* what the same he_add costs where the reference runs it: 2n x (mpi_addm, mpi_mod, mpi_cmp, mpi_sub) in libgcrypt on the host (src/he-add.c:40-45) */
      MPI qh = gcry_mpi_new(0);
      gcry_mpi_rshift(qh, hectx.q[hectx.L], 1);
      const double t0 = now_ms();
      for (unsigned i = 0; i < polyctx.n; i++)
        for (int h = 0; h < 2; h++) {
          MPI r = h ? acc.c1.coeffs[i] : acc.c0.coeffs[i];
          gcry_mpi_addm(r, h ? ct1.c1.coeffs[i] : ct1.c0.coeffs[i], h ? ct2.c1.coeffs[i] : ct2.c0.coeffs[i], hectx.q[hectx.L]);
          gcry_mpi_mod(r, r, hectx.q[hectx.L]);
          if (gcry_mpi_cmp(r, qh) >= 0) gcry_mpi_sub(r, r, hectx.q[hectx.L]);
        }
      printf("  (libgcrypt on the host, the reference's way: one he_add %.1f ms)\n", now_ms() - t0);
    }
    {
      double tcp[CALLS];
      he_ct_t cp;
      poly_alloc(&cp.c0); poly_alloc(&cp.c1);
      he_copy_ct(&cp, &acc);
      for (int i = 0; i < CALLS; i++) { he_neg(&acc); const double t0 = now_ms(); he_copy_ct(&cp, &acc); tcp[i] = now_ms() - t0; }
      qsort(tcp, CALLS, sizeof *tcp, cmp_double);
      const double t0 = now_ms();
      for (unsigned i = 0; i < polyctx.n; i++) { gcry_mpi_set(cp.c0.coeffs[i], acc.c0.coeffs[i]); gcry_mpi_set(cp.c1.coeffs[i], acc.c1.coeffs[i]); }
      printf("  he_copy_ct of a chained ciphertext p50 %.2f p95 %.2f ms (2n mpi_set on the host, the reference's way: %.1f ms)\n", tcp[CALLS / 2], tcp[CALLS * 95 / 100], now_ms() - t0);
    }
    printf("  additive calls in a chain (src/he-add.c): he_add p50 %.2f p95 %.2f ms; he_addpt p50 %.2f p95 %.2f ms; he_neg p50 %.2f p95 %.2f ms\n",
           tadd[CALLS / 2], tadd[CALLS * 95 / 100], tapt[CALLS / 2], tapt[CALLS * 95 / 100], tneg[CALLS / 2], tneg[CALLS * 95 / 100]);
  }
  {
    /* he_gemv's per-diagonal calls (src/he-algo.c:66-78) on a chained ciphertext: he_copy_ct, he_rot, he_mulpt (a new plaintext every time:
     * he_ecd runs on the host), he_add */
    he_evk_t rk[3];
    for (int i = 1; i < 3; i++) {
      rk[i].p0.coeffs = malloc((size_t)hectx.dimevk * polyctx.n * 8); rk[i].p1.coeffs = malloc((size_t)hectx.dimevk * polyctx.n * 8);
      struct rns_ctx *rr = polyctx.rns;
      for (unsigned d = 0; d < hectx.dimevk; d++, rr = rr->next)
        for (unsigned k = 0; k < polyctx.n; k++) { rk[i].p0.coeffs[(size_t)d * polyctx.n + k] = splitmix64(&st) % rr->p; rk[i].p1.coeffs[(size_t)d * polyctx.n + k] = splitmix64(&st) % rr->p; }
    }
    rk[0].p0.coeffs = rk[0].p1.coeffs = NULL;
    he_pt_t diag;
    poly_alloc(&diag.m);
    diag.nu = 1024.0;
    he_ct_t rot, sum;
    poly_alloc(&rot.c0); poly_alloc(&rot.c1); poly_alloc(&sum.c0); poly_alloc(&sum.c1);
    double trot[CALLS], tmpt[CALLS];
    ct1.l = hectx.L;
    he_copy_ct(&sum, &ct1);
    for (int i = -2; i < CALLS; i++) {
      for (unsigned k = 0; k < polyctx.n; k += 64) gcry_mpi_set_ui(diag.m.coeffs[k], (unsigned long)(splitmix64(&st) >> 24));   /* "he_ecd": a plaintext the library has not seen */
      he_copy_ct(&rot, &ct1);
      double t0 = now_ms(); he_rot(&rot, 1 + (i & 1), rk); if (i >= 0) trot[i] = now_ms() - t0;
      t0 = now_ms(); he_mulpt(&rot, &rot, &diag); if (i >= 0) tmpt[i] = now_ms() - t0;
      he_add(&sum, &sum, &rot);
    }
    qsort(trot, CALLS, sizeof *trot, cmp_double); qsort(tmpt, CALLS, sizeof *tmpt, cmp_double);
    printf("  he_gemv's per-diagonal calls in a chain: he_rot p50 %.2f p95 %.2f ms; he_mulpt with a new plaintext p50 %.2f p95 %.2f ms\n",
           trot[CALLS / 2], trot[CALLS * 95 / 100], tmpt[CALLS / 2], tmpt[CALLS * 95 / 100]);
  }
  {
    /* he_inv's own call sequence (src/he-algo.c:130-165) with 8 iterations: every call it makes is one of this library's symbols, so the
     * reference's he_inv, unchanged, runs like this -- on ciphertexts that never leave the device between calls */
    he_pt_t one, two;
    poly_alloc(&one.m); poly_alloc(&two.m);
    one.nu = two.nu = 1.0;
    gcry_mpi_set_ui(one.m.coeffs[0], 1); gcry_mpi_lshift(one.m.coeffs[0], one.m.coeffs[0], 50);
    gcry_mpi_set_ui(two.m.coeffs[0], 2); gcry_mpi_lshift(two.m.coeffs[0], two.m.coeffs[0], 50);
    he_ct_t tmp, an, bn, inv;
    poly_mpi_t *ps2[8] = {&tmp.c0, &tmp.c1, &an.c0, &an.c1, &bn.c0, &bn.c1, &inv.c0, &inv.c1};
    for (int i = 0; i < 8; i++) poly_alloc(ps2[i]);
    const int inv_iters = hectx.L > 8 ? 8 : (int)hectx.L - 1;  /* every iteration and the he_moddown in front spend a level (q = 2^438: 8 levels) */
    for (int mem = 1; mem >= 0; mem--) {
      gpq_mpi_shim_set_poly_slots(mem ? 32 : 0);
      double total = 0;
      for (int pass = 0; pass < 2; pass++) {                  /* the second pass is timed (the first builds the per-level tables) */
        ct1.l = hectx.L;
        const double t0 = now_ms();
        he_copy_ct(&tmp, &ct1);
        he_neg(&tmp);
        he_addpt(&an, &tmp, &two);
        he_moddown(&an);
        he_addpt(&bn, &tmp, &one);
        for (int it = 0; it < inv_iters; it++) {
          he_mul(&bn, &bn, &bn, &rlk);
          he_rs(&bn);
          he_addpt(&tmp, &bn, &one);
          he_mul(&an, &an, &tmp, &rlk);
          he_rs(&an);
        }
        he_copy_ct(&inv, &an);
        total = now_ms() - t0;
      }
      printf("  he_inv's call sequence, %d iterations (%d calls, level %u -> %u), %s: %.1f ms\n", inv_iters, 5 + 5 * inv_iters, hectx.L, inv.l, mem ? "operands resident" : "every call uploads", total);
    }
    gpq_mpi_shim_set_poly_slots(32);
  }
  /* (c) the whole ladder, as he_inv / he_exp walk it (src/he-algo.c:140-160): square and rescale from level L down to level 1 on one
   * ciphertext; the second descent is timed (the first builds the per-level device tables), once with and once without the memory */
  for (int mem = 1; mem >= 0; mem--) {
    gpq_mpi_shim_set_poly_slots(mem ? 32 : 0);
    double total = 0;
    for (int pass = 0; pass < 2; pass++) {
      for (unsigned k = 0; k < polyctx.n; k++) { gcry_mpi_set(ct.c0.coeffs[k], ct1.c0.coeffs[k]); gcry_mpi_set(ct.c1.coeffs[k], ct1.c1.coeffs[k]); }
      ct.l = hectx.L; ct.nu = 1.0; ct.B = 1.0;
      const double t0 = now_ms();
      while (ct.l > 0) { he_mul(&ct, &ct, &ct, &rlk); he_rescale(&ct); }
      total = now_ms() - t0;
    }
    printf("  ladder of %u x (he_mul(&ct, &ct, &ct) + he_rescale), level %u -> 0, %s: %.1f ms\n", hectx.L, hectx.L, mem ? "operands resident" : "every call uploads", total);
    if (gcry_mpi_get_nbits(hectx.q[0]) == 1) {               /* q_0 = 1: mpi_smod by 1 makes every coefficient -1 (src/types.c:108-113) */
      MPI m1 = gcry_mpi_new(0);
      gcry_mpi_set_ui(m1, 1); gcry_mpi_neg(m1, m1);
      int all = 1;
      for (unsigned k = 0; k < polyctx.n; k++) if (gcry_mpi_cmp(ct.c0.coeffs[k], m1) || gcry_mpi_cmp(ct.c1.coeffs[k], m1)) all = 0;
      printf("  level 0 has q_0 = 1: every coefficient %s\n", all ? "is -1, as mpi_smod by 1 leaves it" : "SHOULD be -1");
    }
  }
  gpq_mpi_shim_set_poly_slots(32);
  return 0;
}

/* Random walk over the MPI-typed calls with the resident polynomials in play: every step runs once on a ciphertext the library may hold
 * copies of and once, with gpq_mpi_shim_poly_bypass, on its twin that is always converted and uploaded afresh; after every step the two must be
 * the same integers.  Steps: he_mul with every aliasing pattern, he_rescale, he_moddown, he_rot, he_conj, he_mulpt, host-side edits of a
 * ciphertext (as he_add and friends do), copies between ciphertexts, a smaller number of slots (evictions), a rewritten key. */
static void fill_key(he_evk_t *k, uint64_t seed)
{
  struct rns_ctx *r = polyctx.rns;
  for (unsigned d = 0; d < hectx.dimevk; d++, r = r->next)
    for (unsigned i = 0; i < polyctx.n; i++) {
      k->p0.coeffs[(size_t)d * polyctx.n + i] = splitmix64(&seed) % r->p;
      k->p1.coeffs[(size_t)d * polyctx.n + i] = splitmix64(&seed) % r->p;
    }
}

static int residentfuzz(unsigned logn, unsigned logq, unsigned logDelta, unsigned steps, uint64_t seed)
{
  MPI q = pow2(logq);
  he_ctx_init(logn, q, 1ull << logDelta);
  enum { K = 4 };
  he_ct_t x[K], y[K];                                         /* y[i] is the twin of x[i] */
  const unsigned n = polyctx.n;
  uint64_t st = seed;
  unsigned char *buf = malloc(logq / 8 + 16);
  for (int i = 0; i < K; i++) {
    poly_alloc(&x[i].c0); poly_alloc(&x[i].c1); poly_alloc(&y[i].c0); poly_alloc(&y[i].c1);
    poly_mpi_t *ps[2] = {&x[i].c0, &x[i].c1}, *pt[2] = {&y[i].c0, &y[i].c1};
    for (int h = 0; h < 2; h++)
      for (unsigned k = 0; k < n; k++) {
        const unsigned nb = (logq - 2) / 8;
        for (unsigned b = 0; b < nb; b += 8) { uint64_t v = splitmix64(&st); memcpy(buf + b, &v, 8); }
        MPI t = NULL;
        gcry_mpi_scan(&t, 5, buf, nb, NULL);
        if (splitmix64(&st) & 1) gcry_mpi_neg(t, t);
        gcry_mpi_release(ps[h]->coeffs[k]); ps[h]->coeffs[k] = t;
        gcry_mpi_set(pt[h]->coeffs[k], t);
      }
    x[i].l = y[i].l = hectx.L; x[i].nu = y[i].nu = 1.0; x[i].B = y[i].B = 1.0;
  }
  he_evk_t keys[4];                                           /* rlk, rk[1], rk[2], ck */
  for (int i = 0; i < 4; i++) {
    keys[i].p0.coeffs = malloc((size_t)hectx.dimevk * n * 8); keys[i].p1.coeffs = malloc((size_t)hectx.dimevk * n * 8);
    fill_key(&keys[i], 7000 + i);
  }
  he_evk_t rk[3] = {{{NULL}, {NULL}}, keys[1], keys[2]};
  he_pt_t pt;
  poly_alloc(&pt.m);
  pt.nu = 1024.0;
  for (unsigned i = 0; i < n; i++) { gcry_mpi_set_ui(pt.m.coeffs[i], (unsigned long)(splitmix64(&st) >> 40)); if (i & 1) gcry_mpi_neg(pt.m.coeffs[i], pt.m.coeffs[i]); }
  unsigned count[17] = {0};
  poly_mpi_t sk;                                              /* a dense ternary secret key */
  poly_alloc(&sk);
  for (unsigned i = 0; i < n; i++) { const unsigned t = (unsigned)(splitmix64(&st) % 3); gcry_mpi_set_ui(sk.coeffs[i], t ? 1 : 0); if (t == 2) gcry_mpi_neg(sk.coeffs[i], sk.coeffs[i]); }
  he_pt_t dx, dy;
  poly_alloc(&dx.m); poly_alloc(&dy.m);
#define TWIN(call_x, call_y) do { call_x; gpq_mpi_shim_poly_bypass(1); call_y; gpq_mpi_shim_poly_bypass(0); } while (0)
  for (unsigned step = 0; step < steps; step++) {
    const unsigned op = (unsigned)(splitmix64(&st) % 17), a = (unsigned)(splitmix64(&st) % K), b = (unsigned)(splitmix64(&st) % K), d = (unsigned)(splitmix64(&st) % K);
    int touched = -1;
    if (op <= 2) {                                            /* he_mul: dst may be a, b, both or neither */
      if (x[a].l != x[b].l || x[a].l == 0) continue;
      TWIN(he_mul(&x[d], &x[a], &x[b], &keys[0]), he_mul(&y[d], &y[a], &y[b], &keys[0]));
      touched = (int)d;
    } else if (op == 3) {
      if (x[a].l < 2) continue;
      TWIN(he_rescale(&x[a]), he_rescale(&y[a]));
      touched = (int)a;
    } else if (op == 4) {
      if (x[a].l < 2) continue;
      TWIN(he_moddown(&x[a]), he_moddown(&y[a]));
      touched = (int)a;
    } else if (op == 5) {
      if (x[a].l == 0) continue;
      const int r = 1 + (int)(b & 1);
      TWIN(he_rot(&x[a], r, rk), he_rot(&y[a], r, rk));
      touched = (int)a;
    } else if (op == 6) {
      if (x[a].l == 0) continue;
      TWIN(he_conj(&x[a], &keys[3]), he_conj(&y[a], &keys[3]));
      touched = (int)a;
    } else if (op == 7) {
      if (x[a].l == 0) continue;
      TWIN(he_mulpt(&x[d], &x[a], &pt), he_mulpt(&y[d], &y[a], &pt));
      touched = (int)d;
    } else if (op == 8) {                                     /* the host changes a ciphertext behind the library's back */
      const unsigned k = (unsigned)(splitmix64(&st) % n), how = (unsigned)(splitmix64(&st) % 4);
      poly_mpi_t *px = (splitmix64(&st) & 1) ? &x[a].c0 : &x[a].c1, *py = px == &x[a].c0 ? &y[a].c0 : &y[a].c1;
      if (how == 0) { gcry_mpi_add_ui(px->coeffs[k], px->coeffs[k], 1); gcry_mpi_add_ui(py->coeffs[k], py->coeffs[k], 1); }
      if (how == 1) { gcry_mpi_neg(px->coeffs[k], px->coeffs[k]); gcry_mpi_neg(py->coeffs[k], py->coeffs[k]); }
      if (how == 2) { gcry_mpi_set_ui(px->coeffs[k], 5); gcry_mpi_set_ui(py->coeffs[k], 5); }
      if (how == 3) { MPI t = gcry_mpi_copy(px->coeffs[k]); gcry_mpi_release(px->coeffs[k]); px->coeffs[k] = t; }
    } else if (op == 9 && (splitmix64(&st) & 1)) {            /* x[d] = x[a] through he_copy_ct (src/he-mem.c:88-97) */
      if (a == d) continue;
      TWIN(he_copy_ct(&x[d], &x[a]), he_copy_ct(&y[d], &y[a]));
      touched = (int)d;
    } else if (op == 9) {                                     /* x[d] = x[a] on the host, behind the library's back */
      if (a == d) continue;
      for (unsigned k = 0; k < n; k++) {
        gcry_mpi_set(x[d].c0.coeffs[k], x[a].c0.coeffs[k]); gcry_mpi_set(x[d].c1.coeffs[k], x[a].c1.coeffs[k]);
        gcry_mpi_set(y[d].c0.coeffs[k], y[a].c0.coeffs[k]); gcry_mpi_set(y[d].c1.coeffs[k], y[a].c1.coeffs[k]);
      }
      x[d].l = x[a].l; y[d].l = y[a].l;
    } else if (op == 12) {                                    /* poly_mul as he_dec / he_enc call it: one polynomial of a times one of b, mod q of d's level, into d */
      if (x[d].l == 0) continue;
      TWIN(poly_mul(&x[d].c0, &x[a].c1, &x[b].c0, hectx.dimevk, hectx.q[x[d].l]), poly_mul(&y[d].c0, &y[a].c1, &y[b].c0, hectx.dimevk, hectx.q[y[d].l]));
      touched = (int)d;
    } else if (op == 13) {                                    /* src/he-add.c: he_add / he_sub, any aliasing */
      if (x[a].l != x[b].l || x[a].l == 0) continue;
      if (splitmix64(&st) & 1) TWIN(he_add(&x[d], &x[a], &x[b]), he_add(&y[d], &y[a], &y[b]));
      else TWIN(he_sub(&x[d], &x[a], &x[b]), he_sub(&y[d], &y[a], &y[b]));
      touched = (int)d;
    } else if (op == 14) {                                    /* he_addpt / he_subpt */
      if (x[a].l == 0) continue;
      if (splitmix64(&st) & 1) TWIN(he_addpt(&x[d], &x[a], &pt), he_addpt(&y[d], &y[a], &pt));
      else TWIN(he_subpt(&x[d], &x[a], &pt), he_subpt(&y[d], &y[a], &pt));
      touched = (int)d;
    } else if (op == 15) {
      if (x[a].l == 0) continue;
      TWIN(he_neg(&x[a]), he_neg(&y[a]));
      touched = (int)a;
    } else if (op == 16) {                                    /* he_dec (src/he-encrypt.c:105-125) */
      if (x[a].l == 0) continue;
      TWIN(he_dec(&dx, &x[a], &sk), he_dec(&dy, &y[a], &sk));
      for (unsigned k = 0; k < n; k++)
        if (gcry_mpi_cmp(dx.m.coeffs[k], dy.m.coeffs[k])) { printf("step %u he_dec (a %u, level %u): coefficient %u differs from the twin computed from fresh uploads\n", step, a, x[a].l, k); return 1; }
    } else if (op == 10) {
      gpq_mpi_shim_set_poly_slots(2 + (unsigned)(splitmix64(&st) % 7));
    } else {                                                  /* a key rewritten in place, in one word or in all */
      he_evk_t *k = &keys[splitmix64(&st) % 4];
      if (splitmix64(&st) & 1) fill_key(k, splitmix64(&st));
      else { const size_t at = splitmix64(&st) % ((size_t)hectx.dimevk * n); k->p1.coeffs[at] = k->p1.coeffs[at] > 9 ? k->p1.coeffs[at] - 9 : k->p1.coeffs[at] + 9; }
    }
    count[op]++;
    if (touched >= 0) {
      if (x[touched].l != y[touched].l) { printf("step %u op %u: levels differ\n", step, op); return 1; }
      for (unsigned k = 0; k < n; k++)
        if (gcry_mpi_cmp(x[touched].c0.coeffs[k], y[touched].c0.coeffs[k]) || gcry_mpi_cmp(x[touched].c1.coeffs[k], y[touched].c1.coeffs[k])) {
          printf("step %u op %u (a %u b %u d %u, level %u): coefficient %u differs from the twin computed from fresh uploads\n", step, op, a, b, d, x[touched].l, k);
          return 1;
        }
    }
    /* ladders run out of levels: put a spent ciphertext back on top now and then (both twins alike) */
    if (x[a].l < 2 && (splitmix64(&st) & 1)) { x[a].l = y[a].l = hectx.L; }
  }
  uint64_t confirmed = 0, changed = 0;
  gpq_mpi_shim_poly_stats(&confirmed, &changed);
  printf("residentfuzz ok: %u steps (mul %u rs %u moddown %u rot %u conj %u mulpt %u edit %u copy %u slots %u key %u poly_mul %u add/sub %u addpt/subpt %u neg %u dec %u), %llu operands confirmed, %llu found changed\n", steps,
         count[0] + count[1] + count[2], count[3], count[4], count[5], count[6], count[7], count[8], count[9], count[10], count[11], count[12], count[13], count[14], count[15], count[16],
         (unsigned long long)confirmed, (unsigned long long)changed);
  return 0;
}

/* hectx_init / polyctx_init / poly_*_alloc of the library (weak definitions), printed field by field for the Python side to compare
 * with the restated formulas (src/precomp.c:266-293, :328-450) and with the dims SURVEY.md 8c captured from the reference */
static int ctxcheck(unsigned logn, unsigned logq, unsigned long long Delta)
{
  MPI q = pow2(logq);
  hectx_init(logn, q, 2, Delta);
  printf("poly %u %u %u %u %u %u\n", polyctx.logn, polyctx.n, polyctx.m, polyctx.logq, polyctx.logqub, polyctx.dimub);
  printf("two "); print_mpi(GPQHE_TWO);
  printf("q "); print_mpi(polyctx.q);
  unsigned count = 0;
  for (struct rns_ctx *r = polyctx.rns; r; r = r->next, count++) {
    printf("node %u %llu %llu %llu %llu %llu %llu\n", r->dim, (unsigned long long)r->p, (unsigned long long)r->pinv_mont,
           (unsigned long long)r->pinv_barr, (unsigned long long)r->ninv, (unsigned long long)r->zetas[1], (unsigned long long)r->zetas_inv[polyctx.n / 2]);
    if (r->dim <= 5 || !r->next) {                           /* the big integers of a few prefixes */
      printf("P "); print_mpi(r->P);
      printf("P_2 "); print_mpi(r->P_2);
      for (unsigned d = 0; d < r->dim; d++) { printf("phat %u %llu ", d, (unsigned long long)r->phat_invmp[d]); print_mpi(r->phat[d]); }
    }
  }
  printf("count %u\n", count);
  printf("ring %u %u %.17g %.17g\n", polyctx.ring.cyc_group[1], polyctx.ring.cyc_group[polyctx.n / 2 - 1],
         __real__ polyctx.ring.zetas[1], __imag__ polyctx.ring.zetas[polyctx.m]);
  printf("he %u %u %u %u %u %u %.17g\n", hectx.L, hectx.dim, hectx.dimevk, gcry_mpi_get_nbits(hectx.P), gcry_mpi_get_nbits(hectx.PqL), hectx.slots, hectx.Delta);
  for (unsigned l = 0; l <= hectx.L; l++) { printf("q %u ", l); print_mpi(hectx.q[l]); printf("qh %u ", l); print_mpi(hectx.qh[l]); }
  printf("bnd %.17g %.17g %.17g %.17g %.17g\n", hectx.bnd.Bclean, hectx.bnd.Brs, hectx.bnd.Bks, hectx.bnd.Bmult[0], hectx.bnd.Bmult[hectx.L]);
  poly_mpi_t a;
  poly_rns_t h;
  poly_mpi_alloc(&a);                                       /* src/poly.c:46-51: n fresh MPIs (value 0) */
  poly_rns_alloc(&h, hectx.dimevk);                         /* src/poly.c:60-64 */
  for (unsigned i = 0; i < hectx.dimevk * polyctx.n; i++) h.coeffs[i] = i;
  printf("alloc %u %llu\n", gcry_mpi_get_nbits(a.coeffs[polyctx.n - 1]), (unsigned long long)h.coeffs[hectx.dimevk * polyctx.n - 1]);
  poly_rns_free(&h);
  poly_mpi_free(&a);
  hectx_exit();
  printf("exit %d %d\n", polyctx.rns == NULL, hectx.q == NULL);
  hectx_init(logn, q, 2, Delta);                            /* a context can be built again after hectx_exit */
  printf("again %u %u\n", polyctx.dimub, hectx.dimevk);
  hectx_exit();
  return 0;
}

int main(int argc, char **argv)
{
  if (argc >= 5 && !strcmp(argv[1], "ctxcheck")) return ctxcheck(atoi(argv[2]), atoi(argv[3]), strtoull(argv[4], NULL, 10));
  if (argc >= 2 && !strcmp(argv[1], "polymul")) return polymul(0);
  if (argc >= 2 && !strcmp(argv[1], "polymulodd")) return polymul(1);
  if (argc >= 2 && !strcmp(argv[1], "crt")) return crt();
  if (argc >= 3 && !strcmp(argv[1], "hemul")) return hemul(argv[2]);
  if (argc >= 3 && !strcmp(argv[1], "polymulmono")) return polymulmono(atoi(argv[2]));
  if (argc >= 4 && !strcmp(argv[1], "keygen")) return keygen(atoi(argv[2]), atoi(argv[3]));
  if (argc >= 5 && !strcmp(argv[1], "hemultime")) printf("conversion threads: %u\n", gpq_mpi_shim_set_conversion_threads((unsigned)atoi(argv[4])));
  if (argc >= 4 && !strcmp(argv[1], "hemultime")) return hemultime(atoi(argv[2]), atoi(argv[3]));
  if (argc >= 7 && !strcmp(argv[1], "residentfuzz")) return residentfuzz(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), strtoull(argv[6], NULL, 10));
  return 2;
}
