/*
 * dropin_host.c -- a C host that uses libgpqhe_hip.so the way GPQHE itself
 * would after the swap described in INTEGRATION.md: it owns the global
 * `polyctx` (as src/precomp.c:41 does), walks a linked list of struct rns_ctx
 * and runs the limb loop of poly_mul (src/poly.c:96-103, minus rns_decompose)
 * through the reference-named symbols ntt / invntt / poly_rns_mul, then the
 * d1 = x + y step of he_mul (src/he-mult.c:136) through poly_rns_add.
 *
 * usage: dropin_host <logn> <dim> <seed>   -> prints FNV-1a-64 digests
 * The pytest wrapper compares them with the oracle's.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gpqhe_hip.h"
#include "gpqhe_hip_compat.h"

struct poly_ctx polyctx; /* the reference's global, src/precomp.c:41: a strong definition, like GPQHE's own, in front of the library's weak one */

static uint64_t splitmix64(uint64_t *s)
{
  uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

static uint64_t fnv(const uint64_t *a, size_t n)
{
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; i++)
    for (int b = 0; b < 8; b++) { h ^= (a[i] >> (8 * b)) & 0xff; h *= 0x100000001b3ull; }
  return h;
}

/* dropin_host xform <logn> <dim> <fwd|inv> <infile> <outfile>: the limbs of a uint64[polys][dim][n] slab read from a file, each through
 * the reference-named symbol `ntt` / `invntt` with its own struct rns_ctx, written back -- any input words (the zero-representation
 * and out-of-domain cases of tests/test_ntt_zero_repr_gpu.py). */
static int xform(int argc, char **argv)
{
  if (argc < 7) return 2;
  unsigned logn = (unsigned)atoi(argv[2]), dim = (unsigned)atoi(argv[3]);
  const int inverse = strcmp(argv[4], "inv") == 0;
  size_t n = (size_t)1 << logn;
  memset(&polyctx, 0, sizeof polyctx);
  polyctx.logn = logn; polyctx.n = (unsigned)n; polyctx.m = 2 * (unsigned)n;
  polyctx.logR = 64; polyctx.R = (gpq_u128)1 << 64; polyctx.Rsub1 = polyctx.R - 1;
  polyctx.dimub = dim;
  gpq_ctx *ctx = NULL;
  if (gpq_ctx_create(&ctx, logn, dim, 0) != GPQ_OK) { fprintf(stderr, "%s\n", gpq_last_error()); return 1; }
  struct rns_ctx *nodes = calloc(dim, sizeof *nodes);
  if (gpq_fill_rns_chain(nodes, dim, ctx, 0) != GPQ_OK) { fprintf(stderr, "gpq_fill_rns_chain failed\n"); return 1; }
  polyctx.rns = nodes;
  FILE *f = fopen(argv[5], "rb");
  if (!f) { fprintf(stderr, "cannot read %s\n", argv[5]); return 1; }
  fseek(f, 0, SEEK_END);
  const size_t limbs = (size_t)ftell(f) / (8 * n);                  /* limb k belongs to prime k mod dim */
  fseek(f, 0, SEEK_SET);
  uint64_t *a = malloc(limbs * n * 8);
  if (fread(a, 8, limbs * n, f) != limbs * n) { fprintf(stderr, "cannot read %s\n", argv[5]); return 1; }
  fclose(f);
  for (size_t k = 0; k < limbs; k++) {
    const struct rns_ctx *rns = &nodes[k % dim];
    if (inverse) invntt(a + k * n, rns); else ntt(a + k * n, rns);
  }
  f = fopen(argv[6], "wb");
  if (!f || fwrite(a, 8, limbs * n, f) != limbs * n) { fprintf(stderr, "cannot write %s\n", argv[6]); return 1; }
  fclose(f);
  gpq_dropin_reset();
  gpq_release_rns_chain(nodes);
  gpq_ctx_destroy(ctx);
  free(nodes); free(a);
  return 0;
}

int main(int argc, char **argv)
{
  if (argc > 1 && strcmp(argv[1], "xform") == 0) return xform(argc, argv);
  if (argc < 4) return 2;
  unsigned logn = (unsigned)atoi(argv[1]), dim = (unsigned)atoi(argv[2]);
  uint64_t seed = strtoull(argv[3], NULL, 10);
  size_t n = (size_t)1 << logn;

  /* what polyctx_init would have filled in (src/precomp.c:333-356) */
  memset(&polyctx, 0, sizeof polyctx);
  polyctx.logn = logn; polyctx.n = (unsigned)n; polyctx.m = 2 * (unsigned)n;
  polyctx.logR = 64; polyctx.R = (gpq_u128)1 << 64; polyctx.Rsub1 = polyctx.R - 1;
  polyctx.dimub = dim;

  /* per-prime tables in the reference's own format, taken from the engine's context */
  gpq_ctx *ctx = NULL;
  if (gpq_ctx_create(&ctx, logn, dim, 0) != GPQ_OK) { fprintf(stderr, "%s\n", gpq_last_error()); return 1; }
  struct rns_ctx *nodes = calloc(dim, sizeof *nodes);
  if (gpq_fill_rns_chain(nodes, dim, ctx, 0) != GPQ_OK) { fprintf(stderr, "gpq_fill_rns_chain failed\n"); return 1; }   /* no libgcrypt here: MPI fields stay NULL */
  if (nodes[dim - 1].dim != dim || nodes[dim - 1].next || nodes[0].P || !nodes[0].phat_invmp || nodes[0].phat_invmp[0] != 1) { fprintf(stderr, "bad chain\n"); return 1; }
  polyctx.rns = nodes;

  /* scalar helpers of src/reduce.c on the first prime */
  uint64_t p0 = nodes[0].p;
  if (montgomery_inv(p0) != nodes[0].pinv_mont || barrett_inv(p0) != nodes[0].pinv_barr) { fprintf(stderr, "inv mismatch\n"); return 1; }
  gpq_u128 prod = (gpq_u128)(p0 - 1) * (p0 - 2);
  uint64_t br = barrett_reduce(prod, p0, nodes[0].pinv_barr);
  uint64_t mr = montgomery_reduce(prod, p0, (int64_t)nodes[0].pinv_mont);
  printf("barrett %llu montgomery %llu\n", (unsigned long long)br, (unsigned long long)mr);

  uint64_t *a = malloc(dim * n * 8), *b = malloc(dim * n * 8), *r = malloc(dim * n * 8), *s = malloc(dim * n * 8);
  uint64_t st = seed;
  for (unsigned d = 0; d < dim; d++) for (size_t i = 0; i < n; i++) a[d * n + i] = splitmix64(&st) % nodes[d].p;
  st = seed + 1;
  for (unsigned d = 0; d < dim; d++) for (size_t i = 0; i < n; i++) b[d * n + i] = splitmix64(&st) % nodes[d].p;

  /* src/poly.c:96-103 */
  struct rns_ctx *rns = polyctx.rns;
  for (unsigned d = 0; d < dim; d++) {
    uint64_t *ahat = a + d * n, *bhat = b + d * n;
    ntt(ahat, rns);
    poly_ntt(bhat, rns);                        /* north-star alias of the same symbol */
    poly_rns_mul(&r[d * n], ahat, bhat, rns);
    invntt(&r[d * n], rns);
    poly_rns_add(&s[d * n], ahat, bhat, rns);   /* src/he-mult.c:136 shape */
    poly_rns_mul(ahat, ahat, bhat, rns);        /* aliased output, src/he-mult.c:130 */
    poly_invntt(ahat, rns);
    rns = (d < dim - 1) ? rns->next : rns;
  }
  if (argc > 4) {                               /* PCIe-inclusive latency of the single-limb symbols */
    struct timespec t0, t1;
    const int reps = atoi(argv[4]);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 0; k < reps; k++) { ntt(b, polyctx.rns); invntt(b, polyctx.rns); }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double us = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / (2.0 * reps);
    printf("latency_us %.1f\n", us);
  }
  printf("ntt_b %016llx\nmul %016llx\nadd %016llx\nalias %016llx\n", (unsigned long long)fnv(b, dim * n),
         (unsigned long long)fnv(r, dim * n), (unsigned long long)fnv(s, dim * n), (unsigned long long)fnv(a, dim * n));
  gpq_dropin_reset();
  gpq_release_rns_chain(nodes);
  gpq_ctx_destroy(ctx);
  free(nodes); free(a); free(b); free(r); free(s);
  return 0;
}
