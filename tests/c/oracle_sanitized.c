/* oracle_sanitized.c -- the CPU oracle (oracle/gpqhe_oracle.c, compiled into this program with
 * -fsanitize=address,undefined) replays the he_mul RNS-core known-answer test of SURVEY.md 8c and prints the
 * FNV-1a digests; tests/test_oracle_sanitized.py compares them with tests/golden/survey_8c.json.
 * Test infrastructure: the sanitizers run on the CPU build only (no GPU sanitizer on this pool).
 * usage: oracle_sanitized <logn> <nprimes> <dimA> <dimB> */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../oracle/gpqhe_oracle.c"

static uint64_t fnv1a(const uint64_t *a, size_t words)
{
  uint64_t h = 0xcbf29ce484222325ull;
  const unsigned char *p = (const unsigned char *)a;
  for (size_t i = 0; i < words * 8; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

int main(int argc, char **argv)
{
  if (argc != 5) return 2;
  const unsigned logn = (unsigned)atoi(argv[1]), np = (unsigned)atoi(argv[2]), dA = (unsigned)atoi(argv[3]), dB = (unsigned)atoi(argv[4]);
  orc_ctx *c = orc_ctx_create(logn, np);
  if (!c) return 3;
  const size_t n = orc_ctx_n(c), wa = (size_t)dA * n, wb = (size_t)dB * n;
  uint64_t *in[4], *d[3], *x = malloc(wb * 8), *e0 = malloc(wb * 8), *e1 = malloc(wb * 8), *c0 = malloc(wb * 8), *c1 = malloc(wb * 8);
  for (int i = 0; i < 4; i++) { in[i] = malloc(wa * 8); orc_gen_slab(c, 1000 + (uint64_t)i, dA, in[i]); printf("in%d %016" PRIx64 "\n", i, fnv1a(in[i], wa)); }
  for (int i = 0; i < 3; i++) d[i] = malloc(wa * 8);
  orc_he_mul_tensor(c, dA, d[0], d[1], d[2], in[0], in[1], in[2], in[3]);
  for (int i = 0; i < 3; i++) printf("d%d %016" PRIx64 "\n", i, fnv1a(d[i], wa));
  orc_gen_slab(c, 2000, dB, x); orc_gen_slab(c, 3000, dB, e0); orc_gen_slab(c, 3001, dB, e1);
  orc_keyswitch(c, dB, c0, c1, x, e0, e1);
  printf("c0 %016" PRIx64 "\nc1 %016" PRIx64 "\n", fnv1a(c0, wb), fnv1a(c1, wb));
  /* the single-limb transforms in place, and their round trip */
  uint64_t *a = malloc(n * 8), *b = malloc(n * 8);
  orc_gen_slab(c, 1, 1, a);
  for (size_t i = 0; i < n; i++) b[i] = a[i];
  orc_ntt(c, 0, a);
  printf("ntt %016" PRIx64 "\n", fnv1a(a, n));
  orc_invntt(c, 0, a);
  for (size_t i = 0; i < n; i++) if (a[i] != b[i]) { printf("roundtrip FAILED at %zu\n", i); return 4; }
  printf("roundtrip ok\n");
  for (int i = 0; i < 4; i++) free(in[i]);
  for (int i = 0; i < 3; i++) free(d[i]);
  free(x); free(e0); free(e1); free(c0); free(c1); free(a); free(b);
  orc_ctx_destroy(c);
  return 0;
}
