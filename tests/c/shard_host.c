/*
 * shard_host.c -- a batch of independent ciphertext multiplications sharded over several devices from ONE plain-C process
 * ("one ciphertext per GPU", no exchange inside a transform; src/he-mult.c:116-138 and :58-66 carry no
 * cross-ciphertext state).  Only the C ABI of include/gpqhe_hip.h is used -- no HIP headers, no torch:
 *
 *   shard_host <logn> <dimA> <dimB> <batch> <dev,dev,...> [period [pipe]]
 *
 * Shard s (block partition of the batch, the first batch % shards shards take one more) lives on device <dev_s>: its own
 * WORKER THREAD, context, stream, buffers.  A worker's first action is gpq_bind_thread_to_device(device): it confines itself to the
 * CPUs of its GPU's NUMA node (sysfs, in-process, before its first call that touches the device), so input generation, staging
 * memory and launches stay on the socket the GPU hangs off; eight shards on a two-socket node load over eight PCIe links from both
 * sockets at once.  All workers start before any is waited for; listing a device twice ("0,0") puts two shards with separate
 * contexts and streams on it (what a one-GPU box can exercise).
 * Inputs: ciphertext k uses gen(1000 + 4k .. 1003 + 4k, dimA) and gen(2000 + k, dimB), one key gen(3000 / 3001, dimB) --
 * the synthetic batch of SURVEY.md 8d.  Prints, per ciphertext, the FNV-1a-64 digests of d0, d1, d2, c0, c1; the pytest
 * wrapper compares them with the oracle's.  With a `period` P > 0 ciphertext k carries the inputs of ciphertext k mod P (BASELINE
 * configs[3]'s batch of 512 with P oracle evaluations instead of 512: every ciphertext must print the digests of k mod P).  With `pipe` S > 0
 * every shard runs its ciphertexts as sub-batches of S pipelined over three streams (stages of k on one stream while the outputs of k - 1 leave on
 * another), ordered by gpq_stream_wait alone.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpqhe_hip.h"

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

/* gen(seed, dim) of SURVEY.md 8c: limb-major, a[d n + i] = splitmix64() % p_d from one running state */
static void gen(uint64_t *out, uint64_t seed, unsigned dim, size_t n, const uint64_t *p)
{
  uint64_t st = seed;
  for (unsigned d = 0; d < dim; d++)
    for (size_t i = 0; i < n; i++) out[d * n + i] = splitmix64(&st) % p[d];
}

#define CHECK(x) do { if ((x) != GPQ_OK) { snprintf(h->err, sizeof h->err, "%s: %s", #x, gpq_last_error()); return NULL; } } while (0)

struct shard {
  int index, device; unsigned lo, hi;
  unsigned logn, dimA, dimB, period;
  unsigned pipe;             /* > 0: sub-batches of `pipe` ciphertexts pipelined over three streams (upload | stages | download), gpq_stream_wait between them */
  int bound_cpus;            /* gpq_bind_thread_to_device: CPUs of the GPU's NUMA node this worker is confined to (0 = left where it was) */
  gpq_ctx *ctx; void *stream;
  uint64_t *d_in[4], *d_x, *d_e[2], *d_out[5], *d_wsA, *d_wsB;
  uint64_t *h_out[5];
  char err[600];
  int done;
};

/* One worker thread per shard: placement first (before the thread's first call that touches its device), then its own context, stream,
 * buffers, uploads, the two stages and the downloads -- everything asynchronous on the shard's stream; the thread returns when its stream
 * has drained.  The shards never talk to each other. */
static void *run_shard(void *arg)
{
  struct shard *h = arg;
  const unsigned logn = h->logn, dimA = h->dimA, dimB = h->dimB, period = h->period;
  const size_t n = (size_t)1 << logn, perA = dimA * n, perB = dimB * n;
  const unsigned cnt = h->hi - h->lo;
  h->bound_cpus = gpq_bind_thread_to_device(h->device);
  CHECK(gpq_set_device(h->device));
  CHECK(gpq_ctx_create(&h->ctx, logn, dimB, h->device));
  if (gpq_ctx_device(h->ctx) != h->device) { snprintf(h->err, sizeof h->err, "context on the wrong device"); return NULL; }
  CHECK(gpq_stream_create(&h->stream));
  uint64_t *primes = malloc(dimB * 8), *host = malloc((perA > perB ? perA : perB) * 8);     /* first-touched by this (placed) thread */
  if (!primes || !host) { snprintf(h->err, sizeof h->err, "out of host memory"); return NULL; }
  for (unsigned d = 0; d < dimB; d++) primes[d] = gpq_ctx_const(h->ctx, d, 0);
  for (int i = 0; i < 4; i++) CHECK(gpq_malloc((void **)&h->d_in[i], cnt * perA * 8));
  CHECK(gpq_malloc((void **)&h->d_x, cnt * perB * 8));
  for (int i = 0; i < 2; i++) CHECK(gpq_malloc((void **)&h->d_e[i], perB * 8));
  for (int i = 0; i < 5; i++) {
    const size_t per = i < 3 ? perA : perB;
    CHECK(gpq_malloc((void **)&h->d_out[i], cnt * per * 8));
    CHECK(gpq_malloc_host((void **)&h->h_out[i], cnt * per * 8));       /* page-locked: the download does not block the host */
  }
  CHECK(gpq_malloc((void **)&h->d_wsA, gpq_tensor_workspace_bytes(h->ctx, dimA, cnt)));
  CHECK(gpq_malloc((void **)&h->d_wsB, gpq_keyswitch_workspace_bytes(h->ctx, dimB, cnt)));
  for (unsigned k = h->lo; k < h->hi; k++) {
    const unsigned ks = period ? k % period : k;           /* whose inputs ciphertext k carries */
    if (period && k - h->lo >= period) {                    /* a repeat inside this shard: device-side copies of the first occurrence */
      const unsigned src = k - h->lo - period;
      for (int i = 0; i < 4; i++) CHECK(gpq_copy(h->d_in[i] + (k - h->lo) * perA, h->d_in[i] + src * perA, perA * 8, h->stream));
      CHECK(gpq_copy(h->d_x + (k - h->lo) * perB, h->d_x + src * perB, perB * 8, h->stream));
      continue;
    }
    for (int i = 0; i < 4; i++) {
      gen(host, 1000 + 4 * ks + i, dimA, n, primes);
      CHECK(gpq_upload(h->d_in[i] + (k - h->lo) * perA, host, perA * 8, h->stream));
      CHECK(gpq_stream_sync(h->stream));                 /* `host` is reused (pageable memory: the copy is staged anyway) */
    }
    gen(host, 2000 + ks, dimB, n, primes);
    CHECK(gpq_upload(h->d_x + (k - h->lo) * perB, host, perB * 8, h->stream));
    CHECK(gpq_stream_sync(h->stream));
  }
  for (int i = 0; i < 2; i++) {                            /* the key is replicated on every device */
    gen(host, 3000 + i, dimB, n, primes);
    CHECK(gpq_upload(h->d_e[i], host, perB * 8, h->stream));
    CHECK(gpq_stream_sync(h->stream));
  }
  if (!h->pipe) {
    CHECK(gpq_he_mul_tensor(h->ctx, h->d_out[0], h->d_out[1], h->d_out[2], h->d_in[0], h->d_in[1], h->d_in[2], h->d_in[3], dimA, cnt, h->d_wsA, h->stream));
    CHECK(gpq_keyswitch(h->ctx, h->d_out[3], h->d_out[4], h->d_x, h->d_e[0], h->d_e[1], dimB, cnt, h->d_wsB, h->stream));
    for (int i = 0; i < 5; i++) CHECK(gpq_download(h->h_out[i], h->d_out[i], cnt * (i < 3 ? perA : perB) * 8, h->stream));
    CHECK(gpq_stream_sync(h->stream));
  } else {
    /* The pipelined form: the inputs are on the device already (uploaded above on h->stream = the "upload" stream; a real host would stream them
     * from page-locked memory sub-batch by sub-batch exactly like the downloads below).  Sub-batch k runs its two stages on the compute stream once
     * the upload stream has delivered, and its five output pieces leave on the download stream once the stages are done, while sub-batch k + 1
     * computes: three streams, ordered by gpq_stream_wait only -- the host never blocks until the end.  One context, ONE compute stream. */
    void *cmp, *dn;
    CHECK(gpq_stream_create(&cmp));
    CHECK(gpq_stream_create(&dn));
    for (unsigned k0 = 0; k0 < cnt; k0 += h->pipe) {
      const unsigned m = cnt - k0 < h->pipe ? cnt - k0 : h->pipe;
      CHECK(gpq_stream_wait(cmp, h->stream));               /* (uploads of this sub-batch: here all of them were queued before the loop) */
      CHECK(gpq_he_mul_tensor(h->ctx, h->d_out[0] + k0 * perA, h->d_out[1] + k0 * perA, h->d_out[2] + k0 * perA, h->d_in[0] + k0 * perA, h->d_in[1] + k0 * perA,
                              h->d_in[2] + k0 * perA, h->d_in[3] + k0 * perA, dimA, m, h->d_wsA, cmp));
      CHECK(gpq_keyswitch(h->ctx, h->d_out[3] + k0 * perB, h->d_out[4] + k0 * perB, h->d_x + k0 * perB, h->d_e[0], h->d_e[1], dimB, m, h->d_wsB, cmp));
      CHECK(gpq_stream_wait(dn, cmp));
      for (int i = 0; i < 5; i++) {
        const size_t per = i < 3 ? perA : perB;
        CHECK(gpq_download(h->h_out[i] + k0 * per, h->d_out[i] + k0 * per, m * per * 8, dn));
      }
    }
    CHECK(gpq_stream_sync(dn));
    CHECK(gpq_stream_sync(cmp));
    CHECK(gpq_stream_destroy(cmp));
    CHECK(gpq_stream_destroy(dn));
  }
  free(primes); free(host);
  h->done = 1;
  return NULL;
}

int main(int argc, char **argv)
{
  if (argc < 6) return 2;
  const unsigned logn = (unsigned)atoi(argv[1]), dimA = (unsigned)atoi(argv[2]), dimB = (unsigned)atoi(argv[3]), batch = (unsigned)atoi(argv[4]);
  const size_t n = (size_t)1 << logn, perA = dimA * n, perB = dimB * n;
  int devs[64], shards = 0;
  for (char *tok = strtok(argv[5], ","); tok && shards < 64; tok = strtok(NULL, ",")) devs[shards++] = atoi(tok);
  if (!shards || batch < (unsigned)shards || dimA > dimB) return 2;
  const unsigned period = argc > 6 ? (unsigned)atoi(argv[6]) : 0;
  const unsigned pipe = argc > 7 ? (unsigned)atoi(argv[7]) : 0;
  printf("devices visible %d, shards %d\n", gpq_device_count(), shards);
  for (int s = 0; s < shards; s++)
    if (devs[s] < 0 || devs[s] >= gpq_device_count()) { fprintf(stderr, "device %d is not there\n", devs[s]); return 1; }

  struct shard *sh = calloc((size_t)(unsigned)shards, sizeof *sh);
  pthread_t *th = calloc((size_t)(unsigned)shards, sizeof *th);
  const unsigned base = batch / shards, extra = batch % shards;
  /* every shard's worker starts before any is waited for: the devices (and their PCIe links, and the sockets that generate the inputs) work concurrently */
  for (int s = 0; s < shards; s++) {
    struct shard *h = &sh[s];
    h->index = s; h->device = devs[s];
    h->lo = s * base + ((unsigned)s < extra ? (unsigned)s : extra);
    h->hi = h->lo + base + ((unsigned)s < extra ? 1 : 0);
    h->logn = logn; h->dimA = dimA; h->dimB = dimB; h->period = period; h->pipe = pipe;
    if (pthread_create(&th[s], NULL, run_shard, h) != 0) { fprintf(stderr, "pthread_create failed\n"); return 1; }
  }
  int rc = 0;
  for (int s = 0; s < shards; s++) pthread_join(th[s], NULL);
  for (int s = 0; s < shards; s++)
    if (!sh[s].done) { fprintf(stderr, "shard %d on device %d: %s\n", s, sh[s].device, sh[s].err); rc = 1; }
  if (rc) return rc;
  /* gather: every shard's ciphertexts in batch order */
  for (int s = 0; s < shards; s++) {
    struct shard *h = &sh[s];
    char cpus[256];
    const int local = gpq_device_local_cpus(h->device, NULL, cpus, sizeof cpus);
    printf("shard %d dev %d worker confined to %d cpus (device's node: %d cpus%s%s)\n", s, h->device, h->bound_cpus, local, local ? " " : "", cpus);
    if (gpq_set_device(h->device) != GPQ_OK) return 1;
    for (unsigned k = h->lo; k < h->hi; k++) {
      printf("ct %u dev %d", k, h->device);
      for (int i = 0; i < 5; i++) {
        const size_t per = i < 3 ? perA : perB;
        printf(" %016llx", (unsigned long long)fnv(h->h_out[i] + (k - h->lo) * per, per));
      }
      printf("\n");
    }
    for (int i = 0; i < 4; i++) gpq_free(h->d_in[i]);
    gpq_free(h->d_x); gpq_free(h->d_e[0]); gpq_free(h->d_e[1]); gpq_free(h->d_wsA); gpq_free(h->d_wsB);
    for (int i = 0; i < 5; i++) { gpq_free(h->d_out[i]); gpq_free_host(h->h_out[i]); }
    if (gpq_stream_destroy(h->stream) != GPQ_OK) return 1;
    gpq_ctx_destroy(h->ctx);
  }
  free(sh); free(th);
  return 0;
}
