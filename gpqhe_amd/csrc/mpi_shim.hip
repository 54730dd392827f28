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

namespace {

int g_dev = 0;             // device of the engine context: HIP's current device is per thread, the workers set it for their copies

// one engine context per (logn, chain length), checked against the caller's prime list
gpq_ctx *g_engine = nullptr;

gpq_ctx *engine() {
  if (&polyctx == nullptr || !polyctx.n) die("`polyctx` is not initialised (polyctx_init / hectx_init first)");
  if (g_engine && gpq_ctx_logn(g_engine) == polyctx.logn && gpq_ctx_nprimes(g_engine) >= polyctx.dimub) return g_engine;
  if (g_engine) gpq_ctx_destroy(g_engine);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) die("no HIP device");
  g_dev = dev;
  if (gpq_ctx_create(&g_engine, polyctx.logn, polyctx.dimub, dev) != GPQ_OK) die("cannot build the engine context");
  unsigned d = 0;
  for (const struct rns_ctx *r = polyctx.rns; r && d < polyctx.dimub; r = r->next, ++d)
    if (r->p != gpq_ctx_const(g_engine, d, 0)) die("the caller's prime chain differs from src/precomp.c:358-376");
  return g_engine;
}

// Device buffers of a call go back to a free list by size instead of hipFree: the same shapes come again with the next
// he_mul / he_rs, and hipMalloc + hipFree of a few hundred MB per call cost more than the kernels.  gpq_mpi_shim_release()
// gives the memory back.
std::map<size_t, std::vector<void *>> g_pool;

struct DevBuf {
  void *p = nullptr;
  size_t bytes;
  explicit DevBuf(size_t b) : bytes(b ? b : 8) {
    // the smallest kept buffer that is large enough (and not more than twice the request): slabs of 13 and 14 words share buffers
    for (auto it = g_pool.lower_bound(bytes); it != g_pool.end() && it->first <= 2 * bytes; ++it)
      if (!it->second.empty()) { p = it->second.back(); it->second.pop_back(); bytes = it->first; return; }
    if (gpq_malloc(&p, bytes) != GPQ_OK) die("device allocation failed");
  }
  ~DevBuf() { g_pool[bytes].push_back(p); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  uint64_t *u64() const { return (uint64_t *)p; }
};

void up(const DevBuf &d, const std::vector<uint64_t> &h) { if (gpq_upload(d.p, h.data(), h.size() * 8, nullptr) != GPQ_OK) die("upload failed"); }
void down(std::vector<uint64_t> &h, const DevBuf &d) {
  if (gpq_download(h.data(), d.p, h.size() * 8, nullptr) != GPQ_OK || gpq_stream_sync(nullptr) != GPQ_OK) die("download failed");
}

// Page-locked staging memory for the big slabs, kept by size like the device buffers: copies from / to it are true DMA and run
// while the host threads convert the next range.
std::map<size_t, std::vector<void *>> g_pinned;
struct HostBuf {
  void *p = nullptr;
  size_t bytes;
  explicit HostBuf(size_t b) : bytes(b ? b : 8) {
    for (auto it = g_pinned.lower_bound(bytes); it != g_pinned.end() && it->first <= 2 * bytes; ++it)
      if (!it->second.empty()) { p = it->second.back(); it->second.pop_back(); bytes = it->first; return; }
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) die("page-locked allocation failed");
  }
  ~HostBuf() { g_pinned[bytes].push_back(p); }
  HostBuf(const HostBuf &) = delete;
  HostBuf &operator=(const HostBuf &) = delete;
  uint64_t *u64() const { return (uint64_t *)p; }
};
std::vector<hipEvent_t> g_events;
hipEvent_t event_at(size_t i) {
  while (g_events.size() <= i) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) die("hipEventCreate failed");
    g_events.push_back(e);
  }
  return g_events[i];
}

// Host <-> device staging is in ROWS (W words per coefficient, coefficient after coefficient): a host thread fills or reads its range as one
// sequential stream and the range is one contiguous piece of memory for the DMA; the device turns rows into the kernels' word-major slabs
// and back (gpq_big_transpose, a few microseconds per polynomial).  Polynomials with fewer than 64 coefficients keep the word-major staging.
inline bool staged_in_rows(unsigned n) { return n >= 64; }
// coefficients [lo, hi) of a staged polynomial
void copy_range(void *dst, const void *src, unsigned n, unsigned W, unsigned lo, unsigned hi, hipMemcpyKind kind) {
  if (staged_in_rows(n)) {
    const size_t off = (size_t)lo * W * 8, bytes = (size_t)(hi - lo) * W * 8;
    if (hipMemcpyAsync((char *)dst + off, (const char *)src + off, bytes, kind, nullptr) != hipSuccess) die("slab copy failed");
    return;
  }
  const size_t pitch = (size_t)n * 8;                       // word-major: words [lo, hi) of every one of the W rows, one strided DMA
  if (hipMemcpy2DAsync((char *)dst + (size_t)lo * 8, pitch, (const char *)src + (size_t)lo * 8, pitch, (size_t)(hi - lo) * 8, W, kind, nullptr) != hipSuccess)
    die("slab copy failed");
}

// words [lo, hi) of one polynomial: four multiply-xor lanes (the multiply's latency is covered, the loop runs at memory speed)
uint64_t hash_words(const uint64_t *a, size_t lo, size_t hi) {
  uint64_t h[4] = {0x9e3779b97f4a7c15ull, 0xbf58476d1ce4e5b9ull, 0x94d049bb133111ebull, 0xcbf29ce484222325ull};
  size_t i = lo;
  for (; i + 4 <= hi; i += 4)
    for (int j = 0; j < 4; ++j) { h[j] = (h[j] ^ a[i + j]) * 0xff51afd7ed558ccdull; h[j] ^= h[j] >> 32; }
  for (; i < hi; ++i) { h[0] = (h[0] ^ a[i]) * 0xff51afd7ed558ccdull; h[0] ^= h[0] >> 32; }
  return ((h[0] * 3 + h[1]) * 5 + h[2]) * 7 + h[3];
}
// host threads that convert a polynomial of n coefficients (each its own range), and so the pieces its fingerprint is made of
inline unsigned convert_threads(unsigned n) { return n >= 4096 ? workers().width() : 1; }

// Transfers go in few, large pieces: a copy of a sixteenth of a polynomial (448 KB) moves at 25 GB/s over PCIe here, a whole polynomial
// (7 MB) at 53 GB/s -- about 9 us of fixed cost per copy (tools/copy_probe.hip) -- so the ranges the host threads convert are grouped
// four to a copy: coarse enough for the link, fine enough for conversions and DMA to overlap.
constexpr unsigned kRangesPerCopy = 4;

// MPI polynomials -> device big slabs: every host thread converts its range of a polynomial into page-locked memory; the thread that
// finishes the last range of a group of kRangesPerCopy sends the group off, so conversion of the next ranges / polynomial overlaps the DMA.
// `extra` more tasks (side(0) .. side(extra - 1)) are handed out to the same threads behind the ranges -- the evaluation-key
// fingerprint of a key that is not resident yet, which only reads memory while the conversions compute.
// `prints` (count x convert_threads(n) words, rows staging only): the fingerprint of every converted range, for the resident polynomials below.
void upload_polys(const DevBuf *const dst[], const HostBuf *const stage[], const poly_mpi_t *const src[], int count, unsigned n, unsigned W,
                  unsigned extra = 0, const std::function<void(unsigned)> *side = nullptr, uint64_t *prints = nullptr) {
  if (W < 1 || W > 32) die("coefficients wider than 2047 bits");
  const bool rows = staged_in_rows(n);
  const size_t big = (size_t)W * n * 8;
  DevBuf landing(rows ? big * count : 8);                   // the rows land here; gpq_big_transpose writes the word-major slabs from it
  const unsigned nt = convert_threads(n), per = (n + nt - 1) / nt;
  const unsigned ranges = (n + per - 1) / per, groups = (ranges + kRangesPerCopy - 1) / kRangesPerCopy;
  std::vector<std::atomic<unsigned>> done((size_t)count * groups);
  for (auto &d : done) d.store(0, std::memory_order_relaxed);
  const std::function<void(unsigned)> job = [&](unsigned t) {
    if (t >= nt) { (*side)(t - nt); return; }
    const unsigned lo = t * per, hi = lo + per < n ? lo + per : n;
    if (lo >= hi) return;
    (void)hipSetDevice(g_dev);
    const unsigned g = t / kRangesPerCopy, first = g * kRangesPerCopy, last = first + kRangesPerCopy < ranges ? first + kRangesPerCopy : ranges;
    for (int i = 0; i < count; ++i) {
      if (rows) to_slab_range<true>(stage[i]->u64(), src[i], n, W, lo, hi); else to_slab_range<false>(stage[i]->u64(), src[i], n, W, lo, hi);
      if (prints && rows) prints[(size_t)i * nt + t] = hash_words(stage[i]->u64(), (size_t)lo * W, (size_t)hi * W);   // just written: in cache
      if (done[(size_t)i * groups + g].fetch_add(1, std::memory_order_acq_rel) + 1 == last - first) {      // the group is complete: one copy for all of it
        const unsigned glo = first * per, ghi = last * per < n ? last * per : n;
        copy_range(rows ? (char *)landing.p + big * i : (char *)dst[i]->p, stage[i]->p, n, W, glo, ghi, hipMemcpyHostToDevice);
      }
    }
  };
  if (nt + extra < 2) job(0); else workers().run(nt + extra, job);
  if (rows)
    for (int i = 0; i < count; ++i)
      if (gpq_big_transpose(engine(), dst[i]->u64(), (const uint64_t *)((char *)landing.p + big * i), W, 1, 0, nullptr) != GPQ_OK) die("slab transpose failed");
}

// device big slabs -> the caller's MPIs: the device turns the slabs into rows, the ranges come back one DMA per group of kRangesPerCopy, in
// order, an event behind every one; a host thread converts its range as soon as its group has landed while the later ones are still in flight.
// Phase 1 (download_issue): the transposes, the copies and their events are queued behind the kernels.  Phase 2 (download_convert): the conversions.
// Between the two the host threads are free while the device works -- he_mul / he_rot / he_conj verify the evaluation key there.
void download_issue(const HostBuf *const stage[], const DevBuf *const src[], int count, unsigned n, unsigned W) {
  if (W < 1 || W > 64) die("big slab wider than 64 words");
  const bool rows = staged_in_rows(n);
  const size_t big = (size_t)W * n * 8;
  DevBuf takeoff(rows ? big * count : 8);                   // stream-ordered: safe to hand back to the pool when this returns
  const unsigned nt = convert_threads(n), per = (n + nt - 1) / nt;
  const unsigned ranges = (n + per - 1) / per, groups = (ranges + kRangesPerCopy - 1) / kRangesPerCopy;
  if (rows)                                                 // all the transposes first: the DMA of the first polynomial then runs without a kernel queued between its copies and the next's
    for (int i = 0; i < count; ++i)
      if (gpq_big_transpose(engine(), (uint64_t *)((char *)takeoff.p + big * i), src[i]->u64(), W, 1, 1, nullptr) != GPQ_OK) die("slab transpose failed");
  for (int i = 0; i < count; ++i) {
    const char *from = rows ? (const char *)takeoff.p + big * i : (const char *)src[i]->p;
    for (unsigned g = 0; g < groups; ++g) {
      const unsigned lo = g * kRangesPerCopy * per, hi = (g + 1) * kRangesPerCopy * per < n ? (g + 1) * kRangesPerCopy * per : n;
      copy_range(stage[i]->p, from, n, W, lo, hi, hipMemcpyDeviceToHost);
      if (hipEventRecord(event_at((size_t)i * groups + g), nullptr) != hipSuccess) die("hipEventRecord failed");
    }
  }
}
void download_convert(poly_mpi_t *const dst[], const HostBuf *const stage[], int count, unsigned n, unsigned W, uint64_t *prints = nullptr) {
  const bool rows = staged_in_rows(n);
  const unsigned nt = convert_threads(n), per = (n + nt - 1) / nt;
  const unsigned ranges = (n + per - 1) / per, groups = (ranges + kRangesPerCopy - 1) / kRangesPerCopy;
  const std::function<void(unsigned)> job = [&](unsigned t) {
    (void)hipSetDevice(g_dev);
    const unsigned lo = t * per, hi = lo + per < n ? lo + per : n;
    for (int i = 0; i < count; ++i) {
      if (hipEventSynchronize(g_events[(size_t)i * groups + t / kRangesPerCopy]) != hipSuccess) die("download failed");
      if (rows) from_slab_range<true>(dst[i], stage[i]->u64(), n, W, lo, hi); else from_slab_range<false>(dst[i], stage[i]->u64(), n, W, lo, hi);
      if (prints && rows && lo < hi) prints[(size_t)i * nt + t] = hash_words(stage[i]->u64(), (size_t)lo * W, (size_t)hi * W);
    }
  };
  if (ranges < 2) job(0); else workers().run(ranges, job);
}

// Where the wall time of the last he_mul call went (gpq_mpi_shim_last_timing): conversions + uploads, kernels (HIP events),
// downloads + conversions, whole call.
double g_last_ms[4] = {0, 0, 0, 0};
hipEvent_t g_tick[2] = {nullptr, nullptr};
double wall_ms() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

// Evaluation keys are 2 x dim x n words (47 MB at the headline shape) and the same key multiplies many ciphertexts: the device
// copy is kept, identified by the caller's two pointers and a fingerprint of EVERY word, limb by limb (the reference reads the key
// it is given on every call, src/he-mult.c:60-64: a key edited in place, in however few words, must multiply as edited).  A call at a
// lower level reads fewer limbs of the same key (dimB shrinks with q_l, :51): the resident copy serves every prefix of itself, checked
// over just the limbs in use -- one copy per key, not one per level.  For a key that is resident the fingerprint is computed by the host
// threads WHILE the device works with the resident copy (they would wait for it otherwise: resident_key / key_still_valid; a mismatch
// repeats the device work with the fresh key); for a new key, next to the ciphertext conversions.  gpq_mpi_shim_set_key_check(0) goes
// back to ~1000 sampled words (of the exact length in use) for callers that never edit a key in place.
struct KeySlot { const uint64_t *h0, *h1; unsigned limbs, n; bool full; std::vector<uint64_t> print; void *d0, *d1; size_t cap_words; uint64_t used; };
std::vector<KeySlot> g_keys;
uint64_t g_key_clock = 0;
size_t g_key_slots = 64;      // resident keys (rlk, ck, the rotation keys in use): 45 MiB each at the headline shape (2.9 GB of the 288 GB when all are in use); gpq_mpi_shim_set_key_slots
bool g_key_check_full = true; // gpq_mpi_shim_set_key_check
uint64_t key_print_sampled(const uint64_t *a, const uint64_t *b, size_t words) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; h ^= h >> 29; };
  const size_t step = words > 512 ? words / 509 : 1;     // ~512 samples of each polynomial, plus both ends
  for (size_t i = 0; i < words; i += step) { mix(a[i]); mix(b[i]); }
  for (size_t i = 0; i < 8 && i < words; ++i) { mix(a[i]); mix(b[i]); mix(a[words - 1 - i]); mix(b[words - 1 - i]); }
  return h;
}
struct KeyPrint {                 // fingerprint of the first `limbs` limbs of a host key: one piece per limb, any thread may compute any piece
  const uint64_t *h0, *h1; unsigned limbs, n; bool full; unsigned parts; std::vector<uint64_t> part;
  std::function<void(unsigned)> task;
  KeyPrint(const he_evk_t *key, unsigned limbs_, unsigned n_) : h0(key->p0.coeffs), h1(key->p1.coeffs), limbs(limbs_), n(n_), full(g_key_check_full) {
    parts = full ? limbs : 1;
    part.assign(parts, 0);
    task = [this](unsigned t) {
      if (!full) { part[t] = key_print_sampled(h0, h1, (size_t)limbs * n); return; }
      const size_t lo = (size_t)t * n, hi = lo + n;
      part[t] = hash_words(h0, lo, hi) * 0x100000001b3ull + hash_words(h1, lo, hi);
    };
  }
  size_t words() const { return (size_t)limbs * n; }
  bool covered_by(const KeySlot &k) const {                 // the slot's shape can serve this request at all
    return k.h0 == h0 && k.h1 == h1 && k.n == n && k.full == full && (full ? k.limbs >= limbs : k.limbs == limbs);
  }
  bool matches(const KeySlot &k) const {                    // ... and (part[] computed) holds the words the caller holds now
    return covered_by(k) && std::equal(part.begin(), part.end(), k.print.begin());
  }
};
void drop_key_slot(size_t i) {
  (void)gpq_stream_sync(nullptr);
  (void)gpq_free(g_keys[i].d0); (void)gpq_free(g_keys[i].d1);
  g_keys.erase(g_keys.begin() + i);
}
void forget_key_at(const uint64_t *h0, const uint64_t *h1) {          // the host key at these addresses was rewritten (he_gen*k)
  for (size_t i = g_keys.size(); i-- > 0;)
    if (g_keys[i].h0 == h0 || g_keys[i].h1 == h1 || g_keys[i].h0 == h1 || g_keys[i].h1 == h0) drop_key_slot(i);
}
// A resident copy of the host key at these addresses that covers the limbs in use, whatever its fingerprint: he_mul / he_rot / he_conj
// start the device work with it at once and verify the fingerprint on the host threads WHILE the device works (they would otherwise
// wait for it); a mismatch -- the key was edited in place since -- uploads the key and runs the device work again.
KeySlot *resident_key(const KeyPrint &kp) {
  for (KeySlot &k : g_keys)
    if (kp.covered_by(k)) { k.used = ++g_key_clock; return &k; }
  return nullptr;
}
bool key_still_valid(KeySlot *slot, KeyPrint &kp) {
  if (kp.parts < 2) kp.task(0); else workers().run(kp.parts, kp.task);
  return kp.matches(*slot);
}

void key_on_device(const KeyPrint &kp, uint64_t **d0, uint64_t **d1) {
  const uint64_t *h0 = kp.h0, *h1 = kp.h1;
  const size_t words = kp.words();
  for (KeySlot &k : g_keys)
    if (kp.matches(k)) { k.used = ++g_key_clock; *d0 = (uint64_t *)k.d0; *d1 = (uint64_t *)k.d1; return; }
  KeySlot slot{h0, h1, kp.limbs, kp.n, kp.full, kp.part, nullptr, nullptr, words, ++g_key_clock};
  size_t victim = g_keys.size();
  for (size_t i = 0; i < g_keys.size(); ++i)               // same host key in another state / at another length, else the least recently used
    if (g_keys[i].h0 == h0 && g_keys[i].h1 == h1) victim = i;
  if (victim == g_keys.size() && g_keys.size() >= g_key_slots) {
    victim = 0;
    for (size_t i = 1; i < g_keys.size(); ++i) if (g_keys[i].used < g_keys[victim].used) victim = i;
  }
  if (victim < g_keys.size()) {
    if (g_keys[victim].cap_words >= words) { slot.d0 = g_keys[victim].d0; slot.d1 = g_keys[victim].d1; slot.cap_words = g_keys[victim].cap_words; g_keys.erase(g_keys.begin() + victim); }
    else drop_key_slot(victim);
  }
  if (!slot.d0 && (gpq_malloc(&slot.d0, words * 8) != GPQ_OK || gpq_malloc(&slot.d1, words * 8) != GPQ_OK)) die("device allocation failed");
  if (gpq_upload(slot.d0, h0, words * 8, nullptr) != GPQ_OK || gpq_upload(slot.d1, h1, words * 8, nullptr) != GPQ_OK) die("upload failed");
  g_keys.push_back(slot);
  *d0 = (uint64_t *)slot.d0; *d1 = (uint64_t *)slot.d1;
}

// ---- resident polynomials ----------------------------------------------------------------------------------------------
// GPQHE's own callers chain the calls on one ciphertext (he_mul(&bn, &bn, &bn, rlk); he_rs(&bn); ... src/he-algo.c:140-160): what one call
// writes into the caller's integers is what the next one reads back.  Reading 2 x 65536 scattered libgcrypt integers and sending them
// over PCIe is most of a call (0.7 of 1.9 ms for he_mul), so the device keeps the word-major slab of every polynomial it has seen or
// produced, identified by the caller's coefficient array, the shape and a fingerprint of EVERY word of every coefficient.  When all
// operands of a call (and its key) are resident, the device starts from the resident copies at once and the host threads meanwhile do
// exactly the conversion an upload would have done, into the staging rows, and fingerprint it: a mismatch -- the caller changed the
// integers since, by any means -- uploads the rows that are then already staged and repeats the device work.  The result can never
// depend on a stale copy; a program whose operands are always new pays the fingerprint of its results (in cache, a few per cent).
// Only with the direct integer access (mpi_convert.hpp) and n >= 4096; gpq_mpi_shim_set_poly_slots(0) turns it off.
// `trusted`: a polynomial the caller changed behind the library's back last time (he_add and friends run on the host) is converted and
// uploaded before the device starts next time, as if unknown -- a wrong guess costs a repeated device pass -- until a call finds it unchanged.
struct PolySlot { const gpq_MPI *coeffs; unsigned n, W, parts; uint64_t print; void *d; size_t bytes /* of the buffer: >= W n 8 */; uint64_t used; bool trusted; };
std::vector<PolySlot> g_polys;
uint64_t g_poly_clock = 0;
size_t g_poly_slots = 32;     // 7 MiB each at n = 2^16, 14 words
uint64_t g_poly_hits = 0, g_poly_stale = 0;
std::vector<const void *> g_poly_pinned;   // resident slabs the MPI-typed call in progress reads from: not to be evicted or reused before it returns

bool g_poly_bypass = false;   // gpq_mpi_shim_poly_bypass: calls neither consult nor update the resident polynomials (they stay as they are)
bool poly_cache_on(unsigned n) { return g_poly_slots && !g_poly_bypass && n >= 4096 && staged_in_rows(n) && mpi_direct(); }
uint64_t fold_prints(const uint64_t *part, unsigned parts, unsigned n, unsigned W) {
  uint64_t h = 0xcbf29ce484222325ull ^ ((uint64_t)n << 32 | W);
  for (unsigned t = 0; t < parts; ++t) { h = (h ^ part[t]) * 0x100000001b3ull; h ^= h >> 29; }
  return h;
}
void drop_poly_slot(size_t i) {
  (void)gpq_stream_sync(nullptr);
  (void)gpq_free(g_polys[i].d);
  g_polys.erase(g_polys.begin() + i);
}
// the resident copy of this polynomial at this shape (W = 0: at whatever width it was kept), or null
const PolySlot *resident_poly(const poly_mpi_t *p, unsigned n, unsigned W) {
  for (PolySlot &s : g_polys)
    if (s.coeffs == p->coeffs && s.n == n && (W == 0 || s.W == W) && s.parts == convert_threads(n)) { s.used = ++g_poly_clock; return &s; }
  return nullptr;
}
// `dev` (word-major, W x n) is what the caller's polynomial holds now: keep a copy (stream-ordered device-to-device copy)
void remember_poly(const poly_mpi_t *p, unsigned n, unsigned W, uint64_t print, const void *dev, bool ours = false) {
  if (!poly_cache_on(n)) return;
  const size_t bytes = (size_t)W * n * 8;
  PolySlot slot{p->coeffs, n, W, convert_threads(n), print, nullptr, bytes, ++g_poly_clock, true};
  size_t victim = g_polys.size();
  for (size_t i = 0; i < g_polys.size(); ++i) if (g_polys[i].coeffs == p->coeffs) victim = i;     // one copy per host polynomial
  if (victim < g_polys.size() && !ours)                     // read from the caller: trusted unless it differs from what was kept for it
    slot.trusted = g_polys[victim].n == n && g_polys[victim].W == W && g_polys[victim].print == print;
  if (victim == g_polys.size() && g_polys.size() >= g_poly_slots) {
    // the least recently used goes -- but never a copy the call in progress is reading from (Operands::kept): with few slots the
    // operands of one call can be all there is, and then the newcomer is simply not kept
    auto pinned = [](const void *d) { for (const void *q : g_poly_pinned) if (q == d) return true; return false; };
    for (size_t i = 0; i < g_polys.size(); ++i)
      if (!pinned(g_polys[i].d) && (victim == g_polys.size() || g_polys[i].used < g_polys[victim].used)) victim = i;
    if (victim == g_polys.size()) return;
  }
  if (victim < g_polys.size()) {                            // its buffer serves again if it is large enough (he_rs keeps one word less than he_mul: no free / malloc per call)
    if (g_polys[victim].bytes >= bytes) { slot.d = g_polys[victim].d; slot.bytes = g_polys[victim].bytes; g_polys.erase(g_polys.begin() + victim); }
    else drop_poly_slot(victim);
  }
  if (!slot.d && gpq_malloc(&slot.d, bytes) != GPQ_OK) die("device allocation failed");
  if (slot.d != dev && hipMemcpyAsync(slot.d, dev, (size_t)W * n * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) die("device copy failed");
  g_polys.push_back(slot);
}

// The polynomial operands of one MPI-typed call.  prepare(): an operand with a trusted resident copy is taken from it (x[i] = the kept slab,
// nothing converted yet); the others are converted and uploaded as ever (x[i] = the call's own buffers) and remembered.  After the device
// work is queued, recheck() converts and fingerprints the caller's integers of the operands that were taken from their copies; those that
// differ are uploaded from the rows just staged, x[i] moves to the call's own buffer, and the caller queues the device work again.
struct Operands {
  int count; unsigned n, W, nt; bool cache, resident = false; unsigned kept = 0 /* operands served from their resident copies */, misfits = 0;
  const poly_mpi_t *const *src; const DevBuf *const *dst; const HostBuf *const *stage;
  const uint64_t *x[4]; uint64_t want[4];
  std::vector<uint64_t> prints;
  Operands(int count_, const poly_mpi_t *const s[], const DevBuf *const d[], const HostBuf *const st[], unsigned n_, unsigned W_)
      : count(count_), n(n_), W(W_), nt(convert_threads(n_)), cache(poly_cache_on(n_)), src(s), dst(d), stage(st), prints((size_t)4 * nt, 0) {
    if (count > 4) die("more than four polynomial operands");
  }
  ~Operands() { g_poly_pinned.clear(); }                    // one call at a time (SHIM_CALL), one Operands per call
  Operands(const Operands &) = delete;
  Operands &operator=(const Operands &) = delete;
  void prepare(bool may_speculate, unsigned extra = 0, const std::function<void(unsigned)> *side = nullptr) {
    kept = 0;
    if (cache && may_speculate)
      for (int i = 0; i < count; ++i) {
        const PolySlot *s = resident_poly(src[i], n, W);
        if (s && s->trusted) { x[i] = (const uint64_t *)s->d; want[i] = s->print; kept |= 1u << i; g_poly_pinned.push_back(s->d); }
      }
    resident = kept != 0;
    // the operands without a (trusted) resident copy are converted and uploaded before the device starts, as ever
    const poly_mpi_t *usrc[4]; const DevBuf *udst[4]; const HostBuf *ustage[4]; int idx[4], m = 0;
    for (int i = 0; i < count; ++i)
      if (!(kept >> i & 1)) { usrc[m] = src[i]; udst[m] = dst[i]; ustage[m] = stage[i]; idx[m++] = i; }
    if (!m) {
      if (extra) workers().run(extra, *side);
      return;
    }
    std::vector<uint64_t> uprints((size_t)m * nt, 0);
    upload_polys(udst, ustage, usrc, m, n, W, extra, side, cache ? uprints.data() : nullptr);
    for (int k = 0; k < m; ++k) {
      const int i = idx[k];
      x[i] = dst[i]->u64();
      if (cache) remember_poly(src[i], n, W, fold_prints(&uprints[(size_t)k * nt], nt, n, W), dst[i]->p);
    }
  }
  bool recheck(unsigned extra = 0, const std::function<void(unsigned)> *side = nullptr) {
    const unsigned per = (n + nt - 1) / nt;
    std::atomic<unsigned> misfit{0};
    const std::function<void(unsigned)> job = [&](unsigned t) {
      if (t >= nt) { (*side)(t - nt); return; }
      const unsigned lo = t * per, hi = lo + per < n ? lo + per : n;
      if (lo >= hi) return;
      for (int i = 0; i < count; ++i) {
        if (!(kept >> i & 1)) continue;
        bool bad = false;
        to_slab_range<true>(stage[i]->u64(), src[i], n, W, lo, hi, &bad);
        if (bad) misfit.fetch_or(1u << i, std::memory_order_relaxed);
        prints[(size_t)i * nt + t] = hash_words(stage[i]->u64(), (size_t)lo * W, (size_t)hi * W);
      }
    };
    if (nt + extra < 2) job(0); else workers().run(nt + extra, job);
    bool again = false;
    for (int i = 0; i < count; ++i) {
      if (!(kept >> i & 1)) continue;
      ++g_poly_hits;
      const uint64_t now = fold_prints(&prints[(size_t)i * nt], nt, n, W);
      const bool unfit = misfit.load() >> i & 1;
      if (now == want[i] && !unfit) continue;
      ++g_poly_stale; --g_poly_hits;
      again = true;
      if (unfit) { misfits |= 1u << i; continue; }           // wider than the kept copy: the caller decides (he_mul ends the program, he_rs widens)
      DevBuf landing((size_t)W * n * 8);                     // the caller's integers as they are now: rows already staged by the check
      copy_range(landing.p, stage[i]->p, n, W, 0, n, hipMemcpyHostToDevice);
      if (gpq_big_transpose(engine(), dst[i]->u64(), (const uint64_t *)landing.p, W, 1, 0, nullptr) != GPQ_OK) die("slab transpose failed");
      x[i] = dst[i]->u64();
      remember_poly(src[i], n, W, now, dst[i]->p);
    }
    return again;
  }
};
// the results of a call went into the caller's integers (download_convert with prints): keep the device slabs they came from
void remember_results(poly_mpi_t *const out[], const DevBuf *const dev[], int count, unsigned n, unsigned W, const std::vector<uint64_t> &prints) {
  if (!poly_cache_on(n)) return;
  const unsigned nt = convert_threads(n);
  for (int i = 0; i < count; ++i) remember_poly(out[i], n, W, fold_prints(&prints[(size_t)i * nt], nt, n, W), dev[i]->p, true);
}

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
      if (ops.misfits) return false;
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
      if (ops.misfits) return false;
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
      if (ops.misfits) return false;
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

// ---- src/he-add.c:32-140: he_add, he_sub, he_addpt, he_subpt, he_neg -----------------------------------------------------------
// Big-integer work only (mpi_addm / mpi_subm / mpi_neg, then mpi_smod), no RNS -- but GPQHE's algorithms interleave these with every
// product (he_inv: he_addpt between two he_mul, src/he-algo.c:146-155), each is 2n libgcrypt calls on the host (tens of milliseconds at
// n = 2^16), and a ciphertext the host has added to is one the device copies no longer match.  On the device they are one carry chain
// per coefficient and the centring the rescale kernels already do, on operands that are resident after the call before.
// kind 0: ct = a + b; 1: ct = a - b; 2: ct = a + pt; 3: ct = a - pt; 4: ct = -a (in place); 5: ct = a exactly (he_copy_ct, src/he-mem.c:88-97:
// no reduction -- 2n mpi_set on the host otherwise, and a copy the device knows nothing about)
static void additive(he_ct_t *ct, const he_ct_t *a, const he_ct_t *b, const he_pt_t *pt, int kind) {
  SHIM_CALL();
  need_gcrypt();
  if (&hectx == nullptr || !hectx.q) die("`hectx` is not initialised (hectx_init first)");
  if (kind < 2 && a->l != b->l) die("he_add / he_sub: operands at different levels");              // assert at src/he-add.c:35, :59
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n, l = a->l;
  const double nu = kind >= 4 ? a->nu : kind < 2 ? (a->nu >= b->nu ? a->nu : b->nu) : (a->nu >= pt->nu ? a->nu : pt->nu);   // :37, :61, :84, :107
  const double B = kind >= 4 ? a->B : kind < 2 ? a->B + b->B : a->B;                                                     // :38, :62, :85, :108
  // a copy never looks at q_l (src/he-mem.c:88-97 copies whatever level the source claims)
  const std::vector<uint64_t> qw = kind == 5 ? std::vector<uint64_t>{2} : words_of(hectx.q[l], "he_add: q_l must be positive");
  const bool pow2 = is_pow2(qw);
  const unsigned nbq = kind == 5 ? 2 : G.mpi_get_nbits(hectx.q[l]), logql = nbq - 1;
  poly_mpi_t *out[2] = {&ct->c0, &ct->c1};
  const int count = kind >= 4 ? 2 : kind < 2 ? 4 : 3;
  const poly_mpi_t *in[4] = {&a->c0, &a->c1, kind < 2 ? &b->c0 : kind < 4 ? &pt->m : nullptr, kind < 2 ? &b->c1 : nullptr};
  if (logql == 0 && kind != 5) {                            // q_l = 1: mpi_smod leaves -1 everywhere (see he_rs)
    for (int k = 0; k < 2; ++k)
      for (unsigned i = 0; i < n; ++i) { G.mpi_set_ui(out[k]->coeffs[i], 1); G.mpi_neg(out[k]->coeffs[i], out[k]->coeffs[i]); }
    ct->l = l; ct->nu = nu; ct->B = B;
    return;
  }
  const unsigned Wout = kind == 5 ? 64 : logql / 64 + 1;    // the results are centred mod q_l (a copy keeps every word)
  auto pass = [&](unsigned W, bool kept) -> bool {
    if (W > 32) die("he_add: coefficients wider than 2047 bits");
    const size_t big = (size_t)W * n;
    HostBuf s0(big * 8), s1(big * 8), s2(count > 2 ? big * 8 : 8), s3(count > 3 ? big * 8 : 8), t0s(big * 8), t1s(big * 8);
    DevBuf d0(big * 8), d1(big * 8), d2(count > 2 ? big * 8 : 8), d3(count > 3 ? big * 8 : 8), o0(big * 8), o1(big * 8), scratch(192 * 8);
    const DevBuf *dd[4] = {&d0, &d1, &d2, &d3}, *oo[2] = {&o0, &o1};
    const HostBuf *ss[4] = {&s0, &s1, &s2, &s3}, *ts[2] = {&t0s, &t1s};
    Operands ops(count, in, dd, ss, n, W);
    ops.prepare(kept);
    auto device_work = [&]() {
      int rc;
      if (kind == 5) {
        if (hipMemcpyAsync(o0.p, ops.x[0], big * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess ||
            hipMemcpyAsync(o1.p, ops.x[1], big * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) die("device copy failed");
        download_issue(ts, oo, 2, n, W);
        return;
      }
      if (kind == 4) {
        rc = gpq_big_addsub(c, o0.u64(), ops.x[0], nullptr, W, 1, 2, nullptr);
        if (rc == GPQ_OK) rc = gpq_big_addsub(c, o1.u64(), ops.x[1], nullptr, W, 1, 2, nullptr);
      } else {
        rc = gpq_big_addsub(c, o0.u64(), ops.x[0], ops.x[2], W, 1, kind & 1, nullptr);              // c0 (+/-) the other c0 or the plaintext
        if (rc == GPQ_OK && kind < 2) rc = gpq_big_addsub(c, o1.u64(), ops.x[1], ops.x[3], W, 1, kind & 1, nullptr);
        else if (rc == GPQ_OK && hipMemcpyAsync(o1.p, ops.x[1], big * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) die("device copy failed");   // c1 mod q, :92, :115
      }
      if (rc == GPQ_OK)                                                                              // mpi_smod of both, :43-44 ...
        rc = pow2 ? gpq_he_rs(c, o0.u64(), o1.u64(), W, 0, logql, 1, nullptr)
                  : gpq_he_rs_general(c, o0.u64(), o1.u64(), W, 1ull, qw.data(), (unsigned)qw.size(), 1, scratch.p, nullptr);
      if (rc != GPQ_OK) die("he_add failed");
      download_issue(ts, oo, 2, n, Wout < W ? Wout : W);
    };
    device_work();
    if (ops.resident && ops.recheck()) {
      if (ops.misfits) return false;
      device_work();
    }
    const unsigned Wdown = Wout < W ? Wout : W;
    std::vector<uint64_t> oprints((size_t)2 * ops.nt, 0);
    download_convert(out, ts, 2, n, Wdown, oprints.data());
    remember_results(out, oo, 2, n, Wdown, oprints);
    return true;
  };
  bool done = false;
  if ((pow2 || kind == 5) && poly_cache_on(n)) {            // wrapping sums are harmless below a power of two; any other q_l measures its operands first
    unsigned W = 0;
    bool all = true;
    for (int i = 0; i < count && all; ++i) {
      const PolySlot *k = resident_poly(in[i], n, 0);
      if (!k || !k->trusted || (W && k->W != W)) all = false; else W = k->W;
    }
    if (all && (W >= Wout || kind == 5)) done = pass(W, true);
  }
  if (!done) {
    unsigned bits = nbq;
    for (int i = 0; i < count; ++i) { const unsigned bi = max_bits(in[i], n); if (bi > bits) bits = bi; }
    pass((bits + 1) / 64 + 1, false);                       // one bit of headroom: the sum of two such integers still fits
  }
  ct->l = l; ct->nu = nu; ct->B = B;
}
void he_add(he_ct_t *ct, const he_ct_t *ct1, const he_ct_t *ct2) { additive(ct, ct1, ct2, nullptr, 0); }
void he_sub(he_ct_t *ct, const he_ct_t *ct1, const he_ct_t *ct2) { additive(ct, ct1, ct2, nullptr, 1); }
void he_addpt(he_ct_t *dest, const he_ct_t *src, const he_pt_t *pt) { additive(dest, src, nullptr, pt, 2); }
void he_subpt(he_ct_t *dest, const he_ct_t *src, const he_pt_t *pt) { additive(dest, src, nullptr, pt, 3); }
void he_neg(he_ct_t *ct) { additive(ct, ct, nullptr, nullptr, 4); }
void he_copy_ct(he_ct_t *dest, const he_ct_t *src) { if (dest != src) additive(dest, src, nullptr, nullptr, 5); }      // src/he-mem.c:88-97

// he_rot / he_conj, src/he-automorphism.c:87-115: permute both polynomials, then he_swk (:40-85) in place
static void automorphism(he_ct_t *ct, const he_evk_t *key, bool conj, unsigned rot) {
  SHIM_CALL();
  need_gcrypt();
  if (&hectx == nullptr || !hectx.q) die("`hectx` is not initialised (hectx_init first)");
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n, l = ct->l;
  const std::vector<uint64_t> qw = words_of(hectx.q[l], "he_rot/he_conj: q_l must be positive");
  const bool pow2 = is_pow2(qw);
  const unsigned nbq = G.mpi_get_nbits(hectx.q[l]), logql = nbq - 1, nbPqL = G.mpi_get_nbits(hectx.PqL);
  const unsigned dimB = (nbq + nbPqL + polyctx.logn) / 59 + 1, dimP = hectx.dim;                // src/he-automorphism.c:52
  const unsigned W = logql / 64 + 1;
  const size_t big = (size_t)W * n;
  HostBuf s0(big * 8), s1(big * 8), t0s(big * 8), t1s(big * 8);
  DevBuf a0(big * 8), a1(big * 8), r0(big * 8), r1(big * 8), o0(big * 8), o1(big * 8),
      ws(pow2 ? gpq_he_swk_workspace_bytes(c, W, dimB, dimP, 1) : gpq_he_general_workspace_bytes(c, W, 0, dimB, dimP, 1));
  const DevBuf *dd[2] = {&a0, &a1}, *oo[2] = {&o0, &o1};
  const HostBuf *ss[2] = {&s0, &s1}, *ts[2] = {&t0s, &t1s};
  const poly_mpi_t *in[2] = {&ct->c0, &ct->c1};
  KeyPrint kp(key, dimB, n);
  KeySlot *spec = resident_key(kp);
  Operands ops(2, in, dd, ss, n, W);
  if (spec) ops.prepare(true); else ops.prepare(false, kp.parts, &kp.task);
  uint64_t *k0, *k1;
  if (spec) { k0 = (uint64_t *)spec->d0; k1 = (uint64_t *)spec->d1; } else key_on_device(kp, &k0, &k1);
  auto device_work = [&]() {
    int rc = conj ? gpq_poly_conj(c, r0.u64(), ops.x[0], W, 1, nullptr) : gpq_poly_rot(c, r0.u64(), ops.x[0], W, rot, 1, nullptr);      // :95-96 / :108-109
    if (rc == GPQ_OK) rc = conj ? gpq_poly_conj(c, r1.u64(), ops.x[1], W, 1, nullptr) : gpq_poly_rot(c, r1.u64(), ops.x[1], W, rot, 1, nullptr);
    if (rc == GPQ_OK)                                                                                                                       // :97 / :110
      rc = pow2 ? gpq_he_swk(c, o0.u64(), o1.u64(), r0.u64(), r1.u64(), k0, k1, W, logql, dimB, dimP, 1, ws.p, nullptr)
                : gpq_he_swk_general(c, o0.u64(), o1.u64(), r0.u64(), r1.u64(), k0, k1, W, qw.data(), (unsigned)qw.size(), dimB, dimP, 1,
                                     ws.p, nullptr);
    if (rc != GPQ_OK) die("he_rot/he_conj failed");
    download_issue(ts, oo, 2, n, W);
  };
  device_work();
  bool again = false;
  if (ops.resident) {
    again = ops.recheck(kp.parts, &kp.task);
    if (ops.misfits) die("coefficient does not fit the big slab");
    if (!kp.matches(*spec)) { key_on_device(kp, &k0, &k1); again = true; }
  } else if (spec && !key_still_valid(spec, kp)) {
    key_on_device(kp, &k0, &k1);
    again = true;
  }
  if (again) device_work();
  poly_mpi_t *out[2] = {&ct->c0, &ct->c1};
  std::vector<uint64_t> oprints((size_t)2 * ops.nt, 0);
  download_convert(out, ts, 2, n, W, oprints.data());
  remember_results(out, oo, 2, n, W, oprints);
}
void he_conj(he_ct_t *ct, const he_evk_t *ck) { automorphism(ct, ck, true, 0); }
void he_rot(he_ct_t *ct, const int rot, const he_evk_t *rk) { automorphism(ct, &rk[rot], false, (unsigned)rot); }   // rk[rot], :110

// ---- key generation, src/he-kem.c:74-170 -----------------------------------------------------------------------------
// he_genswk (static in the reference, :74-118) with the hidden polynomial given as a host big slab of W words.  The reference's
// samplers are called in its order (error, then uniform mod P q_L), so a seeded RNG gives the reference's own keys.
static void genswk(he_evk_t *swk, const std::vector<uint64_t> &sp, const std::vector<uint64_t> &hs, unsigned W) {
  SHIM_CALL();
  forget_key_at(swk->p0.coeffs, swk->p1.coeffs);         // the host key is about to be rewritten: its device copy (if any) goes first
  if (!sample_error || !sample_uniform) die("he_gen*k: the host program does not provide sample_error / sample_uniform (src/sample.c)");
  gpq_ctx *c = engine();
  const unsigned n = polyctx.n;
  poly_mpi_t e, p1;
  e.coeffs = (gpq_MPI *)malloc(n * sizeof(gpq_MPI));
  p1.coeffs = (gpq_MPI *)malloc(n * sizeof(gpq_MPI));
  for (unsigned i = 0; i < n; ++i) { e.coeffs[i] = G.mpi_new(0); p1.coeffs[i] = G.mpi_new(0); }
  sample_error(&e);                                                                               // :87
  sample_uniform(&p1, hectx.PqL);                                                                 // :94
  const size_t big = (size_t)W * n, evk = (size_t)hectx.dimevk * n;
  std::vector<uint64_t> he(big), hp(big);
  to_slab(he.data(), &e, n, W);
  to_slab(hp.data(), &p1, n, W);
  for (unsigned i = 0; i < n; ++i) { G.mpi_release(e.coeffs[i]); G.mpi_release(p1.coeffs[i]); }
  free(e.coeffs); free(p1.coeffs);
  const unsigned logqL = G.mpi_get_nbits(hectx.q[hectx.L]) - 1;
  if (!is_pow2(words_of(hectx.q[hectx.L], "he_gen*k: q_L must be positive"))) die("he_gen*k: q_L must be a power of two on this path");
  DevBuf dp(big * 8), ds(big * 8), de(big * 8), dsp(big * 8), k0(evk * 8), k1(evk * 8), ws(gpq_he_genswk_workspace_bytes(c, W, hectx.dim, logqL));
  up(dp, hp); up(ds, hs); up(de, he); up(dsp, sp);
  if (gpq_he_genswk(c, k0.u64(), k1.u64(), dp.u64(), ds.u64(), de.u64(), dsp.u64(), W, hectx.dim, logqL, hectx.dimevk, ws.p, nullptr) != GPQ_OK)
    die("he_genswk failed");
  if (gpq_download(swk->p0.coeffs, k0.p, evk * 8, nullptr) != GPQ_OK || gpq_download(swk->p1.coeffs, k1.p, evk * 8, nullptr) != GPQ_OK ||
      gpq_stream_sync(nullptr) != GPQ_OK) die("download failed");
}

static unsigned keygen_words() {
  SHIM_CALL();
  need_gcrypt();
  if (&hectx == nullptr || !hectx.q) die("`hectx` is not initialised (hectx_init first)");
  return G.mpi_get_nbits(hectx.PqL) / 64 + 1;
}

// a permutation of the secret as the hidden polynomial: poly_conj / poly_rot (src/poly.c:263-283) on the device
static std::vector<uint64_t> permuted(const std::vector<uint64_t> &hs, unsigned W, bool conj, unsigned rot) {
  gpq_ctx *c = engine();
  const size_t big = (size_t)W * polyctx.n;
  DevBuf a(big * 8), r(big * 8);
  up(a, hs);
  const int rc = conj ? gpq_poly_conj(c, r.u64(), a.u64(), W, 1, nullptr) : gpq_poly_rot(c, r.u64(), a.u64(), W, rot, 1, nullptr);
  if (rc != GPQ_OK) die("poly_rot / poly_conj failed");
  std::vector<uint64_t> out(big);
  down(out, r);
  return out;
}

void he_genrlk(he_evk_t *rlk, const poly_mpi_t *sk) {                                              // :120-137
  SHIM_CALL();
  const unsigned W = keygen_words(), n = polyctx.n;
  gpq_ctx *c = engine();
  printf("Generating rlk ... ");
  fflush(stdout);
  std::vector<uint64_t> hs((size_t)W * n), s2((size_t)W * n);
  to_slab(hs.data(), sk, n, W);
  const unsigned nbq = G.mpi_get_nbits(hectx.q[hectx.L]), dim = nbq / 59 + 1;                      // :131
  const std::vector<uint64_t> qw = words_of(hectx.q[hectx.L], "he_genrlk: q_L must be positive");
  {
    DevBuf a(hs.size() * 8), r(hs.size() * 8), ws(gpq_poly_mul_general_workspace_bytes(c, dim, 1));
    up(a, hs);
    const int rc = is_pow2(qw) ? gpq_poly_mul(c, r.u64(), a.u64(), a.u64(), W, dim, nbq - 1, 1, ws.p, nullptr)
                               : gpq_poly_mul_general(c, r.u64(), a.u64(), a.u64(), W, dim, qw.data(), (unsigned)qw.size(), 1, ws.p, nullptr);
    if (rc != GPQ_OK) die("he_genrlk: poly_mul failed");
    down(s2, r);
  }
  genswk(rlk, s2, hs, W);                                                                          // :132
  printf("done.\n");
}

void he_genck(he_evk_t *ck, const poly_mpi_t *sk) {                                                // :140-154
  SHIM_CALL();
  const unsigned W = keygen_words(), n = polyctx.n;
  printf("Generating ck ... ");
  fflush(stdout);
  std::vector<uint64_t> hs((size_t)W * n);
  to_slab(hs.data(), sk, n, W);
  genswk(ck, permuted(hs, W, true, 0), hs, W);
  printf("done.\n");
}

void he_genrk(he_evk_t *rk, const poly_mpi_t *sk) {                                                // :156-170
  SHIM_CALL();
  const unsigned W = keygen_words(), n = polyctx.n;
  printf("Generating rk ... ");
  fflush(stdout);
  std::vector<uint64_t> hs((size_t)W * n);
  to_slab(hs.data(), sk, n, W);
  for (unsigned rot = 0; rot < hectx.slots; ++rot) genswk(&rk[rot], permuted(hs, W, false, rot), hs, W);
  printf("done.\n");
}

// wall milliseconds of the last he_mul(he_ct_t*, ...) call: [0] MPI -> slab conversions and uploads, [1] device kernels,
// [2] downloads and slab -> MPI conversions (includes waiting for [1]), [3] the whole call
void gpq_mpi_shim_last_timing(double ms[4]) { for (int i = 0; i < 4; ++i) ms[i] = g_last_ms[i]; }

// How many evaluation keys stay on the device between calls (default 64; he_rot over many rotation keys -- the gemv of
// src/he-algo.c:63-85 walks rk[0..slots) -- wants as many as it cycles through: 45 MiB each at n = 2^16, 45 limbs).
void gpq_mpi_shim_set_key_slots(unsigned slots) {
  SHIM_CALL();
  g_key_slots = slots ? slots : 1;
  while (g_keys.size() > g_key_slots) {                     // resident keys beyond the new limit go at once, least recently used first
    size_t victim = 0;
    for (size_t i = 1; i < g_keys.size(); ++i) if (g_keys[i].used < g_keys[victim].used) victim = i;
    drop_key_slot(victim);
  }
}
// 1 (default): a resident key is recognised by a fingerprint of every word, computed by the conversion threads beside the
// ciphertext conversions; 0: by ~1000 sampled words (for programs that never edit a key in place; gpq_mpi_shim_forget_keys covers the rest)
void gpq_mpi_shim_set_key_check(int full) { SHIM_CALL(); g_key_check_full = full != 0; }
unsigned gpq_mpi_shim_resident_keys(void) { SHIM_CALL(); return (unsigned)g_keys.size(); }
// 1 (default): libgcrypt integers are read and written limb by limb in place once the layout probe has passed (mpi_convert.hpp);
// 0: every coefficient goes through gcry_mpi_print / gcry_mpi_scan.  Returns whether the direct path is in use afterwards.
int gpq_mpi_shim_set_direct_mpi(int on) { SHIM_CALL(); need_gcrypt(); g_mpi_direct_wanted = on != 0; return mpi_direct() ? 1 : 0; }

// Drops the device copies of evaluation keys (he_mul / he_rot / he_conj keep up to gpq_mpi_shim_set_key_slots of them, recognised by the caller's pointers, the
// length and a fingerprint of every word -- of ~1000 sampled words after gpq_mpi_shim_set_key_check(0), and then a program that rewrites a
// key IN PLACE in a way the samples may miss must call this after the rewrite).  he_gen*k drop the slot of the key they write themselves.
void gpq_mpi_shim_forget_keys(void) {
  SHIM_CALL();
  (void)gpq_stream_sync(nullptr);
  for (KeySlot &k : g_keys) { (void)gpq_free(k.d0); (void)gpq_free(k.d1); }
  g_keys.clear();
}

// Resident polynomials (see PolySlot above): how many device copies of the caller's polynomials are kept between calls (default 32, 7 MiB each at
// n = 2^16 and 14 words; 0 = none: every call converts and uploads its operands before the device starts, as up to round 2).
void gpq_mpi_shim_set_poly_slots(unsigned slots) {
  SHIM_CALL();
  g_poly_slots = slots;
  while (g_polys.size() > g_poly_slots) {
    size_t victim = 0;
    for (size_t i = 1; i < g_polys.size(); ++i) if (g_polys[i].used < g_polys[victim].used) victim = i;
    drop_poly_slot(victim);
  }
}
// Host threads that convert between libgcrypt integers and slabs (default: the hardware threads, at most 16; up to 64).  Takes effect only
// before the first MPI-typed call of the process (the pool is started once); returns the number in use afterwards.
unsigned gpq_mpi_shim_set_conversion_threads(unsigned threads) { SHIM_CALL(); g_workers_wanted = threads; return workers().width(); }
unsigned gpq_mpi_shim_resident_polys(void) { SHIM_CALL(); return (unsigned)g_polys.size(); }
// operands served from a resident copy that the check confirmed / that the check found changed (uploaded again, device work repeated)
void gpq_mpi_shim_poly_stats(uint64_t *confirmed, uint64_t *stale) { SHIM_CALL(); if (confirmed) *confirmed = g_poly_hits; if (stale) *stale = g_poly_stale; }
// testing: while on, the MPI-typed calls convert and upload everything and remember nothing, without touching what is resident
void gpq_mpi_shim_poly_bypass(int on) { SHIM_CALL(); g_poly_bypass = on != 0; }
void gpq_mpi_shim_forget_polys(void) {
  SHIM_CALL();
  (void)gpq_stream_sync(nullptr);
  for (PolySlot &k : g_polys) (void)gpq_free(k.d);
  g_polys.clear();
}

// frees the device buffers the MPI-typed calls keep between calls, and the engine context
void gpq_mpi_shim_release(void) {
  SHIM_CALL();
  (void)gpq_stream_sync(nullptr);
  for (PolySlot &k : g_polys) (void)gpq_free(k.d);
  g_polys.clear();
  for (auto &kv : g_pool) for (void *q : kv.second) (void)gpq_free(q);
  g_pool.clear();
  for (auto &kv : g_pinned) for (void *q : kv.second) (void)hipHostFree(q);
  g_pinned.clear();
  for (KeySlot &k : g_keys) { (void)gpq_free(k.d0); (void)gpq_free(k.d1); }
  g_keys.clear();
  for (hipEvent_t e : g_events) (void)hipEventDestroy(e);
  g_events.clear();
  if (g_engine) { gpq_ctx_destroy(g_engine); g_engine = nullptr; }
}

void he_rs(struct he_ct *ct) { rescale_common(ct, true); }        // src/he-rescale.c:33-54
void he_rescale(struct he_ct *ct) { rescale_common(ct, true); }
void he_moddown(he_ct_t *ct) { rescale_common(ct, false); }       // src/he-rescale.c:56-70

}  // extern "C"

// ======================================================================================================================
// gpq_fill_rns_chain: the per-prime chain `struct rns_ctx` (src/poly.h:28-41) out of an engine context.
//
// The context-construction and storage names of the reference (polyctx_init/exit, hectx_init/exit, poly_mpi_alloc/free,
// poly_rns_alloc/free, the data symbols polyctx, hectx, GPQHE_TWO; src/precomp.c, src/poly.c:46-69) are NOT defined in this
// library: it only REFERENCES polyctx / hectx (weakly).  A GPQHE build keeps its own precomp.o / poly.o and any link order works
// (nothing here can shadow them); a host that is not GPQHE adds -lgpqhe_hip_ctx (ctx_compat.hip), which defines them on top of
// this library.  tests/test_link_order.py checks both orders and the dlopen case.
// ======================================================================================================================
namespace {

struct ChainOwner { struct rns_ctx *nodes; unsigned count; bool mpi; };
std::vector<ChainOwner> g_chains;        // what gpq_fill_rns_chain allocated, for gpq_release_rns_chain

}  // namespace

extern "C" {

// the engine context of the MPI-typed calls for the caller's `polyctx` (ctx_compat.hip builds the prime chain from it)
gpq_ctx *gpq_mpi_shim_engine(void) { SHIM_CALL(); return engine(); }
// addresses of `polyctx` / `hectx` as THIS library is bound to them (null when no object of the process defines them):
// what tests/test_link_order.py compares with the host's own view
const void *gpq_compat_view(const char *name) {
  if (!strcmp(name, "polyctx")) return (const void *)&polyctx;
  if (!strcmp(name, "hectx")) return (const void *)&hectx;
  return nullptr;
}

// Fills nodes[0..count) -- an array the caller owns -- like polyctx_init's loop does (src/precomp.c:359-380): node d
// describes the prefix of d + 1 primes; dim, p, pinv_mont, pinv_barr, ninv as src/precomp.c:246-248; zetas / zetas_inv
// point into the engine context's host tables (Montgomery form, bit-reversed: src/precomp.c:255-263; valid while `ctx`
// lives); phat_invmp as src/precomp.c:287-290 (malloc'ed here).  with_mpi != 0 also builds the libgcrypt integers
// P, P_2, phat[] (src/precomp.c:268-286) -- needs libgcrypt in the process; with 0 they stay NULL (the RNS-level symbols
// ntt / invntt / poly_rns_* never read them).  gpq_release_rns_chain frees what this call allocated.
int gpq_fill_rns_chain(struct rns_ctx *nodes, unsigned count, const gpq_ctx *ctx, int with_mpi) {
  if (!nodes || !ctx || count < 1 || count > gpq_ctx_nprimes(ctx)) return GPQ_ERR_INVALID;
  if (with_mpi) need_gcrypt();
  Words P;
  std::vector<uint64_t> primes(count);
  for (unsigned d = 0; d < count; ++d) {
    struct rns_ctx &r = nodes[d];
    primes[d] = gpq_ctx_const(ctx, d, 0);
    r.dim = d + 1;
    r.p = primes[d];
    r.pinv_mont = gpq_ctx_const(ctx, d, 1);
    r.pinv_barr = gpq_ctx_const(ctx, d, 2);
    r.ninv = gpq_ctx_const(ctx, d, 3);
    r.zetas = const_cast<uint64_t *>(gpq_ctx_zetas(ctx, d, 0));
    r.zetas_inv = const_cast<uint64_t *>(gpq_ctx_zetas(ctx, d, 1));
    r.next = d + 1 < count ? &nodes[d + 1] : nullptr;
    if (d == 0) P.assign(1, primes[0]); else mul_word(P, primes[d]);
    r.phat_invmp = (uint64_t *)malloc((size_t)(d + 1) * sizeof(uint64_t));
    r.P = r.P_2 = nullptr;
    r.phat = nullptr;
    if (with_mpi) {
      r.P = mpi_of(P);
      Words half = P;
      shr1(half);
      r.P_2 = mpi_of(half);
      r.phat = (gpq_MPI *)G.xmalloc((size_t)(d + 1) * sizeof(gpq_MPI));
    }
    for (unsigned k = 0; k <= d; ++k) {
      Words phat = P;
      (void)divmod_word(phat, primes[k]);                      // P / p_k, exact
      Words t = phat;
      const uint64_t res = divmod_word(t, primes[k]);          // (P / p_k) mod p_k
      r.phat_invmp[k] = powm64(res, primes[k] - 2, primes[k]);
      if (with_mpi) r.phat[k] = mpi_of(phat);
    }
  }
  g_chains.push_back(ChainOwner{nodes, count, with_mpi != 0});
  return GPQ_OK;
}

void gpq_release_rns_chain(struct rns_ctx *nodes) {
  for (size_t i = 0; i < g_chains.size(); ++i) {
    if (g_chains[i].nodes != nodes) continue;
    for (unsigned d = 0; d < g_chains[i].count; ++d) {
      struct rns_ctx &r = nodes[d];
      free(r.phat_invmp);
      if (g_chains[i].mpi) {
        G.mpi_release(r.P); G.mpi_release(r.P_2);
        for (unsigned k = 0; k <= d; ++k) G.mpi_release(r.phat[k]);
        G.xfree(r.phat);
      }
      r.phat_invmp = nullptr; r.P = r.P_2 = nullptr; r.phat = nullptr; r.zetas = r.zetas_inv = nullptr;
    }
    g_chains.erase(g_chains.begin() + i);
    return;
  }
}

}  // extern "C"
