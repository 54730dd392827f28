// shim_staging.hpp -- engine context, device / page-locked buffer pools, row staging, MPI <-> slab transfers with their fingerprints.
// Part of the MPI-typed surface: one translation unit (mpi_shim.hip includes these fragments in order); split by concern in round 4.
#pragma once

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

// Fingerprints are KEYED: the four lanes start from, and multiply by, words drawn from getrandom(2) once per process.  Every step
// (h ^ a) * M; h ^= h >> 32 is a bijection of the lane's state, so one changed word always changes its lane; cancelling changes across
// words or lanes needs the secret start words and multiplier.  A party that supplies ciphertexts at a reused buffer address but cannot
// read this process's memory finds a second preimage with probability about 2^-64 per attempt (with the fixed constants of round 3
// the last four words of a range could simply be solved for: ADVICE round 3).
struct HashKey { uint64_t init[4], mul, fold; };
inline const HashKey &hash_key() {
  static const HashKey key = [] {
    HashKey k{{0x9e3779b97f4a7c15ull, 0xbf58476d1ce4e5b9ull, 0x94d049bb133111ebull, 0xcbf29ce484222325ull}, 0xff51afd7ed558ccdull, 0x100000001b3ull};
    uint64_t r[6];
    size_t got = 0;
    while (got < sizeof r) {
      const ssize_t n = getrandom((char *)r + got, sizeof r - got, 0);
      if (n <= 0) break;
      got += (size_t)n;
    }
    if (got != sizeof r) {                                 // no entropy source: time, pid and an address (still not the published constants)
      timespec t;
      clock_gettime(CLOCK_REALTIME, &t);
      uint64_t x = (uint64_t)t.tv_nsec * 0x9e3779b97f4a7c15ull ^ (uint64_t)t.tv_sec << 20 ^ (uint64_t)getpid() << 40 ^ (uint64_t)(uintptr_t)&k;
      for (auto &w : r) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; w = x; x += 0x9e3779b97f4a7c15ull; }
    }
    for (int j = 0; j < 4; ++j) k.init[j] ^= r[j];
    k.mul = r[4] | 1;                                      // odd: multiplication stays a bijection
    k.fold = r[5] | 1;
    return k;
  }();
  return key;
}
// words [lo, hi) of one polynomial: four multiply-xor lanes (the multiply's latency is covered, the loop runs at memory speed)
inline uint64_t hash_words(const uint64_t *a, size_t lo, size_t hi) {
  const HashKey &key = hash_key();
  const uint64_t M = key.mul;
  uint64_t h[4] = {key.init[0], key.init[1], key.init[2], key.init[3]};
  size_t i = lo;
  for (; i + 4 <= hi; i += 4)
    for (int j = 0; j < 4; ++j) { h[j] = (h[j] ^ a[i + j]) * M; h[j] ^= h[j] >> 32; }
  for (; i < hi; ++i) { h[0] = (h[0] ^ a[i]) * M; h[0] ^= h[0] >> 32; }
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

}  // namespace
