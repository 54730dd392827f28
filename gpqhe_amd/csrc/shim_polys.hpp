// shim_polys.hpp -- resident polynomials: the device keeps the slab of every polynomial it has read or written; Operands = the speculation protocol of one call.
// Part of the MPI-typed surface: one translation unit (mpi_shim.hip includes these fragments in order); split by concern in round 4.
#pragma once

namespace {

// ---- resident polynomials ----------------------------------------------------------------------------------------------
// GPQHE's own callers chain the calls on one ciphertext (he_mul(&bn, &bn, &bn, rlk); he_rs(&bn); ... src/he-algo.c:140-160): what one call
// writes into the caller's integers is what the next one reads back.  Reading 2 x 65536 scattered libgcrypt integers and sending them
// over PCIe is most of a call (0.7 of 1.9 ms for he_mul), so the device keeps the word-major slab of every polynomial it has seen or
// produced, identified by the caller's coefficient array, the shape and a fingerprint of EVERY word of every coefficient.  When all
// operands of a call (and its key) are resident, the device starts from the resident copies at once and the host threads meanwhile do
// exactly the conversion an upload would have done, into the staging rows, and fingerprint it: a mismatch -- the caller changed the
// integers since, by any means -- uploads the rows that are then already staged and repeats the device work.  The result depends on a
// stale copy only if a changed polynomial collides with the kept one under the process's keyed 64-bit fingerprint (shim_staging.hpp: about
// 2^-64 per call for a party that cannot read this process's memory); a program whose operands are always new pays the fingerprint of its results (in cache, a few per cent).
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
  const HashKey &key = hash_key();
  uint64_t h = key.init[2] ^ ((uint64_t)n << 32 | W);
  for (unsigned t = 0; t < parts; ++t) { h = (h ^ part[t]) * key.fold; h ^= h >> 29; }
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

}  // namespace
