// mpi_convert.hpp -- host-only half of the MPI-typed surface: libgcrypt's runtime ABI (dlsym, no <gcrypt.h>), the conversions
// between libgcrypt integers and big slabs uint64_t[W][n] (two's complement, word-major), the worker threads they run on and
// the small multiword helpers context construction needs.  No HIP in this file: tests/c/convert_sanitized.cpp compiles it for the
// CPU with AddressSanitizer + UndefinedBehaviorSanitizer and round-trips real libgcrypt MPIs through it (tests/test_convert_sanitized.py);
// mpi_shim.hip includes it for the product.
#pragma once
#include "../../include/gpqhe_hip_compat.h"

#include <dlfcn.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

extern "C" const char *gpq_last_error(void);

namespace {

typedef void *MPI;
struct Gcry {
  MPI (*mpi_new)(unsigned);
  void (*mpi_release)(MPI);
  MPI (*mpi_set)(MPI, MPI);
  MPI (*mpi_set_ui)(MPI, unsigned long);
  void (*mpi_neg)(MPI, MPI);
  int (*mpi_is_neg)(MPI);
  unsigned (*mpi_get_nbits)(MPI);
  int (*mpi_test_bit)(MPI, unsigned);
  unsigned (*mpi_print)(int, unsigned char *, size_t, size_t *, MPI);
  unsigned (*mpi_scan)(MPI *, int, const void *, size_t, size_t *);
  void (*mpi_snatch)(MPI, MPI);
  void (*mpi_set_bit)(MPI, unsigned);
  void *(*xmalloc)(size_t);
  void (*xfree)(void *);
  bool ok = false;
} G;
const int FMT_USG = 5;  // GCRYMPI_FMT_USG: unsigned big-endian magnitude

[[noreturn]] void die(const char *what) {
  errno = EINVAL;  // the reference's error convention (src/reduce.c:95-100, src/precomp.c:344-350)
  fprintf(stderr, "\033[1m\033[31merror:\033[0m \033[1m%s\033[0m. %s (%s)\n", strerror(errno), what, gpq_last_error());
  abort();
}

// ---- direct access to libgcrypt integers ------------------------------------------------------------------------------------
// gcry_mpi_print / gcry_mpi_scan cost ~100 ns per coefficient (a format switch, a byte-order pass, an allocation for scan): at
// n = 2^16 that was 3 of the 3.5 ms of a reference-signature he_mul.  A libgcrypt integer is
//     struct gcry_mpi { int alloced; int nlimbs; int sign; unsigned flags; mpi_limb_t *d; }      (libgcrypt src/mpi.h, unchanged
// since 1.2: little-endian 64-bit limbs of the magnitude, sign apart; flags 1 = secure memory, 4 = opaque, 16 = immutable, 32 = constant)
// so the conversions read and write the limbs in place -- AFTER probe_mpi_layout() has checked, through the public API alone, that
// this process's libgcrypt really lays its integers out like that (values built with gcry_mpi_scan read back through the struct;
// values written through the struct read back with gcry_mpi_print).  If any check fails, or after gpq_mpi_shim_set_direct_mpi(0),
// every conversion goes through gcry_mpi_print / gcry_mpi_scan as before; opaque / immutable / constant integers always do.
struct MpiView { int alloced, nlimbs, sign; unsigned flags; uint64_t *d; };
bool g_mpi_direct = false;        // the probe passed
bool g_mpi_direct_wanted = true;  // gpq_mpi_shim_set_direct_mpi
inline bool mpi_direct() { return g_mpi_direct && g_mpi_direct_wanted; }

void probe_mpi_layout() {
  g_mpi_direct = false;
  if (sizeof(unsigned long) != 8 || sizeof(void *) != 8) return;
  unsigned char bytes[24];
  for (int i = 0; i < 24; ++i) bytes[i] = (unsigned char)(0x11 * (i % 13) + 3 + i);       // big-endian magnitude, top byte non-zero
  uint64_t want[3];
  for (int j = 0; j < 3; ++j) { uint64_t x; memcpy(&x, bytes + 16 - 8 * j, 8); want[j] = __builtin_bswap64(x); }
  MPI a = nullptr;
  if (G.mpi_scan(&a, FMT_USG, bytes, 24, nullptr) || !a) return;
  bool ok = true;
  const MpiView *v = (const MpiView *)a;
  ok = ok && v->nlimbs == 3 && v->alloced >= 3 && v->sign == 0 && !(v->flags & 4) && v->d && !memcmp(v->d, want, 24);
  G.mpi_neg(a, a);
  ok = ok && v->sign == 1 && v->nlimbs == 3 && G.mpi_is_neg(a);
  G.mpi_set_ui(a, 0);
  ok = ok && v->nlimbs == 0;
  G.mpi_set_ui(a, 0x1234567);
  ok = ok && v->nlimbs == 1 && v->sign == 0 && v->d[0] == 0x1234567;
  // writing: grow a fresh integer through the public API, fill the limbs in place, read back with gcry_mpi_print
  MPI b = G.mpi_new(0);
  MpiView *w = (MpiView *)b;
  ok = ok && b && w->nlimbs == 0 && w->sign == 0;
  if (ok) {
    G.mpi_set_bit(b, 64 * 3 - 1);
    ok = w->alloced >= 3 && w->d != nullptr;
  }
  if (ok) {
    memcpy(w->d, want, 24);
    w->nlimbs = 3; w->sign = 1;
    unsigned char back[32];
    size_t nw = 0;
    ok = !G.mpi_print(FMT_USG, back, sizeof back, &nw, b) && nw == 24 && !memcmp(back, bytes, 24) && G.mpi_is_neg(b) && G.mpi_get_nbits(b) == 64 * 2 + (64 - (unsigned)__builtin_clzll(want[2]));
    w->nlimbs = 0; w->sign = 0;
    ok = ok && G.mpi_get_nbits(b) == 0 && !G.mpi_is_neg(b);
  }
  if (b) G.mpi_release(b);
  G.mpi_release(a);
  g_mpi_direct = ok;
}

void need_gcrypt() {
  if (G.ok) return;
  void *h = RTLD_DEFAULT;
  if (!dlsym(h, "gcry_mpi_print")) {
    h = dlopen("libgcrypt.so.20", RTLD_NOW | RTLD_GLOBAL);
    if (!h) die("libgcrypt is not loaded and libgcrypt.so.20 cannot be opened");
  }
  auto get = [&](const char *n) { void *p = dlsym(h, n); if (!p) die(n); return p; };
  G.mpi_new = (MPI(*)(unsigned))get("gcry_mpi_new");
  G.mpi_release = (void (*)(MPI))get("gcry_mpi_release");
  G.mpi_set = (MPI(*)(MPI, MPI))get("gcry_mpi_set");
  G.mpi_set_ui = (MPI(*)(MPI, unsigned long))get("gcry_mpi_set_ui");
  G.mpi_neg = (void (*)(MPI, MPI))get("gcry_mpi_neg");
  G.mpi_is_neg = (int (*)(MPI))get("gcry_mpi_is_neg");
  G.mpi_get_nbits = (unsigned (*)(MPI))get("gcry_mpi_get_nbits");
  G.mpi_test_bit = (int (*)(MPI, unsigned))get("gcry_mpi_test_bit");
  G.mpi_print = (unsigned (*)(int, unsigned char *, size_t, size_t *, MPI))get("gcry_mpi_print");
  G.mpi_scan = (unsigned (*)(MPI *, int, const void *, size_t, size_t *))get("gcry_mpi_scan");
  G.mpi_snatch = (void (*)(MPI, MPI))get("gcry_mpi_snatch");
  G.mpi_set_bit = (void (*)(MPI, unsigned))get("gcry_mpi_set_bit");
  G.xmalloc = (void *(*)(size_t))get("gcry_malloc");
  G.xfree = (void (*)(void *))get("gcry_free");
  G.ok = true;
  probe_mpi_layout();
}

// magnitude of a positive MPI as little-endian words
std::vector<uint64_t> words_of(MPI q, const char *what) {
  const unsigned nb = G.mpi_get_nbits(q);
  if (!nb || G.mpi_is_neg(q)) die(what);
  std::vector<unsigned char> buf((nb + 7) / 8);
  size_t nw = 0;
  if (G.mpi_print(FMT_USG, buf.data(), buf.size(), &nw, q)) die("gcry_mpi_print failed");
  std::vector<uint64_t> w((nb + 63) / 64, 0);
  for (size_t b = 0; b < nw; ++b) w[(nw - 1 - b) >> 3] |= (uint64_t)buf[b] << (8 * ((nw - 1 - b) & 7));
  return w;
}
bool is_pow2(const std::vector<uint64_t> &w) {
  unsigned ones = 0;
  for (uint64_t v : w) ones += __builtin_popcountll(v);
  return ones == 1;
}

// The conversions are per coefficient and independent (gcry_mpi_print only reads its MPI, gcry_mpi_scan / gcry_mpi_set write
// the caller's own, distinct MPIs), so large polynomials are cut into ranges for a few host threads: at n = 2^16 the MPI <-> slab
// conversions are 40 ms of a 42 ms he_mul call on one thread.  The threads are started once and kept (starting 16 threads four
// times per call cost more than the device work of the call).
class Workers {
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable wake_, idle_;
  const std::function<void(unsigned)> *job_ = nullptr;
  unsigned tasks_ = 0, left_ = 0, open_ = 0;
  std::vector<char> taken_;
  uint64_t round_ = 0;

  // Task t belongs to thread t mod width first: the range of a polynomial a thread wrote in one call (results) is the range it reads
  // in the next (the check of a resident operand), out of its own cache.  A thread with nothing of its own left takes any task.
  int pick(unsigned me) {
    if (!left_) return -1;
    const unsigned w = width();
    for (unsigned t = me; t < tasks_; t += w) if (!taken_[t]) return (int)t;
    for (unsigned t = 0; t < tasks_; ++t) if (!taken_[t]) return (int)t;
    return -1;
  }
  void drain(std::unique_lock<std::mutex> &lk, unsigned me) {       // run tasks of the current round until none is left to take
    for (int t; (t = pick(me)) >= 0;) {
      taken_[t] = 1; --left_;
      const std::function<void(unsigned)> *job = job_;
      lk.unlock();
      (*job)((unsigned)t);
      lk.lock();
      if (--open_ == 0) idle_.notify_all();
    }
  }
  void loop(unsigned me) {
    std::unique_lock<std::mutex> lk(mu_);
    uint64_t seen = 0;
    for (;;) {
      wake_.wait(lk, [&] { return round_ != seen; });
      seen = round_;
      drain(lk, me);
    }
  }

 public:
  explicit Workers(unsigned helpers) {
    for (unsigned i = 0; i < helpers; ++i) { th_.emplace_back([this, i] { loop(i + 1); }); th_.back().detach(); }
  }
  unsigned width() const { return (unsigned)th_.size() + 1; }
  void run(unsigned tasks, const std::function<void(unsigned)> &f) {     // the caller works too; returns when every task is done
    std::unique_lock<std::mutex> lk(mu_);
    job_ = &f; tasks_ = tasks; left_ = tasks; open_ = tasks; taken_.assign(tasks, 0); ++round_;
    wake_.notify_all();
    drain(lk, 0);
    idle_.wait(lk, [&] { return open_ == 0; });
    job_ = nullptr; tasks_ = 0;
  }
};
unsigned g_workers_wanted = 0;   // gpq_mpi_shim_set_conversion_threads, before the first MPI-typed call; 0 = min(hardware threads, 16)
Workers &workers() {       // never destroyed: its threads wait detached and vanish with the process
  static Workers *w = [] {
    unsigned nt = g_workers_wanted ? g_workers_wanted : std::thread::hardware_concurrency();
    if (!g_workers_wanted && nt > 16) nt = 16;            // the conversions are bound by memory latency on scattered heap objects: more threads help until the
    if (nt > 64) nt = 64;                                 // host's memory system is busy; 16 is what a one-GPU share of a node usually has
    if (nt < 1) nt = 1;
    return new Workers(nt - 1);
  }();
  return *w;
}

// f(task, lo, hi) over [0, n) cut into ranges; small polynomials stay on the calling thread
template <typename F>
void for_ranges(unsigned n, F f) {
  const unsigned nt = n >= 4096 ? workers().width() : 1;
  if (nt < 2) { f(0u, n); return; }
  const unsigned per = (n + nt - 1) / nt;
  const std::function<void(unsigned)> job = [&](unsigned t) {
    const unsigned lo = t * per, hi = lo + per < n ? lo + per : n;
    if (lo < hi) f(lo, hi);
  };
  workers().run(nt, job);
}

unsigned max_bits(const poly_mpi_t *a, unsigned n) {       // widest coefficient (the MPIs are scattered heap objects: worth the threads)
  std::mutex mu;
  unsigned m = 0;
  for_ranges(n, [&](unsigned lo, unsigned hi) {
    unsigned mine = 0;
    const bool direct = mpi_direct();
    for (unsigned i = lo; i < hi; ++i) {
      unsigned b;
      const MpiView *m = (const MpiView *)a->coeffs[i];
      if (direct && !(m->flags & 4)) {
        unsigned nl = m->nlimbs > 0 ? (unsigned)m->nlimbs : 0;
        while (nl && m->d[nl - 1] == 0) --nl;
        b = nl ? 64 * (nl - 1) + (64 - (unsigned)__builtin_clzll(m->d[nl - 1])) : 0;
      } else b = G.mpi_get_nbits(a->coeffs[i]);
      if (b > mine) mine = b;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (mine > m) m = mine;
  });
  return m;
}

// MPI coefficients -> host big slab [W][n], two's complement.  libgcrypt hands out / takes big-endian magnitude bytes; words are
// assembled eight bytes at a time (the byte-at-a-time form cost as much as gcry_mpi_print itself).
// ROWS = false: the device layout, word j of coefficient i at j*n + i (14 write streams per thread);  ROWS = true: one row of W words per
// coefficient, i*W + j -- one sequential stream, what the MPI-typed calls stage through (the device transposes: bridge_big_transpose)
// `misfit`: a coefficient that does not fit W words is reported there (its words are left as they are) instead of ending the program --
// for the conversions that only CHECK a resident device copy against the caller's integers (mpi_shim.hip, resident polynomials).
template <bool ROWS = false>
void to_slab_range(uint64_t *dst, const poly_mpi_t *a, unsigned n, unsigned W, unsigned lo, unsigned hi, bool *misfit = nullptr) {
  const size_t sj = ROWS ? 1 : n, si = ROWS ? W : 1;
  unsigned char buf[8 * 64 + 8];
  const bool direct = mpi_direct();
  for (unsigned i = lo; i < hi; ++i) {
    MPI v = a->coeffs[i];
    if (direct) {
      // the integers are 65536 separate heap objects, each with its own limb array: two dependent cache misses per coefficient
      // unless they are asked for a few coefficients ahead (struct at i + 12, its limbs at i + 6)
      if (i + 12 < hi) __builtin_prefetch(a->coeffs[i + 12]);
      if (i + 6 < hi) { const uint64_t *d = ((const MpiView *)a->coeffs[i + 6])->d; __builtin_prefetch(d); __builtin_prefetch(d + 8); }
    }
    if (direct && !(((const MpiView *)v)->flags & 4)) {          // limbs in place: little-endian magnitude + sign
      const MpiView *m = (const MpiView *)v;
      unsigned nl = m->nlimbs > 0 ? (unsigned)m->nlimbs : 0;
      while (nl && m->d[nl - 1] == 0) --nl;
      if (nl > W || (nl == W && (m->d[W - 1] >> 63))) {
        if (misfit) { *misfit = true; continue; }
        die("coefficient does not fit the big slab");
      }
      if (m->sign && nl) {
        uint64_t carry = 1;
        for (unsigned j = 0; j < W; ++j) { const uint64_t x = ~(j < nl ? m->d[j] : 0) + carry; carry = carry && x == 0; dst[j * sj + i * si] = x; }
      } else {
        for (unsigned j = 0; j < W; ++j) dst[j * sj + i * si] = j < nl ? m->d[j] : 0;
      }
      continue;
    }
    if (G.mpi_get_nbits(v) > 64 * W - 1) {
      if (misfit) { *misfit = true; continue; }
      die("coefficient does not fit the big slab");
    }
    size_t nw = 0;
    if (G.mpi_print(FMT_USG, buf, 8 * W, &nw, v)) die("gcry_mpi_print failed");
    uint64_t w[64];
    const unsigned full = (unsigned)(nw >> 3), rest = (unsigned)(nw & 7);
    for (unsigned j = 0; j < full; ++j) {                     // word j = bytes [nw - 8 (j + 1), nw - 8 j) of the big-endian string
      uint64_t x;
      memcpy(&x, buf + nw - 8 * (size_t)(j + 1), 8);
      w[j] = __builtin_bswap64(x);
    }
    unsigned used = full;
    if (rest) {
      uint64_t x = 0;
      for (unsigned k = 0; k < rest; ++k) x = (x << 8) | buf[k];
      w[used++] = x;
    }
    for (unsigned j = used; j < W; ++j) w[j] = 0;
    if (G.mpi_is_neg(v)) {
      uint64_t carry = 1;
      for (unsigned j = 0; j < W; ++j) { w[j] = ~w[j] + carry; carry = carry && w[j] == 0; }
    }
    for (unsigned j = 0; j < W; ++j) dst[j * sj + i * si] = w[j];
  }
}
void to_slab(uint64_t *dst, const poly_mpi_t *a, unsigned n, unsigned W) {
  if (W < 1 || W > 32) die("coefficients wider than 2047 bits");      // the device kernels hold at most 32 words; w[64] below is sized for that
  for_ranges(n, [=](unsigned lo, unsigned hi) { to_slab_range(dst, a, n, W, lo, hi); });
}

// host big slab -> existing MPIs (the caller allocated them, src/poly.c:46-51)
template <bool ROWS = false>
void from_slab_range(poly_mpi_t *r, const uint64_t *src, unsigned n, unsigned W, unsigned lo, unsigned hi) {
  const size_t sj = ROWS ? 1 : n, si = ROWS ? W : 1;
  unsigned char buf[8 * 64];
  const bool direct = mpi_direct();
  for (unsigned i = lo; i < hi; ++i) {
    if (direct) {                                             // as in to_slab_range: the destination integers are scattered heap objects
      if (i + 12 < hi) __builtin_prefetch(r->coeffs[i + 12]);
      if (i + 6 < hi) { uint64_t *d = ((MpiView *)r->coeffs[i + 6])->d; if (d) { __builtin_prefetch(d, 1); __builtin_prefetch(d + 8, 1); } }
    }
    uint64_t w[64];
    for (unsigned j = 0; j < W; ++j) w[j] = src[j * sj + i * si];
    const bool neg = w[W - 1] >> 63;
    if (neg) {
      uint64_t carry = 1;
      for (unsigned j = 0; j < W; ++j) { w[j] = ~w[j] + carry; carry = carry && w[j] == 0; }
    }
    unsigned top = W;                                         // words in use
    while (top && w[top - 1] == 0) --top;
    if (direct && !(((MpiView *)r->coeffs[i])->flags & (4 | 16 | 32))) {     // limbs in place (the integer grows through the public API)
      MpiView *m = (MpiView *)r->coeffs[i];
      if (top && (unsigned)m->alloced < top) G.mpi_set_bit(r->coeffs[i], 64 * top - 1);
      if (!top || ((unsigned)m->alloced >= top && m->d)) {
        for (unsigned j = 0; j < top; ++j) m->d[j] = w[j];
        m->nlimbs = (int)top; m->sign = (neg && top) ? 1 : 0;
        continue;
      }
    }
    if (!top) { G.mpi_set_ui(r->coeffs[i], 0); continue; }
    for (unsigned j = 0; j < top; ++j) {                      // big-endian bytes, most significant word first
      const uint64_t x = __builtin_bswap64(w[top - 1 - j]);
      memcpy(buf + 8 * (size_t)j, &x, 8);
    }
    const size_t len = 8 * (size_t)top, skip = (size_t)__builtin_clzll(w[top - 1]) >> 3;
    MPI t = nullptr;
    if (G.mpi_scan(&t, FMT_USG, buf + skip, len - skip, nullptr)) die("gcry_mpi_scan failed");
    if (neg) G.mpi_neg(t, t);
    G.mpi_snatch(r->coeffs[i], t);                            // r->coeffs[i] takes t's limbs (no copy) and t is released
  }
}
void from_slab(poly_mpi_t *r, const uint64_t *src, unsigned n, unsigned W) {
  if (W < 1 || W > 64) die("big slab wider than 64 words");
  for_ranges(n, [=](unsigned lo, unsigned hi) { from_slab_range(r, src, n, W, lo, hi); });
}

typedef std::vector<uint64_t> Words;   // non-negative integer, little-endian 64-bit words, no leading zero word (0 = empty)
typedef unsigned __int128 u128s;

void trim(Words &a) { while (!a.empty() && a.back() == 0) a.pop_back(); }
void mul_word(Words &a, uint64_t m) {
  uint64_t carry = 0;
  for (uint64_t &w : a) { const u128s t = (u128s)w * m + carry; w = (uint64_t)t; carry = (uint64_t)(t >> 64); }
  if (carry) a.push_back(carry);
  trim(a);
}
uint64_t divmod_word(Words &a, uint64_t d) {      // a <- floor(a / d), returns a mod d
  uint64_t rem = 0;
  for (size_t i = a.size(); i-- > 0;) { const u128s t = ((u128s)rem << 64) | a[i]; a[i] = (uint64_t)(t / d); rem = (uint64_t)(t % d); }
  trim(a);
  return rem;
}
Words mul_words(const Words &a, const Words &b) {
  Words r(a.size() + b.size() + 1, 0);
  for (size_t i = 0; i < a.size(); ++i) {
    uint64_t carry = 0;
    for (size_t j = 0; j < b.size(); ++j) { const u128s t = (u128s)a[i] * b[j] + r[i + j] + carry; r[i + j] = (uint64_t)t; carry = (uint64_t)(t >> 64); }
    r[i + b.size()] += carry;
  }
  trim(r);
  return r;
}
unsigned bits_of(const Words &a) { return a.empty() ? 0 : 64 * (unsigned)(a.size() - 1) + (64 - (unsigned)__builtin_clzll(a.back())); }
void shr1(Words &a) { for (size_t i = 0; i < a.size(); ++i) a[i] = (a[i] >> 1) | (i + 1 < a.size() ? a[i + 1] << 63 : 0); trim(a); }

MPI mpi_of(const Words &w) {                       // a fresh libgcrypt integer with this value
  MPI r = G.mpi_new(0);
  if (w.empty()) { G.mpi_set_ui(r, 0); return r; }
  Words padded = w;
  padded.push_back(0);                                    // one more (zero) word: the value is non-negative
  poly_mpi_t one{&r};
  from_slab(&one, padded.data(), 1, (unsigned)padded.size());
  return r;
}
uint64_t powm64(uint64_t b, uint64_t e, uint64_t m) {
  uint64_t r = 1;
  for (b %= m; e; e >>= 1) { if (e & 1) r = (uint64_t)((u128s)r * b % m); b = (uint64_t)((u128s)b * b % m); }
  return r;
}

}  // namespace
