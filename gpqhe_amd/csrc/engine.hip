// engine.hip -- host side of libgpqhe_hip.so: context construction (the RNS
// part of polyctx_init, src/precomp.c:244-264 and :354-380), kernel launchers
// and the C ABI declared in include/gpqhe_hip.h.
#include "../../include/gpqhe_hip.h"
#include "engine_internal.hpp"
#include "ntt_kernels.hpp"

#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>

using namespace gpq;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int gpq_fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
extern "C" const char *gpq_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------
// roctx ranges (engine_internal.hpp)
// ---------------------------------------------------------------------------
#include <dlfcn.h>
#include <atomic>
namespace {
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)(void);
std::atomic<int> g_roctx_state{0};        // 0 = not looked up, 1 = available, 2 = absent
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop = nullptr;
bool roctx_ready() {
  int st = g_roctx_state.load(std::memory_order_acquire);
  if (st == 0) {
    void *h = RTLD_DEFAULT;
    void *push = dlsym(h, "roctxRangePushA");
    for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4"}) {
      if (push) break;
      // only a library that is ALREADY in the process (a profiler loaded it): never pull a tracing runtime in ourselves
      if ((h = dlopen(lib, RTLD_NOW | RTLD_NOLOAD))) push = dlsym(h, "roctxRangePushA");
    }
    if (push) {
      g_roctx_push = (roctx_push_t)push;
      g_roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
    }
    st = (g_roctx_push && g_roctx_pop) ? 1 : 2;
    g_roctx_state.store(st, std::memory_order_release);
  }
  return st == 1;
}
}  // namespace
void gpq_range_push(const char *name) { if (roctx_ready()) (void)g_roctx_push(name); }
void gpq_range_pop() { if (roctx_ready()) (void)g_roctx_pop(); }

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return gpq_fail(GPQ_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// ---------------------------------------------------------------------------
// host number theory (64-bit), own implementation of what precomp.c needs
// ---------------------------------------------------------------------------
constexpr unsigned kMaxPolysPerLaunch = 16384;   // gridDim.y <= 65535 with up to 4 slabs per launch

namespace {

typedef unsigned __int128 u128h;

inline uint64_t mulm(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((u128h)a * b % m); }

uint64_t powm(uint64_t b, uint64_t e, uint64_t m) {
  uint64_t r = 1 % m;
  b %= m;
  while (e) {
    if (e & 1) r = mulm(r, b, m);
    b = mulm(b, b, m);
    e >>= 1;
  }
  return r;
}

// Deterministic Miller-Rabin for 64-bit inputs (7-base set).  The reference
// draws 50 rand() witnesses (src/precomp.c:153-191); both accept exactly the
// primes.
bool is_prime_u64(uint64_t n) {
  if (n < 2) return false;
  for (uint64_t sp : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull}) {
    if (n % sp == 0) return n == sp;
  }
  uint64_t d = n - 1;
  int s = __builtin_ctzll(d);
  d >>= s;
  for (uint64_t a : {2ull, 325ull, 9375ull, 28178ull, 450775ull, 9780504ull, 1795265022ull}) {
    uint64_t x = powm(a, d, n);
    if (x == 0 || x == 1 || x == n - 1) continue;
    bool witness = true;
    for (int r = 1; r < s && witness; ++r) {
      x = mulm(x, x, n);
      if (x == n - 1) witness = false;
    }
    if (witness) return false;
  }
  return true;
}

// distinct prime factors of m (m = p-1, always has a large power of two)
std::vector<uint64_t> prime_factors(uint64_t m) {
  std::vector<uint64_t> f;
  auto strip = [&](uint64_t d) {
    if (m % d == 0) {
      f.push_back(d);
      while (m % d == 0) m /= d;
    }
  };
  strip(2);
  strip(3);
  for (uint64_t d = 5; d * d <= m; d += 6) {
    strip(d);
    strip(d + 2);
  }
  if (m > 1) f.push_back(m);
  return f;
}

// least primitive root, as the loop at src/precomp.c:216-224 finds it
uint64_t least_generator(uint64_t p) {
  const std::vector<uint64_t> f = prime_factors(p - 1);
  for (uint64_t g = 2; g < p; ++g) {
    bool ok = true;
    for (uint64_t q : f)
      if (powm(g, (p - 1) / q, p) == 1) { ok = false; break; }
    if (ok) return g;
  }
  return 0;
}

inline uint32_t bitrev(uint32_t v, unsigned bits) {
  return bits ? (__builtin_bitreverse32(v) >> (32 - bits)) : 0;
}

uint64_t inv_mod_2_64(uint64_t q) {  // Newton; equals the 64-step product of src/reduce.c:36-48 for odd q
  uint64_t x = q;                    // 3 correct bits
  for (int i = 0; i < 6; ++i) x *= 2 - q * x;
  return x;
}

}  // namespace

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
static int alloc_zero_flags(gpq_ctx *c);
static int upload_tables(gpq_ctx *c) {
  const size_t n = c->n, np = c->nprimes;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (c->device < 0 || c->device >= ndev) return gpq_fail(GPQ_ERR_INVALID, "device %d outside 0..%d", c->device, ndev - 1);
  DeviceScope on_device(c->device);
  if (!c->cache) c->cache = new (std::nothrow) gpq_table_cache();
  if (!c->cache) return gpq_fail(GPQ_ERR_NOMEM, "out of host memory");
  HIP_TRY(gpq_table_malloc(c, (void **)&c->d_w, np * n * sizeof(uint64_t)));
  HIP_TRY(gpq_table_malloc(c, (void **)&c->d_winv, np * n * sizeof(uint64_t)));
  HIP_TRY(gpq_table_malloc(c, (void **)&c->d_tabs, np * sizeof(LimbTab)));
  std::vector<uint64_t> wstd(np * n), wistd(np * n);
  std::vector<LimbTab> tabs(np);
  for (size_t d = 0; d < np; ++d) {
    const uint64_t p = c->p[d];
    if (p <= (1ull << 59) || p - (1ull << 59) >= GPQ_FOLD_CMAX)
      return gpq_fail(GPQ_ERR_UNSUPPORTED, "prime %llu is not 2^59 + c with c < %u", (unsigned long long)p, GPQ_FOLD_CMAX);
    const uint64_t pinv = c->pinv_mont[d];
    // Montgomery form -> standard form: z * 2^-64 mod p  (montgomery_reduce, src/reduce.c:59-66)
    auto from_mont = [&](uint64_t z) {
      const uint64_t u = z * pinv;
      const uint64_t t = (uint64_t)(((u128h)u * p) >> 64);
      return t ? p - t : 0;  // hi = 0 here, so hi - t + p
    };
    for (size_t i = 0; i < n; ++i) {
      wstd[d * n + i] = from_mont(c->zetas[d * n + i]);
      wistd[d * n + i] = from_mont(c->zetas_inv[d * n + i]);
    }
    LimbTab &t = tabs[d];
    t.k.p = p; t.k.p2 = 2 * p; t.k.p4 = 4 * p; t.k.c = (uint32_t)(p - (1ull << 59)); t.k.c1 = t.k.c + 1;
    t.k.kx0 = t.k.c1; t.k.kx1 = (uint64_t)t.k.c1 - 4 * p; t.k.ky = 4 * p - 2 * (uint64_t)t.k.c1;
    t.k.kys = 2 * p - 2 * (uint64_t)t.k.c1;
    t.k.p3 = 3 * p; t.k.np3 = (uint64_t)0 - 3 * p; t.k.kx1x = (uint64_t)t.k.c1 - 3 * p; t.k.kyx = 3 * p - 2 * (uint64_t)t.k.c1;
    t.k.np = (uint64_t)0 - p; t.k.np2 = (uint64_t)0 - 2 * p; t.k.np4 = (uint64_t)0 - 4 * p;
    t.ninv = from_mont(c->ninv_mont[d]);
    t.winv1_ninv = n >= 2 ? mulm(wistd[d * n + 1], t.ninv, p) : t.ninv;
    t.ninv_s = t.winv1_ninv_s = TwS{0, 0};
  }
  // split-twiddle pairs (p - w, p - w*2^31 mod p) for the leading limbs whose c allows the single fold ...
  c->nsplit = 0;
  while (c->nsplit < np && c->p[c->nsplit] - (1ull << 59) < GPQ_SPLIT_CMAX) ++c->nsplit;
  // ... and among them the leading limbs whose forward stages may skip every other conditional subtraction (ct_bfly_wide)
  c->nwide = 0;
  while (c->nwide < c->nsplit && c->p[c->nwide] - (1ull << 59) < GPQ_WIDE_CMAX) ++c->nwide;
  c->nsplit_tables = c->nsplit; c->nwide_max = c->nwide;
  c->low9 = c->logn == 17;            // n = 2^17: 8 strided stages over 512-coefficient rows + 9 low stages
  if (c->nsplit) {
    const size_t ns = c->nsplit;
    std::vector<TwS> ws(ns * n), wis(ns * n);
    auto pair_of = [](uint64_t w, uint64_t p) { return TwS{p - w, p - (uint64_t)(((u128h)w << 31) % p)}; };
    for (size_t d = 0; d < ns; ++d) {
      const uint64_t p = c->p[d];
      for (size_t i = 0; i < n; ++i) {
        ws[d * n + i] = pair_of(wstd[d * n + i], p);
        wis[d * n + i] = pair_of(wistd[d * n + i], p);
      }
      tabs[d].ninv_s = pair_of(tabs[d].ninv, p);
      tabs[d].winv1_ninv_s = pair_of(tabs[d].winv1_ninv, p);
    }
    HIP_TRY(gpq_table_malloc(c, (void **)&c->d_ws, ns * n * sizeof(TwS)));
    HIP_TRY(gpq_table_malloc(c, (void **)&c->d_winvs, ns * n * sizeof(TwS)));
    HIP_TRY(hipMemcpy(c->d_ws, ws.data(), ns * n * sizeof(TwS), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_winvs, wis.data(), ns * n * sizeof(TwS), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemcpy(c->d_w, wstd.data(), np * n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_winv, wistd.data(), np * n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_tabs, tabs.data(), np * sizeof(LimbTab), hipMemcpyHostToDevice));
  c->h_tabs = tabs;
  return alloc_zero_flags(c);
}

// gpq_ntt's zero flags: the one device allocation every context -- a peer lane too -- owns from its creation
static int alloc_zero_flags(gpq_ctx *c) {
  DeviceScope on_device(c->device);
  // gpq_ntt's zero flags for launches of up to 4096 (polynomial, limb) units exist from the start, so that a FIRST gpq_ntt inside a
  // stream capture works; larger launches grow them at their first call, outside capture (zero_flags)
  HIP_TRY(hipMalloc((void **)&c->d_zflag, 4096 * sizeof(unsigned)));
  HIP_TRY(hipMemset(c->d_zflag, 0, 4096 * sizeof(unsigned)));
  HIP_TRY(hipStreamSynchronize(nullptr));      // (a device memset is not host-synchronous: done before any stream of the caller's can see the words)
  c->zflag_cap = 4096;
  return GPQ_OK;
}

static void fill_scalar_constants(gpq_ctx *c, size_t d) {
  const uint64_t p = c->p[d];
  c->pinv_mont[d] = inv_mod_2_64(p);                                                  // src/reduce.c:36-48
  c->pinv_barr[d] = (uint64_t)(((u128h)1 << (2 * (64 - __builtin_clzll(p)))) / p);   // src/reduce.c:75-78
  const uint64_t r_mod_p = (uint64_t)(((u128h)1 << 64) % p);
  c->ninv_mont[d] = mulm(powm(c->n, p - 2, p), r_mod_p, p);                           // src/precomp.c:248
}

extern "C" int gpq_ctx_create(gpq_ctx **out, unsigned logn, unsigned nprimes, int device) {
  if (!out || logn < 1 || logn > 17 || nprimes < 1 || nprimes > 4096)
    return gpq_fail(GPQ_ERR_INVALID, "gpq_ctx_create: logn=%u nprimes=%u out of range", logn, nprimes);
  gpq_ctx *c = new (std::nothrow) gpq_ctx();
  if (!c) return gpq_fail(GPQ_ERR_NOMEM, "out of host memory");
  c->device = device; c->logn = logn; c->n = 1u << logn; c->nprimes = nprimes;
  const size_t n = c->n;
  c->p.resize(nprimes); c->pinv_mont.resize(nprimes); c->pinv_barr.resize(nprimes);
  c->ninv_mont.resize(nprimes); c->psi.resize(nprimes);
  c->zetas.resize((size_t)nprimes * n); c->zetas_inv.resize((size_t)nprimes * n);
  uint64_t p = (1ull << 59) + 1;                         // src/precomp.c:358 (GPQHE_LOGP = 59)
  for (size_t d = 0; d < nprimes; ++d) {
    do p += 2 * n; while (!is_prime_u64(p));            // src/precomp.c:372-376
    c->p[d] = p;
    fill_scalar_constants(c, d);
    const uint64_t psi = powm(least_generator(p), (p - 1) / (2 * n), p);  // src/precomp.c:235-242, :251
    const uint64_t psi_inv = powm(psi, p - 2, p);
    const uint64_t r_mod_p = (uint64_t)(((u128h)1 << 64) % p);
    c->psi[d] = psi;
    uint64_t pw = r_mod_p, pwi = r_mod_p;               // psi^i * 2^64 mod p, carried in Montgomery form
    for (uint32_t i = 0; i < n; ++i) {                  // src/precomp.c:255-263
      const uint32_t j = bitrev(i, logn);
      c->zetas[d * n + j] = pw;
      c->zetas_inv[d * n + j] = pwi;
      pw = mulm(pw, psi, p);
      pwi = mulm(pwi, psi_inv, p);
    }
  }
  int rc = upload_tables(c);
  if (rc != GPQ_OK) { gpq_ctx_destroy(c); return rc; }
  *out = c;
  return GPQ_OK;
}

extern "C" int gpq_ctx_create_from_tables(gpq_ctx **out, unsigned logn, unsigned nprimes, const uint64_t *primes,
                                          const uint64_t *const *zetas_mont, const uint64_t *const *zetas_inv_mont,
                                          int device) {
  if (!out || !primes || !zetas_mont || !zetas_inv_mont || logn < 1 || logn > 17 || nprimes < 1)
    return gpq_fail(GPQ_ERR_INVALID, "gpq_ctx_create_from_tables: bad arguments");
  gpq_ctx *c = new (std::nothrow) gpq_ctx();
  if (!c) return gpq_fail(GPQ_ERR_NOMEM, "out of host memory");
  c->device = device; c->logn = logn; c->n = 1u << logn; c->nprimes = nprimes;
  const size_t n = c->n;
  c->p.assign(primes, primes + nprimes);
  c->pinv_mont.resize(nprimes); c->pinv_barr.resize(nprimes); c->ninv_mont.resize(nprimes); c->psi.assign(nprimes, 0);
  c->zetas.resize((size_t)nprimes * n); c->zetas_inv.resize((size_t)nprimes * n);
  for (size_t d = 0; d < nprimes; ++d) {
    if (!(c->p[d] & 1)) { delete c; return gpq_fail(GPQ_ERR_INVALID, "even modulus"); }
    fill_scalar_constants(c, d);
    memcpy(&c->zetas[d * n], zetas_mont[d], n * 8);
    memcpy(&c->zetas_inv[d * n], zetas_inv_mont[d], n * 8);
  }
  int rc = upload_tables(c);
  if (rc != GPQ_OK) { gpq_ctx_destroy(c); return rc; }
  *out = c;
  return GPQ_OK;
}

int gpq_ctx_clone(const gpq_ctx *c, gpq_ctx **out) {
  gpq_ctx *q = new (std::nothrow) gpq_ctx();
  if (!q) return gpq_fail(GPQ_ERR_NOMEM, "out of host memory");
  q->device = c->device; q->logn = c->logn; q->n = c->n; q->nprimes = c->nprimes;
  q->p = c->p; q->pinv_mont = c->pinv_mont; q->pinv_barr = c->pinv_barr; q->ninv_mont = c->ninv_mont; q->psi = c->psi;
  // (no host twiddles: gpq_ctx_zetas is the parent's business; nothing the peer runs reads them)
  q->overlap = 0;
  // every read-only device table is the parent's (engine_internal.hpp: gpq_table_cache)
  q->tables_of = c;
  q->d_w = c->d_w; q->d_winv = c->d_winv; q->d_ws = c->d_ws; q->d_winvs = c->d_winvs; q->d_tabs = c->d_tabs;
  q->h_tabs = c->h_tabs;
  q->nsplit = c->nsplit; q->nsplit_tables = c->nsplit_tables; q->nwide = c->nwide; q->nwide_max = c->nwide_max; q->low9 = c->low9;
  q->cache = c->cache;
  int rc = alloc_zero_flags(q);
  if (rc != GPQ_OK) { gpq_ctx_destroy(q); return rc; }
  *out = q;
  return GPQ_OK;
}

extern "C" void gpq_ctx_destroy(gpq_ctx *c) {
  if (!c) return;
  if (c->peer) {
    if (c->peer_stream) (void)hipStreamSynchronize(c->peer_stream);
    gpq_ctx_destroy(c->peer);                                    // before the tables it borrows go
  }
  if (c->peer_stream) (void)hipStreamDestroy(c->peer_stream);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->peer_ws) (void)hipFree(c->peer_ws);
  if (!c->tables_of) {
    if (c->d_w) (void)hipFree(c->d_w);
    if (c->d_winv) (void)hipFree(c->d_winv);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_winvs) (void)hipFree(c->d_winvs);
    if (c->d_tabs) (void)hipFree(c->d_tabs);
  }
  if (c->d_zflag) (void)hipFree(c->d_zflag);
  if (c->d_zwatch) (void)hipFree(c->d_zwatch);
  gpq_bridge_release(c);
  for (gpq_prof_rec &r : c->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (hipEvent_t e : c->prof_pool) (void)hipEventDestroy(e);
  delete c;
}

// Tests / DESIGN section 6: read-only device memory the context owns (which = 0: transform tables + every bridge constant built so far), what its
// peer lane owns of the same kind (1: zero by construction, the peer borrows), and whether a peer exists (2).
extern "C" size_t gpq_debug_table_bytes(const gpq_ctx *c, int which) {
  if (!c) return 0;
  if (which == 0) return c->cache && !c->tables_of ? c->cache->device_bytes : 0;
  if (which == 1) return c->peer && !c->peer->tables_of && c->peer->cache ? c->peer->cache->device_bytes : 0;
  if (which == 2) return c->peer ? 1 : 0;
  if (which == 3) return c->peer ? (size_t)(c->peer->d_w == c->d_w && c->peer->d_winv == c->d_winv && c->peer->d_ws == c->d_ws && c->peer->d_winvs == c->d_winvs &&
                                            c->peer->d_tabs == c->d_tabs && c->peer->cache == c->cache) : 0;
  return 0;
}

extern "C" unsigned gpq_ctx_logn(const gpq_ctx *c) { return c->logn; }
extern "C" unsigned gpq_ctx_nprimes(const gpq_ctx *c) { return c->nprimes; }
extern "C" int gpq_ctx_device(const gpq_ctx *c) { return c->device; }
extern "C" uint64_t gpq_ctx_const(const gpq_ctx *c, unsigned d, int which) {
  if (d >= c->nprimes) return 0;
  switch (which) {
    case 0: return c->p[d];
    case 1: return c->pinv_mont[d];
    case 2: return c->pinv_barr[d];
    case 3: return c->ninv_mont[d];
    case 4: return c->psi[d];
  }
  return 0;
}
extern "C" const uint64_t *gpq_ctx_zetas(const gpq_ctx *c, unsigned d, int inverse) {
  if (d >= c->nprimes) return nullptr;
  return (inverse ? c->zetas_inv.data() : c->zetas.data()) + (size_t)d * c->n;
}
extern "C" unsigned gpq_dimub(unsigned logn, unsigned logq) { return (1 + logn + 4 * logq) / 59 + 1; }

// ---------------------------------------------------------------------------
// memory / stream helpers
// ---------------------------------------------------------------------------
extern "C" int gpq_malloc(void **dptr, size_t bytes) { HIP_TRY(hipMalloc(dptr, bytes)); return GPQ_OK; }
extern "C" int gpq_free(void *dptr) { HIP_TRY(hipFree(dptr)); return GPQ_OK; }
extern "C" int gpq_upload(void *dst, const void *src, size_t bytes, void *stream) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return GPQ_OK;
}
extern "C" int gpq_download(void *dst, const void *src, size_t bytes, void *stream) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return GPQ_OK;
}
extern "C" int gpq_copy(void *dst, const void *src, size_t bytes, void *stream) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return GPQ_OK;
}
extern "C" int gpq_stream_sync(void *stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return GPQ_OK; }
// Ordering between two streams without stopping the host: work queued on `waiter` after this call starts only when everything queued on `on` up to
// this call has finished.  What a host needs to run uploads, the two stages and downloads of successive sub-batches on three streams (the link is full
// duplex: bench.py's `with_host_scatter.pipelined`, tests/c/shard_host.c `pipe`).  The event lives until the wait has been consumed (destroying a
// recorded event only releases it once it completes).
extern "C" int gpq_stream_wait(void *waiter, void *on) {
  hipEvent_t e;
  HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipError_t rc = hipEventRecord(e, (hipStream_t)on);
  if (rc == hipSuccess) rc = hipStreamWaitEvent((hipStream_t)waiter, e, 0);
  (void)hipEventDestroy(e);
  if (rc != hipSuccess) return gpq_fail(GPQ_ERR_HIP, "gpq_stream_wait: %s", hipGetErrorString(rc));
  return GPQ_OK;
}

// Several devices from one C program (independent ciphertexts, one shard per GPU): the calling thread's current
// device decides where gpq_malloc allocates and where a new stream lives; a context is bound to the device it was created on.
extern "C" int gpq_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
extern "C" int gpq_set_device(int device) { HIP_TRY(hipSetDevice(device)); return GPQ_OK; }
extern "C" int gpq_stream_create(void **stream) {
  if (!stream) return gpq_fail(GPQ_ERR_INVALID, "gpq_stream_create: null pointer");
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void *)s;
  return GPQ_OK;
}
extern "C" int gpq_stream_destroy(void *stream) { HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return GPQ_OK; }
// page-locked host memory: gpq_upload / gpq_download from / to it are asynchronous DMA (with pageable memory they are staged and
// the download waits for the stream)
extern "C" int gpq_malloc_host(void **hptr, size_t bytes) { HIP_TRY(hipHostMalloc(hptr, bytes, hipHostMallocDefault)); return GPQ_OK; }
extern "C" int gpq_free_host(void *hptr) { HIP_TRY(hipHostFree(hptr)); return GPQ_OK; }

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
namespace {

static const char *const kKernelNames[GPQ_K_COUNT] = {"strided_fwd", "strided_inv", "contig_fwd", "contig_inv",
                                                      "tensor_mid", "keyswitch_mid", "pointwise", "small_ntt", "reference_redo",
                                                      "bridge_decompose", "bridge_reconstruct", "bridge_relin_front", "bridge_relin_tail_fused", "bridge_exact_paths", "bridge_rescale",
                                                      "bridge_relin_tail_direct", "bridge_crt_decompose", "bridge_tail_stream"};

int check_shape(const gpq_ctx *c, unsigned dim, unsigned batch, const char *who) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "%s: null context", who);
  if (dim < 1 || dim > c->nprimes) return gpq_fail(GPQ_ERR_INVALID, "%s: dim=%u outside 1..%u", who, dim, c->nprimes);
  if (batch < 1) return gpq_fail(GPQ_ERR_INVALID, "%s: empty batch", who);
  // kernels launch on the calling thread's current device: it must be the one the context (its tables, the caller's slabs) lives on
  int dev = -1;
  if (hipGetDevice(&dev) == hipSuccess && dev != c->device)
    return gpq_fail(GPQ_ERR_INVALID, "%s: the context lives on device %d but the calling thread's current device is %d (gpq_set_device(gpq_ctx_device(ctx)) first)", who, c->device, dev);
  return GPQ_OK;
}

// Cache policy of a launch group's slab traffic (PassArgs::nt).  A group whose slabs fit the 256 MiB Infinity Cache wants the default
// policy: what one kernel writes the next one reads from the cache (NTT pairs at configs[1], batch 64 = 168 MB: 11-15 % slower with nt; at
// 252 MB still 13 % slower).  A group that cannot fit gains 2.4-6 % from nt loads and stores, which pass without evicting the twiddle
// tables and each other's lines (336 MB: +4.6 %; the he_mul core at the headline shape: +2.4 %) -- profiles/r04/v10_nt_policy_ab.txt.
// `slabs` = the slabs of polys x limbs x n words the group moves between its kernels.
static unsigned nt_for(const gpq_ctx *c, size_t polys, unsigned limbs, unsigned slabs) {
  if (c->nt_mode >= 0) return (unsigned)c->nt_mode;
  return polys * limbs * ((size_t)8 << c->logn) * slabs > ((size_t)288 << 20);
}

PassArgs make_args(const gpq_ctx *c, unsigned dim, unsigned nslab, unsigned nt = 0) {
  PassArgs a;
  memset(&a, 0, sizeof a);
  a.nt = nt;
  a.tabs = c->d_tabs;
  a.w = c->d_w;
  a.winv = c->d_winv;
  a.ws = c->d_ws;
  a.winvs = c->d_winvs;
  a.poly_stride = (unsigned long long)dim << c->logn;
  a.logn = c->logn;
  a.limb0 = 0;
  a.nslab = nslab;
  return a;
}

// A pass over the limbs 0..dim-1 of its slabs is launched once per class of limbs present: wide-split (forward
// kernels only: WIDE), split, plain (none at n <= 2^16): f(twiddle type tag, args, limbs).  The classes are
// prefixes of the chain (c grows along it), so each is one contiguous range of limbs.
template <bool WIDE, typename F>
int for_limb_ranges(const gpq_ctx *c, PassArgs a, unsigned dim, const uint64_t **evk0, const uint64_t **evk1, F f) {
  const unsigned lo = a.limb0, hi = a.limb0 + dim;
  auto upto = [&](unsigned bound) { return bound < lo ? lo : (bound > hi ? hi : bound); };
  const unsigned e_wide = WIDE ? upto(c->nwide) : lo, e_split = upto(c->nsplit);
  auto advance = [&](unsigned limbs) {
    const size_t shift = (size_t)limbs << c->logn;
    a.limb0 += limbs;
    for (int i = 0; i < GPQ_MAX_SLABS; ++i) { if (a.src[i]) a.src[i] += shift; if (a.dst[i]) a.dst[i] += shift; }
    if (evk0) *evk0 += shift;
    if (evk1) *evk1 += shift;
    if (a.zflag) a.zflag += limbs;
  };
  int rc;
  if constexpr (WIDE)
    if (e_wide > lo) { if ((rc = f(TwW{}, a, e_wide - lo))) return rc; advance(e_wide - lo); }
  if (e_split > e_wide) { if ((rc = f(TwS{}, a, e_split - e_wide))) return rc; advance(e_split - e_wide); }
  if (hi > e_split && (rc = f(uint64_t{}, a, hi - e_split))) return rc;
  return GPQ_OK;
}

// PassArgs::nt picks the instantiation (a cache policy is an instruction modifier: the flag is a template parameter of the kernels, not a branch
// in them -- as a run-time select the compiler folds the two loads into one and drops the modifier): f(std::true_type / std::false_type)
template <typename F>
static inline void with_nt(unsigned nt, F f) {
  if (nt) f(std::true_type{});
  else f(std::false_type{});
}

template <int M1, int EL, bool INV, bool CANON, typename TW, int CW = 8>
int launch_strided_t(const PassArgs &a, unsigned gy, unsigned gz, hipStream_t s) {
  using G = StridedGeom<M1, EL>;
  with_nt(a.nt, [&](auto nt) {
    hipLaunchKernelGGL((strided_pass<M1, EL, INV, CANON, TW, CW, decltype(nt)::value>), dim3((1u << CW) >> G::CB, gy, gz), dim3(G::T), 0, s, a);
  });
  return GPQ_OK;
}

// strided pass over `polys` polynomials of each of a.nslab slabs, all `dim` limbs
template <bool INV>
int launch_strided(const gpq_ctx *c, const PassArgs &args, unsigned dim, unsigned polys, hipStream_t s) {
  // (forward and inverse passes of a limb must use the same class: the wide class keeps other lazy ranges between kernels)
  return for_limb_ranges<true>(c, args, dim, nullptr, nullptr, [&](auto tag, const PassArgs &a, unsigned limbs) {
    using TW = decltype(tag);
    ProfScope prof(c, INV ? GPQ_K_STRIDED_INV : GPQ_K_STRIDED_FWD, s);
    switch (c->logn) {
      case 13: return launch_strided_t<5, 4, INV, false, TW>(a, polys * a.nslab, limbs, s);
      case 14: return launch_strided_t<6, 4, INV, false, TW>(a, polys * a.nslab, limbs, s);
      case 15: return launch_strided_t<7, 4, INV, false, TW>(a, polys * a.nslab, limbs, s);
      case 16: return launch_strided_t<8, 4, INV, false, TW>(a, polys * a.nslab, limbs, s);
      case 17:   // 8 strided stages over 512-coefficient rows (the 256-row tiles of n = 2^16) + 9 low stages
        return launch_strided_t<8, 4, INV, false, TW, 9>(a, polys * a.nslab, limbs, s);
    }
    return gpq_fail(GPQ_ERR_INVALID, "two-pass NTT needs 13 <= logn <= 17 (got %u)", c->logn);
  });
}

template <bool INV>
int launch_contig(const gpq_ctx *c, const PassArgs &args, unsigned dim, unsigned polys, hipStream_t s) {
  if (args.nslab != 1) return gpq_fail(GPQ_ERR_INVALID, "contig_pass walks one slab");
  return for_limb_ranges<true>(c, args, dim, nullptr, nullptr, [&](auto tag, const PassArgs &a, unsigned limbs) {
    using TW = decltype(tag);
    ProfScope prof(c, INV ? GPQ_K_CONTIG_INV : GPQ_K_CONTIG_FWD, s);
    if (c->low9) {
      const dim3 grid9(c->n >> 11, (polys + CONTIG8_POLYS - 1) / CONTIG8_POLYS, limbs);
      with_nt(a.nt, [&](auto nt) { hipLaunchKernelGGL((contig_pass8<INV, TW, 9, decltype(nt)::value>), grid9, dim3(CONTIG_WAVES * 64), 0, s, a, polys); });
      return (int)GPQ_OK;
    }
    const dim3 grid(c->n >> 12, (polys + CONTIG_POLYS - 1) / CONTIG_POLYS, limbs);
    with_nt(a.nt, [&](auto nt) { hipLaunchKernelGGL((contig_pass<INV, TW, decltype(nt)::value>), grid, dim3(CONTIG_WAVES * 64), 0, s, a, polys); });
    return (int)GPQ_OK;
  });
}

template <bool INV>
int launch_small(const gpq_ctx *c, const PassArgs &args, unsigned dim, unsigned polys, hipStream_t s) {
  return for_limb_ranges<false>(c, args, dim, nullptr, nullptr, [&](auto tag, const PassArgs &a, unsigned limbs) {
    using TW = decltype(tag);
    const dim3 grid(1, polys * a.nslab, limbs);
    ProfScope prof(c, GPQ_K_SMALL, s);
    hipLaunchKernelGGL((small_ntt<INV, TW>), grid, dim3(256), 0, s, a);
    return (int)GPQ_OK;
  });
}

inline bool two_pass(const gpq_ctx *c) { return c->logn > (unsigned)SMALL_MAX_LOGN; }

int after_launch(const char *who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return gpq_fail(GPQ_ERR_HIP, "%s: launch failed: %s", who, hipGetErrorString(e));
  return GPQ_OK;
}

// The zero flags of a forward launch (tables.hpp: PassArgs::zflag), polys * dim words, all zero between calls (ref_zero_redo
// clears what it finds set).  An outgrown buffer is retired, not freed: a HIP graph captured earlier may still use it.
int zero_flags(gpq_ctx *c, size_t count, hipStream_t s, unsigned **out) {
  if (count > c->zflag_cap) {
    // (growing inside a stream capture would put the allocation into the graph: the first call at a new size runs outside capture, as for the CRT scratch)
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap_status) == hipSuccess && cap_status != hipStreamCaptureStatusNone)
      return gpq_fail(GPQ_ERR_INVALID, "the first gpq_ntt at a new batch size allocates its zero flags: run it once outside stream capture");
    DeviceScope on_device(c->device);
    unsigned *fresh = nullptr;
    const size_t cap = count < 4096 ? 4096 : count;
    HIP_TRY(hipMalloc((void **)&fresh, cap * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(fresh, 0, cap * sizeof(unsigned), s));
    if (c->d_zflag) c->retired.push_back(c->d_zflag);
    c->d_zflag = fresh; c->zflag_cap = cap;
  }
  *out = c->d_zflag;
  return GPQ_OK;
}

// src/ntt.c as written, one workgroup per limb (ref_zero_redo in ntt_kernels.hpp): mode 0 = redo the limbs the forward kernels flagged.
int launch_reference(const gpq_ctx *c, const PassArgs &a, unsigned dim, unsigned polys, unsigned mode, hipStream_t s) {
  const unsigned total = polys * dim;
  ProfScope prof(c, GPQ_K_REFERENCE, s);
  hipLaunchKernelGGL(ref_zero_redo, dim3(total < 1024 ? total : 1024), dim3(1024), 0, s, a, polys, dim, mode);
  return GPQ_OK;
}

// in-place forward transform of the slab given in a.src[0] = a.dst[0], in the reference's representation (a.zflag set)
int forward_slabs(gpq_ctx *c, PassArgs a, unsigned dim, unsigned polys, hipStream_t s) {
  int rc;
  if (!two_pass(c)) rc = launch_small<false>(c, a, dim, polys, s);
  else {
    if ((rc = launch_strided<false>(c, a, dim, polys, s)) != GPQ_OK) return rc;
    for (unsigned i = 0; i < a.nslab; ++i) a.src[i] = a.dst[i];
    rc = launch_contig<false>(c, a, dim, polys, s);
  }
  if (rc != GPQ_OK) return rc;
  // gpq_debug_zero_watch: what the forward kernels flagged, and what ref_zero_redo left, as copies on the launch stream
  const size_t words = (size_t)polys * dim;
  const bool watch = c->d_zwatch && a.zflag && words <= c->zwatch_cap;
  if (watch) HIP_TRY(hipMemcpyAsync(c->d_zwatch, a.zflag, words * sizeof(unsigned), hipMemcpyDeviceToDevice, s));
  if ((rc = launch_reference(c, a, dim, polys, 0, s)) != GPQ_OK) return rc;
  if (watch) {
    HIP_TRY(hipMemcpyAsync(c->d_zwatch + c->zwatch_cap, a.zflag, words * sizeof(unsigned), hipMemcpyDeviceToDevice, s));
    c->zwatch_count = words;
  }
  return GPQ_OK;
}

int inverse_slabs(const gpq_ctx *c, PassArgs a, unsigned dim, unsigned polys, hipStream_t s) {
  int rc;
  if (!two_pass(c)) return launch_small<true>(c, a, dim, polys, s);
  if ((rc = launch_contig<true>(c, a, dim, polys, s)) != GPQ_OK) return rc;
  for (unsigned i = 0; i < a.nslab; ++i) a.src[i] = a.dst[i];
  return launch_strided<true>(c, a, dim, polys, s);
}

}  // namespace


extern "C" int gpq_ntt(gpq_ctx *c, uint64_t *slab, unsigned dim, unsigned batch, void *stream) {
  int rc = check_shape(c, dim, batch, "gpq_ntt");
  if (rc) return rc;
  StageRange stage("gpq_ntt");
  if (!slab) return gpq_fail(GPQ_ERR_INVALID, "gpq_ntt: null slab");
  for (unsigned k0 = 0; k0 < batch; k0 += kMaxPolysPerLaunch) {
    const unsigned polys = batch - k0 < kMaxPolysPerLaunch ? batch - k0 : kMaxPolysPerLaunch;
    PassArgs a = make_args(c, dim, 1, nt_for(c, polys, dim, 1));
    a.src[0] = a.dst[0] = slab + (size_t)k0 * ((size_t)dim << c->logn);
    if ((rc = zero_flags(c, (size_t)polys * dim, (hipStream_t)stream, &a.zflag)) != GPQ_OK) return rc;
    a.zstride = dim;
    if ((rc = forward_slabs(c, a, dim, polys, (hipStream_t)stream)) != GPQ_OK) return rc;
  }
  return after_launch("gpq_ntt");
}

// Debug door of the zero watch (tests; HISTORY.md round 5): with the watch on, every gpq_ntt launch group leaves two copies of its flag words
// -- as the forward kernels wrote them and as ref_zero_redo left them -- so that "flag never set", "flag cleared early" and "redo wrong" can be
// told apart.  gpq_debug_zero_flags waits for the device and hands out the copies of the LAST launch group (polys * dim words each).
extern "C" int gpq_debug_zero_watch(gpq_ctx *c, int on) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_debug_zero_watch: null context");
  DeviceScope on_device(c->device);
  if (!on) {
    if (c->d_zwatch) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->d_zwatch)); }
    c->d_zwatch = nullptr; c->zwatch_cap = c->zwatch_count = 0;
    return GPQ_OK;
  }
  if (!c->d_zwatch) {
    const size_t want = (size_t)kMaxPolysPerLaunch * c->nprimes, cap = want < ((size_t)1 << 20) ? want : ((size_t)1 << 20);   // larger launch groups are not watched
    HIP_TRY(hipMalloc((void **)&c->d_zwatch, 2 * cap * sizeof(unsigned)));
    HIP_TRY(hipMemset(c->d_zwatch, 0xff, 2 * cap * sizeof(unsigned)));
    HIP_TRY(hipDeviceSynchronize());
    c->zwatch_cap = cap; c->zwatch_count = 0;
  }
  return GPQ_OK;
}
extern "C" long gpq_debug_zero_flags(gpq_ctx *c, unsigned *before, unsigned *after, size_t capacity) {
  if (!c || !c->d_zwatch) return -1;
  DeviceScope on_device(c->device);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  const size_t words = c->zwatch_count < capacity ? c->zwatch_count : capacity;
  if (before && hipMemcpy(before, c->d_zwatch, words * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (after && hipMemcpy(after, c->d_zwatch + c->zwatch_cap, words * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long)c->zwatch_count;
}

// src/ntt.c:37-73 executed as written (inverse = 0: ntt, else invntt) on every limb of the slab, for ANY input words --
// the kernel gpq_ntt redoes flagged limbs with.  Slow (one workgroup per limb, a barrier per stage); what the drop-in
// `ntt` uses for inputs outside [0, p), and an in-device cross-check of the two-pass kernels.
extern "C" int gpq_ntt_reference(gpq_ctx *c, uint64_t *slab, unsigned dim, unsigned batch, int inverse, void *stream) {
  int rc = check_shape(c, dim, batch, "gpq_ntt_reference");
  if (rc) return rc;
  if (!slab) return gpq_fail(GPQ_ERR_INVALID, "gpq_ntt_reference: null slab");
  for (unsigned k0 = 0; k0 < batch; k0 += kMaxPolysPerLaunch) {
    const unsigned polys = batch - k0 < kMaxPolysPerLaunch ? batch - k0 : kMaxPolysPerLaunch;
    PassArgs a = make_args(c, dim, 1);
    a.src[0] = a.dst[0] = slab + (size_t)k0 * ((size_t)dim << c->logn);
    if ((rc = launch_reference(c, a, dim, polys, inverse ? 2 : 1, (hipStream_t)stream)) != GPQ_OK) return rc;
  }
  return after_launch("gpq_ntt_reference");
}

extern "C" int gpq_invntt(gpq_ctx *c, uint64_t *slab, unsigned dim, unsigned batch, void *stream) {
  int rc = check_shape(c, dim, batch, "gpq_invntt");
  if (rc) return rc;
  StageRange stage("gpq_invntt");
  if (!slab) return gpq_fail(GPQ_ERR_INVALID, "gpq_invntt: null slab");
  for (unsigned k0 = 0; k0 < batch; k0 += kMaxPolysPerLaunch) {
    const unsigned polys = batch - k0 < kMaxPolysPerLaunch ? batch - k0 : kMaxPolysPerLaunch;
    PassArgs a = make_args(c, dim, 1, nt_for(c, polys, dim, 1));
    a.src[0] = a.dst[0] = slab + (size_t)k0 * ((size_t)dim << c->logn);
    if ((rc = inverse_slabs(c, a, dim, polys, (hipStream_t)stream)) != GPQ_OK) return rc;
  }
  return after_launch("gpq_invntt");
}

template <bool MUL>
static int pointwise_api(gpq_ctx *c, uint64_t *r, const uint64_t *x, const uint64_t *y, unsigned dim, unsigned batch,
                         void *stream, const char *who) {
  int rc = check_shape(c, dim, batch, who);
  if (rc) return rc;
  if (!r || !x || !y) return gpq_fail(GPQ_ERR_INVALID, "%s: null slab", who);
  const unsigned bx = c->n >= 512 ? c->n / 512 : 1;
  for (unsigned k0 = 0; k0 < batch; k0 += kMaxPolysPerLaunch) {
    const unsigned polys = batch - k0 < kMaxPolysPerLaunch ? batch - k0 : kMaxPolysPerLaunch;
    const size_t off = (size_t)k0 * ((size_t)dim << c->logn);
    PassArgs a = make_args(c, dim, 1);
    a.src[0] = x + off; a.src[1] = y + off; a.dst[0] = r + off;
    ProfScope prof(c, GPQ_K_POINTWISE, (hipStream_t)stream);
    hipLaunchKernelGGL((pointwise<MUL>), dim3(bx, polys, dim), dim3(256), 0, (hipStream_t)stream, a);
  }
  return after_launch(who);
}

extern "C" int gpq_rns_mul(gpq_ctx *c, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned dim, unsigned batch, void *stream) {
  return pointwise_api<true>(c, r, a, b, dim, batch, stream, "gpq_rns_mul");
}
extern "C" int gpq_rns_add(gpq_ctx *c, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned dim, unsigned batch, void *stream) {
  return pointwise_api<false>(c, r, a, b, dim, batch, stream, "gpq_rns_add");
}

extern "C" int gpq_poly_mul_rns(gpq_ctx *c, uint64_t *r, uint64_t *a, uint64_t *b, unsigned dim, unsigned batch, void *stream) {
  int rc = check_shape(c, dim, batch, "gpq_poly_mul_rns");
  if (rc) return rc;
  if (!r || !a || !b) return gpq_fail(GPQ_ERR_INVALID, "gpq_poly_mul_rns: null slab");
  if (!two_pass(c)) {
    if ((rc = gpq_ntt(c, a, dim, batch, stream))) return rc;
    if ((rc = gpq_ntt(c, b, dim, batch, stream))) return rc;
    if ((rc = gpq_rns_mul(c, r, a, b, dim, batch, stream))) return rc;
    return gpq_invntt(c, r, dim, batch, stream);
  }
  // three kernels instead of seven: strided forward pass on both operands in place, fused middle, strided inverse pass on r
  hipStream_t s = (hipStream_t)stream;
  const size_t poly = (size_t)dim << c->logn;
  for (unsigned k0 = 0; k0 < batch; k0 += kMaxPolysPerLaunch / 2) {
    const unsigned polys = batch - k0 < kMaxPolysPerLaunch / 2 ? batch - k0 : kMaxPolysPerLaunch / 2;
    const unsigned nt = nt_for(c, polys, dim, 3);
    PassArgs f = make_args(c, dim, 2, nt);
    f.src[0] = f.dst[0] = a + k0 * poly; f.src[1] = f.dst[1] = b + k0 * poly;
    if ((rc = launch_strided<false>(c, f, dim, polys, s))) return rc;
    PassArgs m = make_args(c, dim, 1, nt);
    m.src[0] = f.dst[0]; m.src[1] = f.dst[1]; m.dst[0] = r + k0 * poly;
    if ((rc = for_limb_ranges<true>(c, m, dim, nullptr, nullptr, [&](auto tag, const PassArgs &p, unsigned limbs) {
          using TW = decltype(tag);
          ProfScope prof(c, GPQ_K_CONTIG_FWD, s);
          with_nt(p.nt, [&](auto nt) {
            constexpr bool NT = decltype(nt)::value;
            if (c->low9) hipLaunchKernelGGL((polymul_mid8<TW, 9, NT>), dim3(c->n >> 11, polys, limbs), dim3(CONTIG_WAVES * 64), 0, s, p);
            else hipLaunchKernelGGL((polymul_mid8<TW, 8, NT>), dim3(c->n >> 11, polys, limbs), dim3(CONTIG_WAVES * 64), 0, s, p);
          });
          return (int)GPQ_OK;
        }))) return rc;
    PassArgs b2 = make_args(c, dim, 1, nt);
    b2.src[0] = b2.dst[0] = m.dst[0];
    if ((rc = launch_strided<true>(c, b2, dim, polys, s))) return rc;
  }
  return after_launch("gpq_poly_mul_rns");
}

// Limb loop of he_mulpt, src/he-mult.c:179-185: r0 = m (*) x0, r1 = m (*) x1 (negacyclic, per limb); m, x0, x1 are
// overwritten; r0 / r1 may be x0 / x1.
extern "C" int gpq_mulpt_rns(gpq_ctx *c, uint64_t *r0, uint64_t *r1, uint64_t *m, uint64_t *x0, uint64_t *x1, unsigned dim, unsigned batch,
                             void *stream) {
  int rc = check_shape(c, dim, batch, "gpq_mulpt_rns");
  if (rc) return rc;
  if (!r0 || !r1 || !m || !x0 || !x1) return gpq_fail(GPQ_ERR_INVALID, "gpq_mulpt_rns: null slab");
  if (!two_pass(c)) {
    if ((rc = gpq_ntt(c, m, dim, batch, stream)) || (rc = gpq_ntt(c, x0, dim, batch, stream)) || (rc = gpq_ntt(c, x1, dim, batch, stream))) return rc;
    if ((rc = gpq_rns_mul(c, r0, x0, m, dim, batch, stream)) || (rc = gpq_invntt(c, r0, dim, batch, stream))) return rc;
    if ((rc = gpq_rns_mul(c, r1, x1, m, dim, batch, stream))) return rc;
    return gpq_invntt(c, r1, dim, batch, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t poly = (size_t)dim << c->logn;
  for (unsigned k0 = 0; k0 < batch; k0 += kMaxPolysPerLaunch / 4) {
    const unsigned polys = batch - k0 < kMaxPolysPerLaunch / 4 ? batch - k0 : kMaxPolysPerLaunch / 4;
    const unsigned nt = nt_for(c, polys, dim, 5);
    PassArgs f = make_args(c, dim, 3, nt);
    f.src[0] = f.dst[0] = m + k0 * poly; f.src[1] = f.dst[1] = x0 + k0 * poly; f.src[2] = f.dst[2] = x1 + k0 * poly;
    if ((rc = launch_strided<false>(c, f, dim, polys, s))) return rc;
    PassArgs p = make_args(c, dim, 1, nt);
    for (int i = 0; i < 3; ++i) p.src[i] = f.dst[i];
    p.dst[0] = r0 + k0 * poly; p.dst[1] = r1 + k0 * poly;
    if ((rc = for_limb_ranges<true>(c, p, dim, nullptr, nullptr, [&](auto tag, const PassArgs &q, unsigned limbs) {
          using TW = decltype(tag);
          ProfScope prof(c, GPQ_K_CONTIG_FWD, s);
          with_nt(q.nt, [&](auto nt) {
            constexpr bool NT = decltype(nt)::value;
            if (c->low9) hipLaunchKernelGGL((mulpt_mid8<TW, 9, NT>), dim3(c->n >> 11, polys, limbs), dim3(CONTIG_WAVES * 64), 0, s, q);
            else hipLaunchKernelGGL((mulpt_mid8<TW, 8, NT>), dim3(c->n >> 11, polys, limbs), dim3(CONTIG_WAVES * 64), 0, s, q);
          });
          return (int)GPQ_OK;
        }))) return rc;
    PassArgs b2 = make_args(c, dim, 2, nt);
    b2.src[0] = b2.dst[0] = p.dst[0]; b2.src[1] = b2.dst[1] = p.dst[1];
    if ((rc = launch_strided<true>(c, b2, dim, polys, s))) return rc;
  }
  return after_launch("gpq_mulpt_rns");
}

// ---------------------------------------------------------------------------
// fused he_mul RNS core
// ---------------------------------------------------------------------------
static unsigned tensor_chunk(const gpq_ctx *c, unsigned batch) {
  return batch < c->chunk ? batch : c->chunk;
}
static unsigned limb_block(const gpq_ctx *c, unsigned dim) {
  return (c->limb_block && c->limb_block < dim) ? c->limb_block : dim;
}

extern "C" size_t gpq_tensor_workspace_bytes(const gpq_ctx *c, unsigned dim, unsigned batch) {
  if (!two_pass(c)) return 4ull * batch * ((size_t)dim << c->logn) * 8;
  return 4ull * tensor_chunk(c, batch) * ((size_t)dim << c->logn) * 8;
}
extern "C" size_t gpq_keyswitch_workspace_bytes(const gpq_ctx *c, unsigned dim, unsigned batch) {
  if (!two_pass(c)) return 1ull * batch * ((size_t)dim << c->logn) * 8;
  return 1ull * tensor_chunk(c, batch) * ((size_t)dim << c->logn) * 8;
}

extern "C" int gpq_he_mul_tensor(gpq_ctx *c, uint64_t *d0, uint64_t *d1, uint64_t *d2,
                                 const uint64_t *a0, const uint64_t *a1, const uint64_t *b0, const uint64_t *b1,
                                 unsigned dim, unsigned batch, void *workspace, void *stream) {
  int rc = check_shape(c, dim, batch, "gpq_he_mul_tensor");
  if (rc) return rc;
  if (!d0 || !d1 || !d2 || !a0 || !a1 || !b0 || !b1 || !workspace)
    return gpq_fail(GPQ_ERR_INVALID, "gpq_he_mul_tensor: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const size_t poly = (size_t)dim << c->logn;
  uint64_t *ws = (uint64_t *)workspace;
  const uint64_t *in[4] = {a0, a1, b0, b1};

  if (!two_pass(c)) {
    // small rings: unfused sequence, same dataflow as src/he-mult.c:121-136
    uint64_t *t[4];
    for (int i = 0; i < 4; ++i) {
      t[i] = ws + (size_t)i * batch * poly;
      HIP_TRY(hipMemcpyAsync(t[i], in[i], batch * poly * 8, hipMemcpyDeviceToDevice, s));
      if ((rc = gpq_ntt(c, t[i], dim, batch, stream))) return rc;
    }
    if ((rc = gpq_rns_mul(c, d0, t[0], t[2], dim, batch, stream))) return rc;   // c0*c0'
    if ((rc = gpq_rns_mul(c, d2, t[1], t[3], dim, batch, stream))) return rc;   // c1*c1'
    if ((rc = gpq_rns_mul(c, t[0], t[0], t[3], dim, batch, stream))) return rc; // c0*c1'
    if ((rc = gpq_rns_mul(c, t[1], t[1], t[2], dim, batch, stream))) return rc; // c1*c0'
    if ((rc = gpq_rns_add(c, d1, t[0], t[1], dim, batch, stream))) return rc;
    if ((rc = gpq_invntt(c, d0, dim, batch, stream))) return rc;
    if ((rc = gpq_invntt(c, d1, dim, batch, stream))) return rc;
    return gpq_invntt(c, d2, dim, batch, stream);
  }

  // Both operands the same ciphertext (he_mul(&ct, &ct, &ct, rlk), src/he-algo.c:151): two forward transforms instead of four
  const bool square = a0 == b0 && a1 == b1;
  const unsigned nin = square ? 2 : 4;
  const unsigned chunk = tensor_chunk(c, batch);
  const unsigned lblock = limb_block(c, dim);
  PeerLane lane;                                             // two launch groups in flight (engine_internal.hpp: gpq_peer_lane)
  if (batch > chunk && (rc = gpq_peer_lane(c, s, gpq_lane_key(3, dim, chunk, 0, 0, 0), [&](gpq_ctx *q) { return gpq_tensor_workspace_bytes(q, dim, chunk); }, &lane))) return rc;
  for (unsigned k0 = 0; k0 < batch; k0 += chunk) {
    const unsigned polys = (batch - k0 < chunk) ? batch - k0 : chunk;
    if (lane.c && ((k0 / chunk) & 1)) {
      const size_t o = k0 * poly;
      if ((rc = gpq_he_mul_tensor(lane.c, d0 + o, d1 + o, d2 + o, a0 + o, a1 + o, b0 + o, b1 + o, dim, polys, lane.ws, lane.s))) return rc;
      continue;
    }
    // A launch group = `polys` polynomials x `limbs` limbs of every slab: its three kernels run back to back so that
    // what one writes the next one reads while it is still in the Infinity Cache (gpq_set_limb_block).
    for (unsigned l0 = 0; l0 < dim; l0 += lblock) {
      const unsigned limbs = dim - l0 < lblock ? dim - l0 : lblock;
      const size_t loff = (size_t)l0 << c->logn;
      // 1. strided forward pass, inputs -> workspace
      StageRange stage("gpq_he_mul_tensor: strided fwd x4 / tensor_mid8 / strided inv x3");
      const unsigned nt = nt_for(c, polys, limbs, nin + 3);
      PassArgs f = make_args(c, dim, nin, nt);
      f.limb0 = l0;
      for (unsigned i = 0; i < nin; ++i) { f.src[i] = in[i] + k0 * poly + loff; f.dst[i] = ws + (size_t)i * chunk * poly + loff; }
      if ((rc = launch_strided<false>(c, f, limbs, polys, s))) return rc;
      // 2. low forward stages, products, low inverse stages -> outputs
      PassArgs m = make_args(c, dim, 1, nt);
      m.limb0 = l0;
      for (unsigned i = 0; i < nin; ++i) m.src[i] = f.dst[i];
      m.dst[0] = d0 + k0 * poly + loff; m.dst[1] = d1 + k0 * poly + loff; m.dst[2] = d2 + k0 * poly + loff;
      if ((rc = for_limb_ranges<true>(c, m, limbs, nullptr, nullptr, [&](auto tag, const PassArgs &a, unsigned nl) {
            using TW = decltype(tag);
            ProfScope prof(c, GPQ_K_TENSOR_MID, s);
            with_nt(a.nt, [&](auto nt) {
              constexpr bool NT = decltype(nt)::value;
              if (square && c->low9) hipLaunchKernelGGL((tensor_sq_mid8<TW, 9, NT>), dim3(c->n >> 11, polys, nl), dim3(CONTIG_WAVES * 64), 0, s, a);
              else if (square) hipLaunchKernelGGL((tensor_sq_mid8<TW, 8, NT>), dim3(c->n >> 11, polys, nl), dim3(CONTIG_WAVES * 64), 0, s, a);
              else if (c->low9) hipLaunchKernelGGL((tensor_mid8<TW, 9, NT>), dim3(c->n >> 11, polys, nl), dim3(CONTIG_WAVES * 64), 0, s, a);
              else hipLaunchKernelGGL((tensor_mid8<TW, 8, NT>), dim3(c->n >> 11, polys, nl), dim3(CONTIG_WAVES * 64), 0, s, a);
            });
            return (int)GPQ_OK;
          }))) return rc;
      // 3. strided inverse pass in place on the three outputs
      PassArgs b = make_args(c, dim, 3, nt);
      b.limb0 = l0;
      if (c->inv_tabs_override) b.tabs = c->inv_tabs_override;
      for (int i = 0; i < 3; ++i) { b.src[i] = m.dst[i]; b.dst[i] = m.dst[i]; }
      if ((rc = launch_strided<true>(c, b, limbs, polys, s))) return rc;
    }
  }
  const unsigned lanes_used = lane.c ? 2u : 1u;
  if ((rc = gpq_peer_join(c, s, lane))) return rc;
  c->last_lanes = lanes_used;   // (after the nested entry points of the groups, which record their own)
  return after_launch("gpq_he_mul_tensor");
}

extern "C" int gpq_keyswitch(gpq_ctx *c, uint64_t *c0, uint64_t *c1, const uint64_t *x,
                             const uint64_t *evk0, const uint64_t *evk1,
                             unsigned dim, unsigned batch, void *workspace, void *stream) {
  int rc = check_shape(c, dim, batch, "gpq_keyswitch");
  if (rc) return rc;
  if (!c0 || !c1 || !x || !evk0 || !evk1 || !workspace) return gpq_fail(GPQ_ERR_INVALID, "gpq_keyswitch: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const size_t poly = (size_t)dim << c->logn;
  uint64_t *ws = (uint64_t *)workspace;

  if (!two_pass(c)) {
    HIP_TRY(hipMemcpyAsync(ws, x, batch * poly * 8, hipMemcpyDeviceToDevice, s));
    if ((rc = gpq_ntt(c, ws, dim, batch, stream))) return rc;
    for (unsigned k = 0; k < batch; ++k) {  // the key is shared by the batch
      if ((rc = gpq_rns_mul(c, c0 + k * poly, ws + k * poly, evk0, dim, 1, stream))) return rc;
      if ((rc = gpq_rns_mul(c, c1 + k * poly, ws + k * poly, evk1, dim, 1, stream))) return rc;
    }
    if ((rc = gpq_invntt(c, c0, dim, batch, stream))) return rc;
    return gpq_invntt(c, c1, dim, batch, stream);
  }

  const unsigned chunk = tensor_chunk(c, batch);
  const unsigned lblock = limb_block(c, dim);
  PeerLane lane;
  if (batch > chunk && (rc = gpq_peer_lane(c, s, gpq_lane_key(4, dim, chunk, 0, 0, 0), [&](gpq_ctx *q) { return gpq_keyswitch_workspace_bytes(q, dim, chunk); }, &lane))) return rc;
  for (unsigned k0 = 0; k0 < batch; k0 += chunk) {
    const unsigned polys = (batch - k0 < chunk) ? batch - k0 : chunk;
    if (lane.c && ((k0 / chunk) & 1)) {
      const size_t o = k0 * poly;
      if ((rc = gpq_keyswitch(lane.c, c0 + o, c1 + o, x + o, evk0, evk1, dim, polys, lane.ws, lane.s))) return rc;
      continue;
    }
    for (unsigned l0 = 0; l0 < dim; l0 += lblock) {       // launch groups as in gpq_he_mul_tensor
      const unsigned limbs = dim - l0 < lblock ? dim - l0 : lblock;
      const size_t loff = (size_t)l0 << c->logn;
      StageRange stage("gpq_keyswitch: strided fwd / keyswitch_mid8x2 / strided inv x2");
      const unsigned nt = nt_for(c, polys, limbs, 3);
      PassArgs f = make_args(c, dim, 1, nt);
      f.limb0 = l0;
      f.src[0] = x + k0 * poly + loff; f.dst[0] = ws + loff;
      if ((rc = launch_strided<false>(c, f, limbs, polys, s))) return rc;
      KeyswitchArgs m;
      m.p = make_args(c, dim, 1, nt);
      m.p.limb0 = l0;
      m.p.src[0] = f.dst[0]; m.p.dst[0] = c0 + k0 * poly + loff; m.p.dst[1] = c1 + k0 * poly + loff;
      m.evk0 = evk0 + loff; m.evk1 = evk1 + loff;
      if ((rc = for_limb_ranges<true>(c, m.p, limbs, &m.evk0, &m.evk1, [&](auto tag, const PassArgs &a, unsigned nl) {
            using TW = decltype(tag);
            KeyswitchArgs ka{a, m.evk0, m.evk1};
            ProfScope prof(c, GPQ_K_KEYSWITCH_MID, s);
            const dim3 block(CONTIG_WAVES * 64);
            with_nt(a.nt, [&](auto nt) {
              constexpr bool NT = decltype(nt)::value;
              if (polys / 2) {
                if (c->low9) hipLaunchKernelGGL((keyswitch_mid8x2<TW, 9, true, NT>), dim3(c->n >> 11, polys / 2, nl), block, 0, s, ka, 0u);
                else hipLaunchKernelGGL((keyswitch_mid8x2<TW, 8, true, NT>), dim3(c->n >> 11, polys / 2, nl), block, 0, s, ka, 0u);
              }
              if (polys & 1) {           // the odd last polynomial alone (half the arithmetic of a pair that would be stored once)
                if (c->low9) hipLaunchKernelGGL((keyswitch_mid8x2<TW, 9, false, NT>), dim3(c->n >> 11, 1, nl), block, 0, s, ka, polys - 1);
                else hipLaunchKernelGGL((keyswitch_mid8x2<TW, 8, false, NT>), dim3(c->n >> 11, 1, nl), block, 0, s, ka, polys - 1);
              }
            });
            return (int)GPQ_OK;
          }))) return rc;
      PassArgs b = make_args(c, dim, 2, nt);
      b.limb0 = l0;
      if (c->inv_tabs_override) b.tabs = c->inv_tabs_override;
      for (int i = 0; i < 2; ++i) { b.src[i] = m.p.dst[i]; b.dst[i] = m.p.dst[i]; }
      if ((rc = launch_strided<true>(c, b, limbs, polys, s))) return rc;
    }
  }
  const unsigned lanes_used = lane.c ? 2u : 1u;
  if ((rc = gpq_peer_join(c, s, lane))) return rc;
  c->last_lanes = lanes_used;   // (after the nested entry points of the groups, which record their own)
  return after_launch("gpq_keyswitch");
}

// Which butterflies a limb runs is decided by its c = p - 2^59 (modarith.hpp): the first `wide` limbs the wide-split ones, the
// limbs up to `split` the split-twiddle ones, the rest the 7-mad ones.  The context picks the cheapest class each limb
// admits; this call can only move limbs towards the more general (slower) classes -- what the tests use to run every class on
// every limb and compare bit for bit.  Values above what the chain admits are clamped.
extern "C" int gpq_set_nt_policy(gpq_ctx *c, int mode) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_nt_policy: null context");
  if (mode < -1 || mode > 1) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_nt_policy: mode %d (want -1 by working set, 0 never, 1 always)", mode);
  c->nt_mode = mode;
  return GPQ_OK;
}

extern "C" int gpq_set_limb_classes(gpq_ctx *c, unsigned wide, unsigned split) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_limb_classes: null context");
  c->nsplit = split < c->nsplit_tables ? split : c->nsplit_tables;
  c->nwide = wide < c->nwide_max ? wide : c->nwide_max;
  if (c->nwide > c->nsplit) c->nwide = c->nsplit;
  return GPQ_OK;
}

extern "C" int gpq_set_limb_block(gpq_ctx *c, unsigned limbs) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_limb_block: null context");
  c->limb_block = limbs;
  return GPQ_OK;
}

extern "C" int gpq_set_chunk(gpq_ctx *c, unsigned chunk) {
  if (!c || chunk < 1 || chunk > kMaxPolysPerLaunch) return gpq_fail(GPQ_ERR_INVALID, "gpq_set_chunk: bad arguments");
  c->chunk = chunk;
  return GPQ_OK;
}

// ---------------------------------------------------------------------------
// per-kernel profile (HIP events around every launch, on the launch stream)
// ---------------------------------------------------------------------------
extern "C" int gpq_profile_enable(gpq_ctx *c, int on) {
  if (!c) return gpq_fail(GPQ_ERR_INVALID, "gpq_profile_enable: null context");
  c->prof_on = on != 0;
  return GPQ_OK;
}
// ---------------------------------------------------------------------------
// Yardstick kernels (bench.py): what this device's memory system delivers to a plain stream with the library's own access shape --
// 16 bytes per lane, persistent workgroups striding over the buffer, U independent accesses in flight per lane.  kind 0: copy (read + write),
// 1: read only, 2: write only.  The rate of the best configuration, measured in the same process as the kernels it is compared with, is the
// ceiling the HBM-bound kernels are priced against (`of_copy_rate` <= 1 by construction: a kernel that beats it IS the better copy).
// ---------------------------------------------------------------------------
namespace {
typedef unsigned v4w __attribute__((ext_vector_type(4)));
template <int U, int KIND>
__global__ void probe_stream(v4w *__restrict__ dst, const v4w *__restrict__ src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  v4w sink = v4w{0, 0, 0, 0};
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    v4w v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = KIND == 2 ? v4w{(unsigned)i, 1u, 2u, 3u} : __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (KIND == 1) sink ^= v[u];
      else __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
    }
  }
  for (; i < n16; i += stride) {
    if (KIND == 1) sink ^= src[i];
    else dst[i] = KIND == 2 ? v4w{(unsigned)i, 1u, 2u, 3u} : src[i];
  }
  if (KIND == 1 && (sink[0] ^ sink[1] ^ sink[2] ^ sink[3]) == 0x9e3779b9u) dst[0] = sink;     // (keeps the loads alive; practically never taken)
}
template <int KIND>
int launch_probe(void *dst, const void *src, size_t bytes, unsigned blocks, unsigned threads, unsigned unroll, hipStream_t s) {
  const size_t n16 = bytes / 16;
  switch (unroll) {
    case 1: hipLaunchKernelGGL((probe_stream<1, KIND>), dim3(blocks), dim3(threads), 0, s, (v4w *)dst, (const v4w *)src, n16); break;
    case 2: hipLaunchKernelGGL((probe_stream<2, KIND>), dim3(blocks), dim3(threads), 0, s, (v4w *)dst, (const v4w *)src, n16); break;
    case 4: hipLaunchKernelGGL((probe_stream<4, KIND>), dim3(blocks), dim3(threads), 0, s, (v4w *)dst, (const v4w *)src, n16); break;
    case 8: hipLaunchKernelGGL((probe_stream<8, KIND>), dim3(blocks), dim3(threads), 0, s, (v4w *)dst, (const v4w *)src, n16); break;
    default: return gpq_fail(GPQ_ERR_INVALID, "gpq_probe_stream: unroll must be 1, 2, 4 or 8");
  }
  return GPQ_OK;
}
}  // namespace

extern "C" int gpq_probe_stream(void *dst, const void *src, size_t bytes, int kind, unsigned blocks, unsigned threads, unsigned unroll, void *stream) {
  if (!dst || !src || bytes < 16 || (bytes & 15) || blocks < 1 || threads < 64 || threads > 1024 || (threads & 63) || kind < 0 || kind > 2)
    return gpq_fail(GPQ_ERR_INVALID, "gpq_probe_stream: bad arguments");
  int rc = kind == 0 ? launch_probe<0>(dst, src, bytes, blocks, threads, unroll, (hipStream_t)stream)
         : kind == 1 ? launch_probe<1>(dst, src, bytes, blocks, threads, unroll, (hipStream_t)stream)
                     : launch_probe<2>(dst, src, bytes, blocks, threads, unroll, (hipStream_t)stream);
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GPQ_OK : gpq_fail(GPQ_ERR_HIP, "gpq_probe_stream: launch failed: %s", hipGetErrorString(e));
}

extern "C" int gpq_profile_kernels(void) { return GPQ_K_COUNT; }
extern "C" const char *gpq_profile_kernel_name(int k) { return (k >= 0 && k < GPQ_K_COUNT) ? kKernelNames[k] : ""; }

// Waits for the recorded launches, adds their durations per kernel into
// total_ms[GPQ_K_COUNT] / launches[GPQ_K_COUNT], then forgets them.
extern "C" int gpq_profile_collect(gpq_ctx *c, double *total_ms, unsigned long long *launches) {
  if (!c || !total_ms || !launches) return gpq_fail(GPQ_ERR_INVALID, "gpq_profile_collect: bad arguments");
  for (int k = 0; k < GPQ_K_COUNT; ++k) { total_ms[k] = 0; launches[k] = 0; }
  for (gpq_prof_rec &r : c->prof) {
    float ms = 0;
    HIP_TRY(hipEventSynchronize(r.b));
    HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
    total_ms[r.kernel] += ms;
    launches[r.kernel] += 1;
    c->prof_pool.push_back(r.a);
    c->prof_pool.push_back(r.b);
  }
  c->prof.clear();
  return GPQ_OK;
}

// ---------------------------------------------------------------------------
// timers (HIP events recorded on the caller's stream)
// ---------------------------------------------------------------------------
struct gpq_timer { hipEvent_t a, b; };

extern "C" int gpq_timer_create(gpq_timer **t) {
  gpq_timer *x = new (std::nothrow) gpq_timer();
  if (!x) return gpq_fail(GPQ_ERR_NOMEM, "out of host memory");
  HIP_TRY(hipEventCreate(&x->a));
  HIP_TRY(hipEventCreate(&x->b));
  *t = x;
  return GPQ_OK;
}
extern "C" int gpq_timer_start(gpq_timer *t, void *stream) { HIP_TRY(hipEventRecord(t->a, (hipStream_t)stream)); return GPQ_OK; }
extern "C" int gpq_timer_stop(gpq_timer *t, void *stream) { HIP_TRY(hipEventRecord(t->b, (hipStream_t)stream)); return GPQ_OK; }
extern "C" int gpq_timer_elapsed_ms(gpq_timer *t, float *ms) {
  HIP_TRY(hipEventSynchronize(t->b));
  HIP_TRY(hipEventElapsedTime(ms, t->a, t->b));
  return GPQ_OK;
}
extern "C" void gpq_timer_destroy(gpq_timer *t) {
  if (!t) return;
  (void)hipEventDestroy(t->a);
  (void)hipEventDestroy(t->b);
  delete t;
}
