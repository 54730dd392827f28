// convert_sanitized.cpp -- the host-only half of the MPI-typed surface (gpqhe_amd/csrc/mpi_convert.hpp) compiled for the CPU
// with AddressSanitizer + UndefinedBehaviorSanitizer and driven with real libgcrypt integers:
//   * MPI -> big slab -> MPI round trips (single- and multi-threaded) over signs, zeros, every width up to the slab's,
//     two's-complement extremes, and the word-major layout the device kernels read;
//   * the multiword helpers behind polyctx_init / hectx_init (mul_word, divmod_word, mul_words, shr1, bits_of, mpi_of,
//     words_of) against libgcrypt's own arithmetic.
// Prints "ok <checks>" or a line starting with "FAIL".  No GPU, no HIP: this is the -m "not gpu" tier.
#include "../../gpqhe_amd/csrc/mpi_convert.hpp"

extern "C" const char *gpq_last_error(void) { return ""; }

namespace {

uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

typedef int (*cmp_t)(MPI, MPI);
typedef void (*arith3_t)(MPI, MPI, MPI);
typedef void (*mpidiv_t)(MPI, MPI, MPI, MPI, int);
typedef void (*shift_t)(MPI, MPI, unsigned);

MPI random_mpi(uint64_t &st, unsigned bits, bool neg) {
  std::vector<unsigned char> buf((bits + 7) / 8 + 1, 0);
  for (auto &b : buf) b = (unsigned char)splitmix(st);
  if (bits % 8) buf[0] &= (unsigned char)((1u << (bits % 8)) - 1);
  MPI t = nullptr;
  if (G.mpi_scan(&t, FMT_USG, buf.data(), (bits + 7) / 8, nullptr)) die("scan");
  if (neg) G.mpi_neg(t, t);
  return t;
}

}  // namespace

int main() {
  need_gcrypt();
  cmp_t mpi_cmp = (cmp_t)dlsym(RTLD_DEFAULT, "gcry_mpi_cmp");
  arith3_t mpi_mul = (arith3_t)dlsym(RTLD_DEFAULT, "gcry_mpi_mul");
  mpidiv_t mpi_div = (mpidiv_t)dlsym(RTLD_DEFAULT, "gcry_mpi_div");
  shift_t mpi_rshift = (shift_t)dlsym(RTLD_DEFAULT, "gcry_mpi_rshift");
  if (!mpi_cmp || !mpi_mul || !mpi_div || !mpi_rshift) { printf("FAIL libgcrypt symbols\n"); return 1; }
  unsigned long checks = 0;
  uint64_t st = 20261004;

  // ---- round trips: n coefficients, W words, widths 0 .. 64 W - 1 bits, both signs ------------------------------
  // once reading / writing the limbs of struct gcry_mpi in place (the layout probe must have passed against this libgcrypt), once
  // through gcry_mpi_print / gcry_mpi_scan; the slabs of both paths must be the same words
  if (!g_mpi_direct) { printf("FAIL the struct gcry_mpi layout probe did not pass\n"); return 1; }
  for (int mode = 0; mode < 2; ++mode)
  for (unsigned n : {1u, 5u, 128u, 8192u}) {            // 8192 takes the worker threads (>= 4096)
    g_mpi_direct_wanted = mode == 0;
    for (unsigned W : {1u, 2u, 14u, 28u, 32u}) {
      poly_mpi_t a, r;
      a.coeffs = (gpq_MPI *)malloc(n * sizeof(gpq_MPI));
      r.coeffs = (gpq_MPI *)malloc(n * sizeof(gpq_MPI));
      for (unsigned i = 0; i < n; ++i) {
        unsigned bits = (unsigned)(splitmix(st) % (64 * W));          // 0 .. 64 W - 1
        if (i == 0) bits = 0;
        if (i == 1 % n) bits = 64 * W - 1;
        a.coeffs[i] = random_mpi(st, bits, splitmix(st) & 1);
        r.coeffs[i] = (i % 3 == 0) ? G.mpi_new(0)                         // nothing allocated yet: the direct path grows it through gcry_mpi_set_bit
                    : (i % 3 == 1) ? G.mpi_set_ui(G.mpi_new(0), 12345)   // stale content must be overwritten
                                   : random_mpi(st, 64 * 33, true);       // more limbs than the result needs, negative
      }
      if (n > 2) {                                                       // -2^(64W-1) itself: the most negative value a slab holds
        G.mpi_release(a.coeffs[2]);
        MPI one = G.mpi_set_ui(G.mpi_new(0), 1), big = G.mpi_new(0);
        typedef void (*lsh_t)(MPI, MPI, unsigned);
        ((lsh_t)dlsym(RTLD_DEFAULT, "gcry_mpi_lshift"))(big, one, 64 * W - 1);
        G.mpi_neg(big, big);
        G.mpi_release(one);
        a.coeffs[2] = big;
      }
      std::vector<uint64_t> slab((size_t)W * n, 0xdeadbeefdeadbeefull);
      if (n > 2 && G.mpi_get_nbits(a.coeffs[2]) == 64 * W) {
        // to_slab refuses 64 W bits of magnitude (it cannot tell -2^(64W-1) from +2^(64W-1)): convert that one by hand
        MPI keep = a.coeffs[2];
        a.coeffs[2] = G.mpi_set_ui(G.mpi_new(0), 0);
        to_slab(slab.data(), &a, n, W);
        for (unsigned j = 0; j < W; ++j) slab[(size_t)j * n + 2] = j + 1 == W ? 0x8000000000000000ull : 0;
        G.mpi_release(a.coeffs[2]);
        a.coeffs[2] = keep;
      } else {
        to_slab(slab.data(), &a, n, W);
        std::vector<uint64_t> other((size_t)W * n, 0x5555555555555555ull);
        g_mpi_direct_wanted = !g_mpi_direct_wanted;
        to_slab(other.data(), &a, n, W);
        g_mpi_direct_wanted = !g_mpi_direct_wanted;
        if (other != slab) { printf("FAIL direct and gcry_mpi_print slabs differ n=%u W=%u\n", n, W); return 1; }
        ++checks;
      }
      // layout: word j of coefficient i at j*n + i; sign bit = sign of the MPI (zero is non-negative)
      for (unsigned i = 0; i < n; ++i) {
        const bool neg = slab[(size_t)(W - 1) * n + i] >> 63;
        if (neg != (G.mpi_is_neg(a.coeffs[i]) && G.mpi_get_nbits(a.coeffs[i]) != 0)) { printf("FAIL sign n=%u W=%u i=%u\n", n, W, i); return 1; }
        ++checks;
      }
      {
        // the row layout the MPI-typed calls stage through (W words per coefficient): the same words as the word-major slab, transposed,
        // and a round trip of its own
        std::vector<uint64_t> rows((size_t)W * n, 0x1111111111111111ull);
        poly_mpi_t back;
        back.coeffs = (gpq_MPI *)malloc(n * sizeof(gpq_MPI));
        for (unsigned i = 0; i < n; ++i) back.coeffs[i] = G.mpi_set_ui(G.mpi_new(0), 7);
        if (!(n > 2 && G.mpi_get_nbits(a.coeffs[2]) == 64 * W)) {
          to_slab_range<true>(rows.data(), &a, n, W, 0, n);
          for (unsigned i = 0; i < n; ++i)
            for (unsigned j = 0; j < W; ++j)
              if (rows[(size_t)i * W + j] != slab[(size_t)j * n + i]) { printf("FAIL row layout n=%u W=%u i=%u j=%u\n", n, W, i, j); return 1; }
          from_slab_range<true>(&back, rows.data(), n, W, 0, n);
          for (unsigned i = 0; i < n; ++i)
            if (mpi_cmp(a.coeffs[i], back.coeffs[i]) != 0) { printf("FAIL row roundtrip n=%u W=%u i=%u\n", n, W, i); return 1; }
          checks += 2 * n;
        }
        for (unsigned i = 0; i < n; ++i) G.mpi_release(back.coeffs[i]);
        free(back.coeffs);
      }
      from_slab(&r, slab.data(), n, W);
      for (unsigned i = 0; i < n; ++i) {
        if (mpi_cmp(a.coeffs[i], r.coeffs[i]) != 0) { printf("FAIL roundtrip n=%u W=%u i=%u\n", n, W, i); return 1; }
        ++checks;
      }
      if (max_bits(&a, n) != 64 * W - (n > 2 ? 0 : 1) && n > 2) { printf("FAIL max_bits n=%u W=%u\n", n, W); return 1; }
      for (unsigned i = 0; i < n; ++i) { G.mpi_release(a.coeffs[i]); G.mpi_release(r.coeffs[i]); }
      free(a.coeffs); free(r.coeffs);
    }
  }

  // ---- multiword helpers against libgcrypt ---------------------------------------------------------------------------
  for (int t = 0; t < 200; ++t) {
    const unsigned bits_a = 1 + (unsigned)(splitmix(st) % 1800), bits_b = 1 + (unsigned)(splitmix(st) % 900);
    MPI A = random_mpi(st, bits_a, false), B = random_mpi(st, bits_b, false);
    if (!G.mpi_get_nbits(A) || !G.mpi_get_nbits(B)) { G.mpi_release(A); G.mpi_release(B); continue; }
    Words a = words_of(A, "a"), b = words_of(B, "b");
    if (bits_of(a) != G.mpi_get_nbits(A)) { printf("FAIL bits_of\n"); return 1; }
    MPI P = G.mpi_new(0);
    mpi_mul(P, A, B);
    MPI mine = mpi_of(mul_words(a, b));
    if (mpi_cmp(P, mine)) { printf("FAIL mul_words %d\n", t); return 1; }
    G.mpi_release(mine);
    uint64_t m = splitmix(st) | 1;
    Words am = a;
    mul_word(am, m);
    MPI M = G.mpi_set_ui(G.mpi_new(0), m), AM = G.mpi_new(0);
    mpi_mul(AM, A, M);
    mine = mpi_of(am);
    if (mpi_cmp(AM, mine)) { printf("FAIL mul_word %d\n", t); return 1; }
    G.mpi_release(mine);
    Words q = a;
    const uint64_t rem = divmod_word(q, m);
    MPI Q = G.mpi_new(0), R = G.mpi_new(0);
    mpi_div(Q, R, A, M, 0);                                  // positive operands: truncation == floor
    mine = mpi_of(q);
    MPI rm = G.mpi_set_ui(G.mpi_new(0), rem);
    if (mpi_cmp(Q, mine) || mpi_cmp(R, rm)) { printf("FAIL divmod_word %d\n", t); return 1; }
    G.mpi_release(mine); G.mpi_release(rm);
    Words h = a;
    shr1(h);
    MPI H = G.mpi_new(0);
    mpi_rshift(H, A, 1);
    mine = mpi_of(h);
    if (mpi_cmp(H, mine)) { printf("FAIL shr1 %d\n", t); return 1; }
    G.mpi_release(mine);
    checks += 5;
    for (MPI x : {A, B, P, M, AM, Q, R, H}) G.mpi_release(x);
  }
  if (powm64(3, 1ull << 40, 576460752303434497ull) != 1 && powm64(7, 576460752303434496ull, 576460752303434497ull) != 1) { printf("FAIL powm64\n"); return 1; }
  MPI z = mpi_of(Words());
  if (G.mpi_get_nbits(z) != 0) { printf("FAIL zero\n"); return 1; }
  G.mpi_release(z);
  g_mpi_direct_wanted = true;
  printf("ok %lu\n", checks);
  return 0;
}
