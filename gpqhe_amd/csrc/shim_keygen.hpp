// shim_keygen.hpp -- he_genswk / he_genrlk / he_genck / he_genrk (src/he-kem.c:74-170) with the reference's signatures.  Included inside mpi_shim.hip's extern "C" block.
// Part of the MPI-typed surface: one translation unit (mpi_shim.hip includes these fragments in order); split by concern in round 4.
#pragma once

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

