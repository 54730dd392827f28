// shim_additive.hpp -- he_add / he_sub / he_addpt / he_subpt / he_neg (src/he-add.c:32-140), he_rot / he_conj (src/he-automorphism.c:87-115) with the reference's signatures.  Included inside mpi_shim.hip's extern "C" block.
// Part of the MPI-typed surface: one translation unit (mpi_shim.hip includes these fragments in order); split by concern in round 4.
#pragma once

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
      if (ops.misfits) { (void)gpq_stream_sync(nullptr); return false; }   // (as in mpi_shim.hip: queued copies land before the buffers return to the pool)
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

