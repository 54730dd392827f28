// shim_chain.hpp -- gpq_fill_rns_chain / gpq_release_rns_chain and the shim's settings and statistics entry points.
// Part of the MPI-typed surface: one translation unit (mpi_shim.hip includes these fragments in order); split by concern in round 4.
#pragma once

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

