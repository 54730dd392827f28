// engine_internal.hpp -- the context object behind the opaque gpq_ctx handle.
#pragma once
#include <vector>
#include "ntt_kernels.hpp"

struct gpq_ctx {
  int device = 0;
  unsigned logn = 0, n = 0, nprimes = 0;
  unsigned chunk = 4;  // polynomials per fused launch group: keeps the pass-to-pass scratch inside the 256 MiB Infinity Cache
  // host copies in the reference's own representation (struct rns_ctx, src/poly.h:28-41)
  std::vector<uint64_t> p, pinv_mont, pinv_barr, ninv_mont, psi;
  std::vector<uint64_t> zetas, zetas_inv;  // [nprimes][n], Montgomery form, bit-reversed index
  // device tables (standard form)
  uint64_t *d_w = nullptr, *d_winv = nullptr;
  gpq::LimbTab *d_tabs = nullptr;
};

int gpq_fail(int code, const char *fmt, ...);
