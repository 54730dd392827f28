// tables.hpp -- per-prime device tables and the argument block shared by the kernels.
#pragma once
#include "modarith.hpp"

namespace gpq {

struct LimbTab {           // per-prime scalars, array resident in HBM (read with scalar loads)
  PrimeK k;
  uint64_t ninv;           // n^-1 mod p, standard form (reference: rns->ninv is n^-1*2^64, src/precomp.c:248)
  uint64_t winv1_ninv;     // winv[1]*n^-1 mod p : last inverse stage with the scaling folded in
  TwS ninv_s, winv1_ninv_s;  // the same two as split-twiddle constants (modarith.hpp); unset for c >= GPQ_SPLIT_CMAX
};

#define GPQ_MAX_SLABS 4
struct PassArgs {
  const LimbTab *tabs;               // tabs[limb0 + blockIdx.z]
  // Twiddle tables [nprimes][n], standard form, indexed like rns->zetas / rns->zetas_inv
  // (src/precomp.c:255-263).  Kernel-argument pointers: the compiler knows they are
  // global memory, so uniform reads become s_load and the rest global_load.
  const uint64_t *w;
  const uint64_t *winv;
  // The same tables as split-twiddle pairs (modarith.hpp) for the leading limbs whose c allows it
  // ([nsplit][n]); a launch covers either split limbs only or plain limbs only.
  const TwS *ws;
  const TwS *winvs;
  const uint64_t *src[GPQ_MAX_SLABS];
  uint64_t *dst[GPQ_MAX_SLABS];
  unsigned long long poly_stride;    // elements between consecutive polynomials of a slab (= limbs_in_slab * n)
  unsigned logn;
  unsigned limb0;
  unsigned nslab;                    // blockIdx.y = poly * nslab + slab
  // Forward transforms that hand canonical output to the caller (gpq_ntt) note every (polynomial, limb) whose output
  // contains a residue 0: zflag[poly * zstride + blockIdx.z] |= 1.  Those limbs are redone by ref_zero_redo, which
  // reproduces the reference's representation of zero (src/ntt.c:47 stores p, not 0, for a sum x + t == p).
  unsigned *zflag;
  unsigned zstride;
  // Cache policy of the slab accesses of this launch: 1 = nt (streaming).  Read by the HOST launchers only, which pick the NT instantiation
  // of the kernel (engine.hip: with_nt, nt_for -- by the launch group's working set against the Infinity Cache).
  unsigned nt;
};

}  // namespace gpq
