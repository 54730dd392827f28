/*
 * gpqhe_hip_compat.h -- the reference-named symbols libgpqhe_hip.so exports so
 * that GPQHE's own objects link against it unchanged (SURVEY.md section 8b:
 * the boundary is plain link-time C symbols of libgpqhe.so, src/Makefile:55-58).
 *
 * A GPQHE build that uses this library drops src/ntt.c and src/reduce.c from
 * SOURCES and the two pointwise functions from src/poly.c;
 * everything else of GPQHE keeps calling these names.
 *
 * The struct declarations restate the reference's memory layout only
 * (src/poly.h:28-65) with `MPI` spelled as the opaque pointer it is
 * (`typedef struct gcry_mpi *MPI`, src/types.h:47), so that this header does
 * not need <gcrypt.h>.  When compiling GPQHE itself include its own poly.h
 * instead -- the layouts are identical.
 */
#ifndef GPQHE_HIP_COMPAT_H
#define GPQHE_HIP_COMPAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef POLY_H /* GPQHE's own poly.h not seen: declare layout-compatible types */
typedef unsigned __int128 gpq_u128;
typedef void *gpq_MPI;

struct rns_ctx {               /* src/poly.h:28-41 */
  unsigned int dim;
  uint64_t p;
  uint64_t pinv_mont;
  uint64_t pinv_barr;
  uint64_t ninv;
  uint64_t *zetas;
  uint64_t *zetas_inv;
  gpq_MPI P;
  gpq_MPI P_2;
  gpq_MPI *phat;
  uint64_t *phat_invmp;
  struct rns_ctx *next;
};

struct ring_ctx {              /* src/poly.h:43-47 */
  unsigned int *cyc_group;
  _Complex double *zetas;
  double cM;
};

struct poly_ctx {              /* src/poly.h:49-65 */
  unsigned int logn;
  unsigned int n;
  unsigned int m;
  unsigned int logqub;
  unsigned int logq;
  gpq_MPI q;
  unsigned int logR;
  gpq_u128 R;
  gpq_u128 Rsub1;
  unsigned int dimub;
  struct rns_ctx *rns;
  struct ring_ctx ring;
};
#else
typedef u128 gpq_u128;
#endif

#ifndef GPQHE_H /* GPQHE's own gpqhe.h not seen: layout-compatible ciphertext / key / context types */
#ifndef POLY_H
typedef struct poly_mpi { gpq_MPI *coeffs; } poly_mpi_t;          /* src/poly.h:69-72 */
typedef struct poly_rns { uint64_t *coeffs; } poly_rns_t;         /* src/poly.h:74-77 */
#endif
struct bnd_ctx { double Bclean, Brs, Bks; double *Bmult; };       /* src/gpqhe.h:29-34 */
struct he_ctx {                                                   /* src/gpqhe.h:36-50 */
  gpq_MPI P, PqL;
  gpq_MPI *q, *qh;
  gpq_MPI p;
  unsigned int slots;
  double Delta;
  unsigned int L, dim, dimevk;
  struct bnd_ctx bnd;
  double Bmult;
};
typedef struct he_evk { poly_rns_t p0, p1; } he_evk_t;            /* src/gpqhe.h:78-82 */
typedef struct he_ct { unsigned int l; double nu, B; poly_mpi_t c0, c1; } he_ct_t;   /* src/gpqhe.h:84-91 */
typedef struct he_pt { double nu; poly_mpi_t m; } he_pt_t;                            /* src/gpqhe.h:93-97 */
#endif

/* src/ntt.c:37,54 -- in place on one limb of n = polyctx.n coefficients (host
 * memory); n is read from the global `polyctx` like the reference does
 * (src/ntt.c:26,42).  Synchronous: the limb goes to the GPU and back. */
void ntt(uint64_t a[], const struct rns_ctx *rns);
void invntt(uint64_t a[], const struct rns_ctx *rns);
/* north-star spellings of the same two functions (BASELINE.json) */
void poly_ntt(uint64_t a[], const struct rns_ctx *rns);
void poly_invntt(uint64_t a[], const struct rns_ctx *rns);

/* src/poly.c:71-82, declared src/poly.h:84-85.  rhat may alias ahat/bhat
 * (src/he-mult.c:130,183). */
void poly_rns_add(uint64_t rhat[], const uint64_t ahat[], const uint64_t bhat[], const struct rns_ctx *rns);
void poly_rns_mul(uint64_t rhat[], const uint64_t ahat[], const uint64_t bhat[], const struct rns_ctx *rns);

/* src/reduce.c:36,59,75,88 -- scalar host helpers used by precomp.c at init. */
uint64_t montgomery_inv(uint64_t q);
uint64_t montgomery_reduce(gpq_u128 a, uint64_t q, int64_t qinv);
uint64_t barrett_inv(uint64_t q);
uint64_t barrett_reduce(gpq_u128 a, uint64_t q, uint64_t qinv);

/* ---- MPI-typed surface (mpi_shim.hip) ---------------------------------------------------
 * The five names BASELINE.json's north_star lists and the CRT bridge of src/rns.c, with the reference's own signatures and its
 * libgcrypt types.  Coefficients travel MPI -> device big slab -> MPI through libgcrypt's runtime
 * ABI (gcry_mpi_print / gcry_mpi_scan, resolved with dlsym from the libgcrypt the host program
 * already links).  `polyctx` and `hectx` are read like the reference reads them.  Power-of-two moduli
 * (every parameter set of the reference's tests) take the tuned kernels, any other q / q_l / 64-bit
 * Delta the general ones; misuse aborts with the reference's error convention -- there is no CPU path. */
void rns_decompose(uint64_t ahat[], const gpq_MPI a[], const struct rns_ctx *rns);      /* src/rns.c:37: one limb, ahat[i] = a[i] mod rns->p */
void rns_reconstruct(gpq_MPI a, const uint64_t ahat[], const unsigned int i,
                     const struct rns_ctx *rns);                                        /* src/rns.c:60: ONE coefficient, in [0, P); use poly_rns2mpi for a slab */
void poly_rns2mpi(poly_mpi_t *r, const poly_rns_t *rhat, const struct rns_ctx *rns,
                  const gpq_MPI q);                                                     /* src/poly.h:88 */
void poly_mul(poly_mpi_t *r, const poly_mpi_t *a, const poly_mpi_t *b,
              const unsigned int dim, const gpq_MPI q);                                 /* src/poly.h:86-87 */
void he_mul(he_ct_t *ct, const he_ct_t *ct1, const he_ct_t *ct2, const he_evk_t *rlk);  /* src/gpqhe.h:147  */
void he_rs(struct he_ct *ct);                                                           /* src/gpqhe.h:136  */
void he_rescale(struct he_ct *ct);                                                      /* north-star name of he_rs */
void he_moddown(he_ct_t *ct);                                                           /* src/gpqhe.h:137  */
void he_mulpt(struct he_ct *dest, const struct he_ct *src, const struct he_pt *pt);     /* src/gpqhe.h:148  */
/* src/he-add.c:32-140 (decl src/gpqhe.h:140-144): big-integer work only, but interleaved with every product in GPQHE's algorithms
 * (src/he-algo.c:146-155) -- on the device they keep a chain's ciphertexts resident.  A GPQHE build that wants them drops he-add.c. */
void he_add(he_ct_t *ct, const he_ct_t *ct1, const he_ct_t *ct2);                       /* src/gpqhe.h:140  */
void he_sub(he_ct_t *ct, const he_ct_t *ct1, const he_ct_t *ct2);                       /* src/gpqhe.h:141  */
void he_addpt(he_ct_t *dest, const he_ct_t *src, const he_pt_t *pt);                    /* src/gpqhe.h:142  */
void he_subpt(he_ct_t *dest, const he_ct_t *src, const he_pt_t *pt);                    /* src/gpqhe.h:143  */
void he_neg(he_ct_t *ct);                                                               /* src/gpqhe.h:144  */
/* src/he-mem.c:88-97: 2n mpi_set on the host in the reference; here the copy's device slab is a copy of the source's, so the calls that
 * follow on the copy (he_gemv: he_copy_ct, he_rot, he_mulpt, he_add per diagonal, src/he-algo.c:66-78) start on the device.  Optional
 * (guard the one function in he-mem.c). */
void he_copy_ct(struct he_ct *dest, const struct he_ct *src);
/* src/he-encrypt.c:105-125 (decl src/gpqhe.h:130): m = c1 * sk + c0 centred mod q_l -- the caller of poly_mul at the end of every
 * computation; the sum and its centring stay on the device (2n libgcrypt calls on the host in the reference).  Optional. */
void he_dec(struct he_pt *pt, const struct he_ct *ct, const poly_mpi_t *sk);
void he_conj(he_ct_t *ct, const he_evk_t *ck);                                          /* src/gpqhe.h:151  */
void he_rot(he_ct_t *ct, const int rot, const he_evk_t *rk);                            /* src/gpqhe.h:152  */
/* Key generation, src/he-kem.c:120-170 (decl src/gpqhe.h:131-133).  The randomness comes from the host program's own
 * sample_error / sample_uniform (src/sample.c), called in the reference's order; everything else runs on the device. */
void he_genrlk(he_evk_t *rlk, const poly_mpi_t *sk);
void he_genck(he_evk_t *ck, const poly_mpi_t *sk);
void he_genrk(he_evk_t *rk, const poly_mpi_t *sk);

/* ---- context construction and polynomial storage -------------------------------------------------------------------
 * polyctx_init / hectx_init / poly_*_alloc and the data symbols polyctx, hectx, GPQHE_TWO are GPQHE's own (src/precomp.c,
 * src/poly.c:46-69).  libgpqhe_hip.so does not define them (it only reads `polyctx` / `hectx`), so it can sit anywhere on a
 * GPQHE link line.  A host that is not GPQHE gets them from libgpqhe_hip_ctx.so: include/gpqhe_hip_ctx.h. */
struct gpq_ctx;
/* The chain of `struct rns_ctx` nodes (src/poly.h:28-41) for the first `count` primes of an engine context, in a
 * caller-owned array: scalars and table pointers always (tables stay owned by `ctx`), phat_invmp malloc'ed, and with
 * with_mpi != 0 the libgcrypt integers P, P_2, phat[] (src/precomp.c:266-293).  Returns GPQ_OK or GPQ_ERR_INVALID. */
int gpq_fill_rns_chain(struct rns_ctx *nodes, unsigned count, const struct gpq_ctx *ctx, int with_mpi);
void gpq_release_rns_chain(struct rns_ctx *nodes);

/* The engine context the MPI-typed calls use for the caller's `polyctx` (built on first use; ctx_compat.hip takes its prime chain from it). */
struct gpq_ctx *gpq_mpi_shim_engine(void);
/* Address of `polyctx` / `hectx` as this library is bound to them (NULL when no object of the process defines them). */
const void *gpq_compat_view(const char *name);

/* When the host program has no `polyctx` symbol (the library references it
 * weakly), the ring degree for the drop-in calls is set here instead. */
void gpq_dropin_set_logn(unsigned int logn);
/* Releases the device tables the drop-in calls cached (per rns_ctx). */
void gpq_dropin_reset(void);
/* Wall milliseconds of the last he_mul(he_ct_t *, ...) call: [0] MPI -> slab conversions + uploads, [1] device kernels (HIP
 * events), [2] downloads + slab -> MPI conversions (includes waiting for [1]), [3] the whole call. */
void gpq_mpi_shim_last_timing(double ms[4]);
/* Number of evaluation keys the MPI-typed calls keep on the device between calls (default 64, least recently used out; lowering
 * the number evicts at once). */
void gpq_mpi_shim_set_key_slots(unsigned slots);
/* How a resident key is recognised: by the caller's two pointers and a fingerprint of EVERY word, limb by limb (full != 0, the
 * default: a key edited in place multiplies as edited, like the reference, which reads its key on every call), computed by the
 * conversion threads while the device works; a call at a lower level, which reads fewer limbs of the same key (src/he-mult.c:51),
 * is served by the same copy, checked over the limbs it reads.  Or by ~1000 sampled words of the exact length in use (full == 0),
 * for programs that never edit a key in place. */
void gpq_mpi_shim_set_key_check(int full);
unsigned gpq_mpi_shim_resident_keys(void);
/* How coefficients cross between libgcrypt integers and big slabs: on != 0 (default) reads and writes the limbs of `struct gcry_mpi`
 * in place, after a probe through libgcrypt's public API has confirmed the layout in this process (3x faster calls at n = 2^16);
 * 0 goes through gcry_mpi_print / gcry_mpi_scan for every coefficient.  Returns 1 if the direct path is in use afterwards. */
int gpq_mpi_shim_set_direct_mpi(int on);
/* Resident polynomials.  GPQHE chains its calls on one ciphertext (he_mul(&bn, &bn, &bn, rlk); he_rs(&bn); ..., src/he-algo.c:140-160):
 * the device keeps the slab of every polynomial the MPI-typed calls have read or written (`slots` of them, default 32, least recently
 * used out; 0 = none), identified by the caller's coefficient array, the shape and a fingerprint of every word of every coefficient.
 * Operands that are resident are not converted before the device starts: the device works from the copies at once, and the conversion
 * threads meanwhile convert and fingerprint the caller's integers exactly as an upload would; operands the caller changed since are
 * uploaded from the rows then already staged and the device work runs again (and such an operand is converted up front next time).  Results never depend on a stale copy; only with the direct integer access
 * (gpq_mpi_shim_set_direct_mpi) and n >= 4096. */
void gpq_mpi_shim_set_poly_slots(unsigned slots);
unsigned gpq_mpi_shim_resident_polys(void);
/* operands served from a resident copy that the check confirmed / found changed (uploaded again, device work repeated) */
void gpq_mpi_shim_poly_stats(uint64_t *confirmed, uint64_t *stale);
void gpq_mpi_shim_forget_polys(void);
/* testing: while on, calls neither consult nor update the resident polynomials (which stay as they are) */
void gpq_mpi_shim_poly_bypass(int on);
/* Host threads converting between libgcrypt integers and slabs: default min(hardware threads, 16), at most 64.  Only before the first
 * MPI-typed call of the process; returns the number in use. */
unsigned gpq_mpi_shim_set_conversion_threads(unsigned threads);
/* Drops the device copies of the evaluation keys.  Never needed with the default key check; he_genrlk / he_genck / he_genrk drop
 * the copy of the key they write themselves. */
void gpq_mpi_shim_forget_keys(void);
/* The MPI-typed entry points keep staging buffers, device buffers, the key cache and a pool of conversion threads between calls:
 * they serialise on one lock (the reference itself is single-threaded); calls from several host threads are safe and run one
 * after another. */
/* Releases the device buffers and the engine context the MPI-typed calls keep between calls. */
void gpq_mpi_shim_release(void);

#ifdef __cplusplus
}
#endif
#endif /* GPQHE_HIP_COMPAT_H */
