/*
 * gpqhe_hip_ctx.h -- libgpqhe_hip_ctx.so: the reference's context-construction and storage names for hosts that are NOT
 * GPQHE (a GPQHE build keeps its own src/precomp.c / src/poly.c and never links this library).  Link after the engine:
 *     cc host.c -lgpqhe_hip -lgpqhe_hip_ctx -l:libgcrypt.so.20
 * The prime chain, constants and tables come from the engine (bit-identical with src/precomp.c:244-293, pinned in tests/),
 * the MPI fields are built through libgcrypt's runtime ABI.
 */
#ifndef GPQHE_HIP_CTX_H
#define GPQHE_HIP_CTX_H

#include "gpqhe_hip_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

extern struct poly_ctx polyctx;                                                          /* src/precomp.c:41 */
extern struct he_ctx hectx;                                                              /* src/precomp.c:47 */
extern gpq_MPI GPQHE_TWO;                                                                /* src/precomp.c:37 */
void polyctx_init(unsigned int logn, gpq_MPI q);                                        /* src/poly.h:94  */
void polyctx_exit(void);                                                                /* src/poly.h:95  */
void hectx_init(unsigned int logn, gpq_MPI q, unsigned int slots, uint64_t Delta);      /* src/gpqhe.h:100 */
void hectx_exit(void);                                                                  /* src/gpqhe.h:101 */
void poly_mpi_alloc(poly_mpi_t *a);                                                     /* src/poly.h:80  */
void poly_mpi_free(poly_mpi_t *a);                                                      /* src/poly.h:81  */
void poly_rns_alloc(poly_rns_t *a, const unsigned int dim);                             /* src/poly.h:82  */
void poly_rns_free(poly_rns_t *a);                                                      /* src/poly.h:83  */

#ifdef __cplusplus
}
#endif
#endif /* GPQHE_HIP_CTX_H */
