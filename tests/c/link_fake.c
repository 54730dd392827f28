/* link_fake.c -- a stand-in for GPQHE's own libgpqhe.so in tests/test_link_order.py: STRONG definitions of the context /
 * storage names src/precomp.c and src/poly.c define (src/Makefile:55-58 builds them into libgpqhe.so), each recording that it
 * ran.  Own code: nothing of the reference is restated here beyond the two struct layouts of include/gpqhe_hip_compat.h. */
#include <stdlib.h>
#include <string.h>

#include "gpqhe_hip_compat.h"

struct poly_ctx polyctx;          /* src/precomp.c:41 */
struct he_ctx hectx;              /* src/precomp.c:47 */
static int calls[4];

void polyctx_init(unsigned int logn, gpq_MPI q) { (void)q; calls[0]++; memset(&polyctx, 0, sizeof polyctx); polyctx.logn = logn; polyctx.n = 1u << logn; polyctx.dimub = 3; }
void hectx_init(unsigned int logn, gpq_MPI q, unsigned int slots, uint64_t Delta) { (void)Delta; calls[1]++; polyctx_init(logn, q); hectx.slots = slots; hectx.dim = 2; }
void poly_rns_alloc(poly_rns_t *a, const unsigned int dim) { calls[2]++; a->coeffs = calloc((size_t)dim * polyctx.n, 8); }
void poly_rns_free(poly_rns_t *a) { calls[3]++; free(a->coeffs); a->coeffs = NULL; }
const int *fake_calls(void) { return calls; }
const void *fake_view(const char *name) { return strcmp(name, "polyctx") ? (const void *)&hectx : (const void *)&polyctx; }
