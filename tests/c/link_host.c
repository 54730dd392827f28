/* link_host.c -- which definitions of the reference's context / storage names a program and libgpqhe_hip.so end up with, for a
 * given link order (tests/test_link_order.py).  No GPU work: nothing here touches a device.
 *   link_host linked      the executable is linked against the fake GPQHE library and libgpqhe_hip.so (in the order under test)
 *   link_host dlopen <libgpqhe_hip.so>   the executable is linked against the fake only and opens the engine with RTLD_LOCAL */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "gpqhe_hip_compat.h"

extern struct poly_ctx polyctx;
extern struct he_ctx hectx;
void polyctx_init(unsigned int logn, gpq_MPI q);
void hectx_init(unsigned int logn, gpq_MPI q, unsigned int slots, uint64_t Delta);
void poly_rns_alloc(poly_rns_t *a, const unsigned int dim);
void poly_rns_free(poly_rns_t *a);
const int *fake_calls(void);
const void *fake_view(const char *name);
#ifdef LINKED
const void *gpq_compat_view(const char *name);
#endif

static const char *owner(const void *p)
{
  Dl_info info;
  if (!p || !dladdr(p, &info) || !info.dli_fname) return "?";
  const char *s = strrchr(info.dli_fname, '/');
  return s ? s + 1 : info.dli_fname;
}

int main(int argc, char **argv)
{
  const void *(*view)(const char *) = NULL;
#ifdef LINKED
  (void)argc; (void)argv;
  view = gpq_compat_view;
#else
  if (argc < 3) return 2;
  void *h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  view = (const void *(*)(const char *))dlsym(h, "gpq_compat_view");
  if (!view) return 1;
#endif
  hectx_init(5, NULL, 4, 1ull << 30);
  poly_rns_t a;
  poly_rns_alloc(&a, 2);
  poly_rns_free(&a);
  const int *c = fake_calls();
  printf("calls polyctx_init %d hectx_init %d poly_rns_alloc %d poly_rns_free %d\n", c[0], c[1], c[2], c[3]);
  printf("functions polyctx_init %s hectx_init %s poly_rns_alloc %s\n", owner(dlsym(RTLD_DEFAULT, "polyctx_init")),
         owner(dlsym(RTLD_DEFAULT, "hectx_init")), owner(dlsym(RTLD_DEFAULT, "poly_rns_alloc")));
  /* one object each for polyctx / hectx: the host's view, the fake GPQHE's view and the engine library's view must coincide */
  printf("polyctx host %d fake %d engine %d n %u\n", 1, fake_view("polyctx") == (const void *)&polyctx, view("polyctx") == (const void *)&polyctx, polyctx.n);
  printf("hectx host %d fake %d engine %d slots %u\n", 1, fake_view("hectx") == (const void *)&hectx, view("hectx") == (const void *)&hectx, hectx.slots);
  return 0;
}
