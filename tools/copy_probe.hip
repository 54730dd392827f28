// copy_probe.hip -- how the big-slab transfers of the MPI-typed calls move over PCIe (dev tool): one polynomial = uint64_t[W][n],
// sent as (a) 16 two-dimensional copies of W rows x n/16 words (what upload_polys / download_issue do), (b) 16 linear copies of the
// same bytes, (c) one linear copy; both directions, page-locked host memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 65536, W = 14, ranges = 16, per = n / ranges, bytes = W * n * 8;
  const int polys = 4;
  char *h, *d;
  CHECK(hipHostMalloc((void **)&h, bytes * polys, hipHostMallocDefault));
  CHECK(hipMalloc((void **)&d, bytes * polys));
  for (int dir = 0; dir < 2; ++dir) {
    const hipMemcpyKind kind = dir ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
    for (int mode = 0; mode < 3; ++mode) {
      double best = 1e9;
      for (int rep = 0; rep < 20; ++rep) {
        CHECK(hipDeviceSynchronize());
        const double t0 = now_ms();
        for (int p = 0; p < polys; ++p) {
          char *hs = h + p * bytes, *ds = d + p * bytes;
          if (mode == 0)
            for (size_t t = 0; t < ranges; ++t)
              CHECK(hipMemcpy2DAsync((dir ? hs : ds) + t * per * 8, n * 8, (dir ? ds : hs) + t * per * 8, n * 8, per * 8, W, kind, nullptr));
          else if (mode == 1)
            for (size_t t = 0; t < ranges; ++t)
              CHECK(hipMemcpyAsync((dir ? hs : ds) + t * (bytes / ranges), (dir ? ds : hs) + t * (bytes / ranges), bytes / ranges, kind, nullptr));
          else
            CHECK(hipMemcpyAsync(dir ? hs : ds, dir ? ds : hs, bytes, kind, nullptr));
        }
        const double t1 = now_ms();
        CHECK(hipDeviceSynchronize());
        const double t2 = now_ms();
        if (t2 - t0 < best) best = t2 - t0;
        if (rep == 19) printf("%s %-28s %d polys: issue %.3f ms, total %.3f ms (best %.3f ms = %.1f GB/s)\n", dir ? "D2H" : "H2D",
                              mode == 0 ? "16 x 2D (14 rows x 32 KB)" : mode == 1 ? "16 x linear 448 KB" : "1 x linear 7 MB", polys, t1 - t0, t2 - t0, best, polys * bytes / best / 1e6);
      }
    }
  }
  return 0;
}
