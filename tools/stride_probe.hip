// stride_probe -- dev tool (round 5): does the power-of-two 512 KiB stride between the limb rows a bridge wave reads cost anything?
// The access pattern of bridge_tail_stream / bridge_crt_decompose without their arithmetic: persistent workgroups, a wave owns a 1 KiB segment
// (64 lanes x 16 B, non-temporal) of a polynomial and walks ROWS limb rows of it, DEPTH loads in flight; row stride = 512 KiB + pad.
//   hipcc -O3 --offload-arch=gfx950 tools/stride_probe.hip -o tools/stride_probe && tools/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned v4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256) void walk(const v4 *__restrict__ base, v4 *__restrict__ out, size_t row16, size_t poly16, int rows, int tiles, int polys) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  for (size_t t = wave; t < (size_t)tiles * polys; t += nwaves) {
    const v4 *p = base + (t / tiles) * poly16 + (t % tiles) * 64 + lane;
    v4 acc = v4{0, 0, 0, 0};
    int r = 0;
    for (; r + DEPTH <= rows; r += DEPTH) {
      v4 x[DEPTH];
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) x[i] = __builtin_nontemporal_load(p + (size_t)(r + i) * row16);
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) acc ^= x[i];
    }
    for (; r < rows; ++r) acc ^= __builtin_nontemporal_load(p + (size_t)r * row16);
    __builtin_nontemporal_store(acc, out + t * 64 + lane);
  }
}

int main(int argc, char **argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 75, polys = argc > 2 ? atoi(argv[2]) : 64, logn = 16;
  const size_t row_bytes = (size_t)8 << logn;
  const int tiles = (int)(row_bytes / 1024);
  const size_t maxpad = 1 << 20;
  const size_t total = (size_t)polys * rows * (row_bytes + maxpad) + (64 << 20);
  v4 *buf, *out;
  CK(hipMalloc((void **)&buf, total));
  CK(hipMalloc((void **)&out, (size_t)polys * tiles * 1024));
  CK(hipMemset(buf, 1, total));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t pads[] = {0, 256, 512, 1024, 4096, 4096 + 256, 65536 + 256, 3 * 256, 131072 + 4096 + 256};
  for (int round = 0; round < 3; ++round)
    for (size_t pad : pads) {
      const size_t row16 = (row_bytes + pad) / 16, poly16 = row16 * rows;
      for (int blocks : {256 * 2, 256 * 4}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipEventRecord(a));
          hipLaunchKernelGGL((walk<5>), dim3(blocks), dim3(256), 0, 0, buf, out, row16, poly16, rows, tiles, polys);
          CK(hipEventRecord(b));
          CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b));
          if (rep && ms < best) best = ms;
        }
        printf("round %d  rows %d  pad %7zu B  blocks %4d: %.3f ms  %.1f GB/s read\n", round, rows, pad, blocks, best, (double)polys * rows * row_bytes / best / 1e6);
      }
    }
  return 0;
}
