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

// HALF: the tail kernel's real shape -- a wave-level load covers TWO 512-byte segments (lanes 0..31 limb row 4s + 0/1, lanes 32..63 limb row 4s + 2/3:
// bridge_stream.hpp fetch_step), a k step = two such loads -- against the plain 1 KiB-per-row walk above.
template <int DEPTH>
__global__ __launch_bounds__(512) void walk_half(const v4 *__restrict__ base, v4 *__restrict__ out, size_t row16, size_t poly16, int rows, int tiles, int polys) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const int steps = rows / 4;
  for (size_t t = wave; t < (size_t)tiles * 2 * polys; t += nwaves) {         // a tile = 64 coefficients = 512 bytes of a row
    const v4 *p = base + (t / (tiles * 2)) * poly16 + (t % (tiles * 2)) * 32 + r + (size_t)(2 * h) * row16;
    v4 acc = v4{0, 0, 0, 0};
    int s = 0;
    for (; s + DEPTH <= steps; s += DEPTH) {
      v4 x[DEPTH], y[DEPTH];
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) { x[i] = __builtin_nontemporal_load(p + (size_t)(4 * (s + i)) * row16); y[i] = __builtin_nontemporal_load(p + (size_t)(4 * (s + i) + 1) * row16); }
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) acc ^= x[i] ^ y[i];
    }
    for (; s < steps; ++s) acc ^= __builtin_nontemporal_load(p + (size_t)(4 * s) * row16) ^ __builtin_nontemporal_load(p + (size_t)(4 * s + 1) * row16);
    if (lane < 32) __builtin_nontemporal_store(acc, out + t * 32 + r);
  }
}

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
        if (pad == 0 || pad == 256) {
          const int rows4 = rows / 4 * 4;
          float bh = 1e9f;
          for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((walk_half<5>), dim3(blocks / 2), dim3(512), 0, 0, buf, out, row16, poly16, rows4, tiles, polys);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < bh) bh = ms;
          }
          printf("round %d  rows %d  pad %7zu B  workgroups %4d x 512 threads, 512-byte segments on two rows per load (the tail's shape): %.3f ms  %.1f GB/s read\n", round, rows4, pad, blocks / 2, bh, (double)polys * rows4 * row_bytes / bh / 1e6);
        }
      }
    }
  return 0;
}
