// fuse_probe.hip -- memory-pattern probe for the fusion DESIGN.md 9 leaves open: a kernel that is limb-complete per coefficient (reads the 14
// words of a coefficient) AND holds a 256-row strided tile of the forward transform (writes 30 limbs of that tile), so that rns_decompose and
// the strided forward pass become one kernel and 60 words per coefficient leave HBM.  The tile must hold its input on chip: 256 rows x C
// columns x 14 words x 8 B = 28 KB x C -- C = 4 (32-byte segments) is what 160 KB of LDS admits.  This probe runs ONLY the memory pattern
// (no matrix-core product, no butterflies: a copy with that tiling) to see what rate the pattern itself allows, for C = 2 .. 16 (C > 4 keeps
// only what fits: pattern only), with workgroups that share 128-byte lines placed back to back on one XCD.
//   hipcc -O3 --offload-arch=gfx950 tools/fuse_probe.hip -o /tmp/fuse_probe && /tmp/fuse_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned LOGN = 16, N = 1u << LOGN, ROWS = 256, W = 14, LIMBS = 30;

template <int C, bool NT>
__global__ __launch_bounds__(256) void tile_copy(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, unsigned tiles_per_poly, unsigned total) {
  constexpr int KEEP = C <= 4 ? W : (4 * W) / C;                     // word planes kept in LDS (all of them when the tile fits)
  __shared__ uint64_t lds[KEEP * ROWS * C];
  // workgroups b, b + 8, b + 16, ... run on one XCD: give G = 16 / C consecutive ones of them the tiles that share 128-byte lines
  constexpr unsigned G = 16 / C;
  const unsigned b = blockIdx.x, xcd = b & 7, s = b >> 3;
  const unsigned T = (s / G) * (8 * G) + xcd * G + (s % G);
  if (T >= total) return;
  const unsigned poly = T / tiles_per_poly, c0 = (T % tiles_per_poly) * C;
  const unsigned t = threadIdx.x;
  constexpr unsigned PER = ROWS * C / 256;                           // coefficients per thread per plane
  const uint64_t *src = in + (size_t)poly * W * N;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    uint64_t v[PER];
#pragma unroll
    for (unsigned it = 0; it < PER; ++it) {
      const unsigned idx = it * 256 + t, r = idx / C, c = idx % C;
      const uint64_t *p = src + (size_t)w * N + r * 256 + c0 + c;
      v[it] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    if (w < KEEP) {
#pragma unroll
      for (unsigned it = 0; it < PER; ++it) lds[w * ROWS * C + it * 256 + t] = v[it];
    } else {                                                           // (planes that do not fit are folded into kept ones: every load stays live)
#pragma unroll
      for (unsigned it = 0; it < PER; ++it) lds[(w % KEEP) * ROWS * C + it * 256 + t] += v[it];
    }
  }
  __syncthreads();
  uint64_t *dst = out + (size_t)poly * LIMBS * N;
  for (int l = 0; l < (int)LIMBS; ++l) {
#pragma unroll
    for (unsigned it = 0; it < PER; ++it) {
      const unsigned idx = it * 256 + t, r = idx / C, c = idx % C;
      const uint64_t x = lds[(l % KEEP) * ROWS * C + it * 256 + t] + (unsigned)l;
      uint64_t *p = dst + (size_t)l * N + r * 256 + c0 + c;
      if (NT) __builtin_nontemporal_store(x, p); else *p = x;
    }
  }
}

template <int C, bool NT>
static void run(const uint64_t *in, uint64_t *out, unsigned polys) {
  const unsigned tiles_per_poly = 256 / C, total = polys * tiles_per_poly;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((tile_copy<C, NT>), dim3(total), dim3(256), 0, 0, in, out, tiles_per_poly, total);
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tile_copy<C, NT>), dim3(total), dim3(256), 0, 0, in, out, tiles_per_poly, total);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (double)polys * N * (W + LIMBS) * 8;
  printf("C = %2d columns (%3d-byte segments), %s: %.3f ms for %u polynomials = %.0f GB/s  (%.2f ms per 256 polynomials)\n", C, C * 8,
         NT ? "nt     " : "default", ms, polys, bytes / ms / 1e6, ms * 256 / polys);
}

int main() {
  const unsigned polys = 128;                                         // one launch group of gpq_he_mul: 32 ciphertexts x 4 polynomials
  uint64_t *in, *out;
  CHECK(hipMalloc(&in, (size_t)polys * W * N * 8)); CHECK(hipMalloc(&out, (size_t)polys * LIMBS * N * 8));
  CHECK(hipMemset(in, 1, (size_t)polys * W * N * 8));
  for (int round = 0; round < 2; ++round) {
    run<2, true>(in, out, polys); run<4, true>(in, out, polys); run<8, true>(in, out, polys); run<16, true>(in, out, polys);
    run<2, false>(in, out, polys); run<4, false>(in, out, polys); run<8, false>(in, out, polys); run<16, false>(in, out, polys);
  }
  printf("for comparison (profiles/r04/v11_bench.json, 256 polynomials): bridge_decompose 1.15-1.17 ms + the tensor share of the forward strided pass ~1.57 ms\n");
  return 0;
}
