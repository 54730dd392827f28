// mfma_i8_probe.hip -- operand/result lane maps and signedness of v_mfma_i32_32x32x32_i8 on gfx950 (dev tool).
// A[i][k] = (i*7 + k*3) % 251 - 125 (signed), B[k][j] = (k*5 + j*11) % 241 - 120: asymmetric, exact in i32.
// Hypothesis (bf16 map scaled to 16 bytes per lane): lane l (r = l&31, h = l>>5) holds A[r][16h + t], B[16h + t][r], t = 0..15;
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__host__ __device__ inline int Aval(int i, int k) { return (i * 7 + k * 3) % 251 - 125; }
__host__ __device__ inline int Bval(int k, int j) { return (k * 5 + j * 11) % 241 - 120; }

__global__ void probe(int *out) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  union { v4i v; int8_t b[16]; } a, b;
  for (int t = 0; t < 16; ++t) { a.b[t] = (int8_t)Aval(r, 16 * h + t); b.b[t] = (int8_t)Bval(16 * h + t, r); }
  v16i c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
  for (int g = 0; g < 16; ++g) out[l * 16 + g] = c[g];
}

int main() {
  int *d; hipMalloc(&d, 64 * 16 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[64 * 16];
  if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 1; }
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int g = 0; g < 16; ++g) {
      const int col = l & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
      long ref = 0;
      for (int k = 0; k < 32; ++k) ref += (long)Aval(row, k) * Bval(k, col);
      if (ref != h[l * 16 + g]) { if (bad < 8) printf("mismatch lane %d reg %d: got %d want %ld\n", l, g, h[l * 16 + g], ref); ++bad; }
    }
  printf("mfma_i32_32x32x32_i8: %s (%d mismatches): signed x signed, A[r][16h+t], B[16h+t][r], C col=l&31 row=(g&3)+8(g>>2)+4(l>>5)\n", bad ? "HYPOTHESIS WRONG" : "hypothesis confirmed", bad);
  return bad != 0;
}
