// instr_rate.hip -- issue rate of single VALU instructions on gfx950 (inline asm,
// 8 independent chains per lane, every SIMD saturated).  Dev tool.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
constexpr int ITER = 2048;

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ __launch_bounds__(256) void probe(uint64_t *out, uint64_t seed) {
  uint64_t v[8];
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = seed * (threadIdx.x + 1 + 64 * i) + blockIdx.x; a[i] = (uint32_t)(v[i] >> 13) | 1; }
  uint64_t k64 = seed | 1;
  uint32_t k32 = (uint32_t)seed | 1;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(v[i]) : "v"(k64));
      if (OP == 2) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(v[i]) : "v"(a[i]), "v"(k32) : "s20", "s21");
      if (OP == 3) asm volatile("v_cmp_le_u64 vcc, %1, %0\n\tv_cndmask_b32 %2, %2, %3, vcc" : "+v"(v[i]), "+v"(k64), "+v"(a[i]) : "v"(k32) : "vcc");
      if (OP == 4) asm volatile("v_alignbit_b32 %0, %0, %1, 27" : "+v"(a[i]) : "v"(k32));
      if (OP == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 7) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(k32) : "v"(a[(i + 1) & 7]) : "vcc");
      if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(k32) : "vcc");
      if (OP == 9) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 10) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 11) asm volatile("v_lshlrev_b64 %0, 5, %0" : "+v"(v[i]));
      if (OP == 12) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 13) asm volatile("v_cmp_gt_i32 vcc, 0, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(k32) : "vcc");
      if (OP == 14) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 15) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, 0" : "+v"(v[i]) : "v"(a[i]), "v"(k32) : "s20", "s21");
      if (OP == 16) asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n\tv_subb_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(k32) : "v"(a[(i + 1) & 7]) : "vcc");
      if (OP == 17) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 18) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 19) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k32));
      if (OP == 20) asm volatile("v_cmp_le_u64 s[22:23], %0, %1" : : "v"(v[i]), "v"(k64) : "s22", "s23");
      if (OP == 21) asm volatile("v_cmp_le_u32 s[22:23], %0, %1" : : "v"(a[i]), "v"(k32) : "s22", "s23");
      if (OP == 22) asm volatile("v_cndmask_b32 %0, %0, %1, s[24:25]" : "+v"(a[i]) : "v"(k32));
      if (OP == 23) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x0c" : "+v"(a[i]) : "v"(k32));
      if (OP == 24) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
      if (OP == 25) asm volatile("v_mov_b64 %0, %1" : "+v"(v[i]) : "v"(k64));
    }
  }
  uint64_t acc = k64 + k32;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= v[i] + a[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
int run(const char *name, int instr_per, uint64_t *d_out) {
  const int blocks = 256 * 8;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 0x9e3779b97f4a7c15ull);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 0x9e3779b97f4a7c15ull + r);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  double groups = 5.0 * blocks * 256.0 * ITER * 8;       // asm statements executed (per lane)
  double slots = (ms * 1e-3) * (256.0 * 128 * 2.4e9) / groups;  // lane-cycles at 2.4 GHz per statement
  printf("%-34s %8.3f ms  %6.2f lane-cycles per statement (%d instr) -> %.2f per instr\n", name, ms / 5, slots, instr_per, slots / instr_per);
  return 0;
}

int main() {
  uint64_t *d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 8));
  run<0>("v_add_u32", 1, d_out);
  run<1>("v_lshl_add_u64", 1, d_out);
  run<2>("v_mad_u64_u32 (acc)", 1, d_out);
  run<15>("v_mad_u64_u32 (+0)", 1, d_out);
  run<3>("v_cmp_le_u64 + v_cndmask", 2, d_out);
  run<13>("v_cmp_gt_i32 + v_cndmask", 2, d_out);
  run<4>("v_alignbit_b32", 1, d_out);
  run<5>("v_and_b32", 1, d_out);
  run<6>("v_mov_b32", 1, d_out);
  run<7>("v_add_co_u32 + v_addc_co_u32", 2, d_out);
  run<16>("v_sub_co_u32 + v_subb_co_u32", 2, d_out);
  run<8>("v_cndmask_b32", 1, d_out);
  run<9>("v_mul_hi_u32", 1, d_out);
  run<10>("v_mul_u32_u24", 1, d_out);
  run<18>("v_mad_u32_u24", 1, d_out);
  run<11>("v_lshlrev_b64", 1, d_out);
  run<12>("v_add3_u32", 1, d_out);
  run<14>("v_bfi_b32", 1, d_out);
  run<19>("v_min_u32", 1, d_out);
  run<20>("v_cmp_u64", 1, d_out);                       // compare into an SGPR pair, nothing depends on it
  run<21>("v_cmp_u32", 1, d_out);
  run<22>("v_cndmask_b32 (sgpr mask)", 1, d_out);       // select on a mask that no instruction of the loop writes
  run<23>("v_bitop3_b32", 1, d_out);
  run<24>("v_not_b32", 1, d_out);
  run<25>("v_mov_b64", 1, d_out);
  return 0;
}
