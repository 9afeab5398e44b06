// Issue cost of a few VALU instructions on gfx950, in cycles per wave-instruction, with 1 and 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 8
template <int OP> __global__ void __launch_bounds__(1024) k_rate(uint64_t* out, int iters, uint32_t seed) {
  uint32_t a[REP];
  uint64_t w[REP];
  double d[REP];
#pragma unroll
  for (int i = 0; i < REP; ++i) { a[i] = seed + threadIdx.x * 7 + i; w[i] = a[i]; d[i] = 1.0 + a[i] * 1e-9; }
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < REP; ++i) {
      if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "s"(0xD2511F53u) : "vcc");
      if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "s"(0xD2511F53u));
      if (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(0xD2511F53u));
      if (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % REP]));
      if (OP == 4) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
      if (OP == 5) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i]));
      if (OP == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(0x511F53u));
      if (OP == 7) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
      if (OP == 8) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));
      if (OP == 9) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < REP; ++i) acc += a[i] + w[i] + (uint64_t)d[i];
  if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
  out[blockIdx.x * 2 + 1] = acc;
}

template <int OP> static void run(const char* name) {
  uint64_t* d;
  hipMalloc(&d, 4096 * 16);
  const int iters = 4096;
  for (int threads : {256, 1024}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<OP>, dim3(256), dim3(threads), 0, 0, d, iters, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(256), dim3(threads), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: threads/256 waves, each iters*REP instructions
    const double inst_per_simd = (double)(threads / 256) * iters * REP;
    printf("%-16s %d waves/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD (x2.4 GHz = %.2f cycles)\n", name, threads / 256, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
  }
  hipFree(d);
}

int main() {
  run<3>("v_xor_b32");
  run<0>("v_mad_u64_u32");
  run<1>("v_mul_hi_u32");
  run<2>("v_mul_lo_u32");
  run<6>("v_mul_u32_u24");
  run<4>("v_fma_f64");
  run<5>("v_add_f64");
  run<8>("v_mul_f64");
  run<7>("v_rcp_f64");
  run<9>("v_mov_b32_dpp");
  return 0;
}
