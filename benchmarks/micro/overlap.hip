// Microbenchmark: do L2-resident vector loads and independent VALU work (Philox + Box-Muller) of the same wave
// overlap when one 1024-thread workgroup owns a CU?  Mirrors the fill phases of k_tridiag_seg.
//   hipcc -O3 --offload-arch=gfx950 -I../../include -I../../openmcmc_amd/csrc overlap.hip -o overlap && ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "omc_common.h"

template <int MODE>  // 0 loads only, 1 rng only, 2 loads then rng then use, 3 one load per Philox round
__global__ void __launch_bounds__(1024) k(const double* v, int n, int reps, int nblk, double* out, omc_rng_key key) {
  __shared__ double big[11000];  // 88 KB: one workgroup per CU
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* base = v + wave * 640;
  double acc = 0.0;
  for (int r = 0; r < reps; ++r) {
    double x[10];
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int t = 0; t < 10; ++t) x[t] = base[lane + 64 * t];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1 || MODE == 2) {
      for (int b = 0; b < nblk; ++b) {
        double z0, z1;
        omc_normal_pair(omc_rng_block(key, blockIdx.x, (uint32_t)(threadIdx.x * 8 + b + r)), z0, z1);
        acc += z0 * z1;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 3) {
      uint32_t c0 = threadIdx.x * 8 + r, c1 = key.c1, c2 = blockIdx.x, c3 = key.c3_base, k0 = key.k0, k1 = key.k1;
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        x[t] = base[lane + 64 * t];
        omc_philox_round(c0, c1, c2, c3, k0, k1);
        __builtin_amdgcn_sched_barrier(0);
      }
      double z0, z1;
      omc_normal_pair(make_uint4(c0, c1, c2, c3), z0, z1);
      acc += z0 * z1;
      for (int b = 1; b < nblk; ++b) {
        omc_normal_pair(omc_rng_block(key, blockIdx.x, (uint32_t)(threadIdx.x * 8 + b + r)), z0, z1);
        acc += z0 * z1;
      }
    }
    if (MODE != 1) {
#pragma unroll
      for (int t = 0; t < 10; ++t) acc += x[t];
    }
    base = v + ((wave + r + 1) & 15) * 640;  // same 80 KB vector, different slice: L1 cannot help
    __syncthreads();
  }
  big[threadIdx.x] = acc;
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = big[threadIdx.x];
}

template <int MODE>
float run(const double* v, double* out, int reps, int nblk) {
  omc_rng_key key = omc_make_key(1, 2, 0);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, v, 10240, reps, nblk, out, key);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, v, 10240, reps, nblk, out, key);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1000.0f / reps;  // us per repetition
}

int main() {
  double *v, *out;
  hipMalloc(&v, 10240 * 8); hipMemset(v, 0, 10240 * 8);
  hipMalloc(&out, 256 * 1024 * 8);
  const int reps = 64;
  for (int nblk = 1; nblk <= 3; ++nblk) {
    printf("nblk %d: loads %.3f us  rng %.3f us  loads-then-rng %.3f us  interleaved %.3f us (per repetition, 256 WGs)\n", nblk,
           run<0>(v, out, reps, nblk), run<1>(v, out, reps, nblk), run<2>(v, out, reps, nblk), run<3>(v, out, reps, nblk));
  }
  return 0;
}
