// accuracy of v_rcp_f64 / v_rsq_f64 and of their Newton refinements (relative error in ulp of the result)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  double r1 = fma(r, e, r);
  e = fma(-d, r1, 1.0);
  double r2 = fma(r1, e, r1);
  { double e0 = fma(-d, r, 1.0); r2 = fma(r, fma(e0, e0, e0), r); }  // the three-operation form the library uses
  double g = __builtin_amdgcn_rsq(d);
  double s = d * g, h = 0.5 * g;
  double ee = fma(-s, s, d);
  double s1 = fma(ee, h, s);
  ee = fma(-s1, s1, d);
  double s2 = fma(ee, h, s1);
  out[6 * i + 0] = r; out[6 * i + 1] = r1; out[6 * i + 2] = r2; out[6 * i + 3] = s; out[6 * i + 4] = s1; out[6 * i + 5] = s2;
}
int main() {
  const int n = 1 << 20;
  double* hx = new double[n]; double* ho = new double[6 * n];
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = ldexp(0.5 + 0.5 * (rand() / (double)RAND_MAX), (rand() % 60) - 30);
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(ho, dout, 6 * n * 8, hipMemcpyDeviceToHost);
  double worst[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    long double rr = 1.0L / hx[i], ss = sqrtl((long double)hx[i]);
    for (int q = 0; q < 6; ++q) {
      long double ref = q < 3 ? rr : ss;
      double err = (double)(fabsl((long double)ho[6 * i + q] - ref) / fabsl(ref)) / 1.1102230246251565e-16;
      if (err > worst[q]) worst[q] = err;
    }
  }
  printf("max error in units of 2^-53: rcp %.3g, +1 NR %.3g, +2 NR %.3g | sqrt via rsq %.3g, +1 %.3g, +2 %.3g\n", worst[0], worst[1], worst[2], worst[3], worst[4], worst[5]);
  return 0;
}
