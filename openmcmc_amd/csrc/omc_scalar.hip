// Per-chain scalar work: Normal-Gamma conjugate update, log-density pieces, raw random fills.
#include <math.h>

#include "omc_common.h"

__global__ void k_normal_gamma(int64_t C, int64_t chain_offset, double a0, double b0, double half_npos,
                               const double* quad, const double* g_inject, omc_rng_key key, double* out,
                               long long* bad) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double a = a0 + half_npos;
  const double b = b0 + 0.5 * quad[c];
  const double scale = (b == 0.0) ? INFINITY : omc_rcp_nr(b);  // sampler.py:285-286
  bool failed = false;
  const double g = g_inject ? g_inject[c] : omc_standard_gamma(key, chain_offset + c, a, &failed);
  out[c] = g * scale;
  if (failed) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

__global__ void k_scaled_gauss_logpdf(int64_t C, double n, const double* scale, const double* logdet_unscaled,
                                      const double* quad, double* out, int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double s = scale ? scale[c] : 1.0;
  const double lp = 0.5 * (n * log(s) + logdet_unscaled[0] - n * 1.8378770664093453 - s * quad[c]);
  out[c] = accumulate ? out[c] + lp : lp;
}

__global__ void k_gamma_logpdf(int64_t C, const double* x, double shape, double rate, double lnorm, double* out,
                               int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double v = x[c];
  double lp = (v > 0.0) ? lnorm + (shape - 1.0) * log(v) - rate * v : -INFINITY;
  if (v == 0.0 && shape == 1.0) lp = lnorm;
  out[c] = accumulate ? out[c] + lp : lp;
}

__global__ void k_fill_normal(int64_t C, int64_t chain_offset, int64_t n, omc_rng_key key, double* out, int64_t ld) {
  const int64_t c = blockIdx.y;
  const int64_t gc = chain_offset + c;
  const int64_t npairs = (n + 1) / 2;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npairs; p += (int64_t)gridDim.x * blockDim.x) {
    double z0, z1;
    omc_normal_pair(omc_rng_block(key, gc, (uint32_t)p), z0, z1);
    out[c * ld + 2 * p] = z0;
    if (2 * p + 1 < n) out[c * ld + 2 * p + 1] = z1;
  }
}

__global__ void k_fill_u32(int64_t C, int64_t chain_offset, int64_t n_words, omc_rng_key key, uint32_t* out,
                           int64_t ld) {
  const int64_t c = blockIdx.y;
  const int64_t gc = chain_offset + c;
  const int64_t nblk = (n_words + 3) / 4;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (int64_t)gridDim.x * blockDim.x) {
    const uint4 w = omc_rng_block(key, gc, (uint32_t)b);
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    for (int t = 0; t < 4; ++t)
      if (4 * b + t < n_words) out[c * ld + 4 * b + t] = ws[t];
  }
}

static inline unsigned grid1(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

extern "C" {

omc_status omc_normal_gamma_update(omc_ctx* ctx, double a0, double b0, int64_t n_pos, const double* quad,
                                   const double* g_inject, uint64_t draw_index, double* out) {
  if (!ctx || !quad || !out || n_pos < 0) return OMC_INVALID_ARG;
  if (!(a0 + 0.5 * (double)n_pos > 0.0)) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_normal_gamma, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, a0, b0, 0.5 * (double)n_pos, quad, g_inject,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_GAMMA), out, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_scaled_gauss_logpdf(omc_ctx* ctx, int64_t n, const double* scale, const double* logdet_unscaled,
                                   const double* quad, double* out, int32_t accumulate) {
  if (!ctx || n < 1 || !logdet_unscaled || !quad || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_scaled_gauss_logpdf, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream,
                     ctx->n_chains, (double)n, scale, logdet_unscaled, quad, out, (int)accumulate);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_gamma_logpdf(omc_ctx* ctx, const double* x, double shape, double rate, double* out,
                            int32_t accumulate) {
  if (!ctx || !x || !out || !(shape > 0.0) || !(rate > 0.0)) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const double lnorm = shape * log(rate) - lgamma(shape);
  hipLaunchKernelGGL(k_gamma_logpdf, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, x,
                     shape, rate, lnorm, out, (int)accumulate);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_fill_normal(omc_ctx* ctx, int64_t n, uint64_t draw_index, double* out, int64_t ld) {
  if (!ctx || n < 1 || !out || ld < n) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  unsigned gx = grid1((n + 1) / 2, 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(k_fill_normal, dim3(gx, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, n, omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), out, ld);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_fill_philox_u32(omc_ctx* ctx, int64_t n_words, uint64_t draw_index, uint32_t* out, int64_t ld) {
  if (!ctx || n_words < 1 || !out || ld < n_words) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  unsigned gx = grid1((n_words + 3) / 4, 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(k_fill_u32, dim3(gx, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, n_words, omc_make_key(ctx->seed, draw_index, OMC_RNG_RAW), out, ld);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"
