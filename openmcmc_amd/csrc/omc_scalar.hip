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

// omc_log_post_sum: the pieces one after the other, each exactly as its own kernel computes it, summed in order
struct LogpArgs {
  int n;
  omc_logp_piece p[OMC_LOGP_MAX];
  double lnorm[OMC_LOGP_MAX];
};
__global__ void k_log_post_sum(int64_t C, LogpArgs P, double host_const, double* out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < OMC_LOGP_MAX; ++i) {
    if (i >= P.n) break;
    const omc_logp_piece& q = P.p[i];
    double lp;
    if (q.kind == 0) {
      const double s = q.scale ? q.scale[c] : 1.0;
      const double ld = (q.logdet_mult == 1.0) ? q.logdet[0] : q.logdet[0] * q.logdet_mult;
      lp = 0.5 * (q.n * log(s) + ld - q.n * 1.8378770664093453 - s * q.quad[c]);
    } else {
      const double v = q.x[c];
      lp = (v > 0.0) ? P.lnorm[i] + (q.shape - 1.0) * log(v) - q.rate * v : -INFINITY;
      if (v == 0.0 && q.shape == 1.0) lp = P.lnorm[i];
    }
    acc = (i == 0) ? lp : acc + lp;
  }
  out[c] = (host_const != 0.0) ? acc + host_const : acc;
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

// reversible_jump.py:310-373 and :173
// Two output forms: the move probabilities themselves (p_birth_out / p_death_out), or what ReversibleJump.proposal
// goes on to make of them (count_f given: the count is read as the float64 the state holds; outputs the proposed count
// and the two proposal log-densities of reversible_jump.py:142-144 / 189-191,
//   birth: lq_fwd = log p_birth + density, lq_rev = log p_death;  death: lq_fwd = log p_death, lq_rev = log p_birth + density
// with density = dens_const + dens_chain[c]: the log prior density of the last element of every associated parameter).
__global__ void k_rj_move(int64_t C, int64_t chain_offset, long long n_max, double q, const long long* n,
                          const double* u_in, const long long* idx_in, omc_rng_key key, int* birth_out,
                          double* p_birth_out, double* p_death_out, long long* del_out, long long* bad,
                          const double* count_f, const double* dens_chain, double dens_const, double* count_prop_out,
                          double* lq_fwd_out, double* lq_rev_out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long long nc = count_f ? (long long)count_f[c] : n[c];
  if (nc < 1 || nc > n_max || (count_f && (double)nc != count_f[c])) {
    atomicMin((unsigned long long*)bad, (unsigned long long)c);
    birth_out[c] = 0; del_out[c] = -1;
    if (p_birth_out) { p_birth_out[c] = 0.0; p_death_out[c] = 0.0; }
    if (count_prop_out) { count_prop_out[c] = count_f[c]; lq_fwd_out[c] = NAN; lq_rev_out[c] = NAN; }
    return;
  }
  bool birth;
  if (nc == n_max) {
    birth = false;
  } else if (nc == 1) {
    birth = true;
  } else {
    double u;
    if (u_in) {
      u = u_in[c];
    } else {
      const uint4 w = omc_rng_block(key, chain_offset + c, 0u);
      u = omc_u53(w.x, w.y);
    }
    birth = u <= q;
  }
  double pb = q, pd = 1.0 - q;
  if (nc == n_max) pd = 1.0;
  if (nc == n_max - 1 && birth) pd = 1.0;
  if (nc == 1) pb = 1.0;
  if (nc == 2 && !birth) pb = 1.0;
  long long idx = -1;
  if (!birth) {
    if (idx_in) {
      idx = idx_in[c];
    } else {  // unbiased integer in [0, nc): Lemire's multiply-shift with rejection on 32-bit words
      const uint32_t range = (uint32_t)nc;
      const uint32_t thresh = (uint32_t)(-range) % range;
      uint32_t blk = 1u;
      for (;;) {
        const uint4 w = omc_rng_block(key, chain_offset + c, blk++);
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        bool done = false;
        for (int t = 0; t < 4 && !done; ++t) {
          const unsigned long long m = (unsigned long long)ws[t] * range;
          if ((uint32_t)m >= thresh) { idx = (long long)(m >> 32); done = true; }
        }
        if (done || blk > 64u) break;
      }
      if (idx < 0) idx = 0;
    }
  }
  birth_out[c] = birth ? 1 : 0;
  del_out[c] = idx;
  if (p_birth_out) {
    p_birth_out[c] = pb;
    p_death_out[c] = pd;
  }
  if (count_prop_out) {
    const double dens = dens_const + (dens_chain ? dens_chain[c] : 0.0);
    const double lb = log(pb) + dens, ld = log(pd);
    count_prop_out[c] = count_f[c] + (birth ? 1.0 : -1.0);
    lq_fwd_out[c] = birth ? lb : ld;
    lq_rev_out[c] = birth ? ld : lb;
  }
}

// mean / variance over stored iterations (and chains when pooled): one lane per output element,
// coalesced across the element index, Welford accumulation
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

omc_status omc_log_post_sum(omc_ctx* ctx, int32_t n_pieces, const omc_logp_piece* pieces, double host_const, double* out) {
  if (!ctx || n_pieces < 1 || n_pieces > OMC_LOGP_MAX || !pieces || !out) return OMC_INVALID_ARG;
  LogpArgs P{};
  P.n = n_pieces;
  for (int i = 0; i < n_pieces; ++i) {
    const omc_logp_piece& q = pieces[i];
    P.p[i] = q;
    if (q.kind == 0) {
      if (!q.logdet || !q.quad) return OMC_INVALID_ARG;
    } else if (q.kind == 1) {
      if (!q.x || !(q.shape > 0.0) || !(q.rate > 0.0)) return OMC_INVALID_ARG;
      P.lnorm[i] = q.shape * log(q.rate) - lgamma(q.shape);
    } else {
      return OMC_INVALID_ARG;
    }
  }
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_log_post_sum, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, P, host_const, out);
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

omc_status omc_rj_move(omc_ctx* ctx, int64_t n_max, double birth_probability, const int64_t* n, const double* u_inject,
                       const int64_t* idx_inject, uint64_t draw_index, int32_t* birth_out, double* p_birth_out,
                       double* p_death_out, int64_t* del_index_out) {
  if (!ctx || n_max < 1 || n_max > 0x7fffffffLL || !(birth_probability >= 0.0 && birth_probability <= 1.0) || !n ||
      !birth_out || !p_birth_out || !p_death_out || !del_index_out)
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_rj_move, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, ctx->chain_offset,
                     (long long)n_max, birth_probability, (const long long*)n, u_inject, (const long long*)idx_inject,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), birth_out, p_birth_out, p_death_out,
                     (long long*)del_index_out, ctx->d_bad_chain, nullptr, nullptr, 0.0, nullptr, nullptr, nullptr);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_rj_move_densities(omc_ctx* ctx, int64_t n_max, double birth_probability, const double* count,
                                 const double* u_inject, const int64_t* idx_inject, uint64_t draw_index,
                                 const double* density_chain, double density_const, int32_t* birth_out,
                                 int64_t* del_index_out, double* count_prop_out, double* lq_fwd_out, double* lq_rev_out) {
  if (!ctx || n_max < 1 || n_max > 0x7fffffffLL || !(birth_probability >= 0.0 && birth_probability <= 1.0) || !count ||
      !birth_out || !del_index_out || !count_prop_out || !lq_fwd_out || !lq_rev_out)
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_rj_move, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, ctx->chain_offset,
                     (long long)n_max, birth_probability, (const long long*)nullptr, u_inject, (const long long*)idx_inject,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), birth_out, (double*)nullptr, (double*)nullptr,
                     (long long*)del_index_out, ctx->d_bad_chain, count, density_chain, density_const, count_prop_out,
                     lq_fwd_out, lq_rev_out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

// dst[c][j] = src[c][j] for j < count[c], NaN beyond: one draw of a variable-size parameter into its store slab
// (sampler.py:112-116; the reference leaves its NaN fill beyond the live length)
__global__ void k_store_ragged(int64_t C, int64_t width, const double* src, int64_t src_stride, const double* count,
                               double* dst, int64_t dst_stride) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= C * width) return;
  const int64_t c = e / width, j = e - c * width;
  dst[c * dst_stride + j] = ((double)j < count[c]) ? src[c * src_stride + j] : NAN;
}

omc_status omc_store_ragged(omc_ctx* ctx, int64_t width, const double* src, int64_t src_chain_stride, const double* count,
                            double* dst, int64_t dst_chain_stride) {
  if (!ctx || width < 1 || !src || !count || !dst || src_chain_stride < width || dst_chain_stride < width)
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_store_ragged, dim3(grid1(ctx->n_chains * width, 256)), dim3(256), 0, ctx->stream, ctx->n_chains, width,
                     src, src_chain_stride, count, dst, dst_chain_stride);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_store_moments(omc_ctx* ctx, int64_t n_iter, int64_t size, const double* store, int32_t pooled,
                             double* mean_out, double* var_out) {
  if (!ctx || n_iter < 1 || size < 1 || !store || (!mean_out && !var_out)) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  // both summaries are column moments of one row-major matrix (omc_store.hip): pooled [n_iter C][size], per chain [n_iter][C size]
  const int64_t C = ctx->n_chains;
  return omc_col_moments(ctx, store, pooled ? n_iter * C : n_iter, pooled ? size : C * size, mean_out, var_out);
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

// ---- log-density pieces for ragged parameters ---------------------
// sum over a wave in a fixed order (butterfly): every lane ends with the total
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_xor(v, sh, 64);
  return v;
}

// one WAVE per chain (blockDim 256 = 4 chains): the lanes stride over the chain's elements, coalesced; with one thread per
// chain a p = 500 coefficient vector cost 180 us of serial, strided loads and logs
__global__ void __launch_bounds__(256) k_diag_gauss_logpdf(int64_t C, int64_t kmax, const double* x, const double* mean, const double* prec,
                                                           const double* count, double* out, int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;
  const int64_t k = count ? (int64_t)count[c] : kmax;
  double ld = 0.0, q = 0.0;
  for (int64_t j = lane; j < k; j += 64) {
    const double d = prec[c * kmax + j];
    const double r = x[c * kmax + j] - (mean ? mean[c * kmax + j] : 0.0);
    ld += log(d);
    q = fma(d * r, r, q);
  }
  ld = wave_sum_d(ld);
  q = wave_sum_d(q);
  const double lp = 0.5 * (ld - (double)k * 1.8378770664093453 - q);
  if (lane == 0) out[c] = accumulate ? out[c] + lp : lp;
}

__global__ void k_poisson_logpmf(int64_t C, const double* x, double rate, double* out, int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double k = x[c];
  double lp = -INFINITY;
  if (k >= 0.0 && k == floor(k)) lp = (k == 0.0 ? 0.0 : k * log(rate)) - lgamma(k + 1.0) - rate;
  out[c] = accumulate ? out[c] + lp : lp;
}

__global__ void k_count_logpdf(int64_t C, const double* count, double per_element, double* out, int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double lp = per_element * count[c];
  out[c] = accumulate ? out[c] + lp : lp;
}

// (param2, fill2, out2): a second table gathered by the same allocation in the same launch (the mean and the precision
// of a mixture Normal always go together)
__global__ void k_mixture_gather(int64_t C, int64_t kmax, int64_t m, const double* param, int64_t pstride,
                                 const double* alloc, const double* count, double fill, double* out, long long* bad,
                                 const double* param2, int64_t pstride2, double fill2, double* out2) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= C * kmax) return;
  const int64_t c = t / kmax, j = t % kmax;
  double v = fill, v2 = fill2;
  if (!count || (double)j < count[c]) {
    const int64_t a = (int64_t)alloc[t];
    if (a < 0 || a >= m) {
      atomicMin((unsigned long long*)bad, (unsigned long long)c);
    } else {
      v = param[c * pstride + a];
      if (param2) v2 = param2[c * pstride2 + a];
    }
  }
  out[t] = v;
  if (out2) out2[t] = v2;
}

// Gamma.log_p of a ragged (1, k) response (distribution.py:241-261): sum over the live entries, or (last_only) the
// density of the LAST live entry -- what ReversibleJump takes from log_p(..., by_observation=True)[-1]
// (reversible_jump.py:132,143)
__global__ void k_gamma_logpdf_ragged(int64_t C, int64_t kmax, const double* x, const double* count, double shape,
                                      double rate, double lnorm, int last_only, double* out, int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int64_t k = count ? (int64_t)count[c] : kmax;
  double lp = 0.0;
  for (int64_t j = last_only ? (k > 0 ? k - 1 : 0) : 0; j < k; ++j) {
    const double v = x[c * kmax + j];
    double t = (v > 0.0) ? lnorm + (shape - 1.0) * log(v) - rate * v : -INFINITY;
    if (v == 0.0 && shape == 1.0) t = lnorm;
    lp += t;
  }
  out[c] = accumulate ? out[c] + lp : lp;
}

// gradient of a diagonal Gaussian log-density w.r.t. its response: g_j = -prec_j (x_j - mean_j) on the live entries
// (location_scale.py:222-226 with the MixtureParameterMatrix precision of parameter.py:501)
__global__ void k_diag_gauss_grad(int64_t C, int64_t kmax, const double* x, const double* mean, const double* prec,
                                  const double* count, double* grad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= C * kmax) return;
  const int64_t c = t / kmax, j = t % kmax;
  const bool live = !count || (double)j < count[c];
  grad[t] = live ? -prec[t] * (x[t] - (mean ? mean[t] : 0.0)) : 0.0;
}

// out[c] = sum_i (a[c][i] - ca[i]) (b[c][i] - cb[i]): the quadratic form (x - m)' M (x - m) of a DENSE shared M from
// b = M x (one GEMM over all chains) and cb = M m, without forming x - m
__global__ void __launch_bounds__(256) k_centered_rowdot(int64_t n, const double* a, int64_t ld_a, const double* ca,
                                                        const double* b, int64_t ld_b, const double* cb, double* out) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256)
    acc = fma(a[c * ld_a + i] - (ca ? ca[i] : 0.0), b[c * ld_b + i] - (cb ? cb[i] : 0.0), acc);
  for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// MixtureAllocation.sample (sampler.py:340-353): one thread per (chain, element)
__global__ void k_mixture_allocation(int64_t C, int64_t chain_offset, int64_t p, int64_t K, const double* y,
                                     const double* prior, int64_t prior_rows, const double* mean, int64_t mstride,
                                     const double* prec, int64_t pstride, const double* u_in, omc_rng_key key, double* alloc) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= C * p) return;
  const int64_t c = t / p, i = t % p;
  const double yi = y[t];
  const double* pr = prior + (prior_rows > 1 ? i * K : 0);
  double total = 0.0;
  for (int64_t k = 0; k < K; ++k) {
    const double sd = 1.0 / sqrt(prec[c * pstride + k]);  // norm.pdf(y, loc, scale = 1/sqrt(prec))
    const double z = (yi - mean[c * mstride + k]) / sd;
    total += pr[k] * (exp(-(z * z) / 2.0) / 2.5066282746310002 / sd);
  }
  double u;
  if (u_in) {
    u = u_in[t];
  } else {
    const uint4 w = omc_rng_block(key, chain_offset + c, (uint32_t)(i >> 1));
    u = (i & 1) ? omc_u53(w.z, w.w) : omc_u53(w.x, w.y);
  }
  double cum = 0.0;
  int64_t pick = 0;
  for (int64_t k = 0; k < K; ++k) {  // np.sum(U > np.cumsum(prob / total))
    const double sd = 1.0 / sqrt(prec[c * pstride + k]);
    const double z = (yi - mean[c * mstride + k]) / sd;
    cum += pr[k] * (exp(-(z * z) / 2.0) / 2.5066282746310002 / sd) / total;
    pick += (u > cum) ? 1 : 0;
  }
  alloc[t] = (double)pick;
}

__global__ void __launch_bounds__(256) k_categorical_logpmf(int64_t C, int64_t p, int64_t K, const double* alloc, const double* prob,
                                                            int64_t prob_rows, double* out, int accumulate, long long* bad) {
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per chain
  const int lane = threadIdx.x & 63;
  if (c >= C) return;
  double lp = 0.0;
  for (int64_t i = lane; i < p; i += 64) {
    const int64_t a = (int64_t)alloc[c * p + i];
    if (a < 0 || a >= K) {
      atomicMin((unsigned long long*)bad, (unsigned long long)c);
      continue;
    }
    lp += log(prob[(prob_rows > 1 ? i * K : 0) + a]);
  }
  lp = wave_sum_d(lp);
  if (lane == 0) out[c] = accumulate ? out[c] + lp : lp;
}

// NormalGamma.sample for a mixture precision: one wave per (chain, component); the lanes stride over the elements
__global__ void __launch_bounds__(256) k_mixture_normal_gamma(int64_t C, int64_t chain_offset, int64_t p, int64_t K, const double* resid,
                                                              const double* alloc, const double* a0, const double* b0, const double* g_in,
                                                              omc_rng_key key, double* out, long long* bad) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= C * K) return;
  const int64_t c = t / K, k = t % K;
  double cnt = 0.0, ss = 0.0;
  for (int64_t i = lane; i < p; i += 64) {
    if ((int64_t)alloc[c * p + i] == k) {
      const double r = resid[c * p + i];
      cnt += 1.0;
      ss = fma(r, r, ss);
    }
  }
  cnt = wave_sum_d(cnt);
  ss = wave_sum_d(ss);
  if (lane != 0) return;
  const double a = a0[k] + 0.5 * cnt, b = b0[k] + 0.5 * ss;
  const double scale = (b == 0.0) ? INFINITY : 1.0 / b;
  bool failed = false;
  double g;
  if (g_in) {
    g = g_in[t];
  } else {  // component k draws from its own stream: the draw index carries k in bits 32-39 of the 48-bit index
    omc_rng_key kk = key;
    kk.c3_base |= ((uint32_t)k & 0xffu) << 8;
    g = omc_standard_gamma(kk, chain_offset + c, a, &failed);
  }
  out[t] = g * scale;
  if (failed) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

__global__ void k_gamma_logpdf_vec(int64_t C, int64_t K, const double* x, const double* shape, const double* rate,
                                   double* out, int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double lp = 0.0;
  for (int64_t k = 0; k < K; ++k) {
    const double v = x[c * K + k], a = shape[k], b = rate[k];
    lp += (v > 0.0) ? a * log(b) - lgamma(a) + (a - 1.0) * log(v) - b * v : -INFINITY;
  }
  out[c] = accumulate ? out[c] + lp : lp;
}

// LogNormal.log_p (location_scale.py:279-300): out = log x element-wise, sumlog[c] = sum_i log x[c][i]
__global__ void __launch_bounds__(256) k_log_transform(int64_t n, const double* x, int64_t ld_x, double* out, int64_t ld_o,
                                                      double* sumlog) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const double v = log(x[c * ld_x + i]);
    out[c * ld_o + i] = v;
    acc += v;
  }
  for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && sumlog) sumlog[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Uniform.rvs (distribution.py:444-458): lower + range * U, product rounded before the sum as numpy does
__global__ void k_uniform_draw(int64_t C, int64_t chain_offset, int64_t p, const double* lower, const double* range,
                               const double* u_in, omc_rng_key key, uint32_t sub, double* out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= C * p) return;
  const int64_t c = t / p, e = t % p;
  double u;
  if (u_in) {
    u = u_in[t];
  } else {
    const uint4 w = omc_rng_block(key, chain_offset + c, sub + (uint32_t)(e >> 1));
    u = (e & 1) ? omc_u53(w.z, w.w) : omc_u53(w.x, w.y);
  }
  {
#pragma clang fp contract(off)
    const double prod = range[e] * u;
    out[t] = lower[e] + prod;
  }
}

// Poisson.rvs (distribution.py:508-523 -> scipy.stats.poisson.rvs = NumPy's legacy generator, third-party arithmetic:
// random_poisson_mult below rate 10, Hoermann's PTRS transformed rejection with NumPy's own random_loggam from 10 on;
// restated in oracle/prior_draws_ref.py and pinned by tests/golden/prior_draws.npz).  One lane per chain; the uniforms come
// from the injected tape u_in[c * ld_u + 0 ..] (NaN-padded; a draw that runs off its tape is reported through `bad`) or from
// the chain's Philox stream, two per block.
__device__ __forceinline__ double poisson_loggam(double x) {
  const double a[10] = {8.333333333333333e-02, -2.777777777777778e-03, 7.936507936507937e-04, -5.952380952380952e-04,
                        8.417508417508418e-04, -1.917526917526918e-03, 6.410256410256410e-03, -2.955065359477124e-02,
                        1.796443723688307e-01, -1.39243221690590e+00};
  if (x == 1.0 || x == 2.0) return 0.0;
  const int n = (x < 7.0) ? (int)(7.0 - x) : 0;
  double x0 = x + (double)n;
  const double x2 = (1.0 / x0) * (1.0 / x0);
  double gl0 = a[9];
  for (int k = 8; k >= 0; --k) gl0 = gl0 * x2 + a[k];
  double gl = gl0 / x0 + 0.5 * 1.8378770664093453 + (x0 - 0.5) * log(x0) - x0;
  for (int k = 0; k < n; ++k) {
    gl -= log(x0 - 1.0);
    x0 -= 1.0;
  }
  return gl;
}

__global__ void k_poisson_draw(int64_t C, int64_t chain_offset, const double* rate, int64_t rate_stride, const double* u_in,
                               int64_t ld_u, omc_rng_key key, double* out, long long* bad) {
#pragma clang fp contract(off)
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double lam = rate[c * rate_stride];
  uint32_t used = 0;
  bool exhausted = false;
  uint4 w = {0u, 0u, 0u, 0u};
  auto next = [&]() -> double {
    double u;
    if (u_in) {
      u = ((int64_t)used < ld_u) ? u_in[c * ld_u + used] : __builtin_nan("");
      if (!(u == u)) { exhausted = true; u = 0.0; }  // off the tape: 0 ends both loops
    } else {
      if ((used & 1u) == 0u) w = omc_rng_block(key, chain_offset + c, used >> 1);
      u = (used & 1u) ? omc_u53(w.z, w.w) : omc_u53(w.x, w.y);
    }
    ++used;
    return u;
  };
  double k = 0.0;
  if (!(lam >= 0.0)) {
    k = __builtin_nan("");
    exhausted = true;
  } else if (lam == 0.0) {
    k = 0.0;
  } else if (lam < 10.0) {
    const double enlam = exp(-lam);
    double prod = 1.0;
    for (;;) {
      prod *= next();
      if (prod > enlam) k += 1.0;
      else break;
    }
  } else {
    const double slam = sqrt(lam), loglam = log(lam);
    const double b = 0.931 + 2.53 * slam;
    const double a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    const double vr = 0.9277 - 3.6224 / (b - 2);
    for (int guard = 0; guard < 100000; ++guard) {
      const double U = next() - 0.5;
      const double V = next();
      if (exhausted) break;
      const double us = 0.5 - fabs(U);
      k = floor((2 * a / us + b) * U + lam + 0.43);
      if (us >= 0.07 && V <= vr) break;
      if (k < 0.0 || (us < 0.013 && V > us)) continue;
      if ((log(V) + log(invalpha) - log(a / (us * us) + b)) <= (-lam + k * loglam - poisson_loggam(k + 1.0))) break;
    }
  }
  out[c] = k;
  if (exhausted) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

extern "C" {

omc_status omc_poisson_draw(omc_ctx* ctx, const double* rate, int64_t rate_stride, const double* u_inject, int64_t ld_u,
                            uint64_t draw_index, double* out) {
  if (!ctx || !rate || !out || rate_stride < 0 || (u_inject && ld_u < 1)) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_poisson_draw, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, ctx->chain_offset, rate,
                     rate_stride, u_inject, ld_u, omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), out, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_uniform_draw(omc_ctx* ctx, int64_t p, const double* lower, const double* range, const double* u_inject,
                            uint64_t draw_index, uint32_t sub, double* out) {
  if (!ctx || p < 1 || !lower || !range || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_uniform_draw, dim3(grid1(ctx->n_chains * p, 256)), dim3(256), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, p, lower, range, u_inject, omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), sub,
                     out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_mixture_allocation(omc_ctx* ctx, int64_t p, int64_t K, const double* y, const double* prior,
                                  int64_t prior_rows, const double* mean, int64_t mean_stride, const double* prec,
                                  int64_t prec_stride, const double* u_inject, uint64_t draw_index, double* alloc) {
  if (!ctx || p < 1 || K < 1 || !y || !prior || (prior_rows != 1 && prior_rows != p) || !mean || !prec || !alloc)
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_mixture_allocation, dim3(grid1(ctx->n_chains * p, 256)), dim3(256), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, p, K, y, prior, prior_rows, mean, mean_stride, prec, prec_stride, u_inject,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), alloc);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_categorical_logpmf(omc_ctx* ctx, int64_t p, int64_t K, const double* alloc, const double* prob,
                                  int64_t prob_rows, double* out, int32_t accumulate) {
  if (!ctx || p < 1 || K < 1 || !alloc || !prob || (prob_rows != 1 && prob_rows != p) || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_categorical_logpmf, dim3(grid1(ctx->n_chains, 4)), dim3(256), 0, ctx->stream, ctx->n_chains, p, K,
                     alloc, prob, prob_rows, out, (int)accumulate, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_mixture_normal_gamma(omc_ctx* ctx, int64_t p, int64_t K, const double* resid, const double* alloc,
                                    const double* a0, const double* b0, const double* g_inject, uint64_t draw_index,
                                    double* out) {
  if (!ctx || p < 1 || K < 1 || K > 255 || !resid || !alloc || !a0 || !b0 || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_mixture_normal_gamma, dim3(grid1(ctx->n_chains * K, 4)), dim3(256), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, p, K, resid, alloc, a0, b0, g_inject,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_GAMMA), out, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_gamma_logpdf_vec(omc_ctx* ctx, int64_t K, const double* x, const double* shape, const double* rate,
                                double* out, int32_t accumulate) {
  if (!ctx || K < 1 || !x || !shape || !rate || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_gamma_logpdf_vec, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, K, x, shape,
                     rate, out, (int)accumulate);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_log_transform(omc_ctx* ctx, int64_t n, const double* x, int64_t ld_x, double* out, int64_t ld_o,
                             double* sumlog) {
  if (!ctx || n < 1 || !x || !out || ld_x < n || ld_o < n) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_log_transform, dim3((unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, n, x, ld_x, out, ld_o, sumlog);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_centered_rowdot(omc_ctx* ctx, int64_t n, const double* a, int64_t ld_a, const double* center_a,
                               const double* b, int64_t ld_b, const double* center_b, double* out) {
  if (!ctx || n < 1 || !a || !b || ld_a < n || ld_b < n || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_centered_rowdot, dim3((unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, n, a, ld_a, center_a, b, ld_b,
                     center_b, out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_gamma_logpdf_ragged(omc_ctx* ctx, int64_t kmax, const double* x, const double* count, double shape,
                                   double rate, int32_t last_only, double* out, int32_t accumulate) {
  if (!ctx || kmax < 1 || !x || !out || !(shape > 0.0) || !(rate > 0.0)) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const double lnorm = shape * log(rate) - lgamma(shape);
  hipLaunchKernelGGL(k_gamma_logpdf_ragged, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, kmax, x,
                     count, shape, rate, lnorm, (int)last_only, out, (int)accumulate);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_diag_gauss_grad(omc_ctx* ctx, int64_t kmax, const double* x, const double* mean, const double* prec,
                               const double* count, double* grad) {
  if (!ctx || kmax < 1 || !x || !prec || !grad) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_diag_gauss_grad, dim3(grid1(ctx->n_chains * kmax, 256)), dim3(256), 0, ctx->stream, ctx->n_chains,
                     kmax, x, mean, prec, count, grad);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_diag_gauss_logpdf(omc_ctx* ctx, int64_t kmax, const double* x, const double* mean, const double* prec,
                                 const double* count, double* out, int32_t accumulate) {
  if (!ctx || kmax < 1 || !x || !prec || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_diag_gauss_logpdf, dim3(grid1(ctx->n_chains, 4)), dim3(256), 0, ctx->stream, ctx->n_chains, kmax,
                     x, mean, prec, count, out, (int)accumulate);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_poisson_logpmf(omc_ctx* ctx, const double* x, double rate, double* out, int32_t accumulate) {
  if (!ctx || !x || !out || !(rate > 0.0)) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_poisson_logpmf, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, x, rate,
                     out, (int)accumulate);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_count_logpdf(omc_ctx* ctx, const double* count, double per_element, double* out, int32_t accumulate) {
  if (!ctx || !count || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_count_logpdf, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, count,
                     per_element, out, (int)accumulate);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_mixture_gather(omc_ctx* ctx, int64_t kmax, int64_t m, const double* param, int64_t param_stride,
                              const double* alloc, const double* count, double fill, double* out) {
  if (!ctx || kmax < 1 || m < 1 || !param || !alloc || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_mixture_gather, dim3(grid1(ctx->n_chains * kmax, 256)), dim3(256), 0, ctx->stream, ctx->n_chains,
                     kmax, m, param, param_stride, alloc, count, fill, out, ctx->d_bad_chain, nullptr, 0, 0.0, nullptr);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_mixture_gather2(omc_ctx* ctx, int64_t kmax, int64_t m, const double* alloc, const double* count,
                               const double* param_a, int64_t stride_a, double fill_a, double* out_a, const double* param_b,
                               int64_t stride_b, double fill_b, double* out_b) {
  if (!ctx || kmax < 1 || m < 1 || !alloc || !param_a || !out_a || !param_b || !out_b) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_mixture_gather, dim3(grid1(ctx->n_chains * kmax, 256)), dim3(256), 0, ctx->stream, ctx->n_chains,
                     kmax, m, param_a, stride_a, alloc, count, fill_a, out_a, ctx->d_bad_chain, param_b, stride_b, fill_b, out_b);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"
