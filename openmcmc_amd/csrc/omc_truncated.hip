// Truncated Gaussian full conditional: gmrf.gibbs_canonical_truncated_normal (gmrf.py:201-266), the branch
// NormalNormal.sample takes when the parameter's prior has domain limits (sampler.py:199-205).
//
// One scan of single-site Gibbs updates in index order, each from its univariate truncated normal
//     x_i ~ N_[lo_i, hi_i]( (b_i - sum_j Q_ij x_j + Q_ii x_i) / Q_ii , 1 / Q_ii )
// with x_j already updated for j < i (gmrf.py:254-264).  The scan is sequential in i by definition (a
// reordered or coloured scan is a different Markov kernel, not the reference's), so the parallelism is
// over chains: tridiagonal precisions take one lane per chain, dense ones one wave per chain with the
// row product spread over the lanes.  Also Normal.check_domain_response (location_scale.py:169-188).
#include <math.h>

#include "omc_common.h"
#include "omc_truncnorm.h"

static inline unsigned grid1(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

struct TruncTerms {
  int n_terms;
  const double* diag[OMC_MAX_TERMS];
  const double* off[OMC_MAX_TERMS];
  const double* rhs[OMC_MAX_TERMS];
  const double* scale[OMC_MAX_TERMS];
};

__device__ __forceinline__ double trunc_uniform(const double* u_in, int64_t ld, int64_t c, int64_t i, const omc_rng_key& key,
                                                int64_t gc) {
  if (u_in) return u_in[c * ld + i];
  const uint4 w = omc_rng_block(key, gc, (uint32_t)(i >> 1));
  return (i & 1) ? omc_u53(w.z, w.w) : omc_u53(w.x, w.y);
}

__global__ void k_tridiag_gibbs_truncated(int64_t C, int64_t chain_offset, int64_t n, TruncTerms T, const double* rhs_chain,
                                          int64_t ld_rhs, const double* lower, const double* upper, const double* u_in,
                                          int64_t ld_u, omc_rng_key key, double* x, int64_t ld_x, long long* bad) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) s[k] = (k < T.n_terms && T.scale[k]) ? T.scale[k][c] : 1.0;
  double* xc = x + c * ld_x;
  double x_prev = 0.0, off_prev = 0.0;  // x_{i-1} (already updated) and Q_{i,i-1}
  double x_cur = xc[0];
  bool fail = false;
  // row i of Q_c and b_c from the shared vectors; the NEXT site's row is fetched while this site's draw is worked out
  // (the scan is one dependent chain per lane: a load waited for at every site is most of a site's time)
  auto site = [&](int64_t i, double& a, double& o, double& b, double& lo, double& hi, double& xn) {
    a = 0.0; o = 0.0; b = rhs_chain ? rhs_chain[c * ld_rhs + i] : 0.0;
#pragma unroll
    for (int k = 0; k < OMC_MAX_TERMS; ++k) {
      if (k < T.n_terms) {
        a = fma(s[k], T.diag[k] ? T.diag[k][i] : 1.0, a);
        if (T.off[k] && i + 1 < n) o = fma(s[k], T.off[k][i], o);
        if (T.rhs[k]) b = fma(s[k], T.rhs[k][i], b);
      }
    }
    lo = lower ? lower[i] : -INFINITY;
    hi = upper ? upper[i] : INFINITY;
    xn = (i + 1 < n) ? xc[i + 1] : 0.0;
  };
  double a_n, o_n, b_n, lo_n, hi_n, xn_n, u_odd = 0.5;
  site(0, a_n, o_n, b_n, lo_n, hi_n, xn_n);
  for (int64_t i = 0; i < n; ++i) {
    const double a = a_n, o = o_n, b = b_n, lo = lo_n, hi = hi_n, x_next = xn_n;
    if (i + 1 < n) site(i + 1, a_n, o_n, b_n, lo_n, hi_n, xn_n);
    if (!(a > 0.0)) fail = true;
    double mean, sd;
    // 1/a and sqrt(1/a) by the refined hardware reciprocal / reciprocal square root (the divide and sqrt sequences are
    // ~70 dependent instructions on a path that is one dependent chain per lane)
    const double v = (a > 0.0) ? omc_rcp_nr(a) : 1.0;
    {
      const double g = __builtin_amdgcn_rsq((a > 0.0) ? a : 1.0);  // ~ 1/sqrt(a): two Newton steps
      const double h = 0.5 * a;
      double r = g;
      r = fma(r, fma(-h * r, r, 0.5), r);
      r = fma(r, fma(-h * r, r, 0.5), r);
      sd = r;
    }
    if (n == 1) {  // gmrf.py:244-247
      mean = b * v;
    } else {       // gmrf.py:255-262: v_i * (b_i - Q[i,:] @ x + Q_ii x_i), row product in column order
      const double row = fma(o, x_next, fma(a, x_cur, off_prev * x_prev));
      mean = v * ((b - row) + a * x_cur);
    }
    // in-kernel uniforms: one Philox block serves the two sites of a pair
    double uu;
    if (u_in) {
      uu = u_in[c * ld_u + i];
    } else if ((i & 1) == 0) {
      const uint4 w4 = omc_rng_block(key, chain_offset + c, (uint32_t)(i >> 1));
      uu = omc_u53(w4.x, w4.y);
      u_odd = omc_u53(w4.z, w4.w);
    } else {
      uu = u_odd;
    }
    const double xi = omc_truncated_normal_rv_inv(mean, sd, a * sd, lo, hi, uu);  // 1/sd = sqrt(a) = a / sqrt(a)
    xc[i] = xi;
    x_prev = xi;
    off_prev = o;
    x_cur = x_next;
  }
  if (fail) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

// banded precision of bandwidth w (Q_c = sum_k s_k[c] M_k, M_k in the band storage of omc_band_terms: band[d*n + i] =
// M[i+d, i]): the same scan, one lane per chain.  Row i of Q touches x_{i-w..i+w}; the sub-diagonal part of the row comes
// from the columns i-d (band[d*n + i-d]), the super-diagonal part from column i itself (band[d*n + i]).  The row product
// is accumulated in column order like the reference's Q[i, :] @ x (gmrf.py:258).
struct TruncBand {
  int n_terms;
  const double* band[OMC_MAX_TERMS];
  int bw[OMC_MAX_TERMS];
  const double* rhs[OMC_MAX_TERMS];
  const double* scale[OMC_MAX_TERMS];
};

__global__ void k_band_gibbs_truncated(int64_t C, int64_t chain_offset, int64_t n, int w, TruncBand T, const double* rhs_chain,
                                       int64_t ld_rhs, const double* lower, const double* upper, const double* u_in, int64_t ld_u,
                                       omc_rng_key key, double* x, int64_t ld_x, long long* bad) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) s[k] = (k < T.n_terms && T.scale[k]) ? T.scale[k][c] : 1.0;
  double* xc = x + c * ld_x;
  bool fail = false;
  for (int64_t i = 0; i < n; ++i) {
    double a = 0.0, b = rhs_chain ? rhs_chain[c * ld_rhs + i] : 0.0, row = 0.0;
#pragma unroll
    for (int k = 0; k < OMC_MAX_TERMS; ++k)
      if (k < T.n_terms) {
        a = fma(s[k], T.band[k] ? T.band[k][i] : 1.0, a);
        if (T.rhs[k]) b = fma(s[k], T.rhs[k][i], b);
      }
    // columns i-w .. i-1 (already updated), then i, then i+1 .. i+w
    for (int d = w; d >= 1; --d) {
      if (i - d < 0) continue;
      double q = 0.0;
#pragma unroll
      for (int k = 0; k < OMC_MAX_TERMS; ++k)
        if (k < T.n_terms && T.band[k] && d <= T.bw[k]) q = fma(s[k], T.band[k][(int64_t)d * n + (i - d)], q);
      row = fma(q, xc[i - d], row);
    }
    const double xi_old = xc[i];
    row = fma(a, xi_old, row);
    for (int d = 1; d <= w; ++d) {
      if (i + d >= n) break;
      double q = 0.0;
#pragma unroll
      for (int k = 0; k < OMC_MAX_TERMS; ++k)
        if (k < T.n_terms && T.band[k] && d <= T.bw[k]) q = fma(s[k], T.band[k][(int64_t)d * n + i], q);
      row = fma(q, xc[i + d], row);
    }
    if (!(a > 0.0)) fail = true;
    const double lo = lower ? lower[i] : -INFINITY, hi = upper ? upper[i] : INFINITY;
    double mean, sd;
    if (n == 1) {
      mean = b / a;
      sd = 1.0 / sqrt(a);
    } else {
      const double v = 1.0 / a;
      sd = sqrt(v);
      mean = v * ((b - row) + a * xi_old);
    }
    xc[i] = omc_truncated_normal_rv(mean, sd, lo, hi, trunc_uniform(u_in, ld_u, c, i, key, chain_offset + c));
  }
  if (fail) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

// dense precision, one wave per chain: Q_c = sum_k s_k[c] M_k assembled row by row on the fly
__global__ void __launch_bounds__(64) k_dense_gibbs_truncated(int64_t C, int64_t chain_offset, int64_t p, int n_terms,
                                                              const double* m0, const double* m1, const double* m2,
                                                              const double* m3, const double* s0, const double* s1,
                                                              const double* s2, const double* s3, const double* r0,
                                                              const double* r1, const double* r2, const double* r3,
                                                              const double* rhs_chain, int64_t ld_rhs, const double* lower,
                                                              const double* upper, const double* u_in, int64_t ld_u,
                                                              omc_rng_key key, double* x, int64_t ld_x, long long* bad) {
  extern __shared__ double xs[];  // the chain's vector
  const int64_t c = blockIdx.x;
  const int lane = threadIdx.x;
  const double* M[4] = {m0, m1, m2, m3};
  const double* R[4] = {r0, r1, r2, r3};
  double s[4] = {s0 ? s0[c] : 1.0, s1 ? s1[c] : 1.0, s2 ? s2[c] : 1.0, s3 ? s3[c] : 1.0};
  for (int64_t j = lane; j < p; j += 64) xs[j] = x[c * ld_x + j];
  __syncthreads();
  bool fail = false;
  for (int64_t i = 0; i < p; ++i) {
    double dot = 0.0;
    for (int64_t j = lane; j < p; j += 64) {
      double q = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < n_terms) q = fma(s[k], M[k] ? M[k][i * p + j] : (i == j ? 1.0 : 0.0), q);
      dot = fma(q, xs[j], dot);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) dot += __shfl_xor(dot, sh, 64);
    if (lane == 0) {
      double a = 0.0, b = rhs_chain ? rhs_chain[c * ld_rhs + i] : 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < n_terms) {
          a = fma(s[k], M[k] ? M[k][i * p + i] : 1.0, a);
          if (R[k]) b = fma(s[k], R[k][i], b);
        }
      if (!(a > 0.0)) fail = true;
      const double lo = lower ? lower[i] : -INFINITY, hi = upper ? upper[i] : INFINITY;
      double mean, sd;
      if (p == 1) {
        mean = b / a;
        sd = 1.0 / sqrt(a);
      } else {
        const double v = 1.0 / a;
        sd = sqrt(v);
        mean = v * ((b - dot) + a * xs[i]);
      }
      xs[i] = omc_truncated_normal_rv(mean, sd, lo, hi, trunc_uniform(u_in, ld_u, c, i, key, chain_offset + c));
    }
    __syncthreads();
  }
  for (int64_t j = lane; j < p; j += 64) x[c * ld_x + j] = xs[j];
  if (fail && lane == 0) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

// Normal.check_domain_response + log_p = -inf outside the domain (location_scale.py:162-188)
__global__ void k_domain_penalty(int64_t C, int64_t n, const double* x, int64_t ld, const double* lower, const double* upper,
                                 double* out) {
  const int64_t c = blockIdx.x;
  __shared__ int outside;
  if (threadIdx.x == 0) outside = 0;
  __syncthreads();
  int mine = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = x[c * ld + i];
    if ((lower && v < lower[i]) || (upper && v > upper[i])) mine = 1;
  }
  if (mine) outside = 1;
  __syncthreads();
  if (threadIdx.x == 0 && outside) out[c] = -INFINITY;
}

extern "C" {

omc_status omc_tridiag_gibbs_truncated(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms, const double* rhs_chain,
                                       int64_t ld_rhs, const double* lower, const double* upper, const double* u_inject,
                                       int64_t ld_u, uint64_t draw_index, double* x, int64_t ld_x) {
  if (!ctx || n < 1 || !terms || terms->n_terms < 1 || terms->n_terms > OMC_MAX_TERMS || !x || ld_x < n ||
      (rhs_chain && ld_rhs < n) || (u_inject && ld_u < n))
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  TruncTerms T;
  T.n_terms = terms->n_terms;
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    const bool on = k < terms->n_terms;
    T.diag[k] = on ? terms->diag[k] : nullptr;
    T.off[k] = on ? terms->off[k] : nullptr;
    T.rhs[k] = on ? terms->rhs[k] : nullptr;
    T.scale[k] = on ? terms->scale[k] : nullptr;
  }
  hipLaunchKernelGGL(k_tridiag_gibbs_truncated, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, n, T, rhs_chain, ld_rhs, lower, upper, u_inject, ld_u,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), x, ld_x, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_band_gibbs_truncated(omc_ctx* ctx, int64_t n, int64_t w, const omc_band_terms* terms, const double* rhs_chain,
                                    int64_t ld_rhs, const double* lower, const double* upper, const double* u_inject, int64_t ld_u,
                                    uint64_t draw_index, double* x, int64_t ld_x) {
  if (!ctx || n < 1 || w < 0 || w > 128 || !terms || terms->n_terms < 1 || terms->n_terms > OMC_MAX_TERMS || !x || ld_x < n ||
      (rhs_chain && ld_rhs < n) || (u_inject && ld_u < n))
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  TruncBand T;
  T.n_terms = terms->n_terms;
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    const bool on = k < terms->n_terms;
    T.band[k] = on ? terms->band[k] : nullptr;
    T.bw[k] = on ? terms->bw[k] : 0;
    T.rhs[k] = on ? terms->rhs[k] : nullptr;
    T.scale[k] = on ? terms->scale[k] : nullptr;
    if (on && (T.bw[k] < 0 || T.bw[k] > w)) return OMC_INVALID_ARG;
  }
  hipLaunchKernelGGL(k_band_gibbs_truncated, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, n, (int)w, T, rhs_chain, ld_rhs, lower, upper, u_inject, ld_u,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), x, ld_x, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_dense_gibbs_truncated(omc_ctx* ctx, int64_t p, const omc_dense_terms* terms, const double* rhs_chain,
                                     int64_t ld_rhs, const double* lower, const double* upper, const double* u_inject,
                                     int64_t ld_u, uint64_t draw_index, double* x, int64_t ld_x) {
  if (!ctx || p < 1 || p > 8192 || !terms || terms->n_terms < 1 || terms->n_terms > OMC_MAX_TERMS || !x || ld_x < p ||
      (rhs_chain && ld_rhs < p) || (u_inject && ld_u < p))
    return OMC_INVALID_ARG;
  if (terms->diag_chain) return OMC_UNSUPPORTED;  // per-chain diagonal (mixture prior) under a truncated conditional
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const double *m[4] = {0, 0, 0, 0}, *s[4] = {0, 0, 0, 0}, *r[4] = {0, 0, 0, 0};
  for (int k = 0; k < terms->n_terms; ++k) { m[k] = terms->mat[k]; s[k] = terms->scale[k]; r[k] = terms->rhs[k]; }
  hipLaunchKernelGGL(k_dense_gibbs_truncated, dim3((unsigned)ctx->n_chains), dim3(64), (size_t)p * sizeof(double),
                     ctx->stream, ctx->n_chains, ctx->chain_offset, p, (int)terms->n_terms, m[0], m[1], m[2], m[3], s[0], s[1],
                     s[2], s[3], r[0], r[1], r[2], r[3], rhs_chain, ld_rhs, lower, upper, u_inject, ld_u,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), x, ld_x, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_domain_penalty(omc_ctx* ctx, int64_t n, const double* x, int64_t ld, const double* lower,
                              const double* upper, double* out) {
  if (!ctx || n < 1 || !x || ld < n || !out) return OMC_INVALID_ARG;
  if (!lower && !upper) return OMC_OK;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_domain_penalty, dim3((unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, ctx->n_chains, n, x, ld,
                     lower, upper, out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"
