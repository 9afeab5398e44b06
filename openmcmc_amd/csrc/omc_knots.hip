// RandomWalkLoop over the knots of a Gaussian-kernel basis under a regression likelihood, every column of every
// chain in ONE launch.
//
// What it replaces: the loop metropolis_hastings.py:276-289 (one truncated random-walk proposal per knot,
// metropolis_hastings.py:212-269, accept/reject :127-173) for the model of the reference's reversible-jump tests
// (tests/test_reversible_jump.py:24-40: B[i, k] = phi((X_i - theta_k) / s) / s) with the response
//   y ~ N(B beta + offsets, (tau W)^-1),  W diagonal,  theta_k ~ U(lower, upper),
// issued launch by launch from the host it is ~13 launches per knot (proposal, basis column, residual form, densities,
// accept, merges).  The structure that makes one launch possible:
//   * knot k's proposal depends on theta_k only, which no other knot's step changes: all proposals of a chain (the
//     truncated-normal inverse CDF and its two densities are long serial evaluations) are made up front, one per lane;
//   * moving knot k changes the fitted values by beta_k (phi_new - phi_old): the residual r = y - fitted lives in LDS
//     and a step is one pass over it (new column evaluated on the fly, old column read from B), the quadratic form of
//     the proposed state accumulated directly as sum w (r - beta_k d)^2 -- no expanded difference, so the conditioning
//     is that of the two separate evaluations the host route makes;
//   * an accepted move updates r, theta_k and column k of B in place.
// Draw streams, truncated-normal arithmetic and the accept test are the host route's (k_rw_propose, k_mh_accept): both
// routes see the same proposals and uniforms; the log-density DIFFERENCE differs by rounding only (summation order).
// One workgroup per chain; chains with fewer live knots simply finish earlier.
#include "omc_common.h"
#include "omc_truncnorm.h"

#define KN_THREADS 512

__device__ __forceinline__ double kn_block_sum(double v, double* red, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();  // red is free again
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < KN_THREADS / 64; ++k) s += red[k];
  return s;
}

__global__ void __launch_bounds__(KN_THREADS) k_knot_loop(int64_t C, int64_t chain_offset, int64_t n, int64_t kmax, const double* X,
                                                         double scale0, const double* y, const double* add_shared,
                                                         const double* add_chain, const double* w, const double* tau,
                                                         const double* beta, double* theta, const double* count, double* B,
                                                         double step, double lower, double upper, const double* inj_z,
                                                         const double* inj_u, omc_rng_key key, long long* n_accept,
                                                         long long* n_proposal, int* accept_out, double* log_alpha_out) {
  extern __shared__ double sm[];
  double* r = sm;               // [n] residual of the chain's current state
  double* red = r + n;          // [KN_THREADS / 64]
  double* pz = red + KN_THREADS / 64;  // [kmax] proposed knots
  double* plf = pz + kmax;      // [kmax] log q(z | theta)
  double* plr = plf + kmax;     // [kmax] log q(theta | z)
  double* plu = plr + kmax;     // [kmax] log of the accept uniform
  const int64_t c = blockIdx.x;
  const int tid = threadIdx.x;
  int k_live = (int)count[c];
  if (k_live > kmax) k_live = (int)kmax;
  double* Bc = B + c * kmax * n;
  const double* bc = beta + c * kmax;
  double* thc = theta + c * kmax;

  // all proposals of the chain, one per lane (k_rw_propose with p = 1: Philox block 2 k for the proposal of column k,
  // block 2 k + 1 for its accept uniform)
  for (int j = tid; j < k_live; j += KN_THREADS) {
    const double mu = thc[j];
    double d, u;
    if (inj_z) {
      d = inj_z[(int64_t)j * C + c];
    } else {
      const uint4 wv = omc_rng_block(key, chain_offset + c, (uint32_t)(2 * j));
      d = omc_u53(wv.x, wv.y);
    }
    if (inj_u) {
      u = inj_u[(int64_t)j * C + c];
    } else {
      const uint4 wv = omc_rng_block(key, chain_offset + c, (uint32_t)(2 * j + 1));
      u = omc_u53(wv.x, wv.y);
    }
    const double z = omc_truncated_normal_rv(mu, step, lower, upper, d);
    pz[j] = z;
    plf[j] = omc_truncated_normal_log_pdf(z, mu, step, lower, upper);
    plr[j] = omc_truncated_normal_log_pdf(mu, z, step, lower, upper);
    plu[j] = log(u);
  }

  // residual and quadratic form of the current state (k_design_resid_sq's arithmetic: columns in order, zero
  // coefficients skipped)
  double acc = 0.0;
  for (int64_t i = tid; i < n; i += KN_THREADS) {
    double s = 0.0;
    for (int64_t j = 0; j < kmax; ++j) {
      const double cf = bc[j];
      if (cf != 0.0) s = fma(Bc[j * n + i], cf, s);
    }
    const double f = s + (add_chain ? add_chain[c * n + i] : 0.0) + (add_shared ? add_shared[i] : 0.0);
    const double ri = y[i] - f;
    r[i] = ri;
    acc = fma((w ? w[i] : 1.0) * ri, ri, acc);
  }
  double quad = kn_block_sum(acc, red, tid);  // (its barriers also publish r and the proposals)
  const double tc = tau ? tau[c] : 1.0;
  int n_acc = 0;

  for (int j = 0; j < k_live; ++j) {
    const double z = pz[j], bj = bc[j];
    double* col = Bc + (int64_t)j * n;
    acc = 0.0;
    for (int64_t i = tid; i < n; i += KN_THREADS) {
      const double t = (X[i] - z) / scale0;
      const double pn = exp(-(t * t) / 2.0) / 2.5066282746310002 / scale0;  // k_gaussian_basis
      const double rn = fma(-bj, pn - col[i], r[i]);
      acc = fma((w ? w[i] : 1.0) * rn, rn, acc);
    }
    const double quad_n = kn_block_sum(acc, red, tid);
    const double la = (-0.5 * tc * quad_n) + plr[j] - ((-0.5 * tc * quad) + plf[j]);
    const bool ok = plu[j] < la;  // a NaN log_alpha rejects
    if (tid == 0) {
      if (accept_out) accept_out[(int64_t)j * C + c] = ok;
      if (log_alpha_out) log_alpha_out[(int64_t)j * C + c] = la;
    }
    if (ok) {
      for (int64_t i = tid; i < n; i += KN_THREADS) {
        const double t = (X[i] - z) / scale0;
        const double pn = exp(-(t * t) / 2.0) / 2.5066282746310002 / scale0;
        r[i] = fma(-bj, pn - col[i], r[i]);
        col[i] = pn;
      }
      if (tid == 0) thc[j] = z;
      quad = quad_n;
      ++n_acc;
    }
    // (each thread re-reads only the r and col entries it wrote itself: no barrier needed before the next column)
  }
  if (tid == 0) {
    if (n_proposal) n_proposal[c] += k_live;
    if (n_accept) n_accept[c] += n_acc;
  }
}

extern "C" omc_status omc_knot_loop(omc_ctx* ctx, int64_t n, int64_t kmax, const double* X, double scale, const double* y,
                                    const double* add_shared, const double* add_chain, const double* w, const double* tau,
                                    const double* beta, double* theta, const double* count, double* B, double step,
                                    double lower, double upper, const double* inject_z, const double* inject_u,
                                    uint64_t draw_index, int64_t* accept_count, int64_t* proposal_count, int32_t* accept_out,
                                    double* log_alpha_out) {
  if (!ctx || n < 1 || kmax < 1 || !X || !(scale > 0.0) || !y || !beta || !theta || !count || !B || !(step > 0.0) ||
      !(lower < upper))
    return OMC_INVALID_ARG;
  const size_t lds = (size_t)(n + KN_THREADS / 64 + 4 * kmax) * sizeof(double);
  if (lds > 160 * 1024) return OMC_INVALID_ARG;  // the residual must fit the LDS of a CU (n <= ~20 000)
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  if (lds > 48 * 1024)
    OMC_HIP_CHECK(hipFuncSetAttribute((const void*)k_knot_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_knot_loop, dim3((unsigned)ctx->n_chains), dim3(KN_THREADS), lds, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, n, kmax, X, scale, y, add_shared, add_chain, w, tau, beta, theta, count, B, step, lower,
                     upper, inject_z, inject_u, omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), (long long*)accept_count,
                     (long long*)proposal_count, (int*)accept_out, log_alpha_out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}
