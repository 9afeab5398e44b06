// RandomWalkLoop over the knots of a Gaussian-kernel basis under a regression likelihood, every column of every
// chain in one launch (plus one for the proposals).
//
// What it replaces: the loop metropolis_hastings.py:276-289 (one truncated random-walk proposal per knot,
// metropolis_hastings.py:212-269, accept/reject :127-173) for the model of the reference's reversible-jump tests
// (tests/test_reversible_jump.py:24-40: B[i, k] = phi((X_i - theta_k) / s) / s) with the response
//   y ~ N(B beta + offsets, (tau W)^-1),  W diagonal,  theta_k ~ U(lower, upper),
// issued launch by launch from the host it is ~13 launches per knot (proposal, basis column, residual form, densities,
// accept, merges).  The structure that makes one launch possible:
//   * knot k's proposal depends on theta_k only, which no other knot's step changes: all proposals of a chain (the
//     truncated-normal inverse CDF and its two densities are long serial evaluations) are made up front, one per lane;
//   * moving knot k changes the fitted values by beta_k (phi_new - phi_old): the residual r = y - fitted lives in
//     registers (a few rows per thread) and a step is one pass over it (new column evaluated on the fly, old column read
//     from B), the quadratic form of the proposed state accumulated directly as sum w (r - beta_k d)^2 -- no expanded
//     difference, so the conditioning is that of the two separate evaluations the host route makes;
//   * an accepted move updates r, theta_k and column k of B in place.
// Draw streams, truncated-normal arithmetic and the accept test are the host route's (k_rw_propose, k_mh_accept): both
// routes see the same proposals and uniforms; the log-density DIFFERENCE differs by rounding only (summation order).
// One workgroup per chain; chains with fewer live knots simply finish earlier.
#include "omc_common.h"
#include "omc_truncnorm.h"

omc_status omc_ensure_bytes(omc_ctx* ctx, void** buf, size_t* have, size_t need);  // omc_dense.hip

#define KN_THREADS 1024

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

// one basis value, k_gaussian_basis's arithmetic (a unit scale divides exactly: the divisions are skipped)
__device__ __forceinline__ double kn_basis(double x, double knot, double scale0, bool unit) {
  if (unit) {
    const double t = x - knot;
    return exp(-(t * t) / 2.0) / 2.5066282746310002;
  }
  const double t = (x - knot) / scale0;
  return exp(-(t * t) / 2.0) / 2.5066282746310002 / scale0;
}

// All proposals of all chains, one per lane (k_rw_propose with p = 1: Philox block 2 k for the proposal of column k,
// block 2 k + 1 for its accept uniform): prop[c][k] = {z, log q(z | theta), log q(theta | z), log u}.  Its own launch:
// the truncated-normal evaluations are long and register-hungry, the loop kernel below is neither.
__global__ void __launch_bounds__(64) k_knot_propose(int64_t C, int64_t chain_offset, int64_t kmax, const double* theta,
                                                     const double* count, double step, double lower, double upper,
                                                     const double* inj_z, const double* inj_u, omc_rng_key key, double* prop) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= C * kmax) return;
  const int64_t c = e / kmax, j = e - c * kmax;
  if (!((double)j < count[c])) return;
  const double mu = theta[e];
  double d, u;
  if (inj_z) {
    d = inj_z[j * C + c];
  } else {
    const uint4 wv = omc_rng_block(key, chain_offset + c, (uint32_t)(2 * j));
    d = omc_u53(wv.x, wv.y);
  }
  if (inj_u) {
    u = inj_u[j * C + c];
  } else {
    const uint4 wv = omc_rng_block(key, chain_offset + c, (uint32_t)(2 * j + 1));
    u = omc_u53(wv.x, wv.y);
  }
  const double z = omc_truncated_normal_rv(mu, step, lower, upper, d);
  double* out = prop + 4 * e;
  out[0] = z;
  out[1] = omc_truncated_normal_log_pdf(z, mu, step, lower, upper);
  out[2] = omc_truncated_normal_log_pdf(mu, z, step, lower, upper);
  out[3] = log(u);
}

// NR = rows per thread (NR * KN_THREADS >= n; row i = tid + q * KN_THREADS).  The chain's residual, its rows of X and
// the proposed column of the current step stay in registers; the chain's proposals and coefficients sit in LDS; the
// current column of the NEXT step is fetched before the reduction of this one (a step is otherwise one global-load
// latency + one reduction long, and a chain has up to kmax of them in sequence).  NR > 5 would not fit that in 128
// registers: the lean form keeps the residual and the proposed column only and re-reads the rest every step.
template <int NR>
__global__ void __launch_bounds__(KN_THREADS) k_knot_loop(int64_t C, int64_t n, int64_t kmax, const double* X, double scale0,
                                                         const double* y, const double* add_shared, const double* add_chain,
                                                         const double* w, const double* tau, const double* beta, double* theta,
                                                         const double* count, double* B, const double* prop,
                                                         long long* n_accept, long long* n_proposal, int* accept_out,
                                                         double* log_alpha_out) {
  __shared__ double red[KN_THREADS / 64];
  extern __shared__ double sm[];  // [5 * kmax]: proposals {z, lq_fwd, lq_rev, log u} and the coefficients
  const int64_t c = blockIdx.x;
  const int tid = threadIdx.x;
  int k_live = (int)count[c];
  if (k_live > kmax) k_live = (int)kmax;
  double* Bc = B + c * kmax * n;
  double* thc = theta + c * kmax;
  double* sp = sm;
  double* sb = sm + 4 * kmax;
  int* live = (int*)(sm + 5 * kmax);   // the columns with a non-zero coefficient, in order; live[kmax]: how many
  for (int e = tid; e < 4 * k_live; e += KN_THREADS) sp[e] = prop[4 * c * kmax + e];
  for (int e = tid; e < (int)kmax; e += KN_THREADS) sb[e] = beta[c * kmax + e];
  __syncthreads();
  if (tid == 0) {
    int nl = 0;
    for (int j = 0; j < (int)kmax; ++j)
      if (sb[j] != 0.0) live[nl++] = j;
    live[kmax] = nl;
  }
  __syncthreads();
  const bool unit = scale0 == 1.0;
  // Every load below is unconditional: a thread whose row lies beyond n reads row n - 1 and drops the value.  (A load under
  // `if (i < n)` is a branch with the wait for its data inside: the NR rows of a column, then the columns, then y, the offsets,
  // the weights and X came in one memory round trip after the other -- some forty of them before the first step.)
  int ic[NR];
  bool in[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int i = tid + q * KN_THREADS;
    in[q] = i < (int)n;
    ic[q] = in[q] ? i : (int)n - 1;
  }

  // residual and quadratic form of the current state -- k_design_resid_sq's arithmetic: columns in order, zero
  // coefficients skipped (CH columns in flight at a time; a chunk's last columns may repeat the last live one with a zero
  // coefficient, which leaves the sums as they are)
  constexpr bool LEAN = NR > 5;
  constexpr int NK = LEAN ? 1 : NR;
  constexpr int CH = LEAN ? 1 : 4;
  double r[NR], x[NK], wt[NK];
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < NR; ++q) r[q] = 0.0;
  const int n_live_cols = live[kmax];
  for (int t = 0; t < n_live_cols; t += CH) {
    double v[CH][NR], cf[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const bool have = t + u < n_live_cols;
      const int j = live[have ? t + u : n_live_cols - 1];
      cf[u] = have ? sb[j] : 0.0;
      const double* colj = Bc + (int64_t)j * n;
#pragma unroll
      for (int q = 0; q < NR; ++q) v[u][q] = colj[ic[q]];
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
#pragma unroll
      for (int q = 0; q < NR; ++q) r[q] = fma(v[u][q], cf[u], r[q]);
    }
  }
#pragma unroll
  for (int g = 0; g < NR; g += 5) {  // five rows' loads in flight at a time (registers)
    double yv[5], av[5], sv[5], wv[5], xv[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int q = g + u;
      yv[u] = y[ic[q]];
      av[u] = add_chain ? add_chain[c * n + ic[q]] : 0.0;
      sv[u] = add_shared ? add_shared[ic[q]] : 0.0;
      wv[u] = w ? w[ic[q]] : 1.0;
      xv[u] = LEAN ? 0.0 : X[ic[q]];
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int q = g + u;
      if (!LEAN) { x[q] = in[q] ? xv[u] : 0.0; wt[q] = in[q] ? wv[u] : 0.0; }
      if (in[q]) {
        const double f = r[q] + av[u] + sv[u];
        r[q] = yv[u] - f;
        acc = fma(wv[u] * r[q], r[q], acc);
      } else {
        r[q] = 0.0;
      }
    }
  }
  double cn[NK];  // the current column of the step about to run
  if (!LEAN) {
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const double v0 = Bc[ic[q]];
      cn[q] = (k_live > 0 && in[q]) ? v0 : 0.0;
    }
  }
  double quad = kn_block_sum(acc, red, tid);
  const double tc = tau ? tau[c] : 1.0;
  int n_acc = 0;

  for (int j = 0; j < k_live; ++j) {
    const double z = sp[4 * j], bj = sb[j];
    double* col = Bc + (int64_t)j * n;
    double pn[NR], rn[NK];
    acc = 0.0;
#pragma unroll
    for (int g = 0; g < NR; g += 5) {
      // (the lean form re-reads X, the column and the weights every step: unconditional loads, five rows in flight at a time)
      double xq[5], cq[5], wq[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int q = g + u;
        xq[u] = LEAN ? X[ic[q]] : x[q];
        cq[u] = LEAN ? col[ic[q]] : cn[q];
        wq[u] = LEAN ? (w ? w[ic[q]] : 1.0) : wt[q];
      }
      if (LEAN) asm volatile("" ::: "memory");  // (keeps the next five rows' loads behind this group's arithmetic)
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int q = g + u;
        pn[q] = 0.0;
        if (!LEAN) rn[q] = 0.0;
        if (in[q]) {
          pn[q] = kn_basis(xq[u], z, scale0, unit);
          const double rq = fma(-bj, pn[q] - cq[u], r[q]);
          if (!LEAN) rn[q] = rq;
          acc = fma(wq[u] * rq, rq, acc);
        }
      }
    }
    if (!LEAN && j + 1 < k_live) {  // next step's column: in flight under the reduction
#pragma unroll
      for (int q = 0; q < NR; ++q) cn[q] = col[n + ic[q]];
    }
    const double quad_n = kn_block_sum(acc, red, tid);
    const double la = (-0.5 * tc * quad_n) + sp[4 * j + 2] - ((-0.5 * tc * quad) + sp[4 * j + 1]);
    const bool ok = sp[4 * j + 3] < la;  // a NaN log_alpha rejects
    if (tid == 0) {
      if (accept_out) accept_out[(int64_t)j * C + c] = ok;
      if (log_alpha_out) log_alpha_out[(int64_t)j * C + c] = la;
    }
    if (ok) {
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        if (in[q]) {
          r[q] = LEAN ? fma(-bj, pn[q] - col[ic[q]], r[q]) : rn[q];  // (the value the sum above was taken over)
          col[ic[q]] = pn[q];
        }
      }
      if (tid == 0) thc[j] = z;
      quad = quad_n;
      ++n_acc;
    }
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
  const int64_t rows = (n + KN_THREADS - 1) / KN_THREADS;
  const size_t lds = (size_t)(5 * kmax) * sizeof(double) + (size_t)(kmax + 2) * sizeof(int);  // + the list of live columns
  if (rows > 10 || lds > 32 * 1024) return OMC_INVALID_ARG;  // the chain's residual lives in registers: n <= 10 240
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const int64_t Cn = ctx->n_chains;
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->rj_tmp, &ctx->rj_tmp_bytes, (size_t)(4 * Cn * kmax) * sizeof(double));
  if (st != OMC_OK) return st;
  double* prop = (double*)ctx->rj_tmp;
  hipLaunchKernelGGL(k_knot_propose, dim3((unsigned)((Cn * kmax + 63) / 64)), dim3(64), 0, ctx->stream, Cn, ctx->chain_offset,
                     kmax, theta, count, step, lower, upper, inject_z, inject_u,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), prop);
  OMC_HIP_CHECK(hipGetLastError());
#define KN_LAUNCH(NR)                                                                                                    \
  hipLaunchKernelGGL(k_knot_loop<NR>, dim3((unsigned)Cn), dim3(KN_THREADS), lds, ctx->stream, Cn, n, kmax, X, scale, y,   \
                     add_shared, add_chain, w, tau, beta, theta, count, B, prop, (long long*)accept_count,               \
                     (long long*)proposal_count, (int*)accept_out, log_alpha_out)
  if (rows <= 5) KN_LAUNCH(5);
  else KN_LAUNCH(10);
#undef KN_LAUNCH
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}
