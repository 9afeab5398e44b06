// Generic Metropolis-Hastings building blocks, ragged (variable-dimension) chain state and the
// reversible-jump transitions (BASELINE configs[4]).
//
// Ragged state: a parameter whose dimension changes under reversible jump (knots theta, their
// coefficients beta, the basis B) is held padded to n_max with zeros; count[c] (the float64 the
// reference keeps in state["n_basis"]) says how many leading entries of chain c are live.  Zero
// padding makes every linear-algebra kernel mask-free (a dead basis column contributes nothing).
#include <math.h>

#include "omc_common.h"
#include "omc_truncnorm.h"

omc_status omc_ensure_bytes(omc_ctx* ctx, void** buf, size_t* have, size_t need);  // omc_dense.hip

static inline unsigned grid1(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

// a*b + c with the product rounded first (no FMA contraction): what numpy's `mu + z * step` computes
__device__ __forceinline__ double mul_add_2r(double a, double b, double c) {
#pragma clang fp contract(off)
  const double prod = a * b;
  return prod + c;
}

// ------------------------------------------------------------------------------------------------
// RandomWalk.proposal (metropolis_hastings.py:212-269) for one column of a (p, n_rep) parameter
__global__ void k_rw_propose(int64_t C, int64_t chain_offset, int64_t p, const double* x, int64_t xcs, int64_t xes,
                             const double* step, int64_t ses, const double* lower, const double* upper,
                             const double* count, int64_t index, const double* inject, omc_rng_key key, uint32_t sub,
                             double* z, int64_t zcs, int64_t zes, double* lq_fwd, double* lq_rev) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const bool active = !count || (double)index < count[c];
  double f = 0.0, r = 0.0;
  for (int64_t e = 0; e < p; ++e) {
    const double mu = x[c * xcs + e * xes];
    double out = mu;
    if (active) {
      const double s = step[e * ses];
      double d;
      if (inject) {
        d = inject[c * p + e];
      } else {
        const uint4 w = omc_rng_block(key, chain_offset + c, sub + (uint32_t)(e >> 1));
        if (lower) {
          d = (e & 1) ? omc_u53(w.z, w.w) : omc_u53(w.x, w.y);
        } else {
          double n0, n1;
          omc_normal_pair(w, n0, n1);
          d = (e & 1) ? n1 : n0;
        }
      }
      if (lower) {  // truncated proposal and its two densities (metropolis_hastings.py:253-257)
        out = omc_truncated_normal_rv(mu, s, lower[e], upper[e], d);
        f += omc_truncated_normal_log_pdf(out, mu, s, lower[e], upper[e]);
        r += omc_truncated_normal_log_pdf(mu, out, s, lower[e], upper[e]);
      } else {      // symmetric: both densities reported as 0 (metropolis_hastings.py:250-251)
        out = mul_add_2r(d, s, mu);  // two roundings, as numpy's mu + z*step
      }
    }
    z[c * zcs + e * zes] = out;
  }
  lq_fwd[c] = f;
  lq_rev[c] = r;
}

// ManifoldMALA with a DIAGONAL Hessian h (metropolis_hastings.py:301-373): precision h/step^2, L_jj = sqrt(h_j)/step,
// m = x + (1/2) step^2 g / h.  propose != 0: x_other = m + step z / sqrt(h) is written and log q(x_other | x) returned;
// propose == 0: x_other is read and log q(x_other | x) returned (the reverse move).  log q = sum log L_jj - |L'(.-m)|^2/2.
__global__ void k_mala_diag(int64_t C, int64_t chain_offset, int64_t kmax, const double* x, const double* grad,
                            const double* hdiag, const double* count, double step, int propose, const double* z_in,
                            omc_rng_key key, uint32_t sub, double* x_other, double* lq, long long* bad) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int64_t k = count ? (int64_t)count[c] : kmax;
  double sum_log = 0.0, ss = 0.0;
  bool fail = false;
  for (int64_t j = 0; j < kmax; ++j) {
    const int64_t t = c * kmax + j;
    if (j >= k) {
      if (propose) x_other[t] = 0.0;
      continue;
    }
    const double h = hdiag[t];
    if (!(h > 0.0)) fail = true;
    const double prec = h / (step * step);
    const double L = sqrt(prec);
    const double m = x[t] + 0.5 * (grad[t] / prec);
    double xo;
    if (propose) {
      double z;
      if (z_in) {
        z = z_in[t];
      } else {
        double n0, n1;
        omc_normal_pair(omc_rng_block(key, chain_offset + c, sub + (uint32_t)(j >> 1)), n0, n1);
        z = (j & 1) ? n1 : n0;
      }
      xo = m + z / L;
      x_other[t] = xo;
    } else {
      xo = x_other[t];
    }
    const double w = L * (xo - m);
    sum_log += log(L);
    ss = fma(w, w, ss);
  }
  lq[c] = sum_log - 0.5 * ss;
  if (fail) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

// MetropolisHastings._accept_reject_proposal / accept_proposal (metropolis_hastings.py:127-173)
__global__ void k_mh_accept(int64_t C, int64_t chain_offset, const double* lp_cur, const double* lp_prop,
                            const double* lq_fwd, const double* lq_rev, const double* count, int64_t index,
                            const double* u_in, omc_rng_key key, uint32_t sub, int* accept, double* log_alpha,
                            long long* n_accept, long long* n_proposal) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count && !((double)index < count[c])) {
    accept[c] = 0;
    if (log_alpha) log_alpha[c] = NAN;
    return;
  }
  const double la = lp_prop[c] + (lq_rev ? lq_rev[c] : 0.0) - (lp_cur[c] + (lq_fwd ? lq_fwd[c] : 0.0));
  double u;
  if (u_in) {
    u = u_in[c];
  } else {
    const uint4 w = omc_rng_block(key, chain_offset + c, sub);
    u = omc_u53(w.x, w.y);
  }
  const int acc = log(u) < la;  // NaN log_alpha rejects, as in the reference
  accept[c] = acc;
  if (log_alpha) log_alpha[c] = la;
  if (n_proposal) n_proposal[c] += 1;
  if (n_accept) n_accept[c] += acc;
}

__global__ void k_chain_select(int64_t C, int64_t width, const int* accept, const double* src, double* dst) {
  const int64_t c = blockIdx.y;
  if (!accept[c]) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < width; i += (int64_t)gridDim.x * blockDim.x)
    dst[c * width + i] = src[c * width + i];
}

// the same for up to OMC_SELECT_MAX state entries in one launch (blockIdx.z = entry)
struct SelectItems {
  int n;
  int64_t width[OMC_SELECT_MAX];
  const double* src[OMC_SELECT_MAX];
  double* dst[OMC_SELECT_MAX];
};
__global__ void k_chain_select_multi(int64_t C, const int* accept, SelectItems it) {
  const int64_t c = blockIdx.y;
  if (!accept[c]) return;
  const int e = blockIdx.z;
  const int64_t width = it.width[e];
  const double* src = it.src[e];
  double* dst = it.dst[e];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < width; i += (int64_t)gridDim.x * blockDim.x)
    dst[c * width + i] = src[c * width + i];
}

// np.concatenate / np.delete on the ragged axis (reversible_jump.py:131, 175)
__global__ void k_ragged_resize(int64_t C, int64_t rows, int64_t kmax, const double* count, const int* birth,
                                const long long* del, const double* new_vals, const double* src, double* dst,
                                int64_t cs, int64_t rs, int64_t js) {
  const int64_t c = blockIdx.y;
  const int64_t k = (int64_t)count[c];
  const bool b = birth[c] != 0;
  const int64_t idx = b ? -1 : del[c];
  const int64_t total = rows * kmax;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    // walk the axis with the smaller stride fastest so accesses coalesce in either layout
    const int64_t r = (rs <= js) ? t % rows : t / kmax;
    const int64_t j = (rs <= js) ? t / rows : t % kmax;
    double v = 0.0;
    if (b) {
      if (j < k) v = src[c * cs + r * rs + j * js];
      else if (j == k && k < kmax) v = new_vals ? new_vals[c * rows + r] : 0.0;
    } else {
      if (j < idx) v = src[c * cs + r * rs + j * js];
      else if (j < k - 1) v = src[c * cs + r * rs + (j + 1) * js];
    }
    dst[c * cs + r * rs + j * js] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// per-chain design matrices B_c (n x kmax, column j contiguous: B[c*kmax*n + j*n + i])
__global__ void k_design_predict_batched(int64_t C, int64_t n, int64_t kmax, const double* B, const double* coef,
                                         const double* add_chain, const double* add_shared, double alpha,
                                         const double* chain_scale, double* out) {
  const int64_t c = blockIdx.y;
  const double* Bc = B + c * kmax * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    // eight columns at a time: the coefficients first (wave-uniform), then the loads of the live columns issued
    // together, then the sums in column order (a test-and-load per column serialises the memory latencies)
    for (int64_t j0 = 0; j0 < kmax; j0 += 8) {
      double cf[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cf[u] = (j0 + u < kmax) ? coef[c * kmax + j0 + u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) bv[u] = (cf[u] != 0.0) ? Bc[(j0 + u) * n + i] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (cf[u] != 0.0) s = fma(bv[u], cf[u], s);
    }
    double v = alpha * s + (add_chain ? add_chain[c * n + i] : 0.0) + (add_shared ? add_shared[i] : 0.0);
    if (chain_scale) v *= chain_scale[c];
    out[c * n + i] = v;
  }
}

// the same with two nodes per thread (16-byte accesses): n even, every vector 16-byte aligned
__global__ void k_design_predict_batched2(int64_t C, int64_t n, int64_t kmax, const double* B, const double* coef,
                                          const double* add_chain, const double* add_shared, double alpha,
                                          const double* chain_scale, double* out) {
  const int64_t c = blockIdx.y, n2 = n / 2;
  const double2* Bc = reinterpret_cast<const double2*>(B + c * kmax * n);
  const double2* ac = add_chain ? reinterpret_cast<const double2*>(add_chain + c * n) : nullptr;
  const double2* as = add_shared ? reinterpret_cast<const double2*>(add_shared) : nullptr;
  double2* oc = reinterpret_cast<double2*>(out + c * n);
  const double cs = chain_scale ? chain_scale[c] : 1.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    double s0 = 0.0, s1 = 0.0;
    for (int64_t j0 = 0; j0 < kmax; j0 += 8) {
      double cf[8];
      double2 bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cf[u] = (j0 + u < kmax) ? coef[c * kmax + j0 + u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) bv[u] = (cf[u] != 0.0) ? Bc[(j0 + u) * n2 + i] : make_double2(0.0, 0.0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (cf[u] != 0.0) { s0 = fma(bv[u].x, cf[u], s0); s1 = fma(bv[u].y, cf[u], s1); }
    }
    const double2 a = ac ? ac[i] : make_double2(0.0, 0.0), b = as ? as[i] : make_double2(0.0, 0.0);
    double v0 = alpha * s0 + a.x + b.x, v1 = alpha * s1 + a.y + b.y;
    if (chain_scale) { v0 *= cs; v1 *= cs; }
    oc[i] = make_double2(v0, v1);
  }
}

// part[c][blockIdx.x] = sum over this workgroup's rows of w (y - B coef - add_chain - add_shared)^2: the quadratic form
// of the regression likelihood straight from the basis (the fitted values are neither written nor read back).  A
// workgroup owns rows blockIdx.x * 256 * RS_ROWS ...; the 256 partial sums are folded in a fixed tree.
#define RS_ROWS 4
__global__ void __launch_bounds__(256) k_design_resid_sq(int64_t n, int64_t kmax, const double* B, const double* coef,
                                                         const double* add_chain, const double* add_shared, const double* y,
                                                         const double* w, double* part) {
  __shared__ double red[256];
  const int64_t c = blockIdx.y;
  const double* Bc = B + c * kmax * n;
  // gridDim.x == 1: this workgroup walks all row blocks itself and writes the chain's sum (no second launch: with
  // many chains there are enough workgroups anyway, and the sweeps that use this are bound by the launch rate)
  const int64_t n_blocks = (n + 256 * RS_ROWS - 1) / (256 * RS_ROWS);
  const int64_t blk_lo = gridDim.x == 1 ? 0 : blockIdx.x, blk_hi = gridDim.x == 1 ? n_blocks : blockIdx.x + 1;
  double acc = 0.0;
  for (int64_t blk = blk_lo; blk < blk_hi; ++blk) {
  const int64_t base = blk * 256 * RS_ROWS + threadIdx.x;
  double s[RS_ROWS];
#pragma unroll
  for (int q = 0; q < RS_ROWS; ++q) s[q] = 0.0;
  for (int64_t j0 = 0; j0 < kmax; j0 += 4) {
    double cf[4], bv[4][RS_ROWS];
#pragma unroll
    for (int u = 0; u < 4; ++u) cf[u] = (j0 + u < kmax) ? coef[c * kmax + j0 + u] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < RS_ROWS; ++q) {
        const int64_t i = base + q * 256;
        bv[u][q] = (cf[u] != 0.0 && i < n) ? Bc[(j0 + u) * n + i] : 0.0;
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (cf[u] != 0.0) {
#pragma unroll
        for (int q = 0; q < RS_ROWS; ++q) s[q] = fma(bv[u][q], cf[u], s[q]);
      }
  }
#pragma unroll
  for (int q = 0; q < RS_ROWS; ++q) {
    const int64_t i = base + q * 256;
    if (i < n) {
      const double f = s[q] + (add_chain ? add_chain[c * n + i] : 0.0) + (add_shared ? add_shared[i] : 0.0);
      const double r = y[i] - f;
      acc = fma((w ? w[i] : 1.0) * r, r, acc);
    }
  }
  }  // row blocks
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[c * gridDim.x + blockIdx.x] = red[0];
}

// Gaussian-kernel basis B[c][j][i] = phi((X_i - knot_cj) / scale_cj) / scale_cj for the live knots, zero beyond;
// column >= 0 rewrites only that column (a random-walk move of one knot)
__global__ void k_gaussian_basis(int64_t C, int64_t n, int64_t kmax, const double* X, const double* knots,
                                 const double* scales, double scale0, const double* count, const double* prev_count,
                                 int64_t column, double* B) {
  // one workgroup = a block of rows of one chain, all its columns in turn (a grid over the columns would be mostly
  // workgroups with nothing to do: a quarter of the k_max columns are live at BASELINE configs[4])
  const int64_t c = blockIdx.y;
  const int64_t k_live = count ? (int64_t)count[c] : kmax;
  // prev_count: the buffer already holds zeros in every column >= prev_count[c] -- dead columns beyond that are left alone
  int64_t j_hi = kmax;
  if (prev_count) {
    const int64_t kp = (int64_t)prev_count[c];
    j_hi = k_live > kp ? k_live : kp;
    if (j_hi > kmax) j_hi = kmax;
  }
  const int64_t j_lo = column >= 0 ? column : 0;
  if (column >= 0) j_hi = column + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double x = X[i];
    for (int64_t j = j_lo; j < j_hi; ++j) {
      double v = 0.0;
      if (j < k_live) {
        const double th = knots[c * kmax + j];
        const double sc = scales ? scales[c * kmax + j] : scale0;
        const double t = (x - th) / sc;
        v = exp(-(t * t) / 2.0) / 2.5066282746310002 / sc;  // scipy norm.pdf: exp(-x^2/2)/sqrt(2 pi), then / scale
      }
      B[(c * kmax + j) * n + i] = v;
    }
  }
}

// Gram matrix B'WB (kmax x kmax) and B'W r (kmax) of one chain per workgroup.  Only the live columns
// (count[c] of them) are staged; the residual rides along as one more tile column.  Work items are
// (pair (a <= b), row slice): with few live columns the 256 threads split the rows of a tile between them
// and the slices are summed in a fixed order at the end, so the result does not depend on scheduling.
#define GRAM_TR 128
// With blockIdx.y > 1 the rows are cut into gridDim.y contiguous parts, one workgroup each (one workgroup per chain
// walks 40 tiles of 128 rows behind two barriers each and hides no memory latency: 258 us at cfg5); a part writes
// its sums to gram[(c * parts + part) * kmax^2 ...] / rhs[(c * parts + part) * kmax ...] and k_gram_reduce adds the
// parts in order.
// (B_alt, count_alt, select): chain c takes its basis from B_alt / count_alt where select[c] != 0 (the matched
// reversible-jump transition needs the Gram matrix of the LARGER of two bases: proposed for a birth, current for a death)
__global__ void __launch_bounds__(256) k_design_gram_batched(int64_t n, int64_t kmax, const double* B, const double* w,
                                                             const double* resid_shared, const double* resid_chain,
                                                             const double* count, double* gram, double* rhs,
                                                             const double* B_alt, const double* count_alt, const int* select) {
  extern __shared__ double tile[];  // (kmax + 1) x (GRAM_TR + 1) columns + residual, then weights[GRAM_TR]
  const int64_t c = blockIdx.x;
  const int64_t parts = gridDim.y, part = blockIdx.y;
  const int64_t rows_per = ((n + parts - 1) / parts + GRAM_TR - 1) / GRAM_TR * GRAM_TR;  // whole tiles
  const int64_t row_lo = part * rows_per, row_hi = (row_lo + rows_per < n) ? row_lo + rows_per : n;
  gram += (c * parts + part) * kmax * kmax - c * kmax * kmax;  // the indexing below adds c * kmax^2
  if (rhs) rhs += (c * parts + part) * kmax - c * kmax;
  if (select && select[c]) { B = B_alt; count = count_alt; }
  const double* Bc = B + c * kmax * n;
  const int k = count ? (int)count[c] : (int)kmax;   // live columns
  const int K1 = k + 1;                              // + residual
  const int ld = GRAM_TR + 1;
  double* wt = tile + ((int64_t)kmax + 1) * ld;
  const int n_pairs = K1 * (K1 + 1) / 2;
  const int S = n_pairs >= 256 ? 1 : 256 / n_pairs;  // row slices
  const int items = n_pairs * S;
  double acc[3] = {0.0, 0.0, 0.0};  // kmax <= 36 -> at most 3 items per thread
  int pa[3], pb[3], ps[3];
  for (int q = 0; q < 3; ++q) {
    const int item = (int)threadIdx.x + q * 256;
    pa[q] = pb[q] = -1; ps[q] = 0;
    if (item < items) {  // unrank (a <= b) from the row-major upper triangle
      int a = 0, rem = item % n_pairs;
      while (rem >= K1 - a) { rem -= K1 - a; ++a; }
      pa[q] = a; pb[q] = a + rem; ps[q] = item / n_pairs;
    }
  }
  for (int t = threadIdx.x; t < kmax * kmax; t += 256) gram[c * kmax * kmax + t] = 0.0;
  if (rhs) for (int t = threadIdx.x; t < kmax; t += 256) rhs[c * kmax + t] = 0.0;
  for (int64_t i0 = row_lo; i0 < row_hi; i0 += GRAM_TR) {
    const int len = (int)((row_hi - i0 < GRAM_TR) ? row_hi - i0 : GRAM_TR);
    for (int t = threadIdx.x; t < K1 * GRAM_TR; t += 256) {
      const int col = t / GRAM_TR, i = t % GRAM_TR;
      double v = 0.0;
      if (i < len)
        v = (col < k) ? Bc[(int64_t)col * n + i0 + i]
                      : (resid_shared ? resid_shared[i0 + i] : 0.0) - (resid_chain ? resid_chain[c * n + i0 + i] : 0.0);
      tile[col * ld + i] = v;
    }
    for (int i = threadIdx.x; i < GRAM_TR; i += 256) wt[i] = (i < len) ? (w ? w[i0 + i] : 1.0) : 0.0;
    __syncthreads();
    for (int q = 0; q < 3; ++q) {
      if (pa[q] < 0) continue;
      const double* ta = tile + pa[q] * ld;
      const double* tb = tile + pb[q] * ld;
      double s = acc[q];
      for (int i = ps[q]; i < GRAM_TR; i += S) s = fma(ta[i] * wt[i], tb[i], s);
      acc[q] = s;
    }
    __syncthreads();
  }
  // sum the row slices (the tile is free now: S x n_pairs partials)
  if (S > 1) {
    if (pa[0] >= 0) tile[ps[0] * n_pairs + ((int)threadIdx.x % n_pairs)] = acc[0];
    __syncthreads();
    if ((int)threadIdx.x < n_pairs) {
      double s = 0.0;
      for (int sl = 0; sl < S; ++sl) s += tile[sl * n_pairs + threadIdx.x];
      acc[0] = s;
    } else {
      pa[0] = -1;
    }
  }
  for (int q = 0; q < 3; ++q) {
    if (pa[q] < 0) continue;
    const int a = pa[q], b = pb[q];
    if (b < k) {
      gram[c * kmax * kmax + a * kmax + b] = acc[q];
      gram[c * kmax * kmax + b * kmax + a] = acc[q];
    } else if (a < k && rhs) {
      rhs[c * kmax + a] = acc[q];
    }
  }
}

// out[c][t] = sum over the parts, in order (deterministic); a second array (the right-hand sides) rides along
__global__ void k_gram_reduce(int64_t C, int64_t len, int parts, const double* in, double* out, int64_t len2,
                              const double* in2, double* out2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * len) {
    i -= C * len;
    if (!out2 || i >= C * len2) return;
    len = len2; in = in2; out = out2;
  }
  const int64_t c = i / len, t = i - c * len;
  double s = 0.0;
  for (int p = 0; p < parts; ++p) s += in[(c * parts + p) * len + t];
  out[i] = s;
}

// ------------------------------------------------------------------------------------------------
// NormalNormal.sample for a small ragged parameter (sampler.py:176-197 -> gmrf.py:167-198):
//   Q = diag(prior_prec) + lik_scale * G,  rhs = prior_prec * prior_mean + lik_scale * g,  on the live k x k block
// one wave per chain, lane = row, matrix in LDS.
#define SMALL_KMAX 64
__global__ void __launch_bounds__(64) k_small_sample_canonical(int64_t C, int64_t chain_offset, int kmax, const double* gram,
                                                               const double* gram_rhs, const double* lik_scale,
                                                               const double* prior_prec, const double* prior_mean,
                                                               const double* count, const double* z_in, omc_rng_key key,
                                                               double* x_out, double* mu_out, long long* bad) {
  extern __shared__ double sm[];  // L: kmax x (kmax+1), then w[kmax]
  const int64_t c = blockIdx.x;
  const int lane = threadIdx.x;
  const int ld = kmax + 1;
  double* L = sm;
  double* vec = sm + (int64_t)kmax * ld;
  const int k = count ? (int)count[c] : kmax;
  const double tau = lik_scale ? lik_scale[c] : 1.0;
  if (lane < k) {
    for (int j = 0; j < k; ++j) {
      double q = tau * gram[c * kmax * kmax + lane * kmax + j];
      if (j == lane) q += prior_prec[c * kmax + lane];
      L[lane * ld + j] = q;
    }
  }
  __syncthreads();
  // right-looking Cholesky, column by column (natural order, as np.linalg.cholesky / SuperLU without permutation)
  bool fail = false;
  for (int j = 0; j < k; ++j) {
    const double d = L[j * ld + j];
    if (!(d > 0.0)) { fail = true; break; }
    const double sd = sqrt(d);
    double lij = 0.0;
    if (lane > j && lane < k) lij = L[lane * ld + j] / sd;
    __syncthreads();
    if (lane == j) L[j * ld + j] = sd;
    if (lane > j && lane < k) {
      L[lane * ld + j] = lij;
      vec[lane] = lij;
    }
    __syncthreads();
    if (lane > j && lane < k)
      for (int t = j + 1; t <= lane; ++t) L[lane * ld + t] -= lij * vec[t];
    __syncthreads();
  }
  if (fail) {
    if (lane == 0) atomicMin((unsigned long long*)bad, (unsigned long long)c);
    if (lane < kmax) x_out[c * kmax + lane] = NAN;
    return;
  }
  // forward solve L w = rhs, backward solves L' mu = w and L' v = z (serial in j, lanes hold the running rhs)
  double rhs = 0.0, zz = 0.0;
  if (lane < k) {
    rhs = tau * gram_rhs[c * kmax + lane] + prior_prec[c * kmax + lane] * (prior_mean ? prior_mean[c * kmax + lane] : 0.0);
    if (z_in) {
      zz = z_in[c * kmax + lane];
    } else {
      double n0, n1;
      omc_normal_pair(omc_rng_block(key, chain_offset + c, (uint32_t)(lane >> 1)), n0, n1);
      zz = (lane & 1) ? n1 : n0;
    }
  }
  for (int j = 0; j < k; ++j) {
    if (lane == j) { rhs /= L[j * ld + j]; vec[0] = rhs; }
    __syncthreads();
    if (lane > j && lane < k) rhs -= L[lane * ld + j] * vec[0];
    __syncthreads();
  }
  double mu = rhs, v = zz;  // w in rhs
  for (int j = k - 1; j >= 0; --j) {
    if (lane == j) { mu /= L[j * ld + j]; v /= L[j * ld + j]; vec[0] = mu; vec[1] = v; }
    __syncthreads();
    if (lane < j) { mu -= L[j * ld + lane] * vec[0]; v -= L[j * ld + lane] * vec[1]; }
    __syncthreads();
  }
  if (lane < kmax) {
    x_out[c * kmax + lane] = (lane < k) ? mu + v : 0.0;
    if (mu_out) mu_out[c * kmax + lane] = (lane < k) ? mu : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------
// LU with partial pivoting of an m x m matrix held row-per-lane in LDS (ld = row stride), with nrhs
// right-hand-side columns appended at A[:, m .. m+nrhs).  Returns det (sign included); the solution
// overwrites the rhs columns.  One wave; every lane calls it (iscratch: one shared int for the pivot row).
__device__ double lu_solve_wave(double* A, int ld, int m, int nrhs, int* iscratch) {
  const int lane = threadIdx.x;
  double det = 1.0;
  for (int j = 0; j < m; ++j) {
    // pivot: largest |A[i][j]|, i >= j; ties to the lowest row like LAPACK's idamax
    if (lane == 0) {
      int piv = j;
      double best = fabs(A[j * ld + j]);
      for (int i = j + 1; i < m; ++i) {
        const double v = fabs(A[i * ld + j]);
        if (v > best) { best = v; piv = i; }
      }
      iscratch[0] = piv;
    }
    __syncthreads();
    const int piv = iscratch[0];
    if (piv != j) {
      for (int t = lane; t < m + nrhs; t += 64) {
        const double a = A[j * ld + t];
        A[j * ld + t] = A[piv * ld + t];
        A[piv * ld + t] = a;
      }
      det = -det;
    }
    __syncthreads();
    const double pjj = A[j * ld + j];
    det *= pjj;
    if (lane > j && lane < m) {
      const double f = A[lane * ld + j] / pjj;
      A[lane * ld + j] = f;
      for (int t = j + 1; t < m + nrhs; ++t) A[lane * ld + t] -= f * A[j * ld + t];
    }
    __syncthreads();
  }
  // back substitution on the rhs columns: lane = rhs column
  for (int r = lane; r < nrhs; r += 64) {
    for (int i = m - 1; i >= 0; --i) {
      double s = A[i * ld + m + r];
      for (int t = i + 1; t < m; ++t) s -= A[i * ld + t] * A[t * ld + m + r];
      A[i * ld + m + r] = s / A[i * ld + i];
    }
  }
  __syncthreads();
  return det;
}

// ReversibleJump.matched_birth_transition / matched_death_transition (reversible_jump.py:195-308)
__global__ void __launch_bounds__(64) k_rj_matched(int64_t C, int64_t chain_offset, int kmax, const double* gram_cur,
                                                   const double* gram_prop, const double* count, const int* birth,
                                                   const long long* del, const double* coef_cur, double scale,
                                                   int has_limits, double lim_lo, double lim_hi, const double* inject,
                                                   omc_rng_key key, uint32_t sub, double* coef_prop, double* lq_fwd,
                                                   double* lq_rev) {
  extern __shared__ double sm[];  // A: kmax x (2 kmax + 1), F: kmax x (kmax + 2)
  __shared__ int isc[2];
  const int64_t c = blockIdx.x;
  const int lane = threadIdx.x;
  const int lda = 2 * kmax + 1, ldf = kmax + 2;
  double* A = sm;
  double* F = sm + (int64_t)kmax * lda;
  const int k = (int)count[c];
  const bool b = birth[c] != 0;
  if (b && k >= kmax) {  // cannot grow: the move-type kernel never proposes this
    if (lane < kmax) coef_prop[c * kmax + lane] = NAN;
    if (lane == 0) { lq_fwd[c] = NAN; lq_rev[c] = NAN; }
    return;
  }
  const double* Gm = (b ? gram_prop : gram_cur) + c * kmax * kmax;
  const int m = b ? k + 1 : k;          // size of the larger basis
  const int nr = m - 1;                 // columns of the smaller one
  const int idx = b ? m - 1 : (int)del[c];  // the column of the larger basis missing from the smaller
  // A = [ X_big' X_big + 1e-10 I | X_big' X_small ]
  if (lane < m) {
    for (int j = 0; j < m; ++j) A[lane * lda + j] = Gm[lane * kmax + j] + (j == lane ? 1e-10 : 0.0);
    for (int j = 0; j < nr; ++j) A[lane * lda + m + j] = Gm[lane * kmax + (j < idx ? j : j + 1)];
  }
  __syncthreads();
  lu_solve_wave(A, lda, m, nr, isc);  // G = A[:, m..m+nr)  (m x nr)
  double la_f = 0.0, la_r = 0.0;
  if (b) {
    // mu* = G beta; beta*[:-1] = mu*[:-1]; last ~ (truncated) Normal(mu*[-1], scale); log|det F| = log det G[:k, :k]
    double mu = 0.0;
    if (lane < m)
      for (int j = 0; j < nr; ++j) mu = fma(A[lane * lda + m + j], coef_cur[c * kmax + j], mu);
    if (lane < nr)
      for (int j = 0; j < nr; ++j) F[lane * ldf + j] = A[lane * lda + m + j];
    __syncthreads();
    const double det = (nr > 0) ? lu_solve_wave(F, ldf, nr, 0, isc) : 1.0;
    double last = 0.0;
    if (lane == m - 1) {
      double d;
      if (inject) {
        d = inject[c];
      } else {
        const uint4 w = omc_rng_block(key, chain_offset + c, sub);
        if (has_limits) {
          d = omc_u53(w.x, w.y);
        } else {
          double n1;
          omc_normal_pair(w, d, n1);
        }
      }
      if (has_limits) {
        last = omc_truncated_normal_rv(mu, scale, lim_lo, lim_hi, d);
        la_f = omc_truncated_normal_log_pdf(last, mu, scale, lim_lo, lim_hi);
      } else {  // gmrf.sample_normal / multivariate_normal_pdf with Q = 1/scale^2 (reversible_jump.py:255-257)
        last = mul_add_2r(d, scale, mu);
        const double t = (last - mu) / scale;
        la_f = 0.5 * (-2.0 * log(scale) - 1.8378770664093453 - t * t);
      }
      lq_fwd[c] += la_f;
      lq_rev[c] += log(det);
    }
    if (lane < kmax) coef_prop[c * kmax + lane] = (lane < m - 1) ? mu : (lane == m - 1 ? last : 0.0);
  } else {
    // F = G with the unit column e_idx inserted at idx; mu_aug = F^{-1} beta; delete entry idx
    if (lane < m) {
      for (int j = 0; j < m; ++j) {
        double v;
        if (j == idx) v = (lane == idx) ? 1.0 : 0.0;
        else v = A[lane * lda + m + (j < idx ? j : j - 1)];
        F[lane * ldf + j] = v;
      }
      F[lane * ldf + m] = coef_cur[c * kmax + lane];
    }
    __syncthreads();
    const double det = lu_solve_wave(F, ldf, m, 1, isc);
    if (lane == 0) {
      const double del_val = F[idx * ldf + m];
      if (has_limits) {
        la_r = omc_truncated_normal_log_pdf(del_val, 0.0, scale, lim_lo, lim_hi);
      } else {
        const double t = del_val / scale;
        la_r = 0.5 * (-2.0 * log(scale) - 1.8378770664093453 - t * t);
      }
      lq_fwd[c] += log(det);
      lq_rev[c] += la_r;
    }
    if (lane < kmax) {
      double v = 0.0;
      if (lane < m - 1) v = F[(lane < idx ? lane : lane + 1) * ldf + m];
      coef_prop[c * kmax + lane] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host entry points
// Per-chain small symmetric positive definite matrices A_c (k x k, k <= 64; a parameter-dependent Hessian / step^2 of
// ManifoldMALA, metropolis_hastings.py:325-373): A_c v_c, v_c' A_c v_c and log det A_c (natural-order Cholesky), one wave
// per chain, lane = row.  Any output may be NULL.
__global__ void __launch_bounds__(64) k_small_spd_ops(int64_t C, int k, const double* A, const double* v, double* Av_out,
                                                      double* quad_out, double* logdet_out, long long* bad) {
  extern __shared__ double sm[];  // L: k x (k+1), then vec[k]
  const int64_t c = blockIdx.x;
  const int lane = threadIdx.x;
  const int ld = k + 1;
  double* L = sm;
  double* vec = sm + (int64_t)k * ld;
  const double* Ac = A + c * k * k;
  if (lane < k) {
    for (int j = 0; j < k; ++j) L[lane * ld + j] = Ac[lane * k + j];
    vec[lane] = v ? v[c * k + lane] : 0.0;
  }
  __syncthreads();
  if (v && (Av_out || quad_out)) {
    double av = 0.0;
    if (lane < k)
      for (int j = 0; j < k; ++j) av = fma(L[lane * ld + j], vec[j], av);
    if (Av_out && lane < k) Av_out[c * k + lane] = av;
    if (quad_out) {
      double q = (lane < k) ? av * vec[lane] : 0.0;
      for (int s = 32; s >= 1; s >>= 1) q += __shfl_xor(q, s, 64);
      if (lane == 0) quad_out[c] = q;
    }
  }
  if (!logdet_out) return;
  __syncthreads();
  bool fail = false;
  double acc = 0.0;
  for (int j = 0; j < k; ++j) {
    const double d = L[j * ld + j];
    if (!(d > 0.0)) { fail = true; break; }
    const double sd = sqrt(d);
    acc += log(d);
    double lij = 0.0;
    if (lane > j && lane < k) lij = L[lane * ld + j] / sd;
    __syncthreads();
    if (lane > j && lane < k) vec[lane] = lij;
    __syncthreads();
    if (lane > j && lane < k)
      for (int t = j + 1; t <= lane; ++t) L[lane * ld + t] -= lij * vec[t];
    __syncthreads();
  }
  if (lane == 0) {
    logdet_out[c] = fail ? NAN : acc;
    if (fail) atomicMin((unsigned long long*)bad, (unsigned long long)c);
  }
}

extern "C" {

omc_status omc_rw_propose(omc_ctx* ctx, int64_t p, const double* x, int64_t x_chain_stride, int64_t x_elem_stride,
                          const double* step, int64_t step_elem_stride, const double* lower, const double* upper,
                          const double* count, int64_t index, const double* draw_inject, uint64_t draw_index,
                          uint32_t sub, double* z, int64_t z_chain_stride, int64_t z_elem_stride, double* lq_fwd,
                          double* lq_rev) {
  if (!ctx || p < 1 || !x || !step || !z || !lq_fwd || !lq_rev || ((lower == nullptr) != (upper == nullptr)))
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_rw_propose, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, p, x, x_chain_stride, x_elem_stride, step, step_elem_stride, lower, upper, count,
                     index, draw_inject, omc_make_key(ctx->seed, draw_index, lower ? OMC_RNG_UNIFORM : OMC_RNG_NORMAL), sub,
                     z, z_chain_stride, z_elem_stride, lq_fwd, lq_rev);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_mala_diag(omc_ctx* ctx, int64_t kmax, const double* x, const double* grad, const double* hdiag,
                         const double* count, double step, int32_t propose, const double* z_inject, uint64_t draw_index,
                         uint32_t sub, double* x_other, double* lq) {
  if (!ctx || kmax < 1 || !x || !grad || !hdiag || !(step > 0.0) || !x_other || !lq) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_mala_diag, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains, ctx->chain_offset,
                     kmax, x, grad, hdiag, count, step, (int)propose, z_inject,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), sub, x_other, lq, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_mh_accept(omc_ctx* ctx, const double* lp_cur, const double* lp_prop, const double* lq_fwd,
                         const double* lq_rev, const double* count, int64_t index, const double* u_inject,
                         uint64_t draw_index, uint32_t sub, int32_t* accept, double* log_alpha, int64_t* accept_count,
                         int64_t* proposal_count) {
  if (!ctx || !lp_cur || !lp_prop || !accept) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_mh_accept, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, lp_cur, lp_prop, lq_fwd, lq_rev, count, index, u_inject,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), sub, (int*)accept, log_alpha,
                     (long long*)accept_count, (long long*)proposal_count);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_chain_select(omc_ctx* ctx, const int32_t* accept, int64_t width, const double* src, double* dst) {
  if (!ctx || !accept || width < 1 || !src || !dst) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  unsigned gx = grid1(width, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_chain_select, dim3(gx, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, ctx->n_chains, width,
                     (const int*)accept, src, dst);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_chain_select_multi(omc_ctx* ctx, const int32_t* accept, int32_t n_items, const int64_t* widths,
                                  const double* const* srcs, double* const* dsts) {
  if (!ctx || !accept || n_items < 1 || n_items > OMC_SELECT_MAX || !widths || !srcs || !dsts) return OMC_INVALID_ARG;
  SelectItems it{};
  it.n = n_items;
  int64_t wmax = 1;
  for (int e = 0; e < OMC_SELECT_MAX; ++e) {
    const bool on = e < n_items;
    if (on && (widths[e] < 1 || !srcs[e] || !dsts[e])) return OMC_INVALID_ARG;
    it.width[e] = on ? widths[e] : 0;
    it.src[e] = on ? srcs[e] : nullptr;
    it.dst[e] = on ? dsts[e] : nullptr;
    if (on && widths[e] > wmax) wmax = widths[e];
  }
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  // few workgroups per (chain, entry): most chains reject and their workgroups only cost their dispatch
  unsigned gx = grid1(wmax, 256 * 16);
  if (gx > 16) gx = 16;
  hipLaunchKernelGGL(k_chain_select_multi, dim3(gx, (unsigned)ctx->n_chains, (unsigned)n_items), dim3(256), 0, ctx->stream,
                     ctx->n_chains, (const int*)accept, it);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_ragged_resize(omc_ctx* ctx, int64_t rows, int64_t kmax, const double* count, const int32_t* birth,
                             const int64_t* del_index, const double* new_vals, const double* src, double* dst,
                             int64_t chain_stride, int64_t row_stride, int64_t col_stride) {
  if (!ctx || rows < 1 || kmax < 1 || !count || !birth || !del_index || !src || !dst || src == dst)
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  unsigned gx = grid1(rows * kmax, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_ragged_resize, dim3(gx, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, ctx->n_chains, rows,
                     kmax, count, (const int*)birth, (const long long*)del_index, new_vals, src, dst, chain_stride,
                     row_stride, col_stride);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_gaussian_basis(omc_ctx* ctx, int64_t n, int64_t kmax, const double* X, const double* knots,
                              const double* scales, double scale0, const double* count, const double* prev_count,
                              int64_t column, double* B) {
  if (!ctx || n < 1 || kmax < 1 || !X || !knots || !B || column >= kmax || (!scales && !(scale0 > 0.0)) ||
      (prev_count && !count))
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  unsigned gx = grid1(n, 256);
  if (gx > 32) gx = 32;
  hipLaunchKernelGGL(k_gaussian_basis, dim3(gx, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, ctx->n_chains, n, kmax,
                     X, knots, scales, scale0, count, prev_count, column, B);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_design_predict_batched(omc_ctx* ctx, int64_t n, int64_t kmax, const double* B, const double* coef,
                                      const double* add_chain, const double* add_shared, double alpha,
                                      const double* chain_scale, double* out) {
  if (!ctx || n < 1 || kmax < 1 || !B || !coef || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const uintptr_t align = (uintptr_t)B | (uintptr_t)out | (uintptr_t)add_chain | (uintptr_t)add_shared;
  if (n % 2 == 0 && (align & 15u) == 0) {
    unsigned gx = grid1(n / 2, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_design_predict_batched2, dim3(gx, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream,
                       ctx->n_chains, n, kmax, B, coef, add_chain, add_shared, alpha, chain_scale, out);
  } else {
    unsigned gx = grid1(n, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_design_predict_batched, dim3(gx, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream,
                       ctx->n_chains, n, kmax, B, coef, add_chain, add_shared, alpha, chain_scale, out);
  }
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_design_resid_sq_batched(omc_ctx* ctx, int64_t n, int64_t kmax, const double* B, const double* coef,
                                       const double* add_chain, const double* add_shared, const double* y,
                                       const double* w, double* out) {
  if (!ctx || n < 1 || kmax < 1 || !B || !coef || !y || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const int64_t C = ctx->n_chains;
  const int64_t parts = (n + 256 * RS_ROWS - 1) / (256 * RS_ROWS);
  if (parts > 65535) return OMC_UNSUPPORTED;
  if (parts == 1 || C >= 512) {  // one workgroup per chain, straight into out[c]
    hipLaunchKernelGGL(k_design_resid_sq, dim3(1u, (unsigned)C), dim3(256), 0, ctx->stream, n, kmax, B, coef, add_chain,
                       add_shared, y, w, out);
    OMC_HIP_CHECK(hipGetLastError());
    return OMC_OK;
  }
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->mh_work, &ctx->mh_work_bytes, (size_t)C * parts * sizeof(double));
  if (st != OMC_OK) return st;
  hipLaunchKernelGGL(k_design_resid_sq, dim3((unsigned)parts, (unsigned)C), dim3(256), 0, ctx->stream, n, kmax, B, coef,
                     add_chain, add_shared, y, w, ctx->mh_work);
  hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, ctx->stream, C, (int64_t)1, (int)parts,
                     (const double*)ctx->mh_work, out, (int64_t)0, (const double*)nullptr, (double*)nullptr);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

static omc_status design_gram_launch(omc_ctx* ctx, int64_t n, int64_t kmax, const double* B, const double* w,
                                     const double* resid_shared, const double* resid_chain, const double* count,
                                     double* gram, double* rhs, const double* B_alt, const double* count_alt,
                                     const int* select) {
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const size_t lds = (size_t)((kmax + 1) * (GRAM_TR + 1) + GRAM_TR) * sizeof(double);
  const int64_t C = ctx->n_chains;
  // rows in parts of >= 4 tiles, enough parts to put ~8 workgroups on every CU, at most 16
  int64_t parts = (n / (4 * GRAM_TR));
  const int64_t want = (8 * 256 + C - 1) / C;
  if (parts > want) parts = want;
  if (parts > 16) parts = 16;
  if (parts < 2) {
    hipLaunchKernelGGL(k_design_gram_batched, dim3((unsigned)C), dim3(256), lds, ctx->stream, n, kmax, B, w, resid_shared,
                       resid_chain, count, gram, rhs, B_alt, count_alt, select);
  } else {
    const size_t per = (size_t)(kmax * kmax + kmax);
    omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->mh_work, &ctx->mh_work_bytes, (size_t)C * parts * per * sizeof(double));
    if (st != OMC_OK) return st;
    double* gp = ctx->mh_work;
    double* rp = rhs ? gp + (size_t)C * parts * kmax * kmax : nullptr;
    hipLaunchKernelGGL(k_design_gram_batched, dim3((unsigned)C, (unsigned)parts), dim3(256), lds, ctx->stream, n, kmax, B, w,
                       resid_shared, resid_chain, count, gp, rp, B_alt, count_alt, select);
    const int64_t lg = kmax * kmax, total = C * lg + (rhs ? C * kmax : 0);
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, C, lg, (int)parts, gp,
                       gram, kmax, (const double*)rp, rhs);
  }
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_design_gram_batched(omc_ctx* ctx, int64_t n, int64_t kmax, const double* B, const double* w,
                                   const double* resid_shared, const double* resid_chain, const double* count,
                                   double* gram, double* rhs) {
  if (!ctx || n < 1 || kmax < 1 || kmax > 36 || !B || !gram) return OMC_INVALID_ARG;
  return design_gram_launch(ctx, n, kmax, B, w, resid_shared, resid_chain, count, gram, rhs, nullptr, nullptr, nullptr);
}

omc_status omc_design_gram_select(omc_ctx* ctx, int64_t n, int64_t kmax, const double* B, const double* count,
                                  const double* B_alt, const double* count_alt, const int32_t* select, const double* w,
                                  double* gram) {
  if (!ctx || n < 1 || kmax < 1 || kmax > 36 || !B || !B_alt || !select || !gram) return OMC_INVALID_ARG;
  return design_gram_launch(ctx, n, kmax, B, w, nullptr, nullptr, count, gram, nullptr, B_alt, count_alt, (const int*)select);
}

omc_status omc_small_sample_canonical(omc_ctx* ctx, int64_t kmax, const double* gram, const double* gram_rhs,
                                      const double* lik_scale, const double* prior_prec, const double* prior_mean,
                                      const double* count, const double* z_inject, uint64_t draw_index, double* x,
                                      double* mu) {
  if (!ctx || kmax < 1 || kmax > SMALL_KMAX || !gram || !gram_rhs || !prior_prec || !x) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const size_t lds = (size_t)(kmax * (kmax + 1) + kmax + 2) * sizeof(double);
  hipLaunchKernelGGL(k_small_sample_canonical, dim3((unsigned)ctx->n_chains), dim3(64), lds, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, (int)kmax, gram, gram_rhs, lik_scale, prior_prec, prior_mean, count, z_inject,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), x, mu, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_rj_matched_transition(omc_ctx* ctx, int64_t kmax, const double* gram_cur, const double* gram_prop,
                                     const double* count, const int32_t* birth, const int64_t* del_index,
                                     const double* coef_cur, double scale, int32_t has_limits, double lim_lo,
                                     double lim_hi, const double* draw_inject, uint64_t draw_index, uint32_t sub,
                                     double* coef_prop, double* lq_fwd, double* lq_rev) {
  if (!ctx || kmax < 1 || kmax > SMALL_KMAX || !gram_cur || !gram_prop || !count || !birth || !del_index || !coef_cur ||
      !(scale > 0.0) || !coef_prop || !lq_fwd || !lq_rev)
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const size_t lds = (size_t)(kmax * (2 * kmax + 1) + kmax * (kmax + 2)) * sizeof(double);
  hipLaunchKernelGGL(k_rj_matched, dim3((unsigned)ctx->n_chains), dim3(64), lds, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, (int)kmax, gram_cur, gram_prop, count, (const int*)birth,
                     (const long long*)del_index, coef_cur, scale, (int)has_limits, lim_lo, lim_hi, draw_inject,
                     omc_make_key(ctx->seed, draw_index, has_limits ? OMC_RNG_UNIFORM : OMC_RNG_NORMAL), sub, coef_prop,
                     lq_fwd, lq_rev);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_small_spd_ops(omc_ctx* ctx, int64_t k, const double* A, const double* v, double* Av_out, double* quad_out,
                             double* logdet_out) {
  if (!ctx || k < 1 || k > SMALL_KMAX || !A || ((Av_out || quad_out) && !v)) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const size_t lds = (size_t)(k * (k + 1) + k) * sizeof(double);
  hipLaunchKernelGGL(k_small_spd_ops, dim3((unsigned)ctx->n_chains), dim3(64), lds, ctx->stream, ctx->n_chains, (int)k, A, v, Av_out,
                     quad_out, logdet_out, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"
