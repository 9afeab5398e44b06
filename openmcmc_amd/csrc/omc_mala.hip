// Metropolis-Hastings steps (ManifoldMALA, RandomWalk) for a Gaussian target with shared constant
// Hessian, batched over chains as level-3 BLAS on the d x C state matrix (rocBLAS: fp64 MFMA).
// Reference: sampler/metropolis_hastings.py:102-173, 212-269, 301-373; location_scale.py:222-232.
#include <math.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "omc_common.h"

#define OMC_BLAS_CHECK(expr)                                   \
  do {                                                         \
    rocblas_status _s = (expr);                                \
    if (_s != rocblas_status_success) {                        \
      omc_set_error(#expr, hipErrorUnknown);                   \
      return OMC_HIP_ERROR;                                    \
    }                                                          \
  } while (0)

omc_status omc_ensure_blas(omc_ctx* ctx);
omc_status omc_ensure_bytes(omc_ctx* ctx, void** buf, size_t* have, size_t need);

static unsigned gx(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

// out[c][i] = a[c][i] - mu[i]   (mu NULL = 0)
__global__ void k_sub_shared(int64_t d, const double* a, int64_t ld_a, const double* mu, double* out, int64_t ld_o) {
  const int64_t c = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d; i += (int64_t)gridDim.x * blockDim.x)
    out[c * ld_o + i] = a[c * ld_a + i] - (mu ? mu[i] : 0.0);
}
// out = a + alpha * b
__global__ void k_axpby(int64_t d, const double* a, int64_t ld_a, double alpha, const double* b, int64_t ld_b, double* out,
                        int64_t ld_o) {
  const int64_t c = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d; i += (int64_t)gridDim.x * blockDim.x)
    out[c * ld_o + i] = fma(alpha, b[c * ld_b + i], a[c * ld_a + i]);
}
// z[c][:] from the injected array or the chain's normal stream
__global__ void k_draw_normals(int64_t d, int64_t chain_offset, omc_rng_key key, const double* zin, int64_t ld_z,
                               double* z, int64_t ld_o) {
  const int64_t c = blockIdx.y;
  const int64_t npairs = (d + 1) / 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * blockDim.x) {
    double z0, z1;
    if (zin) {
      z0 = zin[c * ld_z + 2 * q];
      z1 = (2 * q + 1 < d) ? zin[c * ld_z + 2 * q + 1] : 0.0;
    } else {
      omc_normal_pair(omc_rng_block(key, chain_offset + c, (uint32_t)q), z0, z1);
    }
    z[c * ld_o + 2 * q] = z0;
    if (2 * q + 1 < d) z[c * ld_o + 2 * q + 1] = z1;
  }
}
// out[c] = sum_i a[c][i]^2
__global__ void __launch_bounds__(256) k_colsumsq(int64_t d, const double* a, int64_t ld_a, double* out) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < d; i += blockDim.x) acc = fma(a[c * ld_a + i], a[c * ld_a + i], acc);
  for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = red[0] + red[1] + red[2] + red[3];
}
// accept/reject: one lane per chain decides, the flag is broadcast for the copy kernel
__global__ void k_mh_decide(int64_t C, int64_t chain_offset, omc_rng_key key, const double* u_in, double dnum,
                            const double* sumlogL, double log_step_term, int mala, const double* ss_fwd,
                            const double* ss_rev, const double* ss_cur, const double* ss_prop, double lp_scale,
                            int* flag, long long* acc_cnt, long long* prop_cnt) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double sl = sumlogL[0];
  // log p (gmrf.py:339-344) with L_Q = step * L:  |L_Q'(x-mu)|^2 = lp_scale * ss
  const double logdetQ = 2.0 * (sl + log_step_term);
  const double lp_cur = 0.5 * (logdetQ - dnum * 1.8378770664093453 - lp_scale * ss_cur[c]);
  const double lp_prop = 0.5 * (logdetQ - dnum * 1.8378770664093453 - lp_scale * ss_prop[c]);
  double log_alpha = lp_prop - lp_cur;
  if (mala) {
    const double lq_fwd = sl - 0.5 * ss_fwd[c], lq_rev = sl - 0.5 * ss_rev[c];  // metropolis_hastings.py:372-373
    log_alpha = lp_prop + lq_rev - (lp_cur + lq_fwd);                            // :155
  }
  double u;
  if (u_in) {
    u = u_in[c];
  } else {
    const uint4 w = omc_rng_block(key, chain_offset + c, 0u);
    u = omc_u53(w.x, w.y);
  }
  const int ok = log(u) < log_alpha;  // :173
  flag[c] = ok;
  if (prop_cnt) prop_cnt[c] += 1;
  if (acc_cnt && ok) acc_cnt[c] += 1;
}
__global__ void k_select_rows(int64_t d, const int* flag, const double* prop, int64_t ld_p, double* x, int64_t ld_x) {
  const int64_t c = blockIdx.y;
  if (!flag[c]) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d; i += (int64_t)gridDim.x * blockDim.x)
    x[c * ld_x + i] = prop[c * ld_p + i];
}
__global__ void k_sumlogdiag(int64_t d, const double* L, double* out) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < d; i += blockDim.x) acc += log(L[i * d + i]);
  for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_scale_copy(int64_t total, const double* a, double scale, double* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = scale * a[i];
}
__global__ void k_latch_info(const int* info, long long* bad) {
  if (info[0] != 0) atomicMin((unsigned long long*)bad, 0ull);
}

struct MhWork {
  double *R, *G, *M, *V, *XP, *T, *ss;  // d x C matrices (ld = d) and 4*C scalars
  int* flag;
};

static omc_status mh_workspace(omc_ctx* ctx, int64_t d, MhWork* w) {
  const int64_t C = ctx->n_chains;
  const size_t mat = (size_t)C * d;
  const size_t need = (13 * mat + 4 * (size_t)C) * sizeof(double) + (size_t)C * sizeof(int) + 64;
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->mh_work, &ctx->mh_work_bytes, need);
  if (st != OMC_OK) return st;
  double* base = ctx->mh_work;
  w->R = base; w->G = base + mat; w->M = base + 2 * mat; w->V = base + 3 * mat; w->XP = base + 4 * mat;
  w->T = base + 5 * mat;  // 8 * mat: T4 and its TRMM image
  w->ss = base + 13 * mat;
  w->flag = (int*)(w->ss + 4 * C);
  return OMC_OK;
}

extern "C" {

omc_status omc_dense_cholesky(omc_ctx* ctx, int64_t d, const double* A, double scale, double* L_out,
                              double* sumlogdiag_out) {
  if (!ctx || d < 1 || d > 32768 || !A || !L_out || !(scale > 0.0)) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  st = omc_ensure_bytes(ctx, (void**)&ctx->dense_info, &ctx->dense_info_bytes, sizeof(int) * (size_t)(ctx->n_chains > 1 ? ctx->n_chains : 1));
  if (st != OMC_OK) return st;
  hipLaunchKernelGGL(k_scale_copy, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d * d, A, scale, L_out);
  OMC_HIP_CHECK(hipGetLastError());
  OMC_BLAS_CHECK(rocsolver_dpotrf((rocblas_handle)ctx->blas, rocblas_fill_lower, (rocblas_int)d, L_out, (rocblas_int)d,
                                  ctx->dense_info));
  hipLaunchKernelGGL(k_latch_info, dim3(1), dim3(1), 0, ctx->stream, ctx->dense_info, ctx->d_bad_chain);
  if (sumlogdiag_out) hipLaunchKernelGGL(k_sumlogdiag, dim3(1), dim3(256), 0, ctx->stream, d, L_out, sumlogdiag_out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

// Shared constant Hessian: the drift matrix A1 = -(L L')^{-1} Q (one potrs) and the explicit L^{-T}
// (one trsm on the identity) are formed once per (Q, L, step) and cached in the context; every step
// is then 3 GEMMs + 1 TRMM instead of 2 GEMMs + 5 TRSMs + 4 TRMMs (TRSM with 512 right-hand sides is
// ~10x slower than the GEMM of the same shape, and the old sequence was launch-bound).
__global__ void k_set_identity(int64_t d, double* A) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d * d; i += (int64_t)gridDim.x * blockDim.x)
    A[i] = (i / d == i % d) ? 1.0 : 0.0;
}
// T4 columns: [ R | XP - M | R' | X - M' ]  (each d x C), from X, mu, XP, M (current) and M' (proposed)
__global__ void k_build_t4(int64_t d, int64_t C, const double* x, int64_t ld_x, const double* mu, const double* xp,
                           const double* m_cur, const double* m_prop, double* t4) {
  const int64_t c = blockIdx.y;
  const int64_t blk = C * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d; i += (int64_t)gridDim.x * blockDim.x) {
    const double xv = x[c * ld_x + i], pv = xp[c * d + i], mv = mu ? mu[i] : 0.0;
    t4[c * d + i] = xv - mv;
    t4[blk + c * d + i] = pv - m_cur[c * d + i];
    t4[2 * blk + c * d + i] = pv - mv;
    t4[3 * blk + c * d + i] = xv - m_prop[c * d + i];
  }
}

static omc_status mala_prepare(omc_ctx* ctx, int64_t d, const double* Q, const double* L, double step) {
  if (ctx->mala_Q == Q && ctx->mala_L == L && ctx->mala_step == step && ctx->mala_d == d && ctx->mala_prep) return OMC_OK;
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->mala_prep, &ctx->mala_prep_bytes, (size_t)3 * d * d * sizeof(double));
  if (st != OMC_OK) return st;
  rocblas_handle h = (rocblas_handle)ctx->blas;
  const rocblas_int di = (rocblas_int)d;
  double* A1 = ctx->mala_prep;            // -(L L')^{-1} Q
  double* LinvT = ctx->mala_prep + d * d; // L^{-T}
  double* Lc = ctx->mala_prep + 2 * d * d; // scratch copy of L (potrs may not alias its factor argument)
  hipLaunchKernelGGL(k_scale_copy, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d * d, Q, -1.0, A1);
  hipLaunchKernelGGL(k_scale_copy, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d * d, L, 1.0, Lc);
  OMC_BLAS_CHECK(rocsolver_dpotrs(h, rocblas_fill_lower, di, di, Lc, di, A1, di));
  hipLaunchKernelGGL(k_set_identity, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d, LinvT);
  const double one = 1.0;
  OMC_BLAS_CHECK(rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                               rocblas_diagonal_non_unit, di, di, &one, L, di, LinvT, di));
  OMC_HIP_CHECK(hipGetLastError());
  ctx->mala_Q = Q; ctx->mala_L = L; ctx->mala_step = step; ctx->mala_d = d;
  return OMC_OK;
}

omc_status omc_mala_step(omc_ctx* ctx, int64_t d, const double* Q, const double* mu, const double* L,
                         const double* sumlogL, double step, const double* z_inject, int64_t ld_z,
                         const double* u_inject, uint64_t draw_index, double* x, int64_t ld_x, int64_t* accept_count,
                         int64_t* proposal_count) {
  if (!ctx || d < 1 || !Q || !L || !sumlogL || !x || ld_x < d || (z_inject && ld_z < d) || !(step > 0.0))
    return OMC_INVALID_ARG;
  const int64_t C = ctx->n_chains;
  if (C > 65535 || 4 * C > 0x7fffffffLL) return OMC_UNSUPPORTED;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  MhWork w;
  st = mh_workspace(ctx, d, &w);
  if (st != OMC_OK) return st;
  st = mala_prepare(ctx, d, Q, L, step);
  if (st != OMC_OK) return st;
  const double* A1 = ctx->mala_prep;
  const double* LinvT = ctx->mala_prep + d * d;
  rocblas_handle h = (rocblas_handle)ctx->blas;
  const rocblas_int di = (rocblas_int)d, Ci = (rocblas_int)C;
  const double one = 1.0, zero = 0.0;
  const dim3 g2(gx(d) > 8 ? 8 : gx(d), (unsigned)C), b2(256);
  hipStream_t s = ctx->stream;
  // workspace roles: R (residual), G (drift), M (current mean), V (draws / M'), XP (proposal), T4 (4 d x C)
  double* T4 = w.T;  // 4 * C * d doubles (mh_workspace sizes T for it)
  double* N4 = w.T + 4 * C * d;
  double *ss = w.ss;  // [4][C]: |L'R|^2, |L'(x'-M)|^2, |L'R'|^2, |L'(x-M')|^2

  // current state: M = x + 1/2 A1 (x - mu)
  hipLaunchKernelGGL(k_sub_shared, g2, b2, 0, s, d, x, ld_x, mu, w.R, d);
  OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, di, Ci, di, &one, A1, di, w.R, di, &zero,
                               w.G, di));
  hipLaunchKernelGGL(k_axpby, g2, b2, 0, s, d, x, ld_x, 0.5, w.G, d, w.M, d);
  // proposal: x' = M + L^{-T} z
  hipLaunchKernelGGL(k_draw_normals, dim3(gx((d + 1) / 2) > 8 ? 8 : gx((d + 1) / 2), (unsigned)C), b2, 0, s, d,
                     ctx->chain_offset, omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), z_inject, ld_z, w.R, d);
  OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, di, Ci, di, &one, LinvT, di, w.R, di,
                               &zero, w.V, di));
  hipLaunchKernelGGL(k_axpby, g2, b2, 0, s, d, w.M, d, 1.0, w.V, d, w.XP, d);
  // proposed state: M' = x' + 1/2 A1 (x' - mu)
  hipLaunchKernelGGL(k_sub_shared, g2, b2, 0, s, d, w.XP, d, mu, w.R, d);
  OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, di, Ci, di, &one, A1, di, w.R, di, &zero,
                               w.G, di));
  hipLaunchKernelGGL(k_axpby, g2, b2, 0, s, d, w.XP, d, 0.5, w.G, d, w.V, d);  // V = M'
  // the four quadratic forms |L'(.)|^2 in one TRMM on [R | x'-M | R' | x-M']
  hipLaunchKernelGGL(k_build_t4, g2, b2, 0, s, d, C, x, ld_x, mu, w.XP, w.M, w.V, T4);
  OMC_BLAS_CHECK(rocblas_dtrmm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                               rocblas_diagonal_non_unit, di, 4 * Ci, &one, L, di, T4, di, N4, di));
  hipLaunchKernelGGL(k_colsumsq, dim3((unsigned)(4 * C)), dim3(256), 0, s, d, N4, d, ss);
  // accept / reject.  L = chol(Q / step^2) => chol(Q) = step * L
  hipLaunchKernelGGL(k_mh_decide, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, C, ctx->chain_offset,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), u_inject, (double)d, sumlogL,
                     (double)d * log(step), 1, ss + C, ss + 3 * C, ss, ss + 2 * C, step * step, w.flag,
                     (long long*)accept_count, (long long*)proposal_count);
  hipLaunchKernelGGL(k_select_rows, g2, b2, 0, s, d, w.flag, w.XP, d, x, ld_x);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_rw_step(omc_ctx* ctx, int64_t d, const double* mu, const double* LQ, const double* sumlogLQ, double step,
                       const double* z_inject, int64_t ld_z, const double* u_inject, uint64_t draw_index, double* x,
                       int64_t ld_x, int64_t* accept_count, int64_t* proposal_count) {
  if (!ctx || d < 1 || !LQ || !sumlogLQ || !x || ld_x < d || (z_inject && ld_z < d) || !(step > 0.0))
    return OMC_INVALID_ARG;
  const int64_t C = ctx->n_chains;
  if (C > 65535) return OMC_UNSUPPORTED;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  MhWork w;
  st = mh_workspace(ctx, d, &w);
  if (st != OMC_OK) return st;
  rocblas_handle h = (rocblas_handle)ctx->blas;
  const rocblas_int di = (rocblas_int)d, Ci = (rocblas_int)C;
  const double one = 1.0;
  const dim3 g2(gx(d) > 8 ? 8 : gx(d), (unsigned)C), b2(256);
  hipStream_t s = ctx->stream;
  double *ss_cur = w.ss + 2 * C, *ss_prop = w.ss + 3 * C;
  hipLaunchKernelGGL(k_sub_shared, g2, b2, 0, s, d, x, ld_x, mu, w.R, d);
  OMC_BLAS_CHECK(rocblas_dtrmm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                               rocblas_diagonal_non_unit, di, Ci, &one, LQ, di, w.R, di, w.T, di));
  hipLaunchKernelGGL(k_colsumsq, dim3((unsigned)C), dim3(256), 0, s, d, w.T, d, ss_cur);
  hipLaunchKernelGGL(k_draw_normals, dim3(gx((d + 1) / 2) > 8 ? 8 : gx((d + 1) / 2), (unsigned)C), b2, 0, s, d,
                     ctx->chain_offset, omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), z_inject, ld_z, w.V, d);
  hipLaunchKernelGGL(k_axpby, g2, b2, 0, s, d, x, ld_x, step, w.V, d, w.XP, d);  // :250
  hipLaunchKernelGGL(k_sub_shared, g2, b2, 0, s, d, w.XP, d, mu, w.R, d);
  OMC_BLAS_CHECK(rocblas_dtrmm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                               rocblas_diagonal_non_unit, di, Ci, &one, LQ, di, w.R, di, w.T, di));
  hipLaunchKernelGGL(k_colsumsq, dim3((unsigned)C), dim3(256), 0, s, d, w.T, d, ss_prop);
  hipLaunchKernelGGL(k_mh_decide, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, C, ctx->chain_offset,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), u_inject, (double)d, sumlogLQ, 0.0, 0,
                     ss_cur, ss_cur, ss_cur, ss_prop, 1.0, w.flag, (long long*)accept_count, (long long*)proposal_count);
  hipLaunchKernelGGL(k_select_rows, g2, b2, 0, s, d, w.flag, w.XP, d, x, ld_x);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"
