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

// Shared constant Hessian: the drift matrix A1 = -(L L')^{-1} Q (one potrs), A1p = I + A1/2 and the explicit
// L^{-T} (one trsm on the identity) are formed once per (Q, L, step) and cached in the context.  The proposal mean
// is then m(x) = A1p x + c0, and a step is 3 GEMMs (x' = A1p x + L^{-T} z accumulated in place, m' = A1p x'),
// one TRMM for the three quadratic forms that need L' (the fourth, |L'(x' - m)|^2, is |z|^2), and three small
// kernels (draw, build, finish) -- 7 launches instead of the first version's 23 (2 GEMMs + 5 TRSMs + 4 TRMMs;
// TRSM with 512 right-hand sides is ~10x slower than the GEMM of the same shape).
__global__ void k_set_identity(int64_t d, double* A) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d * d; i += (int64_t)gridDim.x * blockDim.x)
    A[i] = (i / d == i % d) ? 1.0 : 0.0;
}
// A1p = I + 0.5 * A1  (the proposal mean of a chain is then ONE matrix product: m = A1p x + c0, c0 = -0.5 A1 mu)
__global__ void k_half_plus_identity(int64_t d, const double* A1, double* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d * d; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = 0.5 * A1[i] + ((i / d == i % d) ? 1.0 : 0.0);
}
// a[c][:] += v
__global__ void k_add_shared(int64_t d, double* a, int64_t ld_a, const double* v) {
  const int64_t c = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d; i += (int64_t)gridDim.x * blockDim.x)
    a[c * ld_a + i] += v[i];
}
// T3 columns: [ x - mu | x' - mu | x - m' ]  with m' = mp + c0 (each block d x C)
__global__ void k_build_t3(int64_t d, int64_t C, const double* x, int64_t ld_x, const double* mu, const double* c0,
                           const double* xp, const double* mp, double* t3) {
  const int64_t c = blockIdx.y;
  const int64_t blk = C * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d; i += (int64_t)gridDim.x * blockDim.x) {
    const double xv = x[c * ld_x + i], pv = xp[c * d + i], mv = mu ? mu[i] : 0.0;
    t3[c * d + i] = xv - mv;
    t3[blk + c * d + i] = pv - mv;
    t3[2 * blk + c * d + i] = xv - (mp[c * d + i] + (c0 ? c0[i] : 0.0));
  }
}
// random walk: x' = x + step * z (two roundings like numpy's mu + z*step), T2 = [ x - mu | x' - mu ]
__global__ void k_rw_build(int64_t d, int64_t C, const double* x, int64_t ld_x, const double* mu, double step,
                           const double* z, double* xp, double* t2) {
  const int64_t c = blockIdx.y;
  const int64_t blk = C * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d; i += (int64_t)gridDim.x * blockDim.x) {
    const double xv = x[c * ld_x + i], mv = mu ? mu[i] : 0.0;
    double pv;
    {
#pragma clang fp contract(off)
      const double prod = z[c * d + i] * step;
      pv = xv + prod;
    }
    xp[c * d + i] = pv;
    t2[c * d + i] = xv - mv;
    t2[blk + c * d + i] = pv - mv;
  }
}
// One workgroup per chain: the squared norms the decision needs, the decision, and the move of the accepted
// proposal into the state (metropolis_hastings.py:127-173).  n_cur / n_prop / n_rev: columns of L'(.) (n_rev and z
// NULL for the symmetric random walk); z: the proposal's N(0, I) draw, |z|^2 = |L'(x' - m)|^2.
__global__ void __launch_bounds__(256) k_mh_finish(int64_t d, int64_t chain_offset, omc_rng_key key, const double* u_in,
                                                   const double* sumlogL, double log_step_term, double lp_scale,
                                                   const double* n_cur, const double* n_prop, const double* n_rev,
                                                   const double* z, const double* xp, double* x, int64_t ld_x,
                                                   long long* acc_cnt, long long* prop_cnt) {
  __shared__ double red[4][4];
  __shared__ int accept;
  const int64_t c = blockIdx.x;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int64_t i = threadIdx.x; i < d; i += 256) {
    const double v0 = n_cur[c * d + i], v1 = n_prop[c * d + i];
    a0 = fma(v0, v0, a0);
    a1 = fma(v1, v1, a1);
    if (n_rev) {
      const double v2 = n_rev[c * d + i], v3 = z[c * d + i];
      a2 = fma(v2, v2, a2);
      a3 = fma(v3, v3, a3);
    }
  }
  for (int s = 32; s >= 1; s >>= 1) {
    a0 += __shfl_xor(a0, s, 64); a1 += __shfl_xor(a1, s, 64);
    a2 += __shfl_xor(a2, s, 64); a3 += __shfl_xor(a3, s, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red[0][w] = a0; red[1][w] = a1; red[2][w] = a2; red[3][w] = a3;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ss_cur = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double ss_prop = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double sl = sumlogL[0];
    const double logdetQ = 2.0 * (sl + log_step_term);
    const double dnum = (double)d;
    // log p (gmrf.py:339-344) with L_Q = step * L:  |L_Q'(x-mu)|^2 = lp_scale * ss
    const double lp_cur = 0.5 * (logdetQ - dnum * 1.8378770664093453 - lp_scale * ss_cur);
    const double lp_prop = 0.5 * (logdetQ - dnum * 1.8378770664093453 - lp_scale * ss_prop);
    double log_alpha = lp_prop - lp_cur;
    if (n_rev) {
      const double ss_rev = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
      const double ss_fwd = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
      const double lq_fwd = sl - 0.5 * ss_fwd, lq_rev = sl - 0.5 * ss_rev;  // metropolis_hastings.py:372-373
      log_alpha = lp_prop + lq_rev - (lp_cur + lq_fwd);                      // :155
    }
    double u;
    if (u_in) {
      u = u_in[c];
    } else {
      const uint4 w = omc_rng_block(key, chain_offset + c, 0u);
      u = omc_u53(w.x, w.y);
    }
    const int ok = log(u) < log_alpha;  // :173
    accept = ok;
    if (prop_cnt) prop_cnt[c] += 1;
    if (acc_cnt && ok) acc_cnt[c] += 1;
  }
  __syncthreads();
  if (accept)
    for (int64_t i = threadIdx.x; i < d; i += 256) x[c * ld_x + i] = xp[c * d + i];
}

// lower triangle of a column-major d x d matrix, zeros above the diagonal: the triangular products below run as
// plain GEMMs on this image (rocBLAS's out-of-place TRMM reached 5.6 TFLOP/s at d = 500, its DGEMM 37)
// (grid-stride like every kernel launched with gx(), which caps the grid at 4096 blocks: as plain one-element-per-thread
//  kernels these two left everything beyond 2^20 elements -- d > 1024 -- unwritten)
__global__ void k_lower_copy(int64_t d, const double* L, double* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d * d; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = i / d, row = i - col * d;
    out[i] = row >= col ? L[i] : 0.0;
  }
}

// U = L' as a dense upper-triangular column-major matrix (zeros below the diagonal): the own GEMM takes A as it stands
__global__ void k_lower_transpose(int64_t d, const double* L, double* out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < d * d; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = e / d, i = e - k * d;  // out(i, k) = L(k, i) for k >= i
    out[e] = k >= i ? L[i * d + k] : 0.0;
  }
}

static omc_status mala_prepare(omc_ctx* ctx, int64_t d, const double* Q, const double* L, double step) {
  if (ctx->mala_Q == Q && ctx->mala_L == L && ctx->mala_step == step && ctx->mala_d == d && ctx->mala_prep) return OMC_OK;
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->mala_prep, &ctx->mala_prep_bytes, (size_t)5 * d * d * sizeof(double));
  if (st != OMC_OK) return st;
  rocblas_handle h = (rocblas_handle)ctx->blas;
  const rocblas_int di = (rocblas_int)d;
  double* A1 = ctx->mala_prep;            // -(L L')^{-1} Q
  double* LinvT = ctx->mala_prep + d * d; // L^{-T}
  double* Lc = ctx->mala_prep + 2 * d * d; // copy of L with an explicitly zero upper triangle (potrs; the GEMM form of L'T)
  hipLaunchKernelGGL(k_scale_copy, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d * d, Q, -1.0, A1);
  hipLaunchKernelGGL(k_lower_copy, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d, L, Lc);
  OMC_BLAS_CHECK(rocsolver_dpotrs(h, rocblas_fill_lower, di, di, Lc, di, A1, di));
  hipLaunchKernelGGL(k_set_identity, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d, LinvT);
  const double one = 1.0;
  OMC_BLAS_CHECK(rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                               rocblas_diagonal_non_unit, di, di, &one, L, di, LinvT, di));
  hipLaunchKernelGGL(k_half_plus_identity, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d, A1, ctx->mala_prep + 3 * d * d);
  hipLaunchKernelGGL(k_lower_transpose, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d, L, ctx->mala_prep + 4 * d * d);
  OMC_HIP_CHECK(hipGetLastError());
  ctx->mala_Q = Q; ctx->mala_L = L; ctx->mala_step = step; ctx->mala_d = d;
  return OMC_OK;
}

omc_status omc_mh_invalidate(omc_ctx* ctx) {
  if (!ctx) return OMC_INVALID_ARG;
  ctx->mala_Q = nullptr; ctx->mala_L = nullptr; ctx->mala_step = 0.0; ctx->mala_d = 0;
  ctx->white_L = nullptr; ctx->white_mu = nullptr; ctx->white_d = 0; ctx->white_x = nullptr;
  ctx->rww_x = nullptr; ctx->rww_mu = nullptr;
  ctx->rw_LQ = nullptr; ctx->rw_d = 0;
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
  MhWork w{};
  st = mh_workspace(ctx, d, &w);
  if (st != OMC_OK) return st;
  st = mala_prepare(ctx, d, Q, L, step);
  if (st != OMC_OK) return st;
  const double* A1 = ctx->mala_prep;
  const double* LinvT = ctx->mala_prep + d * d;
  const double* A1p = ctx->mala_prep + 3 * d * d;  // I + A1/2
  const double* Lz = ctx->mala_prep + 2 * d * d;   // L, zeros above the diagonal
  rocblas_handle h = (rocblas_handle)ctx->blas;
  const rocblas_int di = (rocblas_int)d, Ci = (rocblas_int)C;
  const double one = 1.0, zero = 0.0, minus_half = -0.5;
  const dim3 g2(gx(d) > 8 ? 8 : gx(d), (unsigned)C), b2(256);
  hipStream_t s = ctx->stream;
  // workspace roles: Z (draw), XP (proposal x'), V (A1p x'), T3 = [x-mu | x'-mu | x-m'] and its image under L'
  double* Z = w.R;
  double* T3 = w.T;
  double* N3 = w.T + 3 * C * d;
  double* c0 = nullptr;
  if (mu) {  // c0 = -1/2 A1 mu: the constant part of the proposal mean m(x) = A1p x + c0
    c0 = w.G;
    OMC_BLAS_CHECK(rocblas_dgemv(h, rocblas_operation_none, di, di, &minus_half, A1, di, mu, 1, &zero, c0, 1));
  }
  // proposal: x' = m(x) + L^{-T} z = A1p x + L^{-T} z (+ c0): two GEMMs into the same buffer, no element-wise pass
  hipLaunchKernelGGL(k_draw_normals, dim3(gx((d + 1) / 2) > 8 ? 8 : gx((d + 1) / 2), (unsigned)C), b2, 0, s, d,
                     ctx->chain_offset, omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), z_inject, ld_z, Z, d);
  const bool own = !ctx->mh_use_rocblas && d <= 46340;
  const double* Lt = ctx->mala_prep + 4 * d * d;  // L' as a dense upper-triangular matrix
  if (own) {  // one launch: x' = A1p x + L^-T z + c0
    st = omc_dgemm_small(ctx, (int)d, (int)C, A1p, d, x, ld_x, (int)d, LinvT, d, Z, d, (int)d, 0, c0, w.XP, d);
    if (st != OMC_OK) return st;
    st = omc_dgemm_small(ctx, (int)d, (int)C, A1p, d, w.XP, d, (int)d, nullptr, 0, nullptr, 0, 0, 0, nullptr, w.V, d);
    if (st != OMC_OK) return st;
  } else {
    OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, di, Ci, di, &one, A1p, di, x,
                                 (rocblas_int)ld_x, &zero, w.XP, di));
    OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, di, Ci, di, &one, LinvT, di, Z, di, &one,
                                 w.XP, di));
    if (c0) hipLaunchKernelGGL(k_add_shared, g2, b2, 0, s, d, w.XP, d, c0);
    // proposed state's mean: m' = A1p x' (+ c0, added when T3 is built)
    OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, di, Ci, di, &one, A1p, di, w.XP, di, &zero,
                                 w.V, di));
  }
  // |L'(x - mu)|^2, |L'(x' - mu)|^2, |L'(x - m')|^2 in one product; |L'(x' - m)|^2 = |z|^2 needs none
  hipLaunchKernelGGL(k_build_t3, g2, b2, 0, s, d, C, x, ld_x, mu, c0, w.XP, w.V, T3);
  if (own) {
    st = omc_dgemm_small(ctx, (int)d, (int)(3 * C), Lt, d, T3, d, (int)d, nullptr, 0, nullptr, 0, 0, 1, nullptr, N3, d);
    if (st != OMC_OK) return st;
  } else {
    OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, di, 3 * Ci, di, &one, Lz, di, T3, di,
                                 &zero, N3, di));
  }
  // accept / reject and the move of accepted proposals.  L = chol(Q / step^2) => chol(Q) = step * L
  hipLaunchKernelGGL(k_mh_finish, dim3((unsigned)C), dim3(256), 0, s, d, ctx->chain_offset,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), u_inject, sumlogL, (double)d * log(step),
                     step * step, N3, N3 + C * d, N3 + 2 * C * d, Z, w.XP, x, ld_x, (long long*)accept_count,
                     (long long*)proposal_count);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

// ---- the same step in whitened coordinates ---------------------------------------------------------------------
// With L = chol(Q / step^2) the drift matrix of the step above is a multiple of the identity: -(L L')^{-1} Q = -step^2 I,
// m(x) = x - (step^2 / 2)(x - mu).  In a = L'(x - mu) the whole step is element-wise:
//   a' = kappa a + z,  kappa = 1 - step^2 / 2                      (proposal: x' = m(x) + L^{-T} z)
//   |L'(x - mu)|^2 = |a|^2,  |L'(x' - mu)|^2 = |a'|^2              (log p of both states)
//   |L'(x' - m(x))|^2 = |z|^2,  |L'(x - m(x'))|^2 = |a - kappa a'|^2   (log q forward and reverse)
// and only the accepted proposals have to come back to x = mu + L^{-T} a': ONE triangular product per step (plus one
// for a = L'(x - mu) whenever the caller's x is not the one this routine left behind), against the three full and one
// triangular product of omc_mala_step -- 2 d^2 instead of 9 d^2 flop per chain.  Same draws, same decision rule; the
// numbers differ from omc_mala_step's by the rounding of different (shorter) sums.
__global__ void __launch_bounds__(256) k_mala_white(int64_t d, int64_t chain_offset, omc_rng_key nkey, omc_rng_key ukey,
                                                    const double* zin, int64_t ld_z, const double* u_in, const double* sumlogL,
                                                    double log_step_term, double lp_scale, double kappa, double* a,
                                                    double* a_prop, int* accept_out, long long* acc_cnt, long long* prop_cnt,
                                                    double* logp_out) {
  __shared__ double red[4][4];
  __shared__ int accept;
  const int64_t c = blockIdx.x;
  const int64_t npairs = (d + 1) / 2;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int64_t q = threadIdx.x; q < npairs; q += 256) {
    double z[2];
    if (zin) {
      z[0] = zin[c * ld_z + 2 * q];
      z[1] = (2 * q + 1 < d) ? zin[c * ld_z + 2 * q + 1] : 0.0;
    } else {
      omc_normal_pair(omc_rng_block(nkey, chain_offset + c, (uint32_t)q), z[0], z[1]);  // k_draw_normals' mapping
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t i = 2 * q + e;
      if (i >= d) break;
      const double av = a[c * d + i];
      const double ap = fma(kappa, av, z[e]);
      const double rv = fma(-kappa, ap, av);
      a_prop[c * d + i] = ap;
      s0 = fma(av, av, s0);
      s1 = fma(ap, ap, s1);
      s2 = fma(rv, rv, s2);
      s3 = fma(z[e], z[e], s3);
    }
  }
  for (int s = 32; s >= 1; s >>= 1) {
    s0 += __shfl_xor(s0, s, 64); s1 += __shfl_xor(s1, s, 64);
    s2 += __shfl_xor(s2, s, 64); s3 += __shfl_xor(s3, s, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red[0][w] = s0; red[1][w] = s1; red[2][w] = s2; red[3][w] = s3;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // k_mh_finish's decision, term for term
    const double ss_cur = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double ss_prop = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double ss_rev = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    const double ss_fwd = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
    const double sl = sumlogL[0];
    const double logdetQ = 2.0 * (sl + log_step_term);
    const double dnum = (double)d;
    const double lp_cur = 0.5 * (logdetQ - dnum * 1.8378770664093453 - lp_scale * ss_cur);
    const double lp_prop = 0.5 * (logdetQ - dnum * 1.8378770664093453 - lp_scale * ss_prop);
    const double lq_fwd = sl - 0.5 * ss_fwd, lq_rev = sl - 0.5 * ss_rev;
    const double log_alpha = lp_prop + lq_rev - (lp_cur + lq_fwd);
    double u;
    if (u_in) {
      u = u_in[c];
    } else {
      const uint4 w = omc_rng_block(ukey, chain_offset + c, 0u);
      u = omc_u53(w.x, w.y);
    }
    const int ok = log(u) < log_alpha;
    accept = ok;
    if (logp_out) logp_out[c] = ok ? lp_prop : lp_cur;  // log target of the state this step leaves behind
    accept_out[c] = ok;
    if (prop_cnt) prop_cnt[c] += 1;
    if (acc_cnt && ok) acc_cnt[c] += 1;
  }
  __syncthreads();
  if (accept)
    for (int64_t i = threadIdx.x; i < d; i += 256) a[c * d + i] = a_prop[c * d + i];
}

static omc_status white_prepare(omc_ctx* ctx, int64_t d, const double* L, const double* mu) {
  if (ctx->white_L == L && ctx->white_mu == mu && ctx->white_d == d && ctx->white_prep) return OMC_OK;
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->white_prep, &ctx->white_prep_bytes, (size_t)(2 * d * d + 2 * d) * sizeof(double));
  if (st != OMC_OK) return st;
  rocblas_handle h = (rocblas_handle)ctx->blas;
  const rocblas_int di = (rocblas_int)d;
  double* LinvT = ctx->white_prep;          // L^{-T}, upper triangular
  double* Lt = ctx->white_prep + d * d;     // L', upper triangular
  double* negLtmu = ctx->white_prep + 2 * d * d;  // -L' mu
  double* negmu = negLtmu + d;
  hipLaunchKernelGGL(k_set_identity, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d, LinvT);
  const double one = 1.0;
  OMC_BLAS_CHECK(rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                               rocblas_diagonal_non_unit, di, di, &one, L, di, LinvT, di));
  hipLaunchKernelGGL(k_lower_transpose, dim3(gx(d * d)), dim3(256), 0, ctx->stream, d, L, Lt);
  if (mu) {
    hipLaunchKernelGGL(k_scale_copy, dim3(gx(d)), dim3(256), 0, ctx->stream, d, mu, -1.0, negmu);
    st = omc_dgemm_small(ctx, (int)d, 1, Lt, d, negmu, d, (int)d, nullptr, 0, nullptr, 0, 0, 1, nullptr, negLtmu, d);
    if (st != OMC_OK) return st;
  }
  OMC_HIP_CHECK(hipGetLastError());
  ctx->white_L = L; ctx->white_mu = mu; ctx->white_d = d; ctx->white_x = nullptr;
  return OMC_OK;
}

omc_status omc_mala_step_white(omc_ctx* ctx, int64_t d, const double* mu, const double* L, const double* sumlogL, double step,
                               const double* z_inject, int64_t ld_z, const double* u_inject, uint64_t draw_index, double* x,
                               int64_t ld_x, int32_t state_is_current, int64_t* accept_count, int64_t* proposal_count, double* log_p_out) {
  if (!ctx || d < 1 || d > 46340 || !L || !sumlogL || !x || ld_x < d || (z_inject && ld_z < d) || !(step > 0.0))
    return OMC_INVALID_ARG;
  const int64_t C = ctx->n_chains;
  if (C > 65535) return OMC_UNSUPPORTED;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  MhWork w{};
  st = mh_workspace(ctx, d, &w);
  if (st != OMC_OK) return st;
  st = white_prepare(ctx, d, L, mu);
  if (st != OMC_OK) return st;
  st = omc_ensure_bytes(ctx, (void**)&ctx->white_a, &ctx->white_a_bytes, (size_t)C * d * sizeof(double));
  if (st != OMC_OK) return st;
  const double* LinvT = ctx->white_prep;
  const double* Lt = ctx->white_prep + d * d;
  const double* negLtmu = mu ? ctx->white_prep + 2 * d * d : nullptr;
  double* a = ctx->white_a;
  if (!(state_is_current && ctx->white_x == x && ctx->white_ld == ld_x)) {  // a = L'(x - mu) = L'x - L'mu
    st = omc_dgemm_small(ctx, (int)d, (int)C, Lt, d, x, ld_x, (int)d, nullptr, 0, nullptr, 0, 0, 1, negLtmu, a, d);
    if (st != OMC_OK) return st;
  }
  hipLaunchKernelGGL(k_mala_white, dim3((unsigned)C), dim3(256), 0, ctx->stream, d, ctx->chain_offset,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM),
                     z_inject, ld_z, u_inject, sumlogL, (double)d * log(step), step * step, 1.0 - 0.5 * step * step, a, w.XP,
                     w.flag, (long long*)accept_count, (long long*)proposal_count, log_p_out);
  OMC_HIP_CHECK(hipGetLastError());
  // x = mu + L^{-T} a on the chains that accepted (the others keep their x bit for bit)
  st = omc_dgemm_small(ctx, (int)d, (int)C, LinvT, d, a, d, (int)d, nullptr, 0, nullptr, 0, 0, 1, mu, x, ld_x, w.flag);
  if (st != OMC_OK) return st;
  ctx->white_x = x; ctx->white_ld = ld_x;
  return OMC_OK;
}

// ---- several whitened steps per launch ----------------------------------------------------------------------------------
// The whitened step is element-wise in a = L'(x - mu) plus four sums and one decision per chain: nothing in it needs x.  K steps
// are therefore ONE launch that keeps a chain's a in registers (a workgroup per chain, a pair of elements per thread), leaves
// the whitened trajectory a_t behind, and ONE triangular product afterwards maps all K x C states back, x_t = mu + L^-T a_t,
// straight into the store slabs (mcmc.py:105-106 stores the state of every iteration) -- a d x d by d x (K C) product that
// fills the chip, where a product per step (d x C) occupies a quarter of it for 12 us.  Same draws (stream draw_index0 +
// t draw_stride for step t), same sums in the same order, same decision as K calls of omc_mala_step_white: bit-identical a
// and counters, and x of every stored step equal to what the single steps leave (same product, column by column).
}  // extern "C"  (a template)
template <int NP>
__global__ void __launch_bounds__(256) k_mala_white_run(int64_t d, int64_t C, int64_t chain_offset, uint64_t seed, uint64_t draw_index0,
                                                        uint64_t draw_stride, int n_steps, const double* __restrict__ zin, int64_t ld_z,
                                                        const double* __restrict__ u_in, const double* __restrict__ sumlogL,
                                                        double log_step_term, double lp_scale, double kappa, double* __restrict__ a,
                                                        double* __restrict__ a_traj, long long* __restrict__ acc_cnt,
                                                        long long* __restrict__ prop_cnt, double* __restrict__ logp_traj,
                                                        double* __restrict__ logp_last) {
  __shared__ double red[2][4][4];
  const int64_t c = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t npairs = (d + 1) / 2;
  double av[NP][2];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t i = 2 * ((int64_t)tid + 256 * p) + e;
      av[p][e] = i < d ? a[c * d + i] : 0.0;
    }
  const double sl = sumlogL[0];
  const double logdetQ = 2.0 * (sl + log_step_term);
  const double dnum = (double)d;
  long long n_acc = 0;
  double lp_state = 0.0;
  for (int t = 0; t < n_steps; ++t) {
    const omc_rng_key nkey = omc_make_key(seed, draw_index0 + (uint64_t)t * draw_stride, OMC_RNG_NORMAL);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    double ap[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int64_t q = (int64_t)tid + 256 * p;
      ap[p][0] = ap[p][1] = 0.0;
      if (q < npairs) {
        double z[2];
        if (zin) {
          const double* zr = zin + ((int64_t)t * C + c) * ld_z;
          z[0] = zr[2 * q];
          z[1] = (2 * q + 1 < d) ? zr[2 * q + 1] : 0.0;
        } else {
          omc_normal_pair(omc_rng_block(nkey, chain_offset + c, (uint32_t)q), z[0], z[1]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (2 * q + e >= d) break;
          const double v = av[p][e];
          const double pv = fma(kappa, v, z[e]);
          const double rv = fma(-kappa, pv, v);
          ap[p][e] = pv;
          s0 = fma(v, v, s0);
          s1 = fma(pv, pv, s1);
          s2 = fma(rv, rv, s2);
          s3 = fma(z[e], z[e], s3);
        }
      }
    }
    for (int s = 32; s >= 1; s >>= 1) {
      s0 += __shfl_xor(s0, s, 64); s1 += __shfl_xor(s1, s, 64);
      s2 += __shfl_xor(s2, s, 64); s3 += __shfl_xor(s3, s, 64);
    }
    // the step's uniform and its logarithm depend on nothing of the step: formed in front of the barrier, in the shadow of the
    // slowest wave's sums
    double u;
    if (u_in) {
      u = u_in[(int64_t)t * C + c];
    } else {
      const uint4 w = omc_rng_block(omc_make_key(seed, draw_index0 + (uint64_t)t * draw_stride, OMC_RNG_UNIFORM), chain_offset + c, 0u);
      u = omc_u53(w.x, w.y);
    }
    const double log_u = log(u);
    double(*rd)[4] = red[t & 1];  // two buffers: a wave that is a step ahead never writes what a slower one still reads
    if ((tid & 63) == 0) {
      const int w = tid >> 6;
      rd[0][w] = s0; rd[1][w] = s1; rd[2][w] = s2; rd[3][w] = s3;
    }
    __syncthreads();
    // every thread forms the decision (k_mala_white's, term for term): no second barrier, no broadcast
    const double ss_cur = (rd[0][0] + rd[0][1]) + (rd[0][2] + rd[0][3]);
    const double ss_prop = (rd[1][0] + rd[1][1]) + (rd[1][2] + rd[1][3]);
    const double ss_rev = (rd[2][0] + rd[2][1]) + (rd[2][2] + rd[2][3]);
    const double ss_fwd = (rd[3][0] + rd[3][1]) + (rd[3][2] + rd[3][3]);
    const double lp_cur = 0.5 * (logdetQ - dnum * 1.8378770664093453 - lp_scale * ss_cur);
    const double lp_prop = 0.5 * (logdetQ - dnum * 1.8378770664093453 - lp_scale * ss_prop);
    const double lq_fwd = sl - 0.5 * ss_fwd, lq_rev = sl - 0.5 * ss_rev;
    const double log_alpha = lp_prop + lq_rev - (lp_cur + lq_fwd);
    const bool ok = log_u < log_alpha;
    lp_state = ok ? lp_prop : lp_cur;
    n_acc += ok;
    if (ok) {
#pragma unroll
      for (int p = 0; p < NP; ++p) { av[p][0] = ap[p][0]; av[p][1] = ap[p][1]; }
    }
    if (a_traj) {
      double* row = a_traj + ((int64_t)t * C + c) * d;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int64_t i = 2 * ((int64_t)tid + 256 * p) + e;
          if (i < d) row[i] = av[p][e];
        }
    }
    if (tid == 0 && logp_traj) logp_traj[(int64_t)t * C + c] = lp_state;
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t i = 2 * ((int64_t)tid + 256 * p) + e;
      if (i < d) a[c * d + i] = av[p][e];
    }
  if (tid == 0) {
    if (prop_cnt) prop_cnt[c] += n_steps;
    if (acc_cnt) acc_cnt[c] += n_acc;
    if (logp_last && n_steps > 0) logp_last[c] = lp_state;
  }
}

extern "C" {

#define OMC_WHITE_RUN_BLOCK 32  // steps per launch (the whitened trajectory of a block: 32 x C x d doubles of workspace)

omc_status omc_mala_run_white(omc_ctx* ctx, int64_t d, const double* mu, const double* L, const double* sumlogL, double step,
                              const double* z_inject, int64_t ld_z, const double* u_inject, uint64_t draw_index0, uint64_t draw_stride,
                              int64_t n_steps, double* x, int64_t ld_x, int32_t state_is_current, double* x_store, double* logp_store,
                              int64_t* accept_count, int64_t* proposal_count, double* log_p_out) {
  if (!ctx || d < 1 || d > 2048 || !L || !sumlogL || !x || ld_x < d || (z_inject && ld_z < d) || !(step > 0.0) || n_steps < 0)
    return d > 2048 ? OMC_UNSUPPORTED : OMC_INVALID_ARG;
  const int64_t C = ctx->n_chains;
  if (C > 65535) return OMC_UNSUPPORTED;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  st = white_prepare(ctx, d, L, mu);
  if (st != OMC_OK) return st;
  st = omc_ensure_bytes(ctx, (void**)&ctx->white_a, &ctx->white_a_bytes, (size_t)C * d * sizeof(double));
  if (st != OMC_OK) return st;
  const int64_t KB = OMC_WHITE_RUN_BLOCK;
  if (x_store) {
    // two trajectory buffers: the product of block b runs on the side stream while the steps of block b + 1 run here
    st = omc_ensure_bytes(ctx, (void**)&ctx->white_traj, &ctx->white_traj_bytes, (size_t)2 * KB * C * d * sizeof(double));
    if (st != OMC_OK) return st;
    st = omc_ensure_aux(ctx);
    if (st != OMC_OK) return st;
    for (int i = 0; i < 4; ++i)
      if (!ctx->white_ev[i]) OMC_HIP_CHECK(hipEventCreateWithFlags(&ctx->white_ev[i], hipEventDisableTiming));
  }
  const double* LinvT = ctx->white_prep;
  const double* Lt = ctx->white_prep + d * d;
  const double* negLtmu = mu ? ctx->white_prep + 2 * d * d : nullptr;
  double* a = ctx->white_a;
  if (!(state_is_current && ctx->white_x == x && ctx->white_ld == ld_x)) {  // a = L'(x - mu) = L'x - L'mu
    st = omc_dgemm_small(ctx, (int)d, (int)C, Lt, d, x, ld_x, (int)d, nullptr, 0, nullptr, 0, 0, 1, negLtmu, a, d);
    if (st != OMC_OK) return st;
  }
  const int np = (int)(((d + 1) / 2 + 255) / 256);
  int64_t blk = 0;
  for (int64_t t0 = 0; t0 < n_steps; t0 += KB, ++blk) {
    const int nb = (int)((n_steps - t0 < KB) ? n_steps - t0 : KB);
    const int buf = (int)(blk & 1);
    double* traj = x_store ? ctx->white_traj + (size_t)buf * KB * C * d : nullptr;
    if (x_store && blk >= 2) OMC_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->white_ev[2 + buf], 0));  // block b - 2's product has read this buffer
#define OMC_WRUN(NP)                                                                                                                   \
  hipLaunchKernelGGL((k_mala_white_run<NP>), dim3((unsigned)C), dim3(256), 0, ctx->stream, d, C, ctx->chain_offset, ctx->seed,         \
                     draw_index0 + (uint64_t)t0 * draw_stride, draw_stride, nb, z_inject ? z_inject + t0 * C * ld_z : nullptr, ld_z,   \
                     u_inject ? u_inject + t0 * C : nullptr, sumlogL, (double)d * log(step), step * step, 1.0 - 0.5 * step * step, a,    \
                     traj, (long long*)accept_count, (long long*)proposal_count, logp_store ? logp_store + t0 * C : nullptr, log_p_out)
    if (np <= 1) OMC_WRUN(1); else if (np == 2) OMC_WRUN(2); else OMC_WRUN(4);
#undef OMC_WRUN
    OMC_HIP_CHECK(hipGetLastError());
    if (x_store) {  // x_t = mu + L^-T a_t for the whole block, into the store slabs [t][c][:] -- on the side stream
      OMC_HIP_CHECK(hipEventRecord(ctx->white_ev[buf], ctx->stream));
      OMC_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->white_ev[buf], 0));
      hipStream_t mine = ctx->stream;
      ctx->stream = ctx->aux_stream;  // (omc_dgemm_small launches on the context's stream)
      // (a few thousand columns and more: 64 x 64 tiles; below that the step's own small-tile kernel, bit for bit what the
      //  single steps compute)
      if (nb * C >= 4096 && !ctx->mh_use_rocblas)
        st = omc_dgemm_wide(ctx, (int)d, (int)(nb * C), LinvT, d, traj, d, (int)d, 1, mu, x_store + t0 * C * d, d);
      else
        st = omc_dgemm_small(ctx, (int)d, (int)(nb * C), LinvT, d, traj, d, (int)d, nullptr, 0, nullptr, 0, 0, 1, mu, x_store + t0 * C * d, d);
      ctx->stream = mine;
      if (st != OMC_OK) return st;
      OMC_HIP_CHECK(hipEventRecord(ctx->white_ev[2 + buf], ctx->aux_stream));
    }
  }
  if (x_store) {  // join: everything behind this call on the context's stream sees the whole store
    for (int i = 0; i < 2 && i < blk; ++i) OMC_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->white_ev[2 + ((blk - 1 - i) & 1)], 0));
  }
  if (n_steps > 0) {
    if (x_store) {  // the state the run leaves is its last stored one
      OMC_HIP_CHECK(hipMemcpy2DAsync(x, (size_t)ld_x * sizeof(double), x_store + (n_steps - 1) * C * d, (size_t)d * sizeof(double),
                                     (size_t)d * sizeof(double), (size_t)C, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
      st = omc_dgemm_small(ctx, (int)d, (int)C, LinvT, d, a, d, (int)d, nullptr, 0, nullptr, 0, 0, 1, mu, x, ld_x);
      if (st != OMC_OK) return st;
    }
  }
  ctx->white_x = x; ctx->white_ld = ld_x;
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
  MhWork w{};
  st = mh_workspace(ctx, d, &w);
  if (st != OMC_OK) return st;
  rocblas_handle h = (rocblas_handle)ctx->blas;
  const rocblas_int di = (rocblas_int)d, Ci = (rocblas_int)C;
  const double one = 1.0;
  const dim3 g2(gx(d) > 8 ? 8 : gx(d), (unsigned)C), b2(256);
  hipStream_t s = ctx->stream;
  double* Z = w.V;
  double* T2 = w.T;
  double* N2 = w.T + 2 * C * d;
  hipLaunchKernelGGL(k_draw_normals, dim3(gx((d + 1) / 2) > 8 ? 8 : gx((d + 1) / 2), (unsigned)C), b2, 0, s, d,
                     ctx->chain_offset, omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), z_inject, ld_z, Z, d);
  hipLaunchKernelGGL(k_rw_build, g2, b2, 0, s, d, C, x, ld_x, mu, step, Z, w.XP, T2);  // :250
  if (ctx->rw_LQ != LQ || ctx->rw_d != d || !ctx->rw_prep) {
    st = omc_ensure_bytes(ctx, (void**)&ctx->rw_prep, &ctx->rw_prep_bytes, (size_t)2 * d * d * sizeof(double));
    if (st != OMC_OK) return st;
    hipLaunchKernelGGL(k_lower_copy, dim3(gx(d * d)), dim3(256), 0, s, d, LQ, ctx->rw_prep);
    hipLaunchKernelGGL(k_lower_transpose, dim3(gx(d * d)), dim3(256), 0, s, d, LQ, ctx->rw_prep + d * d);
    ctx->rw_LQ = LQ; ctx->rw_d = d;
  }
  const double zero = 0.0;
  if (!ctx->mh_use_rocblas && d <= 46340) {
    st = omc_dgemm_small(ctx, (int)d, (int)(2 * C), ctx->rw_prep + d * d, d, T2, d, (int)d, nullptr, 0, nullptr, 0, 0, 1, nullptr, N2, d);
    if (st != OMC_OK) return st;
  } else {
    OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, di, 2 * Ci, di, &one, ctx->rw_prep, di,
                                 T2, di, &zero, N2, di));
  }
  hipLaunchKernelGGL(k_mh_finish, dim3((unsigned)C), dim3(256), 0, s, d, ctx->chain_offset,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), u_inject, sumlogLQ, 0.0, 1.0, N2, N2 + C * d,
                     (const double*)nullptr, (const double*)nullptr, w.XP, x, ld_x, (long long*)accept_count,
                     (long long*)proposal_count);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

// ---- the random-walk step with the target's quadratic form carried in whitened coordinates -----------------------
// a = L_Q'(x - mu) (L_Q = chol(Q)):  x' = x + step z  =>  a' = a + step L_Q' z, log p(x') - log p(x) = (|a|^2 - |a'|^2) / 2.
// One triangular product per step (L_Q' z) instead of the two of omc_rw_step (L_Q'[x - mu | x' - mu]); x' itself is the
// same two-rounding x + step z as there, so for equal decisions the states are bit-identical.
__global__ void __launch_bounds__(256) k_rw_white(int64_t d, int64_t chain_offset, omc_rng_key ukey, const double* u_in,
                                                  const double* sumlogL, double step, const double* z, const double* wz,
                                                  double* a, double* x, int64_t ld_x, long long* acc_cnt, long long* prop_cnt,
                                                  double* logp_out) {
  __shared__ double red[2][4];
  __shared__ int accept;
  const int64_t c = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int64_t i = threadIdx.x; i < d; i += 256) {
    const double av = a[c * d + i];
    const double ap = fma(step, wz[c * d + i], av);
    s0 = fma(av, av, s0);
    s1 = fma(ap, ap, s1);
  }
  for (int s = 32; s >= 1; s >>= 1) { s0 += __shfl_xor(s0, s, 64); s1 += __shfl_xor(s1, s, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {  // k_mh_finish's decision for the symmetric proposal
    const double ss_cur = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double ss_prop = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double logdetQ = 2.0 * sumlogL[0], dnum = (double)d;
    const double lp_cur = 0.5 * (logdetQ - dnum * 1.8378770664093453 - ss_cur);
    const double lp_prop = 0.5 * (logdetQ - dnum * 1.8378770664093453 - ss_prop);
    const double log_alpha = lp_prop - lp_cur;
    double u;
    if (u_in) {
      u = u_in[c];
    } else {
      const uint4 w = omc_rng_block(ukey, chain_offset + c, 0u);
      u = omc_u53(w.x, w.y);
    }
    const int ok = log(u) < log_alpha;
    accept = ok;
    if (logp_out) logp_out[c] = ok ? lp_prop : lp_cur;  // log target of the state this step leaves behind
    if (prop_cnt) prop_cnt[c] += 1;
    if (acc_cnt && ok) acc_cnt[c] += 1;
  }
  __syncthreads();
  if (accept)
    for (int64_t i = threadIdx.x; i < d; i += 256) {
      a[c * d + i] = fma(step, wz[c * d + i], a[c * d + i]);
      double pv;
      {
#pragma clang fp contract(off)
        const double prod = z[c * d + i] * step;
        pv = x[c * ld_x + i] + prod;
      }
      x[c * ld_x + i] = pv;
    }
}

omc_status omc_rw_step_white(omc_ctx* ctx, int64_t d, const double* mu, const double* LQ, const double* sumlogLQ, double step,
                             const double* z_inject, int64_t ld_z, const double* u_inject, uint64_t draw_index, double* x,
                             int64_t ld_x, int32_t state_is_current, int64_t* accept_count, int64_t* proposal_count, double* log_p_out) {
  if (!ctx || d < 1 || d > 46340 || !LQ || !sumlogLQ || !x || ld_x < d || (z_inject && ld_z < d) || !(step > 0.0))
    return OMC_INVALID_ARG;
  const int64_t C = ctx->n_chains;
  if (C > 65535) return OMC_UNSUPPORTED;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  MhWork w{};
  st = mh_workspace(ctx, d, &w);
  if (st != OMC_OK) return st;
  hipStream_t s = ctx->stream;
  bool fresh = false;
  if (ctx->rw_LQ != LQ || ctx->rw_d != d || !ctx->rw_prep) {
    st = omc_ensure_bytes(ctx, (void**)&ctx->rw_prep, &ctx->rw_prep_bytes, (size_t)2 * d * d * sizeof(double));
    if (st != OMC_OK) return st;
    hipLaunchKernelGGL(k_lower_copy, dim3(gx(d * d)), dim3(256), 0, s, d, LQ, ctx->rw_prep);
    hipLaunchKernelGGL(k_lower_transpose, dim3(gx(d * d)), dim3(256), 0, s, d, LQ, ctx->rw_prep + d * d);
    ctx->rw_LQ = LQ; ctx->rw_d = d;
    fresh = true;
  }
  const double* Lt = ctx->rw_prep + d * d;
  st = omc_ensure_bytes(ctx, (void**)&ctx->rww_mu_neg, &ctx->rww_mu_bytes, (size_t)2 * d * sizeof(double));
  if (st != OMC_OK) return st;
  double* negLtmu = nullptr;
  if (mu) {  // -L_Q' mu, once per (L_Q, mu)
    negLtmu = ctx->rww_mu_neg;
    if (fresh || ctx->rww_mu != mu) {
      hipLaunchKernelGGL(k_scale_copy, dim3(gx(d)), dim3(256), 0, s, d, mu, -1.0, ctx->rww_mu_neg + d);
      st = omc_dgemm_small(ctx, (int)d, 1, Lt, d, ctx->rww_mu_neg + d, d, (int)d, nullptr, 0, nullptr, 0, 0, 1, nullptr, negLtmu, d);
      if (st != OMC_OK) return st;
      fresh = true;
    }
  }
  if (ctx->rww_mu != mu) fresh = true;
  ctx->rww_mu = mu;
  st = omc_ensure_bytes(ctx, (void**)&ctx->rww_a, &ctx->rww_a_bytes, (size_t)C * d * sizeof(double));
  if (st != OMC_OK) return st;
  double* a = ctx->rww_a;
  if (fresh || !(state_is_current && ctx->rww_x == x && ctx->rww_ld == ld_x)) {  // a = L_Q'(x - mu)
    st = omc_dgemm_small(ctx, (int)d, (int)C, Lt, d, x, ld_x, (int)d, nullptr, 0, nullptr, 0, 0, 1, negLtmu, a, d);
    if (st != OMC_OK) return st;
  }
  double* Z = w.V;
  double* WZ = w.XP;
  hipLaunchKernelGGL(k_draw_normals, dim3(gx((d + 1) / 2) > 8 ? 8 : gx((d + 1) / 2), (unsigned)C), dim3(256), 0, s, d,
                     ctx->chain_offset, omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), z_inject, ld_z, Z, d);
  st = omc_dgemm_small(ctx, (int)d, (int)C, Lt, d, Z, d, (int)d, nullptr, 0, nullptr, 0, 0, 1, nullptr, WZ, d);
  if (st != OMC_OK) return st;
  hipLaunchKernelGGL(k_rw_white, dim3((unsigned)C), dim3(256), 0, s, d, ctx->chain_offset,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), u_inject, sumlogLQ, step, Z, WZ, a, x, ld_x,
                     (long long*)accept_count, (long long*)proposal_count, log_p_out);
  OMC_HIP_CHECK(hipGetLastError());
  ctx->rww_x = x; ctx->rww_ld = ld_x;
  return OMC_OK;
}

}  // extern "C"
