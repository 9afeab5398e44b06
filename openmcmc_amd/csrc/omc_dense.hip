// Dense Normal-Normal path (regression-shaped conditionals): batched assembly + rocSOLVER batched
// Cholesky + rocBLAS batched triangular solves, and the design-matrix GEMMs (fp64 MFMA through
// rocBLAS).  Reference call sites: sampler.py:176-197, location_scale.py:234-242, gmrf.py:434,462,481.
#include <math.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "omc_common.h"

struct DenseTermsDev {
  int n_terms;
  const double* mat[OMC_MAX_TERMS];
  const double* rhs[OMC_MAX_TERMS];
  const double* scale[OMC_MAX_TERMS];
  const double* diag_chain;  // [C][p] per-chain diagonal, or NULL
};

#define OMC_BLAS_CHECK(expr)                                   \
  do {                                                         \
    rocblas_status _s = (expr);                                \
    if (_s != rocblas_status_success) {                        \
      omc_set_error(#expr, hipErrorUnknown);                   \
      return OMC_HIP_ERROR;                                    \
    }                                                          \
  } while (0)

omc_status omc_ensure_blas(omc_ctx* ctx) {
  if (!ctx->blas) {
    rocblas_handle h;
    OMC_BLAS_CHECK(rocblas_create_handle(&h));
    OMC_BLAS_CHECK(rocblas_set_stream(h, ctx->stream));
    OMC_BLAS_CHECK(rocblas_set_pointer_mode(h, rocblas_pointer_mode_host));
    ctx->blas = (void*)h;
  }
  return OMC_OK;
}

// side stream, its BLAS handle and the fork / join events of the blocked factorisation (made on first use)
omc_status omc_ensure_aux(omc_ctx* ctx) {
  if (ctx->blas_aux) return OMC_OK;
  OMC_HIP_CHECK(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
  OMC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  OMC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  rocblas_handle h;
  OMC_BLAS_CHECK(rocblas_create_handle(&h));
  OMC_BLAS_CHECK(rocblas_set_stream(h, ctx->aux_stream));
  OMC_BLAS_CHECK(rocblas_set_pointer_mode(h, rocblas_pointer_mode_host));
  ctx->blas_aux = (void*)h;
  return OMC_OK;
}

omc_status omc_ensure_bytes(omc_ctx* ctx, void** buf, size_t* have, size_t need) {
  if (*have >= need) return OMC_OK;
  OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (*buf) OMC_HIP_CHECK(hipFree(*buf));
  *buf = nullptr;
  *have = 0;
  OMC_HIP_CHECK(hipMalloc(buf, need));
  *have = need;
  return OMC_OK;
}

// Q[c] = sum_k s_k[c] M_k, lower triangle only (column-major: element (row, col) at col*p + row, row >= col): the
// factorisations read nothing else, and the upper half would be 1 GB of writes per sweep at p = 1000, C = 256
__global__ void __launch_bounds__(256) k_dense_assemble(DenseTermsDev T, int64_t p, int64_t C, double* Q) {
  const int64_t c = blockIdx.y;
  double sc[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) sc[k] = (k < T.n_terms && T.scale[k]) ? T.scale[k][c] : 1.0;
  double* q = Q + c * p * p;
  const int64_t total = p * p;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / p, cl = i - r * p;
    if (cl < r) continue;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < OMC_MAX_TERMS; ++k) {
      if (k >= T.n_terms) continue;
      v = fma(sc[k], T.mat[k] ? T.mat[k][i] : (r == cl ? 1.0 : 0.0), v);
    }
    if (T.diag_chain && r == cl) v += T.diag_chain[c * p + r];
    q[i] = v;
  }
}

// b[c] = sum_k s_k[c] rhs_k + rhs_chain[c]
__global__ void k_dense_rhs(DenseTermsDev T, int64_t p, int64_t C, const double* rhs_chain, int64_t ld_rhs,
                            double* b, int64_t ld_b) {
  const int64_t c = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p; i += (int64_t)gridDim.x * blockDim.x) {
    double v = rhs_chain ? rhs_chain[c * ld_rhs + i] : 0.0;
#pragma unroll
    for (int k = 0; k < OMC_MAX_TERMS; ++k) {
      if (k >= T.n_terms || !T.rhs[k]) continue;
      v = fma(T.scale[k] ? T.scale[k][c] : 1.0, T.rhs[k][i], v);
    }
    b[c * ld_b + i] = v;
  }
}

// after potrf: latch failures, log det = 2 sum log L_ii
__global__ void __launch_bounds__(256) k_dense_post_factor(int64_t p, int64_t C, const double* L, const int* info,
                                                          double* logdet, long long* bad) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  if (threadIdx.x == 0 && info[c] != 0) atomicMin((unsigned long long*)bad, (unsigned long long)c);
  if (!logdet) return;
  const double* l = L + c * p * p;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < p; i += blockDim.x) acc += log(l[i * p + i]);
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) logdet[c] = 2.0 * (red[0] + red[1] + red[2] + red[3]);
}

// t[c] = w[c] + z[c]   (z injected or from the chain's Philox stream)
__global__ void k_add_draw(int64_t p, int64_t C, int64_t chain_offset, omc_rng_key key, const double* z,
                           int64_t ld_z, double* t, int64_t ld_t) {
  const int64_t c = blockIdx.y;
  const int64_t npairs = (p + 1) / 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * blockDim.x) {
    double z0, z1;
    if (z) {
      z0 = z[c * ld_z + 2 * q];
      z1 = (2 * q + 1 < p) ? z[c * ld_z + 2 * q + 1] : 0.0;
    } else {
      omc_normal_pair(omc_rng_block(key, chain_offset + c, (uint32_t)q), z0, z1);
    }
    t[c * ld_t + 2 * q] += z0;
    if (2 * q + 1 < p) t[c * ld_t + 2 * q + 1] += z1;
  }
}

// Spectral route: in the eigenbasis of the one shared matrix, Q_c = diag(d_c), d_c[i] = a_c + b_c ev[i].
//   t holds V' rhs_c on entry; on exit m_c = t / d (into `mean` if given) and y_c = m_c + z / sqrt(d) (into t);
//   logdet_c = sum log d; a non-positive d latches the chain like a failed factorisation.
__global__ void __launch_bounds__(256) k_spectral_scale(DenseTermsDev T, int kmat, int64_t p, int64_t C, const double* ev,
                                                        int64_t chain_offset, omc_rng_key key, const double* z, int64_t ld_z,
                                                        double* t, int64_t ld_t, double* mean, int64_t ld_m, double* logdet,
                                                        long long* bad) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    if (k >= T.n_terms) continue;
    const double sk = T.scale[k] ? T.scale[k][c] : 1.0;
    if (k == kmat) b = sk;
    else a += sk;
  }
  double acc = 0.0;
  bool neg = false;
  const int64_t npairs = (p + 1) / 2;
  for (int64_t q = threadIdx.x; q < npairs; q += blockDim.x) {
    double z0, z1;
    if (z) {
      z0 = z[c * ld_z + 2 * q];
      z1 = (2 * q + 1 < p) ? z[c * ld_z + 2 * q + 1] : 0.0;
    } else {
      omc_normal_pair(omc_rng_block(key, chain_offset + c, (uint32_t)q), z0, z1);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t i = 2 * q + h;
      if (i >= p) continue;
      const double d = fma(b, ev[i], a);
      neg |= !(d > 0.0);
      const double r = 1.0 / d, m = t[c * ld_t + i] * r;
      if (mean) mean[c * ld_m + i] = m;
      t[c * ld_t + i] = fma(h ? z1 : z0, sqrt(r), m);
      acc += log(d);
    }
  }
  if (neg) atomicMin((unsigned long long*)bad, (unsigned long long)c);
  if (!logdet) return;
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) logdet[c] = red[0] + red[1] + red[2] + red[3];
}

__global__ void k_copy_rows(int64_t p, const double* src, int64_t ld_s, double* dst, int64_t ld_d) {
  const int64_t c = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p; i += (int64_t)gridDim.x * blockDim.x)
    dst[c * ld_d + i] = src[c * ld_s + i];
}

// Xw[i][j] = w[i] * X[i][j]
__global__ void k_scale_rows(int64_t n, int64_t p, const double* X, const double* w, double* out) {
  const int64_t total = n * p;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = X[i] * w[i / p];
}

__global__ void __launch_bounds__(256) k_weighted_resid_sq(int64_t n, const double* y, const double* f, int64_t ld_f,
                                                          const double* w, double* out) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  const double* fc = f + c * ld_f;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const double r = y[i] - fc[i];
    acc = fma((w ? w[i] : 1.0) * r, r, acc);
  }
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------
// Batched triangular solve, one right-hand side per chain:  L w = b  (TRANS = false)  or  L' x = t
// (TRANS = true), L lower, column-major, leading dimension p.  rocBLAS' trsv_strided_batched runs one
// small workgroup per matrix and took 20 ms per call at p = 1000, C = 256 (72 % of the regression
// sweep); here one 256-thread workgroup per chain keeps the right-hand side in LDS and walks L in
// 64-column blocks: a 64 x 64 diagonal solve by one wave (block staged in LDS), then the update of
// the remaining entries by all four waves, streaming L exactly once (p^2/2 doubles) with
// line-complete accesses.  HBM-bound: 4 MB per chain per solve at p = 1000.
#define OMC_TS_B 64
template <bool TRANS>
__global__ void __launch_bounds__(256) k_tri_solve_batched(int64_t p, const double* __restrict__ Lall, int64_t strideL,
                                                           double* __restrict__ xall, int64_t ld_x) {
  extern __shared__ double sm[];
  double* xs = sm;                     // [p]   right-hand side / solution
  double* blk = sm + ((p + 15) & ~15); // [64][65] diagonal block
  const int64_t c = blockIdx.x;
  const double* L = Lall + c * strideL;
  double* x = xall + c * ld_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t i = tid; i < p; i += 256) xs[i] = x[i];
  __syncthreads();
  const int64_t nb = (p + OMC_TS_B - 1) / OMC_TS_B;
  for (int64_t step = 0; step < nb; ++step) {
    const int64_t kb = TRANS ? nb - 1 - step : step;
    const int64_t k0 = kb * OMC_TS_B;
    const int bs = (int)((p - k0) < OMC_TS_B ? (p - k0) : OMC_TS_B);
    if (wave == 0) {
      // stage the diagonal block: lane = row inside the block (rows are contiguous in memory)
      if (bs == OMC_TS_B) {  // all 64 loads on the way before the first is stored (a rolled loop waits for each in turn)
        double v[OMC_TS_B];
#pragma unroll
        for (int j = 0; j < OMC_TS_B; ++j) v[j] = (j <= lane) ? L[(k0 + lane) + (k0 + j) * p] : 0.0;
#pragma unroll
        for (int j = 0; j < OMC_TS_B; ++j) blk[lane * (OMC_TS_B + 1) + j] = v[j];
      } else {
        for (int j = 0; j < bs; ++j)
          blk[lane * (OMC_TS_B + 1) + j] = (lane < bs && j <= lane) ? L[(k0 + lane) + (k0 + j) * p] : 0.0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double xi = (lane < bs) ? xs[k0 + lane] : 0.0;
      const double dinv = (lane < bs) ? 1.0 / blk[lane * (OMC_TS_B + 1) + lane] : 0.0;
      if (!TRANS) {
        for (int j = 0; j < bs; ++j) {
          const double xj = __shfl(xi * dinv, j, 64);          // lane j is final at step j
          if (lane == j) xi = xj;
          if (lane > j && lane < bs) xi = fma(-blk[lane * (OMC_TS_B + 1) + j], xj, xi);
        }
      } else {
        for (int i = bs - 1; i >= 0; --i) {
          const double xv = __shfl(xi * dinv, i, 64);          // lane i is final at step i
          if (lane == i) xi = xv;
          if (lane < i) xi = fma(-blk[i * (OMC_TS_B + 1) + lane], xv, xi);   // L[i][lane] = (L')[lane][i]
        }
      }
      if (lane < bs) xs[k0 + lane] = xi;
    }
    __syncthreads();
    if (!TRANS) {
      // x_i -= sum_j L[i, k0+j] x_{k0+j} for the rows below the block (thread per row, coalesced)
      // (a full block: all 64 loads of a row in flight at once -- with eight, a workgroup keeps 16 KB on the way, a third of
      // what its share of the memory bandwidth needs at 1.5 us latency)
      for (int64_t i = k0 + bs + tid; i < p; i += 256) {
        const double* lp = L + i + k0 * p;
        double acc = 0.0;
        if (bs == OMC_TS_B) {
          double v[OMC_TS_B];
#pragma unroll
          for (int j = 0; j < OMC_TS_B; ++j) v[j] = lp[(int64_t)j * p];
#pragma unroll
          for (int j = 0; j < OMC_TS_B; ++j) acc = fma(v[j], xs[k0 + j], acc);
        } else {
#pragma unroll 8
          for (int j = 0; j < bs; ++j) acc = fma(lp[(int64_t)j * p], xs[k0 + j], acc);
        }
        xs[i] -= acc;
      }
    } else {
      // t_j -= sum_i L[k0+i, j] x_{k0+i} for the columns left of the block (thread per column,
      // each thread streams bs contiguous doubles)
      for (int64_t j = tid; j < k0; j += 256) {
        const double* lp = L + k0 + j * p;
        double acc = 0.0;
        if (bs == OMC_TS_B) {
          double v[OMC_TS_B];
#pragma unroll
          for (int i = 0; i < OMC_TS_B; ++i) v[i] = lp[i];
#pragma unroll
          for (int i = 0; i < OMC_TS_B; ++i) acc = fma(v[i], xs[k0 + i], acc);
        } else {
#pragma unroll 8
          for (int i = 0; i < bs; ++i) acc = fma(lp[i], xs[k0 + i], acc);
        }
        xs[j] -= acc;
      }
    }
    __syncthreads();
  }
  for (int64_t i = tid; i < p; i += 256) x[i] = xs[i];
}

// ------------------------------------------------------------------------------------------
// Batched Cholesky, left-looking by 64-column blocks (column-major, lower; the natural-order factor is unique, so
// this is the reference's np.linalg.cholesky / LAPACK potrf result up to rounding, gmrf.py:481):
//   for each block column J:   A[J:, J] -= L[J:, :J] L[J, :J]'        one strided-batched DGEMM over all chains
//                              L_JJ = chol(A_JJ), A[J+1:, J] L_JJ^-T  k_chol_panel, one workgroup per chain
// rocSOLVER's potrf_strided_batched spent 13 ms on 256 matrices of order 1000 (6.5 TFLOP/s: small-panel kernels
// and a syr2k-based update); here the flops sit in 15 well-shaped batched GEMMs.
#define CH_NB 64
__device__ __forceinline__ void lds_barrier() {  // workgroup barrier that waits for LDS traffic only
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ void __launch_bounds__(256) k_chol_panel(int64_t p, int64_t j0, int nb, double* Qall, int* info,
                                                    long long* bad, int64_t chain0, unsigned long long* dbg) {
#define PANEL_STAMP(i) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[(j0 / CH_NB) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
  PANEL_STAMP(0);
  __shared__ double D[CH_NB][CH_NB + 1];  // diagonal block, then its Cholesky factor (zero outside the live nb x nb)
  __shared__ double LiT[CH_NB][CH_NB];    // transposed inverse of the factor (row t: column t of L_JJ^-1)
  __shared__ double dinv[CH_NB], dsq[CH_NB];
  __shared__ int failed;
  const int64_t c = blockIdx.x;
  double* A = Qall + c * p * p + j0 + j0 * p;  // panel origin: element (r, cc) at A[r + cc * p]
  const int tid = threadIdx.x;
  const int64_t m = p - j0;
  for (int t = tid; t < CH_NB * CH_NB; t += 256) {
    const int r = t % CH_NB, cc = t / CH_NB;
    D[r][cc] = (cc <= r && r < nb) ? A[r + (int64_t)cc * p] : 0.0;
  }
  if (tid < CH_NB) dinv[tid] = 0.0;
  if (tid == 0) failed = 0;
  __syncthreads();
  PANEL_STAMP(1);
  // unblocked right-looking Cholesky of the block: all 256 threads tile the (r, cc) update square 16 x 16, two
  // LDS-only barriers per column (lds_barrier: no wait on vector memory)
  {
    const int tx = tid & 15, ty = tid >> 4;
    for (int k = 0; k < nb; ++k) {
      const double piv = D[k][k];  // final since the update of step k - 1; nobody writes it during this step
      const bool ok = piv > 0.0;
      const double sq = ok ? sqrt(piv) : 1.0;
      const double rinv = 1.0 / sq;
      if (tid > k && tid < nb) D[tid][k] *= rinv;
      if (tid == k) {
        dsq[k] = sq;   // the diagonal goes to its own array: D[k][k] is still being read by slower waves
        dinv[k] = rinv;
        if (!ok) failed = 1;
      }
      lds_barrier();
      for (int r = k + 1 + ty; r < nb; r += 16) {
        const double lr = D[r][k];
        for (int cc = k + 1 + tx; cc <= r; cc += 16) D[r][cc] = fma(-lr, D[cc][k], D[r][cc]);
      }
      lds_barrier();
    }
  }
  PANEL_STAMP(2);
  // inverse of the diagonal factor, one column per group of four adjacent lanes (forward substitution on e_j, the
  // inner sum split four ways and combined with two shuffles), stored transposed:
  // LiT[t][cc] = (L_JJ^-1)[cc][t], so that the row update below reads 64 consecutive doubles per t
  {
    const int j = tid >> 2, part = tid & 3;
    if (part == 0)
      for (int i = 0; i < CH_NB; ++i) LiT[j][i] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = j; i < nb; ++i) {   // j is uniform within a group of four lanes; i runs in lock-step for the group
      double acc = 0.0;
      if (j < nb)
        for (int t = j + part; t < i; t += 4) acc = fma(-D[i][t], LiT[j][t], acc);
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (part == 0 && j < nb) LiT[j][i] = (acc + ((i == j) ? 1.0 : 0.0)) * dinv[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  PANEL_STAMP(3);
  for (int t = tid; t < nb * nb; t += 256) {
    const int r = t % nb, cc = t / nb;
    if (cc <= r) A[r + (int64_t)cc * p] = (cc == r) ? dsq[r] : D[r][cc];
  }
  PANEL_STAMP(4);
  if (tid == 0 && failed) {
    info[c] = (int)j0 + 1;
    atomicMin((unsigned long long*)bad, (unsigned long long)(chain0 + c));
  }
  // rows below the block:  X = A_panel L_JJ^-T, i.e. X[r][cc] = sum_{t <= cc} A[r][t] Linv[cc][t].  One thread per
  // row: 64 accumulators in registers (compile-time indices); the row is fetched eight entries at a time, the next
  // eight while the current ones are used (each fetch is coalesced over the rows); the inverse is read from LDS as
  // wave-wide broadcasts, and the zero half of the triangle is skipped per chunk.
  for (int64_t r = nb + tid; r < m; r += 256) {
    double acc[CH_NB];
#pragma unroll
    for (int cc = 0; cc < CH_NB; ++cc) acc[cc] = 0.0;
    double a_cur[8], a_next[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a_cur[q] = (q < nb) ? A[r + (int64_t)q * p] : 0.0;
#pragma unroll
    for (int t0 = 0; t0 < CH_NB; t0 += 8) {
      if (t0 < nb) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a_next[q] = (t0 + 8 + q < nb) ? A[r + (int64_t)(t0 + 8 + q) * p] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
          for (int cc = t0; cc < CH_NB; ++cc) acc[cc] = fma(a_cur[q], LiT[t0 + q][cc], acc[cc]);
          __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads of later rows of the inverse from piling up
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) a_cur[q] = a_next[q];
      }
    }
#pragma unroll
    for (int cc = 0; cc < CH_NB; ++cc)
      if (cc < nb) A[r + (int64_t)cc * p] = acc[cc];
  }
  PANEL_STAMP(5);
#undef PANEL_STAMP
}

// ---- the panel in two kernels (round 4) ---------------------------------------------------------------------------------------
// Stamps of k_chol_panel at p = 1000 (benchmarks/micro/panel_stamps.py): 44-80 us in the 64 x 64 diagonal factor (two workgroup
// barriers per column), 31 us in its inverse, 20-90 us in the rows below (one thread per row, 64 accumulators, an LDS broadcast
// per multiply-add) -- 168 us per panel, sixteen panels in a row per draw, and the update GEMMs wait for every one of them.
//   k_chol_diag   one WAVE per chain: lane r keeps row r of the block in registers; column k is scaled and the trailing
//                 columns updated with the pivot column's entries broadcast by v_readlane -- no LDS, no barrier on the
//                 column-to-column path; then L^-1 by forward substitution (lane c: column c), L read as LDS broadcasts.
//                 Same operations in the same order per entry as the old factor: bit-identical L.
//   k_panel_rows  X = A_panel L^-T on the matrix cores, in place: a workgroup takes 64 rows x 64 columns, reads its tile of A
//                 whole (two contraction slabs) before it writes; batched over the chains in the grid's z.
typedef double chol_d4 __attribute__((ext_vector_type(4)));
// (omc_choldiag.hip: a translation unit of its own -- 64 fully unrolled column steps take minutes to compile and depend on no
//  header of the library)
void omc_launch_chol_diag(hipStream_t stream, int64_t Cn, int64_t p, int64_t j0, int nb, double* Qall, double* Winv_all, int* info,
                          long long* bad, int64_t chain0);

// rows j0 + nb .. p - 1 of the block column: X[r][cc] = sum_t A[r][t] W[cc][t], in place.  grid (groups of PR_RT row tiles, 1, chains).
// A workgroup walks PR_RT tiles of 64 rows with W staged once; the next tile's 64 x 64 entries of A are in flight (registers) while
// the current one is multiplied, and a tile is written only after every thread has read it: in place is safe (other workgroups
// own other rows).  Traffic-bound by design: 2 x 32 KB per 0.5 Mflop tile.
#define PR_RT 4
__global__ void __launch_bounds__(256) k_panel_rows(int64_t p, int64_t j0, int nb, double* Qall, const double* __restrict__ Winv_all) {
  __shared__ double As[CH_NB][CH_NB + 16];  // As[k][i]: row stride 80: the two rows of a half-wave in different bank halves
  __shared__ double Bs[CH_NB][CH_NB + 2];   // Bs[j][k] = W[j][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t c = blockIdx.z;
  const int64_t m = p - j0 - nb;            // rows below the diagonal block
  double* A = Qall + c * p * p + (j0 + nb) + j0 * p;  // element (r, t) at A[r + t * p]
  const double* W = Winv_all + c * (int64_t)(CH_NB * CH_NB);
  const int64_t tile0 = (int64_t)blockIdx.x * PR_RT;
  const int64_t ntile = (m + 63) / 64;
  double ra[16];
  auto fetch = [&](int64_t tile) {
    const int64_t i0 = tile * 64;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = q * 256 + tid, k = e >> 6, i = e & 63;   // contiguous in i
      ra[q] = (tile < ntile && k < nb && i0 + i < m) ? A[(i0 + i) + (int64_t)k * p] : 0.0;
    }
  };
  fetch(tile0);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = q * 256 + tid;
    Bs[e >> 6][e & 63] = W[e];                               // W stored [j][k] with k contiguous
  }
  for (int t = 0; t < PR_RT && tile0 + t < ntile; ++t) {
    const int64_t i0 = (tile0 + t) * 64;
    __syncthreads();                                          // the previous tile's reads of As are done
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = q * 256 + tid;
      As[e >> 6][e & 63] = ra[q];
    }
    __syncthreads();
    if (t + 1 < PR_RT) fetch(tile0 + t + 1);                  // in flight under the multiplications
    chol_d4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = chol_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < CH_NB; kk += 4) {
      const int kr = kk + (lane >> 4), cl = lane & 15;
      const double af = As[kr][wave * 16 + cl];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, Bs[16 * u + cl][kr], acc[u], 0, 0, 0);
    }
    // C/D of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * register
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = 16 * u + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + wave * 16 + (lane >> 4) + 4 * r;
        if (i < m && j < nb) A[i + (int64_t)j * p] = acc[u][r];
      }
    }
  }
}

// block columns of the chains [c0, c0 + Cn) on one stream
static omc_status potrf_blocked_part(omc_ctx* ctx, rocblas_handle h, hipStream_t stream, int64_t p, double* Q, int64_t c0, int64_t Cn) {
  const double minus_one = -1.0, one = 1.0;
  double* Qp = Q + c0 * p * p;
  for (int64_t j0 = 0; j0 < p; j0 += CH_NB) {
    const int nb = (int)((p - j0 < CH_NB) ? p - j0 : CH_NB);
    if (j0 > 0)
      OMC_BLAS_CHECK(rocblas_dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_transpose,
                                                   (rocblas_int)(p - j0), (rocblas_int)nb, (rocblas_int)j0, &minus_one,
                                                   Qp + j0, (rocblas_int)p, (rocblas_stride)(p * p), Qp + j0, (rocblas_int)p,
                                                   (rocblas_stride)(p * p), &one, Qp + j0 + j0 * p, (rocblas_int)p,
                                                   (rocblas_stride)(p * p), (rocblas_int)Cn));
    if (ctx->dense_panel_old) {
      hipLaunchKernelGGL(k_chol_panel, dim3((unsigned)Cn), dim3(256), 0, stream, p, j0, nb, Qp, ctx->dense_info + c0,
                         ctx->d_bad_chain, c0, c0 == 0 ? ctx->stamps : nullptr);
    } else {
      double* Winv = ctx->dense_winv + c0 * (int64_t)(CH_NB * CH_NB);
      omc_launch_chol_diag(stream, Cn, p, j0, nb, Qp, Winv, ctx->dense_info + c0, ctx->d_bad_chain, c0);
      const int64_t m = p - j0 - nb;
      if (m > 0)
        hipLaunchKernelGGL(k_panel_rows, dim3((unsigned)((m + 64 * PR_RT - 1) / (64 * PR_RT)), 1, (unsigned)Cn), dim3(256), 0, stream, p, j0, nb, Qp, Winv);
    }
    OMC_HIP_CHECK(hipGetLastError());
  }
  return OMC_OK;
}

static omc_status potrf_blocked(omc_ctx* ctx, rocblas_handle h, int64_t p, double* Q, int64_t C) {
  OMC_HIP_CHECK(hipMemsetAsync(ctx->dense_info, 0, (size_t)C * sizeof(int), ctx->stream));
  {
    const omc_status st0 = omc_ensure_bytes(ctx, (void**)&ctx->dense_winv, &ctx->dense_winv_bytes, (size_t)C * CH_NB * CH_NB * sizeof(double));
    if (st0 != OMC_OK) return st0;
  }
  // The panel kernel is one latency-bound workgroup per chain and the update GEMM cannot start before it: in one batch the
  // two alternate and each leaves most of the chip idle in turn.  Two halves of the chains on two streams (fork and join by
  // events: the caller still sees one stream) let one half's panels run under the other half's GEMMs.
  if (!ctx->dense_overlap || C < 64) return potrf_blocked_part(ctx, h, ctx->stream, p, Q, 0, C);
  omc_status st = omc_ensure_aux(ctx);
  if (st != OMC_OK) return st;
  const int64_t C0 = C / 2;
  OMC_HIP_CHECK(hipEventRecord(ctx->ev_fork, ctx->stream));
  OMC_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
  st = potrf_blocked_part(ctx, (rocblas_handle)ctx->blas_aux, ctx->aux_stream, p, Q, C0, C - C0);
  if (st != OMC_OK) return st;
  st = potrf_blocked_part(ctx, h, ctx->stream, p, Q, 0, C0);
  if (st != OMC_OK) return st;
  OMC_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->aux_stream));
  OMC_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return OMC_OK;
}

static omc_status tri_solve(omc_ctx* ctx, rocblas_handle h, bool trans, int64_t p, const double* Q, double* x, int64_t ld,
                            int64_t C) {
  if (p <= 8192) {
    const size_t lds = (((size_t)p + 15) & ~(size_t)15) * sizeof(double) + OMC_TS_B * (OMC_TS_B + 1) * sizeof(double);
    if (trans)
      hipLaunchKernelGGL(k_tri_solve_batched<true>, dim3((unsigned)C), dim3(256), lds, ctx->stream, p, Q, p * p, x, ld);
    else
      hipLaunchKernelGGL(k_tri_solve_batched<false>, dim3((unsigned)C), dim3(256), lds, ctx->stream, p, Q, p * p, x, ld);
    OMC_HIP_CHECK(hipGetLastError());
    return OMC_OK;
  }
  OMC_BLAS_CHECK(rocblas_dtrsv_strided_batched(h, rocblas_fill_lower,
                                               trans ? rocblas_operation_transpose : rocblas_operation_none,
                                               rocblas_diagonal_non_unit, (rocblas_int)p, Q, (rocblas_int)p,
                                               (rocblas_stride)(p * p), x, 1, (rocblas_stride)ld, (rocblas_int)C));
  return OMC_OK;
}

static unsigned gx(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

void omc_dense_release(omc_ctx* ctx) {
  if (ctx->blas) rocblas_destroy_handle((rocblas_handle)ctx->blas);
  ctx->blas = nullptr;
  if (ctx->blas_aux) {
    rocblas_destroy_handle((rocblas_handle)ctx->blas_aux);
    hipEventDestroy(ctx->ev_fork);
    hipEventDestroy(ctx->ev_join);
    hipStreamDestroy(ctx->aux_stream);
    for (int i = 0; i < 4; ++i)
      if (ctx->white_ev[i]) { hipEventDestroy(ctx->white_ev[i]); ctx->white_ev[i] = nullptr; }
    ctx->blas_aux = nullptr;
  }
}

extern "C" {

omc_status omc_dense_sample_canonical(omc_ctx* ctx, int64_t p, const omc_dense_terms* terms, const double* rhs_chain,
                                      int64_t ld_rhs, const double* z_inject, int64_t ld_z, uint64_t draw_index,
                                      double* x_out, int64_t ld_x, double* mean_out, int64_t ld_mean,
                                      double* logdet_out) {
  if (!ctx || p < 1 || p > 32768 || !terms || terms->n_terms < 1 || terms->n_terms > OMC_MAX_TERMS) return OMC_INVALID_ARG;
  if (!x_out || ld_x < p || (rhs_chain && ld_rhs < p) || (z_inject && ld_z < p) || (mean_out && ld_mean < p))
    return OMC_INVALID_ARG;
  const int64_t C = ctx->n_chains;
  if (C > 65535) return OMC_UNSUPPORTED;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  st = omc_ensure_bytes(ctx, (void**)&ctx->dense_factor, &ctx->dense_factor_bytes, (size_t)C * p * p * sizeof(double));
  if (st != OMC_OK) return st;
  st = omc_ensure_bytes(ctx, (void**)&ctx->dense_info, &ctx->dense_info_bytes, (size_t)C * sizeof(int));
  if (st != OMC_OK) return st;
  DenseTermsDev T{};
  T.n_terms = terms->n_terms;
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    const bool on = k < terms->n_terms;
    T.mat[k] = on ? terms->mat[k] : nullptr;
    T.rhs[k] = on ? terms->rhs[k] : nullptr;
    T.scale[k] = on ? terms->scale[k] : nullptr;
  }
  T.diag_chain = terms->diag_chain;
  rocblas_handle h = (rocblas_handle)ctx->blas;
  double* Q = ctx->dense_factor;
  hipLaunchKernelGGL(k_dense_assemble, dim3(gx(p * p) > 64 ? 64 : gx(p * p), (unsigned)C), dim3(256), 0, ctx->stream, T, p, C, Q);
  OMC_HIP_CHECK(hipGetLastError());
  if (p >= ctx->dense_blocked_min && !ctx->dense_use_rocsolver) {
    st = potrf_blocked(ctx, h, p, Q, C);
    if (st != OMC_OK) return st;
  } else {
    OMC_BLAS_CHECK(rocsolver_dpotrf_strided_batched(h, rocblas_fill_lower, (rocblas_int)p, Q, (rocblas_int)p,
                                                    (rocblas_stride)(p * p), ctx->dense_info, (rocblas_int)C));
  }
  hipLaunchKernelGGL(k_dense_post_factor, dim3((unsigned)C), dim3(256), 0, ctx->stream, p, C, Q, ctx->dense_info,
                     logdet_out, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  // w = L^{-1} b in x_out
  hipLaunchKernelGGL(k_dense_rhs, dim3(gx(p), (unsigned)C), dim3(256), 0, ctx->stream, T, p, C, rhs_chain, ld_rhs, x_out, ld_x);
  OMC_HIP_CHECK(hipGetLastError());
  st = tri_solve(ctx, h, false, p, Q, x_out, ld_x, C);
  if (st != OMC_OK) return st;
  if (mean_out) {  // mu = L^{-T} w  (gmrf.py:196)
    hipLaunchKernelGGL(k_copy_rows, dim3(gx(p), (unsigned)C), dim3(256), 0, ctx->stream, p, x_out, ld_x, mean_out, ld_mean);
    OMC_HIP_CHECK(hipGetLastError());
    st = tri_solve(ctx, h, true, p, Q, mean_out, ld_mean, C);
    if (st != OMC_OK) return st;
  }
  // x = L^{-T} (w + z)
  hipLaunchKernelGGL(k_add_draw, dim3(gx((p + 1) / 2), (unsigned)C), dim3(256), 0, ctx->stream, p, C, ctx->chain_offset,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), z_inject, ld_z, x_out, ld_x);
  OMC_HIP_CHECK(hipGetLastError());
  st = tri_solve(ctx, h, true, p, Q, x_out, ld_x, C);
  if (st != OMC_OK) return st;
  return OMC_OK;
}

// ---- spectral route of the dense conjugate draw ---------------------------------------------------------------------
omc_status omc_dense_spectral_prepare(omc_ctx* ctx, int64_t p, const double* M, double* V_out, double* ev_out) {
  if (!ctx || p < 1 || p > 32768 || !M || !V_out || !ev_out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  rocblas_handle h = (rocblas_handle)ctx->blas;
  st = omc_ensure_bytes(ctx, (void**)&ctx->dense_tmp, &ctx->dense_tmp_bytes, (size_t)p * sizeof(double));
  if (st != OMC_OK) return st;
  st = omc_ensure_bytes(ctx, (void**)&ctx->dense_info, &ctx->dense_info_bytes, sizeof(int));
  if (st != OMC_OK) return st;
  OMC_HIP_CHECK(hipMemcpyAsync(V_out, M, (size_t)p * p * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  OMC_BLAS_CHECK(rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, (rocblas_int)p, V_out, (rocblas_int)p, ev_out,
                                  ctx->dense_tmp, ctx->dense_info));
  int info = 0;
  OMC_HIP_CHECK(hipMemcpyAsync(&info, ctx->dense_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (info != 0) { omc_set_error_text("omc_dense_spectral_prepare: the eigensolver did not converge"); return OMC_HIP_ERROR; }
  return OMC_OK;
}

omc_status omc_dense_spectral_sample(omc_ctx* ctx, int64_t p, const omc_dense_terms* terms, int32_t k_mat, const double* V,
                                     const double* ev, const double* rhs_chain, int64_t ld_rhs, const double* z_inject,
                                     int64_t ld_z, uint64_t draw_index, double* x_out, int64_t ld_x, double* mean_out,
                                     int64_t ld_mean, double* logdet_out) {
  if (!ctx || p < 1 || !terms || terms->n_terms < 1 || terms->n_terms > OMC_MAX_TERMS || k_mat < 0 || k_mat >= terms->n_terms)
    return OMC_INVALID_ARG;
  if (!V || !ev || !x_out || ld_x < p || (rhs_chain && ld_rhs < p) || (z_inject && ld_z < p) || (mean_out && ld_mean < p))
    return OMC_INVALID_ARG;
  if (terms->diag_chain || !terms->mat[k_mat]) return OMC_INVALID_ARG;
  for (int k = 0; k < terms->n_terms; ++k)
    if (k != k_mat && terms->mat[k]) return OMC_INVALID_ARG;  // every other term must be a scaled identity
  const int64_t C = ctx->n_chains;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  rocblas_handle h = (rocblas_handle)ctx->blas;
  st = omc_ensure_bytes(ctx, (void**)&ctx->dense_factor, &ctx->dense_factor_bytes, (size_t)2 * C * p * sizeof(double));
  if (st != OMC_OK) return st;
  double* W = ctx->dense_factor;       // [C][p]: V' rhs, then y
  double* Mw = W + C * p;              // [C][p]: m in the eigenbasis
  DenseTermsDev T{};
  T.n_terms = terms->n_terms;
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    const bool on = k < terms->n_terms;
    T.mat[k] = on ? terms->mat[k] : nullptr;
    T.rhs[k] = on ? terms->rhs[k] : nullptr;
    T.scale[k] = on ? terms->scale[k] : nullptr;
  }
  T.diag_chain = nullptr;
  // b_c into x_out, W = V' B  (all chains: one GEMM; [C][p] row-major = column-major p x C)
  hipLaunchKernelGGL(k_dense_rhs, dim3(gx(p), (unsigned)C), dim3(256), 0, ctx->stream, T, p, C, rhs_chain, ld_rhs, x_out, ld_x);
  OMC_HIP_CHECK(hipGetLastError());
  const double one = 1.0, zero = 0.0;
  OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, (rocblas_int)p, (rocblas_int)C, (rocblas_int)p,
                               &one, V, (rocblas_int)p, x_out, (rocblas_int)ld_x, &zero, W, (rocblas_int)p));
  hipLaunchKernelGGL(k_spectral_scale, dim3((unsigned)C), dim3(256), 0, ctx->stream, T, (int)k_mat, p, C, ev, ctx->chain_offset,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL), z_inject, ld_z, W, p, mean_out ? Mw : nullptr, p,
                     logdet_out, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, (rocblas_int)p, (rocblas_int)C, (rocblas_int)p, &one,
                               V, (rocblas_int)p, W, (rocblas_int)p, &zero, x_out, (rocblas_int)ld_x));
  if (mean_out)
    OMC_BLAS_CHECK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, (rocblas_int)p, (rocblas_int)C, (rocblas_int)p, &one,
                                 V, (rocblas_int)p, Mw, (rocblas_int)p, &zero, mean_out, (rocblas_int)ld_mean));
  return OMC_OK;
}

// X is [n][p] row-major = column-major p x n (lda = p).
omc_status omc_gram(omc_ctx* ctx, int64_t n, int64_t p, const double* X, const double* w, double* G_out) {
  if (!ctx || n < 1 || p < 1 || !X || !G_out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  if (!ctx->gram_use_rocblas && p <= 46340) return omc_gram_mfma_launch(ctx, n, p, X, w, G_out);  // own fp64 MFMA kernel (omc_gram.hip)
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  const double* B = X;
  if (w) {
    st = omc_ensure_bytes(ctx, (void**)&ctx->dense_tmp, &ctx->dense_tmp_bytes, (size_t)n * p * sizeof(double));
    if (st != OMC_OK) return st;
    hipLaunchKernelGGL(k_scale_rows, dim3(gx(n * p)), dim3(256), 0, ctx->stream, n, p, X, w, ctx->dense_tmp);
    OMC_HIP_CHECK(hipGetLastError());
    B = ctx->dense_tmp;
  }
  const double one = 1.0, zero = 0.0;
  OMC_BLAS_CHECK(rocblas_dgemm((rocblas_handle)ctx->blas, rocblas_operation_none, rocblas_operation_transpose,
                               (rocblas_int)p, (rocblas_int)p, (rocblas_int)n, &one, X, (rocblas_int)p, B,
                               (rocblas_int)p, &zero, G_out, (rocblas_int)p));
  return OMC_OK;
}

omc_status omc_design_rhs(omc_ctx* ctx, int64_t n, int64_t p, const double* X, const double* w, const double* y,
                          double* out) {
  if (!ctx || n < 1 || p < 1 || !X || !y || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  const double* v = y;
  if (w) {
    st = omc_ensure_bytes(ctx, (void**)&ctx->dense_tmp, &ctx->dense_tmp_bytes, (size_t)n * sizeof(double));
    if (st != OMC_OK) return st;
    hipLaunchKernelGGL(k_scale_rows, dim3(gx(n)), dim3(256), 0, ctx->stream, n, (int64_t)1, y, w, ctx->dense_tmp);
    OMC_HIP_CHECK(hipGetLastError());
    v = ctx->dense_tmp;
  }
  const double one = 1.0, zero = 0.0;
  // out = A v with A = column-major p x n
  OMC_BLAS_CHECK(rocblas_dgemv((rocblas_handle)ctx->blas, rocblas_operation_none, (rocblas_int)p, (rocblas_int)n, &one, X,
                               (rocblas_int)p, v, 1, &zero, out, 1));
  return OMC_OK;
}

// out[c][i] = sum_s part[s][c][i]  (slices of a long contraction, added in slice order: deterministic)
__global__ void __launch_bounds__(256) k_sum_slices(int64_t n, int64_t C, int S, const double* __restrict__ part, double* __restrict__ out,
                                                    int64_t ld_out) {
  const int64_t c = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int s = 0; s < S; ++s) acc += part[((int64_t)s * C + c) * n + i];
    out[c * ld_out + i] = acc;
  }
}

omc_status omc_design_predict(omc_ctx* ctx, int64_t n, int64_t p, const double* X, const double* beta, int64_t ld_beta,
                              double* fitted, int64_t ld_fitted) {
  if (!ctx || n < 1 || p < 1 || !X || !beta || !fitted || ld_beta < p || ld_fitted < n) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  omc_status st = omc_ensure_blas(ctx);
  if (st != OMC_OK) return st;
  const double one = 1.0, zero = 0.0;
  // A short output under a long contraction (the transposed use: A'W applied to per-chain vectors of the observation space,
  // n = parameters, p = observations) leaves the library with a handful of output tiles -- 16 workgroups for 1000 x 256 x
  // 10 000, 1.6 ms.  Then the contraction is cut into slices (one strided-batched GEMM, a slice per batch entry) and the
  // slices are added in order.
  const int64_t C = ctx->n_chains;
  const int64_t tiles = ((n + 127) / 128) * ((C + 127) / 128);
  if (p >= 4096 && tiles < 128) {
    int64_t S = 256 / tiles;
    if (S > p / 512) S = p / 512;
    if (S > 64) S = 64;
    if (S >= 2) {
      const int64_t ks = p / S;  // the last slice takes the remainder in a call of its own
      st = omc_ensure_bytes(ctx, (void**)&ctx->slice_buf, &ctx->slice_buf_bytes, (size_t)S * C * n * sizeof(double));
      if (st != OMC_OK) return st;
      double* part = ctx->slice_buf;
      OMC_BLAS_CHECK(rocblas_dgemm_strided_batched((rocblas_handle)ctx->blas, rocblas_operation_transpose, rocblas_operation_none,
                                                   (rocblas_int)n, (rocblas_int)C, (rocblas_int)ks, &one, X, (rocblas_int)p,
                                                   (rocblas_stride)ks, beta, (rocblas_int)ld_beta, (rocblas_stride)ks, &zero, part,
                                                   (rocblas_int)n, (rocblas_stride)(n * C), (rocblas_int)(S - 1)));
      const int64_t k_last = p - (S - 1) * ks;
      OMC_BLAS_CHECK(rocblas_dgemm((rocblas_handle)ctx->blas, rocblas_operation_transpose, rocblas_operation_none, (rocblas_int)n,
                                   (rocblas_int)C, (rocblas_int)k_last, &one, X + (S - 1) * ks, (rocblas_int)p,
                                   beta + (S - 1) * ks, (rocblas_int)ld_beta, &zero, part + (S - 1) * n * C, (rocblas_int)n));
      int64_t gx_ = (n + 255) / 256;
      if (gx_ > 64) gx_ = 64;
      hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)gx_, (unsigned)C), dim3(256), 0, ctx->stream, n, C, (int)S, part, fitted, ld_fitted);
      OMC_HIP_CHECK(hipGetLastError());
      return OMC_OK;
    }
  }
  // fitted (col-major n x C, ld = ld_fitted) = A' (n x p) * Beta (col-major p x C, ld = ld_beta)
  OMC_BLAS_CHECK(rocblas_dgemm((rocblas_handle)ctx->blas, rocblas_operation_transpose, rocblas_operation_none,
                               (rocblas_int)n, (rocblas_int)ctx->n_chains, (rocblas_int)p, &one, X, (rocblas_int)p, beta,
                               (rocblas_int)ld_beta, &zero, fitted, (rocblas_int)ld_fitted));
  return OMC_OK;
}

omc_status omc_chain_copy(omc_ctx* ctx, int64_t n, const double* src, int64_t ld_src, double* dst, int64_t ld_dst) {
  if (!ctx || n < 1 || !src || !dst || ld_src < n || ld_dst < n) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  if (src == dst && ld_src == ld_dst) return OMC_OK;
  unsigned g = gx(n);
  if (g > 64) g = 64;
  hipLaunchKernelGGL(k_copy_rows, dim3(g, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, n, src, ld_src, dst, ld_dst);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_weighted_resid_sq(omc_ctx* ctx, int64_t n, const double* y, const double* fitted, int64_t ld_fitted,
                                 const double* w, double* out) {
  if (!ctx || n < 1 || !y || !fitted || ld_fitted < n || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_weighted_resid_sq, dim3((unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, n, y, fitted,
                     ld_fitted, w, out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"
