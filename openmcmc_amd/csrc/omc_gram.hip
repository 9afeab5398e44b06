// G = X' diag(w) X on the fp64 matrix cores (gfx950: v_mfma_f64_16x16x4_f64).
//
// What it replaces: the one dense contraction of the conjugate update and of the gradient branch,
//   grad.T @ Q @ grad with grad = X' (location_scale.py:238-241), Q = diag(w) -- 2 p^2 n flop, 2e10 at BASELINE
// configs[1] (n = 10 000 observations, p = 1 000 coefficients); chain- and iteration-invariant, formed once per model.
//
// Shape of the problem on MI355X: the output is small (p x p = 8 MB), the contraction long (n).  A tiling of the
// output alone gives 8 x 8 tiles of 128 -- a quarter of the 256 CUs.  So:
//   * only the tiles on and below the diagonal are computed (G is symmetric: 36 of 64 at p = 1000), the rest mirrored;
//   * the contraction is cut into `splits` slices, one workgroup per (tile, slice), the slices' partial tiles go to a
//     workspace and a second kernel adds them in a fixed order (deterministic: no atomics) and mirrors;
//   * a workgroup = 4 waves in 2 x 2, each wave a 64 x 64 block = 4 x 4 MFMA tiles of 16 x 16 (128 accumulator
//     registers); per 4 rows of X a wave reads 4 + 4 operand values per lane from LDS and issues 16 MFMAs;
//   * X is row-major [n][p]: a slab of BK rows x 128 columns is BK contiguous 1-KB pieces -- coalesced 8-byte loads
//     straight into the [row][column] LDS image both operands want (A[i][k] = X[k][i], B[k][j] = w_k X[k][j]: lane l
//     takes [k0 + l/16][i0 + l%16] for either).  Row stride 128 + 16 doubles: the two rows a half-wave touches fall
//     into different halves of the 64 banks.  The weights ride in LDS next to the slab and are applied to the B
//     operand in registers (no scaled copy of X: the library route needed an n x p temporary);
//   * the next slab is fetched into registers while the current one is multiplied; two workgroups share a CU.
#include "omc_common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define GR_TS 128   // output tile
#define GR_BK 16    // rows of X per slab
#define GR_LD (GR_TS + 16)

__global__ void __launch_bounds__(256, 2) k_gram_mfma(int64_t n, int p, const double* __restrict__ X, const double* __restrict__ w,
                                                      double* __restrict__ part, int ntile, int64_t kchunk) {
  __shared__ double As[GR_BK][GR_LD];
  __shared__ double Bs[GR_BK][GR_LD];
  __shared__ double ws[GR_BK];
  // tile pair below or on the diagonal from the linear index: bi >= bj
  int t = blockIdx.x, bi = 0;
  while (t >= bi + 1) { t -= bi + 1; ++bi; }
  const int bj = t;
  const bool diag_tile = bi == bj;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t kbeg = (int64_t)blockIdx.y * kchunk;
  const int64_t kend = (kbeg + kchunk < n) ? kbeg + kchunk : n;
  const int ci = bi * GR_TS, cj = bj * GR_TS;

  double4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};

  // slab element e = q * 256 + tid : row e / 128, column e % 128  (8 per panel and thread)
  double ra[8], rb[8], rw = 0.0;
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = q * 256 + tid, r = e >> 7, c = e & 127;
      const int64_t k = k0 + r;
      const bool rowok = k < kend;
      ra[q] = (rowok && ci + c < p) ? X[k * p + ci + c] : 0.0;
      if (!diag_tile) rb[q] = (rowok && cj + c < p) ? X[k * p + cj + c] : 0.0;
    }
    if (tid < GR_BK) rw = (k0 + tid < kend) ? (w ? w[k0 + tid] : 1.0) : 0.0;
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = q * 256 + tid, r = e >> 7, c = e & 127;
      As[r][c] = ra[q];
      if (!diag_tile) Bs[r][c] = rb[q];
    }
    if (tid < GR_BK) ws[tid] = rw;
  };

  fetch(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += GR_BK) {
    __syncthreads();  // the previous slab has been consumed
    stage();
    __syncthreads();
    if (k0 + GR_BK < kend) fetch(k0 + GR_BK);  // in flight under the multiplications below
    const double(*Bp)[GR_LD] = diag_tile ? As : Bs;
#pragma unroll
    for (int kk = 0; kk < GR_BK; kk += 4) {
      const int kr = kk + (lane >> 4), cl = lane & 15;
      const double wk = ws[kr];
      double a[4], b[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = As[kr][wr * 64 + m * 16 + cl];
#pragma unroll
      for (int m = 0; m < 4; ++m) b[m] = Bp[kr][wc * 64 + m * 16 + cl] * wk;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[q], acc[m][q], 0, 0, 0);
    }
  }
  // C/D of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * register
  double* out = part + (int64_t)blockIdx.y * p * p;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ci + wr * 64 + m * 16 + (lane >> 4) + 4 * r, j = cj + wc * 64 + q * 16 + (lane & 15);
        if (i < p && j < p) out[(int64_t)i * p + j] = acc[m][q][r];
      }
}

// G[i][j] = sum over the slices, in index order, of element (max(i,j), min(i,j)) of the partial tiles
__global__ void __launch_bounds__(256) k_gram_reduce(int p, int splits, const double* __restrict__ part, double* __restrict__ G) {
  const int64_t total = (int64_t)p * p;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / p), j = (int)(e - (int64_t)i * p);
    // everything above the diagonal is the mirror image of what lies below it -- also inside the diagonal tiles, whose
    // two halves were both computed but with the weight on different operands (equal only up to rounding)
    const int64_t src = (i >= j) ? e : (int64_t)j * p + i;
    double s = 0.0;
    for (int k = 0; k < splits; ++k) s += part[(int64_t)k * total + src];
    G[e] = s;
  }
}

extern "C" omc_status omc_gram_mfma_launch(omc_ctx* ctx, int64_t n, int64_t p, const double* X, const double* w, double* G_out) {
  const int ntile = (int)((p + GR_TS - 1) / GR_TS);
  const int pairs = ntile * (ntile + 1) / 2;
  int dev_cus = 256;
  hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
  // two workgroups per CU; slices at least a few slabs long
  int splits = (2 * dev_cus) / pairs;
  if (splits < 1) splits = 1;
  const int64_t max_splits = (n + 4 * GR_BK - 1) / (4 * GR_BK);
  if (splits > max_splits) splits = (int)max_splits;
  int64_t kchunk = (n + splits - 1) / splits;
  kchunk = (kchunk + GR_BK - 1) / GR_BK * GR_BK;
  splits = (int)((n + kchunk - 1) / kchunk);
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->dense_tmp, &ctx->dense_tmp_bytes, (size_t)splits * p * p * sizeof(double));
  if (st != OMC_OK) return st;
  hipLaunchKernelGGL(k_gram_mfma, dim3((unsigned)pairs, (unsigned)splits), dim3(256), 0, ctx->stream, n, (int)p, X, w,
                     ctx->dense_tmp, ntile, kchunk);
  OMC_HIP_CHECK(hipGetLastError());
  int64_t grid = (p * p + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (int)p, splits, ctx->dense_tmp, G_out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}
