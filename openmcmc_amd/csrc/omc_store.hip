// On-device posterior summaries of the sample store: quantiles of store[param] without a sort
// and without moving the store (3 GB and more at cfg3) off the GPU.  The reference keeps its store in host arrays
// (mcmc.py:105-111, sampler/sampler.py:89-118) and users call np.quantile on them; this is that call for the device-resident
// store of C chains.
//
// Both summaries are column-wise order statistics of one dense row-major matrix [R][K]:
//   pooled over chains and iterations : R = n_iter * C rows, K = size columns          (store is [n_iter][C][size])
//   per chain                         : R = n_iter rows,     K = C * size columns
// Exact selection by most-significant-digit radix refinement on the order-preserving 64-bit image of a double: eight
// passes of 8 bits; a pass histograms, per target order statistic, the next digit of the values that still share the
// target's prefix; a scan picks the digit the target's rank falls into.  A quantile needs two neighbouring order statistics
// (np.quantile's default "linear" method), so nq quantiles are 2 nq targets that refine side by side in the same passes.
//
// Bound: HBM (8 reads of the matrix per group of four quantiles, nothing written but histograms).  A workgroup owns a
// tile of 16 adjacent columns (128-byte row pieces) and a slice of the rows; its histograms live in LDS (ds_add_u32) and
// reach the global ones as one atomic per touched bin.
#include <math.h>

#include "omc_common.h"

namespace {

constexpr int Q_TARGETS = 8;   // order statistics refined side by side (4 quantiles)
constexpr int Q_COLS = 16;     // columns per workgroup
constexpr int Q_ROWSTEP = 16;  // rows per step of a 256-thread workgroup

__device__ __forceinline__ uint64_t q_key(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  if (v != v) return ~0ull;  // every NaN sorts last, like np.sort
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double q_val(uint64_t k) {
  const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// np.quantile, method "linear": virtual index (n - 1) q, its floor and the next index (lib/_function_base_impl.py:
// _get_indexes / _get_gamma) for a column with nv valid values -> the two ranks and the interpolation weight
__device__ __forceinline__ void q_ranks(int64_t nv, double q, int64_t& lo, int64_t& hi, double& frac) {
  const double h = (double)(nv - 1) * q;
  lo = (int64_t)floor(h);
  if (lo < 0) lo = 0;
  if (lo > nv - 1) lo = nv - 1;
  hi = lo + 1 < nv ? lo + 1 : nv - 1;
  if (h >= (double)(nv - 1)) lo = hi = nv - 1;
  if (nv <= 0) lo = hi = 0;
  frac = (nv > 0 && h < (double)(nv - 1)) ? h - (double)lo : 0.0;
}

// one wave on one row of 256 counts: the bin the rank falls into and the count in front of it.  (No lane claims the rank
// only when the column has no valid value at all -- the result is NaN then anyway.)
__device__ __forceinline__ void q_pick(const uint32_t* hrow, int lane, int64_t rk, int& dg_out, int64_t& before_out) {
  const uint4 c = *reinterpret_cast<const uint4*>(hrow + 4 * lane);
  const int64_t mine = (int64_t)c.x + c.y + c.z + c.w;
  int64_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  const int64_t excl = incl - mine;
  const unsigned long long m = __ballot(rk >= excl && rk < incl);
  const int src = m ? __ffsll((long long)m) - 1 : 63;
  int dg = 4 * lane;
  int64_t before = excl;
  if (rk >= before + c.x) { before += c.x; ++dg; if (rk >= before + c.y) { before += c.y; ++dg; if (rk >= before + c.z) { before += c.z; ++dg; } } }
  dg_out = __shfl(dg, src, 64);
  before_out = __shfl(before, src, 64);
}

// numpy's _lerp (lib/_function_base_impl.py), operation by operation: a + (b - a) t, and b - (b - a)(1 - t) where t >= 0.5.
// (HIP's __dmul_rn / __dadd_rn are plain operators the compiler may still fuse into an fma: contraction is switched off here)
__device__ __noinline__ double q_lerp(double a, double b, double t) {
#pragma clang fp contract(off)
  const double diff = b - a;
  const double up = a + diff * t, down = b - diff * (1.0 - t);
  return t >= 0.5 ? down : up;
}

// hist [T][Kc][256], prefix [T][Kc]; pass 0 fills target 0's histogram only (no prefix yet: the same for every target)
__global__ void __launch_bounds__(256) k_q_hist(const double* __restrict__ data, int64_t R, int64_t K, int64_t k0, int64_t Kc, int pass, int T,
                                                const uint64_t* __restrict__ prefix, uint32_t* __restrict__ hist,
                                                uint32_t* __restrict__ nan_count, int64_t rows_per_block) {
  extern __shared__ uint32_t lh[];  // [Tl][Q_COLS][256], Tl = 1 in pass 0
  const int Tl = pass == 0 ? 1 : T;
  const int tid = threadIdx.x;
  for (int i = tid; i < Tl * Q_COLS * 256; i += 256) lh[i] = 0;
  const int col = tid & (Q_COLS - 1);
  const int64_t kc = (int64_t)blockIdx.x * Q_COLS + col;
  const bool live = kc < Kc;
  uint64_t pf[Q_TARGETS];
#pragma unroll
  for (int t = 0; t < Q_TARGETS; ++t) pf[t] = (pass > 0 && t < T && live) ? prefix[(int64_t)t * Kc + kc] : 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  const int shift = 56 - 8 * pass;
  const double* p = data + k0 + kc;
  uint32_t nans = 0;
  if (live) {
    int64_t r = r0 + (tid >> 4);
    // four rows in flight per thread: the loop is a chain of independent loads feeding LDS atomics
    for (; r + 3 * Q_ROWSTEP < r1; r += 4 * Q_ROWSTEP) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(r + u * Q_ROWSTEP) * K];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t key = q_key(v[u]);
        if (pass == 0) {
          nans += key == ~0ull;
          atomicAdd(&lh[col * 256 + (int)(key >> 56)], 1u);
        } else {
          const uint64_t hi = key >> (shift + 8);
          const int dg = (int)(key >> shift) & 255;
#pragma unroll
          for (int t = 0; t < Q_TARGETS; ++t)
            if (t < T && hi == pf[t]) atomicAdd(&lh[(t * Q_COLS + col) * 256 + dg], 1u);
        }
      }
    }
    for (; r < r1; r += Q_ROWSTEP) {
      const uint64_t key = q_key(p[r * K]);
      if (pass == 0) {
        nans += key == ~0ull;
        atomicAdd(&lh[col * 256 + (int)(key >> 56)], 1u);
      } else {
        const uint64_t hi = key >> (shift + 8);
        const int dg = (int)(key >> shift) & 255;
#pragma unroll
        for (int t = 0; t < Q_TARGETS; ++t)
          if (t < T && hi == pf[t]) atomicAdd(&lh[(t * Q_COLS + col) * 256 + dg], 1u);
      }
    }
    if (pass == 0 && nans) atomicAdd(&nan_count[kc], nans);
  }
  __syncthreads();
  const int64_t kbase = (int64_t)blockIdx.x * Q_COLS;
  for (int i = tid; i < Tl * Q_COLS * 256; i += 256) {
    const uint32_t c = lh[i];
    if (!c) continue;
    const int t = i / (Q_COLS * 256), cl = (i / 256) % Q_COLS, dg = i & 255;
    if (kbase + cl < Kc) atomicAdd(&hist[((int64_t)t * Kc + kbase + cl) * 256 + dg], c);
  }
}

// one wave per (target, column): the digit the target's rank falls into; the histogram is cleared for the next pass
__global__ void __launch_bounds__(256) k_q_scan(int pass, int T, int64_t Kc, int64_t R, uint32_t* __restrict__ hist, uint64_t* __restrict__ prefix,
                                                int64_t* __restrict__ rank, const uint32_t* __restrict__ nan_count, const double* __restrict__ q,
                                                double* __restrict__ frac) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (int64_t)T * Kc) return;
  const int t = (int)(w / Kc);
  const int64_t kc = w % Kc;
  int64_t rk;
  if (pass == 0) {
    int64_t lo, hi;
    double fr;
    q_ranks(R - (int64_t)nan_count[kc], q[t >> 1], lo, hi, fr);
    rk = (t & 1) ? hi : lo;
    if (!(t & 1) && lane == 0) frac[(int64_t)(t >> 1) * Kc + kc] = fr;
  } else {
    rk = rank[(int64_t)t * Kc + kc];
  }
  uint32_t* hrow = hist + ((int64_t)(pass == 0 ? 0 : t) * Kc + kc) * 256;  // (pass 0: one histogram serves every target)
  int dg;
  int64_t before;
  q_pick(hrow, lane, rk, dg, before);
  if (lane == 0) {
    const uint64_t old = pass == 0 ? 0 : prefix[(int64_t)t * Kc + kc];
    prefix[(int64_t)t * Kc + kc] = (old << 8) | (uint64_t)dg;
    rank[(int64_t)t * Kc + kc] = rk - before;
  }
  // (pass 0's shared row is cleared by the host behind this launch; later passes clear their own)
  if (pass > 0) *reinterpret_cast<uint4*>(hrow + 4 * lane) = make_uint4(0, 0, 0, 0);
}

__global__ void k_q_finish(int nq, int64_t Kc, int64_t k0, int64_t K, int64_t R, const uint64_t* __restrict__ prefix,
                           const double* __restrict__ frac, const uint32_t* __restrict__ nan_count, int omit_nan, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)nq * Kc) return;
  const int j = (int)(i / Kc);
  const int64_t kc = i % Kc;
  const uint32_t nn = nan_count[kc];
  double r;
  if ((int64_t)nn >= R || (nn && !omit_nan)) {
    r = __longlong_as_double(0x7ff8000000000000LL);
  } else {
    const double a = q_val(prefix[(int64_t)(2 * j) * Kc + kc]), b = q_val(prefix[(int64_t)(2 * j + 1) * Kc + kc]);
    const double t = frac[(int64_t)j * Kc + kc];
    r = q_lerp(a, b, t);
  }
  out[(int64_t)j * K + k0 + kc] = r;
}

// Many columns (the per-chain summaries: C x size of them): a workgroup OWNS OC adjacent columns -- all their rows -- and runs the
// eight passes in one launch; histograms, prefixes and ranks never leave LDS, and the re-reads of the workgroup's column
// strip come out of L2 / MALL for short stores.  (The row-sliced route above needs 8 KB of global histogram per column
// and two launches per pass: at 10 M columns that is hundreds of chunks.)
constexpr int OC = 8;            // columns per workgroup: 64-byte row pieces, 64 KB of histograms -> two workgroups per CU
constexpr int OROW = 256 / OC;   // rows per step

__global__ void __launch_bounds__(256) k_q_owned(const double* __restrict__ data, int64_t R, int64_t K, int T, int nq,
                                                 const double* __restrict__ q, int omit_nan, double* __restrict__ out) {
  extern __shared__ uint32_t lh[];  // [T][OC][256]
  __shared__ uint64_t s_prefix[Q_TARGETS][OC];
  __shared__ int64_t s_rank[Q_TARGETS][OC];
  __shared__ double s_frac[Q_TARGETS / 2][OC];
  __shared__ uint32_t s_nan[OC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = tid & (OC - 1);
  const int64_t k = (int64_t)blockIdx.x * OC + col;
  const bool live = k < K;
  const double* p = data + (live ? k : 0);
  if (tid < OC) s_nan[tid] = 0;
  for (int pass = 0; pass < 8; ++pass) {
    const int Tl = pass == 0 ? 1 : T;
    for (int i = tid; i < Tl * OC * 256; i += 256) lh[i] = 0;
    uint64_t pf[Q_TARGETS];
#pragma unroll
    for (int t = 0; t < Q_TARGETS; ++t) pf[t] = (pass > 0 && t < T) ? s_prefix[t][col] : 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    if (live) {
      uint32_t nans = 0;
      auto take = [&](double v) {
        const uint64_t key = q_key(v);
        if (pass == 0) {
          nans += key == ~0ull;
          atomicAdd(&lh[col * 256 + (int)(key >> 56)], 1u);
        } else {
          const uint64_t hi = key >> (shift + 8);
          const int dg = (int)(key >> shift) & 255;
#pragma unroll
          for (int t = 0; t < Q_TARGETS; ++t)
            if (t < T && hi == pf[t]) atomicAdd(&lh[(t * OC + col) * 256 + dg], 1u);
        }
      };
      int64_t r = tid / OC;
      for (; r + 3 * OROW < R; r += 4 * OROW) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[(r + u * OROW) * K];
#pragma unroll
        for (int u = 0; u < 4; ++u) take(v[u]);
      }
      for (; r < R; r += OROW) take(p[r * K]);
      if (pass == 0 && nans) atomicAdd(&s_nan[col], nans);
    }
    __syncthreads();
    for (int pr = wave; pr < T * OC; pr += 4) {  // (target, column) pairs, one wave each
      const int t = pr / OC, c = pr % OC;
      int64_t rk;
      if (pass == 0) {
        int64_t lo, hi;
        double fr;
        q_ranks(R - (int64_t)s_nan[c], q[t >> 1], lo, hi, fr);
        rk = (t & 1) ? hi : lo;
        if (!(t & 1) && lane == 0) s_frac[t >> 1][c] = fr;
      } else {
        rk = s_rank[t][c];
      }
      int dg;
      int64_t before;
      q_pick(&lh[((pass == 0 ? 0 : t) * OC + c) * 256], lane, rk, dg, before);
      if (lane == 0) {
        s_prefix[t][c] = ((pass == 0 ? 0 : s_prefix[t][c]) << 8) | (uint64_t)dg;
        s_rank[t][c] = rk - before;
      }
    }
    __syncthreads();
  }
  if (tid < nq * OC) {
    const int j = tid / OC, c = tid % OC;
    const int64_t kk = (int64_t)blockIdx.x * OC + c;
    if (kk < K) {
      const uint32_t nn = s_nan[c];
      double r;
      if ((int64_t)nn >= R || (nn && !omit_nan)) r = __longlong_as_double(0x7ff8000000000000LL);
      else r = q_lerp(q_val(s_prefix[2 * j][c]), q_val(s_prefix[2 * j + 1][c]), s_frac[j][c]);
      out[(int64_t)j * K + kk] = r;
    }
  }
}

// Short columns (a chain's few hundred stored iterations): with 8-bit digits a pass zeroes and scans 64 KB of histograms to place
// a handful of values (139 ms for 10 M columns of 128).  Here the digits are 4 bits wide: sixteen passes, 16-bin histograms
// (T x 16 columns x 16 bins = 8 KB), a (target, column) pair picked by sixteen lanes -- four pairs per wave at once.
constexpr int SC = 16;            // columns per workgroup: 128-byte row pieces
constexpr int SROW = 256 / SC;

__global__ void __launch_bounds__(256) k_q_owned4(const double* __restrict__ data, int64_t R, int64_t K, int T, int nq,
                                                  const double* __restrict__ q, int omit_nan, double* __restrict__ out) {
  __shared__ uint32_t lh[Q_TARGETS][SC][16];
  __shared__ uint64_t s_prefix[Q_TARGETS][SC];
  __shared__ int64_t s_rank[Q_TARGETS][SC];
  __shared__ double s_frac[Q_TARGETS / 2][SC];
  __shared__ uint32_t s_nan[SC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = tid & (SC - 1);
  const int64_t k = (int64_t)blockIdx.x * SC + col;
  const bool live = k < K;
  const double* p = data + (live ? k : 0);
  if (tid < SC) s_nan[tid] = 0;
  for (int pass = 0; pass < 16; ++pass) {
    for (int i = tid; i < Q_TARGETS * SC * 16; i += 256) (&lh[0][0][0])[i] = 0;
    uint64_t pf[Q_TARGETS];
#pragma unroll
    for (int t = 0; t < Q_TARGETS; ++t) pf[t] = (pass > 0 && t < T) ? s_prefix[t][col] : 0;
    __syncthreads();
    const int shift = 60 - 4 * pass;
    if (live) {
      uint32_t nans = 0;
      for (int64_t r = tid / SC; r < R; r += SROW) {
        const uint64_t key = q_key(p[r * K]);
        if (pass == 0) {
          nans += key == ~0ull;
          atomicAdd(&lh[0][col][(int)(key >> 60)], 1u);
        } else {
          const uint64_t hi = key >> (shift + 4);
          const int dg = (int)(key >> shift) & 15;
#pragma unroll
          for (int t = 0; t < Q_TARGETS; ++t)
            if (t < T && hi == pf[t]) atomicAdd(&lh[t][col][dg], 1u);
        }
      }
      if (pass == 0 && nans) atomicAdd(&s_nan[col], nans);
    }
    __syncthreads();
    // (target, column) pairs: sixteen lanes each, four pairs per wave and step
    const int sub = lane & 15, grp = lane >> 4;
    for (int base = 0; base < T * SC; base += 16) {   // sixteen pairs per step of the workgroup: four per wave
      const int pidx = base + 4 * wave + grp;
      const bool on = pidx < T * SC;
      const int t = on ? pidx / SC : 0, c = on ? pidx % SC : 0;
      int64_t rk = 0;
      if (pass == 0) {
        int64_t lo, hi;
        double fr;
        q_ranks(R - (int64_t)s_nan[c], q[t >> 1], lo, hi, fr);
        rk = (t & 1) ? hi : lo;
        if (on && !(t & 1) && sub == 0) s_frac[t >> 1][c] = fr;
      } else {
        rk = s_rank[t][c];
      }
      const int64_t mine = (int64_t)lh[pass == 0 ? 0 : t][c][sub];
      int64_t incl = mine;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const int64_t o = __shfl_up(incl, d, 16);
        if (sub >= d) incl += o;
      }
      const int64_t excl = incl - mine;
      const unsigned long long m = (__ballot(rk >= excl && rk < incl) >> (16 * grp)) & 0xffffull;
      const int dg = m ? __ffsll((long long)m) - 1 : 15;  // (no bin claims the rank only when the column has no valid value: NaN anyway)
      const int64_t before = __shfl(excl, 16 * grp + dg, 64);
      if (on && sub == 0) {
        s_prefix[t][c] = ((pass == 0 ? 0 : s_prefix[t][c]) << 4) | (uint64_t)dg;
        s_rank[t][c] = rk - before;
      }
    }
    __syncthreads();
  }
  if (tid < nq * SC) {
    const int j = tid / SC, c = tid % SC;
    const int64_t kk = (int64_t)blockIdx.x * SC + c;
    if (kk < K) {
      const uint32_t nn = s_nan[c];
      double r;
      if ((int64_t)nn >= R || (nn && !omit_nan)) r = __longlong_as_double(0x7ff8000000000000LL);
      else r = q_lerp(q_val(s_prefix[2 * j][c]), q_val(s_prefix[2 * j + 1][c]), s_frac[j][c]);
      out[(int64_t)j * K + kk] = r;
    }
  }
}

// mean and unbiased variance of every column of [R][K]: row slices combined by Chan's pairwise update through a small
// [slices][2][K] scratch (deterministic: fixed slice boundaries, fixed combination order)
__global__ void __launch_bounds__(256) k_col_moments_part(const double* __restrict__ data, int64_t R, int64_t K, int64_t rows_per_block,
                                                          double* __restrict__ part /*[slices][3][K]*/) {
  __shared__ double sm[3][Q_ROWSTEP][Q_COLS];
  const int tid = threadIdx.x, col = tid & (Q_COLS - 1), rr = tid >> 4;
  const int64_t k = (int64_t)blockIdx.x * Q_COLS + col;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  double cnt = 0.0, mean = 0.0, m2 = 0.0;
  if (k < K)
    for (int64_t r = r0 + rr; r < r1; r += Q_ROWSTEP) {
      const double v = data[r * K + k];
      cnt += 1.0;
      const double d = v - mean;
      mean += d / cnt;
      m2 = fma(d, v - mean, m2);
    }
  sm[0][rr][col] = cnt; sm[1][rr][col] = mean; sm[2][rr][col] = m2;
  __syncthreads();
  if (rr == 0 && k < K) {
    for (int j = 1; j < Q_ROWSTEP; ++j) {
      const double cb = sm[0][j][col], mb = sm[1][j][col], qb = sm[2][j][col];
      if (cb == 0.0) continue;
      const double tot = cnt + cb, d = mb - mean;
      mean += d * (cb / tot);
      m2 += qb + d * d * (cnt * cb / tot);
      cnt = tot;
    }
    double* o = part + (int64_t)blockIdx.y * 3 * K;
    o[k] = cnt; o[K + k] = mean; o[2 * K + k] = m2;
  }
}
__global__ void k_col_moments_join(int64_t K, int slices, const double* __restrict__ part, double* __restrict__ mean_out, double* __restrict__ var_out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double cnt = 0.0, mean = 0.0, m2 = 0.0;
  for (int s = 0; s < slices; ++s) {
    const double* o = part + (int64_t)s * 3 * K;
    const double cb = o[k], mb = o[K + k], qb = o[2 * K + k];
    if (cb == 0.0) continue;
    const double tot = cnt + cb, d = mb - mean;
    mean += d * (cb / tot);
    m2 += qb + d * d * (cnt * cb / tot);
    cnt = tot;
  }
  if (mean_out) mean_out[k] = mean;
  if (var_out) var_out[k] = cnt > 1.0 ? m2 / (cnt - 1.0) : 0.0;
}

// every `every`-th stored iteration, packed: out[j] = store[first + j * every]  (rows of `row` doubles)
__global__ void k_thin_rows(const double* __restrict__ src, int64_t row, int64_t first, int64_t every, int64_t n_out, double* __restrict__ dst) {
  const int64_t j = blockIdx.y;
  const double* s = src + (first + j * every) * row;
  double* d = dst + j * row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

}  // namespace

extern "C" {

omc_status omc_store_quantiles(omc_ctx* ctx, int64_t n_iter, int64_t size, const double* store, int32_t pooled, int32_t n_q,
                               const double* q, int32_t omit_nan, double* out) {
  if (!ctx || n_iter < 1 || size < 1 || !store || n_q < 1 || !q || !out) return OMC_INVALID_ARG;
  for (int j = 0; j < n_q; ++j)
    if (!(q[j] >= 0.0 && q[j] <= 1.0)) return OMC_INVALID_ARG;  // np.quantile: "Quantiles must be in the range [0, 1]"
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const int64_t C = ctx->n_chains;
  const int64_t R = pooled ? n_iter * C : n_iter, K = pooled ? size : C * size;
  // workspace per column of a chunk: histograms 8 x 1 KB, prefixes and ranks 8 x 8 B each, fractions 4 x 8 B, NaN count
  const size_t per_col = (size_t)Q_TARGETS * 256 * 4 + (size_t)Q_TARGETS * 16 + (Q_TARGETS / 2) * 8 + 4;
  const size_t budget = (size_t)256 << 20;
  int64_t Kc = (int64_t)(budget / per_col) & ~(int64_t)(Q_COLS - 1);
  if (Kc > K) Kc = (K + Q_COLS - 1) & ~(int64_t)(Q_COLS - 1);
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->store_ws, &ctx->store_ws_bytes, (size_t)Kc * per_col + 64 + Q_TARGETS * 8);
  if (st != OMC_OK) return st;
  char* ws = (char*)ctx->store_ws;
  uint32_t* hist = (uint32_t*)ws;                                   ws += (size_t)Kc * Q_TARGETS * 256 * 4;
  uint64_t* prefix = (uint64_t*)ws;                                 ws += (size_t)Kc * Q_TARGETS * 8;
  int64_t* rank = (int64_t*)ws;                                     ws += (size_t)Kc * Q_TARGETS * 8;
  double* frac = (double*)ws;                                       ws += (size_t)Kc * (Q_TARGETS / 2) * 8;
  double* dq = (double*)ws;                                         ws += (Q_TARGETS / 2) * 8;
  uint32_t* nanc = (uint32_t*)ws;
  hipStream_t s = ctx->stream;
  // enough columns to fill the chip with column-owning workgroups (and rows few enough for one workgroup to walk): one launch
  const bool owned = K >= 8192 && R <= ((int64_t)1 << 22) && (K + OC - 1) / OC <= 0x7fffffffLL;
  for (int j0 = 0; j0 < n_q; j0 += Q_TARGETS / 2) {
    const int nq = (n_q - j0 < Q_TARGETS / 2) ? n_q - j0 : Q_TARGETS / 2;
    const int T = 2 * nq;
    // (the quantile levels of this group; pageable host memory: the copy has left the host buffer when the call returns)
    OMC_HIP_CHECK(hipMemcpyAsync(dq, q + j0, nq * sizeof(double), hipMemcpyHostToDevice, s));
    if (owned && R <= 2048) {  // short columns: 4-bit digits (see k_q_owned4)
      hipLaunchKernelGGL(k_q_owned4, dim3((unsigned)((K + SC - 1) / SC)), dim3(256), 0, s, store, R, K, T, nq, dq, (int)omit_nan,
                         out + (int64_t)j0 * K);
      OMC_HIP_CHECK(hipGetLastError());
      continue;
    }
    if (owned) {
      hipLaunchKernelGGL(k_q_owned, dim3((unsigned)((K + OC - 1) / OC)), dim3(256), (size_t)T * OC * 256 * 4, s, store, R, K, T, nq, dq,
                         (int)omit_nan, out + (int64_t)j0 * K);
      OMC_HIP_CHECK(hipGetLastError());
      continue;
    }
    for (int64_t k0 = 0; k0 < K; k0 += Kc) {
      const int64_t kc = (K - k0 < Kc) ? K - k0 : Kc;
      // every target's rows start at zero (a fresh workspace holds anything, and an earlier call laid its rows out for another
      // column count); from then on a pass leaves its rows cleared for the next one
      OMC_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)T * kc * 256 * 4, s));
      OMC_HIP_CHECK(hipMemsetAsync(nanc, 0, (size_t)kc * 4, s));
      const unsigned tiles = (unsigned)((kc + Q_COLS - 1) / Q_COLS);
      // row slices: enough workgroups to fill the chip (256 CUs x a few), no slice shorter than 64 rows
      int64_t slices = (4096 + tiles - 1) / tiles;
      if (slices > (R + 63) / 64) slices = (R + 63) / 64;
      if (slices < 1) slices = 1;
      if (slices > 65535) slices = 65535;
      const int64_t rpb = (R + slices - 1) / slices;
      for (int pass = 0; pass < 8; ++pass) {
        const size_t lds = (size_t)(pass == 0 ? 1 : T) * Q_COLS * 256 * 4;
        hipLaunchKernelGGL(k_q_hist, dim3(tiles, (unsigned)slices), dim3(256), lds, s, store, R, K, k0, kc, pass, T, prefix, hist, nanc, rpb);
        hipLaunchKernelGGL(k_q_scan, dim3((unsigned)(((int64_t)T * kc + 3) / 4)), dim3(256), 0, s, pass, T, kc, R, hist, prefix, rank, nanc, dq, frac);
        if (pass == 0) OMC_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)kc * 256 * 4, s));
      }
      hipLaunchKernelGGL(k_q_finish, dim3((unsigned)(((int64_t)nq * kc + 255) / 256)), dim3(256), 0, s, nq, kc, k0, K, R, prefix, frac, nanc,
                         (int)omit_nan, out + (int64_t)j0 * K);
      OMC_HIP_CHECK(hipGetLastError());
    }
  }
  return OMC_OK;
}

omc_status omc_store_thin(omc_ctx* ctx, int64_t n_iter, int64_t size, const double* store, int64_t first, int64_t every, double* out,
                          int64_t* n_out) {
  if (!ctx || n_iter < 1 || size < 1 || !store || first < 0 || first >= n_iter || every < 1 || !out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const int64_t row = ctx->n_chains * size;
  const int64_t n = (n_iter - first + every - 1) / every;
  if (n_out) *n_out = n;
  for (int64_t j0 = 0; j0 < n; j0 += 65535) {
    const int64_t nj = (n - j0 < 65535) ? n - j0 : 65535;
    unsigned gx = (unsigned)((row + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_thin_rows, dim3(gx, (unsigned)nj), dim3(256), 0, ctx->stream, store, row, first + j0 * every, every, nj, out + j0 * row);
  }
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"

// the column moments behind omc_store_moments (omc_scalar.hip keeps the entry point)
omc_status omc_col_moments(omc_ctx* ctx, const double* data, int64_t R, int64_t K, double* mean_out, double* var_out) {
  const unsigned tiles = (unsigned)((K + Q_COLS - 1) / Q_COLS);
  int64_t slices = (2048 + tiles - 1) / tiles;
  if (slices > (R + 255) / 256) slices = (R + 255) / 256;
  if (slices < 1) slices = 1;
  if (slices > 1024) slices = 1024;
  const int64_t rpb = (R + slices - 1) / slices;
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->store_ws, &ctx->store_ws_bytes, (size_t)slices * 3 * K * sizeof(double));
  if (st != OMC_OK) return st;
  double* part = (double*)ctx->store_ws;
  hipLaunchKernelGGL(k_col_moments_part, dim3(tiles, (unsigned)slices), dim3(256), 0, ctx->stream, data, R, K, rpb, part);
  hipLaunchKernelGGL(k_col_moments_join, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ctx->stream, K, (int)slices, part, mean_out, var_out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}
