// Banded precisions of any bandwidth w (RW2 / seasonal / lattice GMRFs):
// gmrf.sample_normal_canonical (gmrf.py:167-198) with Q_c = sum_k s_k[c] M_k, every M_k symmetric with
// bandwidth <= w, factorised in NATURAL order like the reference's unpermuted SuperLU / LAPACK route
// (gmrf.py:489-520), so that the draw matches the reference path-wise for the same z.
//
// One workgroup per chain (16 x 16 threads; a single 8 x 8 wave for bandwidth <= 7).  Right-looking banded Cholesky on a ring of the w+1 "open"
// columns held in LDS: column j is scaled by 1/sqrt(pivot), written to the per-chain factor workspace
// (column-major band: column j = w+1 contiguous doubles), used to update the (w x w)/2 trailing entries of
// the ring (threads tile the (a, b) square, no integer division), and its ring slot is reloaded with column
// j+w+1.  The forward substitution rides along (the right-hand side is one more ring).  The backward pass
// solves L'x = u + z (mean and draw share one pass: x = L^-T L^-1 b + L^-T z) with the dot product over the
// band spread over the lanes.  Two barriers per column; the work per column is O(w^2), so the cost is the
// band flops n w^2 for wide bands and barrier latency for narrow ones (tridiagonal models take the segmented
// kernel of omc_tridiag.hip instead).
#include <math.h>

#include "omc_common.h"

omc_status omc_ensure_bytes(omc_ctx* ctx, void** buf, size_t* have, size_t need);  // omc_dense.hip
bool omc_band_blocked_launch(omc_ctx* ctx, int64_t n, int w, const void* terms, const double* rhs_chain, int64_t ld_rhs,
                             const double* z_inject, int64_t ld_z, omc_rng_key key, double* Lws, double* x, int64_t ld_x, double* mean,
                             int64_t ld_mean, double* logdet);  // omc_bandwide.hip

#define BAND_WMAX 128

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the vector-memory
// counter, which would expose the latency of the prefetched global loads (and of the factor stores) on every
// column; inside the two loops the threads communicate through LDS alone.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct BandTermsDev {
  int n_terms;
  const double* band[OMC_MAX_TERMS];  // [ (bw+1) x n ], band[d*n + i] = M[i+d, i]; NULL = identity
  int bw[OMC_MAX_TERMS];
  const double* rhs[OMC_MAX_TERMS];
  const double* scale[OMC_MAX_TERMS];
};

__device__ __forceinline__ double band_entry(const BandTermsDev& T, const double* s, int64_t n, int64_t col, int d) {
  // Q[col + d, col]
  if (col + d >= n) return 0.0;
  double v = 0.0;
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    if (k < T.n_terms) {
      if (T.band[k]) {
        if (d <= T.bw[k]) v = fma(s[k], T.band[k][(int64_t)d * n + col], v);
      } else if (d == 0) {
        v += s[k];
      }
    }
  }
  return v;
}

__device__ __forceinline__ double band_rhs(const BandTermsDev& T, const double* s, int64_t n, int64_t col, const double* rc) {
  if (col >= n) return 0.0;
  double b = rc ? rc[col] : 0.0;
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k)
    if (k < T.n_terms && T.rhs[k]) b = fma(s[k], T.rhs[k][col], b);
  return b;
}

template <int BAND_TX, int BAND_TY>
__global__ void __launch_bounds__(BAND_TX* BAND_TY) k_band_sample(int64_t C, int64_t chain_offset, int64_t n, int w,
                                                                   BandTermsDev T, const double* rhs_chain, int64_t ld_rhs,
                                                                   const double* z_in, int64_t ld_z, omc_rng_key key,
                                                                   double* Lws, double* x, int64_t ld_x, double* mean,
                                                                   int64_t ld_mean, double* logdet, long long* bad) {
  extern __shared__ double sm[];
  const int W1 = w + 1;
  double* ring = sm;                    // W1 x W1: ring[(col % W1) * W1 + d] = open entry Q[col + d, col]
  double* rring = ring + (int64_t)W1 * W1;  // W1: open right-hand side
  double* lcol = rring + W1;            // W1: the column being eliminated
  double* misc = lcol + W1;             // [0] u_j, [1] fail flag
  const int64_t c = blockIdx.x;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int tid = ty * BAND_TX + tx;
  const int nthreads = BAND_TX * BAND_TY;
  double s[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) s[k] = (k < T.n_terms && T.scale[k]) ? T.scale[k][c] : 1.0;
  double* Lc = Lws + c * n * W1;
  double* xc = x + c * ld_x;

  // open the first w+1 columns
  for (int t = tid; t < W1 * W1; t += nthreads) {
    const int col = t / W1, d = t % W1;
    ring[col * W1 + d] = band_entry(T, s, n, col, d);
  }
  const double* rc = rhs_chain ? rhs_chain + c * ld_rhs : nullptr;
  for (int t = tid; t < W1; t += nthreads) rring[t] = band_rhs(T, s, n, t, rc);
  if (tid == 0) misc[1] = 0.0;
  __syncthreads();

  double ld_mant = 1.0;
  long long ld_exp = 0;
  // entries of the column that will be opened at the END of step j (column j + w + 1) are fetched one step ahead,
  // so the global-load latency hides behind a whole elimination step instead of stalling every barrier
  double pre = 0.0;
  if (tid < W1) pre = band_entry(T, s, n, (int64_t)W1, tid);
  else if (tid == W1) pre = band_rhs(T, s, n, (int64_t)W1, rc);
  int slot = 0;  // j % W1, kept incrementally (a 64-bit modulo per access would dominate the step)
  for (int64_t j = 0; j < n; ++j, slot = (slot + 1 == W1) ? 0 : slot + 1) {
    double pre_next = 0.0;
    if (tid < W1) pre_next = band_entry(T, s, n, j + 1 + W1, tid);
    else if (tid == W1) pre_next = band_rhs(T, s, n, j + 1 + W1, rc);
    const double pivot = ring[slot * W1];
    const bool ok = pivot > 0.0;
    // 1/sqrt(pivot) by rsq + two Newton steps; the column is scaled by multiplication (fp64 sqrt and divisions
    // on the serial path cost more than the whole trailing update for narrow bands)
    double rinv = 1.0;
    if (ok) {
      const double g = __builtin_amdgcn_rsq(pivot);
      const double h = 0.5 * g;
      double sq = pivot * g;                 // ~ sqrt(pivot)
      double e = fma(-sq, sq, pivot);
      sq = fma(e, h, sq);
      e = fma(-sq, sq, pivot);
      sq = fma(e, h, sq);
      rinv = omc_rcp_nr(sq);
    }
    if (tid < W1) {
      // the diagonal slot of the stored factor holds 1/L_jj (what the backward pass multiplies by)
      const double l = (tid == 0) ? rinv : ring[slot * W1 + tid] * rinv;
      lcol[tid] = l;
      Lc[j * W1 + tid] = l;
    }
    if (tid == 0) {
      const double u = rring[slot] * rinv;
      misc[0] = u;
      xc[j] = u;  // forward-substituted right-hand side, overwritten by the draw in the backward pass
      if (!ok) misc[1] = 1.0;
      // log det = sum log pivot, accumulated as a product with the exponent split off (one log at the end)
      ld_mant *= __builtin_amdgcn_frexp_mant(ok ? pivot : 1.0);
      ld_exp += __builtin_amdgcn_frexp_exp(ok ? pivot : 1.0);
      ld_exp += __builtin_amdgcn_frexp_exp(ld_mant);
      ld_mant = __builtin_amdgcn_frexp_mant(ld_mant);
    }
    lds_barrier();
    // trailing update: Q[j+a, j+b] -= l_a l_b, 1 <= b <= a <= w, and the right-hand side
    for (int a = 1 + ty; a <= w; a += BAND_TY) {
      const double la = lcol[a];
      for (int b = 1 + tx; b <= a; b += BAND_TX) {
        int cslot = slot + b;
        if (cslot >= W1) cslot -= W1;
        ring[cslot * W1 + (a - b)] = fma(-la, lcol[b], ring[cslot * W1 + (a - b)]);
      }
    }
    if (tid >= 1 && tid <= w) {
      int rslot = slot + tid;
      if (rslot >= W1) rslot -= W1;
      rring[rslot] = fma(-lcol[tid], misc[0], rring[rslot]);
    }
    // the eliminated column's slot takes column j + w + 1 (fetched during the previous step)
    if (tid < W1) ring[slot * W1 + tid] = pre;
    else if (tid == W1) rring[slot] = pre;
    pre = pre_next;
    lds_barrier();
  }
  const bool failed = misc[1] != 0.0;
  if (tid == 0) {
    if (logdet) logdet[c] = log(ld_mant) + (double)ld_exp * 0.69314718055994530942;
    if (failed) atomicMin((unsigned long long*)bad, (unsigned long long)c);
  }
  __syncthreads();
  if (failed) {
    for (int64_t i = tid; i < n; i += nthreads) xc[i] = NAN;
    return;
  }

  // t = u + z (all threads), mean needs u alone: keep it in the mean buffer
  double* mc = mean ? mean + c * ld_mean : nullptr;
  for (int64_t i = tid; i < n; i += nthreads) {
    double z;
    if (z_in) {
      z = z_in[c * ld_z + i];
    } else {
      double n0, n1;
      omc_normal_pair(omc_rng_block(key, chain_offset + c, (uint32_t)(i >> 1)), n0, n1);
      z = (i & 1) ? n1 : n0;
    }
    const double u = xc[i];
    if (mc) mc[i] = u;
    xc[i] = u + z;
  }
  __syncthreads();

  // backward pass  L' x = t:  x_j = (t_j - sum_{d=1..w} L[j+d, j] x_{j+d}) / L_jj ;  ring of the last w solutions
  double* xr = ring;        // W1 entries: x_{col} at xr[col % W1]
  double* mr = ring + W1;   // the same for the mean
  double* red = ring + 2 * W1;  // per-wave partial sums (2 x 4)
  for (int t = tid; t < 2 * W1; t += nthreads) ring[t] = 0.0;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int n_waves = (w + 63) / 64;  // waves that carry band entries (w <= 128 -> at most 2)
  const int d = tid + 1;  // band offset handled by this thread
  // the factor column and the right-hand side of step j - 1 are fetched while step j is reduced
  double l_cur = (d <= w) ? Lc[(n - 1) * W1 + d] : 0.0;
  double ljj_cur = 0.0, t_cur = 0.0, tm_cur = 0.0;
  if (tid == 0) {
    ljj_cur = Lc[(n - 1) * W1];
    t_cur = xc[n - 1];
    if (mc) tm_cur = mc[n - 1];
  }
  int bslot = (int)((n - 1) % W1);  // j % W1, kept incrementally
  for (int64_t j = n - 1; j >= 0; --j, bslot = (bslot == 0) ? W1 - 1 : bslot - 1) {
    double l_next = 0.0, ljj_next = 0.0, t_next = 0.0, tm_next = 0.0;
    if (j > 0) {
      if (d <= w) l_next = Lc[(j - 1) * W1 + d];
      if (tid == 0) {
        ljj_next = Lc[(j - 1) * W1];
        t_next = xc[j - 1];
        if (mc) tm_next = mc[j - 1];
      }
    }
    double px = 0.0, pm = 0.0;
    if (d <= w && j + d < n) {
      int sl = bslot + d;
      if (sl >= W1) sl -= W1;
      px = l_cur * xr[sl];
      if (mc) pm = l_cur * mr[sl];
    }
    if (wave < n_waves || wave == 0) {
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1) {
        px += __shfl_xor(px, sh, 64);
        pm += __shfl_xor(pm, sh, 64);
      }
      if (lane == 0) { red[wave] = px; red[4 + wave] = pm; }
    }
    lds_barrier();
    if (tid == 0) {
      double sx = red[0], smn = red[4];
      if (n_waves > 1) { sx += red[1]; smn += red[5]; }
      const double xv = (t_cur - sx) * ljj_cur;  // ljj_cur holds 1/L_jj
      xc[j] = xv;
      xr[bslot] = xv;
      if (mc) {
        const double mv = (tm_cur - smn) * ljj_cur;
        mc[j] = mv;
        mr[bslot] = mv;
      }
    }
    l_cur = l_next; ljj_cur = ljj_next; t_cur = t_next; tm_cur = tm_next;
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// Narrow bands (w <= 8): ONE LANE PER CHAIN.  The (w+1) x (w+1) window of open columns lives in registers (static
// indices, shifted by one column per step), so a column costs a few dozen register-only instructions instead of two
// workgroup barriers and a dozen LDS round trips; 64 chains share a wave and every global access is coalesced
// across them: the factor and the forward-substituted right-hand side go to a workspace laid out [column][entry][chain],
// the draw is handed back chain-major through a 64 x 64 LDS transpose tile.  The kernel is latency-bound (one
// dependent chain of n columns per lane), so the loop bodies are kept short: ONE banded term (the prior; the other
// terms are identities, the usual likelihoods) and running pointers instead of index arithmetic.
struct BandLaneArgs {
  const double* band;      // [(bw+1) x n] of the banded term
  int bw;
  const double* s_band;    // [C] or NULL
  const double* s_ident[OMC_MAX_TERMS - 1];  // scales of the identity terms ([C] or NULL = 1); n_ident of them
  int n_ident;
  const double* rhs[OMC_MAX_TERMS];          // shared right-hand sides with their scales
  const double* s_rhs[OMC_MAX_TERMS];
  int n_rhs;
  const double* rhs_t;     // per-chain right-hand side, transposed: element (column i, chain c) at rhs_t[i * ld_t + c], or NULL
  int64_t ld_t;            // (a lane beyond the last chain repeats the last chain's work, right-hand side included: the
                           //  duplicates write the same values to the same workspace slots)
};

// rhs_chain [C][ld] -> [n][ld_t] (chains contiguous: what a lane-per-chain kernel reads coalesced), 64 x 64 tiles through LDS
__global__ void __launch_bounds__(256) k_band_rhs_transpose(int64_t C, int64_t n, const double* __restrict__ src, int64_t ld,
                                                            double* __restrict__ dst, int64_t ld_t) {
  __shared__ double tile[64][65];
  const int64_t i0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int64_t c = c0 + r, i = i0 + tx;
    tile[r][tx] = (c < C && i < n) ? src[c * ld + i] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int64_t i = i0 + r;
    if (i < n) dst[i * ld_t + c0 + tx] = tile[tx][r];
  }
}

template <int W>
__global__ void __launch_bounds__(64) k_band_lane(int64_t C, int64_t chain_offset, int64_t n, BandLaneArgs P,
                                                  const double* z_in, int64_t ld_z, omc_rng_key key, double* Lws,
                                                  double* x, int64_t ld_x, double* mean, int64_t ld_mean, double* logdet,
                                                  long long* bad, const int* group_flag) {
  constexpr int W1 = W + 1;
  if (group_flag && !group_flag[blockIdx.x]) return;  // fallback use: only the groups the segmented route gave up on
  __shared__ double tile_x[64][65];
  __shared__ double tile_m[64][65];
  const int lane = threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int64_t c = c0 + lane;
  const bool live = c < C;
  const int64_t cc = live ? c : C - 1;
  const double sb = P.s_band ? P.s_band[cc] : 1.0;
  double sid = 0.0;
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS - 1; ++k)
    if (k < P.n_ident) sid += P.s_ident[k] ? P.s_ident[k][cc] : 1.0;
  double sr[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) sr[k] = (k < P.n_rhs && P.s_rhs[k]) ? P.s_rhs[k][cc] : 1.0;
  // workspace: L[(j*W1 + d)*C + c] (d = 0 holds 1/L_jj), then u[j*C + c]
  double* Lp = Lws + cc;
  double* Up = Lws + (int64_t)n * W1 * C + cc;
  // band rows are zero-padded to n entries, so Q[col + d, col] = sb * band[d][col] (+ sid on the diagonal) for col < n
  const double* brow[W1];
#pragma unroll
  for (int d = 0; d < W1; ++d) brow[d] = (d <= P.bw) ? P.band + (int64_t)d * n : nullptr;
  auto column = [&](int64_t col, double (&q)[W1], double& r) {
    if (col < n) {
#pragma unroll
      for (int d = 0; d < W1; ++d) q[d] = brow[d] ? sb * brow[d][col] : 0.0;
      q[0] += sid;
      double b = P.rhs_t ? P.rhs_t[col * P.ld_t + cc] : 0.0;
#pragma unroll
      for (int k = 0; k < OMC_MAX_TERMS; ++k)
        if (k < P.n_rhs) b = fma(sr[k], P.rhs[k][col], b);
      r = b;
    } else {
#pragma unroll
      for (int d = 0; d < W1; ++d) q[d] = 0.0;
      r = 0.0;
    }
  };
  double A[W1][W1];  // A[b][d] = current value of Q[j+b+d, j+b]
  double R[W1];      // current right-hand side at j+b
#pragma unroll
  for (int b = 0; b < W1; ++b) column((int64_t)b, A[b], R[b]);
  double ld_mant = 1.0;
  long long ld_exp = 0;
  bool fail = false;
  // the column that enters the window at the end of step j is fetched during step j - 1
  double pre[W1], preR;
  column((int64_t)W1, pre, preR);
  for (int64_t j = 0; j < n; ++j) {
    double pre_next[W1], preR_next;
    column(j + 1 + W1, pre_next, preR_next);
    const double pivot = A[0][0];
    const bool ok = pivot > 0.0;
    fail |= !ok;
    double rinv = 1.0;
    if (ok) {
      const double g = __builtin_amdgcn_rsq(pivot);
      const double h = 0.5 * g;
      double sq = pivot * g;
      double e = fma(-sq, sq, pivot);
      sq = fma(e, h, sq);
      e = fma(-sq, sq, pivot);
      sq = fma(e, h, sq);
      rinv = omc_rcp_nr(sq);
    }
    double l[W1];
    l[0] = rinv;
#pragma unroll
    for (int d = 1; d < W1; ++d) l[d] = A[0][d] * rinv;
    const double u = R[0] * rinv;
#pragma unroll
    for (int d = 0; d < W1; ++d) Lp[(int64_t)d * C] = l[d];
    Lp += (int64_t)W1 * C;
    *Up = u;
    Up += C;
    ld_mant *= __builtin_amdgcn_frexp_mant(ok ? pivot : 1.0);
    ld_exp += __builtin_amdgcn_frexp_exp(ok ? pivot : 1.0);
    if ((j & 15) == 15) {
      ld_exp += __builtin_amdgcn_frexp_exp(ld_mant);
      ld_mant = __builtin_amdgcn_frexp_mant(ld_mant);
    }
    // trailing update and shift of the window by one column
#pragma unroll
    for (int b = 1; b < W1; ++b) {
#pragma unroll
      for (int d = 0; d < W1; ++d) {
        double v = A[b][d];
        if (b + d < W1) v = fma(-l[b + d], l[b], v);  // Q[j+a, j+b] -= l_a l_b with a = b + d
        A[b - 1][d] = v;
      }
      R[b - 1] = fma(-l[b], u, R[b]);
    }
#pragma unroll
    for (int d = 0; d < W1; ++d) { A[W][d] = pre[d]; pre[d] = pre_next[d]; }
    R[W] = preR;
    preR = preR_next;
  }
  if (logdet && live) logdet[c] = log(ld_mant) + (double)ld_exp * 0.69314718055994530942;
  if (fail && live) atomicMin((unsigned long long*)bad, (unsigned long long)c);

  // backward pass  L' x = u + z (and L' m = u), 64 columns at a time into the transpose tiles; Lp / Up now point
  // one column past the end and walk back
  const int64_t gc = chain_offset + cc;
  const double* zrow = z_in ? z_in + cc * ld_z : nullptr;
  double z_even = 0.0;
  double xs[W1], ms[W1];  // xs[d] = x_{j+d} for d = 1..W (xs[0] unused)
#pragma unroll
  for (int d = 0; d < W1; ++d) { xs[d] = 0.0; ms[d] = 0.0; }
  double lnext[W1], unext;
  Lp -= (int64_t)W1 * C;
  Up -= C;
#pragma unroll
  for (int d = 0; d < W1; ++d) lnext[d] = Lp[(int64_t)d * C];
  unext = *Up;
  for (int64_t jb = ((n - 1) / 64) * 64; jb >= 0; jb -= 64) {
    const int len = (int)((n - jb < 64) ? n - jb : 64);
    for (int t = len - 1; t >= 0; --t) {
      const int64_t j = jb + t;
      double lcol[W1];
#pragma unroll
      for (int d = 0; d < W1; ++d) lcol[d] = lnext[d];
      const double u = unext;
      if (j > 0) {
        Lp -= (int64_t)W1 * C;
        Up -= C;
#pragma unroll
        for (int d = 0; d < W1; ++d) lnext[d] = Lp[(int64_t)d * C];
        unext = *Up;
      }
      double z;
      if (zrow) {
        z = zrow[j];
      } else if ((j & 1) || j == n - 1) {  // a Philox block gives the draws of columns 2q and 2q+1: made once, used twice
        double n0, n1;
        omc_normal_pair(omc_rng_block(key, gc, (uint32_t)(j >> 1)), n0, n1);
        z = (j & 1) ? n1 : n0;
        z_even = n0;
      } else {
        z = z_even;
      }
      double ax = u + z, am = u;
#pragma unroll
      for (int d = 1; d < W1; ++d) {
        ax = fma(-lcol[d], xs[d], ax);
        am = fma(-lcol[d], ms[d], am);
      }
      const double xv = fail ? NAN : ax * lcol[0], mv = am * lcol[0];
#pragma unroll
      for (int d = W; d > 1; --d) { xs[d] = xs[d - 1]; ms[d] = ms[d - 1]; }
      xs[1] = xv; ms[1] = mv;
      tile_x[lane][t] = xv;
      if (mean) tile_m[lane][t] = mv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int r = 0; r < 64; ++r) {
      if (c0 + r < C && lane < len) {
        x[(c0 + r) * ld_x + jb + lane] = tile_x[r][lane];
        if (mean) mean[(c0 + r) * ld_mean + jb + lane] = tile_m[r][lane];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ------------------------------------------------------------------------------------------------
// The same lane-per-chain factorisation with the n columns of a chain cut into SEGMENTS, one wave per (64 chains,
// segment), each wave its own workgroup so that the segments of a chain group spread over the CUs: k_band_lane is one
// dependent chain of 2 n steps per lane with nothing to overlap it -- 12.6 ms for n = 10 000 whatever the number of
// chains.  (Sixteen segment waves inside ONE workgroup were tried first: they share a CU, and at ~100 instructions a
// step the four waves of a SIMD take 0.7 us per step between them -- 3 ms.)  What lets a segment start without its
// predecessors:
//   * pivots (nonlinear): the window of partially eliminated columns forgets its starting value geometrically (the
//     decay of the precision's Green's function), so a segment starts `ov` columns early from the raw matrix entries and
//     throws the warm-up away.  That this was enough is CHECKED: the window a segment ends with must agree with the one
//     its successor started from to `tol` relative to the pivot; if any join of any of a group's 64 chains fails (a very
//     weak likelihood: long memory) the group is redone in one piece by k_band_lane.  No silent loss of accuracy.
//   * forward and backward substitution (linear): exact.  A segment is run from a zero incoming state together with W
//     unit incoming states (its affine map: W x W matrix + W vector); the maps are composed in order -- in two levels,
//     blocks of BSEG_BS segments whose map the last wave of the block to finish a phase leaves behind -- and the
//     segment's true solution follows from its true incoming state.
// Three launches, the launch boundaries being the only synchronisation between phases (inside a phase: one counter per
// block, never waited for):
//   PHASE 0  factor + forward map:  L, the zero-state u and its W unit responses, join windows, log-det parts; per
//            block: forward map, joins inside the block, log-det part
//   PHASE 1  (first segment's wave: joins between blocks, the group's verdict, log det;) incoming forward state, backward
//            map of the segment (draws made, nothing stored); per block: backward map
//   PHASE 2  incoming backward state, the segment's x (draws made again) and mean, written to the caller's chain-major
//            rows through an LDS tile
// No loop both loads and stores what it waits for: on this hardware a load returns behind every older store (one
// in-order counter), which made a combined loop pay a store acknowledgement per step.  Natural-order Cholesky
// throughout: the factor, u = L^-1 b and x = L^-T (u + z) are k_band_lane's to rounding (the updates of a column are
// added in another order, the incoming states through block maps).  The number of segments: see
// omc_band_sample_canonical (chosen for the SIMDs).
#define BSEG_MAX 128
#define BSEG_BS 8      // segments per block (two-level composition of the incoming states)
#define BSEG_BROW 32   // scratch entries per block: forward map (W*W + W), backward map (W*W + 2 W)
#define BSEG_RC 32  // columns per staged piece of a per-chain right-hand side
template <int W, int PHASE>
__global__ void __launch_bounds__(64) k_band_seg(int64_t C, int64_t chain_offset, int64_t n, BandLaneArgs P, int nseg,
                                                 int64_t mseg, int ov, double tol, const double* z_in, int64_t ld_z,
                                                 omc_rng_key key, double* Lws, double* x_out, int64_t ld_x, double* mean_out,
                                                 int64_t ld_mean, double* scratch,
                                                 const int* gate, int* group_flag, double* logdet, long long* bad,
                                                 unsigned long long* n_fallback, int* blk_count) {
  constexpr int W1 = W + 1;
  constexpr int NJ = W * (W + 1) / 2;  // window entries that carry eliminated columns' updates: A[b][d] with b + d < W
  constexpr int NF = W * W + W;        // forward map: G (W x W) and g (W)
  constexpr int NB = W * W + 2 * W;    // backward map: H, h (draw) and hm (mean)
  // per (group, segment): start window, end window, G/g, H/h/hm, log-det part, fail; then what PHASE 1 leaves for PHASE 2:
  // the segment's true incoming forward state and (first segment's row only) whether the chain is positive definite
  constexpr int NS = 2 * NJ + NF + NB + 2 + W + 1;
  constexpr int O_JS = 0, O_JE = NJ, O_F = 2 * NJ, O_B = 2 * NJ + NF, O_LD = 2 * NJ + NF + NB, O_FAIL = O_LD + 1;
  constexpr int O_DIN = O_FAIL + 1, O_CF = O_DIN + W;
  constexpr int O_BV = 2 * W * W + 3 * W;  // block rows: forward map, backward map, then joins ok / failed / log-det part
  static_assert(O_BV + 3 <= BSEG_BROW, "block scratch row");
  static_assert(NS <= 48, "scratch rows are allocated 48 entries long");
  __shared__ double stage_all[2 * (W1 + OMC_MAX_TERMS) * 64];
  __shared__ double rc_tile[PHASE == 0 ? BSEG_RC : 1][64];  // per-chain right-hand side of the current BSEG_RC columns
  // PHASE 2: the results of 32 columns x 64 chains on their way to the caller's chain-major rows (256 B of a row at a time)
  __shared__ double out_x[PHASE == 2 ? 32 : 1][65], out_m[PHASE == 2 ? 32 : 1][65];
  const int lane = threadIdx.x, seg = blockIdx.x;
  const int64_t grp = blockIdx.y;
  // second attempt with a longer warm-up: only the groups whose joins did not close the first time (gate is what the
  // first attempt's PHASE 1 left; this attempt's PHASE 1 writes its verdict elsewhere, so the gate is stable for a launch)
  if (PHASE != 2 && gate && !gate[grp]) return;
  const int64_t c0 = grp * 64;
  const int64_t c = c0 + lane;
  const bool live = c < C;
  const int64_t cc = live ? c : C - 1;
  // Segments in blocks of BSEG_BS: the LAST wave of a block to finish a phase composes the block's map from its segments'
  // maps (one batch of loads) and leaves it behind the segments' rows; a wave of the next phase then walks over whole
  // blocks and over the segments of its own block only.  Every wave walking over every segment before (or behind) it read
  // O(segments^2) rows of this scratch -- 100 MB per phase at 64 segments and 1024 chains, a quarter of the factor's own
  // bytes, out of L2/MALL, and the longest walk set the phase's length.
  const int nblk = (nseg + BSEG_BS - 1) / BSEG_BS;
  double* sc = scratch + grp * ((int64_t)nseg * NS + (int64_t)nblk * BSEG_BROW) * 64;  // [seg][entry][lane], then [block][entry][lane]
  double* bsc = sc + (int64_t)nseg * NS * 64;
  auto bput = [&](int bk, int e, double v) { bsc[((int64_t)bk * BSEG_BROW + e) * 64 + lane] = v; };
  auto bget = [&](int bk, int e) -> double { return bsc[((int64_t)bk * BSEG_BROW + e) * 64 + lane]; };
  // The rows the last wave of a block reads from its fellows go out as agent-scope (write-through) stores and come in as
  // agent-scope loads: a release FENCE at agent scope writes the whole L2 back -- the factor's half gigabyte sits there
  // dirty -- and 1024 waves doing that at their ends cost 40-70 us per phase (measured, twice).
  auto sput_ag = [&](int sg, int e, double v) {
    __hip_atomic_store((unsigned long long*)&sc[((int64_t)sg * NS + e) * 64 + lane], (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto sget_ag = [&](int sg, int e) -> double {
    return __longlong_as_double((long long)__hip_atomic_load((unsigned long long*)&sc[((int64_t)sg * NS + e) * 64 + lane],
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  };
  // has every other wave of this wave's block finished the phase?  (counters zeroed per launch by the host)
  // Memory-model note: the rows travel as RELAXED agent-scope stores, ordered in front of the RELAXED counter increment by
  // `s_waitcnt vmcnt(0)` (a write-through store has reached the L2 -- the agent's point of coherence -- when vmcnt counts it
  // down), and the reader's agent-scope loads bypass its own L1.  That is a property of gfx942/gfx950's memory pipeline, not a
  // happens-before edge of the HIP/LLVM memory model; the formal route (release fence / acquire at AGENT scope) costs a
  // write-back of the whole dirty L2 per wave (see above).  Hence the architecture check below, and
  // benchmarks/determinism_all.py band as the gating run after any toolchain bump.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_band_seg's block hand-over relies on gfx950's write-through stores + in-order vmcnt; use agent-scope release/acquire on the counter elsewhere"
#endif
  auto last_of_block = [&](int bk) -> bool {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's write-through rows are at their destination
    const int members = (bk + 1) * BSEG_BS <= nseg ? BSEG_BS : nseg - bk * BSEG_BS;
    int last_one = 0;
    if (lane == 0)
      last_one = __hip_atomic_fetch_add(&blk_count[grp * nblk + bk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1;
    if (!__builtin_amdgcn_readfirstlane(last_one)) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
  };
  auto sput = [&](int sg, int e, double v) { sc[((int64_t)sg * NS + e) * 64 + lane] = v; };
  auto sget = [&](int sg, int e) -> double { return sc[((int64_t)sg * NS + e) * 64 + lane]; };
  const double sb = P.s_band ? P.s_band[cc] : 1.0;
  double sid = 0.0;
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS - 1; ++k)
    if (k < P.n_ident) sid += P.s_ident[k] ? P.s_ident[k][cc] : 1.0;
  double sr[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) sr[k] = (k < P.n_rhs && P.s_rhs[k]) ? P.s_rhs[k][cc] : 1.0;
  // workspace: L[(j*W1 + d)*C + c] (d = 0 holds 1/L_jj), then per column the zero-state u and its W unit responses
  double* const L0 = Lws + cc;
  double* const U0 = Lws + (int64_t)n * W1 * C + cc;   // U[(j*W1 + k)*C + c]: k = 0 zero-state u, k = 1..W responses
  const double* brow[W1];
#pragma unroll
  for (int d = 0; d < W1; ++d) brow[d] = (d <= P.bw) ? P.band + (int64_t)d * n : nullptr;
  const int64_t lo = (int64_t)seg * mseg, hi = (lo + mseg < n) ? lo + mseg : n;

  if constexpr (PHASE == 0) {
    // Shared (chain-independent) entries of a column, staged 64 columns at a time: lane t fetches column 64 k + t of every
    // array (coalesced) a whole chunk ahead, the chunk is handed to LDS when the recurrence reaches it, and a step reads
    // its column as broadcasts.
    const int NV = W1 + P.n_rhs;
    double* const stage = stage_all;
    double sreg[W1 + OMC_MAX_TERMS];
    int64_t staged = -1, inreg = -1;
    auto fetch_chunk = [&](int64_t ch) {
      const int64_t col = ch * 64 + lane;
#pragma unroll
      for (int d = 0; d < W1; ++d) sreg[d] = (brow[d] && col < n) ? brow[d][col] : 0.0;
#pragma unroll
      for (int k2 = 0; k2 < OMC_MAX_TERMS; ++k2) sreg[W1 + k2] = (k2 < P.n_rhs && col < n) ? P.rhs[k2][col] : 0.0;
      inreg = ch;
    };
    auto need_chunk = [&](int64_t ch) {  // uniform over the wave
      if (ch <= staged) return;
      if (inreg != ch) fetch_chunk(ch);
      double* dst = stage + (ch & 1) * NV * 64;
#pragma unroll
      for (int d = 0; d < W1; ++d) dst[d * 64 + lane] = sreg[d];
#pragma unroll
      for (int k2 = 0; k2 < OMC_MAX_TERMS; ++k2)
        if (k2 < P.n_rhs) dst[(W1 + k2) * 64 + lane] = sreg[W1 + k2];
      staged = ch;
      fetch_chunk(ch + 1);
    };
    // A per-chain right-hand side (transposed: a lane's value of a column is one coalesced load) is staged the same way,
    // BSEG_RC columns at a time: the piece after the current one waits in registers (fetched a whole piece ahead of its
    // use) and goes to this wave's LDS tile when the recurrence reaches it.
    double rcreg[BSEG_RC];
    int64_t rc_staged = -1, rc_inreg = -1;
    auto rc_fetch = [&](int64_t pc) {
#pragma unroll
      for (int t = 0; t < BSEG_RC; ++t) {
        const int64_t col = pc * BSEG_RC + t;
        rcreg[t] = (col < n) ? P.rhs_t[col * P.ld_t + cc] : 0.0;
      }
      rc_inreg = pc;
    };
    auto rc_need = [&](int64_t pc) {  // uniform over the wave; columns are asked for in ascending order
      if (pc <= rc_staged) return;
      if (rc_inreg != pc) rc_fetch(pc);
#pragma unroll
      for (int t = 0; t < BSEG_RC; ++t) rc_tile[t][lane] = rcreg[t];
      rc_staged = pc;
      rc_fetch(pc + 1);
    };
    auto column = [&](int64_t col, double (&q)[W1], double& r) {
      if (col < n) {
        need_chunk(col >> 6);
        const double* src = stage + ((col >> 6) & 1) * NV * 64 + (col & 63);
#pragma unroll
        for (int d = 0; d < W1; ++d) q[d] = sb * src[d * 64];
        q[0] += sid;
        double b = 0.0;
        if (P.rhs_t) {
          rc_need(col / BSEG_RC);
          b = rc_tile[col % BSEG_RC][lane];
        }
#pragma unroll
        for (int k2 = 0; k2 < OMC_MAX_TERMS; ++k2)
          if (k2 < P.n_rhs) b = fma(sr[k2], src[(W1 + k2) * 64], b);
        r = b;
      } else {
#pragma unroll
        for (int d = 0; d < W1; ++d) q[d] = 0.0;
        r = 0.0;
      }
    };
    const int64_t j0 = (seg == 0 || lo < ov) ? 0 : lo - ov;
    bool fail = false;
    double A[W1][W1], raw[W1];
    need_chunk(j0 >> 6);
#pragma unroll
    for (int b = 0; b < W1; ++b) column(j0 + b, A[b], raw[b]);
    double pre[W1], preR;
    column(j0 + W1, pre, preR);
    double Rc[W1], E[W][W1];
#pragma unroll
    for (int b = 0; b < W1; ++b) Rc[b] = 0.0;
#pragma unroll
    for (int k = 0; k < W; ++k)
#pragma unroll
      for (int b = 0; b < W1; ++b) E[k][b] = (b == k) ? 1.0 : 0.0;
    double ld_mant = 1.0;
    long long ld_exp = 0;
    double* Lp = L0 + lo * (int64_t)W1 * C;
    double* Up = U0 + lo * (int64_t)W1 * C;
    for (int64_t j = j0; j < hi; ++j) {
      double pre_next[W1], preR_next;
      column(j + 1 + W1, pre_next, preR_next);
      const bool mine = j >= lo;
      if (j == lo) {  // what this segment starts from: checked against its predecessor's end
        int e = 0;
#pragma unroll
        for (int b = 0; b < W; ++b)
#pragma unroll
          for (int d = 0; d < W; ++d)
            if (b + d < W) sput_ag(seg, O_JS + e++, A[b][d]);
      }
      const double pivot = A[0][0];
      const bool ok = pivot > 0.0;
      if (mine) fail |= !ok;
      double rinv = 1.0;
      if (ok) {
        const double g = __builtin_amdgcn_rsq(pivot);
        const double h = 0.5 * g;
        double sq = pivot * g;
        double e = fma(-sq, sq, pivot);
        sq = fma(e, h, sq);
        e = fma(-sq, sq, pivot);
        sq = fma(e, h, sq);
        rinv = omc_rcp_nr(sq);
      }
      double l[W1];
      l[0] = rinv;
#pragma unroll
      for (int d = 1; d < W1; ++d) l[d] = A[0][d] * rinv;
      double u = 0.0, uk[W];
#pragma unroll
      for (int k = 0; k < W; ++k) uk[k] = 0.0;
      if (mine) {
#pragma unroll
        for (int d = 0; d < W1; ++d) Lp[(int64_t)d * C] = l[d];
        Lp += (int64_t)W1 * C;
        u = (raw[0] + Rc[0]) * rinv;
        Up[0] = u;
#pragma unroll
        for (int k = 0; k < W; ++k) {
          uk[k] = E[k][0] * rinv;
          Up[(int64_t)(k + 1) * C] = uk[k];
        }
        Up += (int64_t)W1 * C;
        ld_mant *= __builtin_amdgcn_frexp_mant(ok ? pivot : 1.0);
        ld_exp += __builtin_amdgcn_frexp_exp(ok ? pivot : 1.0);
        if ((j & 15) == 15) {
          ld_exp += __builtin_amdgcn_frexp_exp(ld_mant);
          ld_mant = __builtin_amdgcn_frexp_mant(ld_mant);
        }
      }
#pragma unroll
      for (int b = 1; b < W1; ++b) {
#pragma unroll
        for (int d = 0; d < W1; ++d) {
          double v = A[b][d];
          if (b + d < W1) v = fma(-l[b + d], l[b], v);
          A[b - 1][d] = v;
        }
        raw[b - 1] = raw[b];
        if (mine) {
          Rc[b - 1] = fma(-l[b], u, Rc[b]);
#pragma unroll
          for (int k = 0; k < W; ++k) E[k][b - 1] = fma(-l[b], uk[k], E[k][b]);
        }
      }
#pragma unroll
      for (int d = 0; d < W1; ++d) { A[W][d] = pre[d]; pre[d] = pre_next[d]; }
      raw[W] = preR;
      preR = preR_next;
      if (mine) {
        Rc[W] = 0.0;
#pragma unroll
        for (int k = 0; k < W; ++k) E[k][W] = 0.0;
      }
    }
    int e = 0;
#pragma unroll
    for (int b = 0; b < W; ++b)
#pragma unroll
      for (int d = 0; d < W; ++d)
        if (b + d < W) sput_ag(seg, O_JE + e++, A[b][d]);
    e = 0;
#pragma unroll
    for (int k = 0; k < W; ++k)
#pragma unroll
      for (int b = 0; b < W; ++b) sput_ag(seg, O_F + e++, E[k][b]);  // d(out state b) / d(in state k)
#pragma unroll
    for (int b = 0; b < W; ++b) sput_ag(seg, O_F + e++, Rc[b]);
    sput_ag(seg, O_LD, log(ld_mant) + (double)ld_exp * 0.69314718055994530942);
    sput_ag(seg, O_FAIL, fail ? 1.0 : 0.0);
    {  // the block's forward map, by the last of its waves to get here: out[b] = g[b] + sum_k G[k][b] in[k]
      const int bk = seg / BSEG_BS;
      if (!last_of_block(bk)) return;
      double mp[BSEG_BS][NF];
#pragma unroll
      for (int i = 0; i < BSEG_BS; ++i) {
        const int sg = (bk * BSEG_BS + i < nseg) ? bk * BSEG_BS + i : nseg - 1;
#pragma unroll
        for (int e2 = 0; e2 < NF; ++e2) mp[i][e2] = sget_ag(sg, O_F + e2);
      }
      double G[W][W], g[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        g[k] = 0.0;
#pragma unroll
        for (int b = 0; b < W; ++b) G[k][b] = (k == b) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int i = 0; i < BSEG_BS; ++i) {
        if (bk * BSEG_BS + i < nseg) {
          double nG[W][W], ng[W];
#pragma unroll
          for (int b = 0; b < W; ++b) {
            double a = mp[i][W * W + b];
#pragma unroll
            for (int k = 0; k < W; ++k) a = fma(mp[i][k * W + b], g[k], a);
            ng[b] = a;
#pragma unroll
            for (int j = 0; j < W; ++j) {
              double t = 0.0;
#pragma unroll
              for (int k = 0; k < W; ++k) t = fma(mp[i][k * W + b], G[j][k], t);
              nG[j][b] = t;
            }
          }
#pragma unroll
          for (int b = 0; b < W; ++b) {
            g[b] = ng[b];
#pragma unroll
            for (int j = 0; j < W; ++j) G[j][b] = nG[j][b];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < W; ++j)
#pragma unroll
        for (int b = 0; b < W; ++b) bput(bk, j * W + b, G[j][b]);
#pragma unroll
      for (int b = 0; b < W; ++b) bput(bk, W * W + b, g[b]);
      // ... and what the group's verdict needs from this block: the joins inside it, any failed pivot, its log-det part
      bool okb = true, failb = false;
      double ldb = 0.0;
#pragma unroll
      for (int i = 0; i < BSEG_BS; ++i) {
        const int sg = bk * BSEG_BS + i;
        if (sg < nseg) {
          failb |= sget_ag(sg, O_FAIL) != 0.0;
          ldb += sget_ag(sg, O_LD);
          if (i + 1 < BSEG_BS && sg + 1 < nseg) {
            const double scale = fabs(sget_ag(sg, O_JE));
            bool good = true;
#pragma unroll
            for (int e2 = 0; e2 < NJ; ++e2) good &= fabs(sget_ag(sg + 1, O_JS + e2) - sget_ag(sg, O_JE + e2)) <= tol * scale;
            okb &= good;
          }
        }
      }
      bput(bk, O_BV, okb ? 1.0 : 0.0);
      bput(bk, O_BV + 1, failb ? 1.0 : 0.0);
      bput(bk, O_BV + 2, ldb);
    }
    return;
  } else {
    // ---- joins of the whole group: the first segment's wave of PHASE 1 looks at all of them and leaves the verdict (the
    // group goes to a longer warm-up, then to k_band_lane, as a whole).  The other waves of PHASE 1 do not wait for it:
    // what they compute for a group that is redone is simply not used, and PHASE 2 reads the verdict.  (Every wave of
    // both phases used to repeat this loop -- 8 loads a segment, a third of the phases' time between them.)
    // (no short-circuits and a fixed unroll: the loads of eight segments go out together -- one at a time this loop is
    // 64 round trips to memory, longer than the segment's own work)
    bool failed = false;
    double din[W];
#pragma unroll
    for (int b = 0; b < W; ++b) din[b] = 0.0;
    if (PHASE == 1) {
      if (seg == 0) {
        // (the joins inside a block, its failures and its log-det part were looked at by the block's last wave of PHASE 0;
        // left here: the joins between blocks)
        bool okj = true;
        double ld_sum = 0.0;
#pragma unroll 8
        for (int bk = 0; bk < nblk; ++bk) {
          okj &= bget(bk, O_BV) != 0.0;
          failed |= bget(bk, O_BV + 1) != 0.0;
          ld_sum += bget(bk, O_BV + 2);
          const int sg = (bk + 1) * BSEG_BS - 1;  // the block's last segment against the next block's first
          if (sg + 1 < nseg) {
            const double scale = fabs(sget(sg, O_JE));
            bool good = true;
#pragma unroll
            for (int e = 0; e < NJ; ++e) good &= fabs(sget(sg + 1, O_JS + e) - sget(sg, O_JE + e)) <= tol * scale;
            okj &= good;
          }
        }
        const bool chain_bad = live && !failed && !okj;  // (a chain that is not positive definite is reported, not retried)
        if (__builtin_amdgcn_readfirstlane((int)(__ballot(chain_bad) != 0ull))) {
          if (lane == 0) {
            group_flag[grp] = 1;
            atomicAdd(n_fallback, 1ull);  // first attempt: a retry; second attempt: a group handed to k_band_lane
          }
          return;
        }
        if (lane == 0) group_flag[grp] = 0;
        sput(0, O_CF, failed ? 1.0 : 0.0);
        if (live) {
          if (logdet) logdet[c] = ld_sum;
          if (failed) atomicMin((unsigned long long*)bad, (unsigned long long)c);
        }
      }
      // ---- the segment's true incoming forward state (kept for PHASE 2): whole blocks, then the segments of its own block
      const int myblk = seg / BSEG_BS;
#pragma unroll 8
      for (int bk = 0; bk < myblk; ++bk) {
        double nx[W];
#pragma unroll
        for (int b = 0; b < W; ++b) {
          double a = bget(bk, W * W + b);
#pragma unroll
          for (int k = 0; k < W; ++k) a = fma(bget(bk, k * W + b), din[k], a);
          nx[b] = a;
        }
#pragma unroll
        for (int b = 0; b < W; ++b) din[b] = nx[b];
      }
#pragma unroll 8
      for (int sg = myblk * BSEG_BS; sg < seg; ++sg) {
        double nx[W];
#pragma unroll
        for (int b = 0; b < W; ++b) {
          double a = sget(sg, O_F + W * W + b);
#pragma unroll
          for (int k = 0; k < W; ++k) a = fma(sget(sg, O_F + k * W + b), din[k], a);
          nx[b] = a;
        }
#pragma unroll
        for (int b = 0; b < W; ++b) din[b] = nx[b];
      }
#pragma unroll
      for (int b = 0; b < W; ++b) sput(seg, O_DIN + b, din[b]);
    } else {
      if (group_flag[grp]) return;
      failed = sget(0, O_CF) != 0.0;
#pragma unroll
      for (int b = 0; b < W; ++b) din[b] = sget(seg, O_DIN + b);
    }
    // ---- its incoming backward state (PHASE 2): composed from the last segment down
    const bool last = seg == nseg - 1;
    double xs[W1], ms[W1], Xk[W][W1];
#pragma unroll
    for (int d = 0; d < W1; ++d) { xs[d] = 0.0; ms[d] = 0.0; }
    if (PHASE == 2 && !last) {
      // from beyond the last segment (state 0) down: whole blocks above this one, then the segments of its own block
      double xin[W], min_[W];
#pragma unroll
      for (int b = 0; b < W; ++b) { xin[b] = 0.0; min_[b] = 0.0; }
      const int myblk = seg / BSEG_BS;
      constexpr int O_BB = W * W + W;  // the block's backward map behind its forward map
#pragma unroll 4
      for (int bk = nblk - 1; bk > myblk; --bk) {
        double nx[W], nm[W];
#pragma unroll
        for (int b = 0; b < W; ++b) {
          double a = bget(bk, O_BB + W * W + b), am = bget(bk, O_BB + W * W + W + b);
#pragma unroll
          for (int k = 0; k < W; ++k) {
            const double hkb = bget(bk, O_BB + k * W + b);
            a = fma(hkb, xin[k], a);
            am = fma(hkb, min_[k], am);
          }
          nx[b] = a; nm[b] = am;
        }
#pragma unroll
        for (int b = 0; b < W; ++b) { xin[b] = nx[b]; min_[b] = nm[b]; }
      }
      const int top = ((myblk + 1) * BSEG_BS < nseg ? (myblk + 1) * BSEG_BS : nseg) - 1;  // last segment of this block
#pragma unroll 8
      for (int sg = top; sg > seg; --sg) {
        double nx[W], nm[W];
#pragma unroll
        for (int b = 0; b < W; ++b) {
          double a = sget(sg, O_B + W * W + b), am = sget(sg, O_B + W * W + W + b);
#pragma unroll
          for (int k = 0; k < W; ++k) {
            const double hkb = sget(sg, O_B + k * W + b);
            a = fma(hkb, xin[k], a);
            am = fma(hkb, min_[k], am);
          }
          nx[b] = a; nm[b] = am;
        }
#pragma unroll
        for (int b = 0; b < W; ++b) { xin[b] = nx[b]; min_[b] = nm[b]; }
      }
#pragma unroll
      for (int d = 1; d < W1; ++d) { xs[d] = xin[d - 1]; ms[d] = min_[d - 1]; }
    }
#pragma unroll
    for (int k = 0; k < W; ++k)
#pragma unroll
      for (int d = 0; d < W1; ++d) Xk[k][d] = (d - 1 == k) ? 1.0 : 0.0;
    // ---- backward over the segment's columns hi-1 ... lo.  PHASE 1 carries the W unit incoming states and stores
    // nothing (the last segment's state is final as it stands and needs no map); PHASE 2 stores x and the mean.
    constexpr int PD = 4;  // columns in flight
    const bool fail_chain = failed;
    const int64_t gc = chain_offset + cc;
    const double* zrow = z_in ? z_in + cc * ld_z : nullptr;
    const double* Lp = L0 + (hi - 1) * (int64_t)W1 * C;
    const double* Up = U0 + (hi - 1) * (int64_t)W1 * C;
    // (PHASE 2 writes the caller's rows itself: a column's 64 results go to an LDS tile, and a tile of 32 columns -- aligned
    // in the row, so that neighbouring segments share no 128-byte line but the one their border cuts -- leaves as 64
    // pieces of 256 B.  The [column][chain] copy and its transpose launch, 82 MB each way per array, are gone.)
    int64_t out_blk = (hi - 1) >> 5;
    auto flush_tile = [&](int64_t blk) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int64_t col = blk * 32 + (lane & 31);
      const bool col_ok = col >= lo && col < hi;
#pragma unroll 4
      for (int it = 0; it < 32; ++it) {
        const int r = 2 * it + (lane >> 5);
        if (col_ok && c0 + r < C) {
          x_out[(c0 + r) * ld_x + col] = out_x[lane & 31][r];
          if (mean_out) mean_out[(c0 + r) * ld_mean + col] = out_m[lane & 31][r];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // A slot keeps what it loaded as it came (combining the responses with the incoming state at the fetch would make the
    // wave wait for the loads it has just issued) and an injected draw travels with its column's slot: as a load of its
    // own next to the use it was the youngest load, and the wait the compiler has to put in front of the use -- on both
    // sides of the branch -- emptied the queue at every column, the generated-draw case included: one memory latency per
    // column, 2.7 us measured, with the other three columns' loads never in flight.
    const bool inj = z_in != nullptr;  // uniform
    double lb[PD][W1], ub[PD][W1], zb[PD];
    auto fetch = [&](int t, int64_t back) {  // column (hi - 1 - back) into slot t
#pragma unroll
      for (int d = 0; d < W1; ++d) lb[t][d] = Lp[(d - back * W1) * C];
#pragma unroll
      for (int k = 0; k < W1; ++k) ub[t][k] = Up[(k - back * W1) * C];
      if (inj) zb[t] = zrow[hi - 1 - back];
    };
    const bool long_seg = hi - lo >= 2 * PD;  // uniform; (always, but for the last segment of a short chain)
    if (long_seg) {
#pragma unroll
      for (int t = 0; t < PD; ++t) fetch(t, t);
    } else {
#pragma unroll
      for (int t = 0; t < PD; ++t) {
#pragma unroll
        for (int d = 0; d < W1; ++d) { lb[t][d] = 0.0; ub[t][d] = 0.0; }
        zb[t] = 0.0;
        if (hi - 1 - t >= lo) fetch(t, t);
      }
    }
    double z_even = 0.0;
    bool have_even = false;
    int64_t back0 = 0;
    auto column = [&](int t, int64_t j) {  // column j from slot t
      double u = ub[t][0];
#pragma unroll
      for (int k = 0; k < W; ++k) u = fma(ub[t][k + 1], din[k], u);  // u = zero-state u + responses . incoming
      double z;
      if (inj) {
        z = zb[t];
      } else if ((j & 1) || !have_even) {  // a Philox block gives the draws of columns 2q and 2q+1
        double n0, n1;
        omc_normal_pair(omc_rng_block(key, gc, (uint32_t)(j >> 1)), n0, n1);
        z = (j & 1) ? n1 : n0;
        z_even = n0;
        have_even = (j & 1) != 0;
      } else {
        z = z_even;
        have_even = false;
      }
      double ax = u + z, am = u, ak[W];
#pragma unroll
      for (int k2 = 0; k2 < W; ++k2) ak[k2] = 0.0;
#pragma unroll
      for (int d = 1; d < W1; ++d) {
        ax = fma(-lb[t][d], xs[d], ax);
        am = fma(-lb[t][d], ms[d], am);
        if (PHASE == 1) {
#pragma unroll
          for (int k2 = 0; k2 < W; ++k2) ak[k2] = fma(-lb[t][d], Xk[k2][d], ak[k2]);
        }
      }
      const double l0 = lb[t][0];
      const double xv = ax * l0, mv = am * l0;
#pragma unroll
      for (int d = W; d > 1; --d) {
        xs[d] = xs[d - 1]; ms[d] = ms[d - 1];
#pragma unroll
        for (int k2 = 0; k2 < W; ++k2) Xk[k2][d] = Xk[k2][d - 1];
      }
      xs[1] = xv; ms[1] = mv;
#pragma unroll
      for (int k2 = 0; k2 < W; ++k2) Xk[k2][1] = ak[k2] * l0;
      if (PHASE == 2) {
        if ((j >> 5) != out_blk) {  // uniform
          flush_tile(out_blk);
          out_blk = j >> 5;
        }
        out_x[j & 31][lane] = fail_chain ? NAN : xv;
        if (mean_out) out_m[j & 31][lane] = mv;
      }
    };
    int64_t jt = hi - 1;
    // whole groups whose columns and refills all lie inside the segment: nothing in the loop but the recurrence and its loads
    // (the next group's loads go out first and land in a second set of registers, taken over when the group's own four
    // columns are done: refilling a slot right after its column leaves the last slot's loads no time before the next
    // group needs ... everything, since the compiler renames the slots by a copy at the loop's end)
    for (; jt - (2 * PD - 1) >= lo; jt -= PD, back0 += PD) {
      double lbn[PD][W1], ubn[PD][W1], zbn[PD];
#pragma unroll
      for (int t = 0; t < PD; ++t) {
        const int64_t back = back0 + t + PD;
#pragma unroll
        for (int d = 0; d < W1; ++d) lbn[t][d] = Lp[(d - back * W1) * C];
#pragma unroll
        for (int k = 0; k < W1; ++k) ubn[t][k] = Up[(k - back * W1) * C];
        zbn[t] = inj ? zrow[hi - 1 - back] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < PD; ++t) column(t, jt - t);
#pragma unroll
      for (int t = 0; t < PD; ++t) {
#pragma unroll
        for (int d = 0; d < W1; ++d) { lb[t][d] = lbn[t][d]; ub[t][d] = ubn[t][d]; }
        zb[t] = zbn[t];
      }
    }
    for (; jt >= lo; jt -= PD, back0 += PD) {
#pragma unroll
      for (int t = 0; t < PD; ++t) {
        const int64_t j = jt - t;
        if (j >= lo) column(t, j);
        if (j - PD >= lo) fetch(t, back0 + t + PD);
      }
    }
    if (PHASE == 2) flush_tile(out_blk);
    if (PHASE == 1) {
      int e = 0;
#pragma unroll
      for (int k = 0; k < W; ++k)
#pragma unroll
        for (int b = 0; b < W; ++b) sput_ag(seg, O_B + e++, Xk[k][b + 1]);
#pragma unroll
      for (int b = 0; b < W; ++b) sput_ag(seg, O_B + e++, xs[b + 1]);
#pragma unroll
      for (int b = 0; b < W; ++b) sput_ag(seg, O_B + e++, ms[b + 1]);
      // the block's backward map (its segments from the top one down), by the last of its waves to get here:
      // out[b] = h[b] + sum_k H[k][b] in[k], the same H for the draw's and the mean's states
      const int bk = seg / BSEG_BS;
      if (!last_of_block(bk)) return;
      double mp[BSEG_BS][NB];
      const int top = ((bk + 1) * BSEG_BS < nseg ? (bk + 1) * BSEG_BS : nseg) - 1;
#pragma unroll
      for (int i = 0; i < BSEG_BS; ++i) {
        const int sg = (top - i >= bk * BSEG_BS) ? top - i : bk * BSEG_BS;
#pragma unroll
        for (int e2 = 0; e2 < NB; ++e2) mp[i][e2] = sget_ag(sg, O_B + e2);
      }
      double H[W][W], h[W], hm[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        h[k] = 0.0; hm[k] = 0.0;
#pragma unroll
        for (int b = 0; b < W; ++b) H[k][b] = (k == b) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int i = 0; i < BSEG_BS; ++i) {
        if (top - i >= bk * BSEG_BS) {
          double nH[W][W], nh[W], nhm[W];
#pragma unroll
          for (int b = 0; b < W; ++b) {
            double a = mp[i][W * W + b], am = mp[i][W * W + W + b];
#pragma unroll
            for (int k = 0; k < W; ++k) {
              a = fma(mp[i][k * W + b], h[k], a);
              am = fma(mp[i][k * W + b], hm[k], am);
            }
            nh[b] = a; nhm[b] = am;
#pragma unroll
            for (int j = 0; j < W; ++j) {
              double t = 0.0;
#pragma unroll
              for (int k = 0; k < W; ++k) t = fma(mp[i][k * W + b], H[j][k], t);
              nH[j][b] = t;
            }
          }
#pragma unroll
          for (int b = 0; b < W; ++b) {
            h[b] = nh[b]; hm[b] = nhm[b];
#pragma unroll
            for (int j = 0; j < W; ++j) H[j][b] = nH[j][b];
          }
        }
      }
      constexpr int O_BB = W * W + W;
#pragma unroll
      for (int j = 0; j < W; ++j)
#pragma unroll
        for (int b = 0; b < W; ++b) bput(bk, O_BB + j * W + b, H[j][b]);
#pragma unroll
      for (int b = 0; b < W; ++b) { bput(bk, O_BB + W * W + b, h[b]); bput(bk, O_BB + W * W + W + b, hm[b]); }
    }
  }
}

// Does the term list fit k_band_lane (exactly one banded term, the others identities)?
static bool band_lane_args(const BandTermsDev& T, BandLaneArgs* P) {
  P->band = nullptr; P->bw = 0; P->s_band = nullptr; P->n_ident = 0; P->n_rhs = 0; P->rhs_t = nullptr; P->ld_t = 0;
  for (int k = 0; k < T.n_terms; ++k) {
    if (T.band[k]) {
      if (P->band) return false;
      P->band = T.band[k]; P->bw = T.bw[k]; P->s_band = T.scale[k];
    } else {
      P->s_ident[P->n_ident++] = T.scale[k];
    }
    if (T.rhs[k]) { P->rhs[P->n_rhs] = T.rhs[k]; P->s_rhs[P->n_rhs] = T.scale[k]; ++P->n_rhs; }
  }
  return P->band != nullptr;
}

template <int W>
static void launch_band_lane(omc_ctx* ctx, int64_t n, const BandLaneArgs& P, const double* z, int64_t ld_z, omc_rng_key key,
                             double* x, int64_t ld_x, double* mean, int64_t ld_mean, double* logdet,
                             const int* group_flag = nullptr) {
  hipLaunchKernelGGL((k_band_lane<W>), dim3((unsigned)((ctx->n_chains + 63) / 64)), dim3(64), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, n, P, z, ld_z, key, ctx->workspace, x, ld_x, mean, ld_mean, logdet, ctx->d_bad_chain,
                     group_flag);
}

// quad[c] = (x_c - m)' M (x_c - m) for a shared band matrix (NormalGamma.sample sampler.py:276,284; gmrf.py:343-344)
__global__ void __launch_bounds__(256) k_band_quadform(int64_t C, int64_t n, int w, const double* band, const double* center,
                                                       const double* x, int64_t ld, double* quad) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  const double* xc = x + c * ld;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const double ri = xc[i] - (center ? center[i] : 0.0);
    double row = (band ? band[i] : 1.0) * ri;
    for (int d = 1; d <= w && i + d < n; ++d) {
      const double rj = xc[i + d] - (center ? center[i + d] : 0.0);
      row = fma(2.0 * band[(int64_t)d * n + i], rj, row);
    }
    acc = fma(ri, row, acc);
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) acc += __shfl_xor(acc, sh, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) quad[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[c][i] (+)= scale[c] * sum_j M[i][j] v_c[j] for a shared symmetric band matrix in lower band storage (row d = the d-th
// sub-diagonal): the per-chain right-hand-side pieces of a hierarchical model on the band route (sampler.py:181-192)
__global__ void __launch_bounds__(256) k_band_matvec_chain(int64_t n, int w, const double* __restrict__ band, const double* __restrict__ v,
                                                           int64_t ld_v, const double* __restrict__ scale, double* __restrict__ out,
                                                           int64_t ld_out, int accumulate) {
  const int64_t c = blockIdx.y;
  const double* vc = v + c * ld_v;
  double* oc = out + c * ld_out;
  const double s = scale ? scale[c] : 1.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double acc = (band ? band[i] : 1.0) * vc[i];
    for (int d = 1; d <= w; ++d) {
      if (i + d < n) acc = fma(band[(int64_t)d * n + i], vc[i + d], acc);      // M[i+d][i] below the diagonal = M[i][i+d]
      if (i - d >= 0) acc = fma(band[(int64_t)d * n + i - d], vc[i - d], acc);  // M[i][i-d]
    }
    oc[i] = accumulate ? fma(s, acc, oc[i]) : s * acc;
  }
}

extern "C" {

omc_status omc_band_sample_canonical(omc_ctx* ctx, int64_t n, int64_t w, const omc_band_terms* terms,
                                     const double* rhs_chain, int64_t ld_rhs, const double* z_inject, int64_t ld_z,
                                     uint64_t draw_index, double* x, int64_t ld_x, double* mean, int64_t ld_mean,
                                     double* logdet) {
  if (!ctx || n < 1 || w < 0 || w > BAND_WMAX || !terms || terms->n_terms < 1 || terms->n_terms > OMC_MAX_TERMS || !x ||
      ld_x < n || (rhs_chain && ld_rhs < n) || (z_inject && ld_z < n) || (mean && ld_mean < n) || x == mean ||
      (z_inject && (z_inject == x || z_inject == mean)))
    return OMC_INVALID_ARG;
  BandTermsDev T{};
  T.n_terms = terms->n_terms;
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    const bool on = k < terms->n_terms;
    T.band[k] = on ? terms->band[k] : nullptr;
    T.bw[k] = on ? (int)terms->bw[k] : 0;
    T.rhs[k] = on ? terms->rhs[k] : nullptr;
    T.scale[k] = on ? terms->scale[k] : nullptr;
    if (on && (terms->bw[k] < 0 || terms->bw[k] > w)) return OMC_INVALID_ARG;
  }
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const int64_t Cn = ctx->n_chains;
  const int64_t groups = (Cn + 63) / 64;
  BandLaneArgs LP{};
  // Bands of width 4 .. 8 on up to 3072 chains take the blocked workgroup-per-chain kernel as well: a lane per chain is a lone
  // wave per SIMD at 1.4-2.1 us a column whatever the number of chains (14-21 ms per draw at n = 10 000, measured), the blocked
  // kernel in its four-wave forms 3.9 ms per 768 chains, 4.6 ms per 1024 (four workgroups to a CU): 18 ms at 4096
  // ("band_algo" 1 keeps the lane kernel, 3 takes the blocked one for every width).
  const bool blocked_first = ctx->band_algo == 3 || (ctx->band_algo == 0 && w >= 4 && Cn <= 3072);
  const bool lane_fits = w >= 1 && w <= 8 && ctx->band_algo != 2 && !blocked_first && band_lane_args(T, &LP);
  // Segmented route: segments of at least 96 columns and at least half the warm-up, their number chosen for the SIMDs.
  // The factor phase is bound by instruction issue, not by latency (measured: 0.5 us a column for a wave alone on its
  // SIMD, 0.9-1.0 us when two share one), so what counts is (segment + warm-up) x the waves the fullest SIMD gets:
  // 1664 waves on 1024 SIMDs took as long as 2048 would have.  A small charge per segment stands for what every wave
  // does once per segment of its group (joins, composition of the incoming states).
  const int ov = ctx->band_seg_overlap;
  int64_t min_seg = ov / 2 > 96 ? ov / 2 : 96;
  int dev_cus = 256;
  hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
  const int64_t simds = 4 * (int64_t)(dev_cus > 0 ? dev_cus : 256);
  int nseg = 0;
  {
    int64_t best = -1;
    for (int cand = 2; cand <= BSEG_MAX; ++cand) {
      const int64_t ms = (n + cand - 1) / cand;
      if (ms < min_seg) break;
      const int64_t rounds = (cand * groups + simds - 1) / simds;
      const int64_t cost = 2 * (ms + ov) * rounds + cand;
      if (best < 0 || cost < best) { best = cost; nseg = cand; }
    }
  }
  if (ctx->band_seg_count >= 2 && (int64_t)ctx->band_seg_count * 8 <= n) nseg = ctx->band_seg_count;
  const bool segmented = lane_fits && w <= 3 && nseg >= 2 && ctx->band_algo != 1;
  // factor, then per column the zero-state u and its w unit responses (segmented route) / u alone
  const size_t base_doubles = (size_t)Cn * n * (segmented ? 2 * (w + 1) : (w + 2));
  const int64_t mseg = segmented ? (n + nseg - 1) / nseg : 0;
  while (segmented && (int64_t)(nseg - 1) * mseg >= n) --nseg;  // no empty segment
  const int nblk = (nseg + BSEG_BS - 1) / BSEG_BS;
  const size_t scr_doubles = (size_t)groups * ((size_t)nseg * 48 + (size_t)nblk * BSEG_BROW) * 64;
  // + flags (2 per group) and the block counters of the four launches that compose (ints)
  const size_t seg_doubles = segmented ? scr_doubles + (size_t)groups + 2 * (size_t)groups * nblk + 2 : 0;
  const size_t rt_doubles = (lane_fits && rhs_chain) ? (size_t)n * groups * 64 : 0;  // the per-chain right-hand side, transposed
  omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->workspace, &ctx->workspace_bytes,
                                   (base_doubles + seg_doubles + rt_doubles) * sizeof(double));
  if (st != OMC_OK) return st;
  if (rt_doubles) {
    double* rt = ctx->workspace + base_doubles + seg_doubles;
    hipLaunchKernelGGL(k_band_rhs_transpose, dim3((unsigned)((n + 63) / 64), (unsigned)groups), dim3(256), 0, ctx->stream, Cn, n,
                       rhs_chain, ld_rhs, rt, groups * 64);
    OMC_HIP_CHECK(hipGetLastError());
    LP.rhs_t = rt;
    LP.ld_t = groups * 64;
  }
  if (segmented) {
    double* Lws = ctx->workspace;
    double* scratch = ctx->workspace + base_doubles;
    int* flags = (int*)(scratch + scr_doubles);
    int* counts = flags + 2 * groups;  // [launch 0..3][group][block], zero at the start of every call
    const size_t cnt_n = (size_t)groups * nblk;
    hipMemsetAsync(counts, 0, 4 * cnt_n * sizeof(int), ctx->stream);
    const omc_rng_key lane_key = omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL);
    const double tol = 1e-13;
    const dim3 sg((unsigned)nseg, (unsigned)groups);
    // flags: [0..groups) verdict of the first attempt (gate of the second), [groups..2 groups) final verdict
    int* flag1 = flags;
    int* flag2 = flags + groups;
    const int ov2 = 4 * ov;  // second attempt for the groups whose joins did not close (a long-memory prior): 4 x the warm-up
#define OMC_BSEG(Wv, PH, OV, GATE, FLAG, CNT, BCNT)                                                                               \
  hipLaunchKernelGGL((k_band_seg<Wv, PH>), sg, dim3(64), 0, ctx->stream, Cn, ctx->chain_offset, n, LP, nseg, mseg, OV, tol,       \
                     z_inject, ld_z, lane_key, Lws, x, ld_x, mean, ld_mean, scratch, (const int*)(GATE), FLAG, logdet,           \
                     ctx->d_bad_chain, CNT, BCNT)
#define OMC_BSEG_ALL(Wv)                                                                                                          \
  do {                                                                                                                            \
    OMC_BSEG(Wv, 0, ov, nullptr, flag1, ctx->d_fallbacks + 3, counts);                                                            \
    OMC_BSEG(Wv, 1, ov, nullptr, flag1, ctx->d_fallbacks + 3, counts + cnt_n);                                                    \
    hipMemcpyAsync(flag2, flag1, (size_t)groups * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream);                             \
    OMC_BSEG(Wv, 0, ov2, flag1, flag2, ctx->d_fallbacks + 2, counts + 2 * cnt_n);                                                 \
    OMC_BSEG(Wv, 1, ov2, flag1, flag2, ctx->d_fallbacks + 2, counts + 3 * cnt_n);                                                 \
    OMC_BSEG(Wv, 2, ov, nullptr, flag2, ctx->d_fallbacks + 2, nullptr);                                                           \
  } while (0)
    if (w == 1) OMC_BSEG_ALL(1); else if (w == 2) OMC_BSEG_ALL(2); else OMC_BSEG_ALL(3);
#undef OMC_BSEG_ALL
#undef OMC_BSEG
    flags = flag2;
    // groups whose joins did not close: in one piece (the kernel returns at once for the others)
    if (w == 1) launch_band_lane<1>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet, flags);
    else if (w == 2) launch_band_lane<2>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet, flags);
    else launch_band_lane<3>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet, flags);
    OMC_HIP_CHECK(hipGetLastError());
    return OMC_OK;
  }
  if (lane_fits) {
    // narrow band: one lane per chain, window in registers (see k_band_lane)
    const omc_rng_key lane_key = omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL);
    switch ((int)w) {
      case 1: launch_band_lane<1>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet); break;
      case 2: launch_band_lane<2>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet); break;
      case 3: launch_band_lane<3>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet); break;
      case 4: launch_band_lane<4>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet); break;
      case 5: launch_band_lane<5>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet); break;
      case 6: launch_band_lane<6>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet); break;
      case 7: launch_band_lane<7>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet); break;
      default: launch_band_lane<8>(ctx, n, LP, z_inject, ld_z, lane_key, x, ld_x, mean, ld_mean, logdet); break;
    }
    OMC_HIP_CHECK(hipGetLastError());
    return OMC_OK;
  }
  const int W1 = (int)w + 1;
  size_t lds = ((size_t)W1 * W1 + 2 * W1 + 2) * sizeof(double);
  const size_t lds_back = (size_t)(2 * W1 + 8) * sizeof(double);
  if (lds < lds_back) lds = lds_back;
  const omc_rng_key key = omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL);
  // wide bands (lattice GMRFs): NB columns per step, the window update on the matrix cores (omc_bandwide.hip); "band_algo" 3
  // forces it for any bandwidth, 2 keeps the column-at-a-time kernel
  if ((blocked_first || (ctx->band_algo != 2 && w >= 9)) &&
      omc_band_blocked_launch(ctx, n, (int)w, &T, rhs_chain, ld_rhs, z_inject, ld_z, key, ctx->workspace, x, ld_x, mean, ld_mean, logdet)) {
    OMC_HIP_CHECK(hipGetLastError());
    return OMC_OK;
  }
  if (w <= 7) {
    // narrow band: one wave per chain -- the two barriers per column cost nothing inside a single wave, and the
    // 8 x 8 thread tile already covers the (w x w)/2 update
    hipLaunchKernelGGL((k_band_sample<8, 8>), dim3((unsigned)ctx->n_chains), dim3(8, 8), lds, ctx->stream, ctx->n_chains,
                       ctx->chain_offset, n, (int)w, T, rhs_chain, ld_rhs, z_inject, ld_z, key, ctx->workspace, x, ld_x, mean,
                       ld_mean, logdet, ctx->d_bad_chain);
  } else {
    hipLaunchKernelGGL((k_band_sample<16, 16>), dim3((unsigned)ctx->n_chains), dim3(16, 16), lds, ctx->stream, ctx->n_chains,
                       ctx->chain_offset, n, (int)w, T, rhs_chain, ld_rhs, z_inject, ld_z, key, ctx->workspace, x, ld_x, mean,
                       ld_mean, logdet, ctx->d_bad_chain);
  }
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_band_quadform(omc_ctx* ctx, int64_t n, int64_t w, const double* band, const double* center, const double* x,
                             int64_t ld, double* quad) {
  if (!ctx || n < 1 || w < 0 || (w > 0 && !band) || !x || ld < n || !quad) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_band_quadform, dim3((unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, ctx->n_chains, n, (int)w, band,
                     center, x, ld, quad);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_band_matvec_chain(omc_ctx* ctx, int64_t n, int64_t w, const double* band, const double* v, int64_t ld_v,
                                 const double* scale, double* out, int64_t ld_out, int32_t accumulate) {
  if (!ctx || n < 1 || w < 0 || (w > 0 && !band) || !v || !out || ld_v < n || ld_out < n || v == out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  int64_t gx_ = (n + 255) / 256;
  if (gx_ > 64) gx_ = 64;
  hipLaunchKernelGGL(k_band_matvec_chain, dim3((unsigned)gx_, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, n, (int)w, band, v,
                     ld_v, scale, out, ld_out, (int)accumulate);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"
