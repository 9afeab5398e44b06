// Wide bands (16 <= w <= 128: lattice GMRFs with bandwidth sqrt(n); gmrf.py:489-520 on a sparse
// precision of that shape), BLOCKED: the same natural-order band Cholesky, forward and backward substitution as k_band_sample
// (omc_band.hip) -- one workgroup per chain, the open columns in an LDS ring -- but NB columns per step instead of one.
//
// k_band_sample pays two workgroup barriers per column for a rank-1 update of the (w x w)/2 window (3.6 us per column at
// w = 100: 145 ms per draw of 1024 chains at n = 10 000, 0.75 TFLOP/s) and two more per column on the way back.  Here a block
// of NB columns is: (1) the NB x NB diagonal block factorised by ONE wave in registers (rows in lanes, the pivot column's entries
// by v_readlane: no LDS, no barrier between its columns), (2) the w x NB panel below it solved against that block, a thread per
// row, (3) the trailing (w x w)/2 window updated by a rank-NB product on the MATRIX CORES (v_mfma_f64_16x16x4_f64: 16 x 16
// tiles of P P'), the right-hand side riding along as one more row -- four barriers per NB columns.  The backward pass takes NB
// columns per step as well: the products with the part of the solution behind the block spread over the lanes (16 per
// column), the NB x NB triangle by one lane per right-hand side.
// Same factor layout in the workspace as k_band_sample ([column][w + 1], 1 / L_jj in the diagonal slot), same random
// streams, same log det accumulation; results agree with it to rounding (another summation order), parity tests as for it.
#include <math.h>
#include <string.h>

#include "omc_common.h"

#define BAND_WMAX_W 128

namespace {

struct BandTermsW {  // (the image of omc_band.hip's BandTermsDev: kept in step by hand, both are filled from omc_band_terms)
  int n_terms;
  const double* band[OMC_MAX_TERMS];  // [(bw+1) x n], band[d*n + i] = M[i+d, i]; NULL = identity
  int bw[OMC_MAX_TERMS];
  const double* rhs[OMC_MAX_TERMS];
  const double* scale[OMC_MAX_TERMS];
};

__device__ __forceinline__ void lds_barrier_w() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double entry_w(const BandTermsW& T, const double* s, int64_t n, int64_t col, int d) {
  if (col >= n || col + d >= n) return 0.0;
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
__device__ __forceinline__ double rhs_w(const BandTermsW& T, const double* s, int64_t n, int64_t col, const double* rc) {
  if (col >= n) return 0.0;
  double b = rc ? rc[col] : 0.0;
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k)
    if (k < T.n_terms && T.rhs[k]) b = fma(s[k], T.rhs[k][col], b);
  return b;
}

__device__ __forceinline__ double readlane_d(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// column K of the NB x NB diagonal block (rows in lanes 0 .. NB-1, row r's entries D[0 .. r] in registers)
template <int K, int NB>
struct DiagStep {
  static __device__ __forceinline__ void run(double (&D)[NB], double (&dinv)[NB], bool& failed, double& ld_mant, long long& ld_exp, int live) {
    const double piv = readlane_d(D[K], K);
    const bool ok = piv > 0.0 || K >= live;   // (columns beyond the chain's end are identity padding)
    failed |= !ok;
    double rinv = 1.0, sq = 1.0;
    if (ok && K < live) {  // 1/sqrt(pivot) by rsq + two Newton steps, as k_band_sample does it
      const double g = __builtin_amdgcn_rsq(piv);
      const double h = 0.5 * g;
      sq = piv * g;
      double e = fma(-sq, sq, piv);
      sq = fma(e, h, sq);
      e = fma(-sq, sq, piv);
      sq = fma(e, h, sq);
      rinv = omc_rcp_nr(sq);
      ld_mant *= __builtin_amdgcn_frexp_mant(piv);
      ld_exp += __builtin_amdgcn_frexp_exp(piv);
      ld_exp += __builtin_amdgcn_frexp_exp(ld_mant);
      ld_mant = __builtin_amdgcn_frexp_mant(ld_mant);
    }
    dinv[K] = rinv;
    const double lk = D[K] * rinv;
    D[K] = ((int)(threadIdx.x & 63) == K) ? sq : lk;
#pragma unroll
    for (int cc = K + 1; cc < NB; ++cc) D[cc] = fma(-lk, readlane_d(lk, cc), D[cc]);
    DiagStep<K + 1, NB>::run(D, dinv, failed, ld_mant, ld_exp, live);
  }
};
template <int NB>
struct DiagStep<NB, NB> {
  static __device__ __forceinline__ void run(double (&)[NB], double (&)[NB], bool&, double&, long long&, int) {}
};

typedef double wide_d4 __attribute__((ext_vector_type(4)));

template <int NB, int NT>
__global__ void __launch_bounds__(NT) k_band_blocked(int64_t C, int64_t chain_offset, int64_t n, int w, BandTermsW T, const double* rhs_chain,
                                                      int64_t ld_rhs, const double* z_in, int64_t ld_z, omc_rng_key key, double* Lws, double* x,
                                                      int64_t ld_x, double* mean, int64_t ld_mean, double* logdet, long long* bad, unsigned long long* dbg) {
  extern __shared__ double sm[];
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define WSTAMP(i) do { if (dbg) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; } } while (0)
  const int W1 = w + 1;
  const int WS = w + NB;                 // columns of the window (ring slots)
  const int WP = (w + 15) & ~15;         // panel rows, padded to whole tiles
  constexpr int PS = NB + 1;             // panel row stride
  double* ring = sm;                               // WS x W1: ring[slot(col) * W1 + d] = open entry Q[col + d, col]
  double* rring = ring + (int64_t)WS * W1;         // WS: open right-hand side
  double* P = rring + WS;                          // WP x PS: the panel below the diagonal block (zero outside the band)
  double* Ld = P + (int64_t)WP * PS;               // NB x PS: the diagonal block's factor
  double* dv = Ld + NB * PS;                       // NB: 1 / L_jj of the block
  double* Us = dv + NB;                            // NB: forward-substituted right-hand side of the block
  double* misc = Us + NB;                          // [0] fail flag
  const int64_t c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) s[k] = (k < T.n_terms && T.scale[k]) ? T.scale[k][c] : 1.0;
  double* Lc = Lws + c * n * W1;
  double* xc = x + c * ld_x;
  const double* rc = rhs_chain ? rhs_chain + c * ld_rhs : nullptr;

  // open the first WS columns (slot of column col = col % WS, kept incrementally below)
  for (int t = tid; t < WS * W1; t += NT) {
    const int col = t / W1, d = t % W1;
    ring[col * W1 + d] = entry_w(T, s, n, col, d);
  }
  for (int t = tid; t < WS; t += NT) rring[t] = rhs_w(T, s, n, t, rc);
  if (tid == 0) misc[0] = 0.0;
  for (int t = tid; t < WP * PS; t += NT) P[t] = 0.0;  // (rows w .. WP - 1 pad the last tile: never written again)
  __syncthreads();

  double ld_mant = 1.0;
  long long ld_exp = 0;
  // the NB entering columns (their W1 band entries and their right-hand side, entry W1): 256 / NB threads per column, no division
  constexpr int TPC = NT / NB;                                   // threads per column
  constexpr int NPRE = (BAND_WMAX_W + 2 + TPC - 1) / TPC;         // entries per thread
  const int pcol = tid / TPC, pq = tid % TPC;                    // (NB is a power of two: shifts)
  int slot0 = 0;  // slot of column j
  for (int64_t j = 0; j < n; j += NB) {
    const int nb = (int)((n - j < NB) ? n - j : NB);
    // the NB columns that enter the window at the end of this block: their raw entries are REQUESTED now, all of them before any
    // is used (a load consumed inside a divergent branch is waited for on the spot: seven round trips in a row), combined and
    // stored in S4
    double raw[NPRE][OMC_MAX_TERMS], rawc[NPRE];
    {
      const int64_t col = j + WS + pcol;
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        const int d = pq + q * TPC;
        rawc[q] = 0.0;
#pragma unroll
        for (int k = 0; k < OMC_MAX_TERMS; ++k) {
          raw[q][k] = 0.0;
          if (k < T.n_terms && col < n) {
            if (d < W1) {
              if (T.band[k] && d <= T.bw[k] && col + d < n) raw[q][k] = T.band[k][(int64_t)d * n + col];
            } else if (d == W1 && T.rhs[k]) {
              raw[q][k] = T.rhs[k][col];
            }
          }
        }
        if (d == W1 && rc && col < n) rawc[q] = rc[col];
      }
    }
    WSTAMP(0);
    // ---- S1 (+ S2): the block column as ONE tall right-looking factorisation in registers.  Lanes 0 .. NB-1 of every taking-part
    // wave hold the diagonal block's rows (the same in each of these waves: the pivot column's entries reach the other lanes by
    // v_readlane), the lanes behind them 64 - NB rows of the panel below -- and one of them the right-hand side, which is one more
    // row of the matrix being factorised.  Scaling column K and updating the columns behind it is then the SAME instruction for
    // block, panel and right-hand side: the panel's triangular solve (5.4 k cycles and a barrier as a phase of its own) costs
    // nothing beyond the diagonal block's factorisation.
    {
      constexpr int RPW = 64 - NB;                       // panel rows per wave
      const int npw = (w + 1 + RPW - 1) / RPW;           // waves that take part: w panel rows + the right-hand side
      if (wave < npw) {
        double D[NB], dinv[NB];
        const int i = RPW * wave + (lane - NB);          // panel row of this lane (lane >= NB); i == w: the right-hand side
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          int sl = slot0 + b;
          if (sl >= WS) sl -= WS;
          double v;
          if (lane < NB) {
            const bool in = lane < nb && b < nb && b <= lane && lane - b <= w;  // (a band narrower than the block: zeros beyond it)
            v = in ? ring[sl * W1 + (lane - b)] : ((b == lane && lane >= nb) ? 1.0 : 0.0);
          } else if (i < w) {
            const int d = NB + i - b;                    // row j + NB + i against column j + b
            v = (b < nb && d <= w && j + NB + i < n) ? ring[sl * W1 + d] : 0.0;
          } else {
            v = (i == w && b < nb) ? rring[sl] : 0.0;
          }
          D[b] = v;
        }
        bool failed = false;
        DiagStep<0, NB>::run(D, dinv, failed, ld_mant, ld_exp, nb);
        if (lane < NB) {
          if (wave == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) Ld[lane * PS + b] = (b <= lane) ? D[b] : 0.0;
          }
        } else if (i < w) {
#pragma unroll
          for (int b = 0; b < NB; ++b) P[i * PS + b] = D[b];
        } else if (i == w) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            Us[b] = D[b];
            if (b < nb) xc[j + b] = D[b];  // forward-substituted right-hand side, overwritten by the draw in the backward pass
          }
        }
        if (wave == 0 && lane == 0) {
#pragma unroll
          for (int b = 0; b < NB; ++b) dv[b] = dinv[b];
          if (failed) misc[0] = 1.0;
        }
      }
    }
    lds_barrier_w();
    WSTAMP(1);
    WSTAMP(2);
    // ---- S3: the block's columns of the factor go to the workspace; the trailing window takes P P' on the matrix cores
    if (pcol < nb) {
      double* col = Lc + (j + pcol) * W1;
      for (int d = pq; d < W1; d += TPC) {
        double v;
        if (d == 0) v = dv[pcol];                                   // the diagonal slot holds 1 / L_jj
        else if (pcol + d < NB) v = (pcol + d < nb) ? Ld[(pcol + d) * PS + pcol] : 0.0;
        else v = (pcol + d - NB < w) ? P[(pcol + d - NB) * PS + pcol] : 0.0;
        col[d] = v;
      }
    }
    {
      const int nt = WP / 16;
      const int ntiles = nt * (nt + 1) / 2;
      const int cl = lane & 15, kr = lane >> 4;
      for (int tile = wave; tile < ntiles; tile += NT / 64) {
        // tile -> (ti >= tj), rows 16 ti .., columns 16 tj ..
        int ti = 0, rem = tile;
        while (rem > ti) { rem -= ti + 1; ++ti; }
        const int tj = rem;
        wide_d4 acc = wide_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(P[(16 * ti + cl) * PS + 4 * ks + kr], P[(16 * tj + cl) * PS + 4 * ks + kr], acc, 0, 0, 0);
        // C/D of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * register
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ri = 16 * ti + kr + 4 * r, ci = 16 * tj + cl;
          if (ri >= ci && ri < w && j + NB + ri < n) {
            int sl = slot0 + NB + ci;
            if (sl >= WS) sl -= WS;
            ring[sl * W1 + (ri - ci)] -= acc[r];
          }
        }
      }
    }
    if (tid < w && j + NB + tid < n) {
      double acc = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) acc = fma(P[tid * PS + b], Us[b], acc);
      int sl = slot0 + NB + tid;
      if (sl >= WS) sl -= WS;
      rring[sl] -= acc;
    }
    lds_barrier_w();
    WSTAMP(3);
    // ---- S4: the block's slots take the columns j + WS .. j + WS + NB - 1
    {
      int sl = slot0 + pcol;
      if (sl >= WS) sl -= WS;
      const int64_t col = j + WS + pcol;
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        const int d = pq + q * TPC;
        double v = rawc[q];
#pragma unroll
        for (int k = 0; k < OMC_MAX_TERMS; ++k) {
          if (k < T.n_terms) {
            if (d < W1 && !T.band[k]) { if (d == 0 && col < n) v += s[k]; }   // an identity term
            else v = fma(s[k], raw[q][k], v);
          }
        }
        if (d < W1) ring[sl * W1 + d] = v;
        else if (d == W1) rring[sl] = v;
      }
    }
    slot0 += NB;
    if (slot0 >= WS) slot0 -= WS;
    lds_barrier_w();
    WSTAMP(4);
  }
  const bool failed = misc[0] != 0.0;
  if (tid == 0) {
    if (logdet) logdet[c] = log(ld_mant) + (double)ld_exp * 0.69314718055994530942;
    if (failed) atomicMin((unsigned long long*)bad, (unsigned long long)c);
  }
  __syncthreads();
  if (failed) {
    for (int64_t i = tid; i < n; i += NT) xc[i] = NAN;
    return;
  }

  // t = u + z (all threads); the mean needs u alone: kept in the mean buffer (as k_band_sample)
  double* mc = mean ? mean + c * ld_mean : nullptr;
  for (int64_t i = tid; i < n; i += NT) {
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

  WSTAMP(5);
  // ---- backward pass L' x = t, NB columns per step.  xs / ms: the last WS solutions, slot = column % WS
  double* xs = sm;
  double* ms = xs + WS;
  double* Sx = ms + WS;        // NB: sum over the rows behind the block, per column
  double* Sm = Sx + NB;
  double* Lb = Sm + NB;        // NB x PS: the block's own triangle of the factor, Lb[a][b] = L[j+a][j+b], a > b; dinv on the diagonal
  for (int t = tid; t < 2 * WS; t += NT) sm[t] = 0.0;
  __syncthreads();
  const int64_t nblk = (n + NB - 1) / NB;
  int slotJ = (int)(((nblk - 1) * NB) % WS);  // slot of the block's first column, kept incrementally
  for (int64_t J = nblk - 1; J >= 0; --J) {
    const int64_t j = J * NB;
    const int nb = (int)((n - j < NB) ? n - j : NB);
    // S1: 256 / NB lanes per column: s_b = sum over the rows behind the block of L[j+b+d][j+b] x[j+b+d]; the block's own triangle goes
    // to LDS on the way
    // (wave 0's first 2 NB lanes ask for their right-hand side entries now: used in S2)
    double tval = 0.0;
    if (wave == 0 && lane < 2 * NB && (lane & (NB - 1)) < nb && (lane < NB || mc)) tval = (lane < NB ? xc : mc)[j + (lane & (NB - 1))];
    {
      const int b = pcol, q = pq;
      double px = 0.0, pm = 0.0;
      // the column's entries: all requested before the first is used
      double lv[NPRE];
#pragma unroll
      for (int t = 0; t < NPRE; ++t) {
        const int d = q + t * TPC;
        lv[t] = (b < nb && d <= w) ? Lc[(j + b) * W1 + d] : 0.0;
      }
      if (b < nb) {
        int sl = slotJ + b + q;
        while (sl >= WS) sl -= WS;
#pragma unroll
        for (int t = 0; t < NPRE; ++t) {
          const int d = q + t * TPC;
          if (d <= w) {
            const double l = lv[t];
            if (d == 0) Lb[b * PS + b] = l;                               // 1 / L_jj
            else if (b + d < nb) Lb[(b + d) * PS + b] = l;                // inside the block
            else if (j + b + d < n) {
              px = fma(l, xs[sl], px);
              if (mc) pm = fma(l, ms[sl], pm);
            }
          }
          sl += TPC;
          while (sl >= WS) sl -= WS;
        }
      }
#pragma unroll
      for (int sh = TPC / 2; sh > 0; sh >>= 1) {
        px += __shfl_xor(px, sh, 64);
        pm += __shfl_xor(pm, sh, 64);
      }
      if (q == 0) { Sx[b] = px; Sm[b] = pm; }
    }
    lds_barrier_w();
    WSTAMP(6);
    // S2: the NB x NB triangle by the first 2 NB lanes of wave 0: lane b (draw) and lane NB + b (mean) keep their own unknown; the
    // unknowns are finished from the last one up and handed to the lanes in front by v_readlane
    if (wave == 0) {
      const bool is_m = lane >= NB;
      const int b = lane & (NB - 1);
      const bool on = lane < 2 * NB && b < nb && (!is_m || mc);
      double* dst = is_m ? mc : xc;
      double acc = on ? tval - (is_m ? Sm[b] : Sx[b]) : 0.0;
      const double dinv_b = (lane < 2 * NB && b < nb) ? Lb[b * PS + b] : 0.0;
      double xv = 0.0;
#pragma unroll
      for (int a = NB - 1; a >= 0; --a) {
        // unknown a is complete in lanes a and NB + a
        const double fin = acc * dinv_b;
        if (b == a) xv = fin;
        const double xa = readlane_d(fin, a), ma = readlane_d(fin, NB + a);
        const double l = (a < nb && b < a && a - b <= w && lane < 2 * NB) ? Lb[a * PS + b] : 0.0;
        acc = fma(-l, is_m ? ma : xa, acc);
      }
      if (on) {
        dst[j + b] = xv;
        int sl = slotJ + b;
        if (sl >= WS) sl -= WS;
        (is_m ? ms : xs)[sl] = xv;
      }
    }
    slotJ -= NB;
    if (slotJ < 0) slotJ += WS;
    lds_barrier_w();
    WSTAMP(7);
  }
  if (dbg && blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) dbg[i] = tacc[i];
#undef WSTAMP
}

}  // namespace

// LDS bytes of a block size NB at bandwidth w (the factor phase is the larger one)
static size_t blocked_lds(int w, int NB) {
  const size_t W1 = (size_t)w + 1, WS = (size_t)w + NB, WP = ((size_t)w + 15) & ~(size_t)15, PS = (size_t)NB + 1;
  return (WS * W1 + WS + WP * PS + (size_t)NB * PS + 2 * (size_t)NB + 2) * sizeof(double);
}

// terms: omc_band.hip's BandTermsDev (the same layout as BandTermsW above); Lws: [C][n][w + 1] doubles.  Returns false if no block
// size fits the 160 KB of LDS (the caller then takes k_band_sample).
bool omc_band_blocked_launch(omc_ctx* ctx, int64_t n, int w, const void* terms, const double* rhs_chain, int64_t ld_rhs,
                             const double* z_inject, int64_t ld_z, omc_rng_key key, double* Lws, double* x, int64_t ld_x, double* mean,
                             int64_t ld_mean, double* logdet) {
  if (w < 1 || w > BAND_WMAX_W) return false;
  BandTermsW T;
  memcpy(&T, terms, sizeof(T));
  const size_t limit = 160 * 1024;
  if (blocked_lds(w, 16) <= limit) {
    hipLaunchKernelGGL((k_band_blocked<16, 512>), dim3((unsigned)ctx->n_chains), dim3(512), blocked_lds(w, 16), ctx->stream, ctx->n_chains,
                       ctx->chain_offset, n, w, T, rhs_chain, ld_rhs, z_inject, ld_z, key, Lws, x, ld_x, mean, ld_mean, logdet,
                       ctx->d_bad_chain, ctx->stamps);
    return true;
  }
  if (blocked_lds(w, 8) <= limit) {
    hipLaunchKernelGGL((k_band_blocked<8, 512>), dim3((unsigned)ctx->n_chains), dim3(512), blocked_lds(w, 8), ctx->stream, ctx->n_chains,
                       ctx->chain_offset, n, w, T, rhs_chain, ld_rhs, z_inject, ld_z, key, Lws, x, ld_x, mean, ld_mean, logdet,
                       ctx->d_bad_chain, ctx->stamps);
    return true;
  }
  return false;
}
