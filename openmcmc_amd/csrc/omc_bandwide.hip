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
#define BAND_W16_MAX 115  // 16 columns per step up to here (LDS: blocked_lds(116, 16) > 160 KB), 8 beyond

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

// column K of the NB x NB diagonal block (rows in lanes 0 .. NB-1, row r's entries D[0 .. r] in registers; the lanes behind them
// hold rows of the panel and the right-hand side, which take the same steps).  No branch on the pivot: a pivot that is not
// positive is COUNTED (the chain's results are replaced by NaN at the end) and the arithmetic goes on with whatever it gives.
// dkeep: lane K keeps 1 / L_KK.
template <int K, int NB>
struct DiagStep {
  static __device__ __forceinline__ void run(double (&D)[NB], double& dkeep, int& nfail, double& ld_mant, long long& ld_exp, bool keep_ld) {
    const double piv = readlane_d(D[K], K);
    nfail += (piv > 0.0) ? 0 : 1;
    // 1/sqrt(pivot) by rsq + two Newton steps, as k_band_sample does it
    const double g = __builtin_amdgcn_rsq(piv);
    const double h = 0.5 * g;
    double sq = piv * g;
    double e = fma(-sq, sq, piv);
    sq = fma(e, h, sq);
    e = fma(-sq, sq, piv);
    sq = fma(e, h, sq);
    const double rinv = omc_rcp_nr(sq);
    if (keep_ld) {  // (uniform: one wave keeps the log determinant; identity padding beyond the chain's end adds log 1)
      ld_mant *= __builtin_amdgcn_frexp_mant(piv);
      ld_exp += __builtin_amdgcn_frexp_exp(piv);
      ld_exp += __builtin_amdgcn_frexp_exp(ld_mant);
      ld_mant = __builtin_amdgcn_frexp_mant(ld_mant);
    }
    if ((int)(threadIdx.x & 63) == K) dkeep = rinv;
    const double lk = D[K] * rinv;
    D[K] = lk;  // (lane K: L_KK to rounding; the diagonal is never read from here, 1 / L_KK is)
#pragma unroll
    for (int cc = K + 1; cc < NB; ++cc) D[cc] = fma(-lk, readlane_d(lk, cc), D[cc]);
    DiagStep<K + 1, NB>::run(D, dkeep, nfail, ld_mant, ld_exp, keep_ld);
  }
};
template <int NB>
struct DiagStep<NB, NB> {
  static __device__ __forceinline__ void run(double (&)[NB], double&, int&, double&, long long&, bool) {}
};

typedef double wide_d4 __attribute__((ext_vector_type(4)));

template <int NB, int NT>
__global__ void __launch_bounds__(NT) k_band_blocked(int64_t C, int64_t chain_offset, int64_t n, int w, BandTermsW T, const double* rhs_chain,
                                                      int64_t ld_rhs, const double* z_in, int64_t ld_z, omc_rng_key key, double* Lws, double* x,
                                                      int64_t ld_x, double* mean, int64_t ld_mean, double* logdet, long long* bad, unsigned long long* dbg) {
  extern __shared__ double sm[];
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter(), twork = 0;
#define WSTAMP(i) do { if (dbg) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; } } while (0)
  const int W1 = w + 1;
  const int WS = w + NB;                 // columns of the window (ring slots)
  const int WP = (w + 15) & ~15;         // panel rows, padded to whole tiles
  constexpr int PS = NB + 1;             // panel row stride
  double* ring = sm;                               // WS x W1: ring[slot(col) * W1 + d] = open entry Q[col + d, col]
  double* rring = ring + (int64_t)WS * W1;         // WS: open right-hand side
  // a factorised block column, TWO copies (the block being applied and the one factorised ahead of it), FBS doubles apart:
  double* P = rring + WS;                          // WP x PS: the panel below the diagonal block (zero outside the band)
  double* Ld = P + (int64_t)WP * PS;               // NB x PS: the diagonal block's factor
  double* dv = Ld + NB * PS;                       // NB: 1 / L_jj of the block
  double* Us = dv + NB;                            // NB: forward-substituted right-hand side of the block
  const int FBS = WP * PS + NB * PS + 2 * NB;
  double* misc = P + 2 * FBS;                      // [0] fail flag
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
  if (tid == 0) misc[0] = misc[1] = 0.0;  // [0] a pivot was not positive, [1] a zero to read
  for (int t = tid; t < WP * PS; t += NT) P[t] = P[FBS + t] = 0.0;  // (rows w .. WP - 1 pad the last tile: never written again)
  __syncthreads();

  double ld_mant = 1.0;
  long long ld_exp = 0;
  // the NB entering columns (their W1 band entries and their right-hand side, entry W1): NT / NB threads per column, no division.
  // What does not change from block to block -- which term has an entry at this thread's distance from the diagonal, and where
  // its row of the band storage starts -- is worked out once: a block then asks for its entries with a compare and a load each.
  constexpr int TPC = NT / NB;                                   // threads per column
  constexpr int WMAX_NB = (NB == 16) ? BAND_W16_MAX : BAND_WMAX_W;  // widest band this block size is launched for
  constexpr int NPRE = (WMAX_NB + 2 + TPC - 1) / TPC;             // entries per thread
  const int pcol = tid / TPC, pq = tid % TPC;                    // (NB is a power of two: shifts)
  const double* pbase[NPRE][OMC_MAX_TERMS];                      // term k's row d of the band (or its right-hand side), NULL: no entry
  const double* pchain[NPRE];                                    // the chain's own right-hand side (entry W1 only)
  int plim[NPRE];                                                // the entry exists for columns < plim
#pragma unroll
  for (int q = 0; q < NPRE; ++q) {
    const int d = pq + q * TPC;
    plim[q] = (d < W1) ? (int)n - d : (d == W1 ? (int)n : 0);
    pchain[q] = (d == W1) ? rc : nullptr;
#pragma unroll
    for (int k = 0; k < OMC_MAX_TERMS; ++k) {
      const double* p = nullptr;
      if (k < T.n_terms) {
        if (d < W1) { if (T.band[k] && d <= T.bw[k]) p = T.band[k] + (int64_t)d * n; }
        else if (d == W1) p = T.rhs[k];
      }
      pbase[q][k] = p;
    }
  }
  // look-ahead: block j + NB's columns are complete once the FIRST tile column of block j's window update is in, so its
  // factorisation (one to three waves, the serial part: 14 k of the 30 k cycles of a block) runs beside the rest of that update
  // on the other waves, into the second copy of P / Ld / dv / Us.  Bands narrower than a block (w < NB: the columns entering
  // at the end of block j are already part of block j + NB) keep the plain order.
  const bool ahead = w >= NB;
  constexpr int RPW = 64 - NB;                       // panel rows per wave in the block factorisation
  const int npw = (w + 1 + RPW - 1) / RPW;           // waves that take part in it: w panel rows + the right-hand side
  // ---- the block column as ONE tall right-looking factorisation in registers.  Lanes 0 .. NB-1 of every taking-part wave hold
  // the diagonal block's rows (the same in each of these waves: the pivot column's entries reach the other lanes by v_readlane),
  // the lanes behind them 64 - NB rows of the panel below -- and one of them the right-hand side, which is one more row of the
  // matrix being factorised.  Scaling column K and updating the columns behind it is then the SAME instruction for block, panel
  // and right-hand side: the panel's triangular solve costs nothing beyond the diagonal block's factorisation.
  // (the waves that factorise run at raised priority: they share their SIMDs with waves of the window update, and every issue slot
  // lost to those is on the critical path)
  const int frow_i = RPW * wave + (lane - NB);                    // panel row of this lane (lane >= NB); == w: the right-hand side
  const bool f_rhs = lane >= NB && frow_i == w;
  const int f_rowoff = lane < NB ? lane : NB + frow_i;            // row jj + f_rowoff of the matrix (the rhs lane: any value >= NB)
  const bool f_row = lane < NB || frow_i < w;
  // where the lane's entry against column jj + b sits: ring[slot * W1 + f_rowoff - b], the right-hand side rring[slot] (= ring[WS W1 + slot])
  const int f_mul = f_rhs ? 1 : W1, f_base = f_rhs ? WS * W1 : f_rowoff, f_dec = f_rhs ? 0 : 1;
  const int zero_at = (int)(misc + 1 - ring);                     // misc[1] == 0.0
  auto factor_block = [&](const int64_t jj, const int slot, double* Pn, double* Ldn, double* dvn, double* Usn) {
    __builtin_amdgcn_s_setprio(3);
    const int nbj = (int)((n - jj < NB) ? n - jj : NB);
    double D[NB];
    // the columns b of the block this lane holds an entry against: b_lo .. b_hi (empty for a row beyond the chain's end and for an
    // idle lane; a band narrower than the block leaves zeros beyond it)
    const int offmax = (f_row && jj + f_rowoff < n) ? w : -1;
    int b_lo = f_rhs ? 0 : ((f_rowoff - offmax > 0) ? f_rowoff - offmax : 0);
    const int b_hi = (f_rowoff < nbj - 1) ? f_rowoff : nbj - 1;
    if (b_hi < b_lo) b_lo = 0x10000;                       // (empty: no b passes the test below)
    const unsigned span = (unsigned)(b_hi < b_lo ? 0 : b_hi - b_lo);
    // the entry's place in the ring: slot (block-uniform, wraps) * f_mul + f_base - b * f_dec; lanes without an entry read a zero
    // kept in LDS for them, so that the sixteen reads are issued back to back with nothing between a read and its use
    int sl = slot, rest = f_base;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bool in = (unsigned)(b - b_lo) <= span;
      D[b] = ring[in ? sl * f_mul + rest : zero_at];
      rest -= f_dec;
      if (++sl == WS) sl = 0;
    }
    if (nbj < NB) {  // the last block of a chain whose length is no multiple of NB: identity padding
      asm volatile("" ::: "memory");  // (a real branch: taken once per chain at most)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (lane == b && b >= nbj) D[b] = 1.0;
    }
    int nfail = 0;
    double dkeep = 0.0;
    DiagStep<0, NB>::run(D, dkeep, nfail, ld_mant, ld_exp, wave == 0);
    // rows of the block (first wave only; entries on and above the diagonal are never read), rows of the panel, the right-hand side
    double* dst = lane < NB ? (wave == 0 ? Ldn + lane * PS : nullptr) : (frow_i < w ? Pn + frow_i * PS : (f_rhs ? Usn : nullptr));
    if (dst) {
#pragma unroll
      for (int b = 0; b < NB; ++b) dst[b] = D[b];
    }
    if (tid < NB) dvn[tid] = dkeep;
    if (tid == 0 && nfail) misc[0] = 1.0;
    __builtin_amdgcn_s_setprio(0);
  };
  // one 16 x 16 tile of P P' off the window (rows 16 ti .., columns 16 tj .. behind the block)
  const int cl = lane & 15, kr = lane >> 4;
  auto window_tile = [&](const int64_t j, const int slot0, const double* Pc, const int ti, const int tj) {
    wide_d4 acc = wide_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pc[(16 * ti + cl) * PS + 4 * ks + kr], Pc[(16 * tj + cl) * PS + 4 * ks + kr], acc, 0, 0, 0);
    // C/D of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * register
    const int ci = 16 * tj + cl;
    int sl = slot0 + NB + ci;
    if (sl >= WS) sl -= WS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ri = 16 * ti + kr + 4 * r;
      if (ri >= ci && ri < w && j + NB + ri < n) ring[sl * W1 + (ri - ci)] -= acc[r];
    }
  };
  const int nt = WP / 16;  // tiles per side of the window (<= 8)
  // block j's columns of the factor to the workspace (lanes along a column's entries) and the tiles (ti >= tj >= 1) of its window
  // update, by the waves w0 .. NT / 64 - 1
  auto store_and_tiles = [&](const int64_t j, const int nb, const int slot0, const double* Pc, const double* Ldc, const double* dvc,
                             const int w0) {
    const int nwk = NT / 64 - w0, wk = wave - w0;
    for (int b = wk; b < nb; b += nwk) {
      double* col = Lc + (j + b) * W1;
      for (int d = lane; d < W1; d += 64) {
        double v;
        if (d == 0) v = dvc[b];                                     // the diagonal slot holds 1 / L_jj
        else if (b + d < NB) v = (b + d < nb) ? Ldc[(b + d) * PS + b] : 0.0;
        else v = (b + d - NB < w) ? Pc[(b + d - NB) * PS + b] : 0.0;
        col[d] = v;
      }
    }
    const int ntiles = nt * (nt - 1) / 2;
    for (int tile = wk; tile < ntiles; tile += nwk) {
      int ti = 0, rem = tile;
      while (rem > ti) { rem -= ti + 1; ++ti; }
      window_tile(j, slot0, Pc, ti + 1, rem + 1);
    }
  };
  // S4: block j's slots take the columns j + WS .. j + WS + NB - 1 (requested at the top of the block)
  // (macro: the raw entries live in registers of the enclosing scope)
#define BAND_REFILL()                                                                      \
  do {                                                                                     \
    int sl_ = slot0 + pcol;                                                                \
    if (sl_ >= WS) sl_ -= WS;                                                              \
    const int64_t col_ = j + WS + pcol;                                                    \
    _Pragma("unroll") for (int q = 0; q < NPRE; ++q) {                                     \
      const int d = pq + q * TPC;                                                          \
      double v = rawc[q];                                                                  \
      _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) {                          \
        if (k < T.n_terms) {                                                               \
          if (d < W1 && !T.band[k]) { if (d == 0 && col_ < n) v += s[k]; } /* identity */  \
          else v = fma(s[k], raw[q][k], v);                                                \
        }                                                                                  \
      }                                                                                    \
      if (d < W1) ring[sl_ * W1 + d] = v;                                                  \
      else if (d == W1) rring[sl_] = v;                                                    \
    }                                                                                      \
  } while (0)
  // The loop starts one block early (j = -NB: nothing to apply, block 0 to factorise), so that the factorisation is written once.
  int slot0 = WS - NB;     // slot of column j
  int cur = 1;             // which copy of P / Ld / dv / Us holds block j
  for (int64_t j = -NB; j < n; j += NB) {
    const bool apply = j >= 0;
    const int nb = (int)((n - j < NB) ? n - j : NB);
    const double* Pc = P + cur * FBS, * Ldc = Ld + cur * FBS, * dvc = dv + cur * FBS, * Usc = Us + cur * FBS;
    const int nxt = cur ^ 1;
    int slot1 = slot0 + NB;
    if (slot1 >= WS) slot1 -= WS;
    // the NB columns that enter the window at the end of this block: their raw entries are REQUESTED now, all of them before any
    // is used (a load consumed inside a divergent branch is waited for on the spot: seven round trips in a row), combined and
    // stored in S4
    double raw[NPRE][OMC_MAX_TERMS], rawc[NPRE];
    if (apply) {
      const int col = (int)j + WS + pcol;
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        const bool in = col < plim[q];
        rawc[q] = (in && pchain[q]) ? pchain[q][col] : 0.0;
#pragma unroll
        for (int k = 0; k < OMC_MAX_TERMS; ++k) raw[q][k] = (in && pbase[q][k]) ? pbase[q][k][col] : 0.0;
      }
      WSTAMP(0);
      // ---- S2: what block j + NB waits for: the first tile column of the window update and the right-hand side
      for (int ti = wave; ti < nt; ti += NT / 64) window_tile(j, slot0, Pc, ti, 0);
      if (tid < w && j + NB + tid < n) {
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) acc = fma(Pc[tid * PS + b], Usc[b], acc);
        int sl = slot1 + tid;
        if (sl >= WS) sl -= WS;
        rring[sl] -= acc;
      }
      // (the block's forward-substituted right-hand side: overwritten by the draw in the backward pass)
      if (tid >= NT - 64 && lane < nb) xc[j + lane] = Usc[lane];
      if (!ahead) store_and_tiles(j, nb, slot0, Pc, Ldc, dvc, 0);
      lds_barrier_w();
      WSTAMP(2);
      if (!ahead) {
        BAND_REFILL();
        lds_barrier_w();
        WSTAMP(4);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        rawc[q] = 0.0;
#pragma unroll
        for (int k = 0; k < OMC_MAX_TERMS; ++k) raw[q][k] = 0.0;
      }
    }
    // ---- S3: block j + NB factorised by the first waves, beside the rest of block j's work on the others
    const unsigned long long tw0 = dbg ? __builtin_readcyclecounter() : 0;
    if (wave < npw) {
      if (j + NB < n) factor_block(j + NB, slot1, P + nxt * FBS, Ld + nxt * FBS, dv + nxt * FBS, Us + nxt * FBS);
    } else if (ahead && apply) {
      store_and_tiles(j, nb, slot0, Pc, Ldc, dvc, npw);
    }
    if (dbg) twork += __builtin_readcyclecounter() - tw0;   // (this wave's own work in S3, without the wait at the barrier)
    lds_barrier_w();
    WSTAMP(3);
    if (ahead && apply) {
      BAND_REFILL();
      lds_barrier_w();
      WSTAMP(4);
    }
    slot0 = slot1;
    cur = nxt;
  }
#undef BAND_REFILL
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
  if (dbg && blockIdx.x == 0 && lane == 0) dbg[8 + wave] = twork;   // S3 per wave
#undef WSTAMP
}

}  // namespace

// LDS bytes of a block size NB at bandwidth w (the factor phase is the larger one)
static size_t blocked_lds(int w, int NB) {
  const size_t W1 = (size_t)w + 1, WS = (size_t)w + NB, WP = ((size_t)w + 15) & ~(size_t)15, PS = (size_t)NB + 1;
  return (WS * W1 + WS + 2 * (WP * PS + (size_t)NB * PS + 2 * (size_t)NB) + 2) * sizeof(double);
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
  if (w <= BAND_W16_MAX && blocked_lds(w, 16) <= limit) {
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
