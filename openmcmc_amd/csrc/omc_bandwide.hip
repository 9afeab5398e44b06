// Wide bands (16 <= w <= 128: lattice GMRFs with bandwidth sqrt(n); gmrf.py:489-520 on a sparse
// precision of that shape), BLOCKED: the same natural-order band Cholesky, forward and backward substitution as k_band_sample
// (omc_band.hip) -- one workgroup per chain, the open columns in an LDS ring -- but NB columns per step instead of one.
//
// k_band_sample pays two workgroup barriers per column for a rank-1 update of the (w x w)/2 window (3.6 us per column at
// w = 100: 145 ms per draw of 1024 chains at n = 10 000, 0.75 TFLOP/s) and two more per column on the way back.  Here a block
// of NB columns is
//   (1) the block column -- the NB x NB diagonal block, the w x NB panel below it and the right-hand side as one more row --
//       factorised as ONE tall right-looking factorisation in registers by one to three waves (rows in lanes, the pivot column's
//       entries by v_readlane: no LDS, no barrier between its columns);
//   (2) the trailing (w x w)/2 window updated by a rank-NB product on the MATRIX CORES (v_mfma_f64_16x16x4_f64: 16 x 16 tiles of
//       P P'), the block's columns of the factor written to the workspace.
// Block j + NB is factorised AHEAD: it only needs the first tile column of block j's update, so (1) for block j + NB runs beside
// the rest of (2) for block j, out of a second copy of the block column in LDS; the rest of (2) is a list of items the waves take
// from a counter in LDS, the factorising waves too once they are done.  Three barriers per NB columns.
// The backward pass takes NB columns per step and ONE barrier: the products with the solution two blocks and more behind are
// summed for the next block by seven waves while the first solves the current one (the rows of the block behind it and the block's
// own triangle across lanes).  Every request to global memory is made a block before it is used.
// Same factor layout in the workspace as k_band_sample ([column][w + 1], 1 / L_jj in the diagonal slot), same random
// streams; results agree with it to rounding (another summation order), parity tests as for it.  Sums are taken in fixed orders:
// results do not depend on timing.
#include <math.h>
#include <string.h>

#include <type_traits>

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

// v of the lane a DPP control word names (quad permutations, row mirrors), both halves of the double
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true));
}

// column K of the NB x NB diagonal block (rows in lanes 0 .. NB-1, row r's entries D[0 .. r] in registers; the lanes behind them
// hold rows of the panel and the right-hand side, which take the same steps).  No branch on the pivot: a pivot that is not
// positive is COUNTED (the chain's results are replaced by NaN at the end) and the arithmetic goes on with whatever it gives.
// dkeep: lane K keeps 1 / L_KK.
template <int K, int NB>
struct DiagStep {
  static __device__ __forceinline__ void run(double (&D)[NB], double& dkeep, int& nfail) {
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
    if ((int)(threadIdx.x & 63) == K) dkeep = rinv;
    const double lk = D[K] * rinv;
    D[K] = lk;  // (lane K: L_KK to rounding; the diagonal is never read from here, 1 / L_KK is)
#pragma unroll
    for (int cc = K + 1; cc < NB; ++cc) D[cc] = fma(-lk, readlane_d(lk, cc), D[cc]);
    DiagStep<K + 1, NB>::run(D, dkeep, nfail);
  }
};
template <int NB>
struct DiagStep<NB, NB> {
  static __device__ __forceinline__ void run(double (&)[NB], double&, int&) {}
};

typedef double wide_d4 __attribute__((ext_vector_type(4)));

// MT: the number of terms the kernel is compiled for (2 covers most models: their entries a block ahead are half the registers)
// WPE: waves per SIMD the register allocation leaves room for (4: 128 registers -- two 512-thread or four 256-thread workgroups on a CU)
template <int NB, int NT, int MT, int WPE>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE, 8))) k_band_blocked(int64_t C, int64_t chain_offset, int64_t n, int w, BandTermsW T, const double* rhs_chain,
                                                      int64_t ld_rhs, const double* z_in, int64_t ld_z, omc_rng_key key, double* Lws, double* x,
                                                      int64_t ld_x, double* mean, int64_t ld_mean, double* logdet, long long* bad, unsigned long long* dbg) {
  static_assert(NT == 512 || NT == 256, "eight waves: one per tile of the window's first tile column; four for bands up to 15");
  extern __shared__ double sm[];
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter(), twork = 0, tback = 0;
#define WSTAMP(i) do { if (dbg) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; } } while (0)
  const int W1 = w + 1;
  const int WS = w + NB;                 // columns of the window (ring slots)
  const int WP = (w + 15) & ~15;         // panel rows, padded to whole tiles
  constexpr int PS = NB + 1;             // panel row stride
  double* ring = sm;                               // WS x W1: ring[slot(col) * W1 + d] = open entry Q[col + d, col]
  double* rring = ring + (int64_t)WS * W1;         // WS: open right-hand side
  // a factorised block column, TWO copies (the block being applied and the one factorised ahead of it), FBS doubles apart:
  double* Ld = rring + WS;                         // NB x PS: the diagonal block's factor (its strict lower triangle is what is read)
  double* P = Ld + NB * PS;                        // WP x PS: the panel below it, right behind: row r of the block column = Ld[r * PS ..]
  double* dv = P + (int64_t)WP * PS;               // NB: 1 / L_jj of the block
  double* Us = dv + NB;                            // NB: forward-substituted right-hand side of the block
  const int FBS = WP * PS + NB * PS + 2 * NB;
  double* misc = Ld + 2 * FBS;                     // [0] fail flag, [1] a zero to read, [2 .. 65] a slot per lane to write to in vain, [66] a counter
  const int64_t c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s[MT];
#pragma unroll
  for (int k = 0; k < MT; ++k) s[k] = (k < T.n_terms && T.scale[k]) ? T.scale[k][c] : 1.0;
  double* Lc = Lws + c * n * W1;
  double* xc = x + c * ld_x;
  const double* rc = rhs_chain ? rhs_chain + c * ld_rhs : nullptr;

  // open the first WS columns (slot of column col = col % WS, kept incrementally below)
  for (int t = tid; t < WS * W1; t += NT) {
    const int col = t / W1, d = t % W1;
    ring[col * W1 + d] = entry_w(T, s, n, col, d);
  }
  for (int t = tid; t < WS; t += NT) rring[t] = rhs_w(T, s, n, t, rc);
  if (tid == 0) {
    misc[0] = misc[1] = 0.0;  // [0] a pivot was not positive, [1] a zero to read
    *(int*)(misc + 66) = 0;   // [66] the next item of the block's list (see drain_items)
  }
  for (int t = tid; t < WP * PS; t += NT) P[t] = P[FBS + t] = 0.0;  // (rows w .. WP - 1 pad the last tile: never written again)
  __syncthreads();

  double ld_mant = 1.0;   // (threads 0 .. NB-1: the product of this lane's 1 / L_KK, see factor_block)
  int ld_exp = 0;
  // the NB entering columns (their W1 band entries and their right-hand side, entry W1): NT / NB threads per column, no division.
  // What does not change from block to block -- which term has an entry at this thread's distance from the diagonal, and where
  // its row of the band storage starts -- is worked out once: a block then asks for its entries with a compare and a load each.
  constexpr int TPC = NT / NB;                                   // threads per column
  // widest band this instantiation is launched for (the four-wave form: bands narrower than a block, several workgroups per CU)
  constexpr int WMAX_NB = (NT == 256) ? ((NB == 16) ? 15 : 64) : ((NB == 16) ? BAND_W16_MAX : BAND_WMAX_W);
  constexpr int NPRE = (WMAX_NB + 2 + TPC - 1) / TPC;             // entries per thread
  const int pcol = tid / TPC, pq = tid % TPC;                    // (NB is a power of two: shifts)
  const double* pbase[NPRE][MT];                      // term k's row d of the band (or its right-hand side), NULL: no entry
  const double* pchain[NPRE];                                    // the chain's own right-hand side (entry W1 only)
  int plim[NPRE];                                                // the entry exists for columns < plim
#pragma unroll
  for (int q = 0; q < NPRE; ++q) {
    const int d = pq + q * TPC;
    plim[q] = (d < W1) ? (int)n - d : (d == W1 ? (int)n : 0);
    pchain[q] = (d == W1) ? rc : nullptr;
#pragma unroll
    for (int k = 0; k < MT; ++k) {
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
  const int dump_at = zero_at + 1 + lane;
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
    asm volatile("" : "+v"(rest));  // (not worth sixteen registers held through the whole kernel to save a subtraction each)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bool in = (unsigned)(b - b_lo) <= span;
      D[b] = ring[in ? __mul24(sl, f_mul) + rest : zero_at];
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
    DiagStep<0, NB>::run(D, dkeep, nfail);
    // log det: lane K of the first wave keeps the running product of its 1 / L_KK over the blocks as mantissa and exponent
    // (identity padding beyond the chain's end: factors 1); the lanes are put together at the end
    if (tid < NB) {
      ld_mant *= __builtin_amdgcn_frexp_mant(dkeep);
      ld_exp += __builtin_amdgcn_frexp_exp(dkeep);
      ld_exp += __builtin_amdgcn_frexp_exp(ld_mant);
      ld_mant = __builtin_amdgcn_frexp_mant(ld_mant);
    }
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
  // (two tiles at a time where there are two: the operand reads, the chained matrix-core steps and the read-modify-write of
  // the ring are each a latency a lone tile waits out)
  auto window_tiles = [&](auto two_c, const int64_t j, const int slot0, const double* Pc, const int ti0, const int tj0, const int ti1,
                          const int tj1) {
    constexpr bool two = decltype(two_c)::value;
    double a0[NB / 4], b0[NB / 4], a1[NB / 4], b1[NB / 4];
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks) {
      a0[ks] = Pc[(16 * ti0 + cl) * PS + 4 * ks + kr];
      b0[ks] = Pc[(16 * tj0 + cl) * PS + 4 * ks + kr];
      a1[ks] = two ? Pc[(16 * ti1 + cl) * PS + 4 * ks + kr] : 0.0;
      b1[ks] = two ? Pc[(16 * tj1 + cl) * PS + 4 * ks + kr] : 0.0;
    }
    wide_d4 acc0 = wide_d4{0.0, 0.0, 0.0, 0.0}, acc1 = wide_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks], b0[ks], acc0, 0, 0, 0);
      if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ks], b1[ks], acc1, 0, 0, 0);
    }
    // C/D of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * register
    const int ci0 = 16 * tj0 + cl, ci1 = 16 * tj1 + cl;
    int sl0 = slot0 + NB + ci0, sl1 = slot0 + NB + ci1;
    if (sl0 >= WS) sl0 -= WS;
    if (sl1 >= WS) sl1 -= WS;
    const int rows_left = (n - j - NB < (int64_t)w) ? (int)(n - j - NB) : w;   // rows behind the block that exist
    // (lanes outside the band or beyond the chain's end take the same steps on a slot of their own that nobody reads)
    int at0[4], at1[4];
    double c0[4], c1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ri0 = 16 * ti0 + kr + 4 * r, ri1 = 16 * ti1 + kr + 4 * r;
      at0[r] = (ri0 >= ci0 && ri0 < rows_left) ? sl0 * W1 + (ri0 - ci0) : dump_at;
      c0[r] = ring[at0[r]];
      if (two) {
        at1[r] = (ri1 >= ci1 && ri1 < rows_left) ? sl1 * W1 + (ri1 - ci1) : dump_at;
        c1[r] = ring[at1[r]];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ring[at0[r]] = c0[r] - acc0[r];
      if (two) ring[at1[r]] = c1[r] - acc1[r];
    }
  };
  const int nt = WP / 16;  // tiles per side of the window (<= 8)
  // What is left of block j once block j + NB can be factorised -- the tiles (ti >= tj >= 1) of its window update, two at a time,
  // and its columns of the factor, to be written to the workspace (lanes along a column's entries: 1 / L_jj in the diagonal slot,
  // then the rows of the block column, which lie one behind the other in LDS) -- is a list of items the waves TAKE from a
  // counter in LDS: the waves that factorise join when they are done, and nobody waits for the slowest share.  The items touch
  // disjoint entries: who takes which does not change a bit of the result.
  int* const q_next = (int*)(misc + 66);
  auto drain_items = [&](const int64_t j, const int nb, const int slot0, const int cur) {
    const int ntiles = nt * (nt - 1) / 2, n_tile_items = (ntiles + 1) / 2, n_items = n_tile_items + (nb + 1) / 2;
    const int ld_at = (int)(Ld - ring) + cur * FBS, dv_at = (int)(dv - ring) + cur * FBS;
    const double* Pc = P + cur * FBS;
    for (;;) {
      int it = 0;
      if (lane == 0) it = atomicAdd(q_next, 1);
      it = __builtin_amdgcn_readfirstlane(it);
      if (it >= n_items) break;
      if (it < n_tile_items) {
        int ti0 = 0, tj0 = 2 * it;
        while (tj0 > ti0) { tj0 -= ti0 + 1; ++ti0; }
        if (2 * it + 1 < ntiles) {
          int ti1 = ti0, tj1 = tj0 + 1;   // the tile after (ti0, tj0) in the same order
          if (tj1 > ti1) { ++ti1; tj1 = 0; }
          window_tiles(std::true_type{}, j, slot0, Pc, ti0 + 1, tj0 + 1, ti1 + 1, tj1 + 1);
        } else {
          window_tiles(std::false_type{}, j, slot0, Pc, ti0 + 1, tj0 + 1, ti0 + 1, tj0 + 1);
        }
      } else {
        const int b0 = 2 * (it - n_tile_items);
        for (int b = b0; b < b0 + 2 && b < nb; ++b) {
          double* col = Lc + (j + b) * W1;
          for (int d = lane; d < W1; d += 64) {
            const int row = b + d;
            const bool ok = row < nb || (row >= NB && row - NB < w);
            col[d] = ring[d == 0 ? dv_at + b : (ok ? ld_at + row * PS + b : zero_at)];
          }
        }
      }
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
      _Pragma("unroll") for (int k = 0; k < MT; ++k) {                          \
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
    const double* Pc = P + cur * FBS, * Usc = Us + cur * FBS;
    const int nxt = cur ^ 1;
    int slot1 = slot0 + NB;
    if (slot1 >= WS) slot1 -= WS;
    // the NB columns that enter the window at the end of this block: their raw entries are REQUESTED now, all of them before any
    // is used (a load consumed inside a divergent branch is waited for on the spot: seven round trips in a row), combined and
    // stored in S4
    double raw[NPRE][MT], rawc[NPRE];
    if (apply) {
      const int col = (int)j + WS + pcol;
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        const bool in = col < plim[q];
        rawc[q] = (in && pchain[q]) ? pchain[q][col] : 0.0;
#pragma unroll
        for (int k = 0; k < MT; ++k) raw[q][k] = (in && pbase[q][k]) ? pbase[q][k][col] : 0.0;
      }
      WSTAMP(0);
      // ---- S2: what block j + NB waits for: the first tile column of the window update and the right-hand side
      if (wave < nt) window_tiles(std::false_type{}, j, slot0, Pc, wave, 0, wave, 0);   // (nt <= 8 = NT / 64)
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
      if (!ahead) drain_items(j, nb, slot0, cur);
      lds_barrier_w();
      WSTAMP(2);
      if (!ahead) {
        BAND_REFILL();
        if (tid == 0) *q_next = 0;
        lds_barrier_w();
        WSTAMP(4);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        rawc[q] = 0.0;
#pragma unroll
        for (int k = 0; k < MT; ++k) raw[q][k] = 0.0;
      }
    }
    // ---- S3: block j + NB factorised by the first waves, beside the rest of block j's work on the others
    const unsigned long long tw0 = dbg ? __builtin_readcyclecounter() : 0;
    if (wave < npw && j + NB < n) factor_block(j + NB, slot1, P + nxt * FBS, Ld + nxt * FBS, dv + nxt * FBS, Us + nxt * FBS);
    if (ahead && apply) drain_items(j, nb, slot0, cur);
    if (dbg) twork += __builtin_readcyclecounter() - tw0;   // (this wave's own work in S3, without the wait at the barrier)
    lds_barrier_w();
    WSTAMP(3);
    if (ahead && apply) {
      BAND_REFILL();
      if (tid == 0) *q_next = 0;
      lds_barrier_w();
      WSTAMP(4);
    }
    slot0 = slot1;
    cur = nxt;
  }
#undef BAND_REFILL
  const bool failed = misc[0] != 0.0;
  if (wave == 0) {
    // log det Q = -2 sum log(1 / L_KK): the lanes' shares in lane order
    const double share = (lane < NB) ? log(ld_mant) + (double)ld_exp * 0.69314718055994530942 : 0.0;
    double total = 0.0;
#pragma unroll
    for (int K = 0; K < NB; ++K) total += readlane_d(share, K);
    if (tid == 0) {
      if (logdet) logdet[c] = -2.0 * total;
      if (failed) atomicMin((unsigned long long*)bad, (unsigned long long)c);
    }
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
  // ---- backward pass L' x = t, NB columns per step, ONE barrier per step.  xs / ms: the last WB solutions, slot = column % WB.
  // Column b of block J needs sum_d L[j+b+d][j+b] x[j+b+d] over the rows behind it.  The rows from block J + 2 on ("far":
  // d >= 2 NB - b) are known a whole step before block J is solved: waves 1 .. 7 add them up for block J - 1 WHILE wave 0 solves
  // block J -- the rows of block J + 1 ("near"), the far sums handed over through LDS, and the block's own triangle across its
  // lanes.  Every sum is taken in a fixed order (partial sums per quad of threads, then in quad order): results do not depend on
  // timing.
  // (the ring of solutions has w + 2 NB slots, not w + NB: block J + 1 is still being copied out of it by wave 1 while wave 0
  // writes block J -- with w < NB the two blocks would share slots in a ring of w + NB)
  const int WB = w + 2 * NB;
  double* xs = sm;
  double* ms = xs + WB;
  double* Sx = ms + WB;        // NB: sum over the rows behind the block, per column
  double* Sm = Sx + NB;
  double* Lb = Sm + NB;        // NB x PS: the block's own triangle of the factor, Lb[a][b] = L[j+a][j+b], a > b; dinv on the diagonal
  constexpr int TPF = (NT - 64) / NB;   // threads per column for the far rows
  constexpr int QPC = TPF / 4;          // their quads per column
  constexpr int NPF = (WMAX_NB > NB) ? (WMAX_NB - NB + TPF - 1) / TPF : 1;  // far entries per thread (d = 2 NB - b .. w: at most w - NB of them, b = NB - 1)
  constexpr int LPC = 64 / NB;          // lanes of wave 0 per column for the block's own and the near rows
  constexpr int KN = 2 * NB / LPC;      // entries per lane there (d = 0 .. 2 NB - 1 - b)
  static_assert(TPF % 4 == 0 && TPF * NB == NT - 64, "quads of far threads do not straddle columns");
  const int dump_b = NB * PS + 4 * NB * QPC + lane;  // (behind Part: a slot per lane to write to in vain)
  const int diag_b = NB * PS + 4 * NB * QPC + 64;    // (and the block's 1 / L_jj: NB of them)
  double* Part = Lb + NB * PS;  // [2][2][NB][QPC]: the far sums of the quads, for blocks of even and odd number, draw and mean
  for (int t = tid; t < 2 * WB; t += NT) sm[t] = 0.0;
  for (int t = tid; t < NB * PS; t += NT) Lb[t] = 0.0;
  if (tid < NB) Lb[diag_b + tid] = 0.0;
  __syncthreads();
  const int64_t nblk = (n + NB - 1) / NB;
  const int ft = tid - 64, fb = ft >= 0 ? ft / TPF : 0, fq = ft >= 0 ? ft % TPF : 0;   // far threads: column, place in the column
  const int gb = lane / LPC, gg = lane % LPC;                                         // wave 0: column, place in the column
  const bool t_lane = wave == 0 && lane < 2 * NB && (lane < NB || mc);
  // entries of the factor and of the right-hand side, asked for a step before they are used
  // (every request is an UNCONDITIONAL load -- entries that do not exist are asked for at the chain's first factor entry and
  // masked when they are used: a load under a condition becomes a branch with a wait for the data behind it, one round trip to
  // memory after the other)
  double fnext[NPF], nnext[KN], tnext = 0.0;
  unsigned fmask_next = 0, nmask_next = 0;   // bit k: entry k exists; bit 31: the right-hand side entry does
  const int ni = (int)n;
  // what does not change from block to block: which of this thread's entries lie inside the band (bit k), and the rows they meet
  // counted from the block's first column (far rows: 2 NB + fq + TPF k, the same for every column of the block)
  unsigned fband = 0, nband = 0;
#pragma unroll
  for (int k = 0; k < NPF; ++k) fband |= (2 * NB - fb + fq + TPF * k <= w) ? 1u << k : 0u;
#pragma unroll
  for (int k = 0; k < KN; ++k) nband |= (gg + LPC * k <= w && gb + gg + LPC * k < 2 * NB) ? 1u << k : 0u;
  const double* const tsrc = (lane < NB || !mc) ? xc : mc;
  auto request = [&](const int Jt) {  // block Jt's entries for this thread's role (Jt == -1: none)
    const int j0 = Jt * NB;
    if (wave == 0) {
      const double* col = Lc + (int64_t)(j0 + gb) * W1;
      nmask_next = 0;
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const int d = gg + LPC * k;
        const bool ok = ((nband >> k) & 1) && Jt >= 0 && j0 + gb + d < ni;
        nmask_next |= ok ? 1u << k : 0u;
        const double* at = ok ? col + d : Lc;
        nnext[k] = *at;
      }
      const int tb = lane & (NB - 1);
      const bool ok = Jt >= 0 && t_lane && j0 + tb < ni;
      nmask_next |= ok ? 1u << 31 : 0u;
      tnext = tsrc[ok ? j0 + tb : 0];
    } else {
      const double* col = Lc + (int64_t)(j0 + fb) * W1 + (2 * NB - fb + fq);
      fmask_next = 0;
#pragma unroll
      for (int k = 0; k < NPF; ++k) {
        const bool ok = ((fband >> k) & 1) && Jt >= 0 && j0 + 2 * NB + fq + TPF * k < ni;
        fmask_next |= ok ? 1u << k : 0u;
        const double* at = ok ? col + TPF * k : Lc;
        fnext[k] = *at;
      }
    }
  };
  // far sums of block Jt (waves 1 .. 7; slotT: slot of its first column)
  auto far_part = [&](const int Jt, const int slotT) {
    double fl[NPF];
    const unsigned fm = fmask_next;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (written out: see the note at the forward pass' entering columns)
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      fl[k] = fnext[k];
      asm volatile("" : "+v"(fl[k]));
    }
    request(Jt - 1);
    double px = 0.0, pm = 0.0;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      // an entry that exists meets a row no further than w + NB - 1 = w + NB - 1 from the block's first column: one wrap at most
      int sl = slotT + 2 * NB + fq + TPF * k;
      if (sl >= WB) sl -= WB;
      const bool ok = (fm >> k) & 1;
      sl = ok ? sl : 0;
      const double l = ok ? fl[k] : 0.0;
      px = fma(l, xs[sl], px);
      if (mc) pm = fma(l, ms[sl], pm);
    }
    px += dpp_d<0xB1>(px);  // quad: xor 1
    px += dpp_d<0x4E>(px);  //       xor 2
    if (mc) {
      pm += dpp_d<0xB1>(pm);
      pm += dpp_d<0x4E>(pm);
    }
    if ((fq & 3) == 0) {
      double* dst = Part + (Jt & 1) * 2 * NB * QPC + fb * QPC + (fq >> 2);
      dst[0] = px;
      if (mc) dst[NB * QPC] = pm;
    }
  };
  // a solved block goes to global memory a step later, from the ring, by wave 1: a store is not acknowledged for a microsecond,
  // and the wave that solves waits for "all its memory operations" at the top of every step (its requests of a step ago)
  auto store_block = [&](const int64_t Jb, const int slotB) {
    const int b = lane & (NB - 1);
    const int64_t i = Jb * NB + b;
    if (lane < 2 * NB && i < n && (lane < NB || mc)) {
      int sl = slotB + b;
      if (sl >= WB) sl -= WB;
      (lane < NB ? xc : mc)[i] = (lane < NB ? xs : ms)[sl];
    }
  };
  int slotJ = (int)(((nblk - 1) * NB) % WB);  // slot of the block's first column, kept incrementally
  request((int)nblk - 1);
  if (wave > 0) far_part((int)nblk - 1, slotJ);    // (nothing behind the last block but zeros: its sums are written all the same)
  lds_barrier_w();
  WSTAMP(6);
  for (int64_t J = nblk - 1; J >= 0; --J) {
    const int64_t j = J * NB;
    const int nb = (int)((n - j < NB) ? n - j : NB);
    int slotP = slotJ - NB;                   // slot of block J - 1's first column
    if (slotP < 0) slotP += WB;
    const unsigned long long tb0 = dbg ? __builtin_readcyclecounter() : 0;
    if (wave > 0) {
      if (J > 0) far_part((int)J - 1, slotP);
      if (wave == 1 && J + 1 < nblk) store_block(J + 1, slotJ + NB >= WB ? slotJ + NB - WB : slotJ + NB);
    } else {
      double nl[KN];
      const unsigned nm = nmask_next;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        nl[k] = nnext[k];
        asm volatile("" : "+v"(nl[k]));
      }
      double tval = tnext;
      asm volatile("" : "+v"(tval));
      if (!(nm >> 31)) tval = 0.0;
      request((int)J - 1);
      // the block's own triangle to LDS, the near rows (block J + 1) against the solutions, LPC lanes per column
      double px = 0.0, pm = 0.0;
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const int d = gg + LPC * k, r = gb + d;
        const bool ok = (nm >> k) & 1;
        const bool mine = r < NB;                              // inside the block
        // (rows of entries that exist lie no further than w + NB - 1 from the block's first column)
        int sl = slotJ + r;
        if (sl >= WB) sl -= WB;
        sl = (ok && !mine) ? sl : 0;
        const double l = (ok && !mine) ? nl[k] : 0.0;
        px = fma(l, xs[sl], px);
        if (mc) pm = fma(l, ms[sl], pm);
        Lb[(ok && mine) ? (d == 0 ? diag_b + gb : r * PS + gb) : dump_b] = nl[k];   // (d == 0: 1 / L_jj, kept apart)
      }
      // + the far sums of this lane's share of the quads
      const double* part = Part + (int)(J & 1) * 2 * NB * QPC + gb * QPC;
#pragma unroll
      for (int k = gg; k < QPC; k += LPC) {
        px += part[k];
        if (mc) pm += part[NB * QPC + k];
      }
      px += dpp_d<0xB1>(px);
      px += dpp_d<0x4E>(px);
      if (LPC == 8) px += dpp_d<0x141>(px);  // (quads agree by now: mirroring a half row swaps them)
      if (mc) {
        pm += dpp_d<0xB1>(pm);
        pm += dpp_d<0x4E>(pm);
        if (LPC == 8) pm += dpp_d<0x141>(pm);
      }
      if (gg == 0) { Sx[gb] = px; Sm[gb] = pm; }
      // the NB x NB triangle by the first 2 NB lanes: lane b (draw) and lane NB + b (mean) keep their own unknown; the unknowns
      // are finished from the last one up and handed to the lanes in front by v_readlane.  (One wave: its LDS operations are
      // carried out in order, no barrier between the writes above and the reads below.)
      const bool is_m = lane >= NB;
      const int b = lane & (NB - 1);
      const bool on = lane < 2 * NB && b < nb && (!is_m || mc);
      const double sb = (is_m ? Sm : Sx)[b];
      double acc = on ? tval - sb : 0.0;
      const double dinv_b = Lb[diag_b + b];
      // this lane's column of the triangle, read in one go and without a condition: what the steps above never write -- on and
      // above the diagonal, outside the band, rows beyond the chain's end (the first block solved is the only short one) -- is
      // zero from the start.  An unknown is complete once the unknowns behind it have been taken off: lane b's acc does not
      // change after step b + 1 (its entries against the unknowns in front of it are those zeros).
      double lcol[NB];
#pragma unroll
      for (int a = 0; a < NB; ++a) lcol[a] = Lb[a * PS + b];
      if (mc) {
#pragma unroll
        for (int a = NB - 1; a > 0; --a) {
          const double fin = acc * dinv_b;   // (complete in lanes a and NB + a)
          const double xa = readlane_d(fin, a), ma = readlane_d(fin, NB + a);
          acc = fma(-lcol[a], is_m ? ma : xa, acc);
        }
      } else {
#pragma unroll
        for (int a = NB - 1; a > 0; --a) acc = fma(-lcol[a], readlane_d(acc * dinv_b, a), acc);
      }
      const double xv = acc * dinv_b;
      if (on) {
        int sl = slotJ + b;
        if (sl >= WB) sl -= WB;
        (is_m ? ms : xs)[sl] = xv;
      }
    }
    slotJ = slotP;
    if (dbg) tback += __builtin_readcyclecounter() - tb0;   // (this wave's own work, without the wait at the barrier)
    lds_barrier_w();
    WSTAMP(7);
  }
  if (wave == 1) store_block(0, slotJ + NB >= WB ? slotJ + NB - WB : slotJ + NB);
  if (dbg && blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) dbg[i] = tacc[i];
  if (dbg && blockIdx.x == 0 && lane == 0) {
    dbg[8 + wave] = twork;    // S3 per wave
    dbg[16 + wave] = tback;   // a step of the backward pass per wave
  }
#undef WSTAMP
}

}  // namespace

// LDS bytes of a block size NB at bandwidth w with NT threads: the larger of the two passes' images (the factor phase's, except for
// bands of a few entries)
static size_t blocked_lds(int w, int NB, int NT = 512) {
  const size_t W1 = (size_t)w + 1, WS = (size_t)w + NB, WP = ((size_t)w + 15) & ~(size_t)15, PS = (size_t)NB + 1;
  const size_t fwd = WS * W1 + WS + 2 * (WP * PS + (size_t)NB * PS + 2 * (size_t)NB) + 2 + 64 + 2;
  // the backward pass lays its own image over the same memory: two solution rings of w + 2 NB, the block's sums and triangle, the
  // far sums of the quads (two blocks, draw and mean), a slot per lane to write to in vain, the block's 1 / L_jj
  const size_t qpc = (size_t)(NT - 64) / NB / 4;
  const size_t bwd = 2 * ((size_t)w + 2 * NB) + 2 * (size_t)NB + (size_t)NB * PS + 4 * (size_t)NB * qpc + 64 + (size_t)NB;
  return (fwd > bwd ? fwd : bwd) * sizeof(double);
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
  const bool few = T.n_terms <= 2;
#define OMC_BLOCKED_LAUNCH(NB_, NT_, MT_, WPE_)                                                                                       \
  hipLaunchKernelGGL((k_band_blocked<NB_, NT_, MT_, WPE_>), dim3((unsigned)ctx->n_chains), dim3(NT_), blocked_lds(w, NB_, NT_), ctx->stream, \
                     ctx->n_chains, ctx->chain_offset, n, w, T, rhs_chain, ld_rhs, z_inject, ld_z, key, Lws, x, ld_x, mean, ld_mean,  \
                     logdet, ctx->d_bad_chain, ctx->stamps)
  // Which form, by what fits a CU (measured on 10 000-node lattices, profiles/r04q_band.txt):
  //  * bands narrower than a block take four waves per chain (one tile, one factorising wave); with more chains than three
  //    workgroups per CU hold, the form compiled for 128 registers puts four there (1024 chains at w = 8: 8.0 -> 4.6 ms);
  //  * bands up to ~64 on more chains than CUs: 8 columns per step at 128 registers -- four waves per chain and four workgroups
  //    to a CU while their LDS fits (w <= 55: 1024 chains at w = 32 15.6 -> 6.2 ms, at w = 16 15.6 -> 5.6 ms), eight waves and
  //    two to a CU beyond (w = 64: 16.4 -> 14.6 ms); slower where one workgroup per CU is all there is (3.9 -> 4.9 ms at 256
  //    chains);
  //  * otherwise 16 columns per step, eight waves, one workgroup per CU (8 columns where the window would not fit the LDS).
  // "band_blocked_threads": 0 this choice; 512 eight waves and no register limit whatever the shape; 4 / 16 / 8 the 128-register
  // forms wherever they apply (A/B runs and tests).  Only the forms compiled for two terms have the 128-register variants.
  int dev_cus = 256;
  hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
  const int forced = ctx->band_blocked_threads;
  if (w <= 15 && forced != 512 && forced != 8) {
    if (few && (forced == 4 || (forced == 0 && ctx->n_chains > 3 * (int64_t)dev_cus))) OMC_BLOCKED_LAUNCH(16, 256, 2, 4);
    else if (few) OMC_BLOCKED_LAUNCH(16, 256, 2, 1);
    else OMC_BLOCKED_LAUNCH(16, 256, OMC_MAX_TERMS, 1);
    return true;
  }
  if (few && w <= 64 && 4 * blocked_lds(w, 8, 256) <= limit && (forced == 16 || (forced == 0 && ctx->n_chains > (int64_t)dev_cus))) {
    OMC_BLOCKED_LAUNCH(8, 256, 2, 4);   // 8 columns per step, four waves, four workgroups to a CU
    return true;
  }
  if (few && 2 * blocked_lds(w, 8) <= limit && (forced == 8 || (forced == 0 && ctx->n_chains > (int64_t)dev_cus))) {
    OMC_BLOCKED_LAUNCH(8, 512, 2, 4);   // 8 columns per step, eight waves, two workgroups to a CU
    return true;
  }
  if (w <= BAND_W16_MAX && blocked_lds(w, 16) <= limit) {
    if (few) OMC_BLOCKED_LAUNCH(16, 512, 2, 1);
    else OMC_BLOCKED_LAUNCH(16, 512, OMC_MAX_TERMS, 1);
    return true;
  }
  if (blocked_lds(w, 8) <= limit) {
    if (few) OMC_BLOCKED_LAUNCH(8, 512, 2, 1);
    else OMC_BLOCKED_LAUNCH(8, 512, OMC_MAX_TERMS, 1);
    return true;
  }
#undef OMC_BLOCKED_LAUNCH
  return false;
}
