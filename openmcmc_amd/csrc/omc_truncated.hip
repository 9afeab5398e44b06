// Truncated Gaussian full conditional: gmrf.gibbs_canonical_truncated_normal (gmrf.py:201-266), the branch
// NormalNormal.sample takes when the parameter's prior has domain limits (sampler.py:199-205).
//
// One scan of single-site Gibbs updates in index order, each from its univariate truncated normal
//     x_i ~ N_[lo_i, hi_i]( (b_i - sum_j Q_ij x_j + Q_ii x_i) / Q_ii , 1 / Q_ii )
// with x_j already updated for j < i (gmrf.py:254-264).  The scan is sequential in i by definition (a
// reordered or coloured scan is a different Markov kernel, not the reference's), so the parallelism is
// over chains: tridiagonal precisions take one lane per chain, dense ones one wave per chain with the
// row product spread over the lanes.  Also Normal.check_domain_response (location_scale.py:169-188).
#include <math.h>

#include "omc_common.h"
#include "omc_truncnorm.h"

static inline unsigned grid1(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

struct TruncTerms {
  int n_terms;
  const double* diag[OMC_MAX_TERMS];
  const double* off[OMC_MAX_TERMS];
  const double* rhs[OMC_MAX_TERMS];
  const double* scale[OMC_MAX_TERMS];
};

__device__ __forceinline__ double trunc_uniform(const double* u_in, int64_t ld, int64_t c, int64_t i, const omc_rng_key& key,
                                                int64_t gc) {
  if (u_in) return u_in[c * ld + i];
  const uint4 w = omc_rng_block(key, gc, (uint32_t)(i >> 1));
  return (i & 1) ? omc_u53(w.z, w.w) : omc_u53(w.x, w.y);
}

// One wave = 64 chains, one lane each, 64 sites at a time:
//   * the block's x (and the per-chain right-hand side, if any) comes in as 64 coalesced rows through an LDS transpose
//     tile and leaves the same way once the block is done.  A site-by-site store followed by the next site's load
//     costs a store acknowledgement per site on this hardware (a load returns behind every older store: 2 us a site,
//     20 ms a scan); the shared vectors of the block are staged in LDS too and read as broadcasts;
//   * a site's draw is x = mean + sd * t with t the standard truncated-normal quantile.  When the limits are far out
//     (the usual case) t = Phi^-1(u) does not depend on the mean, i.e. not on the neighbours: it is evaluated ahead of
//     the one dependent chain of the scan (mean_i needs x_{i-1}) instead of inside it; a site near a limit takes the full
//     log-space route as before.
// Round 4: NINE waves per 64 chains.  A wave alone on its SIMD issues one instruction per ~8.5 cycles whatever their
// dependences, and 250 of a site's ~290 instructions (Philox rounds, the far-limits quantile, 1/Q_ii and its square root) do not
// depend on the state: the sites of a 16-site round are split over eight producing waves (two each: the same statement-by-statement
// vector code as before) and left in LDS, while wave 0 walks the dependent chain (mean_i needs x_{i-1}: ~30 instructions a
// site) of the round made in the step before -- a two-stage pipeline with a barrier per step.  Same formulas on the same values in the same order: results are bit-identical to the one-wave form.
#define TG_LD 65
#define TG_BLK 16   // sites per production / scan round (2 per producing wave)
#define TG_NW 9     // waves: wave 0 walks the dependent chain, waves 1 .. 8 produce
#define TG_NF 5     // fields kept per site and chain: zf, uu, v, sd, b
template <bool INJ, int NT>
__global__ void __launch_bounds__(64 * TG_NW) k_tridiag_gibbs_truncated(int64_t C, int64_t chain_offset, int64_t n, TruncTerms T,
                                                                const double* rhs_chain, int64_t ld_rhs, const double* lower,
                                                                const double* upper, const double* u_in, int64_t ld_u,
                                                                omc_rng_key key, double* x, int64_t ld_x, long long* bad) {
  extern __shared__ double sm[];
  double* xt = sm;                                   // [64 chains][65]: x of the block's 64 sites + the first of the next
  double* rt = sm + 64 * TG_LD;                      // [64][65] per-chain right-hand side (only if rhs_chain)
  double* stage = rt + (rhs_chain ? 64 * TG_LD : 0); // [3 n_terms + 2][64]: diag / off / rhs of every term, lower, upper
  double* fld = stage + (3 * NT + 2) * 64;           // [2][TG_BLK][TG_NF][64]: what the production leaves for the scan, two rounds
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int64_t c = c0 + lane;
  const bool live = c < C;
  const int64_t cc = live ? c : C - 1;
  double s[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) s[k] = (k < T.n_terms && T.scale[k]) ? T.scale[k][cc] : 1.0;
  double* st_lo = stage + 3 * NT * 64;
  double* st_hi = st_lo + 64;
  double x_prev = 0.0, off_prev = 0.0;  // x_{i-1} (already updated) and Q_{i,i-1}
  bool fail = false;
  const int64_t gc = chain_offset + cc;
  const double* urow = u_in ? u_in + cc * ld_u : nullptr;
  // Everything of a site that does not depend on the state -- uniform, far-limits quantile, row of Q, 1/Q_ii and
  // 1/sqrt(Q_ii) -- for TG_U sites at a time in one straight piece of code (vectors of TG_U: statement by statement): the wave is alone on its SIMD, so the only
  // thing that can fill the latency of one site's long dependent chains (Philox rounds, two Horner chains, log, sqrt) is
  // the same work of its neighbours
  constexpr int TG_U = 2;  // sites per producing wave and round: eight waves x 2 = TG_BLK
  const bool single = n == 1;
  struct Site { double uu, zf, a, o, b, lo, hi, v, sd; bool zok; };
  for (int64_t i0 = 0; i0 < n; i0 += 64) {
    const int len = (int)((n - i0 < 64) ? n - i0 : 64);
    // ---- block in: rows of 64 sites (coalesced), one more column for the last site's right neighbour
    for (int r = wave; r < 64; r += TG_NW) {
      const bool okr = c0 + r < C && lane < len;
      xt[r * TG_LD + lane] = okr ? x[(c0 + r) * ld_x + i0 + lane] : 0.0;
      if (rhs_chain) rt[r * TG_LD + lane] = okr ? rhs_chain[(c0 + r) * ld_rhs + i0 + lane] : 0.0;
    }
    if (wave == 1) xt[lane * TG_LD + 64] = (live && i0 + 64 < n) ? x[c * ld_x + i0 + 64] : 0.0;
    if (wave == 0) {
      const int64_t i = i0 + lane;
      const bool oki = lane < len;
      for (int k = 0; k < NT; ++k) {
        stage[(3 * k + 0) * 64 + lane] = (oki && T.diag[k]) ? T.diag[k][i] : 1.0;
        stage[(3 * k + 1) * 64 + lane] = (oki && T.off[k] && i + 1 < n) ? T.off[k][i] : 0.0;
        stage[(3 * k + 2) * 64 + lane] = (oki && T.rhs[k]) ? T.rhs[k][i] : 0.0;
      }
      st_lo[lane] = (oki && lower) ? lower[i] : -INFINITY;
      st_hi[lane] = (oki && upper) ? upper[i] : INFINITY;
    }
    __syncthreads();
    double* xrow = xt + lane * TG_LD;
    const double* rrow = rt + lane * TG_LD;
    // The block's rounds as a two-stage pipeline: in step k the eight producing waves make round k while wave 0 walks round
    // k - 1 (made in the step before, in the other buffer); a barrier between steps.  The pipeline drains at the block's end:
    // the next block's staged vectors replace this one's.
    const int n_rounds = (len + TG_BLK - 1) / TG_BLK;
    for (int step = 0; step <= n_rounds; ++step) {
      if (wave > 0 && step < n_rounds) {
      const int tb = step * TG_BLK;
      double* const fbuf = fld + (int64_t)(step & 1) * TG_BLK * TG_NF * 64;
      const int t0 = tb + TG_U * (wave - 1);  // this wave's sites of the round
      Site S[TG_U];
      // uniforms: sites 2m and 2m+1 share a Philox block (t0 and i0 are multiples of TG_U): TG_U / 2 blocks, their rounds interleaved
      typedef omc_dv<TG_U> dvu;
      dvu u4;
      if (INJ) {
#pragma unroll
        for (int q = 0; q < TG_U; ++q) u4[q] = urow[i0 + ((t0 + q < len) ? t0 + q : len - 1)];
      } else {
        const uint32_t blk = (uint32_t)((i0 + t0) >> 1);
        const uint32_t c2 = (uint32_t)gc, c3 = key.c3_base | ((uint32_t)((uint64_t)gc >> 32) & 0xffu) << 16;
        uint32_t w0[TG_U / 2], w1[TG_U / 2], w2[TG_U / 2], w3[TG_U / 2], k0[TG_U / 2], k1[TG_U / 2];
#pragma unroll
        for (int b = 0; b < TG_U / 2; ++b) { w0[b] = blk + b; w1[b] = key.c1; w2[b] = c2; w3[b] = c3; k0[b] = key.k0; k1[b] = key.k1; }
#pragma unroll
        for (int r = 0; r < 10; ++r)
#pragma unroll
          for (int b = 0; b < TG_U / 2; ++b) omc_philox_round_r(r, w0[b], w1[b], w2[b], w3[b], k0[b], k1[b]);
#pragma unroll
        for (int b = 0; b < TG_U / 2; ++b) { u4[2 * b] = omc_u53(w0[b], w1[b]); u4[2 * b + 1] = omc_u53(w2[b], w3[b]); }
      }
      dvu usafe;
#pragma unroll
      for (int q = 0; q < TG_U; ++q) {
        S[q].uu = u4[q];
        S[q].zok = u4[q] > 1e-15 && u4[q] < 1.0 - 1e-15;
        usafe[q] = S[q].zok ? u4[q] : 0.5;
      }
      const dvu z4 = omc_ndtri_as241_nbv<TG_U>(usafe);
      dvu a4, o4, b4;
#pragma unroll
      for (int q = 0; q < TG_U; ++q) {
        const int t = (t0 + q < len) ? t0 + q : len - 1;  // (a short last group repeats its last site: unused)
        double a = 0.0, o = 0.0, b = rhs_chain ? rrow[t] : 0.0;
#pragma unroll
        for (int k = 0; k < NT; ++k) {  // (staged with the defaults of absent vectors: 1, 0, 0; NT is a template parameter:
          a = fma(s[k], stage[(3 * k + 0) * 64 + t], a);  //  a run-time count made every term of every site its own
          o = fma(s[k], stage[(3 * k + 1) * 64 + t], o);  //  basic block with an exposed LDS round trip)
          b = fma(s[k], stage[(3 * k + 2) * 64 + t], b);
        }
        a4[q] = a; o4[q] = o; b4[q] = b;
        S[q].lo = st_lo[t]; S[q].hi = st_hi[t];
      }
      // 1/a and 1/sqrt(a) by the refined hardware reciprocal / reciprocal square root, the four together
      dvu asafe;
#pragma unroll
      for (int q = 0; q < TG_U; ++q) asafe[q] = (a4[q] > 0.0) ? a4[q] : 1.0;
      const dvu v4 = omc_rcp_nrv<TG_U>(asafe);
      dvu sd4;
      {
        dvu g;
#pragma unroll
        for (int q = 0; q < TG_U; ++q) g[q] = __builtin_amdgcn_rsq(asafe[q]);
        const dvu h = dvu(0.5) * a4;
        dvu r = g;
        r = omc_fmav<TG_U>(r, omc_fmav<TG_U>(-h * r, r, dvu(0.5)), r);
        r = omc_fmav<TG_U>(r, omc_fmav<TG_U>(-h * r, r, dvu(0.5)), r);
        sd4 = r;
      }
#pragma unroll
      for (int q = 0; q < TG_U; ++q) {
        S[q].zf = z4[q];
        S[q].a = a4[q]; S[q].o = o4[q]; S[q].b = b4[q];
        S[q].v = (a4[q] > 0.0) ? v4[q] : 1.0;
        S[q].sd = sd4[q];
      }
      // what the scan needs of them goes to LDS (a, o and the limits it takes from the staged vectors itself)
#pragma unroll
      for (int q = 0; q < TG_U; ++q) {
        double* f = fbuf + (int64_t)((TG_U * (wave - 1) + q) * TG_NF) * 64 + lane;
        f[0 * 64] = S[q].zf; f[1 * 64] = S[q].uu; f[2 * 64] = S[q].v; f[3 * 64] = S[q].sd; f[4 * 64] = S[q].b;
      }
      }  // producers
      // ---- the scan proper: one dependent chain, wave 0, one round behind
      if (wave == 0 && step > 0) {
        const int tb = (step - 1) * TG_BLK;
        const double* const fbuf = fld + (int64_t)((step - 1) & 1) * TG_BLK * TG_NF * 64;
#pragma unroll 4
        for (int qq = 0; qq < TG_BLK; ++qq) {
          const int t = tb + qq;
          if (t < len) {
            const double* f = fbuf + (int64_t)(qq * TG_NF) * 64 + lane;
            const double zf = f[0 * 64], uu = f[1 * 64], v = f[2 * 64], sd = f[3 * 64], bb = f[4 * 64];
            const bool zok = uu > 1e-15 && uu < 1.0 - 1e-15;
            double a = 0.0, o = 0.0;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
              a = fma(s[k], stage[(3 * k + 0) * 64 + t], a);
              o = fma(s[k], stage[(3 * k + 1) * 64 + t], o);
            }
            const double x_cur = xrow[t], x_next = xrow[t + 1];
            if (!(a > 0.0)) fail = true;
            const double inv_sd = a * sd;  // sqrt(a)
            // gmrf.py:255-262: v_i * (b_i - Q[i,:] @ x + Q_ii x_i), row product in column order; n = 1: b v (gmrf.py:244-247)
            const double row = fma(o, x_next, fma(a, x_cur, off_prev * x_prev));
            const double mean = single ? bb * v : v * ((bb - row) + a * x_cur);
            const double as = (st_lo[t] - mean) * inv_sd, bs = (st_hi[t] - mean) * inv_sd;
            double xi = fma(zf, sd, mean);  // omc_truncnorm_ppf's far-limits branch: t = Phi^-1(u), x = t * sd + mean
            if (!(as < -13.0 && bs > 13.0 && zok)) xi = omc_truncnorm_ppf(uu, as, bs) * sd + mean;
            xrow[t] = xi;
            x_prev = xi;
            off_prev = o;
          }
        }
      }
      __syncthreads();
    }
    // ---- block out
    for (int r = wave; r < 64; r += TG_NW)
      if (c0 + r < C && lane < len) x[(c0 + r) * ld_x + i0 + lane] = xt[r * TG_LD + lane];
    __syncthreads();
  }
  if (wave == 0 && fail && live) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

// banded precision of bandwidth w (Q_c = sum_k s_k[c] M_k, M_k in the band storage of omc_band_terms: band[d*n + i] =
// M[i+d, i]): the same scan, one lane per chain.  Row i of Q touches x_{i-w..i+w}; the sub-diagonal part of the row comes
// from the columns i-d (band[d*n + i-d]), the super-diagonal part from column i itself (band[d*n + i]).  The row product
// is accumulated in column order like the reference's Q[i, :] @ x (gmrf.py:258).
struct TruncBand {
  int n_terms;
  const double* band[OMC_MAX_TERMS];
  int bw[OMC_MAX_TERMS];
  const double* rhs[OMC_MAX_TERMS];
  const double* scale[OMC_MAX_TERMS];
};

__global__ void k_band_gibbs_truncated(int64_t C, int64_t chain_offset, int64_t n, int w, TruncBand T, const double* rhs_chain,
                                       int64_t ld_rhs, const double* lower, const double* upper, const double* u_in, int64_t ld_u,
                                       omc_rng_key key, double* x, int64_t ld_x, long long* bad) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) s[k] = (k < T.n_terms && T.scale[k]) ? T.scale[k][c] : 1.0;
  double* xc = x + c * ld_x;
  bool fail = false;
  for (int64_t i = 0; i < n; ++i) {
    double a = 0.0, b = rhs_chain ? rhs_chain[c * ld_rhs + i] : 0.0, row = 0.0;
#pragma unroll
    for (int k = 0; k < OMC_MAX_TERMS; ++k)
      if (k < T.n_terms) {
        a = fma(s[k], T.band[k] ? T.band[k][i] : 1.0, a);
        if (T.rhs[k]) b = fma(s[k], T.rhs[k][i], b);
      }
    // columns i-w .. i-1 (already updated), then i, then i+1 .. i+w
    for (int d = w; d >= 1; --d) {
      if (i - d < 0) continue;
      double q = 0.0;
#pragma unroll
      for (int k = 0; k < OMC_MAX_TERMS; ++k)
        if (k < T.n_terms && T.band[k] && d <= T.bw[k]) q = fma(s[k], T.band[k][(int64_t)d * n + (i - d)], q);
      row = fma(q, xc[i - d], row);
    }
    const double xi_old = xc[i];
    row = fma(a, xi_old, row);
    for (int d = 1; d <= w; ++d) {
      if (i + d >= n) break;
      double q = 0.0;
#pragma unroll
      for (int k = 0; k < OMC_MAX_TERMS; ++k)
        if (k < T.n_terms && T.band[k] && d <= T.bw[k]) q = fma(s[k], T.band[k][(int64_t)d * n + i], q);
      row = fma(q, xc[i + d], row);
    }
    if (!(a > 0.0)) fail = true;
    const double lo = lower ? lower[i] : -INFINITY, hi = upper ? upper[i] : INFINITY;
    double mean, sd;
    if (n == 1) {
      mean = b / a;
      sd = 1.0 / sqrt(a);
    } else {
      const double v = 1.0 / a;
      sd = sqrt(v);
      mean = v * ((b - row) + a * xi_old);
    }
    xc[i] = omc_truncated_normal_rv(mean, sd, lo, hi, trunc_uniform(u_in, ld_u, c, i, key, chain_offset + c));
  }
  if (fail) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

// dense precision, one wave per chain: Q_c = sum_k s_k[c] M_k assembled row by row on the fly
__global__ void __launch_bounds__(64) k_dense_gibbs_truncated(int64_t C, int64_t chain_offset, int64_t p, int n_terms,
                                                              const double* m0, const double* m1, const double* m2,
                                                              const double* m3, const double* s0, const double* s1,
                                                              const double* s2, const double* s3, const double* r0,
                                                              const double* r1, const double* r2, const double* r3,
                                                              const double* rhs_chain, int64_t ld_rhs, const double* lower,
                                                              const double* upper, const double* u_in, int64_t ld_u,
                                                              omc_rng_key key, double* x, int64_t ld_x, long long* bad) {
  extern __shared__ double xs[];  // the chain's vector
  const int64_t c = blockIdx.x;
  const int lane = threadIdx.x;
  const double* M[4] = {m0, m1, m2, m3};
  const double* R[4] = {r0, r1, r2, r3};
  double s[4] = {s0 ? s0[c] : 1.0, s1 ? s1[c] : 1.0, s2 ? s2[c] : 1.0, s3 ? s3[c] : 1.0};
  for (int64_t j = lane; j < p; j += 64) xs[j] = x[c * ld_x + j];
  __syncthreads();
  bool fail = false;
  for (int64_t i = 0; i < p; ++i) {
    double dot = 0.0;
    for (int64_t j = lane; j < p; j += 64) {
      double q = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < n_terms) q = fma(s[k], M[k] ? M[k][i * p + j] : (i == j ? 1.0 : 0.0), q);
      dot = fma(q, xs[j], dot);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) dot += __shfl_xor(dot, sh, 64);
    if (lane == 0) {
      double a = 0.0, b = rhs_chain ? rhs_chain[c * ld_rhs + i] : 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < n_terms) {
          a = fma(s[k], M[k] ? M[k][i * p + i] : 1.0, a);
          if (R[k]) b = fma(s[k], R[k][i], b);
        }
      if (!(a > 0.0)) fail = true;
      const double lo = lower ? lower[i] : -INFINITY, hi = upper ? upper[i] : INFINITY;
      double mean, sd;
      if (p == 1) {
        mean = b / a;
        sd = 1.0 / sqrt(a);
      } else {
        const double v = 1.0 / a;
        sd = sqrt(v);
        mean = v * ((b - dot) + a * xs[i]);
      }
      xs[i] = omc_truncated_normal_rv(mean, sd, lo, hi, trunc_uniform(u_in, ld_u, c, i, key, chain_offset + c));
    }
    __syncthreads();
  }
  for (int64_t j = lane; j < p; j += 64) x[c * ld_x + j] = xs[j];
  if (fail && lane == 0) atomicMin((unsigned long long*)bad, (unsigned long long)c);
}

// Normal.check_domain_response + log_p = -inf outside the domain (location_scale.py:162-188)
__global__ void k_domain_penalty(int64_t C, int64_t n, const double* x, int64_t ld, const double* lower, const double* upper,
                                 double* out) {
  const int64_t c = blockIdx.x;
  __shared__ int outside;
  if (threadIdx.x == 0) outside = 0;
  __syncthreads();
  int mine = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = x[c * ld + i];
    if ((lower && v < lower[i]) || (upper && v > upper[i])) mine = 1;
  }
  if (mine) outside = 1;
  __syncthreads();
  if (threadIdx.x == 0 && outside) out[c] = -INFINITY;
}

extern "C" {

omc_status omc_tridiag_gibbs_truncated(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms, const double* rhs_chain,
                                       int64_t ld_rhs, const double* lower, const double* upper, const double* u_inject,
                                       int64_t ld_u, uint64_t draw_index, double* x, int64_t ld_x) {
  if (!ctx || n < 1 || !terms || terms->n_terms < 1 || terms->n_terms > OMC_MAX_TERMS || !x || ld_x < n ||
      (rhs_chain && ld_rhs < n) || (u_inject && ld_u < n))
    return OMC_INVALID_ARG;
  for (int k = 0; k < terms->n_terms; ++k)
    if (terms->center_chain[k]) return OMC_UNSUPPORTED;  // (per-chain centres: the exact draw only)
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  TruncTerms T{};
  T.n_terms = terms->n_terms;
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    const bool on = k < terms->n_terms;
    T.diag[k] = on ? terms->diag[k] : nullptr;
    T.off[k] = on ? terms->off[k] : nullptr;
    T.rhs[k] = on ? terms->rhs[k] : nullptr;
    T.scale[k] = on ? terms->scale[k] : nullptr;
  }
  const size_t lds = (size_t)((rhs_chain ? 2 : 1) * 64 * TG_LD + (3 * T.n_terms + 2) * 64 + 2 * TG_BLK * TG_NF * 64) * sizeof(double);
#define OMC_TG_LAUNCH(INJv, NTv)                                                                                               \
  do {                                                                                                                          \
    if (lds > 48 * 1024)                                                                                                        \
      OMC_HIP_CHECK(hipFuncSetAttribute((const void*)(k_tridiag_gibbs_truncated<INJv, NTv>),                                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                 \
    hipLaunchKernelGGL((k_tridiag_gibbs_truncated<INJv, NTv>), dim3(grid1(ctx->n_chains, 64)), dim3(64 * TG_NW), lds, ctx->stream,      \
                       ctx->n_chains, ctx->chain_offset, n, T, rhs_chain, ld_rhs, lower, upper, u_inject, ld_u,                 \
                       omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), x, ld_x, ctx->d_bad_chain);                        \
  } while (0)
#define OMC_TG_TERMS(INJv)                                                                                                      \
  do {                                                                                                                          \
    if (T.n_terms == 1) OMC_TG_LAUNCH(INJv, 1);                                                                                 \
    else if (T.n_terms == 2) OMC_TG_LAUNCH(INJv, 2);                                                                            \
    else if (T.n_terms == 3) OMC_TG_LAUNCH(INJv, 3);                                                                            \
    else OMC_TG_LAUNCH(INJv, 4);                                                                                                \
  } while (0)
  if (u_inject) OMC_TG_TERMS(true); else OMC_TG_TERMS(false);
#undef OMC_TG_TERMS
#undef OMC_TG_LAUNCH
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_band_gibbs_truncated(omc_ctx* ctx, int64_t n, int64_t w, const omc_band_terms* terms, const double* rhs_chain,
                                    int64_t ld_rhs, const double* lower, const double* upper, const double* u_inject, int64_t ld_u,
                                    uint64_t draw_index, double* x, int64_t ld_x) {
  if (!ctx || n < 1 || w < 0 || w > 128 || !terms || terms->n_terms < 1 || terms->n_terms > OMC_MAX_TERMS || !x || ld_x < n ||
      (rhs_chain && ld_rhs < n) || (u_inject && ld_u < n))
    return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  TruncBand T{};
  T.n_terms = terms->n_terms;
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    const bool on = k < terms->n_terms;
    T.band[k] = on ? terms->band[k] : nullptr;
    T.bw[k] = on ? terms->bw[k] : 0;
    T.rhs[k] = on ? terms->rhs[k] : nullptr;
    T.scale[k] = on ? terms->scale[k] : nullptr;
    if (on && (T.bw[k] < 0 || T.bw[k] > w)) return OMC_INVALID_ARG;
  }
  hipLaunchKernelGGL(k_band_gibbs_truncated, dim3(grid1(ctx->n_chains, 64)), dim3(64), 0, ctx->stream, ctx->n_chains,
                     ctx->chain_offset, n, (int)w, T, rhs_chain, ld_rhs, lower, upper, u_inject, ld_u,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), x, ld_x, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_dense_gibbs_truncated(omc_ctx* ctx, int64_t p, const omc_dense_terms* terms, const double* rhs_chain,
                                     int64_t ld_rhs, const double* lower, const double* upper, const double* u_inject,
                                     int64_t ld_u, uint64_t draw_index, double* x, int64_t ld_x) {
  if (!ctx || p < 1 || p > 8192 || !terms || terms->n_terms < 1 || terms->n_terms > OMC_MAX_TERMS || !x || ld_x < p ||
      (rhs_chain && ld_rhs < p) || (u_inject && ld_u < p))
    return OMC_INVALID_ARG;
  if (terms->diag_chain) return OMC_UNSUPPORTED;  // per-chain diagonal (mixture prior) under a truncated conditional
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const double *m[4] = {0, 0, 0, 0}, *s[4] = {0, 0, 0, 0}, *r[4] = {0, 0, 0, 0};
  for (int k = 0; k < terms->n_terms; ++k) { m[k] = terms->mat[k]; s[k] = terms->scale[k]; r[k] = terms->rhs[k]; }
  hipLaunchKernelGGL(k_dense_gibbs_truncated, dim3((unsigned)ctx->n_chains), dim3(64), (size_t)p * sizeof(double),
                     ctx->stream, ctx->n_chains, ctx->chain_offset, p, (int)terms->n_terms, m[0], m[1], m[2], m[3], s[0], s[1],
                     s[2], s[3], r[0], r[1], r[2], r[3], rhs_chain, ld_rhs, lower, upper, u_inject, ld_u,
                     omc_make_key(ctx->seed, draw_index, OMC_RNG_UNIFORM), x, ld_x, ctx->d_bad_chain);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_domain_penalty(omc_ctx* ctx, int64_t n, const double* x, int64_t ld, const double* lower,
                              const double* upper, double* out) {
  if (!ctx || n < 1 || !x || ld < n || !out) return OMC_INVALID_ARG;
  if (!lower && !upper) return OMC_OK;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_domain_penalty, dim3((unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, ctx->n_chains, n, x, ld,
                     lower, upper, out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"
