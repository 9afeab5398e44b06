// Batched tridiagonal GMRF conjugate-Gibbs draw for gfx950:  x_c ~ N(Q_c^{-1} b_c, Q_c^{-1}).
//
// What it computes (per chain, fp64), following gmrf.sample_normal_canonical of the reference
// (gmrf.py:167-198; factor gmrf.py:489-520, solves gmrf.py:414-462, draw gmrf.py:29-61):
//     D_0 = a_0,  l_{i-1} = b_{i-1}/D_{i-1},  D_i = a_i - l_{i-1} b_{i-1}      (SuperLU's unpivoted LU pivots U_ii)
//     u_i = r_i - l_{i-1} u_{i-1}                                                (L w = b, w_i = u_i/sqrt(D_i))
//     g_i = u_i/D_i + z_i/sqrt(D_i),   x_i = g_i - l_i x_{i+1}                   (L'x = w + z)
// where L = L_lu diag(sqrt(D)) is the natural-order Cholesky factor, so x equals the reference's
// draw for the same z (not merely in distribution).
//
// Two kernels:
//   k_tridiag_serial : one lane per chain, any n, streams l through a workspace.  Simple; the
//                      cross-check for the fast path and the fallback for very long chains.
//   k_tridiag_seg    : the fast path.  A chain is cut into segments of M consecutive nodes, one
//                      lane per segment (up to 1024 lanes = one workgroup per chain); each lane
//                      keeps its 3*M working values in registers.  The three serial recurrences
//                      (pivots, forward substitution, backward substitution) are each solved as
//                      "local pass + scan over segments + local pass":
//                        pivots   : D_i = f_i(D_{i-1}) is a Moebius map; a division-free 2x2
//                                   product per segment + a scan gives every segment's incoming
//                                   pivot to ~1e-7..1e-15; Newton multiple-shooting on the true
//                                   recurrence (one affine scan per sweep) then makes the
//                                   segment joins consistent to a few ulp;
//                        forward / backward substitution: affine maps, one scan each.
//                      Global memory is touched only through wave-private LDS tiles that turn
//                      the lane-owns-M-consecutive-nodes layout into fully coalesced 512-B
//                      wave accesses.  HBM traffic per chain-update is the x store (8n B) plus
//                      shared vectors served from L2; nothing is spilled between the passes.
//                      With gamma blocks attached the same launch also performs the
//                      Normal-Gamma updates and log_post of the sweep (omc_gmrf_sweep).
#include <math.h>
#include <time.h>

#include "omc_common.h"

#ifndef OMC_EARLY_LAST_PAIR
#define OMC_EARLY_LAST_PAIR 1  // SIG 2: the last pair of draws made before the scales arrive too (one value in the tile row's pad slot, one in registers; both in registers: 68 spilled bytes, 16.0 against 14.6 us per sweep at 128 chains)
#endif
#ifndef OMC_GENERIC_PARK
#define OMC_GENERIC_PARK 1  // SIG 0, M <= 10: draws made under the loads of the tile fills and parked in LDS (0: all in the forward pass; A/B builds)
#endif
#ifndef OMC_EARLY_DEFER_STORE
#define OMC_EARLY_DEFER_STORE 0  // SIG 2: 1 = x stored behind the hand-over instead of inside the quadratic-form pass (measured: 13.51 against 13.38 us per sweep at 128 chains)
#endif
#ifndef OMC_EARLY_ONE_POLLER
#define OMC_EARLY_ONE_POLLER 1  // SIG 2: wave 0 polls the hand-over line in memory, the other waves poll its LDS copy
#endif
#ifndef OMC_PAIR_IN_SCAN_WINDOW
#define OMC_PAIR_IN_SCAN_WINDOW 0  // SIG 1: 1 = waves 1..15 make their last buffered pair of draws while wave 0 scans the Moebius wave totals (measured: 80.7 against 80.3 us per sweep)
#endif
#ifndef OMC_EARLY_PFQ_AHEAD
#define OMC_EARLY_PFQ_AHEAD 0  // SIG 2: 1 = the quadratic forms prefetch issued before the forward substitution instead of before the reverse scan (measured: 13.4 against 12.7 us per sweep at 128 chains)
#endif
#ifndef OMC_SHIFT_PREFETCH
#define OMC_SHIFT_PREFETCH 0  // SIG 3: 1 = the chain's centre slice prefetched with the shared centre under the reverse scan (144 spilled bytes)
#endif
#ifndef OMC_SHIFT_PARK_C
#define OMC_SHIFT_PARK_C 1  // SIG 3: the draws' LDS slots take the chain's centre slice (HBM) instead of the off-diagonal slice (L2)
#endif
#ifndef OMC_JOIN_OR_LIB
#define OMC_JOIN_OR_LIB 0  // 1: the join test through __syncthreads_or (three barriers; A/B builds)
#endif

struct TermsDev {
  int n_terms;
  const double* diag[OMC_MAX_TERMS];
  const double* off[OMC_MAX_TERMS];
  const double* rhs[OMC_MAX_TERMS];
  const double* center[OMC_MAX_TERMS];
  const double* scale[OMC_MAX_TERMS];
};
// per-chain part of the terms' centres [C][ld] (omc_tridiag_terms::center_chain).  Its own struct, at the END of the kernel
// arguments: the structure-specialised instantiation never reads it, and with the fields inside TermsDev the shifted
// argument offsets alone cost that instantiation 1 us per sweep (same-box A/B).
struct CentreChain {
  const double* v;  // the chains' vectors [C][ld], or NULL
  int64_t ld;
  int quad_skip;    // bit k: term k's fused quadratic form is not wanted (generic instantiation; option "tridiag_quad_skip")
  int k;            // the term it belongs to (ONE term per launch: with the code unrolled over all four terms the generic
                    // instantiation spilled 100 bytes per lane)
};

struct GammaDev {
  int enabled;
  double a0, b0, half_npos;
  double lnorm;  // a0*log(b0) - lgamma(a0), host-computed
  const double* g_inject;
  double* store;
  double* scale_out;  // writable alias of T.scale[k]
  const double* logdet_unscaled;
  omc_rng_key key;
};

// omc_gmrf_run: several sweeps of the same chains in ONE launch (blockIdx = sweep * C + chain).  What differs from
// sweep to sweep is small and wave-uniform; it sits in the kernel arguments, indexed by the sweep.
#define OMC_RUN_MAX 32
struct SweepRec {
  uint64_t draw;      // draw index of the sweep's standard-normal stream; the Gamma streams are draw + gdraw[k]
  double* x;          // where the draw goes: the sweep's store slab, or the scratch slab
  double* log_post;   // or NULL
  int64_t slot_off;   // offset (in doubles) of the sweep's slot in the per-chain scalar stores; < 0: not stored
};
// hand-over of a chain's freshly drawn scales from the workgroup of sweep s to the one of sweep s+1, which may sit on
// another XCD: data-tagged 8-byte granules {32 bits of the double, 32-bit tag}, written and read with agent-scope
// (sc1) accesses -- no flag, no fence, no ordering between granules needed (MI355X_MICROARCH.md, hand-off forms).
// One 128-byte line per chain: [term k][half] at word 2 k + half.
#define OMC_HANDOFF_WORDS 16

struct TriArgs {
  TermsDev T;
  int64_t n, C, chain_offset;
  const double* rhs_chain; int64_t ld_rhs;
  const double* z; int64_t ld_z;
  int zero_z;
  omc_rng_key key;
  double* x; int64_t ld_x;
  double* quad;
  double* logdet;
  long long* bad;
  double perturb_start;           // tests only: relative error put on every segment's Moebius start value
  int newton_max;                 // Newton corrections of the segment joins before the sequential fallback takes over
  unsigned long long* fallbacks;  // diagnostic counter: chains whose pivot joins went through the sequential fallback
  double* work;
  // fused sweep (omc_gmrf_sweep)
  unsigned long long* stamps;  // diagnostic: [chain][wave][16] s_memtime at phase boundaries, or NULL
  int fused;
  GammaDev gb[OMC_MAX_TERMS];
  // generic workgroup-per-chain instantiation: the same blocks and streams in device memory, where wave 0's lanes index them
  // by their term (indexing the ARGUMENT copy per lane makes the compiler keep a private image of all blocks: 360 bytes of
  // scratch per lane stored by every wave at entry, 160 instead of 107 us per sweep)
  const GammaDev* gb_dev;
  const unsigned long long* gdraw_dev;
  double* log_post;
  // several sweeps per launch (n_sweeps > 0; workgroup-per-chain form only)
  int n_sweeps;
  int reenter;                     // 1, 2: a workgroup restarts itself as its chain's next sweep
  int block_sweeps;                // ... for this many sweeps in a row; then a fresh workgroup (block index + C) takes the chain
                                   // over through the global hand-over line.  n_sweeps: one workgroup per chain for the launch
  int early_draws;                 // 1: all buffered pairs of draws are made before the scales are waited for (see the kernel)
  uint32_t epoch;                  // tag of sweep 0's inputs + 1 = tag its outputs carry; unique per context over launches
  uint64_t seed;
  uint64_t gdraw[OMC_MAX_TERMS];   // Gamma stream of term k = sweep's draw index + gdraw[k]
  unsigned long long* handoff;     // [C][OMC_HANDOFF_WORDS]
  unsigned long long* timeouts;    // counter: hand-overs that did not arrive (dispatch-order assumption broken)
  // diagnostic sweep clock: wave 0 of a (sweep, chain) workgroup leaves {s_memrealtime at entry, at exit} in record
  // (sweep_times_pos + sweep) mod sweep_times_cap of the ring [cap][C][2]; NULL = off
  unsigned long long* sweep_times;
  int64_t sweep_times_cap, sweep_times_pos;
  SweepRec rec[OMC_RUN_MAX];
  CentreChain cc;
};

__device__ __forceinline__ bool run_mode(const TriArgs& A) { return A.n_sweeps > 0; }
__device__ __forceinline__ omc_rng_key sweep_gamma_key(const TriArgs& A, int sw, const GammaDev& g, uint64_t gd) {
  return run_mode(A) ? omc_make_key(A.seed, A.rec[sw].draw + gd, OMC_RNG_GAMMA) : g.key;
}
__device__ __forceinline__ double* sweep_gamma_store(const TriArgs& A, int sw, const GammaDev& g) {
  if (!run_mode(A)) return g.store;
  const int64_t off = A.rec[sw].slot_off;
  return (g.store && off >= 0) ? g.store + off : nullptr;
}
__device__ __forceinline__ double* sweep_log_post(const TriArgs& A, int sw) { return run_mode(A) ? A.rec[sw].log_post : A.log_post; }

__device__ __forceinline__ double fast_rcp(double d) { return omc_rcp_nr(d); }

// Normal-Gamma updates + log_post of one chain, run by one lane (sampler.py:252-288, model.py:57-70)
__device__ __forceinline__ void sweep_epilogue(const TriArgs& A, int64_t c, const double* quad) {
  double lp = 0.0;
  bool failed = false;
  const double nd = (double)A.n;
  _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < A.T.n_terms) {
    const GammaDev& g = A.gb[k];
    double s = A.T.scale[k] ? A.T.scale[k][c] : 1.0;
    if (g.enabled) {
      const double a = g.a0 + g.half_npos;
      const double b = g.b0 + 0.5 * quad[k];
      const double sc = (b == 0.0) ? INFINITY : omc_rcp_nr(b);
      const double gd = g.g_inject ? g.g_inject[c] : omc_standard_gamma(g.key, A.chain_offset + c, a, &failed);
      s = gd * sc;
      g.scale_out[c] = s;
      if (g.store) g.store[c] = s;
    }
    if (A.log_post) {
      double lpk = 0.5 * (nd * log(s) + g.logdet_unscaled[0] - nd * 1.8378770664093453 - s * quad[k]);
      if (g.enabled) lpk += g.lnorm + (g.a0 - 1.0) * log(s) - g.b0 * s;
      lp += lpk;
    }
  }
  if (A.log_post) A.log_post[c] = lp;
  if (failed) atomicMin((unsigned long long*)A.bad, (unsigned long long)c);
}

__device__ __forceinline__ double read_lane_d(double v, int l) {  // l wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// log-posterior of one sweep from the per-term scales, quadratic forms and log-determinants: lanes 16 k (k = term)
// hold s, qk, ldet of their term; lane 0 stores the sum (model.py:57-70 -> gmrf.py:321-348, distribution.py:241-261).
// Split from the epilogue so that a self-restarting workgroup can leave it to a wave that has slack (see the kernel).
template <bool DEV = false>
__device__ __forceinline__ void sweep_log_post_wave(const TriArgs& A, int64_t c, int lane, double s, double qk, double ldet,
                                                    double* lp_out) {
  const int k = lane >> 4, j = lane & 15;
  const bool term_on = k < A.T.n_terms;
  GammaDev g;
  if constexpr (DEV) {
    g = A.gb_dev[k];
  } else {
    g = A.gb[0];
#pragma unroll
    for (int t = 1; t < OMC_MAX_TERMS; ++t) {
      if (k == t) g = A.gb[t];
    }
  }
  double lp = 0.0;
  if (term_on && j == 0) {
    const double nd = (double)A.n;
    // the lean fdlibm log kernel (< 1 ulp) for finite positive scales, the library's log for the rest (zero-rate guard:
    // scale = inf)
    const double ls = (s > 0.0 && s < INFINITY) ? omc_log_unit(s) : log(s);
    lp = 0.5 * (nd * ls + ldet - nd * 1.8378770664093453 - s * qk);
    if (g.enabled) lp += g.lnorm + (g.a0 - 1.0) * ls - g.b0 * s;
  }
  // terms are summed in order 0,1,2,3 as the serial epilogue does
  const double t0 = read_lane_d(lp, 0), t1 = read_lane_d(lp, 16), t2 = read_lane_d(lp, 32), t3 = read_lane_d(lp, 48);
  if (lane == 0) lp_out[c] = ((t0 + t1) + t2) + t3;
}

// The same epilogue spread over the 64 lanes of one wave (the workgroup-per-chain kernel runs it on
// wave 0 while the other waves are already storing x): lanes 16k..16k+15 belong to term k and each
// evaluates one Marsaglia-Tsang attempt; the lowest accepted attempt is the serial answer.
// The same epilogue spread over the 64 lanes of one wave (the workgroup-per-chain kernel runs it on
// wave 0): lanes 16k..16k+15 belong to term k.
//
// Part 1, `sweep_gamma_draws_wave`: the standard-gamma draws Gamma(a,1).  They depend only on the
// prior shape and the node count, not on the data, so the kernel makes them at its very start, in the
// shadow of the first global loads; each lane evaluates one Marsaglia-Tsang attempt, the lowest
// accepted attempt is the serial answer.  Part 2, `sweep_epilogue_wave`: scale by 1/b once the
// quadratic forms are known, store, log_post.
template <bool DEV = false>
__device__ __forceinline__ double sweep_gamma_draws_wave(const TriArgs& A, int64_t c, int lane, bool* failed, int sw = 0) {
  const int k = lane >> 4, j = lane & 15;
  const bool term_on = k < A.T.n_terms;
  GammaDev g;
  uint64_t gdr;
  if constexpr (DEV) {
    g = A.gb_dev[k];
    gdr = A.gdraw_dev[k];
  } else {
    g = A.gb[0];
    gdr = A.gdraw[0];
#pragma unroll
    for (int t = 1; t < OMC_MAX_TERMS; ++t)
      if (k == t) { g = A.gb[t]; gdr = A.gdraw[t]; }
  }
  g.key = sweep_gamma_key(A, sw, g, gdr);
  const bool draw = term_on && g.enabled;
  double gd = 0.0;
  if (__ballot(draw) == 0ull) return gd;
  if (draw && g.g_inject) {
    gd = g.g_inject[c];
  } else if (draw) {
    const omc_gamma_prep p = omc_gamma_prepare(g.key, A.chain_offset + c, g.a0 + g.half_npos);
    double v = 0.0;
    // Attempt 0 alone first: it is accepted with probability > 0.95 (-> 1 for large shapes), mostly by the
    // log-free squeeze test, and with one lane per term active the wave rarely has to walk the log branch
    // that some lane of a full 16-attempt evaluation nearly always needs.
    bool ok = (j == 0) && omc_gamma_attempt(g.key, A.chain_offset + c, p, 0u, v);
    const unsigned long long first = __ballot(ok), want = __ballot(j == 0);
    if (first != want) {  // wave-uniform: some term's first attempt was rejected -> evaluate the other 15 as well
      if (j != 0) ok = omc_gamma_attempt(g.key, A.chain_offset + c, p, (uint32_t)j, v);
    }
    const unsigned long long m = (__ballot(ok) >> (16 * k)) & 0xffffull;  // accepted attempts of this group
    if (m == 0ull) {  // astronomically rare: continue serially on the group's first lane
      if (j == 0) {
        ok = false;
        for (uint32_t at = 16; at < 256 && !ok; ++at) ok = omc_gamma_attempt(g.key, A.chain_offset + c, p, at, v);
        *failed = !ok;
        gd = ok ? v : p.boost * p.d;
      }
    } else {
      gd = __shfl(v, __ffsll((long long)m) - 1 + 16 * k, 64);
    }
  }
  return gd;
}

template <bool DEV = false>
__device__ __forceinline__ void sweep_epilogue_wave(const TriArgs& A, int64_t c, double q0, double q1, double q2, double q3,
                                                    double s_old, double ldet, double gd, bool failed, int lane, int sw = 0,
                                                    unsigned long long* lds_hand = nullptr, bool defer_lp = false,
                                                    double* lds_q = nullptr) {
  const int k = lane >> 4, j = lane & 15;
  const bool term_on = k < A.T.n_terms;
  // per-lane copy of this lane's term, selected with compile-time indices (a dynamically indexed
  // kernel-argument array would be spilled to scratch)
  GammaDev g;
  double s = s_old;  // this lane's term; scalars were loaded before the quad phase
  if constexpr (DEV) {
    g = A.gb_dev[k];
  } else {
    g = A.gb[0];
#pragma unroll
    for (int t = 1; t < OMC_MAX_TERMS; ++t) {
      if (k == t) g = A.gb[t];
    }
  }
  const double qk = (k == 0) ? q0 : ((k == 1) ? q1 : ((k == 2) ? q2 : q3));
  if (!term_on) s = 1.0;
  double* const lp_out = sweep_log_post(A, sw);
  // deferred log-posterior: the quadratic forms go to LDS ahead of the scale granules (one wave's LDS writes land in
  // order: whoever has seen the granules finds these)
  if (lp_out && defer_lp && term_on && j == 0) __hip_atomic_store(lds_q + k, qk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (term_on && g.enabled) {
    const double b = g.b0 + 0.5 * qk;
    s = gd * ((b == 0.0) ? INFINITY : omc_rcp_nr(b));  // sampler.py:285-287
    if (j == 0) {
      if (run_mode(A)) {  // hand the new scale to the workgroup of the chain's next sweep (same launch): FIRST -- a consumer
                          // on another CU waits for exactly these stores, and vector-memory operations leave in order
        const uint32_t tag = A.epoch + (uint32_t)sw + 1u;
        unsigned long long* h = A.handoff + c * OMC_HANDOFF_WORDS + 2 * k;
        const unsigned long long lo = ((unsigned long long)tag << 32) | (uint32_t)__double2loint(s);
        const unsigned long long hi = ((unsigned long long)tag << 32) | (uint32_t)__double2hiint(s);
        if (lds_hand) {  // self-restarting workgroup: the consumer is this workgroup -- the same granules through LDS
          __hip_atomic_store(lds_hand + 2 * k, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(lds_hand + 2 * k + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __hip_atomic_store(h, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(h + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // the caller's scale array: written by the launch's last sweep only (two XCDs' write-through stores to one
      // address within a launch have no defined order)
      if (!run_mode(A) || sw == A.n_sweeps - 1) g.scale_out[c] = s;
      double* const st = sweep_gamma_store(A, sw, g);
      if (st) st[c] = s;
    }
  }
  if (lp_out && !defer_lp) sweep_log_post_wave<DEV>(A, c, lane, s, qk, ldet, lp_out);
  if (failed) atomicMin((unsigned long long*)A.bad, (unsigned long long)c);
}

// ------------------------------------------------------------------------------------------
// serial kernel
__global__ void __launch_bounds__(64) k_tridiag_serial(TriArgs A) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= A.C) return;
  const int nt = A.T.n_terms;
  const int64_t n = A.n;
  double sc[OMC_MAX_TERMS];
  _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < nt) sc[k] = A.T.scale[k] ? A.T.scale[k][c] : 1.0;
  double* lw = A.work + c * n;
  double* xo = A.x + c * A.ld_x;
  const int64_t gc = A.chain_offset + c;
  bool bad = false;
  double lp = 0.0, bprev = 0.0, u = 0.0, logdet = 0.0, zodd = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    double av = 0.0, bv = 0.0, rv = 0.0;
    _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < nt) {
      av = fma(sc[k], A.T.diag[k] ? A.T.diag[k][i] : 1.0, av);
      if (A.T.off[k] && i < n - 1) bv = fma(sc[k], A.T.off[k][i], bv);
      if (A.T.rhs[k]) rv = fma(sc[k], A.T.rhs[k][i], rv);
    }
    if (A.rhs_chain) rv += A.rhs_chain[c * A.ld_rhs + i];
    double D = fma(-lp, bprev, av);
    bad |= !(D > 0.0);
    double rD = fast_rcp(D);
    u = fma(-lp, u, rv);
    double zi;
    if (A.z) {
      zi = A.z[c * A.ld_z + i];
    } else if (A.zero_z) {
      zi = 0.0;
    } else if ((i & 1) == 0) {
      omc_normal_pair(omc_rng_block(A.key, gc, (uint32_t)(i >> 1)), zi, zodd);
    } else {
      zi = zodd;
    }
    xo[i] = fma(u, rD, zi * sqrt(rD));
    logdet -= log(rD);
    lp = bv * rD;
    lw[i] = lp;
    bprev = bv;
  }
  const bool want_quad = A.quad || A.fused;
  double acc[OMC_MAX_TERMS] = {0, 0, 0, 0}, rnext[OMC_MAX_TERMS] = {0, 0, 0, 0};
  double x = 0.0;
  for (int64_t i = n - 1; i >= 0; --i) {
    x = fma(-lw[i], x, xo[i]);
    xo[i] = x;
    if (want_quad) {
      _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < nt) {
        double r = x - (A.T.center[k] ? A.T.center[k][i] : 0.0);
        double dk = A.T.diag[k] ? A.T.diag[k][i] : 1.0;
        double ok = (A.T.off[k] && i < n - 1) ? A.T.off[k][i] : 0.0;
        acc[k] += dk * r * r + 2.0 * ok * r * rnext[k];
        rnext[k] = r;
      }
    }
  }
  if (A.quad)
    _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < nt) A.quad[k * A.C + c] = acc[k];
  if (A.logdet) A.logdet[c] = logdet;
  if (bad) atomicMin((unsigned long long*)A.bad, (unsigned long long)c);
  if (A.fused) sweep_epilogue(A, c, acc);
}

// ------------------------------------------------------------------------------------------
// segmented kernel: scan machinery
struct Mob { double a, b, c, d; };  // 2x2 matrix [[a,b],[c,d]] acting as D -> (aD+b)/(cD+d)
struct Aff { double p, q; };        // v -> p + q v

__device__ __forceinline__ Mob mob_norm(Mob m) {
  double mx = fmax(fmax(fabs(m.a), fabs(m.b)), fmax(fabs(m.c), fabs(m.d)));
  int e = (mx > 0.0 && mx < INFINITY) ? ilogb(mx) : 0;
  double s = ldexp(1.0, -e);  // exact power of two: the map is unchanged
  return Mob{m.a * s, m.b * s, m.c * s, m.d * s};
}
// later-after-earlier composition.  The Moebius product is left unscaled: a product of k matrices whose
// largest entries lie in [1, 2) has entries below 2^(2k-1), so the scans rescale (`renorm`, an exact power
// of two: the map is unchanged) once per 16-lane row pass and per fold, not once per product.
__device__ __forceinline__ Mob compose(const Mob& L, const Mob& E) {
  return Mob{fma(L.a, E.a, L.b * E.c), fma(L.a, E.b, L.b * E.d), fma(L.c, E.a, L.d * E.c), fma(L.c, E.b, L.d * E.d)};
}
__device__ __forceinline__ Aff compose(const Aff& L, const Aff& E) { return Aff{fma(L.q, E.p, L.p), L.q * E.q}; }
__device__ __forceinline__ Mob renorm(const Mob& m) { return mob_norm(m); }
__device__ __forceinline__ Aff renorm(const Aff& f) { return f; }

__device__ __forceinline__ Mob shfl(const Mob& v, int d, int w, bool rev) {
  return rev ? Mob{__shfl_down(v.a, d, w), __shfl_down(v.b, d, w), __shfl_down(v.c, d, w), __shfl_down(v.d, d, w)}
             : Mob{__shfl_up(v.a, d, w), __shfl_up(v.b, d, w), __shfl_up(v.c, d, w), __shfl_up(v.d, d, w)};
}
__device__ __forceinline__ Aff shfl(const Aff& v, int d, int w, bool rev) {
  return rev ? Aff{__shfl_down(v.p, d, w), __shfl_down(v.q, d, w)} : Aff{__shfl_up(v.p, d, w), __shfl_up(v.q, d, w)};
}

// Exclusive scan of `v` over the lanes of one chain, in segment order (or reverse order).
// Wd = lanes of this chain inside one wave (power of two); for MULTI the chain spans nw waves
// and `lds` (>= nw entries) carries the wave totals.  Every lane of the block must call it.
template <class T, bool MULTI>
__device__ __forceinline__ T excl_scan(T v, const T ident, int pos, int Wd, bool rev, T* lds, int wave, int nw) {
  const int p = rev ? (Wd - 1 - pos) : pos;  // rank in scan order inside the wave
  for (int d = 1; d < Wd; d <<= 1) {
    T o = shfl(v, d, Wd, rev);
    if (p >= d) v = compose(v, o);
    if (d & 0x2a) v = renorm(v);  // every other doubling step
  }
  v = renorm(v);
  T e = shfl(v, 1, Wd, rev);
  if (p == 0) e = ident;
  if (MULTI) {
    if (p == Wd - 1) lds[wave] = v;
    __syncthreads();
    T pre = ident;
    if (!rev) {
      for (int w = 0; w < wave; ++w) pre = renorm(compose(lds[w], pre));
    } else {
      for (int w = nw - 1; w > wave; --w) pre = renorm(compose(lds[w], pre));
    }
    e = compose(e, pre);
    __syncthreads();
  }
  return e;
}

// value held by the previous segment's lane (identity for the first segment)
template <bool MULTI>
__device__ __forceinline__ void prev_lane2(double& v0, double& v1, double id0, double id1, int pos, int Wd,
                                           double* lds, int wave) {
  double a = __shfl_up(v0, 1, Wd), b = __shfl_up(v1, 1, Wd);
  if (MULTI) {
    if (pos == Wd - 1) { lds[2 * wave] = v0; lds[2 * wave + 1] = v1; }
    __syncthreads();
    if (pos == 0 && wave > 0) { a = lds[2 * (wave - 1)]; b = lds[2 * (wave - 1) + 1]; }
    if (pos == 0 && wave == 0) { a = id0; b = id1; }
    __syncthreads();
  } else if (pos == 0) {
    a = id0; b = id1;
  }
  v0 = a; v1 = b;
}

template <bool MULTI>
__device__ __forceinline__ double group_sum(double v, int Wd, double* lds, int wave, int nw) {
  for (int d = Wd >> 1; d >= 1; d >>= 1) v += __shfl_xor(v, d, Wd);
  if (MULTI) {
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += lds[w];
    __syncthreads();
    v = t;
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// Full-wave (64 lanes = 64 consecutive segments of one chain) scans on DPP lane shifts: a shift is
// one v_mov_dpp per 32-bit word instead of a ds_bpermute round trip.  Lanes without a source
// receive the identity, so no lane needs a conditional.
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138

// `fill` is always a compile-time identity element (0.0 or 1.0) at the call sites: a word of it that is zero
// is produced by the instruction's own bound_ctrl zero fill instead of a preloaded destination register
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v, double fill) {
  const int flo = __double2loint(fill), fhi = __double2hiint(fill);
  int lo, hi;
  if (__builtin_constant_p(flo) && flo == 0) lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  else lo = __builtin_amdgcn_update_dpp(flo, __double2loint(v), CTRL, 0xf, 0xf, false);
  if (__builtin_constant_p(fhi) && fhi == 0) hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  else hi = __builtin_amdgcn_update_dpp(fhi, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ Mob dpp_mov(const Mob& v, const Mob& f) {
  return Mob{dpp_mov<CTRL>(v.a, f.a), dpp_mov<CTRL>(v.b, f.b), dpp_mov<CTRL>(v.c, f.c), dpp_mov<CTRL>(v.d, f.d)};
}
template <int CTRL> __device__ __forceinline__ Aff dpp_mov(const Aff& v, const Aff& f) {
  return Aff{dpp_mov<CTRL>(v.p, f.p), dpp_mov<CTRL>(v.q, f.q)};
}
// row_bcast15 / row_bcast31 (GFX9 DPP): the last lane of a row -> every lane of the next row / lane 31 -> rows 2
// and 3.  Rows not selected by ROW_MASK keep `fill` (the identity), so composing with the result is a no-op there.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_bcast(double v, double fill) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ Mob dpp_bcast(const Mob& v, const Mob& f) {
  return Mob{dpp_bcast<CTRL, ROW_MASK>(v.a, f.a), dpp_bcast<CTRL, ROW_MASK>(v.b, f.b), dpp_bcast<CTRL, ROW_MASK>(v.c, f.c),
             dpp_bcast<CTRL, ROW_MASK>(v.d, f.d)};
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ Aff dpp_bcast(const Aff& v, const Aff& f) {
  return Aff{dpp_bcast<CTRL, ROW_MASK>(v.p, f.p), dpp_bcast<CTRL, ROW_MASK>(v.q, f.q)};
}
__device__ __forceinline__ double read_lane(double v, int l) {  // l must be wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ Mob read_lane(const Mob& v, int l) {
  return Mob{read_lane(v.a, l), read_lane(v.b, l), read_lane(v.c, l), read_lane(v.d, l)};
}
__device__ __forceinline__ Aff read_lane(const Aff& v, int l) { return Aff{read_lane(v.p, l), read_lane(v.q, l)}; }

// inclusive scan inside each row of 16 lanes, forward (REV = false) or from the high lane down
template <class T, bool REV>
__device__ __forceinline__ T row_scan(T v, const T& id) {
  if (!REV) {
    v = compose(v, dpp_mov<DPP_ROW_SHR(1)>(v, id));
    v = compose(v, dpp_mov<DPP_ROW_SHR(2)>(v, id));
    v = compose(v, dpp_mov<DPP_ROW_SHR(4)>(v, id));
    v = compose(v, dpp_mov<DPP_ROW_SHR(8)>(v, id));
  } else {
    v = compose(v, dpp_mov<DPP_ROW_SHL(1)>(v, id));
    v = compose(v, dpp_mov<DPP_ROW_SHL(2)>(v, id));
    v = compose(v, dpp_mov<DPP_ROW_SHL(4)>(v, id));
    v = compose(v, dpp_mov<DPP_ROW_SHL(8)>(v, id));
  }
  return renorm(v);
}

// Workgroup barrier that orders LDS traffic only.  `__syncthreads()` is a fence as well: it waits for every
// outstanding vector-memory operation of the wave (vmcnt(0)) -- here that would be the 80 KB of x stores and the
// LDS-DMA transfers, which no other wave ever reads; the hand-overs of the scans and reductions go through LDS.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Exclusive scan over all lanes of the workgroup (one chain), in segment order or reversed.
// `lds` holds one entry per wave.  Every lane of the block must call it.
// ONE_WAVE: the scan over the wave totals is done by wave 0 alone and handed out through `lds2` behind a second
// barrier, instead of redundantly by every wave -- worth it for the Moebius elements, whose 16-lane row scan
// is ~140 vector-ALU instructions per wave (x 16 waves on 4 SIMDs) against a few hundred cycles of one wave.
// `idle_work` (ONE_WAVE only): run by every wave but wave 0 while that one scans the totals and they would wait for it
struct omc_no_idle_work { __device__ __forceinline__ void operator()() const {} };
template <class T, bool REV, bool ONE_WAVE = false, class F = omc_no_idle_work>
__device__ __forceinline__ T excl_scan_wg(T v, const T id, T* lds, int lane, int wave, int nw, T* lds2 = nullptr, F idle_work = F()) {
  v = row_scan<T, REV>(v, id);
  // row totals sit in the last (first) lane of each row; fold the preceding rows in
  const int row = lane >> 4;
  if (!REV) {
    // the classic wave64 pattern: lane 15 -> row 1 and lane 47 -> row 3, then lane 31 -> rows 2 and 3
    v = compose(v, dpp_bcast<0x142, 0xA>(v, id));
    v = compose(v, dpp_bcast<0x143, 0xC>(v, id));
  } else {  // no mirrored broadcast exists: fold through readlanes
    const T t3 = read_lane(v, 48), t2 = read_lane(v, 32), t1 = read_lane(v, 16);
    const T p1 = compose(t2, t3), p0 = compose(t1, p1);
    const T pre = row == 2 ? t3 : (row == 1 ? p1 : (row == 0 ? p0 : id));
    v = compose(v, pre);
  }
  T e = REV ? dpp_mov<DPP_WAVE_SHL1>(v, id) : dpp_mov<DPP_WAVE_SHR1>(v, id);
  if (nw > 1) {
    if (lane == (REV ? 0 : 63)) lds[wave] = v;  // wave total
    lds_barrier();
    const int w = __builtin_amdgcn_readfirstlane(wave);
    const int src = REV ? w + 1 : w - 1;
    if (ONE_WAVE) {
      if (w == 0) {
        // The wave totals arrive unscaled from two folds; products of Moebius matrices of a precision of magnitude
        // lambda shrink by ~1/lambda per factor, so sixteen of them in a row underflowed for lambda >= 1e6 on chains of
        // twelve and more waves (0/0 start values).  Rescaled here, the row pass sees factors in [1, 2) like the one
        // inside a wave.
        T t = (lane < nw) ? renorm(lds[lane]) : id;
        t = row_scan<T, REV>(t, id);
        if (lane < nw) lds2[lane] = t;
      } else {
        idle_work();
      }
      lds_barrier();
      if (src >= 0 && src < nw) e = compose(e, lds2[src]);
    } else {
      T t = (lane < nw) ? lds[lane] : id;        // nw <= 16: one row
      t = row_scan<T, REV>(t, id);
      if (src >= 0 && src < nw) e = compose(e, read_lane(t, src));
    }
    // no trailing barrier: consecutive calls must use different `lds` buffers (the barrier of
    // the next call then orders this call's reads before the buffer is written again)
  }
  return e;
}

// previous segment's (v0, v1); (id0, id1) for the first segment of the chain
__device__ __forceinline__ void prev_lane2_wg(double& v0, double& v1, double id0, double id1, double* lds, int lane,
                                              int wave, int nw) {
  double a = dpp_mov<DPP_WAVE_SHR1>(v0, id0), b = dpp_mov<DPP_WAVE_SHR1>(v1, id1);
  if (nw > 1) {
    if (lane == 63) { lds[2 * wave] = v0; lds[2 * wave + 1] = v1; }
    lds_barrier();
    if (lane == 0 && wave > 0) { a = lds[2 * (wave - 1)]; b = lds[2 * (wave - 1) + 1]; }
    // no trailing barrier: see excl_scan_wg
  }
  v0 = a; v1 = b;
}

// sums of the first nt accumulators over the workgroup with a single barrier; lds: [4][16]
__device__ __forceinline__ void sum4_wg(const double (&v)[OMC_MAX_TERMS], double (&out)[OMC_MAX_TERMS], int nt, double* lds,
                                        int lane, int wave, int nw) {
  double t[OMC_MAX_TERMS] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    if (k >= nt) continue;  // wave-uniform
    double x = v[k];
    x += dpp_mov<DPP_ROW_SHR(1)>(x, 0.0);
    x += dpp_mov<DPP_ROW_SHR(2)>(x, 0.0);
    x += dpp_mov<DPP_ROW_SHR(4)>(x, 0.0);
    x += dpp_mov<DPP_ROW_SHR(8)>(x, 0.0);
    t[k] = (read_lane(x, 15) + read_lane(x, 31)) + (read_lane(x, 47) + read_lane(x, 63));
  }
  if (nw > 1) {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < OMC_MAX_TERMS; ++k)
        if (k < nt) lds[k * 16 + wave] = t[k];
    }
    lds_barrier();
    // lane 16 k + w holds wave w's partial sum of term k; one row reduction serves all terms
    double x = ((lane & 15) < nw && (lane >> 4) < nt) ? lds[lane] : 0.0;
    x += dpp_mov<DPP_ROW_SHR(1)>(x, 0.0);
    x += dpp_mov<DPP_ROW_SHR(2)>(x, 0.0);
    x += dpp_mov<DPP_ROW_SHR(4)>(x, 0.0);
    x += dpp_mov<DPP_ROW_SHR(8)>(x, 0.0);
    t[0] = read_lane(x, 15); t[1] = read_lane(x, 31); t[2] = read_lane(x, 47); t[3] = read_lane(x, 63);
  }
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) out[k] = t[k];
}

__device__ __forceinline__ double sum_wg(double v, double* lds, int lane, int wave, int nw) {
  v += dpp_mov<DPP_ROW_SHR(1)>(v, 0.0);
  v += dpp_mov<DPP_ROW_SHR(2)>(v, 0.0);
  v += dpp_mov<DPP_ROW_SHR(4)>(v, 0.0);
  v += dpp_mov<DPP_ROW_SHR(8)>(v, 0.0);
  double t = (read_lane(v, 15) + read_lane(v, 31)) + (read_lane(v, 47) + read_lane(v, 63));
  if (nw > 1) {
    if (lane == 0) lds[wave] = t;
    lds_barrier();
    double u = 0.0;
    for (int w = 0; w < nw; ++w) u += lds[w];
    t = u;  // no trailing barrier: every call site owns its 16-entry slot of `lds`
  }
  return t;
}

// Workgroup-wide OR of a per-lane flag with ONE LDS barrier: every wave leaves its ballot in its own word of `slot`, all
// read the row behind the barrier.  (`__syncthreads_or` is a library reduction of three `s_barrier`s -- clear, `ds_or`,
// read -- and a full `__syncthreads` fence each; the join test sits on every chain-update's critical path.)  No trailing
// barrier: the next call on the same `slot` must lie behind another barrier.
__device__ __forceinline__ bool any_wg(int need, int* slot, int lane, int wave, int nw) {
  const bool mine = __ballot(need) != 0ull;
  if (nw <= 1) return mine;
  if (lane == 0) slot[wave] = mine ? 1 : 0;
  lds_barrier();
  const int f = (lane < nw) ? slot[lane] : 0;
  return __ballot(f) != 0ull;
}

// ------------------------------------------------------------------------------------------
// Wave-private LDS tile: converts between "lane owns M consecutive nodes" (registers) and
// "64 consecutive lanes touch 64 consecutive doubles" (global memory).  Tile element e
// (0 <= e < 64*M; e = lane'*M + j) lives at tile[e + e/M]: row stride M+1 doubles, so the
// per-lane reads at stride M+1 (odd) are bank-conflict free for ds_read_b64.
template <int M, bool MULTI>
struct Geom {
  int lane, wave, G;   // G: lanes per chain (sub-wave groups) when !MULTI
  int64_t chain0;      // MULTI: the chain; else first chain of this wave
  __device__ __forceinline__ int64_t node(int e) const {
    const int lp = e / M, j = e - lp * M;
    const int seg = MULTI ? (wave * 64 + lp) : (lp & (G - 1));
    return (int64_t)seg * M + j;
  }
  __device__ __forceinline__ int64_t chain(int e) const { return MULTI ? chain0 : chain0 + (e / M) / G; }
};

__device__ __forceinline__ void wave_lds_fence() {
  // DS operations of one wave execute in order; this only stops the compiler from moving them.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// In front of an LDS-DMA (global_load_lds) into a region this wave has been reading: the DMA's write reaches
// LDS through the vector-memory path, not the DS queue, so "DS operations execute in order" does not cover
// it -- a ds_read that has been issued but not yet served could see the new bytes.  Wait until every DS
// operation of the wave has returned (lgkmcnt(0); vmcnt and expcnt left alone).
__device__ __forceinline__ void lds_reads_done() {
  wave_lds_fence();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  wave_lds_fence();
}

// shared vector v[0..lim) -> tile (fill beyond lim); every sub-wave group then reads rows 0..G-1
template <int M, bool MULTI>
__device__ __forceinline__ void tile_fill_shared(double* tile, const Geom<M, MULTI>& g, const double* v, int64_t lim,
                                                 double fill) {
  wave_lds_fence();
#pragma unroll 2
  for (int t = 0; t < M; ++t) {
    const int e = t * 64 + g.lane;
    const int64_t nd = g.node(e);
    tile[e + e / M] = (nd < lim) ? v[nd] : fill;
  }
  wave_lds_fence();
}
// per-chain vector v[chain*ld + node]; one tile row per lane
template <int M, bool MULTI>
__device__ __forceinline__ void tile_fill_chain(double* tile, const Geom<M, MULTI>& g, const double* v, int64_t ld,
                                                int64_t lim, int64_t C, double fill) {
  wave_lds_fence();
#pragma unroll 2
  for (int t = 0; t < M; ++t) {
    const int e = t * 64 + g.lane;
    const int64_t nd = g.node(e), ch = g.chain(e);
    tile[e + e / M] = (nd < lim && ch < C) ? v[ch * ld + nd] : fill;
  }
  wave_lds_fence();
}
template <int M, bool MULTI>
__device__ __forceinline__ void tile_store_chain(const double* tile, const Geom<M, MULTI>& g, double* v, int64_t ld,
                                                 int64_t lim, int64_t C) {
  wave_lds_fence();
#pragma unroll 2
  for (int t = 0; t < M; ++t) {
    const int e = t * 64 + g.lane;
    const int64_t nd = g.node(e), ch = g.chain(e);
    if (nd < lim && ch < C) v[ch * ld + nd] = tile[e + e / M];
  }
  wave_lds_fence();
}

// Per-chain combination of the shared term vectors, formed while the tile is filled (coalesced):
//   DIAG: a = sum_k s_k diag_k (1 beyond n), OFF: b = sum_k s_k off_k, RHS: r = sum_k s_k rhs_k + rhs_chain
enum { COMB_DIAG = 0, COMB_OFF = 1, COMB_RHS = 2 };
// nodes per lane handled per batch of loads (memory-level parallelism vs registers)
#define OMC_CH(M) ((M) % 5 == 0 ? 5 : 4)
template <int M, bool MULTI, int WHICH>
__device__ __forceinline__ void tile_fill_comb(double* tile, const Geom<M, MULTI>& g, const TriArgs& A,
                                               const double (&sc)[OMC_MAX_TERMS]) {
  const int nt = A.T.n_terms;
  const int64_t n = A.n;
  wave_lds_fence();
#pragma unroll 2
  for (int t = 0; t < M; ++t) {
    const int e = t * 64 + g.lane;
    const int64_t nd = g.node(e), ch = g.chain(e);
    double v = (WHICH == COMB_DIAG) ? 1.0 : 0.0;
    const int64_t lim = (WHICH == COMB_OFF) ? n - 1 : n;
    if (nd < lim) {
      v = 0.0;
      _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < nt) {
        const double* src = (WHICH == COMB_DIAG) ? A.T.diag[k] : (WHICH == COMB_OFF ? A.T.off[k] : A.T.rhs[k]);
        if (!src && WHICH != COMB_DIAG) continue;
        const double sk = MULTI ? sc[k] : ((A.T.scale[k] && ch < A.C) ? A.T.scale[k][ch] : 1.0);
        v = fma(sk, src ? src[nd] : 1.0, v);
      }
      if (WHICH == COMB_RHS && A.rhs_chain && ch < A.C) v += A.rhs_chain[ch * A.ld_rhs + nd];
    }
    tile[e + e / M] = v;
  }
  wave_lds_fence();
}

// Workgroup-per-chain form of the tile traffic.  A wave's tile is 64 rows (segments) of M nodes at row
// stride M+1.  In the coalesced mapping step t moves tile elements e = 64 t + lane (node wbase + e: one
// aligned 512-byte request per wave and step); element e lives at tile[e + e/M].  With 64 t = M A_t + B_t
// (compile-time) and lane = M q0 + r0:  e/M = A_t + q0 + (r0 >= M - B_t), so the address is a lane-only
// base (lane + q0), an immediate (64 t + A_t) and a one-bit carry: a compare and a select per element instead
// of running index arithmetic on the vector ALU (the kernel is bound by VALU issue).
template <int M>
struct TileMap {
  static constexpr int LU = 64;                 // lanes in use
  static constexpr int NS = M;                  // steps per tile
  static constexpr int CH = (M % 5 == 0) ? 5 : 4;  // steps per batch of loads (memory-level parallelism vs registers)
  __device__ static __forceinline__ int lane_base(int lane) { return lane + lane / M; }
  __device__ static __forceinline__ int lane_col(int lane) { return lane % M; }
  __device__ static constexpr int upto(int t) { return 64 * t; }  // elements of steps [0, t)
  // tile element of step t; tl = tile + lane_base, r0 = lane_col
  template <class P>
  __device__ static __forceinline__ P* elem(P* tl, int r0, int t) {
    const int At = (64 * t) / M, Bt = (64 * t) % M;
    P* p = tl + (64 * t + At);
    return (Bt != 0 && r0 >= M - Bt) ? p + 1 : p;
  }
  // its successor in node order: the next column, or column 0 of the next row
  template <class P>
  __device__ static __forceinline__ P* succ(P* p, int r0, int t) {
    const int Bt = (64 * t) % M;
    return (r0 == M - 1 - Bt) ? p + 2 : p + 1;
  }
};

// number of this wave's tile elements that lie below `lim` (wave-uniform)
template <int M>
__device__ __forceinline__ int wave_valid(int wave_u, int lim) {
  const int v = lim - wave_u * 64 * M;
  return v < 0 ? 0 : (v > 64 * M ? 64 * M : v);
}

// Right-hand-side part of the terms with a per-chain centre (omc_tridiag_terms::center_chain): v_i += s_k (M_k c_k)_i with
// c_k the chain's vector, in the coalesced mapping -- three predicated loads of c per node (the shifted ones come out of
// the cache lines the first one brought), nothing staged.  Wave-uniform skip when no term has one.
template <int M, int CH>
__device__ __forceinline__ void rhs_center_chain(double (&v)[CH], const TriArgs& A, const double (&sc)[OMC_MAX_TERMS], bool chain_ok,
                                                 int64_t cc, int wbase, int lane, int t0, int cnt, int nvalid) {
  if (!A.cc.v || !chain_ok) return;  // wave-uniform
  const int n = (int)A.n, kc = A.cc.k;
  const double* c = A.cc.v + cc * A.cc.ld + wbase;
  // the term's vectors and scale by wave-uniform selects (a dynamic index into the kernel arguments would cost a private copy)
  const double* dk = kc == 0 ? A.T.diag[0] : (kc == 1 ? A.T.diag[1] : (kc == 2 ? A.T.diag[2] : A.T.diag[3]));
  const double* ok = kc == 0 ? A.T.off[0] : (kc == 1 ? A.T.off[1] : (kc == 2 ? A.T.off[2] : A.T.off[3]));
  const double sk = kc == 0 ? sc[0] : (kc == 1 ? sc[1] : (kc == 2 ? sc[2] : sc[3]));
  // (plain predicated loads, element by element: batching them -- all loads of the batch first, at clamped positions -- was
  // no faster and cost the generic instantiation 100 bytes of scratch per lane)
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    if (t >= cnt) continue;
    const int idx = lane + (t0 + t) * 64, i = wbase + idx;
    if (idx >= nvalid) continue;
    double r = (dk ? (dk + wbase)[(unsigned)idx] : 1.0) * c[(unsigned)idx];
    if (ok) {
      if (i > 0) r = fma((ok + wbase)[idx - 1], c[idx - 1], r);
      if (i + 1 < n) r = fma((ok + wbase)[(unsigned)idx], c[(unsigned)idx + 1u], r);
    }
    v[t] = fma(sk, r, v[t]);
  }
}

// Per-chain combination of the shared term vectors, formed while the tile is filled.  All loads of a
// batch are issued back to back (L2 latency is paid once per batch) and only then combined.  A batch
// that lies wholly inside the vector takes the test-free path; the chain's last wave takes the
// predicated one for its boundary batch and only writes fill values beyond it.
struct omc_no_work { __device__ __forceinline__ void operator()(int) const {} };
// `under_loads(b)`: work that depends on nothing, run once per batch b while that batch's first loads are in flight
template <int M, int WHICH, bool CCH = false, class F = omc_no_work>
__device__ __forceinline__ void tile_fill_comb_wg(double* tile, int lane, int wave, int lbase, const TriArgs& A,
                                                  const double (&sc)[OMC_MAX_TERMS], bool chain_ok, int64_t cc,
                                                  F under_loads = F()) {
  constexpr bool OVL = !__is_same(F, omc_no_work);
  using TM = TileMap<M>;
  constexpr int CH = TM::CH;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nt = A.T.n_terms;
  const int n = (int)A.n;
  const int wbase = wave_u * 64 * M;
  const int nvalid = wave_valid<M>(wave_u, (WHICH == COMB_OFF) ? n - 1 : n);
  const double* rc = (WHICH == COMB_RHS && A.rhs_chain && chain_ok) ? A.rhs_chain + cc * A.ld_rhs + wbase : nullptr;
  const double fillv = (WHICH == COMB_DIAG) ? 1.0 : 0.0;
  double* tl = tile + lbase;
  const int r0 = TM::lane_col(lane);
  wave_lds_fence();
  {
#pragma unroll
    for (int t0 = 0; t0 < TM::NS; t0 += CH) {
      const int cnt = (TM::NS - t0 < CH) ? TM::NS - t0 : CH;
      double v[CH];
#pragma unroll
      for (int t = 0; t < CH; ++t) v[t] = 0.0;
      if (TM::upto(t0 + cnt) <= nvalid) {  // wave-uniform: the whole batch is inside
        // With work to overlap (OVL): the loads of the first two terms, then the work that depends on nothing, then their
        // combination; further terms one by one.  Without: every term loads and combines in turn (fewest registers).
        constexpr int KF = OVL ? 2 : 0;
        double ldf[KF > 0 ? KF : 1][CH];
        if constexpr (OVL) {
#pragma unroll
          for (int k = 0; k < KF; ++k) {
            if (k >= nt) continue;
            const double* src = (WHICH == COMB_DIAG) ? A.T.diag[k] : (WHICH == COMB_OFF ? A.T.off[k] : A.T.rhs[k]);
            if (!src) continue;
            const double* ps = src + wbase;
#pragma unroll
            for (int t = 0; t < CH; ++t)
              if (t < cnt) ldf[k][t] = ps[(unsigned)(lane + (t0 + t) * TM::LU)];
          }
          __builtin_amdgcn_sched_barrier(0);
          under_loads(t0 / CH);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < OMC_MAX_TERMS; ++k) {
          if (k >= nt) continue;
          const double* src = (WHICH == COMB_DIAG) ? A.T.diag[k] : (WHICH == COMB_OFF ? A.T.off[k] : A.T.rhs[k]);
          if (!src) {
            if (WHICH == COMB_DIAG) {
#pragma unroll
              for (int t = 0; t < CH; ++t) v[t] += sc[k];
            }
            continue;
          }
          if (k < KF) {
#pragma unroll
            for (int t = 0; t < CH; ++t)
              if (t < cnt) v[t] = fma(sc[k], ldf[k < KF ? k : 0][t], v[t]);
          } else {
            const double* ps = src + wbase;
            double ld[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t)
              if (t < cnt) ld[t] = ps[(unsigned)(lane + (t0 + t) * TM::LU)];
#pragma unroll
            for (int t = 0; t < CH; ++t)
              if (t < cnt) v[t] = fma(sc[k], ld[t], v[t]);
          }
        }
        if (WHICH == COMB_RHS && rc) {
          double ld[CH];
#pragma unroll
          for (int t = 0; t < CH; ++t)
            if (t < cnt) ld[t] = rc[(unsigned)(lane + (t0 + t) * TM::LU)];
#pragma unroll
          for (int t = 0; t < CH; ++t)
            if (t < cnt) v[t] += ld[t];
        }
        if (WHICH == COMB_RHS && CCH) rhs_center_chain<M, CH>(v, A, sc, chain_ok, cc, wbase, lane, t0, cnt, nvalid);
#pragma unroll
        for (int t = 0; t < CH; ++t)
          if (t < cnt) *TM::elem(tl, r0, t0 + t) = v[t];
      } else {
        if constexpr (OVL) under_loads(t0 / CH);
#pragma unroll
        for (int k = 0; k < OMC_MAX_TERMS; ++k) {
          if (k >= nt) continue;
          const double* src = (WHICH == COMB_DIAG) ? A.T.diag[k] : (WHICH == COMB_OFF ? A.T.off[k] : A.T.rhs[k]);
          if (!src) {
            if (WHICH == COMB_DIAG) {
#pragma unroll
              for (int t = 0; t < CH; ++t) v[t] += sc[k];
            }
            continue;
          }
          const double* ps = src + wbase;
#pragma unroll
          for (int t = 0; t < CH; ++t) {
            const int idx = lane + (t0 + t) * TM::LU;
            if (t < cnt && idx < nvalid) v[t] = fma(sc[k], ps[(unsigned)idx], v[t]);
          }
        }
#pragma unroll
        for (int t = 0; t < CH; ++t) {
          const int idx = lane + (t0 + t) * TM::LU;
          if (t >= cnt) continue;
          if (WHICH == COMB_RHS && rc && idx < nvalid) v[t] += rc[(unsigned)idx];
        }
        if (WHICH == COMB_RHS && CCH) rhs_center_chain<M, CH>(v, A, sc, chain_ok, cc, wbase, lane, t0, cnt, nvalid);
#pragma unroll
        for (int t = 0; t < CH; ++t) {
          const int idx = lane + (t0 + t) * TM::LU;
          if (t >= cnt) continue;
          if (idx < 64 * M) *TM::elem(tl, r0, t0 + t) = (idx < nvalid) ? v[t] : fillv;
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the next batch's loads from being hoisted over this one
    }
  }
  wave_lds_fence();
}

// sqrt(1/D) that scales a standard-normal draw: one Newton step on v_rsq_f64 (4e-15 relative, measured in
// benchmarks/micro/rcp_acc.hip) -- the draw's own scale, nothing downstream amplifies it
__device__ __forceinline__ double fast_sqrt(double r) {
  const double g = __builtin_amdgcn_rsq(r);
  const double s = r * g;
  return fma(fma(-s, s, r), 0.5 * g, s);
}

// diagnostic phase stamps (guide section 7, in-kernel stamps): lane 0 of every wave, only when enabled
#if OMC_NO_STAMPS
#define OMC_STAMP(k) do { } while (0)
#else
#define OMC_STAMP(k)                                                                                  \
  do {                                                                                                \
    if (A.stamps) {                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                              \
      const unsigned long long _t = __builtin_amdgcn_s_memtime();                                     \
      if (lane == 0 && chain_ok) A.stamps[((c * 16) + wave) * 16 + (k)] = _t;                          \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    }                                                                                                 \
  } while (0)
#endif

// join residual (relative) below which the pivots are accepted: ~72 ulp; the Moebius start already
// meets it for well-conditioned chains, weakly coupled ones take one or two Newton corrections
#define OMC_NEWTON_TOL 1.6e-14
#define OMC_NEWTON_MAX 4
#ifndef OMC_PARK_OFF
#define OMC_PARK_OFF 1
#endif
#ifndef OMC_PARK_DIAG
#define OMC_PARK_DIAG 1
#endif
// timing what-ifs (benchmarks/ab_headline.py builds variants with these; results are wrong by construction)
#ifndef OMC_WHATIF_NOSTORE
#define OMC_WHATIF_NOSTORE 0
#endif
// the draw is written once and never read back by this kernel: streaming (nt) stores measured 0.6 % faster
#ifndef OMC_STORE_NT
#define OMC_STORE_NT 1
#endif
#ifndef OMC_PREFETCH_QUAD
#define OMC_PREFETCH_QUAD 1
#endif
#ifndef OMC_WHATIF_NOQLOAD
#define OMC_WHATIF_NOQLOAD 0
#endif

// Quadratic forms (x - m_k)' M_k (x - m_k) of one wave's 64*M nodes in the coalesced mapping: x comes
// back from the tile (x_{i+1} = the next tile element; the slot behind the tile's last row holds the
// first x of the next wave), the shared vectors straight from L2.
template <int M>
__device__ __forceinline__ void quad_wg(const double* tile, int lane, int wave_u, int lbase, const TriArgs& A,
                                        double (&acc)[OMC_MAX_TERMS], int64_t cc) {
  using TM = TileMap<M>;
  constexpr int CH = TM::CH;
  const int nt = A.T.n_terms, n32 = (int)A.n;
  const int wbase = wave_u * 64 * M;
  const int nrem = n32 - wbase;  // nodes of the chain from this wave's first one on (may exceed the tile)
  const int nvalid = nrem < 64 * M ? (nrem < 0 ? 0 : nrem) : 64 * M;
  const double* tl = tile + lbase;
  const int r0 = TM::lane_col(lane);
  {
#pragma unroll
    for (int t0 = 0; t0 < TM::NS; t0 += CH) {
      const int cnt = (TM::NS - t0 < CH) ? TM::NS - t0 : CH;
      double xv[CH], xn[CH];  // x_i and x_{i+1}
      if (TM::upto(t0 + cnt) < nrem) {  // wave-uniform: i + 1 < n for every node of the batch
#pragma unroll
        for (int t = 0; t < CH; ++t)
          if (t < cnt) {
            xv[t] = *TM::elem(tl, r0, t0 + t);
            xn[t] = *TM::succ(TM::elem(tl, r0, t0 + t), r0, t0 + t);
          }
#pragma unroll
        for (int k = 0; k < OMC_MAX_TERMS; ++k) {
          if (k >= nt || ((A.cc.quad_skip >> k) & 1)) continue;  // (wave-uniform)
          const double *ck = A.T.center[k], *dk = A.T.diag[k], *ok = A.T.off[k];
          const double* cck = (A.cc.v && A.cc.k == k) ? A.cc.v + cc * A.cc.ld : nullptr;
          double ri[CH], rn[CH], dv[CH], ov[CH];
#pragma unroll
          for (int t = 0; t < CH; ++t) {
            if (t >= cnt) continue;
            const unsigned off = (unsigned)(lane + (t0 + t) * TM::LU);
            ri[t] = ck ? (ck + wbase)[off] : 0.0;
            rn[t] = (ck && ok) ? (ck + wbase)[off + 1u] : 0.0;
            if (cck) {  // per-chain part of the centre
              ri[t] += (cck + wbase)[off];
              if (ok) rn[t] += (cck + wbase)[off + 1u];
            }
            dv[t] = dk ? (dk + wbase)[off] : 1.0;
            ov[t] = ok ? (ok + wbase)[off] : 0.0;
          }
          if (ok) {
#pragma unroll
            for (int t = 0; t < CH; ++t) {
              if (t >= cnt) continue;
              const double a = xv[t] - ri[t], bnx = xn[t] - rn[t];
              acc[k] = fma(fma(2.0 * ov[t], bnx, dv[t] * a), a, acc[k]);
            }
          } else {
#pragma unroll
            for (int t = 0; t < CH; ++t) {
              if (t >= cnt) continue;
              const double a = xv[t] - ri[t];
              acc[k] = fma(dv[t] * a, a, acc[k]);
            }
          }
        }
      } else if (t0 * TM::LU < nvalid) {  // the chain's boundary batch
#pragma unroll
        for (int t = 0; t < CH; ++t) {
          if (t >= cnt) continue;
          const int idx = lane + (t0 + t) * TM::LU;
          const bool in = idx < nvalid, in1 = in && idx + 1 < nrem;
          const double x0 = in ? *TM::elem(tl, r0, t0 + t) : 0.0, x1 = in1 ? *TM::succ(TM::elem(tl, r0, t0 + t), r0, t0 + t) : 0.0;
#pragma unroll
          for (int k = 0; k < OMC_MAX_TERMS; ++k) {
            if (k >= nt || ((A.cc.quad_skip >> k) & 1)) continue;
            const double *ck = A.T.center[k], *dk = A.T.diag[k], *ok = A.T.off[k];
            const double* cck = (A.cc.v && A.cc.k == k) ? A.cc.v + cc * A.cc.ld : nullptr;
            const double a = x0 - ((ck && in) ? (ck + wbase)[(unsigned)idx] : 0.0) - ((cck && in) ? (cck + wbase)[(unsigned)idx] : 0.0);
            const double bnx = x1 - ((ck && ok && in1) ? (ck + wbase)[(unsigned)idx + 1u] : 0.0)
                                  - ((cck && ok && in1) ? (cck + wbase)[(unsigned)idx + 1u] : 0.0);
            const double d = in ? (dk ? (dk + wbase)[(unsigned)idx] : 1.0) : 0.0;
            const double o = (ok && in1) ? (ok + wbase)[(unsigned)idx] : 0.0;
            acc[k] = fma(fma(2.0 * o, bnx, d * a), a, acc[k]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Structure-specialised form of the workgroup-per-chain kernel (template parameter SIG).
//   SIG 0: any term structure (every pointer tested at run time).
//   SIG 1: the GMRF smoother of examples/4 and BASELINE configs[2]: two terms in either order,
//            term I = scaled identity precision (diag, off absent) with rhs and center,
//            term P = tridiagonal precision (diag, off present) without rhs and center,
//          so  a = sI + sP diagP,  b = sP offP,  r = sI rhsI (+ rhs_chain),
//              qI = |x - centerI|^2,  qP = x' M_P x.
// Knowing the structure at compile time removes the pointer tests and makes the load phases explicit, so
// that work which depends on nothing can be placed in their shadow: with one workgroup per CU all waves are
// in the same phase, and a phase that only waits for L2 (80 KB per vector and workgroup at ~30 B/clk per
// CU, 1.3 us) is otherwise dead time for the vector ALU.  The standard-normal draws are such work (pure
// functions of the Philox counter, 1.45 us per pair and workgroup): all but the last pair of a segment are
// generated while the precision and right-hand-side vectors are in flight and parked in LDS (`lds_z`,
// lane-private slots).  benchmarks/micro/overlap.hip measures the effect in isolation.  (Holding
// prefetched vectors of a later phase in registers instead was tried: at 128 VGPRs it spills, and the spill
// traffic costs more than the overlap gains.)
template <int M, bool FULLW>
__device__ __forceinline__ void coal_load(double (&v)[M], const double* base, int lane, int nvalid) {
#pragma unroll
  for (int t = 0; t < M; ++t) {
    const int idx = lane + 64 * t;
    v[t] = (FULLW || idx < nvalid) ? base[(unsigned)idx] : 0.0;
  }
}
template <int M>
__device__ __forceinline__ void coal_load(double (&v)[M], const double* base, int lane, int nvalid) {
  if (nvalid == 64 * M) coal_load<M, true>(v, base, lane, nvalid);
  else coal_load<M, false>(v, base, lane, nvalid);
}
// One pair of draws (Philox block `block` of the chain, Box-Muller) with the M loads of a coalesced vector
// issued one per Philox round: the loads drain while the wave computes (issued back to back in front of
// the arithmetic they also overlap, but less: a wave blocks at issue once the CU's vector-memory queue is
// full).
template <int M, bool FULLW>
__device__ __forceinline__ void draws_over_load(const omc_rng_key& key, int64_t gc, uint32_t block, double& z0, double& z1,
                                                double (&v)[M], const double* base, int lane, int nvalid) {
  constexpr int LPR = (M + 9) / 10;  // loads per round
  uint32_t c0 = block, c1 = key.c1, c2 = (uint32_t)gc;
  uint32_t c3 = key.c3_base | ((uint32_t)((uint64_t)gc >> 32) & 0xffu) << 16;
  uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#pragma unroll
    for (int q = 0; q < LPR; ++q) {
      const int t = r * LPR + q;
      if (t < M) {
        const int idx = lane + 64 * t;
        v[t] = (FULLW || idx < nvalid) ? base[(unsigned)idx] : 0.0;
      }
    }
    omc_philox_round_r(r, c0, c1, c2, c3, k0, k1);
    __builtin_amdgcn_sched_barrier(0);
  }
  omc_normal_pair(make_uint4(c0, c1, c2, c3), z0, z1);
}

// is_smoother: host-side test of the SIG 1 structure
static bool is_smoother(const TermsDev& T) {
  if (T.n_terms != 2) return false;
  for (int i = 0; i < 2; ++i) {  // i: the identity term
    const int p = 1 - i;
    if (!T.diag[i] && !T.off[i] && T.rhs[i] && T.center[i] && T.diag[p] && T.off[p] && !T.rhs[p] && !T.center[p]) return true;
  }
  return false;
}

// SIG 3: the smoother whose tridiagonal term carries the per-chain centre (and nothing else differs: no per-chain offsets,
// identity term with or without a shared centre)
static bool is_shifted_smoother(const TermsDev& T, const CentreChain& cc) {
  if (T.n_terms != 2 || !cc.v || cc.k < 0) return false;
  const int p = cc.k, i = 1 - cc.k;
  return !T.diag[i] && !T.off[i] && ((T.rhs[i] != nullptr) == (T.center[i] != nullptr)) && T.diag[p] && T.off[p] && !T.rhs[p] &&
         !T.center[p];
}

// Register plan per lane (M nodes): Y = b -> l ; W = 1/D -> g -> x (draws are consumed as they are made).
// The combined diagonal a (then the right-hand side r) lives in the wave's LDS tile.
template <int M, bool MULTI, int MAXT, int SIG = 0>
__global__ void __launch_bounds__(MAXT) k_tridiag_seg(TriArgs A, int G) {
  static_assert(SIG == 0 || MULTI, "specialised structures exist for the workgroup-per-chain form only");
  // SIG 1, 2: the two-term smoother.  SIG 2 is its WAITING form -- the (sweep, chain) grid with at most half as many chains
  // as CUs, in-kernel draws: the workgroup of a chain's next sweep sits on an idle CU until the previous sweep's scales
  // arrive, so everything that does not depend on them is done up front (all three vector loads, the buffered draws, the
  // Normal-Gamma standard draws), and the poll of the hand-over line is tight.  Never self-restarting.
  // SIG 3 (round 3): the smoother whose tridiagonal term is centred at a PER-CHAIN vector c (a hierarchical model's sampled
  // prior mean, or the sampled field a mean block is conditioned on: omc_tridiag_terms.center_chain on that term).
  // Evaluated by a shift: x = c + e, where e is the plain smoother's draw for the identity term centred at ys - c --
  //   Q e = sP P c + sI ys - Q c = sI (ys - c)   --
  // so the stencil product P c is never formed and all the specialised kernel has to do differently is element-wise: the
  // right-hand side sI (ys - c), the identity term's quadratic form around ys - c (= (x - ys)'(x - ys) of the shifted x), and
  // x = c + e on the way out; the tridiagonal term's quadratic form e'Pe IS (x - c)'P(x - c), with the parking scheme intact.
  // Same conditional law and the same draw for the same z up to rounding (the two right-hand sides are equal in exact
  // arithmetic).  ys may be absent (zeros).
  constexpr bool SMO = SIG != 0;
  constexpr bool EARLY = SIG == 2;
  constexpr bool SHIFT = SIG == 3;
  using TM = TileMap<M>;
  constexpr int NWMAX = MAXT / 64;
  __shared__ double lds_tile[NWMAX][64 * (M + 1) + 2];  // + the successor slot of the last row (quad_wg)
  __shared__ Mob lds_mob[16], lds_mob2[16];
  __shared__ double lds_g[64];     // wave 0's Normal-Gamma standard draws, start of kernel -> epilogue
  __shared__ unsigned long long lds_hand[2 * OMC_MAX_TERMS];  // self-restarting workgroups: the scales from sweep to sweep
  __shared__ double lds_q[OMC_MAX_TERMS];  // ... and the quadratic forms of the sweep whose log-posterior is finished by the next
  __shared__ Aff lds_aff[4][16];   // scans alternate buffers instead of paying a trailing barrier
  __shared__ double lds_x[2][32];  // neighbour exchange of the Newton passes (alternating)
  __shared__ double lds_d[6][16];  // reductions: one slot per call site
  __shared__ int lds_any[16];      // any_wg of the join test
  // SIG 1: pairs of draws per lane made ahead of the forward pass (all but the last; at most 8: LDS)
  // SIG 0, M <= 10 (round 3): the generic instantiation parks the same pairs -- its LDS image leaves 66 KB free -- and makes
  // them under the loads of its three tile fills (`fill_draws`), where the vector ALU used to idle; it generated all of a
  // segment's draws inside the forward pass (10 000 cycles of pure vector-ALU time on the critical path).
  constexpr bool PARKZ = SMO || (MULTI && M <= 10 && OMC_GENERIC_PARK);
  constexpr int NZB = PARKZ ? (M / 2 - 1 > 8 ? 8 : M / 2 - 1) : 0;
  __shared__ double lds_z[PARKZ ? NWMAX : 1][NZB > 0 ? 2 * NZB : 1][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int Wd = MULTI ? 64 : G;
  int64_t c;
  int s;
  Geom<M, MULTI> geo;
  geo.lane = lane; geo.wave = wave; geo.G = G;
  int sw = 0;  // sweep of this workgroup inside the launch (omc_gmrf_run: blockIdx = sweep * C + chain)
  bool restarted = false;  // this workgroup came here by its own restart (its predecessor sweep ran in this very workgroup)
  int left = 0;            // sweeps this workgroup will still restart itself for
  unsigned c_blk = 0;
  if (MULTI) {
    unsigned blk = blockIdx.x;
    // (Several sweeps per launch: the sweep index picks the sweep's record out of the kernel arguments.  Round 2 saw
    // "private copies" of the argument struct appear in the generic instantiation whenever such an index was added and blamed
    // the dynamic index; the cause was LLVM's limit of 300 users in the transform that forwards reads of a by-value kernel
    // argument to the kernel-argument segment -- see the note in the Makefile.  With the limit raised every instantiation
    // takes the sweep index, and none uses scratch.)
    if (A.n_sweeps > 1) {
      // A fresh workgroup: block index = (block of sweeps) * C + chain, and it starts at the block's first sweep.  A restarted
      // one carries what the restart put into the workgroup-id register: bit 31, the sweeps still to follow in its block
      // (bits 30:26) and the virtual index sweep * C + chain (OMC_REENTER at the end of the kernel).
      restarted = (blk >> 31) != 0u;
      const unsigned vblk = restarted ? (blk & 0x03ffffffu) : blk;
      const unsigned q = vblk / (unsigned)A.C;
      c_blk = vblk - q * (unsigned)A.C;
      if (restarted) {
        sw = (int)q;
        left = (int)((blk >> 26) & 31u);
      } else {
        const int g = (A.reenter && A.block_sweeps > 0) ? A.block_sweeps : 1;
        sw = (int)q * g;
        left = (A.n_sweeps - sw < g ? A.n_sweeps - sw : g) - 1;
      }
      blk = c_blk;
    }
    c = blk;
    s = threadIdx.x;
    geo.chain0 = c;
  } else {
    const int cpw = 64 / G;
    geo.chain0 = ((int64_t)blockIdx.x * nw + wave) * cpw;
    c = geo.chain0 + lane / G;
    s = lane % G;
  }
  double* tile = lds_tile[wave];
  const double* trow = tile + (MULTI ? lane : s) * (M + 1);  // shared vectors: every group reads rows 0..G-1
  double* crow = tile + lane * (M + 1);                      // per-chain data: one row per lane
  const int lbase = TileMap<M>::lane_base(lane);             // this lane's element of step 0 (coalesced mapping)
  const int pos = MULTI ? lane : s;
  const bool chain_ok = c < A.C;
  const int64_t cc = chain_ok ? c : 0;
  const int64_t n = A.n;
  const int64_t i0 = (int64_t)s * M;
  const int nt = (SMO) ? 2 : A.T.n_terms;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int r0 = TM::lane_col(lane);
  double* tl = tile + lbase;

  double sc[OMC_MAX_TERMS];
  // Sweeps after the first of a launch take the scales their Normal-Gamma blocks redraw from the hand-over line of
  // the chain (written by the workgroup of the previous sweep, possibly on another XCD); the loads are issued here
  // and examined where the scales are first needed (`take_scales`), behind the first pair of draws.
  const bool handed = MULTI && sw > 0;
  // a self-restarting workgroup takes them from its own LDS (written by its wave 0 a moment ago: a poll there costs a
  // hundred cycles, a poll of the global line a trip to L2)
  const bool hand_lds = SIG == 1 && A.reenter != 0 && restarted;
  // LDS comes as the previous workgroup on this CU left it -- possibly this very kernel under another context, whose
  // tags count from 1 like ours: the launch's first sweep wipes the granules (tag 0 is never waited for) long before
  // its epilogue writes them and the second sweep looks
  if (SIG == 1 && A.reenter != 0 && !restarted && threadIdx.x < 2 * OMC_MAX_TERMS)
    __hip_atomic_store(lds_hand + threadIdx.x, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  unsigned long long hw[2 * OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    sc[k] = 1.0;
    hw[2 * k] = hw[2 * k + 1] = 0ull;
    if (k < nt && A.T.scale[k]) {
      if (handed && A.gb[k].enabled) {
        if (!hand_lds) {  // (the LDS granules are read where they are needed: a read there costs nothing worth hiding)
          const unsigned long long* h = A.handoff + cc * OMC_HANDOFF_WORDS + 2 * k;
          hw[2 * k] = __hip_atomic_load(h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hw[2 * k + 1] = __hip_atomic_load(h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        sc[k] = A.T.scale[k][cc];
      }
    }
  }
  // is there a scale that travels from sweep to sweep at all (then waiting for it orders everything else its producer
  // wrote to LDS before it)?
  auto any_handed_f = [&]() {  // (recomputed from the kernel arguments where it is asked: nothing to keep live)
    bool any = false;
#pragma unroll
    for (int k = 0; k < OMC_MAX_TERMS; ++k) any |= (k < nt && A.T.scale[k] && A.gb[k].enabled);
    return any;
  };
  auto take_scales = [&]() {
    if (!handed) return;
    bool lost = false;  // the hand-over never came (reported through `timeouts`): this sweep runs on NaN scales, so that
                        // whatever it stores is recognisably not a sample
    const uint32_t want = A.epoch + (uint32_t)sw;
    auto tags_ok = [&]() {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < OMC_MAX_TERMS; ++k)
        if (k < nt && A.T.scale[k] && A.gb[k].enabled)
          ok = ok && (uint32_t)(hw[2 * k] >> 32) == want && (uint32_t)(hw[2 * k + 1] >> 32) == want;
      return __builtin_amdgcn_readfirstlane((int)ok) != 0;  // every lane loaded the same words
    };
    if (EARLY && OMC_EARLY_ONE_POLLER && wave_u != 0) {
      // SIG 2: wave 0 alone polls the chain's hand-over line in memory; the other waves wait at a BARRIER (no polling
      // traffic of their own, released together the moment wave 0 arrives) and then read what wave 0 left in LDS.
      // Sixteen waves polling back to back got in each other's way: the per-wave timeline showed the last wave seeing
      // the scales 4 000 cycles after the first, and the first scan waits for the last wave.  The LDS words were wiped
      // at the workgroup's start (behind a barrier): LDS arrives as the CU's previous workgroup left it, and that may have
      // been another chain's sweep with exactly the tag waited for here.
      lds_barrier();
#pragma unroll
      for (int k = 0; k < OMC_MAX_TERMS; ++k)
        if (k < nt && A.T.scale[k] && A.gb[k].enabled) {
          hw[2 * k] = __hip_atomic_load(lds_hand + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          hw[2 * k + 1] = __hip_atomic_load(lds_hand + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      lost = !tags_ok();  // (wave 0 passes NaN on under the right tag when the hand-over never came; this cannot fail)
    } else if (hand_lds) {
      // The producer is this workgroup's wave 0, still in the previous sweep's epilogue if this wave is ahead of it: a
      // loop of LDS reads only (no vector-memory wait in it: the previous sweep's x stores are still draining).
      // Bounded; a hand-over that never comes is reported.
      for (int spin = 0;; ++spin) {
#pragma unroll
        for (int k = 0; k < OMC_MAX_TERMS; ++k)
          if (k < nt && A.T.scale[k] && A.gb[k].enabled) {
            hw[2 * k] = __hip_atomic_load(lds_hand + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            hw[2 * k + 1] = __hip_atomic_load(lds_hand + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        if (tags_ok()) break;
        if (spin >= (1 << 22)) {
          if (threadIdx.x == 0 && chain_ok) atomicAdd(A.timeouts, 1ull);
          lost = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    } else {
      // In-order dispatch puts the producer (a lower block index) on the chip first, so this loop normally never
      // turns; it is bounded all the same (about a second), and a hand-over that never comes is reported.
      for (int spin = 0;; ++spin) {
        if (tags_ok()) break;
        // (SIG 2: the consumer was on its CU long before the producer finished -- the poll interval is part of every
        // chain-update's latency; 64 cycles instead of 4096 between looks, the bound scaled to the same ~1 s)
        if (spin >= (EARLY ? (1 << 22) : (1 << 19))) {
          if (threadIdx.x == 0 && chain_ok) atomicAdd(A.timeouts, 1ull);
          lost = true;
          break;
        }
        if (!EARLY) __builtin_amdgcn_s_sleep(64);  // (SIG 2: back-to-back looks, a load round trip apart)
#pragma unroll
        for (int k = 0; k < OMC_MAX_TERMS; ++k)
          if (k < nt && A.T.scale[k] && A.gb[k].enabled) {
            const unsigned long long* h = A.handoff + cc * OMC_HANDOFF_WORDS + 2 * k;
            hw[2 * k] = __hip_atomic_load(h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hw[2 * k + 1] = __hip_atomic_load(h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
      }
    }
#pragma unroll
    for (int k = 0; k < OMC_MAX_TERMS; ++k)
      if (k < nt && A.T.scale[k] && A.gb[k].enabled)
        sc[k] = lost ? __builtin_nan("") : __hiloint2double((int)(uint32_t)hw[2 * k + 1], (int)(uint32_t)hw[2 * k]);
    if (EARLY && OMC_EARLY_ONE_POLLER && wave_u == 0 && lane == 0) {  // pass the scales (or the NaN of a lost hand-over) on
#pragma unroll
      for (int k = 0; k < OMC_MAX_TERMS; ++k)
        if (k < nt && A.T.scale[k] && A.gb[k].enabled) {
          const unsigned long long tg = (unsigned long long)want << 32;
          __hip_atomic_store(lds_hand + 2 * k, tg | (uint32_t)__double2loint(sc[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(lds_hand + 2 * k + 1, tg | (uint32_t)__double2hiint(sc[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (EARLY && OMC_EARLY_ONE_POLLER && wave_u == 0) lds_barrier();  // releases the fifteen waves waiting for these words
  };
  // per-sweep arguments (draw stream, output slab)
  auto nkey_f = [&]() -> omc_rng_key { return (MULTI && run_mode(A)) ? omc_make_key(A.seed, A.rec[sw].draw, OMC_RNG_NORMAL) : A.key; };
  auto x_out = [&]() -> double* { return (MULTI && run_mode(A)) ? A.rec[sw].x : A.x; };
  // SIG 1: which of the two terms is the tridiagonal one (wave-uniform; selects, not indexed kernel arguments)
  const bool p_first = SMO && A.T.diag[0] != nullptr;
  double sP = 1.0, sI = 1.0;  // the two scales by role; selected where first needed (a use up here would put the wait for
                              // the scalar loads in front of the first vector loads and draws)
  const double* const vPd = p_first ? A.T.diag[0] : A.T.diag[1];
  const double* const vPo = p_first ? A.T.off[0] : A.T.off[1];
  const double* const vIr = p_first ? A.T.rhs[1] : A.T.rhs[0];
  const double* const vIc = p_first ? A.T.center[1] : A.T.center[0];
  const double* const vSh = (SHIFT && A.cc.v) ? A.cc.v + cc * A.cc.ld : nullptr;  // SIG 3: this chain's centre vector c

  // diagnostic sweep clock: the constant-rate counter at this wave's entry, kept in two scalar registers to its exit
  unsigned long long t_enter = 0ull;
  if (MULTI && A.sweep_times) t_enter = __builtin_amdgcn_s_memrealtime();
  double Y[M], W[M];
  OMC_STAMP(0);
  // Normal-Gamma standard draws, made up front (see sweep_gamma_draws_wave) by the chain's last wave: its
  // tile is the one that may be partly empty, so it has the least other work
  const bool epi_wave = MULTI && A.fused && wave == 0;
  // the draws are parked in LDS until the epilogue: two registers that would otherwise be live (or, as the
  // compiler prefers, spilled to scratch by every wave) across the whole kernel.  SIG 1 makes them later, where
  // wave 0's SIMD has issue slots to spare (the opening phase is bound by the vector ALU there).
  if (!SMO && epi_wave && chain_ok) {
    bool f = false;
    const double g = sweep_gamma_draws_wave<MULTI && !SMO>(A, c, lane, &f, sw);
    lds_g[lane] = f ? -g : g;  // a Gamma draw is positive; the sign flags a draw that did not terminate
  }

  OMC_STAMP(1);
  // draws: stream position of this segment; SIG 1 makes all but the last pair ahead of the forward pass
  const int64_t gc = A.chain_offset + cc;
  const uint32_t blk0 = (uint32_t)(i0 >> 1);
  const bool gen_z = !A.z && !A.zero_z;
  // SIG 1: may this wave's off-diagonal slice be parked in the draws' LDS slots (see the forward pass)?
  // (SIG 3 parks the chain's own centre slice there instead: that one comes from HBM, the off-diagonal slice from L2)
  const double* const vPark = (SHIFT && OMC_SHIFT_PARK_C) ? vSh : vPo;
  const bool park_off = OMC_PARK_OFF && SMO && gen_z && (A.quad || A.fused) &&
                        wave_valid<M>(wave_u, (int)n - ((SHIFT && OMC_SHIFT_PARK_C) ? 0 : 1)) == 64 * M &&
                        (reinterpret_cast<uintptr_t>(vPark) & 15u) == 0;
  const bool park_diag = OMC_PARK_DIAG && SMO && (A.quad || A.fused) && wave_valid<M>(wave_u, (int)n) == 64 * M &&
                         (reinterpret_cast<uintptr_t>(vPd) & 15u) == 0;
  // SIG 2, the chain's last (partly empty) wave.  It cannot take the LDS-DMA parking as it stands (the transfers would read
  // past the end of the shared vectors) and used to fetch its three quadratic-form vectors inside the phase itself; with
  // a CU to itself per chain every wave waits for that one at the reduction's barrier (the per-wave timeline: 2 000 cycles).
  // Here it gets its own variant: the diagonal slice read back from the staged tile before x overwrites it, the
  // off-diagonal slice parked by transfers whose source is clamped to the last whole 16-byte pair (the consumer masks by
  // index; an odd last element comes from a scalar load), the centre vector prefetched with predicated loads.  The
  // arithmetic and its order are those of the other forms: results stay bit-identical.
  const int e_nv = EARLY ? wave_valid<M>(wave_u, (int)n) : 0, e_nvo = EARLY ? wave_valid<M>(wave_u, (int)n - 1) : 0;
  const bool e_partial = EARLY && (A.quad || A.fused) && e_nv > 0 && e_nv < 64 * M && !(A.rhs_chain && chain_ok);
  const bool park_off_p = OMC_PARK_OFF && e_partial && gen_z && e_nvo >= 2 && (reinterpret_cast<uintptr_t>(vPo) & 15u) == 0;
  double e_edge_o = 0.0;

  // Fewer chains than CUs ((sweep, chain) grid): this workgroup has been placed on an idle CU while the chain's previous
  // sweep is still running elsewhere, and all it can do until that sweep's scales arrive is what does not depend on them --
  // the loads and the draws.  Then ALL buffered pairs are made up here (nothing else is live yet), not spread over the
  // phases behind the hand-over where they would sit on the chain's critical path from sweep to sweep.
  // That is the SIG 2 instantiation (the host picks it for such launches): the three shared vectors are requested first
  // (60 registers that nothing else wants yet), the draws are made while they travel, wave 0 adds the Normal-Gamma standard
  // draws (functions of the priors only), and only then are the scales waited for -- with a tight poll: what follows the
  // hand-over is the chain's critical path from sweep to sweep, and a poll interval is on it.
  // (SIG 1 keeps the run-time form of the early draws: the host no longer asks for it, but without this block the
  // register allocator spills three registers of the hot path)
  const bool early1 = !EARLY && SMO && gen_z && A.early_draws != 0;
  const bool gen_late = gen_z && !EARLY && !early1;
  if constexpr (SMO && !EARLY) {
    if (early1) {
#pragma unroll
      for (int jb = 0; jb < NZB; ++jb) {
        double z0, z1;
        omc_normal_pair(omc_rng_block(nkey_f(), gc, blk0 + (uint32_t)jb), z0, z1);
        lds_z[wave][2 * jb][lane] = z0;
        lds_z[wave][2 * jb + 1][lane] = z1;
      }
    }
  }
  double pre[M];  // SIG 1, 2: the right-hand side vector
  // SIG 2 stages the three vectors while it waits, UNSCALED, in the mapping the recurrences read them in: the off-diagonal
  // slice in Y, the right-hand side in Rrow (both through the tile's transpose), the diagonal slice in the tile itself.
  // When the scales arrive b = sP Y, a_j = sP tile_j + sI and r_j = sI Rrow_j are single operations at the places that read
  // them -- the same values, bit for bit, as the scaled images the other forms write into the tile -- and the three tile
  // fills (thirty LDS operations per wave, bound by the CU's LDS bandwidth: ~1 us) are off the chain's critical path.
  double Rrow[EARLY ? M : 1], ebm1_raw = 0.0, ezl0 = 0.0, ezl1 = 0.0;
  if constexpr (EARLY) {
    // draws first (light on registers), the loads behind them: the workgroup waits for its scales far longer than a load
    // takes, so nothing has to travel under the draws -- and sixty registers of loads in flight beside them would spill
    if (gen_z) {
#pragma unroll
      for (int jb = 0; jb < NZB; ++jb) {
        double z0, z1;
        omc_normal_pair(omc_rng_block(nkey_f(), gc, blk0 + (uint32_t)jb), z0, z1);
        lds_z[wave][2 * jb][lane] = z0;
        lds_z[wave][2 * jb + 1][lane] = z1;
      }
    }
    if (epi_wave && chain_ok) {
      bool f = false;
      const double g = sweep_gamma_draws_wave(A, c, lane, &f, sw);
      lds_g[lane] = f ? -g : g;
    }
    if (OMC_EARLY_LAST_PAIR && gen_z) {
      // the segment's last pair as well: one value goes into the pad slot of this lane's tile row (the slot that staggers
      // the rows over the banks: no tile operation of this form touches it -- the one that would, the transfer of the
      // diagonal slice, does not happen here), the other stays in a register pair
      omc_normal_pair(omc_rng_block(nkey_f(), gc, blk0 + (uint32_t)NZB), ezl0, ezl1);
      crow[M] = ezl0;
    }
    __builtin_amdgcn_sched_barrier(0);
    const int wbase = wave_u * 64 * M;
    const int env = wave_valid<M>(wave_u, (int)n), envo = wave_valid<M>(wave_u, (int)n - 1);
    ebm1_raw = vPo[(i0 > 0 && i0 < n) ? i0 - 1 : 0];
    auto stage = [&](const double* base, int nvalid) {  // coalesced loads -> the wave's tile (zeros beyond the vector's end)
      double tmp[M];
      coal_load<M>(tmp, base, lane, nvalid);
      wave_lds_fence();
#pragma unroll
      for (int t = 0; t < M; ++t) *TM::elem(tl, r0, t) = tmp[t];
      wave_lds_fence();
    };
    if (!(A.rhs_chain && chain_ok)) {
      stage(vIr + wbase, env);
#pragma unroll
      for (int j = 0; j < M; ++j) Rrow[j] = crow[j];
    }
    stage(vPo + wbase, envo);
#pragma unroll
    for (int j = 0; j < M; ++j) Y[j] = crow[j];
    stage(vPd + wbase, env);  // stays in the tile until the pivots are final
    __builtin_amdgcn_sched_barrier(0);
    if (OMC_EARLY_ONE_POLLER && handed) {  // (workgroup-uniform) the LDS hand-over words: wiped before anybody looks
      if (threadIdx.x < 2 * OMC_MAX_TERMS) __hip_atomic_store(lds_hand + threadIdx.x, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      lds_barrier();
    }
  }
  // SIG 0: one parked pair of draws, made under the loads of a tile fill (see PARKZ)
  auto fill_draws = [&](int jb) {
    if constexpr (PARKZ && !SMO) {
      if (gen_z && jb < NZB) {
        double z0, z1;
        omc_normal_pair(omc_rng_block(nkey_f(), gc, blk0 + (uint32_t)jb), z0, z1);
        lds_z[wave][2 * jb][lane] = z0;
        lds_z[wave][2 * jb + 1][lane] = z1;
      }
    }
  };
  // ---- conditional precision: b -> Y (registers), a -> LDS tile ----
  double bm1 = 0.0;  // coupling b_{i0-1} into the segment
  if constexpr (SMO) {
    const int wbase = wave_u * 64 * M;
    const int nv = wave_valid<M>(wave_u, (int)n), nvo = wave_valid<M>(wave_u, (int)n - 1);
    // one vector (20 registers) in flight beside the generation of one pair of draws: more than that spills
    auto vec_and_draws = [&](double (&v)[M], const double* base, int nvalid, int jb) {
      if (gen_late && jb < NZB) {
        double z0, z1;
        if (nvalid == 64 * M) draws_over_load<M, true>(nkey_f(), gc, blk0 + (uint32_t)jb, z0, z1, v, base, lane, nvalid);
        else draws_over_load<M, false>(nkey_f(), gc, blk0 + (uint32_t)jb, z0, z1, v, base, lane, nvalid);
        lds_z[wave][2 * jb][lane] = z0;
        lds_z[wave][2 * jb + 1][lane] = z1;
      } else {
        coal_load<M>(v, base, lane, nvalid);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    {
      double po[M];
      double bm1_raw;
      if constexpr (EARLY) {
        bm1_raw = ebm1_raw;
      } else {
        // b_{i0-1}: only loaded here; any arithmetic on it would put a wait for all loads in front of the draws
        bm1_raw = vPo[(i0 > 0 && i0 < n) ? i0 - 1 : 0];
        vec_and_draws(po, vPo + wbase, nvo, 0);
      }
      take_scales();
      if (SIG == 1 && A.reenter == 2 && handed && hand_lds && wave_u == 1 && chain_ok && any_handed_f()) {
        // the previous sweep's log-posterior, left here by its epilogue (scales: just taken; quadratic forms: LDS)
        double* const lp_prev = A.rec[sw - 1].log_post;
        if (lp_prev) {
          const int k = lane >> 4;
          const double sk = (k == 0) ? sc[0] : ((k == 1) ? sc[1] : ((k == 2) ? sc[2] : sc[3]));
          const double qk = (k < nt) ? __hip_atomic_load(lds_q + (k < nt ? k : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0.0;
          double ldk = 0.0;
          _Pragma("unroll") for (int t = 0; t < OMC_MAX_TERMS; ++t)
            if (t < nt && k == t && A.gb[t].logdet_unscaled) ldk = A.gb[t].logdet_unscaled[0];
          sweep_log_post_wave(A, c, lane, (k < nt) ? sk : 1.0, qk, ldk, lp_prev);
        }
      }
      sP = p_first ? sc[0] : sc[1];
      sI = p_first ? sc[1] : sc[0];
      if constexpr (EARLY) {
#pragma unroll
        for (int j = 0; j < M; ++j) Y[j] *= sP;
        if (!(A.rhs_chain && chain_ok)) {  // r = sI rhs: scaled here, where nothing else is live yet
#pragma unroll
          for (int j = 0; j < M; ++j) Rrow[j] *= sI;
        }
      } else {
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < M; ++t) *TM::elem(tl, r0, t) = sP * po[t];
        wave_lds_fence();
      }
      bm1 = (i0 > 0 && i0 < n) ? sP * bm1_raw : 0.0;
    }
    if constexpr (EARLY) {
      OMC_STAMP(2);
    } else {
      double pd[M];
      vec_and_draws(pd, vPd + wbase, nv, 1);
#pragma unroll
      for (int j = 0; j < M; ++j) Y[j] = crow[j];
      OMC_STAMP(2);
      wave_lds_fence();
      if (nv == 64 * M) {
#pragma unroll
        for (int t = 0; t < M; ++t) *TM::elem(tl, r0, t) = fma(sP, pd[t], sI);
      } else {
#pragma unroll
        for (int t = 0; t < M; ++t) *TM::elem(tl, r0, t) = (lane + 64 * t < nv) ? fma(sP, pd[t], sI) : 1.0;
      }
      wave_lds_fence();
    }
  } else {
    take_scales();
    if (MULTI) tile_fill_comb_wg<M, COMB_OFF>(tile, lane, wave, lbase, A, sc, chain_ok, cc, [&](int b) { if (b == 0) fill_draws(0); });
    else tile_fill_comb<M, MULTI, COMB_OFF>(tile, geo, A, sc);
#pragma unroll
    for (int j = 0; j < M; ++j) Y[j] = crow[j];
    if (i0 > 0 && i0 < n)
      _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < nt)
        if (A.T.off[k]) bm1 = fma(sc[k], A.T.off[k][i0 - 1], bm1);
    OMC_STAMP(2);
    if (MULTI) tile_fill_comb_wg<M, COMB_DIAG>(tile, lane, wave, lbase, A, sc, chain_ok, cc, [&](int b) { if (b == 0) fill_draws(1); });
    else tile_fill_comb<M, MULTI, COMB_DIAG>(tile, geo, A, sc);
  }
  const double* arow = crow;
  // the combined diagonal as the recurrences read it (SIG 2: scaled on the way out of the tile, see above)
  auto a_at = [&](int j) -> double {
    // (beyond the chain's end the staged image is 0, so a = sI there instead of the other forms' 1: those nodes are
    // decoupled from the chain -- b = 0 -- and take part in no result; any positive pivot serves)
    if constexpr (EARLY) return fma(sP, arow[j], sI);
    else return arow[j];
  };

  OMC_STAMP(3);
  // ---- Moebius product of the segment, scan -> incoming pivot ----
  double Dst;
  double Dnext0 = 0.0;  // the start value the NEXT segment derives from the same scan (up to rounding)
  {
    Mob m{1.0, 0.0, 0.0, 1.0};
    double bp = bm1;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const double b2 = bp * bp, aj = a_at(j);
      const double na = fma(aj, m.a, -b2 * m.c), nb = fma(aj, m.b, -b2 * m.d);
      m.c = m.a; m.d = m.b; m.a = na; m.b = nb;
      bp = Y[j];
      if ((j & 7) == 7) {  // cheap guard against overflow inside long segments: scale by the exponent of the leading entry
        const int ex = -__builtin_amdgcn_frexp_exp(m.a);
        m = Mob{ldexp(m.a, ex), ldexp(m.b, ex), ldexp(m.c, ex), ldexp(m.d, ex)};
      }
    }
    m = mob_norm(m);
    OMC_STAMP(4);
    const Mob idm{1.0, 0.0, 0.0, 1.0};
    // SIG 1: waves 1..15 make their last buffered pair of draws while wave 0 scans the wave totals (2 000 cycles in which
    // they would wait at the second barrier); wave 0 makes its own in the right-hand-side phase as before
    auto pair_in_window = [&]() {
      if constexpr (SIG == 1 && OMC_PAIR_IN_SCAN_WINDOW) {
        if (gen_late && NZB > 3) {
          double z0, z1;
          omc_normal_pair(omc_rng_block(nkey_f(), gc, blk0 + (uint32_t)(NZB - 1)), z0, z1);
          lds_z[wave][2 * (NZB - 1)][lane] = z0;
          lds_z[wave][2 * (NZB - 1) + 1][lane] = z1;
        }
      }
    };
    const Mob E = MULTI ? excl_scan_wg<Mob, false, true>(m, idm, lds_mob, lane, wave, nw, lds_mob2, pair_in_window)
                        : excl_scan<Mob, false>(m, idm, pos, Wd, false, lds_mob, wave, nw);
    Dst = (E.a + E.b) / (E.c + E.d);
    if (A.perturb_start != 0.0 && s > 0) Dst *= 1.0 + A.perturb_start;  // tests: a start the join test must reject
    if (MULTI) {
      const Mob inc = compose(m, E);
      Dnext0 = (inc.a + inc.b) * fast_rcp(inc.c + inc.d);
      if (A.perturb_start != 0.0) Dnext0 *= 1.0 + A.perturb_start;  // tests: the successor's start is spoiled the same way
    }
  }

  OMC_STAMP(5);
  // ---- true pivot recurrence, Newton multiple shooting on the segment joins ----
  bool bad = false;
  double lin = 0.0;  // l_{i0-1}
  // one pass of the true recurrence over the segment from Dst: W = 1/D, returns the last pivot
  auto pivot_pass = [&]() -> double {
    wave_lds_fence();  // re-read a from LDS every pass instead of keeping a register copy
    const double rst = fast_rcp(Dst);
    lin = bm1 * rst;
    double lp = lin, bprev = bm1, Dend = Dst;
    bool badp = false;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const double D = fma(-lp, bprev, a_at(j));
      badp |= !(D > 0.0);
      const double r = fast_rcp(D);
      W[j] = r;
      bprev = Y[j];
      lp = bprev * r;
      Dend = D;
    }
    bad = badp;
    return Dend;
  };
  double Dend = pivot_pass();
  // Join test.  Workgroup form: every segment tests the join at its END against the start value its successor
  // took from the Moebius scan, which it can compute itself (Dnext0): no neighbour exchange, one barrier.  97 %
  // of the cfg3 chains stop here.  Otherwise (and in the sub-wave form) the joins are tested at the segment
  // starts with an exchange, and corrected: Newton on the joins with Jacobian prod l^2, one affine scan each.
  bool settled = false;
  if (MULTI) {
    const bool has_next = i0 + M < n;
    const int need = (has_next && fabs(Dend - Dnext0) > OMC_NEWTON_TOL * fabs(Dnext0)) ? 1 : 0;  // false for NaN: -> `bad`
#if OMC_JOIN_OR_LIB
    settled = !__syncthreads_or(need);
#else
    settled = !any_wg(need, lds_any, lane, wave, nw);
#endif
  }
  for (int it = 0; !settled; ++it) {
    double J = lin * lin;  // d(last pivot)/d(start pivot) of this segment = prod of l^2 over it
#pragma unroll
    for (int j = 0; j + 1 < M; ++j) {
      const double l = Y[j] * W[j];
      J *= l * l;
    }
    double Dp = Dend, Jp = J;
    if (MULTI) prev_lane2_wg(Dp, Jp, Dst, 0.0, lds_x[it & 1], lane, wave, nw);
    else prev_lane2<false>(Dp, Jp, Dst, 0.0, pos, Wd, lds_x[0], wave);
    const bool joined = (s > 0 && i0 < n);
    const double e = joined ? (Dp - Dst) : 0.0;
    if (!joined) Jp = 0.0;
    const int need = (fabs(e) > OMC_NEWTON_TOL * fabs(Dst)) ? 1 : 0;  // false for NaN: falls through to `bad`
    const int any = MULTI ? __syncthreads_or(need) : (__ballot(need) != 0ull);  // (rare path: any_wg here costs the hot path 3 spilled registers)
    if (!any) break;
    if (it >= A.newton_max) {
      // Newton has not brought every join below the tolerance (a recurrence that is not contractive over a
      // segment: weak coupling, or |l| > 1 on a stretch).  Nothing is left to chance from here: the joins are
      // made consistent by the sequential recurrence itself.  Every pass starts each segment from the TRUE end
      // value of its predecessor's last pass, so after pass k the first k+1 segments carry exactly the pivots of
      // the serial kernel, and where the recurrence contracts, the rest converges geometrically at the same time;
      // the loop stops when every join meets the tolerance the Newton path accepts, at the latest after one pass per
      // segment.  (It used to insist on bit-equal joins.  On a homogeneous chain -- every segment the same map, as in
      // the headline model -- that is a worst case by construction: the map has two floating-point fixed points one
      // ulp apart, the part of the chain that converged from the spoiled starts sits on the other one than the part
      // propagated from the chain's head, and the border between them moves one segment per pass: all ~1000 passes,
      // 1.4 ms per chain-update measured by benchmarks/join_fallback_cost.py, for a difference of one ulp.)
      // About 1.4 us per pass; rare; counted in `fallbacks`.
      if (A.fallbacks && chain_ok && s == 0) atomicAdd(A.fallbacks, 1ull);
      const int S = MULTI ? (int)blockDim.x : Wd;
      for (int pass = 0; pass < S; ++pass) {
        double Dq = Dend, Jq = 0.0;
        if (MULTI) prev_lane2_wg(Dq, Jq, Dst, 0.0, lds_x[pass & 1], lane, wave, nw);
        else prev_lane2<false>(Dq, Jq, Dst, 0.0, pos, Wd, lds_x[0], wave);
        // (a NaN pivot compares false: it is `bad`, not a reason to go on)
        const int moved = (joined && fabs(Dq - Dst) > OMC_NEWTON_TOL * fabs(Dst)) ? 1 : 0;
        const int some = MULTI ? __syncthreads_or(moved) : (__ballot(moved) != 0ull);
        if (!some) break;
        if (joined) Dst = Dq;
        Dend = pivot_pass();
      }
      break;
    }
    const Aff own{e, Jp};
    const Aff ex = MULTI ? excl_scan_wg<Aff, false>(own, Aff{0.0, 1.0}, lds_aff[it & 1], lane, wave, nw)
                          : excl_scan<Aff, false>(own, Aff{0.0, 1.0}, pos, Wd, false, lds_aff[0], wave, nw);
    Dst += fma(Jp, ex.p, e);  // delta_s = e_s + J_{s-1} delta_{s-1}
    Dend = pivot_pass();
  }
  OMC_STAMP(6);
  double logdet = 0.0;
  // a chain with a non-positive pivot is reported through `bad`; its lanes continue on 1/D = 1 so that nothing
  // downstream sees the square root of a negative number (wave-uniform branch: no per-node selects)
  if (__ballot(bad) != 0ull) {
#pragma unroll
    for (int j = 0; j < M; ++j) W[j] = bad ? 1.0 : W[j];
  }
#pragma unroll
  for (int j = 0; j < M; ++j) {
    Y[j] *= W[j];               // l_j = b_j / D_j
    if (A.logdet && i0 + j < n) logdet -= log(W[j]);
  }

  OMC_STAMP(7);
  // ---- right-hand side -> tile; forward substitution (local affine map, scan, true pass) ----
  bool rhs_done = false;
  if constexpr (SMO) {
    // per-chain offsets (rhs_chain) go through the general fill below; the draws are made ahead in either case
    const bool with_offsets = A.rhs_chain && chain_ok;
    if (SHIFT && !with_offsets && !vIr) {  // SIG 3 without a shared centre: nothing to load, the pair is made plainly
#pragma unroll
      for (int t = 0; t < M; ++t) pre[t] = 0.0;
      if (gen_late && NZB > 2) {
        double z0, z1;
        omc_normal_pair(omc_rng_block(nkey_f(), gc, blk0 + 2u), z0, z1);
        lds_z[wave][4][lane] = z0;
        lds_z[wave][5][lane] = z1;
      }
    } else if (!with_offsets && !EARLY) {  // (SIG 2 asked for the vector at its start)
      const int nvr = wave_valid<M>(wave_u, (int)n);
      const double* base = vIr + wave_u * 64 * M;
      if (gen_late && NZB > 2) {
        double z0, z1;
        if (nvr == 64 * M) draws_over_load<M, true>(nkey_f(), gc, blk0 + 2u, z0, z1, pre, base, lane, nvr);
        else draws_over_load<M, false>(nkey_f(), gc, blk0 + 2u, z0, z1, pre, base, lane, nvr);
        lds_z[wave][4][lane] = z0;
        lds_z[wave][5][lane] = z1;
      } else {
        coal_load<M>(pre, base, lane, nvr);
      }
    }
    double csh[SHIFT ? M : 1];  // SIG 3: the chain's centre slice, requested here so that it travels under the next pair of draws
    if constexpr (SHIFT) {
      if (!with_offsets) {
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < M; ++t) *TM::elem(tl, r0, t) = sI * pre[t];  // (frees `pre`: one vector in flight beside a pair of draws)
        wave_lds_fence();
        coal_load<M>(csh, vSh + wave_u * 64 * M, lane, wave_valid<M>(wave_u, (int)n));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (gen_late) {
#pragma unroll
      for (int jb = 2; jb < NZB; ++jb) {
        if (jb == 2 && !with_offsets) continue;  // made under the load above
        if (SIG == 1 && OMC_PAIR_IN_SCAN_WINDOW && NZB > 3 && jb == NZB - 1 && wave_u != 0) continue;  // made in the Moebius scan's window
        double z0, z1;
        omc_normal_pair(omc_rng_block(nkey_f(), gc, blk0 + (uint32_t)jb), z0, z1);
        lds_z[wave][2 * jb][lane] = z0;
        lds_z[wave][2 * jb + 1][lane] = z1;
      }
    }
    if (!with_offsets) {
      if constexpr (SHIFT) {  // r = sI (ys - c): the shared part is in the tile already
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < M; ++t) {
          double* pe = TM::elem(tl, r0, t);
          *pe = fma(-sI, csh[t], *pe);
        }
        wave_lds_fence();
      } else if constexpr (!EARLY) {  // (SIG 2: Rrow holds the scaled vector since the scales arrived)
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < M; ++t) *TM::elem(tl, r0, t) = sI * pre[t];
        wave_lds_fence();
      }
      rhs_done = true;
    }
  }
  if (rhs_done) {
  } else if (MULTI) {
    // (per-chain centres: SIG 0 only; SIG 0 makes two more pairs of draws under this fill's loads)
    if constexpr (SMO) tile_fill_comb_wg<M, COMB_RHS, false>(tile, lane, wave, lbase, A, sc, chain_ok, cc);
    else tile_fill_comb_wg<M, COMB_RHS, true>(tile, lane, wave, lbase, A, sc, chain_ok, cc, [&](int b) { fill_draws(2 + b); });
    if constexpr (EARLY) {  // (per-chain offsets: the general fill above; read out once, like the staged vector)
#pragma unroll
      for (int j = 0; j < M; ++j) Rrow[j] = crow[j];
    }
  } else {
    tile_fill_comb<M, MULTI, COMB_RHS>(tile, geo, A, sc);
  }
  constexpr bool PFQ = SMO && OMC_PREFETCH_QUAD;
  constexpr int PFQ_LOADS = M + (OMC_PREFETCH_QUAD > 1 ? M - 2 * NZB : 0);  // loads the prefetch puts behind the LDS-DMA
  static_assert(!PFQ || PFQ_LOADS <= 15, "vmcnt immediate");
  double qcp[PFQ ? M : 1], qop[PFQ ? M : 1];
  double qcc[SHIFT ? M : 1];  // SIG 3: the chain's centre slice in the coalesced mapping (quadratic form's centre, x = c + e)
  bool pfq = false;
  // vector-memory loads issue_pfq has put on the wire, counted WHERE they are issued: the count-based wait in front of the parked
  // diagonal (below) is taken only if this says that at least PFQ_LOADS loads went out behind the transfer -- the wait's safety
  // follows from the counter, not from a remark about which paths issue loads (round 3's race was such a remark going stale)
  int pfq_behind = 0;
  auto issue_pfq = [&]() {
    if constexpr (PFQ) {
      const bool wq = A.quad || A.fused;
      pfq = wq && park_off && wave_valid<M>(wave_u, (int)n) == 64 * M;  // wave-uniform; other waves load in the phase itself
      __builtin_amdgcn_sched_barrier(0);  // not into the forward pass: its registers are all taken
      if constexpr (EARLY) {
        if (e_partial) {  // (wave-uniform) the last wave's centre slice, predicated
          const int wbase = wave_u * 64 * M;
#pragma unroll
          for (int t = 0; t < M; ++t) qcp[t] = (lane + 64 * t < e_nv) ? (vIc + wbase)[(unsigned)(lane + 64 * t)] : 0.0;
        }
      }
      if (pfq) {
        const int wbase = wave_u * 64 * M;
        if (!SHIFT || vIc) {
#pragma unroll
          for (int t = 0; t < M; ++t) qcp[t] = (vIc + wbase)[(unsigned)(lane + 64 * t)];
          pfq_behind += M;
        } else {
#pragma unroll
          for (int t = 0; t < M; ++t) qcp[t] = 0.0;
        }
        if constexpr (SHIFT && OMC_SHIFT_PREFETCH) {
#pragma unroll
          for (int t = 0; t < M; ++t) qcc[t] = (vSh + wbase)[(unsigned)(lane + 64 * t)];
          pfq_behind += M;
        }
        if (OMC_PREFETCH_QUAD > 1) {
#pragma unroll
          for (int t = 2 * NZB; t < M; ++t) qop[t] = (vPark + wbase)[(unsigned)(lane + 64 * t)];
          pfq_behind += M - 2 * NZB;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  OMC_STAMP(8);
  if (EARLY && OMC_EARLY_PFQ_AHEAD) issue_pfq();
  auto r_at = [&](int j) -> double {  // the right-hand side as the forward substitution reads it
    if constexpr (EARLY) return Rrow[j];
    else return crow[j];
  };
  {
    Aff f{0.0, 1.0};
    double lp = lin;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      f.p = fma(-lp, f.p, r_at(j));
      f.q = -lp * f.q;
      lp = Y[j];
    }
    double u = (MULTI ? excl_scan_wg<Aff, false>(f, Aff{0.0, 1.0}, lds_aff[2], lane, wave, nw)
                      : excl_scan<Aff, false>(f, Aff{0.0, 1.0}, pos, Wd, false, lds_aff[0], wave, nw)).p;
    lp = lin;
    OMC_STAMP(9);
    // g_j = u_j/D_j + z_j/sqrt(D_j); the draws are produced here, pair by pair, so that no array of
    // z ever has to be kept in registers next to l and 1/D (Philox + Box-Muller interleave with the
    // serial u recurrence; the scheduling barrier keeps the five bodies from being overlapped)
    const double* zin = A.z ? A.z + cc * A.ld_z + i0 : nullptr;
#pragma unroll
    for (int j = 0; j < M; j += 2) {
      double z0 = 0.0, z1 = 0.0;
      if (zin) {
        if (i0 + j < n) z0 = zin[j];
        if (i0 + j + 1 < n) z1 = zin[j + 1];
        // injected draws (tests): waited for HERE.  Left pending, these loads meet the in-kernel-draw path at the join below,
        // and the compiler's wait-count pass -- which must assume either predecessor -- then puts an `s_waitcnt vmcnt(0)`
        // into the shared code: behind the quadratic-form prefetches of SIG 1 that wait exposed the whole L2 latency of
        // twelve loads on every sweep of the production path.
        if (SMO) __builtin_amdgcn_s_waitcnt(0x0F70);
      } else if (PARKZ && (j >> 1) < NZB) {
        if (gen_z) { z0 = lds_z[wave][j][lane]; z1 = lds_z[wave][j + 1][lane]; }
      } else if (!A.zero_z) {
        if constexpr (SMO) {
          // The parked draws have all been read: their LDS slots now take the first 128 NZB entries of this
          // wave's slice of the off-diagonal vector for the quadratic forms (LDS-DMA, no registers), under the
          // generation of the segment's last pair of draws.
          if (j == 2 * NZB && park_off) {
            lds_reads_done();
#pragma unroll
            for (int k = 0; k < NZB; ++k)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vPark + wave_u * 64 * M + 128 * k + 2 * lane),
                                               (__attribute__((address_space(3))) void*)&lds_z[wave][2 * k][0], 16, 0, 0);
          }
          if constexpr (EARLY) {
            if (j == 2 * NZB && park_off_p) {  // (wave-uniform)
              lds_reads_done();
              const int last_pair = (e_nvo - 2) & ~1;
#pragma unroll
              for (int k = 0; k < NZB; ++k) {
                const int e = 128 * k + 2 * lane;
                const int ec = (e + 1 < e_nvo) ? e : last_pair;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vPo + wave_u * 64 * M + ec),
                                                 (__attribute__((address_space(3))) void*)&lds_z[wave][2 * k][0], 16, 0, 0);
              }
              if (e_nvo & 1) e_edge_o = vPo[wave_u * 64 * M + e_nvo - 1];
            }
          }
        }
        if constexpr (EARLY && OMC_EARLY_LAST_PAIR) {
          z0 = crow[M]; z1 = ezl1;  // (made while the scales were waited for: one pair of draws less on the critical path)
        } else {
          omc_normal_pair(omc_rng_block(nkey_f(), gc, blk0 + (uint32_t)(j >> 1)), z0, z1);
        }
      }
      u = fma(-lp, u, r_at(j));
      W[j] = fma(u, W[j], z0 * fast_sqrt(W[j]));
      lp = Y[j];
      u = fma(-lp, u, r_at(j + 1));
      W[j + 1] = fma(u, W[j + 1], z1 * fast_sqrt(W[j + 1]));
      lp = Y[j + 1];
      if (!(PARKZ && (j >> 1) < NZB - 1)) __builtin_amdgcn_sched_barrier(0);  // parked draws: let the pairs pipeline
    }
  }

  OMC_STAMP(10);
  const bool want_quad = A.quad || A.fused;
  // SIG 1: the tile's right-hand side is dead now; until x is written into it, it takes this wave's slice of
  // the tridiagonal term's diagonal (LDS-DMA, contiguous image), which the back pass below reads in the row
  // mapping for the x' diag x part of the quadratic form -- one vector less to wait for afterwards
  double aPd = 0.0;
  double eqd[EARLY ? M : 1];  // SIG 2, last wave: its diagonal slice in the coalesced mapping, taken before x overwrites the tile
  // SIG 2 without per-chain offsets: the wave's diagonal slice has been sitting in the tile, unscaled, since the workgroup
  // started (nothing wrote the tile after the pivots): no transfer, the back pass reads the staged rows
  const bool diag_staged = EARLY && !(A.rhs_chain && chain_ok);
  if constexpr (SMO) {
    if (park_diag && !diag_staged) {
      lds_reads_done();
#pragma unroll
      for (int k = 0; k < (64 * M) / 128; ++k)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vPd + wave_u * 64 * M + 128 * k + 2 * lane),
                                         (__attribute__((address_space(3))) void*)(tile + 128 * k), 16, 0, 0);
    }
  }
  // SIG 1: the rest of what the quadratic forms read from L2 -- the centre vector and the part of the off-diagonal
  // slice that did not fit beside the parked draws -- is fetched here into registers: the reverse scan and the back
  // pass (a tenth of the wave's lifetime, light on registers) cover its latency, the quadratic-form phase then reads
  // nothing from memory and its x stores start a load round trip earlier (OMC_PREFETCH_QUAD; measured on
  // benchmarks/ab_headline.py).
  // (SIG 2 has issued the prefetch before its forward substitution, which generates no draws there: the loads'
  // issue overlaps arithmetic instead of holding up the reverse scan, see OMC_EARLY_PFQ_AHEAD)
  if (!(EARLY && OMC_EARLY_PFQ_AHEAD)) issue_pfq();
  // ---- backward substitution: local affine map, reverse scan, true pass ----
  double xnext;
  {
    Aff f{0.0, 1.0};
#pragma unroll
    for (int j = M - 1; j >= 0; --j) {
      f.p = fma(-Y[j], f.p, W[j]);
      f.q = -Y[j] * f.q;
    }
    xnext = (MULTI ? excl_scan_wg<Aff, true>(f, Aff{0.0, 1.0}, lds_aff[3], lane, wave, nw)
                   : excl_scan<Aff, false>(f, Aff{0.0, 1.0}, pos, Wd, true, lds_aff[0], wave, nw)).p;
    double x = xnext;
    OMC_STAMP(11);
    if constexpr (EARLY) {
      if (e_partial) {
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < M; ++t) eqd[t] = *TM::elem(tl, r0, t);
      }
    }
    if (SMO && park_diag) {
      // the LDS-DMA has landed: vector-memory operations retire in order, so it is enough that no more than the prefetch
      // loads issued BEHIND it are still out (they are not needed before the quadratic forms)
      // (SIG 3 without a shared centre issues NO prefetch loads -- issue_pfq sets the slice to zero --, and then "at most
      // PFQ_LOADS still out" says nothing about the transfer: the diagonal was read before it had landed now and then, and
      // the late transfer overwrote the x this wave had meanwhile put into the tile.  One chain in a few thousand sweeps of
      // the hierarchical smoother at n = 10 000 x 1024 chains, found by benchmarks/determinism_hier.py.)
      if (!diag_staged) {
        // (issue_pfq ran behind the transfer -- not in the EARLY && OMC_EARLY_PFQ_AHEAD order -- and left at least PFQ_LOADS
        //  loads behind it: then "at most PFQ_LOADS still out" means the transfer is not among them)
        if (PFQ && !(EARLY && OMC_EARLY_PFQ_AHEAD) && pfq_behind >= PFQ_LOADS) __builtin_amdgcn_s_waitcnt(0x0F70 | PFQ_LOADS);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      wave_lds_fence();
      const double* drow = diag_staged ? crow : tile + lane * M;  // the staged rows (padded), or the transfer's contiguous image
#pragma unroll
      for (int j = M - 1; j >= 0; --j) {
        x = fma(-Y[j], x, W[j]);
        W[j] = x;
        aPd = fma(drow[j] * x, x, aPd);
      }
    } else {
#pragma unroll
      for (int j = M - 1; j >= 0; --j) {
        x = fma(-Y[j], x, W[j]);
        W[j] = x;
      }
    }
  }
  OMC_STAMP(12);
  double qsum[OMC_MAX_TERMS] = {0, 0, 0, 0};
  double my_scale = 1.0, my_logdet = 0.0;  // epilogue scalars of this lane's term (wave 0)
  if (MULTI) {
    // ---- store + fused quadratic forms, both in the coalesced mapping: lane handles nodes
    //      wbase + t*64 + lane; x comes back from the tile, the shared vectors straight from L2 ----
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < M; ++j) crow[j] = W[j];
    // x at the first node of the NEXT wave's tile = what the reverse scan handed this wave's last segment as
    // its successor value; it goes behind the last row so that every node finds x_{i+1} one element on
    if (lane == 63) tile[64 * (M + 1)] = xnext;
    wave_lds_fence();  // wave-private tile: no workgroup barrier needed
    double acc[OMC_MAX_TERMS] = {0, 0, 0, 0};
    // scalars of the epilogue: issue their loads now so the latency hides behind the quad phase
    if (epi_wave) {  // lane group k = lane >> 4 serves term k
      _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < nt && (lane >> 4) == k) {
        if (A.T.scale[k]) my_scale = handed ? sc[k] : A.T.scale[k][cc];
        if constexpr (SMO) {
          if (sweep_log_post(A, sw) && A.gb[k].logdet_unscaled) my_logdet = A.gb[k].logdet_unscaled[0];
        }
      }
      if constexpr (!SMO) {  // (the blocks' device image: see TriArgs::gb_dev)
        const double* const ldp = ((lane >> 4) < nt) ? A.gb_dev[lane >> 4].logdet_unscaled : nullptr;
        if (sweep_log_post(A, sw) && ldp) my_logdet = ldp[0];
      }
    }
    if constexpr (SMO) {
      // Normal-Gamma standard draws (functions of the priors only): here, in front of the load-bound phase of
      // the quadratic forms, wave 0's delay costs nothing -- the other waves' loads keep the L2 path busy
      if (!EARLY && epi_wave && chain_ok) {  // (SIG 2: made at the start, while the scales were waited for)
        bool f = false;
        const double g = sweep_gamma_draws_wave(A, c, lane, &f, sw);
        lds_g[lane] = f ? -g : g;
      }
      const int nv = wave_valid<M>(wave_u, (int)n);
      double qc[M], qd[M], qo[M];
      {
        const int wbase = wave_u * 64 * M;
        const int nvq = want_quad ? nv : 0, nvo = want_quad ? wave_valid<M>(wave_u, (int)n - 1) : 0;
        {
          if (park_off && SHIFT && OMC_SHIFT_PARK_C) {  // the parked slice is the chain's centre; the off-diagonal comes from L2
            const double* zf = &lds_z[wave][0][0];
            coal_load<M>(qo, vPo + wbase, lane, nvo);
            // the transfers (older than these M loads) have landed.  The M loads are M instructions issued right here on every
            // path: park_off says the wave is full, so nvo >= 64 M - 1 and no load has all its lanes predicated off
            __builtin_amdgcn_s_waitcnt(0x0F70 | M);
            wave_lds_fence();
#pragma unroll
            for (int t = 0; t < M; ++t)
              if (t < 2 * NZB) qcc[t] = zf[64 * t + lane];
#pragma unroll
            for (int t = 0; t < M; ++t) {
              if (t < 2 * NZB) continue;
              if constexpr (PFQ && OMC_PREFETCH_QUAD > 1) qcc[t] = pfq ? qop[t] : (vSh + wbase)[(unsigned)(lane + 64 * t)];
              else qcc[t] = (vSh + wbase)[(unsigned)(lane + 64 * t)];
            }
          } else if (park_off) {
            const double* zf = &lds_z[wave][0][0];
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the LDS-DMA has landed (the reverse scan's barrier drained it already)
            wave_lds_fence();
#pragma unroll
            for (int t = 0; t < M; ++t)
              if (t < 2 * NZB) qo[t] = zf[64 * t + lane];
#pragma unroll
            for (int t = 0; t < M; ++t) {
              if (t < 2 * NZB) continue;
              if constexpr (PFQ && OMC_PREFETCH_QUAD > 1) qo[t] = pfq ? qop[t] : (vPo + wbase)[(unsigned)(lane + 64 * t)];
              else qo[t] = OMC_WHATIF_NOQLOAD ? 0.25 : (vPo + wbase)[(unsigned)(lane + 64 * t)];
            }
          } else if (EARLY && park_off_p) {
            const double* zf = &lds_z[wave][0][0];
            __builtin_amdgcn_s_waitcnt(0x0F70);  // the transfers (and the centre prefetch behind them) have landed
            wave_lds_fence();
#pragma unroll
            for (int t = 0; t < M; ++t) {
              const int idx = lane + 64 * t;
              double v = 0.0;
              if (t < 2 * NZB) v = zf[idx];
              else if (idx < nvo) v = (vPo + wbase)[(unsigned)idx];  // (a last wave of more than 512 nodes)
              qo[t] = (idx < nvo) ? (((nvo & 1) && idx == nvo - 1) ? e_edge_o : v) : 0.0;
            }
          } else {
            coal_load<M>(qo, vPo + wbase, lane, nvo);
          }
        }
        if (EARLY && e_partial) {
#pragma unroll
          for (int t = 0; t < M; ++t) qd[t] = eqd[t];
        } else if (park_diag) {
#pragma unroll
          for (int t = 0; t < M; ++t) qd[t] = 0.0;  // that part is in aPd already
        } else {
          coal_load<M>(qd, vPd + wbase, lane, nvq);
        }
        if (OMC_WHATIF_NOQLOAD) {
#pragma unroll
          for (int t = 0; t < M; ++t) qc[t] = 1.0;
        } else if (PFQ && (pfq || (EARLY && e_partial))) {
#pragma unroll
          for (int t = 0; t < M; ++t) qc[t] = qcp[t];
        } else if (SHIFT && !vIc) {
#pragma unroll
          for (int t = 0; t < M; ++t) qc[t] = 0.0;
        } else {
          coal_load<M>(qc, vIc + wbase, lane, nvq);
        }
        if constexpr (SHIFT) {
          if (!(PFQ && pfq && OMC_SHIFT_PREFETCH) && !(park_off && OMC_SHIFT_PARK_C)) coal_load<M>(qcc, vSh + wbase, lane, nv);
#pragma unroll
          for (int t = 0; t < M; ++t) qc[t] -= qcc[t];  // the identity term's centre in e-coordinates: ys - c
        }
      }
      double aI = 0.0, aP = aPd;
      // x leaves from the same pass, behind the loads issued above (vmcnt retires in order: nothing waits on the
      // x stream, and the 80 KB of stores drain under the reduction and the epilogue instead of after them)
      double* const xb = x_out();
      double* xo = (xb && chain_ok) ? xb + cc * A.ld_x + wave_u * 64 * M : nullptr;
      // SIG 2: the stores wait until the quadratic forms are reduced and wave 0 has handed the new scales over (the block
      // at the kernel's end): between two hand-overs nothing uses the memory pipeline that the next hand-over does not need
      if (EARLY && OMC_EARLY_DEFER_STORE && want_quad) xo = nullptr;
      if (!want_quad) {
        if (xo) {
#pragma unroll
          for (int t = 0; t < M; ++t)
            if (lane + 64 * t < nv) xo[(unsigned)(lane + 64 * t)] = SHIFT ? *TM::elem(tl, r0, t) + qcc[t] : *TM::elem(tl, r0, t);
        }
      } else if (nv == 64 * M && park_diag) {  // the diagonal part is in aPd already
#pragma unroll
        for (int t = 0; t < M; ++t) {
          const double* pe = TM::elem(tl, r0, t);
          const double xv = *pe, xn = *TM::succ(pe, r0, t), a = xv - qc[t];
          aI = fma(a, a, aI);
          aP = fma(2.0 * qo[t] * xn, xv, aP);
          if (xo && !OMC_WHATIF_NOSTORE) {
            const double xs = SHIFT ? xv + qcc[t] : xv;  // (SIG 3: x = c + e)
            if (OMC_STORE_NT) __builtin_nontemporal_store(xs, &xo[(unsigned)(lane + 64 * t)]);
            else xo[(unsigned)(lane + 64 * t)] = xs;
          }
        }
      } else if (nv == 64 * M) {
#pragma unroll
        for (int t = 0; t < M; ++t) {
          const double* pe = TM::elem(tl, r0, t);
          const double xv = *pe, xn = *TM::succ(pe, r0, t), a = xv - qc[t];
          aI = fma(a, a, aI);
          aP = fma(fma(2.0 * qo[t], xn, qd[t] * xv), xv, aP);
          if (xo && !OMC_WHATIF_NOSTORE) {
            const double xs = SHIFT ? xv + qcc[t] : xv;  // (SIG 3: x = c + e)
            if (OMC_STORE_NT) __builtin_nontemporal_store(xs, &xo[(unsigned)(lane + 64 * t)]);
            else xo[(unsigned)(lane + 64 * t)] = xs;
          }
        }
      } else {  // the chain's last wave: nodes beyond n hold finite fill values, their vectors were loaded as 0
#pragma unroll
        for (int t = 0; t < M; ++t) {
          const double* pe = TM::elem(tl, r0, t);
          const double xv = *pe, xn = *TM::succ(pe, r0, t), a = (lane + 64 * t < nv) ? xv - qc[t] : 0.0;
          aI = fma(a, a, aI);
          aP = fma(fma(2.0 * qo[t], xn, qd[t] * xv), xv, aP);
          if (xo && lane + 64 * t < nv) xo[(unsigned)(lane + 64 * t)] = SHIFT ? xv + qcc[t] : xv;
        }
      }
      acc[0] = p_first ? aP : aI;
      acc[1] = p_first ? aI : aP;
    } else {
      if (want_quad) quad_wg<M>(tile, lane, wave_u, lbase, A, acc, cc);
    }
    OMC_STAMP(13);
    if (want_quad) {
      sum4_wg(acc, qsum, nt, &lds_d[0][0], lane, wave, nw);  // all terms behind one barrier
      _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < nt) {
        if (A.quad && s == 0 && chain_ok) A.quad[k * A.C + c] = qsum[k];
      }
    }
  } else {
  if (A.x) {
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < M; ++j) crow[j] = W[j];
    tile_store_chain<M, MULTI>(tile, geo, A.x, A.ld_x, n, A.C);
  }

  // ---- fused quadratic forms (x - m_k)' M_k (x - m_k), segment-local form ----
  if (want_quad) {
    double X[M];  // residuals
    _Pragma("unroll") for (int k = 0; k < OMC_MAX_TERMS; ++k) if (k < nt) {
      // residual of this segment in X, residual of the next segment's first node in rn
      double rn = 0.0;
      if (A.T.center[k]) {
        tile_fill_shared<M, MULTI>(tile, geo, A.T.center[k], n, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j) X[j] = W[j] - trow[j];
        if (i0 + M < n) rn = xnext - A.T.center[k][i0 + M];
      } else {
#pragma unroll
        for (int j = 0; j < M; ++j) X[j] = W[j];
        if (i0 + M < n) rn = xnext;
      }
      double acc = 0.0;
      if (A.T.diag[k]) {
        tile_fill_shared<M, MULTI>(tile, geo, A.T.diag[k], n, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j) acc = fma(trow[j] * X[j], X[j], acc);
      } else {
#pragma unroll
        for (int j = 0; j < M; ++j)
          if (i0 + j < n) acc = fma(X[j], X[j], acc);
      }
      if (A.T.off[k]) {
        tile_fill_shared<M, MULTI>(tile, geo, A.T.off[k], n - 1, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j) acc = fma(2.0 * trow[j] * X[j], (j + 1 < M) ? X[(j + 1) % M] : rn, acc);
      }
      qsum[k] = group_sum<false>(acc, Wd, lds_d[0], wave, nw);
      if (A.quad && s == 0 && chain_ok) A.quad[k * A.C + c] = qsum[k];
    }
  }
  }
  OMC_STAMP(14);
  if (A.logdet) {
    const double t = MULTI ? sum_wg(logdet, lds_d[4], lane, wave, nw) : group_sum<false>(logdet, Wd, lds_d[0], wave, nw);
    if (s == 0 && chain_ok) A.logdet[c] = t;
  }
  if (bad && chain_ok) atomicMin((unsigned long long*)A.bad, (unsigned long long)c);
  if (MULTI) {
    if (epi_wave && chain_ok) {
      const double g = lds_g[lane];
      // restart without a barrier: the log-posterior of this sweep is left to wave 1 of the next one (it has the slack
      // this wave does not: everyone waits for the wave that ran the epilogue at the next sweep's first barrier)
      const bool defer_lp = SIG == 1 && A.reenter == 2 && left > 0 && nw > 1 && any_handed_f();
      sweep_epilogue_wave<!SMO>(A, c, qsum[0], qsum[1], qsum[2], qsum[3], my_scale, my_logdet, fabs(g), g < 0.0, lane, sw,
                          (SIG == 1 && A.reenter) ? lds_hand : nullptr, defer_lp, lds_q);
    }
    // x leaves last: a load issued behind a store would have to wait for the store to be
    // acknowledged (vmcnt retires in order); this way nothing ever waits on the x stream.
    // (SIG 1 has stored it from its quadratic-form pass already.)
    const bool store_here = !SMO || (EARLY && OMC_EARLY_DEFER_STORE && (A.quad || A.fused));
    double* const xb = store_here ? x_out() : nullptr;
    if (store_here && xb && chain_ok) {
      double* xo = xb + cc * A.ld_x + wave_u * 64 * M;
      const int nvalid = wave_valid<M>(wave_u, (int)n);
      {
        if (nvalid == 64 * M) {
#pragma unroll
          for (int t = 0; t < TM::NS; ++t)
            xo[(unsigned)(lane + t * TM::LU)] = *TM::elem(tl, r0, t);
        } else {
#pragma unroll
          for (int t = 0; t < TM::NS; ++t) {
            const int idx = lane + t * TM::LU;
            if (idx < nvalid) xo[(unsigned)idx] = *TM::elem(tl, r0, t);
          }
        }
      }
    }
  } else if (A.fused && s == 0 && chain_ok) {
    sweep_epilogue(A, c, qsum);
  }
  OMC_STAMP(15);
  if (MULTI && A.sweep_times && threadIdx.x == 0 && chain_ok) {
    // wave 0 is the one that runs the epilogue: its exit is the end of the chain's sweep (self-restarting workgroups: its
    // next entry follows at once, so consecutive records of a chain tile the launch)
    const unsigned long long t_exit = __builtin_amdgcn_s_memrealtime();
    int64_t r = A.sweep_times_pos + sw;
    if (r >= A.sweep_times_cap) r -= A.sweep_times_cap;
    unsigned long long* const p = A.sweep_times + (r * A.C + c) * 2;
    p[0] = t_enter;
    p[1] = t_exit;
  }
  if (MULTI && SIG == 1 && A.reenter && left > 0) {
    // Restart as the workgroup of the chain's next sweep: same code from its first instruction, with the three
    // registers a fresh workgroup is handed (kernel-argument pointer, workgroup id, work-item id) set to what the
    // dispatcher would have put there for block index + C.  Nothing else is live at a kernel's entry.  What this
    // buys over a fresh workgroup: the x stores of this sweep drain under the next sweep's loads and draws instead of
    // holding the CU until they are acknowledged, and there is no dispatch gap between the sweeps of a chain.
    // (vmcnt is not zero on re-entry -- the waits of the next sweep only become conservative.)
    if (A.reenter != 2) lds_barrier();  // every wave is done with this sweep's LDS image (2: see DESIGN, no barrier)
    const uint64_t kptr = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
    const uint64_t kargs = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kptr) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(kptr >> 32)) << 32);
    const uint32_t next_blk = (uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(0x80000000u | ((uint32_t)(left - 1) << 26) | ((uint32_t)(sw + 1) * (uint32_t)A.C + c_blk)));
    const uint32_t tid = threadIdx.x;
    // (device code may not name a kernel, so the entry point is reached through its linker symbol: a name that does
    // not match an instantiation fails the link, not the run)
#define OMC_REENTER(Mv, MAXTv, SIGv)                                                                                  \
  if constexpr (M == Mv && MAXT == MAXTv && SIG == SIGv)                                                              \
    asm volatile("s_mov_b64 exec, -1\n\ts_getpc_b64 s[4:5]\n\t"                                                      \
                 "s_add_u32 s4, s4, _Z13k_tridiag_segILi" #Mv "ELb1ELi" #MAXTv "ELi" #SIGv "EEv7TriArgsi@rel32@lo+4\n\t" \
                 "s_addc_u32 s5, s5, _Z13k_tridiag_segILi" #Mv "ELb1ELi" #MAXTv "ELi" #SIGv "EEv7TriArgsi@rel32@hi+12\n\t" \
                 "s_setpc_b64 s[4:5]" ::"{s[0:1]}"(kargs), "{s2}"(next_blk), "{v0}"(tid) : "memory", "s4", "s5")
    OMC_REENTER(8, 1024, 1);
    OMC_REENTER(10, 1024, 1);
#undef OMC_REENTER
  }
}

// ------------------------------------------------------------------------------------------
// Self-check of what the restart above takes for granted.  `s_setpc_b64` to the kernel's first instruction reproduces a
// fresh workgroup only if the kernel descriptor asks the dispatcher for exactly the three registers the restart sets:
// two user SGPRs (the kernel-argument pointer, nothing else: no dispatch / queue pointer, no dispatch id, no flat-scratch
// init, no preloaded kernel arguments), workgroup id x as the only system SGPR, the packed work-item id in v0, and no
// private segment.  The descriptors of the re-entered instantiations are read HERE, from the code object the runtime
// actually loaded (their `.kd` linker symbols), and compared on the host before the first restarting launch; a mismatch
// (another compiler, another flag) switches the restarting form off for the process instead of producing wrong chains.
// tests/test_kernel_resources.py checks the same facts at build time without a GPU.
#define OMC_KD_WORDS(Mv, MAXTv, SIGv, dst)                                                                               \
  do {                                                                                                                    \
    uint64_t kd_;                                                                                                         \
    asm volatile("s_getpc_b64 s[4:5]\n\t"                                                                                \
                 "s_add_u32 s4, s4, _Z13k_tridiag_segILi" #Mv "ELb1ELi" #MAXTv "ELi" #SIGv "EEv7TriArgsi.kd@rel32@lo+4\n\t"  \
                 "s_addc_u32 s5, s5, _Z13k_tridiag_segILi" #Mv "ELb1ELi" #MAXTv "ELi" #SIGv "EEv7TriArgsi.kd@rel32@hi+12\n\t" \
                 "s_mov_b64 %0, s[4:5]"                                                                                   \
                 : "=s"(kd_)::"s4", "s5");                                                                               \
    const uint32_t* w_ = (const uint32_t*)kd_;                                                                            \
    (dst)[0] = w_[1];  /* PRIVATE_SEGMENT_FIXED_SIZE */                                                                    \
    (dst)[1] = w_[13]; /* COMPUTE_PGM_RSRC2 */                                                                             \
    (dst)[2] = w_[14]; /* kernel code properties (low half), kernarg preload spec (high half) */                           \
  } while (0)

__global__ void k_reentry_probe(uint32_t* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  OMC_KD_WORDS(8, 1024, 1, out);
  OMC_KD_WORDS(10, 1024, 1, out + 3);
}

// the expectation, on the three descriptor words the probe returns (LLVM AMDGPUUsage, "Kernel Descriptor")
static bool reentry_descriptor_ok(uint32_t private_size, uint32_t rsrc2, uint32_t props_preload) {
  const uint32_t props = props_preload & 0xffffu, preload = props_preload >> 16;
  const bool user_sgprs = ((rsrc2 >> 1) & 0x1fu) == 2u && (props & 0x7fu) == 0x08u;  // kernarg segment pointer only
  const bool system_sgprs = ((rsrc2 >> 7) & 0xfu) == 0x1u;                           // workgroup id x; no y, z, info
  const bool no_private = private_size == 0u && (rsrc2 & 1u) == 0u && (props & (1u << 11)) == 0u;
  const bool wave64 = (props & (1u << 10)) == 0u;
  return user_sgprs && system_sgprs && no_private && wave64 && preload == 0u;
}

static int g_reentry_ok = -1;  // -1: not probed yet (process-wide: one code object)
static bool reentry_abi_ok(omc_ctx* ctx) {
  if (g_reentry_ok >= 0) return g_reentry_ok != 0;
  uint32_t* d = nullptr;
  uint32_t h[6] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
  bool ok = hipMalloc(&d, sizeof(h)) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_reentry_probe, dim3(1), dim3(64), 0, ctx->stream, d);
    ok = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
         hipStreamSynchronize(ctx->stream) == hipSuccess;
    hipFree(d);
  }
  ok = ok && reentry_descriptor_ok(h[0], h[1], h[2]) && reentry_descriptor_ok(h[3], h[4], h[5]);
  g_reentry_ok = ok ? 1 : 0;
  if (!ok) omc_set_error_text("omc_gmrf_run: the kernel descriptor does not match the restart's entry state; self-restarting workgroups are off");
  return ok;
}
int omc_reentry_probe_result(omc_ctx* ctx) { return reentry_abi_ok(ctx) ? 1 : 0; }

// ------------------------------------------------------------------------------------------
// small helpers
__global__ void k_tridiag_matvec(int64_t n, const double* diag, const double* off, const double* v, double* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double t = (diag ? diag[i] : 1.0) * v[i];
    if (off) {
      if (i > 0) t = fma(off[i - 1], v[i - 1], t);
      if (i < n - 1) t = fma(off[i], v[i + 1], t);
    }
    out[i] = t;
  }
}

// out[c][:] (+)= scale[c] * M v_c for per-chain vectors v_c (M tridiagonal from diag / off; NULL diag = identity)
__global__ void __launch_bounds__(256) k_tridiag_matvec_chain(int64_t n, const double* diag, const double* off, const double* v,
                                                              int64_t ld_v, const double* scale, double* out, int64_t ld_o,
                                                              int accumulate) {
  const int64_t c = blockIdx.y;
  const double s = scale ? scale[c] : 1.0;
  const double* vc = v + c * ld_v;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double t = (diag ? diag[i] : 1.0) * vc[i];
    if (off) {
      if (i > 0) t = fma(off[i - 1], vc[i - 1], t);
      if (i < n - 1) t = fma(off[i], vc[i + 1], t);
    }
    out[c * ld_o + i] = accumulate ? fma(s, t, out[c * ld_o + i]) : s * t;
  }
}

// out[c][:] = a x_c + b y_c   (y per chain, or shared with ld_y == 0)
__global__ void __launch_bounds__(256) k_chain_lincomb(int64_t n, double a, const double* x, int64_t ld_x, double b, const double* y,
                                                       int64_t ld_y, double* out, int64_t ld_o) {
  const int64_t c = blockIdx.y;
  // b == 0: y is not read at all (the BLAS convention: 0 * inf or 0 * NaN in y must not reach the result -- a scale of
  // +inf is what the Normal-Gamma update's zero-rate guard produces, sampler.py:285-286)
  if (b == 0.0) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
      out[c * ld_o + i] = a * x[c * ld_x + i];
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[c * ld_o + i] = fma(a, x[c * ld_x + i], b * y[c * ld_y + i]);
}

// one workgroup per chain: quad[k][c] = (x-m_k)' M_k (x-m_k)
// One pass over the chain's row for all NT terms: the row (the only per-chain operand) is read once, a vector a term does
// not have is not read at all (uniform branches on flags set before the loop: an identity term around a shared centre
// runs at the row's bandwidth), and the element indices are 32-bit on uniform base pointers (scalar-base loads, no
// 64-bit address arithmetic per load).  Every term's sum is accumulated in the order it always was.
template <int NT, bool CCV>
__device__ __forceinline__ void quadform_row(const TermsDev& T, const CentreChain& CC, int64_t n, const double* xc, const double* ccv,
                                             double (&acc)[OMC_MAX_TERMS]) {
  bool hd[NT], ho[NT], hc[NT], hcc[NT], any_off = false;
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    hd[k] = T.diag[k] != nullptr; ho[k] = T.off[k] != nullptr && n > 1; hc[k] = T.center[k] != nullptr;
    hcc[k] = CCV && CC.k == k;
    any_off |= ho[k];
  }
  const unsigned nn = (unsigned)n, step = blockDim.x;  // (n < 2^31: checked by the entry point)
  for (unsigned i = threadIdx.x; i < nn; i += step) {
    const bool has_next = i + 1 < nn;
    const unsigned in = has_next ? i + 1 : i;
    const double xi = xc[i];
    double xn = 0.0, ci = 0.0, cn = 0.0;
    if (any_off) xn = xc[in];
    if constexpr (CCV) {
      ci = ccv[i];
      if (any_off) cn = ccv[in];
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const double r = xi - (hc[k] ? T.center[k][i] : 0.0) - (hcc[k] ? ci : 0.0);
      acc[k] = fma((hd[k] ? T.diag[k][i] : 1.0) * r, r, acc[k]);
      if (ho[k] && has_next) {
        const double rn = xn - (hc[k] ? T.center[k][in] : 0.0) - (hcc[k] ? cn : 0.0);
        acc[k] = fma(2.0 * T.off[k][i] * r, rn, acc[k]);
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_tridiag_quadform(TermsDev T, CentreChain CC, int64_t n, int64_t C, const double* x,
                                                         int64_t ld_x, double* quad) {
  __shared__ double red[OMC_MAX_TERMS][4];
  const int64_t c = blockIdx.x;
  const double* xc = x + c * ld_x;
  const double* ccv = CC.v ? CC.v + c * CC.ld : nullptr;
  double acc[OMC_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) acc[k] = 0.0;
  if (ccv) {
    switch (T.n_terms) {
      case 1: quadform_row<1, true>(T, CC, n, xc, ccv, acc); break;
      case 2: quadform_row<2, true>(T, CC, n, xc, ccv, acc); break;
      case 3: quadform_row<3, true>(T, CC, n, xc, ccv, acc); break;
      default: quadform_row<4, true>(T, CC, n, xc, ccv, acc); break;
    }
  } else {
    switch (T.n_terms) {
      case 1: quadform_row<1, false>(T, CC, n, xc, ccv, acc); break;
      case 2: quadform_row<2, false>(T, CC, n, xc, ccv, acc); break;
      case 3: quadform_row<3, false>(T, CC, n, xc, ccv, acc); break;
      default: quadform_row<4, false>(T, CC, n, xc, ccv, acc); break;
    }
  }
#pragma unroll
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    double a = acc[k];
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x < OMC_MAX_TERMS && (int)threadIdx.x < T.n_terms) {
    const int k = threadIdx.x;
    quad[k * C + c] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
  }
}

// log det of one shared tridiagonal matrix of any length: D_i = a_i - b_{i-1}^2 / D_{i-1}, sum log D_i (gmrf.py:489-520)
__global__ void __launch_bounds__(64) k_tridiag_logdet_serial(int64_t n, const double* diag, const double* off, double* logdet,
                                                              long long* bad) {
  if (blockIdx.x != 0) return;
  // One dependent chain -- but the chain is the recurrence alone.  The wave fetches 64 columns at a time (one coalesced
  // load per vector, the next 64 while the current ones are consumed), every lane runs the same recurrence on values
  // handed round by v_readlane (no branch, no load inside the chain), and the logarithm is taken of a running product of
  // mantissas, once per 64 columns.  (One lane with its loads inside the chain: a trip to memory per column, 7.6 ms at
  // n = 20 000 and 22 ms at n = 50 000 -- more than ten sweeps of the model this is the set-up of.)
  const int lane = threadIdx.x;
  auto fetch = [&](int64_t i0, double& a, double& b) {  // padding: a = 1, b = 0 (pivot 1, log 0)
    const int64_t i = i0 + lane;
    a = (diag && i < n) ? diag[i] : 1.0;
    b = (off && i >= 1 && i < n) ? off[i - 1] : 0.0;
  };
  auto bcast = [&](double v, int t) -> double {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), t), __builtin_amdgcn_readlane(__double2loint(v), t));
  };
  double D = 1.0, mant = 1.0;
  long long ex = 0;
  bool neg = false;
  double a_cur, b_cur, a_nxt, b_nxt;
  fetch(0, a_cur, b_cur);
  for (int64_t i0 = 0; i0 < n; i0 += 64) {
    fetch(i0 + 64, a_nxt, b_nxt);
#pragma unroll
    for (int t = 0; t < 64; ++t) {
      const double a = bcast(a_cur, t), b = bcast(b_cur, t);
      D = fma(-b * omc_rcp_nr(D), b, a);  // (the first column: b = 0)
      const bool ok = D > 0.0;
      neg |= !ok;
      const double Dp = ok ? D : 1.0;
      mant *= __builtin_amdgcn_frexp_mant(Dp);
      ex += __builtin_amdgcn_frexp_exp(Dp);
      if ((t & 15) == 15) {
        ex += __builtin_amdgcn_frexp_exp(mant);
        mant = __builtin_amdgcn_frexp_mant(mant);
      }
    }
    a_cur = a_nxt; b_cur = b_nxt;
  }
  if (lane == 0) {
    logdet[0] = neg ? NAN : log(mant) + (double)ex * 0.69314718055994530942;
    if (neg) atomicMin((unsigned long long*)bad, 0ull);
  }
}

// ------------------------------------------------------------------------------------------
// host side
static bool terms_to_dev(const omc_tridiag_terms* t, TermsDev* d, CentreChain* cc = nullptr) {
  if (!t || t->n_terms < 1 || t->n_terms > OMC_MAX_TERMS) return false;
  d->n_terms = t->n_terms;
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    const bool on = k < t->n_terms;
    d->diag[k] = on ? t->diag[k] : nullptr;
    d->off[k] = on ? t->off[k] : nullptr;
    d->rhs[k] = on ? t->rhs[k] : nullptr;
    d->center[k] = on ? t->center[k] : nullptr;
    d->scale[k] = on ? t->scale[k] : nullptr;
  }
  if (cc) {  // at most one term with a per-chain centre per call
    *cc = CentreChain{};
    for (int k = 0; k < t->n_terms; ++k)
      if (t->center_chain[k]) {
        cc->k = cc->v ? -1 : k;  // (-1: more than one -- the entry points answer OMC_UNSUPPORTED)
        cc->v = t->center_chain[k];
      }
    cc->ld = t->ld_center_chain;
  }
  return true;
}
static bool has_center_chain(const CentreChain& cc) { return cc.v != nullptr; }

static void args_defaults(omc_ctx* ctx, TriArgs* A, int64_t n) {
  A->T = TermsDev{};  // every pointer null, no terms: a caller that fills the terms by hand cannot leave a field behind
  A->cc = CentreChain{};
  A->n = n; A->C = ctx->n_chains; A->chain_offset = ctx->chain_offset;
  A->rhs_chain = nullptr; A->ld_rhs = 0;
  A->z = nullptr; A->ld_z = 0; A->zero_z = 0;
  A->key = omc_make_key(ctx->seed, 0, OMC_RNG_NORMAL);
  A->x = nullptr; A->ld_x = 0; A->quad = nullptr; A->logdet = nullptr;
  A->bad = ctx->d_bad_chain;
  A->fallbacks = ctx->d_fallbacks;
  A->newton_max = ctx->tridiag_newton_max;
  A->perturb_start = ctx->tridiag_perturb_ppb * 1e-9;
  A->work = nullptr;
  A->fused = 0;
  A->stamps = ctx->stamps;
  A->sweep_times = nullptr; A->sweep_times_cap = 0; A->sweep_times_pos = 0;  // (omc_gmrf_run switches the sweep clock on)
  A->log_post = nullptr;
  A->gb_dev = nullptr; A->gdraw_dev = nullptr;
  A->n_sweeps = 0; A->reenter = 0; A->block_sweeps = 0; A->early_draws = 0; A->epoch = 0; A->seed = ctx->seed; A->handoff = nullptr; A->timeouts = ctx->d_fallbacks + 1;
  for (int k = 0; k < OMC_MAX_TERMS; ++k) A->gdraw[k] = 0;
  for (int i = 0; i < OMC_RUN_MAX; ++i) { A->rec[i].draw = 0; A->rec[i].x = nullptr; A->rec[i].log_post = nullptr; A->rec[i].slot_off = -1; }
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    A->gb[k].enabled = 0; A->gb[k].a0 = A->gb[k].b0 = A->gb[k].half_npos = A->gb[k].lnorm = 0.0;
    A->gb[k].g_inject = nullptr; A->gb[k].store = nullptr; A->gb[k].scale_out = nullptr;
    A->gb[k].logdet_unscaled = nullptr; A->gb[k].key = A->key;
  }
}

// CLOCK_MONOTONIC in seconds: the clock of Python's time.perf_counter on Linux, so a caller can place the launch log on
// its own time axis
static double omc_host_clock() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// nodes-per-lane variants: M -> (largest n, launch bound of the one-chain-per-workgroup form)
template <int M> struct SegCfg;
// SMOOTHER: 1 where the structure-specialised instantiation (SIG 1) exists -- the variants auto_seg picks
template <> struct SegCfg<8>  { static constexpr int MAXT = 1024; static constexpr int SMOOTHER = 1; };
template <> struct SegCfg<10> { static constexpr int MAXT = 1024; static constexpr int SMOOTHER = 1; };
template <> struct SegCfg<16> { static constexpr int MAXT = 640; static constexpr int SMOOTHER = 0; };  // measured at cfg3 with SIG 1:
template <> struct SegCfg<20> { static constexpr int MAXT = 512; static constexpr int SMOOTHER = 0; };  // 125 and 112 us against 102 for M = 10
template <> struct SegCfg<32> { static constexpr int MAXT = 512; static constexpr int SMOOTHER = 0; };
static int64_t seg_max_n(int seg) {
  switch (seg) {
    case 8: return 8 * 1024;
    case 10: return 10 * 1024;
    case 16: return 16 * 640;
    case 20: return 20 * 512;
    case 32: return 32 * 512;
  }
  return 0;
}

template <int M>
static bool launch_seg(omc_ctx* ctx, const TriArgs& A_in) {
  TriArgs A = A_in;
  const int S = (int)((A.n + M - 1) / M);
  if (S <= 64) {
    const int G = pow2_ceil(S);
    const int64_t chains_per_block = 4 * (64 / G);
    const int64_t grid = (A.C + chains_per_block - 1) / chains_per_block;
    hipLaunchKernelGGL((k_tridiag_seg<M, false, 256>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, G);
  } else {
    const int threads = 64 * ((S + 63) / 64);
    // the specialised instantiation owns the CU (its LDS image is sized for a full workgroup); a short chain
    // leaves room for a second workgroup of the generic one, which then wins (n = 2000: 38 against 53 us)
    const bool special = SegCfg<M>::SMOOTHER && !ctx->tridiag_generic && is_smoother(A.T) && !has_center_chain(A.cc) && A.n >= 2 &&
                         2 * threads > SegCfg<M>::MAXT;
    // self-restarting workgroups: the specialised instantiation only (it keeps no private memory, so the three entry
    // registers are all a restart has to reproduce)
    if (!special) A.reenter = 0;
    if (!special && A.fused) {
      // the Normal-Gamma blocks in device memory for the generic instantiation (stream-ordered copy: the previous launch
      // has read its image by the time this one is written)
      const size_t gb_bytes = sizeof(A.gb), gd_bytes = sizeof(A.gdraw);
      if (!ctx->d_gamma_tab && hipMalloc(&ctx->d_gamma_tab, gb_bytes + gd_bytes) != hipSuccess) return false;
      if (hipMemcpyAsync(ctx->d_gamma_tab, A.gb, gb_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return false;
      if (hipMemcpyAsync((char*)ctx->d_gamma_tab + gb_bytes, A.gdraw, gd_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return false;
      A.gb_dev = (const GammaDev*)ctx->d_gamma_tab;
      A.gdraw_dev = (const unsigned long long*)((char*)ctx->d_gamma_tab + gb_bytes);
    }
    // workgroup-per-chain form: one workgroup per (sweep, chain), or per chain when the workgroups restart themselves
    const int64_t wg_per_chain = A.n_sweeps <= 0 ? 1 : (!A.reenter ? A.n_sweeps : (A.block_sweeps > 0 ? (A.n_sweeps + A.block_sweeps - 1) / A.block_sweeps : 1));
    const unsigned wg_grid = (unsigned)(A.C * wg_per_chain);
    // the shifted smoother (SIG 3): single-sweep launches without per-chain offsets (not the fused sweep: its epilogue's
    // log-posterior would need the shared centre's terms spelled out for e-coordinates)
    const bool shifted = SegCfg<M>::SMOOTHER && !ctx->tridiag_generic && is_shifted_smoother(A.T, A.cc) && !A.rhs_chain && !A.fused &&
                         A.n_sweeps <= 0 && A.n >= 2 && 2 * threads > SegCfg<M>::MAXT;
    if (shifted)
      hipLaunchKernelGGL((k_tridiag_seg<M, true, SegCfg<M>::MAXT, 3 * SegCfg<M>::SMOOTHER>), dim3(wg_grid), dim3(threads), 0,
                         ctx->stream, A, threads);
    else if (special && SegCfg<M>::SMOOTHER && A.early_draws && !A.reenter && !A.z && !A.zero_z)  // the waiting form (see the kernel)
      hipLaunchKernelGGL((k_tridiag_seg<M, true, SegCfg<M>::MAXT, 2 * SegCfg<M>::SMOOTHER>), dim3(wg_grid), dim3(threads), 0,
                         ctx->stream, A, threads);
    else if (special)
      hipLaunchKernelGGL((k_tridiag_seg<M, true, SegCfg<M>::MAXT, SegCfg<M>::SMOOTHER>), dim3(wg_grid), dim3(threads), 0,
                         ctx->stream, A, threads);
    else
      hipLaunchKernelGGL((k_tridiag_seg<M, true, SegCfg<M>::MAXT>), dim3(wg_grid), dim3(threads), 0, ctx->stream,
                         A, threads);
  }
  return true;
}

static bool launch_seg_any(omc_ctx* ctx, const TriArgs& A, int seg) {
  switch (seg) {
    case 8: return launch_seg<8>(ctx, A);
    case 10: return launch_seg<10>(ctx, A);
    case 16: return launch_seg<16>(ctx, A);
    case 20: return launch_seg<20>(ctx, A);
    case 32: return launch_seg<32>(ctx, A);
  }
  return false;
}

static int auto_seg(int64_t n) {
  // fewest nodes per lane that still fits one workgroup (measured on cfg3: 10 beats 16 and 20)
  if (n <= seg_max_n(8)) return 8;
  if (n <= seg_max_n(10)) return 10;
  return 32;
}

// true if launch_tridiag would run the workgroup-per-chain form of the segmented kernel for n nodes (the only form
// that takes several sweeps per launch)
static bool takes_wg_per_chain(const omc_ctx* ctx, int64_t n) {
  if (ctx->tridiag_algo == 1) return false;
  const int seg = ctx->tridiag_seg ? ctx->tridiag_seg : auto_seg(n);
  if (n > seg_max_n(seg)) return false;
  return (n + seg - 1) / seg > 64;
}

// true if launch_tridiag would pick the structure-specialised instantiation (the one that takes several sweeps per launch)
static bool takes_specialised(const omc_ctx* ctx, const TermsDev& T, int64_t n) {
  if (!takes_wg_per_chain(ctx, n) || ctx->tridiag_generic || !is_smoother(T) || n < 2) return false;
  const int seg = ctx->tridiag_seg ? ctx->tridiag_seg : auto_seg(n);
  const int threads = 64 * (int)(((n + seg - 1) / seg + 63) / 64);
  switch (seg) {
    case 8: return SegCfg<8>::SMOOTHER && 2 * threads > SegCfg<8>::MAXT;
    case 10: return SegCfg<10>::SMOOTHER && 2 * threads > SegCfg<10>::MAXT;
    case 16: return SegCfg<16>::SMOOTHER && 2 * threads > SegCfg<16>::MAXT;
    case 20: return SegCfg<20>::SMOOTHER && 2 * threads > SegCfg<20>::MAXT;
    case 32: return SegCfg<32>::SMOOTHER && 2 * threads > SegCfg<32>::MAXT;
  }
  return false;
}

static omc_status launch_tridiag(omc_ctx* ctx, TriArgs& A) {
  int algo = ctx->tridiag_algo;
  int seg = ctx->tridiag_seg;
  const int64_t n = A.n;
  if (algo == 0) algo = (n <= seg_max_n(32)) ? 2 : 1;
  if (algo == 2) {
    if (seg == 0) seg = auto_seg(n);
    if (n > seg_max_n(seg)) {
      if (ctx->tridiag_algo == 2) return OMC_UNSUPPORTED;
      algo = 1;
    }
  }
  if (algo == 2) {
    if (!launch_seg_any(ctx, A, seg)) return OMC_INVALID_ARG;
  } else {
    if (!A.x) return OMC_INVALID_ARG;
    const size_t need = (size_t)A.C * (size_t)n * sizeof(double);
    if (ctx->workspace_bytes < need) {
      OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (ctx->workspace) OMC_HIP_CHECK(hipFree(ctx->workspace));
      ctx->workspace = nullptr;
      ctx->workspace_bytes = 0;
      OMC_HIP_CHECK(hipMalloc(&ctx->workspace, need));
      ctx->workspace_bytes = need;
    }
    A.work = ctx->workspace;
    const int64_t grid = (A.C + 63) / 64;
    hipLaunchKernelGGL(k_tridiag_serial, dim3((unsigned)grid), dim3(64), 0, ctx->stream, A);
  }
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

// ------------------------------------------------------------------------------------------
// Chains too long for one workgroup (n > 16 384).  The one-lane-per-chain kernel is a cliff there (28 ms per sweep at
// n = 20 000 x 1024 chains); the segmented kernels of the band route (omc_band.hip, bandwidth 1) take any n at 2-5 ms.
// The tridiagonal terms are put into that route's band storage (a stream-ordered device copy of 2 n doubles per term into
// a context buffer), the draw is omc_band_sample_canonical -- same natural-order factor, same draw streams, equal to the
// tridiagonal kernels' result to rounding -- and what the fused sweep adds (quadratic forms, Normal-Gamma updates, log
// posterior) follows as the library's own separate launches.
__global__ void __launch_bounds__(256) k_band_from_tridiag(int64_t n, const double* diag, const double* off, double* band) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    band[i] = diag ? diag[i] : 1.0;
    band[n + i] = (off && i < n - 1) ? off[i] : 0.0;
  }
}

static bool long_chain_route(const omc_ctx* ctx, int64_t n) { return ctx->tridiag_algo == 0 && n > seg_max_n(32); }

static omc_status long_chain_draw(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* t, const double* rhs_chain, int64_t ld_rhs,
                                  const double* z, int64_t ld_z, uint64_t draw_index, double* x, int64_t ld_x, double* mean,
                                  int64_t ld_mean, double* logdet) {
  omc_band_terms bt;
  bt.n_terms = t->n_terms;
  const size_t per = 2 * (size_t)n * sizeof(double);
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  {
    const omc_status st = omc_ensure_bytes(ctx, (void**)&ctx->long_band, &ctx->long_band_bytes, per * OMC_MAX_TERMS);
    if (st != OMC_OK) return st;
  }
  for (int k = 0; k < OMC_MAX_TERMS; ++k) {
    bt.band[k] = nullptr; bt.bw[k] = 0; bt.rhs[k] = nullptr; bt.scale[k] = nullptr;
    if (k >= t->n_terms) continue;
    bt.rhs[k] = t->rhs[k];
    bt.scale[k] = t->scale[k];
    if (t->diag[k] || t->off[k]) {
      double* b = ctx->long_band + (size_t)k * 2 * (size_t)n;
      int64_t grid = (n + 255) / 256;
      if (grid > 1024) grid = 1024;
      hipLaunchKernelGGL(k_band_from_tridiag, dim3((unsigned)grid), dim3(256), 0, ctx->stream, n, t->diag[k], t->off[k], b);
      bt.band[k] = b;
      bt.bw[k] = t->off[k] ? 1 : 0;
    }
  }
  OMC_HIP_CHECK(hipGetLastError());
  return omc_band_sample_canonical(ctx, n, 1, &bt, rhs_chain, ld_rhs, z, ld_z, draw_index, x, ld_x, mean, ld_mean, logdet);
}

extern "C" {

int32_t omc_reentry_descriptor_ok(uint32_t private_segment_fixed_size, uint32_t compute_pgm_rsrc2, uint32_t properties_and_preload) {
  return reentry_descriptor_ok(private_segment_fixed_size, compute_pgm_rsrc2, properties_and_preload) ? 1 : 0;
}

omc_status omc_tridiag_sample_canonical(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms,
                                        const double* rhs_chain, int64_t ld_rhs, const double* z_inject,
                                        int64_t ld_z, uint64_t draw_index, double* x_out, int64_t ld_x,
                                        double* mean_out, int64_t ld_mean, double* quad_out, double* logdet_out) {
  if (!ctx || n < 1 || !x_out || ld_x < n) return OMC_INVALID_ARG;
  if ((rhs_chain && ld_rhs < n) || (z_inject && ld_z < n) || (mean_out && ld_mean < n)) return OMC_INVALID_ARG;
  TriArgs A{};
  args_defaults(ctx, &A, n);
  if (!terms_to_dev(terms, &A.T, &A.cc)) return OMC_INVALID_ARG;
  if (has_center_chain(A.cc) && (A.cc.k < 0 || !takes_wg_per_chain(ctx, n) || A.cc.ld < n)) return OMC_UNSUPPORTED;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  if (long_chain_route(ctx, n)) {
    omc_status st = long_chain_draw(ctx, n, terms, rhs_chain, ld_rhs, z_inject, ld_z, draw_index, x_out, ld_x, mean_out, ld_mean,
                                    logdet_out);
    if (st != OMC_OK || !quad_out) return st;
    return omc_tridiag_quadform(ctx, n, terms, x_out, ld_x, quad_out);
  }
  A.rhs_chain = rhs_chain; A.ld_rhs = ld_rhs;
  A.key = omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL);
  if (mean_out) {  // mu = Q^{-1} b is the same solve with z = 0 (gmrf.py:196)
    A.zero_z = 1;
    A.x = mean_out; A.ld_x = ld_mean;
    omc_status st = launch_tridiag(ctx, A);
    if (st != OMC_OK) return st;
  }
  A.z = z_inject; A.ld_z = ld_z; A.zero_z = 0;
  A.x = x_out; A.ld_x = ld_x; A.quad = quad_out; A.logdet = logdet_out;
  A.cc.quad_skip = quad_out ? ctx->tridiag_quad_skip : 0;  // (honoured by the generic workgroup-per-chain instantiation only)
  return launch_tridiag(ctx, A);
}

omc_status omc_gmrf_sweep(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms, const omc_gamma_block* blocks,
                          const double* rhs_chain, int64_t ld_rhs, const double* z_inject, int64_t ld_z,
                          uint64_t draw_index, double* x_out, int64_t ld_x, double* log_post_out) {
  if (!ctx || n < 1 || !x_out || ld_x < n || !blocks) return OMC_INVALID_ARG;
  if ((rhs_chain && ld_rhs < n) || (z_inject && ld_z < n)) return OMC_INVALID_ARG;
  TriArgs A{};
  args_defaults(ctx, &A, n);
  if (!terms_to_dev(terms, &A.T, &A.cc)) return OMC_INVALID_ARG;
  if (has_center_chain(A.cc) && (A.cc.k < 0 || !takes_wg_per_chain(ctx, n) || A.cc.ld < n)) return OMC_UNSUPPORTED;
  for (int k = 0; k < A.T.n_terms; ++k) {
    const omc_gamma_block& b = blocks[k];
    GammaDev& g = A.gb[k];
    g.enabled = b.enabled ? 1 : 0;
    if (g.enabled) {
      if (!terms->scale[k] || b.n_pos < 0 || !(b.a0 + 0.5 * (double)b.n_pos > 0.0)) return OMC_INVALID_ARG;
      if (log_post_out && !(b.a0 > 0.0 && b.b0 > 0.0)) return OMC_INVALID_ARG;
    }
    if (log_post_out && !b.logdet_unscaled) return OMC_INVALID_ARG;
    g.a0 = b.a0; g.b0 = b.b0; g.half_npos = 0.5 * (double)b.n_pos;
    g.lnorm = (g.enabled && log_post_out) ? b.a0 * log(b.b0) - lgamma(b.a0) : 0.0;
    g.g_inject = b.g_inject; g.store = b.store;
    g.scale_out = const_cast<double*>(terms->scale[k]);
    g.logdet_unscaled = b.logdet_unscaled;
    g.key = omc_make_key(ctx->seed, b.draw_index, OMC_RNG_GAMMA);
  }
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  if (long_chain_route(ctx, n)) {
    // the sweep as separate launches: draw, quadratic forms, Normal-Gamma updates, log posterior (model.py:57-70)
    const int nt = terms->n_terms;
    const int64_t C = ctx->n_chains;
    {
      const omc_status st0 = omc_ensure_bytes(ctx, (void**)&ctx->long_quad, &ctx->long_quad_bytes, (size_t)OMC_MAX_TERMS * C * sizeof(double));
      if (st0 != OMC_OK) return st0;
    }
    omc_status st = long_chain_draw(ctx, n, terms, rhs_chain, ld_rhs, z_inject, ld_z, draw_index, x_out, ld_x, nullptr, 0, nullptr);
    if (st != OMC_OK) return st;
    st = omc_tridiag_quadform(ctx, n, terms, x_out, ld_x, ctx->long_quad);
    if (st != OMC_OK) return st;
    for (int k = 0; k < nt; ++k) {
      const omc_gamma_block& b = blocks[k];
      if (!b.enabled) continue;
      double* const sc = const_cast<double*>(terms->scale[k]);
      st = omc_normal_gamma_update(ctx, b.a0, b.b0, b.n_pos, ctx->long_quad + k * C, b.g_inject, b.draw_index, sc);
      if (st != OMC_OK) return st;
      if (b.store) {
        st = omc_chain_copy(ctx, 1, sc, 1, b.store, 1);
        if (st != OMC_OK) return st;
      }
    }
    if (log_post_out) {
      int first = 1;
      for (int k = 0; k < nt; ++k) {
        const omc_gamma_block& b = blocks[k];
        st = omc_scaled_gauss_logpdf(ctx, n, terms->scale[k], b.logdet_unscaled, ctx->long_quad + k * C, log_post_out, first ? 0 : 1);
        if (st != OMC_OK) return st;
        first = 0;
        if (b.enabled) {
          st = omc_gamma_logpdf(ctx, terms->scale[k], b.a0, b.b0, log_post_out, 1);
          if (st != OMC_OK) return st;
        }
      }
    }
    return OMC_OK;
  }
  A.rhs_chain = rhs_chain; A.ld_rhs = ld_rhs;
  A.z = z_inject; A.ld_z = ld_z;
  A.key = omc_make_key(ctx->seed, draw_index, OMC_RNG_NORMAL);
  A.x = x_out; A.ld_x = ld_x;
  A.zero_z = ctx->debug_zero_z;
  A.fused = 1;
  A.log_post = log_post_out;
  return launch_tridiag(ctx, A);
}

omc_status omc_gmrf_run(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms, const omc_gamma_block* blocks,
                        int64_t n_burn, int64_t n_iter, int64_t n_thin, uint64_t draw_index0, uint64_t draws_per_sweep,
                        double* x_store, int64_t ld_x, int64_t x_slot_stride, int64_t first_slot, int64_t n_slots,
                        double* log_post_store, double* scratch_x) {
  if (!ctx || !terms || !blocks || n_burn < 0 || n_iter < 0 || n_thin < 1 || n_slots < 1 || first_slot < 0 || !x_store ||
      !scratch_x || x_slot_stride < ctx->n_chains * ld_x)
    return OMC_INVALID_ARG;
  const int64_t C = ctx->n_chains;
  omc_gamma_block b[OMC_MAX_TERMS];
  // mcmc.py:97-98: every iteration, burn-in included, is n_thin sweeps
  const int64_t burn = n_burn * n_thin, total = burn + n_iter * n_thin;
  if (n < 1 || ld_x < n) return OMC_INVALID_ARG;
  if (ctx->run_sweeps_per_launch > 1 && takes_wg_per_chain(ctx, n) && C * (int64_t)OMC_RUN_MAX < (int64_t)1 << 31) {
    // Several sweeps per launch: workgroup (sweep, chain) = block sweep * C + chain.  A chain's sweep s+1 needs only the
    // scales its sweep s drew; they travel through the chain's hand-over line (see OMC_HANDOFF_WORDS).  Blocks are
    // dispatched in index order, so a producer is always on the chip before its consumer; the consumer's wait is
    // bounded anyway and a hand-over that never came is reported by omc_ctx_status.  What this buys: the ramp, tail
    // and boundary of a launch (7.4 us against 21 us per round of workgroups) are paid once per OMC_RUN_MAX sweeps.
    TriArgs A{};
    args_defaults(ctx, &A, n);
    if (!terms_to_dev(terms, &A.T, &A.cc)) return OMC_INVALID_ARG;
  if (has_center_chain(A.cc) && (A.cc.k < 0 || !takes_wg_per_chain(ctx, n) || A.cc.ld < n)) return OMC_UNSUPPORTED;
    for (int k = 0; k < A.T.n_terms; ++k) {
      const omc_gamma_block& bk = blocks[k];
      GammaDev& g = A.gb[k];
      g.enabled = bk.enabled ? 1 : 0;
      if (g.enabled) {
        if (!terms->scale[k] || bk.n_pos < 0 || !(bk.a0 + 0.5 * (double)bk.n_pos > 0.0)) return OMC_INVALID_ARG;
        if (log_post_store && !(bk.a0 > 0.0 && bk.b0 > 0.0)) return OMC_INVALID_ARG;
        if (bk.g_inject) return OMC_INVALID_ARG;
      }
      if (log_post_store && !bk.logdet_unscaled) return OMC_INVALID_ARG;
      g.a0 = bk.a0; g.b0 = bk.b0; g.half_npos = 0.5 * (double)bk.n_pos;
      g.lnorm = (g.enabled && log_post_store) ? bk.a0 * log(bk.b0) - lgamma(bk.a0) : 0.0;
      g.g_inject = nullptr; g.store = bk.store;
      g.scale_out = const_cast<double*>(terms->scale[k]);
      g.logdet_unscaled = bk.logdet_unscaled;
      A.gdraw[k] = bk.draw_index;
    }
    OMC_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->d_handoff) {
      const size_t bytes = (size_t)C * OMC_HANDOFF_WORDS * sizeof(unsigned long long);
      OMC_HIP_CHECK(hipMalloc(&ctx->d_handoff, bytes));
      OMC_HIP_CHECK(hipMemsetAsync(ctx->d_handoff, 0, bytes, ctx->stream));
      ctx->run_epoch = 1;
    }
    A.handoff = ctx->d_handoff;
    A.ld_x = ld_x;
    A.zero_z = ctx->debug_zero_z;
    A.fused = 1;
    // (the generic instantiation: one sweep per launch, see the kernel)
    const int per = ctx->run_sweeps_per_launch < OMC_RUN_MAX ? ctx->run_sweeps_per_launch : OMC_RUN_MAX;
    // Which form of the launch: one self-restarting workgroup per chain (a chain's sweeps stay on one CU: no dispatch
    // gaps, restart under the epilogue) pays when the chains fill the CUs in whole rounds; otherwise one workgroup per
    // (sweep, chain) -- the dispatcher then balances the CUs sweep by sweep, and with fewer chains than CUs the next
    // sweep of a chain starts on an idle CU under the tail of the previous one (384 chains: 31.4 against 39.2 us per
    // sweep, 128 chains: 19.0 against 19.9; 256 and 1024 chains: the restarting form by 6 % and 3 %).
    static int cached_cus[64];  // (per device: the attribute query is a driver call on every omc_gmrf_run otherwise)
    int dev_cus = (ctx->device >= 0 && ctx->device < 64) ? cached_cus[ctx->device] : 0;
    if (dev_cus <= 0) {
      dev_cus = 256;
      hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
      if (ctx->device >= 0 && ctx->device < 64) cached_cus[ctx->device] = dev_cus;
    }
    const int64_t rounds = (C + dev_cus - 1) / dev_cus;
    const bool whole_rounds = C >= dev_cus && (double)C >= 0.95 * (double)(rounds * dev_cus);
    A.reenter = (whole_rounds || ctx->run_reenter_force) ? ctx->run_reenter : 0;
    if (!takes_specialised(ctx, A.T, n) || has_center_chain(A.cc)) A.reenter = 0;  // the generic instantiation: one workgroup per (sweep, chain)
    if (A.reenter && per > 1 && !reentry_abi_ok(ctx)) A.reenter = 0;  // (see k_reentry_probe)
    if (A.reenter && C * (int64_t)per >= ((int64_t)1 << 26)) A.reenter = 0;  // (a restart's register carries sweep * C + chain in 26 bits)
    A.early_draws = (A.reenter == 0 && 2 * C <= dev_cus && !has_center_chain(A.cc)) ? 1 : 0;  // a waiting workgroup per chain has a CU to itself
    // (sweep, chain) grid: two sweeps of one launch must not write the same store slot -- their workgroups are not ordered
    // against each other (the self-restarting form walks a chain's sweeps in order and may lap the ring)
    const int64_t stored_per_launch_max = (A.reenter == 0 && n_slots < per) ? n_slots : per;
    ctx->launch_log_n = 0; ctx->launch_log_total = 0;
    const bool clock_on = ctx->sweep_times && ctx->sweep_times_cap >= OMC_SWEEP_RING_MIN;
    if (ctx->run_ev_begin) OMC_HIP_CHECK(hipEventRecord(ctx->run_ev_begin, ctx->stream));
    for (int64_t t0 = 0; t0 < total;) {
      int k_sw = (int)(total - t0 < per ? total - t0 : per);
      if (stored_per_launch_max < per) {  // end the launch before a store slot would repeat inside it
        int64_t stored_seen = 0;
        for (int i = 0; i < k_sw; ++i) {
          const int64_t t = t0 + i;
          if (t >= burn && ((t - burn + 1) % n_thin == 0) && ++stored_seen > stored_per_launch_max) { k_sw = i; break; }
        }
      }
      for (int i = 0; i < k_sw; ++i) {
        const int64_t t = t0 + i;
        const bool stored = t >= burn && ((t - burn + 1) % n_thin == 0);
        const int64_t it = stored ? (t - burn + 1) / n_thin - 1 : 0;
        const int64_t slot = (first_slot + it) % n_slots;
        A.rec[i].draw = draw_index0 + (uint64_t)t * draws_per_sweep;
        // a draw that is neither stored nor the run's last is not written at all: the next sweep redraws x from its full
        // conditional without reading it (80 KB per chain and sweep of stores that burn-in and thinning would throw away)
        A.rec[i].x = stored ? x_store + slot * x_slot_stride : (t == total - 1 ? scratch_x : nullptr);
        A.rec[i].log_post = (stored && log_post_store) ? log_post_store + slot * C : nullptr;
        A.rec[i].slot_off = stored ? slot * C : -1;
      }
      A.n_sweeps = k_sw;
      // Self-restarting workgroups in blocks: a workgroup walks `block_sweeps` sweeps of its chain, then a fresh one (block
      // index + C, dispatched when a CU falls free) takes the chain over through the global hand-over line.  One block
      // per launch ties a chain to one CU for the whole launch and the launch ends with its slowest CU (four rounds of
      // 20 sweeps: a tail of ~60 us measured by the sweep clock); shorter blocks let the dispatcher level the CUs.
      A.block_sweeps = k_sw;
      if (A.reenter && ctx->run_block_sweeps > 0 && ctx->run_block_sweeps < k_sw) A.block_sweeps = ctx->run_block_sweeps;
      A.epoch = ctx->run_epoch;
      ctx->run_epoch += (uint32_t)k_sw;
      // the non-specialised paths still read these
      A.key = omc_make_key(ctx->seed, A.rec[0].draw, OMC_RNG_NORMAL);
      A.x = A.rec[0].x;
      A.log_post = A.rec[0].log_post;
      if (clock_on) {
        A.sweep_times = ctx->sweep_times; A.sweep_times_cap = ctx->sweep_times_cap; A.sweep_times_pos = ctx->sweep_times_pos;
      }
      const double t_begin = omc_host_clock();
      omc_status st = launch_tridiag(ctx, A);
      if (st != OMC_OK) return st;
      if (ctx->launch_log_n < OMC_LAUNCH_LOG_MAX) {
        omc_ctx::LaunchRec& r = ctx->launch_log[ctx->launch_log_n++];
        r.t_begin = t_begin; r.t_end = omc_host_clock(); r.n_sweeps = k_sw; r.form = A.reenter;
        r.ring_pos = clock_on ? ctx->sweep_times_pos : -1;
      }
      ctx->launch_log_total++;
      if (clock_on) ctx->sweep_times_pos = (ctx->sweep_times_pos + k_sw) % ctx->sweep_times_cap;
      t0 += k_sw;
    }
    if (ctx->run_ev_end) OMC_HIP_CHECK(hipEventRecord(ctx->run_ev_end, ctx->stream));
    return OMC_OK;
  }
  ctx->launch_log_n = 0; ctx->launch_log_total = 0;
  for (int64_t t = 0; t < total; ++t) {
    const bool stored = t >= burn && ((t - burn + 1) % n_thin == 0);
    const int64_t i = stored ? (t - burn + 1) / n_thin - 1 : 0;
    const int64_t slot = (first_slot + i) % n_slots;
    const uint64_t base = draw_index0 + (uint64_t)t * draws_per_sweep;
    for (int k = 0; k < terms->n_terms && k < OMC_MAX_TERMS; ++k) {
      b[k] = blocks[k];
      b[k].draw_index = base + blocks[k].draw_index;
      b[k].store = (stored && blocks[k].store) ? blocks[k].store + slot * C : nullptr;
    }
    const double t_begin = omc_host_clock();
    omc_status st = omc_gmrf_sweep(ctx, n, terms, b, nullptr, 0, nullptr, 0, base,
                                   stored ? x_store + slot * x_slot_stride : scratch_x, ld_x,
                                   (stored && log_post_store) ? log_post_store + slot * C : nullptr);
    if (st != OMC_OK) return st;
    if (ctx->launch_log_n < OMC_LAUNCH_LOG_MAX) {
      omc_ctx::LaunchRec& r = ctx->launch_log[ctx->launch_log_n++];
      r.t_begin = t_begin; r.t_end = omc_host_clock(); r.n_sweeps = 1; r.form = 0; r.ring_pos = -1;
    }
    ctx->launch_log_total++;
  }
  return OMC_OK;
}

int32_t omc_tridiag_takes_center_chain(omc_ctx* ctx, int64_t n) { return (ctx && n >= 1 && takes_wg_per_chain(ctx, n)) ? 1 : 0; }

omc_status omc_tridiag_quadform(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms, const double* x,
                                int64_t ld_x, double* quad_out) {
  if (!ctx || n < 1 || !x || ld_x < n || !quad_out) return OMC_INVALID_ARG;
  if (n >= (int64_t)1 << 31) return OMC_UNSUPPORTED;  // (32-bit element indices in the kernel)
  TermsDev T{};
  CentreChain CC{};
  if (!terms_to_dev(terms, &T, &CC)) return OMC_INVALID_ARG;
  if (has_center_chain(CC) && CC.k < 0) return OMC_UNSUPPORTED;
  if (has_center_chain(CC) && CC.ld < n) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_tridiag_quadform, dim3((unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, T, CC, n,
                     ctx->n_chains, x, ld_x, quad_out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_tridiag_matvec(omc_ctx* ctx, int64_t n, const double* diag, const double* off, const double* v,
                              double* out) {
  if (!ctx || n < 1 || !v || !out || v == out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  const int64_t grid = (n + 255) / 256;
  hipLaunchKernelGGL(k_tridiag_matvec, dim3((unsigned)(grid > 2048 ? 2048 : grid)), dim3(256), 0, ctx->stream, n,
                     diag, off, v, out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_tridiag_matvec_chain(omc_ctx* ctx, int64_t n, const double* diag, const double* off, const double* v, int64_t ld_v,
                                    const double* scale, double* out, int64_t ld_out, int32_t accumulate) {
  if (!ctx || n < 1 || !v || !out || ld_v < n || ld_out < n || v == out) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  int64_t gx_ = (n + 255) / 256;
  if (gx_ > 64) gx_ = 64;
  hipLaunchKernelGGL(k_tridiag_matvec_chain, dim3((unsigned)gx_, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, n, diag, off, v, ld_v,
                     scale, out, ld_out, (int)accumulate);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_chain_lincomb(omc_ctx* ctx, int64_t n, double a, const double* x, int64_t ld_x, double b, const double* y, int64_t ld_y,
                             double* out, int64_t ld_out) {
  if (!ctx || n < 1 || !x || !y || !out || ld_x < n || ld_out < n || (ld_y != 0 && ld_y < n)) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  int64_t gx_ = (n + 255) / 256;
  if (gx_ > 64) gx_ = 64;
  hipLaunchKernelGGL(k_chain_lincomb, dim3((unsigned)gx_, (unsigned)ctx->n_chains), dim3(256), 0, ctx->stream, n, a, x, ld_x, b, y, ld_y,
                     out, ld_out);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_tridiag_logdet(omc_ctx* ctx, int64_t n, const double* diag, const double* off, double* logdet) {
  if (!ctx || n < 1 || !logdet) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  if (n > seg_max_n(32)) {  // beyond one workgroup: the plain recurrence on one lane (set-up time only, once per model)
    hipLaunchKernelGGL(k_tridiag_logdet_serial, dim3(1), dim3(64), 0, ctx->stream, n, diag, off, logdet, ctx->d_bad_chain);
    OMC_HIP_CHECK(hipGetLastError());
    return OMC_OK;
  }
  TriArgs A{};
  args_defaults(ctx, &A, n);
  A.T.n_terms = 1;  // (args_defaults has cleared every pointer of the terms)
  A.T.diag[0] = diag;
  A.T.off[0] = off;
  A.C = 1; A.chain_offset = 0;
  A.zero_z = 1;
  A.logdet = logdet;
  // the segmented kernel needs no x storage; use it regardless of the algo option
  launch_seg_any(ctx, A, auto_seg(n));
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

}  // extern "C"
