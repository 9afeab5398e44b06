// Shared internals of libomcmc_hip.so (gfx950 only).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "omcmc_hip.h"

#define OMC_LAUNCH_LOG_MAX 64
#define OMC_SWEEP_RING_MIN 64  // >= 2 * OMC_RUN_MAX: a launch never laps its own records
struct omc_ctx {
  int device;
  int64_t n_chains;
  uint64_t seed;
  int64_t chain_offset;
  hipStream_t stream;
  bool own_stream;
  long long* d_bad_chain;  // device word: min local chain index with a non-positive pivot, or LLONG_MAX
  unsigned long long* d_fallbacks;  // device word behind d_bad_chain: chain-updates that took the sequential join fallback
                                    // (d_fallbacks + 1: hand-overs of a several-sweeps launch that never arrived; + 2: groups of the
                                    //  segmented band route handed to the one-piece kernel; + 3: groups that needed its second attempt)
  void* d_gamma_tab;                // generic tridiagonal instantiation: device image of a launch's Normal-Gamma blocks and streams
  unsigned long long* d_handoff;    // omc_gmrf_run: [n_chains][16] hand-over lines, allocated on first use
  uint32_t run_epoch;               // tag counter of the hand-over lines
  int run_sweeps_per_launch;        // omc_gmrf_run: sweeps per launch (1 = one launch per sweep)
  int run_reenter;                  // omc_gmrf_run: 1, 2 = a chain's workgroup restarts itself for the next sweep of the launch
  int run_block_sweeps;             // omc_gmrf_run: sweeps a self-restarting workgroup walks before a fresh one takes over (0: the whole launch)
  int run_reenter_force;            // 1: take that form whatever the chain count (tests); 0: only when the chains fill the CUs in whole rounds
  hipEvent_t run_ev_begin, run_ev_end;  // options "run_event_begin" / "run_event_end": caller-owned events omc_gmrf_run records around its launches
  double* long_band; size_t long_band_bytes;  // chains beyond one workgroup: the terms in band storage (omc_tridiag.hip, long_chain_draw)
  double* long_quad; size_t long_quad_bytes;  // ... and the sweep's quadratic forms [term][chain]
  double* workspace;       // scratch for the serial kernel (l vectors), grown on demand
  size_t workspace_bytes;
  void* store_ws; size_t store_ws_bytes;  // omc_store.hip: histograms / partial moments of the store summaries
  // dense path (omc_dense.hip): rocBLAS handle and workspaces, created on first use
  void* blas;
  // blocked dense factorisation: second half of the chains on a side stream (forked from / joined into `stream` by events),
  // so that one half's panel kernel (one workgroup per chain, latency-bound) runs under the other half's update GEMM
  void* blas_aux;
  hipStream_t aux_stream;
  hipEvent_t ev_fork, ev_join;
  int dense_blocked_min;  // option "dense_blocked_min": smallest order that takes the blocked factorisation (rocSOLVER below)
  int dense_overlap;  // option "dense_overlap": 1 (default) = split the chains in two halves when there are >= 64; 0 = one batch
  double* dense_factor; size_t dense_factor_bytes;
  double* dense_winv; size_t dense_winv_bytes;  // blocked factorisation: inverses of the current diagonal blocks [C][64][64]
  int dense_panel_old;  // option "dense_panel_old": 1 = the one-kernel panel of rounds 1-3 (cross-checks)
  int* dense_info; size_t dense_info_bytes;
  double* slice_buf; size_t slice_buf_bytes;  // partial products of a sliced contraction (omc_design_predict) [S][C][n]
  double* dense_tmp; size_t dense_tmp_bytes;
  double* rj_tmp; size_t rj_tmp_bytes;   // omc_knot_loop: the proposals of all knots
  double* mh_work; size_t mh_work_bytes;  // omc_mala.hip
  double* mala_prep; size_t mala_prep_bytes;  // cached drift matrix and L^{-T} of the current (Q, L, step)
  const double* mala_Q; const double* mala_L; double mala_step; int64_t mala_d;
  double* white_prep; size_t white_prep_bytes; const double* white_L; const double* white_mu; int64_t white_d;  // omc_mala_step_white
  double* white_a; size_t white_a_bytes; const double* white_x; int64_t white_ld;  // a = L'(x - mu) of the state at white_x
  double* white_traj; size_t white_traj_bytes;  // omc_mala_run_white: the whitened states of two blocks of steps [2][32][C][d]
  hipEvent_t white_ev[4];  // ... trajectory of block b written (0, 1) / mapped back by the product on the side stream (2, 3)
  double* rww_a; size_t rww_a_bytes; const double* rww_x; int64_t rww_ld; const double* rww_mu; double* rww_mu_neg; size_t rww_mu_bytes;  // omc_rw_step_white
  double* rw_prep; size_t rw_prep_bytes; const double* rw_LQ; int64_t rw_d;  // omc_rw_step: LQ with a zero upper triangle
  int tridiag_algo;  // 0 auto, 1 serial, 2 segmented
  int tridiag_seg;   // 0 auto, else nodes per lane
  int debug_zero_z;  // diagnostic: skip the draw generation (timing what-if only)
  int tridiag_newton_max;  // Newton join corrections before the sequential fallback (default OMC_NEWTON_MAX = 4; 0 forces the fallback: tests)
  int tridiag_quad_skip;    // bit k: the generic tridiagonal draw does not take term k's fused quadratic form (its quad_out entry is left unwritten)
  int tridiag_perturb_ppb;  // tests only: relative error (parts per billion) put on the Moebius start values of the segment joins
  int tridiag_generic;  // 1: never take the structure-specialised instantiation of the segmented kernel (tests)
  int band_algo;  // 0 auto, 1 lane-per-chain in one piece (narrow bands), 2 workgroup-per-chain
  int band_seg_overlap;  // segmented lane kernel: columns of warm-up before a segment (default 192)
  int band_seg_count;    // segmented lane kernel: number of segments (0 = chosen for the SIMDs; tuning and tests)
  int band_blocked_threads;  // blocked band kernel: 0 = form chosen by what fits a CU; 512, 4, 8 force one (A/B, tests; omc_bandwide.hip)
  int mh_gemm_ksplit;  // omc_dgemm_small: groups of four waves per workgroup cutting the contraction (1, 2 or 4)
  int mh_use_rocblas;  // 1: the products of the fused Metropolis-Hastings steps through rocBLAS DGEMM instead of omc_dgemm_small (cross-checks)
  int gram_use_rocblas;  // 1: X' diag(w) X through rocBLAS (scaled copy of X + DGEMM) instead of the own MFMA kernel (cross-checks)
  int dense_use_rocsolver;  // 1: factor dense precisions with rocSOLVER's batched potrf instead of the blocked route
  unsigned long long* stamps;  // diagnostic phase stamps of the segmented kernel (NULL = off)
  // diagnostic sweep clock (options "sweep_times_ptr" / "sweep_times_cap"): the workgroup-per-chain tridiagonal kernel writes
  // {s_memrealtime at entry, at exit} of every (sweep, chain) into a caller-owned ring [cap][n_chains][2] of uint64
  unsigned long long* sweep_times;
  int64_t sweep_times_cap, sweep_times_pos;
  // launch log of the last omc_gmrf_run call (omc_ctx_launch_log): host clock around every kernel launch it issued
  struct LaunchRec { double t_begin, t_end; int n_sweeps, form; int64_t ring_pos; };
  LaunchRec launch_log[OMC_LAUNCH_LOG_MAX];
  int launch_log_n, launch_log_total;
};

void omc_set_error(const char* what, hipError_t e);
void omc_set_error_text(const char* text);  // any other library failure (RCCL) for omc_last_error()
void omc_dense_release(omc_ctx* ctx);
int omc_reentry_probe_result(omc_ctx* ctx);  // omc_tridiag.hip: 1 if the loaded kernel descriptors match what a self-restart reproduces
// omc_gemm.hip: small-state fp64 MFMA GEMM, C = A0 B0 (+ A1 B1) (+ addv per column), column-major
omc_status omc_dgemm_small(omc_ctx* ctx, int M, int N, const double* A0, int64_t lda0, const double* B0, int64_t ldb0, int K0,
                           const double* A1, int64_t lda1, const double* B1, int64_t ldb1, int K1, int tri, const double* addv,
                           double* Cout, int64_t ldc, const int* colmask = nullptr, int ksplit = 0);
omc_status omc_ensure_bytes(omc_ctx* ctx, void** buf, size_t* have, size_t need);  // grow-on-demand workspace (omc_dense.hip)
// omc_gemm.hip: C = A B (+ addv per column) for thousands of columns (64 x 64 tiles, one operand pair)
omc_status omc_dgemm_wide(omc_ctx* ctx, int M, int N, const double* A, int64_t lda, const double* B, int64_t ldb, int K, int tri,
                          const double* addv, double* Cout, int64_t ldc);
omc_status omc_ensure_aux(omc_ctx* ctx);  // side stream + its events and BLAS handle, made on first use (omc_dense.hip)
omc_status omc_col_moments(omc_ctx* ctx, const double* data, int64_t R, int64_t K, double* mean_out, double* var_out);  // omc_store.hip
extern "C" omc_status omc_gram_mfma_launch(omc_ctx* ctx, int64_t n, int64_t p, const double* X, const double* w, double* G_out);  // omc_gram.hip  // destroys the rocBLAS handle if one was created

#define OMC_HIP_CHECK(expr)                  \
  do {                                       \
    hipError_t _e = (expr);                  \
    if (_e != hipSuccess) {                  \
      omc_set_error(#expr, _e);              \
      return OMC_HIP_ERROR;                  \
    }                                        \
  } while (0)

#define OMC_NO_BAD_CHAIN 0x7fffffffffffffffLL

// ------------------------------------------------------------------------------------------
// Random streams: Philox4x32-10 (Salmon et al., SC'11), the generator rocRAND's
// rocrand_philox4x32_10 implements.  Counter layout (documented in DESIGN.md):
//   key = (seed_lo, seed_hi)
//   ctr = (block, draw_index_lo, global_chain_lo, purpose<<24 | global_chain_hi(8b)<<16 | draw_index_hi(16b))
// so a draw is a pure function of (seed, global chain id, draw_index, position): independent of
// launch geometry and of how chains are sharded over GPUs.
enum : uint32_t { OMC_RNG_NORMAL = 0, OMC_RNG_GAMMA = 1, OMC_RNG_UNIFORM = 2, OMC_RNG_RAW = 3 };

struct omc_rng_key {
  uint32_t k0, k1;    // seed
  uint32_t c1;        // draw_index lo
  uint32_t c3_base;   // purpose and draw_index hi; chain hi bits are OR-ed in per chain
};

__host__ __device__ inline omc_rng_key omc_make_key(uint64_t seed, uint64_t draw_index, uint32_t purpose) {
  omc_rng_key k;
  k.k0 = (uint32_t)seed;
  k.k1 = (uint32_t)(seed >> 32);
  k.c1 = (uint32_t)draw_index;
  k.c3_base = (purpose << 24) | (uint32_t)((draw_index >> 32) & 0xffffu);
  return k;
}

// one round, key schedule included.  XOR3: hi ^ c ^ k as one v_bitop3_b32 (truth table 0x96 = three-input XOR, new on
// gfx950) where the compiler's own selection leaves two v_xor_b32 -- 14 vector instructions per block.  Used from round 3
// on, where all four words vary per lane in every kernel; in the workgroup-per-chain kernels only the block index differs
// between the lanes at first, and the plain form lets the scalar unit do the wave-uniform part of rounds 0-2.
template <bool XOR3>
__device__ __forceinline__ void omc_philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t& k0,
                                                 uint32_t& k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  uint32_t n0, n2;
  if constexpr (XOR3) {
    n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
    n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
  } else {
    n0 = hi1 ^ c1 ^ k0;
    n2 = hi0 ^ c3 ^ k1;
  }
  c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
  k0 += W0; k1 += W1;
}

#ifndef OMC_XOR3_FROM
#define OMC_XOR3_FROM 3  // first round that uses the three-input XOR (10 = never: A/B builds)
#endif
// round r of an unrolled loop (r folds to a constant there)
__device__ __forceinline__ void omc_philox_round_r(int r, uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t& k0,
                                                   uint32_t& k1) {
  if (r >= OMC_XOR3_FROM) omc_philox_round<true>(c0, c1, c2, c3, k0, k1);
  else omc_philox_round<false>(c0, c1, c2, c3, k0, k1);
}

__device__ __forceinline__ uint4 omc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                   uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) omc_philox_round_r(r, c0, c1, c2, c3, k0, k1);
  return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ uint4 omc_rng_block(const omc_rng_key& k, int64_t global_chain, uint32_t block) {
  uint32_t c2 = (uint32_t)global_chain;
  uint32_t c3 = k.c3_base | ((uint32_t)((uint64_t)global_chain >> 32) & 0xffu) << 16;
  return omc_philox4x32_10(block, k.c1, c2, c3, k.k0, k.k1);
}

// 53-bit uniform in (0, 1] from two words (same mapping as rocRAND's box_muller_double).
__device__ __forceinline__ double omc_u53(uint32_t lo, uint32_t hi) {
  unsigned long long v = (unsigned long long)lo ^ ((unsigned long long)hi << 21);
  return 0x1.0p-53 + (double)v * 0x1.0p-53;
}

// ---- lean fp64 elementary functions for the Box-Muller transform -------------------------------
// The generic libm entry points carry range/special-case handling the transform never needs
// (u in (0,1], angle in (0,2]); these restate the classic fdlibm kernels (e_log.c, k_sin.c,
// k_cos.c: argument reduction + minimax polynomial, < 1 ulp) for exactly those domains.
__host__ __device__ inline double omc_rcp_nr(double d) {
#if !defined(__HIP_DEVICE_COMPILE__)
  return 1.0 / d;
#else
  // v_rcp_f64 is good to 4.6e-8; r (1 + e + e^2) with e = 1 - d r leaves e^3 ~ 1e-22 and one rounding (three
  // operations where two Newton steps take four)
  const double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  return fma(r, fma(e, e, e), r);
#endif
}
__device__ __forceinline__ double omc_sqrt_nr(double r) {  // r > 0, normal range
  const double g = __builtin_amdgcn_rsq(r);
  double s = r * g;
  const double h = 0.5 * g;
  double e = fma(-s, s, r);
  s = fma(e, h, s);
  e = fma(-s, s, r);
  return fma(e, h, s);
}
// fma with a constant addend (or factor) taken from a scalar register pair.  A VOP3 fp64 instruction cannot
// carry a 64-bit literal, and for a constant addend the compiler's choice is v_fmac with the constant first
// copied into the destination: two v_mov_b32 per polynomial coefficient (36 of the 188 vector instructions of
// one pair of draws), i.e. vector-ALU issue slots -- what the sampling kernels are bound by -- for something
// two s_mov_b32 on the scalar unit do beside them.
__device__ __forceinline__ double omc_fma_vvs(double a, double b, double c_scalar) {  // a*b + c
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_scalar));
  return d;
}
__device__ __forceinline__ double omc_fma_vsv(double a, double b_scalar, double c) {  // a*b + c
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_scalar), "v"(c));
  return d;
}

__device__ __forceinline__ double omc_mul_vs(double a, double b_scalar) {  // a*b
  double d;
  asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b_scalar));
  return d;
}
__device__ __forceinline__ double omc_add_vs(double a, double b_scalar) {  // a+b
  double d;
  asm("v_add_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b_scalar));
  return d;
}

// log(u) for a positive normal u (written for 2^-53 <= u <= 1, the uniforms; nothing in it depends on that range)
__device__ __forceinline__ double omc_log_unit(double u) {
  int k = __builtin_amdgcn_frexp_exp(u);        // u = m * 2^k, m in [0.5, 1)
  double m = __builtin_amdgcn_frexp_mant(u);
  if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; }
  const double f = m - 1.0;
  const double s = f * omc_rcp_nr(2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * omc_fma_vvs(w, omc_fma_vsv(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
                                    3.999999999940941908e-01);
  const double t2 = z * omc_fma_vvs(w, omc_fma_vvs(w, omc_fma_vsv(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                                   2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
  return omc_mul_vs(dk, 6.93147180369123816490e-01) - ((hfsq - fma(s, hfsq + R, omc_mul_vs(dk, 1.90821492927058770002e-10))) - f);
}
// sin(pi a), cos(pi a) for 0 < a <= 2
__device__ __forceinline__ void omc_sincospi02(double a, double& sn, double& cs) {
  const double n = rint(2.0 * a);               // 0..4 quarter turns
  const double t = omc_mul_vs(fma(-0.5, n, a), 3.14159265358979311600e+00);  // |t| <= pi/4, reduction exact
  const double z = t * t;
  const double ps = omc_fma_vvs(z, omc_fma_vvs(z, omc_fma_vvs(z, omc_fma_vsv(z, 1.58969099521155010221e-10,
                                                                           -2.50507602534068634195e-08),
                                                          2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                8.33333333332248946124e-03);
  const double s0 = fma(z * t, omc_fma_vvs(z, ps, -1.66666666666666324348e-01), t);
  const double pc = omc_fma_vvs(z, omc_fma_vvs(z, omc_fma_vvs(z, omc_fma_vvs(z, omc_fma_vsv(z, -1.13596475577881948265e-11,
                                                                                         2.08757232129817482790e-09),
                                                                        -2.75573143513906633035e-07),
                                                          2.48015872894767294178e-05), -1.38888888888741095749e-03),
                                4.16666666666666019037e-02);
  const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)n & 3;
  const double sa = (q & 1) ? c0 : s0, ca = (q & 1) ? s0 : c0;
  sn = (q == 2 || q == 3) ? -sa : sa;
  cs = (q == 1 || q == 2) ? -ca : ca;
}

// ---- host/device shims: the Box-Muller map below also compiles for the host, where tests/ checks it against
// the NumPy model (tests/philox_model.py) without a GPU ----
#if defined(__HIP_DEVICE_COMPILE__)
#define OMC_ON_DEVICE 1
#else
#define OMC_ON_DEVICE 0
#endif
__host__ __device__ inline double omc_bits_to_double(uint32_t hi, uint32_t lo) {
#if OMC_ON_DEVICE
  return __hiloint2double((int)hi, (int)lo);
#else
  const uint64_t b = ((uint64_t)hi << 32) | lo;
  double d;
  __builtin_memcpy(&d, &b, 8);
  return d;
#endif
}
// double in [1, 2) whose 52 mantissa bits are the top 20 bits of `hi20src` and all of `lo`
__host__ __device__ inline double omc_unit_mantissa(uint32_t lo, uint32_t hi20src) {
#if OMC_ON_DEVICE
  return omc_bits_to_double(__builtin_amdgcn_alignbit(0x3FFu, hi20src, 12), lo);  // one v_alignbit_b32
#else
  return omc_bits_to_double(0x3FF00000u | (hi20src >> 12), lo);
#endif
}
// v with the sign bit of `signsrc` (bit 31)
__host__ __device__ inline double omc_with_sign_of(double v, uint32_t signsrc) {
#if OMC_ON_DEVICE
  const uint32_t hi = (uint32_t)__double2hiint(v);
  return omc_bits_to_double((hi & 0x7fffffffu) | (signsrc & 0x80000000u), (uint32_t)__double2loint(v));  // one v_bfi_b32
#else
  uint64_t b;
  __builtin_memcpy(&b, &v, 8);
  b = (b & 0x7fffffffffffffffull) | ((uint64_t)(signsrc & 0x80000000u) << 32);
  __builtin_memcpy(&v, &b, 8);
  return v;
#endif
}
#if OMC_ON_DEVICE
#define OMC_FMA_VVS(a, b, c) omc_fma_vvs(a, b, c)
#define OMC_MUL_VS(a, b) omc_mul_vs(a, b)
#define OMC_ADD_VS(a, b) omc_add_vs(a, b)
#else
#define OMC_FMA_VVS(a, b, c) fma(a, b, c)
#define OMC_MUL_VS(a, b) ((a) * (b))
#define OMC_ADD_VS(a, b) ((a) + (b))
#endif

// One Philox block -> two independent N(0,1): Box-Muller in fp64, 128 random bits per pair (the budget of
// rocRAND's box_muller_double), arranged for the vector ALU the sampling kernels are bound by:
//  * both uniforms are spliced straight into mantissas (one v_alignbit_b32 each) instead of being converted
//    from 64-bit integers (13 instructions each, two of them quarter-rate multiplies):
//        radius   u = 2 - m,  m in [1,2) from w.x and the top 20 bits of w.y      -> u in (0,1] on a 2^-52 grid
//        angle    t = (m' - 3/2) pi/2,  m' from w.z and the top 20 bits of w.w    -> t in [-pi/4, pi/4)
//  * the point (cos t, sin t) on the arc around +x is carried to the whole circle by a random reflection
//    x -> -x (bit 11 of w.w) and a random swap of the coordinates (bit 10): one v_bfi_b32 and four
//    v_cndmask_b32 in place of a quadrant reduction;
//  * -2 log u comes out of the fdlibm log kernel directly (doubled coefficients, one constant for ln 2: u
//    never needs the hi/lo split, |k ln2| and |log1p f| do not cancel on (0,1]), and its square root takes
//    one Newton step on v_rsq_f64 (4e-15 relative: the draw's own scale).
// 70 vector instructions per pair where the integer-conversion form took 111; tests/philox_model.py is the
// host model of exactly this map.
__host__ __device__ inline void omc_normal_pair(uint4 w, double& n0, double& n1) {
  // ---- radius: V = -2 log u ----
  const double u = 2.0 - omc_unit_mantissa(w.x, w.y);
#if OMC_ON_DEVICE
  int k = __builtin_amdgcn_frexp_exp(u);  // u = g 2^k, g in [0.5, 1)
  double g = __builtin_amdgcn_frexp_mant(u);
#else
  int k;
  double g = frexp(u, &k);
#endif
  if (g < 0.70710678118654752440) { g += g; k -= 1; }
  const double f = g - 1.0;
  const double s = f * omc_rcp_nr(2.0 + f);
  const double z = s * s;
  // 2 R(z) = 2 z (Lg1 + z (Lg2 + ... + z Lg7)), e_log.c coefficients doubled
  double p = OMC_ADD_VS(OMC_MUL_VS(z, 2.959639721023317182e-01), 3.062767539841874664e-01);
  p = OMC_FMA_VVS(z, p, 3.636714432323610024e-01);
  p = OMC_FMA_VVS(z, p, 4.444439686429956792e-01);
  p = OMC_FMA_VVS(z, p, 5.714285748732478298e-01);
  p = OMC_FMA_VVS(z, p, 7.999999999881883816e-01);
  p = OMC_FMA_VVS(z, p, 1.333333333333347026e+00);
  const double ff = f * f;
  // -2 log u = -2 k ln2 - 2 f + f^2 - s (f^2 + 2R)
  double V = fma(-s, fma(z, p, ff), ff);
  V = fma(f, -2.0, V);
  V = fma((double)k, -1.38629436111989061883e+00, V);
  V = fmax(V, 1e-300);  // u == 1 -> radius 0, not NaN
#if OMC_ON_DEVICE
  const double rs = __builtin_amdgcn_rsq(V);
  const double r0 = V * rs;
  const double r = fma(fma(-r0, r0, V), 0.5 * rs, r0);
#else
  const double r = sqrt(V);
#endif
  // ---- angle ----
  const double t = OMC_MUL_VS(OMC_ADD_VS(omc_unit_mantissa(w.z, w.w), -1.5), 1.57079632679489655800e+00);
  const double q = t * t;
  // k_sin.c / k_cos.c kernels in plain Horner form (|t| <= pi/4)
  double ps = OMC_ADD_VS(OMC_MUL_VS(q, 1.58969099521155010221e-10), -2.50507602534068634195e-08);
  ps = OMC_FMA_VVS(q, ps, 2.75573137070700676789e-06);
  ps = OMC_FMA_VVS(q, ps, -1.98412698298579493134e-04);
  ps = OMC_FMA_VVS(q, ps, 8.33333333332248946124e-03);
  ps = OMC_FMA_VVS(q, ps, -1.66666666666666324348e-01);
  const double sn = fma(t * q, ps, t);
  double pc = OMC_ADD_VS(OMC_MUL_VS(q, -1.13596475577881948265e-11), 2.08757232129817482790e-09);
  pc = OMC_FMA_VVS(q, pc, -2.75573143513906633035e-07);
  pc = OMC_FMA_VVS(q, pc, 2.48015872894767294178e-05);
  pc = OMC_FMA_VVS(q, pc, -1.38888888888741095749e-03);
  pc = OMC_FMA_VVS(q, pc, 4.16666666666666019037e-02);
  pc = fma(q, pc, -0.5);
  const double cs = omc_with_sign_of(fma(q, pc, 1.0), w.w << 20);  // random reflection x -> -x (bit 11)
  const bool swap = (int32_t)(w.w << 21) < 0;                          // random swap (bit 10)
  n0 = (swap ? cs : sn) * r;
  n1 = (swap ? sn : cs) * r;
}

// Marsaglia & Tsang (2000) Gamma(a,1), a > 0, from the chain's Philox stream.
// Attempt j (j = 0, 1, ...) consumes blocks 1+2j (two N(0,1) candidates) and 2+2j (their two
// uniforms); block 0 feeds the a < 1 boost Gamma(a) = Gamma(a+1) U^(1/a).  The draw is the first
// accepted candidate in (attempt, candidate) order, so evaluating attempts on different lanes and
// taking the lowest accepted one gives the same value as the serial loop.
struct omc_gamma_prep { double d, cst, boost; };

__device__ __forceinline__ omc_gamma_prep omc_gamma_prepare(const omc_rng_key& key, int64_t gc, double a) {
  omc_gamma_prep p;
  p.boost = 1.0;
  if (a < 1.0) {
    const uint4 w = omc_rng_block(key, gc, 0u);
    p.boost = exp(omc_log_unit(omc_u53(w.x, w.y)) / a);
    a += 1.0;
  }
  p.d = a - 1.0 / 3.0;
  p.cst = omc_rcp_nr(omc_sqrt_nr(9.0 * p.d));
  return p;
}

// acceptance test of one candidate (x ~ N(0,1), u ~ U(0,1])
__device__ __forceinline__ bool omc_gamma_candidate(const omc_gamma_prep& p, double x, double u, double& value) {
  double v = fma(p.cst, x, 1.0);
  if (!(v > 0.0)) return false;
  v = v * v * v;
  const double x2 = x * x;
  if (u < 1.0 - 0.0331 * x2 * x2 || omc_log_unit(u) < fma(p.d, (1.0 - v) + log(v), 0.5 * x2)) {
    value = p.boost * p.d * v;
    return true;
  }
  return false;
}

// one attempt: returns true and the Gamma(a,1) value if one of its two candidates is accepted
__device__ __forceinline__ bool omc_gamma_attempt(const omc_rng_key& key, int64_t gc, const omc_gamma_prep& p,
                                                  uint32_t attempt, double& value) {
  double x0, x1;
  omc_normal_pair(omc_rng_block(key, gc, 1u + 2u * attempt), x0, x1);
  const uint4 w = omc_rng_block(key, gc, 2u + 2u * attempt);
  if (omc_gamma_candidate(p, x0, omc_u53(w.x, w.y), value)) return true;
  return omc_gamma_candidate(p, x1, omc_u53(w.z, w.w), value);
}

__device__ inline double omc_standard_gamma(const omc_rng_key& key, int64_t gc, double a, bool* failed) {
  const omc_gamma_prep p = omc_gamma_prepare(key, gc, a);
  double v;
  for (uint32_t attempt = 0; attempt < 256; ++attempt)
    if (omc_gamma_attempt(key, gc, p, attempt, v)) return v;
  *failed = true;
  return p.boost * p.d;
}
