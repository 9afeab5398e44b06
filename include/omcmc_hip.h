/*
 * omcmc_hip.h -- C ABI of libomcmc_hip.so, the MI355X (gfx950) sampler core.
 *
 * The reference (sede-open/openMCMC v1.0.7) is pure Python and has no FFI: the boundary it
 * offers is the duck-typed plugin API MCMCSampler.sample(state)->state called from
 * MCMC.run_mcmc (mcmc.py:97-100).  This header is what a ctypes binding for that path needs:
 * every entry point replaces the arithmetic behind one group of reference call sites, cited
 * per function as file:line relative to /root/reference/src/openmcmc/.  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no torch / STL types; every `const double*` / `double*` is a DEVICE pointer
 *     (hipMalloc, or torch.Tensor.data_ptr() of a float64 ROCm tensor) unless marked [host];
 *   - "C" = number of chains held by the context; per-chain vectors are chain-major:
 *     element i of chain c lives at base[c*ld + i] (ld >= n, leading stride in elements);
 *     per-chain scalars are arrays of length C; vectors shared by all chains have length n;
 *   - all arithmetic is IEEE fp64;
 *   - calls on one context are issued to that context's HIP stream and return without
 *     synchronising; numerical failures (non-positive pivot) are latched on the device and
 *     read with omc_ctx_status(), which synchronises;
 *   - every function returns an omc_status; nothing is retained after omc_ctx_destroy;
 *   - every entry point that consumes randomness takes an optional pointer to injected draws
 *     (the parity tests replay the reference's draws through it) and otherwise uses the
 *     in-kernel Philox4x32-10 stream keyed by (seed, global chain id, draw_index), so results
 *     do not depend on how chains are sharded over GPUs.
 */
#ifndef OMCMC_HIP_H
#define OMCMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  OMC_OK = 0,
  OMC_INVALID_ARG = 1,  /* -> ValueError  (gmrf.py:508-509, gmrf.py:149-150)                */
  OMC_NOT_POSDEF = 2,   /* -> numpy.linalg.LinAlgError (gmrf.py:518 dense fallback raises)   */
  OMC_HIP_ERROR = 3,    /* -> RuntimeError                                                   */
  OMC_UNSUPPORTED = 4   /* -> NotImplementedError                                            */
} omc_status;

typedef struct omc_ctx omc_ctx;

#define OMC_MAX_TERMS 4
#define OMC_SELECT_MAX 8   /* state entries per omc_chain_select_multi call */

/* A conditional precision in "shared structure x per-chain scalar" form
 *     Q_c = sum_k scale[k][c] * M_k,   M_k symmetric tridiagonal, shared by all chains,
 *     rhs_c = sum_k scale[k][c] * rhs[k]  (+ optional per-chain vector),
 * which is what NormalNormal.sample assembles from ScaledMatrix precisions (sampler.py:176-192,
 * parameter.py:329): term 0 is usually the prior (lambda*P, rhs = P m), the others likelihoods
 * (tau*W, rhs = A'W(y-d)).  center[k] is the vector the quadratic form of term k is taken
 * around: quad[k][c] = (x_c - center_k)' M_k (x_c - center_k), the quantity NormalGamma.sample
 * needs (sampler.py:276,284) and Normal.log_p reuses (gmrf.py:343-344).                    */
typedef struct {
  int32_t n_terms;                      /* 1..OMC_MAX_TERMS                                  */
  const double* diag[OMC_MAX_TERMS];    /* [n]   shared; NULL = all ones                     */
  const double* off[OMC_MAX_TERMS];     /* [n-1] shared; NULL = zeros (diagonal M_k)         */
  const double* rhs[OMC_MAX_TERMS];     /* [n]   shared M_k m_k;  NULL = zeros               */
  const double* center[OMC_MAX_TERMS];  /* [n]   shared m_k;      NULL = zeros               */
  const double* scale[OMC_MAX_TERMS];   /* [C]   per-chain scalar; NULL = 1                  */
  /* Per-chain part of a term's centre (ABI 2): a hierarchical model's prior mean that is itself sampled, or a
   * sampled response whose mean is the parameter (sampler.py:181-192 with state[...] differing per chain).  Term k
   * is then centred at center[k] + center_chain[k][c]:  rhs_c += scale[k][c] * M_k center_chain[k][c]  (formed
   * inside the launch: no product vector travels through memory) and quad[k][c] is taken around that centre.
   * NULL = none; at most ONE term of a call may have one.  Taken by the workgroup-per-chain form of the draw
   * (omc_tridiag_takes_center_chain) and by omc_tridiag_quadform; every other entry point -- and a call with two
   * such terms -- returns OMC_UNSUPPORTED.                                                                */
  const double* center_chain[OMC_MAX_TERMS];  /* [C][ld_center_chain]                        */
  int64_t ld_center_chain;
} omc_tridiag_terms;

/* ---- context ---------------------------------------------------------------------------- */

/* device: HIP device ordinal.  stream: the hipStream_t every call of this context is issued to
 * (e.g. torch.cuda.current_stream().cuda_stream; 0 is the legacy default stream); with
 * create_stream != 0 `stream` is ignored and the library creates and owns a non-blocking stream.
 * chain_id_offset: global id of local chain 0 (rank * chains_per_rank), used only to key the
 * random stream.                                                                             */
omc_status omc_ctx_create(int32_t device, int64_t n_chains, uint64_t seed, int64_t chain_id_offset,
                          void* stream, int32_t create_stream, omc_ctx** out);
omc_status omc_ctx_destroy(omc_ctx* ctx);
/* Synchronises the stream.  *first_bad_chain = -1 if no failure has been latched since the
 * last call, else the smallest LOCAL chain index whose factorisation met a non-positive pivot
 * (returns OMC_NOT_POSDEF; the latch is cleared).                                            */
omc_status omc_ctx_status(omc_ctx* ctx, int64_t* first_bad_chain);
omc_status omc_ctx_synchronize(omc_ctx* ctx);
/* Tuning knobs, by name: "tridiag_algo" (0 auto, 1 serial lane-per-chain, 2 segmented),
 * "tridiag_seg" (nodes per lane: 0 auto, 8, 10, 16, 20, 32), "tridiag_generic" (1: never use the
 * instantiation specialised for the two-term smoother structure; for cross-checks), "tridiag_quad_skip" (bit k set: the generic
 * workgroup-per-chain draw skips term k's fused quadratic form -- a hierarchical sweep knows which of them a later block will
 * invalidate --, that quad_out entry is then meaningless), "tridiag_newton_max" (0..64,
 * default 4: Newton corrections of the segment joins before the sequential fallback takes over; 0 forces it),
 * "tridiag_perturb_ppb" (tests: relative error, in 1e-9, put on the segments' start pivots so that the join test must
 * reject them), "run_sweeps_per_launch" (1..32, default 32: sweeps omc_gmrf_run issues per launch), "run_reenter" (0..2, default 2:
 * within such a launch a chain's workgroup restarts itself for the next sweep instead of one workgroup per sweep and chain;
 * 2: without a barrier in front of the restart, the scales handed from sweep to sweep through LDS; by default the
 * restarting form is taken when the chains fill the CUs in whole rounds and the (sweep, chain) grid otherwise, an
 * explicit setting holds for every chain count),
 * "band_algo" (0 auto; 1 narrow bands one lane per chain in ONE piece; 2 one workgroup per chain, a column per step; 3 one
 * workgroup per chain in blocks of 16 columns, the next block factorised ahead, the window update on the matrix cores -- auto
 * takes it from w = 9, and from w = 4 on up to 3072 chains, where a lane per chain leaves the SIMDs to lone waves), "band_seg_overlap"
 * (8..65536, default 192: columns of warm-up before a segment of the segmented narrow-band route), "band_seg_count" (0, 2..128;
 * default 0: the number of segments of that route is chosen for the device's SIMDs; same results to rounding),
 * "band_blocked_threads" (0, 4, 8, 16 or 512; default 0: the blocked band kernel picks its form by what fits a CU -- four waves per
 * chain for bands narrower than 16, 128-register forms with several workgroups to a CU when there are more chains than CUs;
 * 512 keeps eight waves and one workgroup per CU, 4 / 16 / 8 force the 128-register forms: same results to rounding),
 * "dense_overlap" (0/1, default 1: the blocked dense factorisation runs the two halves of the chains on two streams,
 * forked from and joined into the context's stream by events; bit-identical results), "dense_use_rocsolver" (1: rocSOLVER's
 * potrf instead of the blocked route; cross-checks), "dense_blocked_min" (default 144: smallest order that takes the blocked
 * factorisation; rocSOLVER's small kernels below), "dense_panel_old" (1: the blocked factorisation's panel as ONE kernel per
 * block column, as in rounds 1-3, instead of the register-resident diagonal factor + matrix-core row update; same factor bit for
 * bit, cross-checks and A/B timing).
 * Diagnostics of omc_gmrf_run: "sweep_times_cap" then "sweep_times_ptr" = capacity (in sweeps, >= 64) and device address
 * of a caller-owned ring [cap][C][2] of uint64: wave 0 of the workgroup of every (sweep, chain) leaves the device's
 * constant-rate counter (s_memrealtime; rate: counter "wall_clock_khz") at its entry and at its exit in record
 * (position + sweep) mod cap; the position restarts at 0 when either option is set and advances by the sweeps of every
 * launch (counter "sweep_times_pos"); 0 = off (the default).  One 16-byte store per workgroup and sweep.
 * "run_event_begin" / "run_event_end" = a caller-owned hipEvent_t (as an integer; 0 = none, the default): omc_gmrf_run records
 * it on the context's stream in front of its first / behind its last launch (several-sweeps-per-launch route).
 * Unknown name -> OMC_INVALID_ARG.                                                            */
omc_status omc_ctx_set_option(omc_ctx* ctx, const char* name, int64_t value);
/* Diagnostic counters, by name (synchronises): "tridiag_join_fallbacks" = chain-updates of the segmented
 * tridiagonal kernel whose pivot joins did not meet the Newton tolerance within its iteration limit and were
 * made consistent by the sequential recurrence instead (same pivots as the serial kernel; slow, rare);
 * "band_join_retries" = groups of 64 chains of the segmented narrow-band route whose segment joins did not close
 * within the warm-up and were redone with four times the warm-up; "band_join_fallbacks" = those that did not close
 * then either and were factorised in one piece instead; "run_handoff_timeouts" = sweeps of a several-sweeps launch whose
 * scales never arrived from the chain's previous sweep (that sweep and the chain's later ones then run on NaN scales, and
 * omc_ctx_status reports the run as failed).  Not counters, no synchronisation: "wall_clock_khz" (rate of the sweep clock),
 * "sweep_times_pos"; "reenter_abi_ok" = 1 if the kernel descriptors of the loaded code object ask the dispatcher for exactly the
 * entry state a self-restarting workgroup reproduces (read back from the device on first use; 0 switches "run_reenter" off
 * for the process).                                                                                             */
omc_status omc_ctx_counter(omc_ctx* ctx, const char* name, int64_t* value);
/* Launch log of the LAST omc_gmrf_run call on this context [host]: *n_launches = kernel launches it issued; for the first
 * min(cap, 64) of them out[5 i .. 5 i + 4] = {host CLOCK_MONOTONIC seconds just before and just after the launch call,
 * sweeps in the launch, launch form (0: one workgroup per (sweep, chain); 1, 2: self-restarting workgroups), ring position
 * of the launch's first sweep in the sweep clock or -1}.  Together with the sweep clock it tells where the time of a slow
 * run went: the host issuing late, the queue starting late, or sweeps that ran long (and which chains').          */
omc_status omc_ctx_launch_log(omc_ctx* ctx, double* out, int64_t cap, int64_t* n_launches);
/* [host, no GPU] The test behind "reenter_abi_ok", on three words of an AMDGPU kernel descriptor (bytes 4-7, 52-55 and
 * 56-59: PRIVATE_SEGMENT_FIXED_SIZE, COMPUTE_PGM_RSRC2, kernel code properties | kernarg preload spec << 16): 1 if a
 * workgroup of that kernel is handed exactly s[0:1] = kernel-argument pointer, s2 = workgroup id x, v0 = work-item id and
 * has no private segment -- the entry state omc_gmrf_run's self-restarting workgroups reproduce by hand.  The build-time
 * test (tests/test_kernel_resources.py) feeds it the descriptors of the compiled code object.                        */
int32_t omc_reentry_descriptor_ok(uint32_t private_segment_fixed_size, uint32_t compute_pgm_rsrc2, uint32_t properties_and_preload);
const char* omc_last_error(void);          /* [host] text of the last HIP failure, thread-local */
int32_t omc_abi_version(void);

/* ---- Normal/GMRF conjugate-Gibbs draw, tridiagonal precision -------------------------------
 * Replaces, for every chain at once, gmrf.sample_normal_canonical (gmrf.py:167-198):
 *   L = sparse_cholesky(Q) (gmrf.py:489-520), mu = cho_solve((L,True), b) (gmrf.py:437-462),
 *   x = mu + solve(L', z) (gmrf.py:29-61, 414-434),  z ~ N(0, I),
 * with Q and b assembled as in NormalNormal.sample (sampler.py:176-192).
 *   rhs_chain  [C][ld_rhs] optional per-chain addition to b (NULL = none)
 *   z_inject   [C][ld_z]   injected N(0,1) draws (NULL = generate, keyed by draw_index)
 *   x_out      [C][ld_x]   the draw
 *   mean_out   [C][ld_mean] optional mu = Q^{-1} b (NULL = skip)
 *   quad_out   [n_terms][C] optional quadratic forms around center[k] (NULL = skip)
 *   logdet_out [C]         optional log det Q_c = 2 sum log L_ii (NULL = skip)
 * Chains longer than one workgroup takes (n > 16 384) go through the segmented band kernels (omc_band_sample_canonical,
 * w = 1), and that route is stricter in two ways: z_inject must not alias x_out or mean_out (OMC_INVALID_ARG; the
 * workgroup-per-chain kernels accept z_inject == x_out), and the option "tridiag_quad_skip" is ignored -- every requested
 * quad_out row is computed (by omc_tridiag_quadform behind the draw).                                  */
omc_status omc_tridiag_sample_canonical(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms,
                                        const double* rhs_chain, int64_t ld_rhs,
                                        const double* z_inject, int64_t ld_z, uint64_t draw_index,
                                        double* x_out, int64_t ld_x,
                                        double* mean_out, int64_t ld_mean,
                                        double* quad_out, double* logdet_out);

/* ---- one whole Gibbs sweep of the "GMRF smoother" block structure in ONE launch ------------
 * Equivalent to the sampler list [NormalNormal(x), NormalGamma(scale_0), NormalGamma(scale_1), ...]
 * of MCMC.run_mcmc (mcmc.py:99-100) followed by the per-iteration bookkeeping (mcmc.py:105-108),
 * for a Gaussian block whose conditional precision is given by `terms`:
 *   1. x_c ~ N(Q_c^{-1} b_c, Q_c^{-1})                      (as omc_tridiag_sample_canonical)
 *   2. for every term k with blocks[k].enabled: scale[k][c] ~ Gamma(a0 + n_pos/2,
 *      rate = b0 + quad_k/2), written IN PLACE into terms->scale[k] (and to blocks[k].store)
 *      (as omc_normal_gamma_update with blocks[k].draw_index);
 *   3. log_post_out[c] = sum_k log N(. ; center_k, (scale_k M_k)^{-1}) + sum_{k enabled}
 *      log Gamma(scale_k; a0, b0) with the NEW scales (model.py:57-70), if log_post_out != NULL;
 *      needs blocks[k].logdet_unscaled = device scalar log det M_k for every term.
 * The draw goes straight to x_out, which may be a slab of the device-resident store.         */
typedef struct {
  int32_t enabled;                /* 0: scale[k] is a fixed input, no prior term in log_post   */
  double a0, b0;                  /* Gamma prior shape, rate (sampler.py:278-279)               */
  int64_t n_pos;                  /* #{diag(M_k) > 0}  (sampler.py:283)                         */
  const double* g_inject;         /* [C] injected Gamma(a,1) draws, or NULL                     */
  uint64_t draw_index;            /* keys this block's gamma stream when g_inject == NULL       */
  double* store;                  /* [C] optional copy of the new scale (store[param]), or NULL */
  const double* logdet_unscaled;  /* device scalar log det M_k; NULL allowed iff no log_post    */
} omc_gamma_block;

/* 1 if omc_tridiag_sample_canonical / omc_gmrf_sweep accept terms->center_chain for chains of n nodes under the
 * context's current options (the workgroup-per-chain form of the kernel), else 0.                            */
int32_t omc_tridiag_takes_center_chain(omc_ctx* ctx, int64_t n);

/* The log-posterior of a sweep in ONE launch: out[c] = host_const + sum over the pieces, in the order given, of
 *   kind 0  0.5 (n log s_c + mult * logdet - n log 2 pi - s_c quad_c)      Normal.log_p with a scalar x shared-matrix precision
 *           (location_scale.py:145-167 -> gmrf.py:321-348; quad from omc_*_quadform or a draw's fused form), s = 1 if scale NULL
 *   kind 1  log Gamma(x_c; shape, rate)                                     Gamma.log_p (distribution.py:241-261)
 * -- the same arithmetic, piece by piece and in the same order, as omc_scaled_gauss_logpdf / omc_gamma_logpdf called with
 * accumulate (model.py:57-70 sums the members one after the other).  At most OMC_LOGP_MAX pieces.                          */
#define OMC_LOGP_MAX 8
typedef struct {
  int32_t kind;
  double n;                /* kind 0: dimension x replicates                       */
  const double* scale;     /* kind 0: [C] or NULL                                  */
  const double* logdet;    /* kind 0: device scalar log det M                      */
  double logdet_mult;      /* kind 0: usually 1 (replicates: their number)         */
  const double* quad;      /* kind 0: [C]                                          */
  const double* x;         /* kind 1: [C]                                          */
  double shape, rate;      /* kind 1                                               */
} omc_logp_piece;
omc_status omc_log_post_sum(omc_ctx* ctx, int32_t n_pieces, const omc_logp_piece* pieces /* host */, double host_const,
                            double* out);

omc_status omc_gmrf_sweep(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms,
                          const omc_gamma_block* blocks /* [n_terms], host */,
                          const double* rhs_chain, int64_t ld_rhs,
                          const double* z_inject, int64_t ld_z, uint64_t draw_index,
                          double* x_out, int64_t ld_x, double* log_post_out);

/* The whole MCMC.run_mcmc loop (mcmc.py:97-111) for that sampler list, issued from C: (n_burn + n_iter) *
 * n_thin sweeps back to back with no host work in between.  Sweep t (0-based) uses draw index
 * draw_index0 + t * draws_per_sweep for the Gaussian block and that base + blocks[k].draw_index for
 * gamma block k.  Iteration i (the last sweep of every group of n_thin after the burn-in) is stored:
 *   x      -> x_store + slot * x_slot_stride        ([C][ld_x] slab; written directly by the sweep)
 *   scale  -> blocks[k].store + slot * C            (when blocks[k].store != NULL)
 *   log_p  -> log_post_store + slot * C             (when != NULL)
 * with slot = (first_slot + i) % n_slots.  A sweep that is not stored leaves its draw in scratch_x [C][ld_x] if it is
 * the last of the run (the state to continue from); the draws of other unstored sweeps may not be written at all.  */
omc_status omc_gmrf_run(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms, const omc_gamma_block* blocks,
                        int64_t n_burn, int64_t n_iter, int64_t n_thin, uint64_t draw_index0,
                        uint64_t draws_per_sweep, double* x_store, int64_t ld_x, int64_t x_slot_stride,
                        int64_t first_slot, int64_t n_slots, double* log_post_store, double* scratch_x);

/* quad_out[k][c] = (x_c - center_k)' M_k (x_c - center_k) for an existing x (sampler.py:276-284
 * when the Gaussian block was not just drawn; gmrf.py:343-344).                              */
omc_status omc_tridiag_quadform(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms,
                                const double* x, int64_t ld_x, double* quad_out);

/* Shared-vector helpers used once per model, not per sweep:
 *   out = M v for a shared tridiagonal M (rhs[k] = M_k m_k, sampler.py:183)
 *   *logdet [device, 1] = log det M via the same factorisation (gmrf.py:342), status latched */
omc_status omc_tridiag_matvec(omc_ctx* ctx, int64_t n, const double* diag, const double* off,
                              const double* v, double* out);
/* Per-chain vectors on the right-hand side (hierarchical models: a sampled prior mean, a sampled response):
 *   omc_tridiag_matvec_chain: out[c] (+)= scale[c] * M v_c -- the term Q_rsp @ mean of sampler.py:181-183 when the mean is
 *     itself per chain, and W (y_c - d) of sampler.py:190-192 when the response is (M tridiagonal; NULL diag = identity);
 *   omc_chain_lincomb: out[c] = a x_c + b y_c (y per chain, or shared with ld_y = 0): the residual of location_scale.py:
 *     153-160 when both sides of a Normal are per chain; b == 0 means y is not read (an inf or NaN there does not reach out). */
omc_status omc_tridiag_matvec_chain(omc_ctx* ctx, int64_t n, const double* diag, const double* off, const double* v, int64_t ld_v,
                                    const double* scale, double* out, int64_t ld_out, int32_t accumulate);
omc_status omc_chain_lincomb(omc_ctx* ctx, int64_t n, double a, const double* x, int64_t ld_x, double b, const double* y, int64_t ld_y,
                             double* out, int64_t ld_out);
/* dst[c][0..n) = src[c][0..n) for every chain: the current value of a parameter into its slab of the sample store
 * (sampler/sampler.py:89-118, `store[param][:, [iteration]] = state[param]`) -- a copy kernel on the context's stream (the
 * runtime's device-to-device memcpy moved the 82 MB of a 10 000 x 1024 state at 0.26 TB/s).                        */
omc_status omc_chain_copy(omc_ctx* ctx, int64_t n, const double* src, int64_t ld_src, double* dst, int64_t ld_dst);
omc_status omc_tridiag_logdet(omc_ctx* ctx, int64_t n, const double* diag, const double* off,
                              double* logdet);

/* ---- Normal/Normal conjugate-Gibbs draw, DENSE conditional precision ------------------------
 * The same update as omc_tridiag_sample_canonical for Q_c = sum_k scale[k][c] * M_k with M_k shared
 * dense symmetric p x p matrices (NULL = identity): the regression case of NormalNormal.sample where
 * the likelihood Hessian is the Gram matrix G = A' W A (sampler.py:185-186 -> location_scale.py:
 * 238-241) and b = P m + A' W (y - d) (sampler.py:183,190-192).  Per chain: dense Cholesky
 * (gmrf.py:481 np.linalg.cholesky = LAPACK potrf; here rocSOLVER potrf_strided_batched),
 * mu = cho_solve (gmrf.py:462), x = mu + L^{-T} z (gmrf.py:434; the reference calls a general LU
 * solve there, the result is the same triangular solve).  Factors live in a workspace owned by
 * the context (C * p * p doubles, allocated on first use).                                      */
typedef struct {
  int32_t n_terms;                      /* 1..OMC_MAX_TERMS                                  */
  const double* mat[OMC_MAX_TERMS];     /* [p*p] shared symmetric; NULL = identity           */
  const double* rhs[OMC_MAX_TERMS];     /* [p]   shared; NULL = zeros                        */
  const double* scale[OMC_MAX_TERMS];   /* [C]   per-chain scalar; NULL = 1                  */
  const double* diag_chain;             /* [C][p] per-chain diagonal added to Q_c (a mixture prior precision
                                           diag(prec[allocation]), parameter.py:501); NULL = none           */
} omc_dense_terms;

omc_status omc_dense_sample_canonical(omc_ctx* ctx, int64_t p, const omc_dense_terms* terms,
                                      const double* rhs_chain, int64_t ld_rhs,
                                      const double* z_inject, int64_t ld_z, uint64_t draw_index,
                                      double* x_out, int64_t ld_x,
                                      double* mean_out, int64_t ld_mean, double* logdet_out);

/* Spectral route of the same conditional draw for Q_c = a_c I + b_c M with ONE shared symmetric M -- the regression
 * likelihood's Gram matrix next to a scalar prior precision (examples/3, BASELINE configs[1]), where only two scalars
 * differ from chain to chain and from sweep to sweep.  M = V diag(ev) V' once per model (omc_dense_spectral_prepare,
 * rocSOLVER dsyevd), then  mu_c = V diag(1/(a_c + b_c ev)) V' b_c  and  x_c = mu_c + V diag((a_c + b_c ev)^-1/2) z_c:
 * two (three with mean_out) GEMMs over all chains and one scaling kernel instead of C factorisations of order p
 * (p^3/3 flop per chain).  mu_c and log det Q_c = sum_i log(a_c + b_c ev_i) are the reference's values
 * (gmrf.py:196, 339) up to rounding; x_c is a draw from the same N(mu_c, Q_c^-1) but not the reference's
 * path-wise image of z (that is L^-T z, gmrf.py:61): replays with injected draws use omc_dense_sample_canonical.
 *   terms: as above; terms->mat[k_mat] is M, every other term a scaled identity (mat NULL), no diag_chain.       */
omc_status omc_dense_spectral_prepare(omc_ctx* ctx, int64_t p, const double* M, double* V_out, double* ev_out);
omc_status omc_dense_spectral_sample(omc_ctx* ctx, int64_t p, const omc_dense_terms* terms, int32_t k_mat, const double* V,
                                     const double* ev, const double* rhs_chain, int64_t ld_rhs, const double* z_inject,
                                     int64_t ld_z, uint64_t draw_index, double* x_out, int64_t ld_x, double* mean_out,
                                     int64_t ld_mean, double* logdet_out);

/* Design-matrix helpers; X is [n][p] row-major (a NumPy array as is), shared by all chains; w is an
 * optional [n] diagonal weight (NULL = ones).
 *   omc_gram:            G[p][p] = X' diag(w) X                 (location_scale.py:238-241; fp64 MFMA GEMM)
 *   omc_design_rhs:      out[p]  = X' diag(w) y                 (sampler.py:192)
 *   omc_design_predict:  fitted[c][:] = X beta_c                (parameter.py:196; one GEMM for all chains)
 *   omc_weighted_resid_sq: out[c] = (y - f_c)' diag(w) (y - f_c) (sampler.py:276,284)              */
omc_status omc_gram(omc_ctx* ctx, int64_t n, int64_t p, const double* X, const double* w, double* G_out);
omc_status omc_design_rhs(omc_ctx* ctx, int64_t n, int64_t p, const double* X, const double* w,
                          const double* y, double* out);
omc_status omc_design_predict(omc_ctx* ctx, int64_t n, int64_t p, const double* X,
                              const double* beta, int64_t ld_beta, double* fitted, int64_t ld_fitted);
omc_status omc_weighted_resid_sq(omc_ctx* ctx, int64_t n, const double* y, const double* fitted,
                                 int64_t ld_fitted, const double* w, double* out);

/* ---- Metropolis-Hastings on a Gaussian target with a shared constant Hessian -----------------
 * Target Normal("x", mean=mu, precision=Q) with Q dense d x d and mu shared by all chains (cfg4).
 *
 * omc_dense_cholesky: L = chol(scale * A) (lower, LAPACK potrf as gmrf.py:481), *sumlogdiag =
 *   sum_i log L_ii (device scalar); used once per model for L = chol(H / step^2)
 *   (metropolis_hastings.py:345-346) and L_Q = chol(Q) (gmrf.py:339).
 *
 * omc_mala_step: one ManifoldMALA.sample for every chain (metropolis_hastings.py:102-125, 301-373):
 *   g = -Q (x - mu), H = Q (location_scale.py:222-232); m = x + 1/2 (H/step^2)^{-1} g (347);
 *   x' = m + L^{-T} z (317 -> gmrf.py:61); log q = sum log L_ii - 1/2 |L'(. - m)|^2 forward and
 *   reverse (350-373); log alpha = lp' + q_rev - lp - q_fwd (155); accept iff log u < log alpha (173).
 *   As level-3 BLAS on the d x C state matrix.  Because H is constant, the drift matrix
 *   -(H/step^2)^{-1} Q and L^{-T} are formed once per (Q, L, step) (cached in the context, so Q and L
 *   must stay unchanged while they are in use); a step is then 4 GEMMs (the product with L' runs on a copy
 *   of L with an explicitly zero upper triangle).
 * omc_rw_step: one untruncated RandomWalk.sample (metropolis_hastings.py:212-269): x' = x + step z.
 *   The zero-padded copy of LQ is cached in the context the same way (LQ must stay unchanged while in use).
 *   z_inject [C][ld_z] / u_inject [C]: injected N(0,1) / U(0,1) draws (NULL = generate);
 *   accept_count / proposal_count [C] int64 (NULL = not kept): the AcceptRate counters
 *   (metropolis_hastings.py:25-66), INT path: bit-exact.                                        */
omc_status omc_dense_cholesky(omc_ctx* ctx, int64_t d, const double* A, double scale, double* L_out,
                              double* sumlogdiag_out);
omc_status omc_mala_step(omc_ctx* ctx, int64_t d, const double* Q, const double* mu, const double* L,
                         const double* sumlogL, double step, const double* z_inject, int64_t ld_z,
                         const double* u_inject, uint64_t draw_index, double* x, int64_t ld_x,
                         int64_t* accept_count, int64_t* proposal_count);
/* The same step when L = chol(Q / step^2) (what ManifoldMALA factorises for a Gaussian target: H = Q), in whitened
 * coordinates a = L'(x - mu): the drift -(L L')^{-1} Q is then -step^2 I, so
 *   a' = (1 - step^2/2) a + z;  |L'(x-mu)|^2 = |a|^2, |L'(x'-mu)|^2 = |a'|^2, |L'(x'-m)|^2 = |z|^2,
 *   |L'(x-m')|^2 = |a - (1 - step^2/2) a'|^2  -- every term of log alpha (metropolis_hastings.py:155, 350-373) element-wise,
 * and only accepted proposals are mapped back, x = mu + L^{-T} a' (one triangular product; rejected chains keep their x
 * bit for bit).  Same draws and decision rule as omc_mala_step; results agree to the rounding of the two evaluation
 * orders.  The context keeps a for the state at x: state_is_current != 0 promises that x has not been written by anyone
 * else since this function last returned for the same x (otherwise a = L'(x - mu) is recomputed: one more product).
 * L and mu must stay unchanged while in use (omc_mh_invalidate drops what was derived from them).                */
/* log_p_out (both whitened steps; NULL = not wanted): [C] the log density of the target at the state the step leaves behind
 * -- what Model.log_p would evaluate for the one-Normal model of this route (mcmc.py:99-111), taken from the quantities the
 * acceptance test has formed anyway.                                                                                  */
omc_status omc_mala_step_white(omc_ctx* ctx, int64_t d, const double* mu, const double* L, const double* sumlogL, double step,
                               const double* z_inject, int64_t ld_z, const double* u_inject, uint64_t draw_index, double* x,
                               int64_t ld_x, int32_t state_is_current, int64_t* accept_count, int64_t* proposal_count, double* log_p_out);
/* n_steps of omc_mala_step_white in a row -- the loop of MCMC.run_mcmc over a one-sampler model (mcmc.py:97-111) -- issued
 * as ONE launch per block of 32 steps: a chain's whitened state stays in registers for the block, and one triangular
 * product per block maps the block's states back into the store.  Step t draws from stream draw_index0 + t * draw_stride
 * (what the loop would pass as draw_index); results are bit-identical to n_steps single calls (state, counters, log_p).
 *   z_inject [n_steps][C][ld_z], u_inject [n_steps][C]: injected draws (NULL = generate);
 *   x_store    [n_steps][C][d] or NULL: the state after every step (sampler.store of every iteration, sampler.py:89-118);
 *   logp_store [n_steps][C]    or NULL: the target's log density at those states (mcmc.py:108);
 *   x: the current state on entry (see state_is_current), the state after the last step on return; log_p_out [C] or NULL.
 * d <= 2048 (OMC_UNSUPPORTED beyond: use the single step).                                                            */
omc_status omc_mala_run_white(omc_ctx* ctx, int64_t d, const double* mu, const double* L, const double* sumlogL, double step,
                              const double* z_inject, int64_t ld_z, const double* u_inject, uint64_t draw_index0,
                              uint64_t draw_stride, int64_t n_steps, double* x, int64_t ld_x, int32_t state_is_current,
                              double* x_store, double* logp_store, int64_t* accept_count, int64_t* proposal_count, double* log_p_out);
omc_status omc_rw_step(omc_ctx* ctx, int64_t d, const double* mu, const double* LQ, const double* sumlogLQ,
                       double step, const double* z_inject, int64_t ld_z, const double* u_inject,
                       uint64_t draw_index, double* x, int64_t ld_x, int64_t* accept_count,
                       int64_t* proposal_count);
/* The same random-walk step with a = L_Q'(x - mu) carried by the context (LQ = chol(Q)): a' = a + step L_Q' z,
 * log p(x') - log p(x) = (|a|^2 - |a'|^2)/2 -- one triangular product per step instead of two; x' = x + step z is
 * formed exactly as in omc_rw_step.  state_is_current as for omc_mala_step_white.                                  */
omc_status omc_rw_step_white(omc_ctx* ctx, int64_t d, const double* mu, const double* LQ, const double* sumlogLQ, double step,
                             const double* z_inject, int64_t ld_z, const double* u_inject, uint64_t draw_index, double* x,
                             int64_t ld_x, int32_t state_is_current, int64_t* accept_count, int64_t* proposal_count, double* log_p_out);
/* omc_mh_invalidate: drops the matrices cached for the last (Q, L, step) / LQ.  The cache is keyed by device
 * addresses; call this whenever Q, L or LQ were rewritten in place or re-created (a new buffer may land on a recycled
 * address).  The reference has nothing to invalidate: it refactorises every step (metropolis_hastings.py:345-346). */
omc_status omc_mh_invalidate(omc_ctx* ctx);

/* ---- Normal-Gamma conjugate update ---------------------------------------------------------
 * NormalGamma.sample (sampler.py:252-288) for a scalar precision per chain:
 *   a = a0 + n_pos/2, b = b0 + quad[c]/2, out[c] = Gamma(a, scale = 1/b); b == 0 -> scale inf.
 *   g_inject [C] injected standard-gamma draws Gamma(a,1) (NULL = generate, Marsaglia-Tsang). */
omc_status omc_normal_gamma_update(omc_ctx* ctx, double a0, double b0, int64_t n_pos,
                                   const double* quad, const double* g_inject,
                                   uint64_t draw_index, double* out);

/* ---- log-density pieces of Model.log_p (model.py:57-70) ------------------------------------
 * Gaussian with precision scale[c]*M, M shared (location_scale.py:145-167 -> gmrf.py:321-348):
 *   lp = 0.5*(n*log(scale[c]) + logdet_M - n*log(2 pi) - scale[c]*quad[c])
 * Gamma prior (distribution.py:241-261): lp = a log b - lgamma(a) + (a-1) log x - b x.
 * accumulate != 0 adds into out[c] instead of overwriting.                                    */
omc_status omc_scaled_gauss_logpdf(omc_ctx* ctx, int64_t n, const double* scale,
                                   const double* logdet_unscaled /* device, 1 */,
                                   const double* quad, double* out, int32_t accumulate);
omc_status omc_gamma_logpdf(omc_ctx* ctx, const double* x, double shape, double rate,
                            double* out, int32_t accumulate);

/* ---- reversible-jump move bookkeeping (INT / index path: bit-exact) ---------------------------
 * ReversibleJump.get_move_type, get_move_probabilities and the deletion index of death_proposal
 * (reversible_jump.py:310-373, 173) for every chain:
 *   n[c] == n_max -> death; n[c] == 1 -> birth; otherwise birth iff u <= birth_probability, where the
 *   uniform is consumed ONLY in that last case (as in the reference);
 *   p_birth/p_death with the edge cases at n_max, n_max-1, 1, 2;
 *   del_index[c] = randint(0, n[c]) for a death, -1 for a birth.
 *   u_inject [C] / idx_inject [C] (int64): injected draws (NULL = Philox: uniform from block 0,
 *   unbiased bounded integer by Lemire's rejection from block 1).  Any n[c] < 1 or > n_max is
 *   latched as a failure of that chain (the reference raises ValueError for n == 0).           */
omc_status omc_rj_move(omc_ctx* ctx, int64_t n_max, double birth_probability, const int64_t* n,
                       const double* u_inject, const int64_t* idx_inject, uint64_t draw_index,
                       int32_t* birth_out, double* p_birth_out, double* p_death_out, int64_t* del_index_out);
/* The same move with the outputs ReversibleJump.proposal makes of it (reversible_jump.py:122-144, 165-191): count [C] is
 * the float64 count the state holds; count_prop = count +- 1; with d = density_const + density_chain[c] (the log prior
 * density of the last element of every associated parameter, :132,143; density_chain may be NULL)
 *   birth: lq_fwd = log p_birth + d, lq_rev = log p_death;   death: lq_fwd = log p_death, lq_rev = log p_birth + d.  */
omc_status omc_rj_move_densities(omc_ctx* ctx, int64_t n_max, double birth_probability, const double* count,
                                 const double* u_inject, const int64_t* idx_inject, uint64_t draw_index,
                                 const double* density_chain, double density_const, int32_t* birth_out,
                                 int64_t* del_index_out, double* count_prop_out, double* lq_fwd_out, double* lq_rev_out);

/* One draw of a variable-size parameter into its store slab (sampler.py:112-116): dst[c][j] = src[c][j] for
 * j < count[c], NaN beyond, j < width; chain c at src + c*src_chain_stride / dst + c*dst_chain_stride.          */
omc_status omc_store_ragged(omc_ctx* ctx, int64_t width, const double* src, int64_t src_chain_stride, const double* count,
                            double* dst, int64_t dst_chain_stride);

/* ---- banded precisions of any bandwidth (RW2, seasonal, lattice GMRFs) --------
 * The same conditional draw as omc_tridiag_sample_canonical for Q_c = sum_k scale[k][c] * M_k with every M_k
 * symmetric of bandwidth bw[k] <= w (w <= 128), factorised in natural order (the reference's unpermuted SuperLU /
 * LAPACK route, gmrf.py:489-520, so the draw matches path-wise for the same z):
 *   band[k]  [(bw[k]+1) x n], band[k][d*n + i] = M_k[i+d, i] (d-th sub-diagonal contiguous, as scipy's
 *            M.diagonal(-d) padded to n); NULL = identity (bw 0);
 *   rhs[k]   [n] shared M_k m_k or NULL;  scale[k] [C] or NULL = 1;
 *   x = Q^{-1} b + L^{-T} z; mean (optional) = Q^{-1} b; logdet (optional) [C] = log det Q_c.
 * x, mean and z_inject are three different arrays (OMC_INVALID_ARG otherwise: the narrow-band route writes rows of
 * x while other rows' draws are still being read).
 * Uses a per-context workspace of C * n * (w+1) doubles for the factor (twice that on the segmented route).   */
typedef struct {
  int32_t n_terms;
  const double* band[OMC_MAX_TERMS];
  int32_t bw[OMC_MAX_TERMS];
  const double* rhs[OMC_MAX_TERMS];
  const double* scale[OMC_MAX_TERMS];
} omc_band_terms;

omc_status omc_band_sample_canonical(omc_ctx* ctx, int64_t n, int64_t w, const omc_band_terms* terms,
                                     const double* rhs_chain, int64_t ld_rhs, const double* z_inject, int64_t ld_z,
                                     uint64_t draw_index, double* x, int64_t ld_x, double* mean, int64_t ld_mean,
                                     double* logdet);
/* quad[c] = (x_c - center)' M (x_c - center) for one shared band matrix (NormalGamma.sample sampler.py:276,284;
 * Normal.log_p gmrf.py:343-344); band NULL with w = 0 is the identity, center NULL = 0.                      */
omc_status omc_band_quadform(omc_ctx* ctx, int64_t n, int64_t w, const double* band, const double* center,
                             const double* x, int64_t ld, double* quad);
/* out[c] (+)= scale[c] * M v_c for a shared band matrix and per-chain vectors: Q_rsp @ mean with a sampled prior mean
 * (sampler.py:181-183) and W (y_c - d) with a sampled response (sampler.py:190-192) on the band route; band NULL with
 * w = 0 is the identity, scale NULL = 1.                                                                          */
omc_status omc_band_matvec_chain(omc_ctx* ctx, int64_t n, int64_t w, const double* band, const double* v, int64_t ld_v,
                                 const double* scale, double* out, int64_t ld_out, int32_t accumulate);

/* ---- truncated Gaussian full conditional ------------------------------------
 * gmrf.gibbs_canonical_truncated_normal (gmrf.py:201-266), the branch NormalNormal.sample takes when the
 * parameter's prior has domain limits (sampler.py:199-205): ONE scan of single-site updates in index order,
 *   x_i ~ N_[lower_i, upper_i]( (b_i - sum_j Q_ij x_j + Q_ii x_i) / Q_ii, 1/Q_ii ),  x_j already updated for j < i,
 * each by inverse CDF of a uniform (gmrf.py:264 -> scipy truncnorm.rvs == ppf(U)).  x is updated in place
 * (the scan starts from the current state).  Q and b in the same "shared structure x per-chain scalar" forms as
 * the untruncated entry points; lower / upper: device [n] (NULL = unbounded on that side).
 * u_inject [C][ld_u] injected uniforms; in-kernel uniforms are Philox blocks i/2 of (seed, chain, draw_index).
 * The scan is sequential in i by definition, the parallelism is over chains (tridiagonal: one lane per chain;
 * dense: one wave per chain, p <= 8192).                                                                    */
omc_status omc_tridiag_gibbs_truncated(omc_ctx* ctx, int64_t n, const omc_tridiag_terms* terms,
                                       const double* rhs_chain, int64_t ld_rhs, const double* lower,
                                       const double* upper, const double* u_inject, int64_t ld_u,
                                       uint64_t draw_index, double* x, int64_t ld_x);
/* the same scan for a banded precision (band storage and terms of omc_band_sample_canonical), one lane per chain */
omc_status omc_band_gibbs_truncated(omc_ctx* ctx, int64_t n, int64_t w, const omc_band_terms* terms, const double* rhs_chain,
                                    int64_t ld_rhs, const double* lower, const double* upper, const double* u_inject, int64_t ld_u,
                                    uint64_t draw_index, double* x, int64_t ld_x);
omc_status omc_dense_gibbs_truncated(omc_ctx* ctx, int64_t p, const omc_dense_terms* terms,
                                     const double* rhs_chain, int64_t ld_rhs, const double* lower,
                                     const double* upper, const double* u_inject, int64_t ld_u,
                                     uint64_t draw_index, double* x, int64_t ld_x);
/* Normal.log_p returns -inf when the response lies outside [lower, upper] (location_scale.py:162-188):
 *   out[c] = -inf for every chain with an element of x[c] below lower or above upper; other chains untouched. */
omc_status omc_domain_penalty(omc_ctx* ctx, int64_t n, const double* x, int64_t ld, const double* lower,
                              const double* upper, double* out);

/* ---- generic Metropolis-Hastings blocks, ragged state, reversible-jump transitions ----------------
 * Ragged parameters (dimension changes under reversible jump: knots, their coefficients, the basis)
 * are held padded to n_max with zeros; count[c] -- the float64 the reference keeps in
 * state["n_basis"] -- is the number of live leading entries of chain c.  `count`/`index` pairs below
 * gate a chain: it takes part iff count == NULL or index < count[c] (RandomWalkLoop over knots:
 * metropolis_hastings.py:286-288 runs range(n_basis) steps, chains with fewer knots sit the step out).
 * `sub` is the first Philox block the call may use within (seed, chain, draw_index): lets one sampler
 * call make several independent draws.
 *
 * omc_rw_propose: RandomWalk.proposal (metropolis_hastings.py:212-269) for p elements per chain,
 *   x[c*x_chain_stride + e*x_elem_stride] -> z[c*z_chain_stride + e*z_elem_stride]:
 *   lower/upper == NULL: z = x + step*N(0,1), lq_fwd = lq_rev = 0 (symmetric, :250-251);
 *   else truncated normal on [lower[e], upper[e]] (device, p each) by inverse CDF of a uniform
 *   (gmrf.py:269-292 -> scipy truncnorm.rvs == ppf(U)), lq_fwd = sum_e log q(z|x), lq_rev = sum_e log q(x|z)
 *   (:255-257 -> gmrf.py:295-318).  step[e*step_elem_stride] device (stride 0 = scalar step).
 *   draw_inject [C][p]: the uniforms (truncated) or normals (untruncated) to use instead of Philox.   */
omc_status omc_rw_propose(omc_ctx* ctx, int64_t p, const double* x, int64_t x_chain_stride,
                          int64_t x_elem_stride, const double* step, int64_t step_elem_stride,
                          const double* lower, const double* upper, const double* count, int64_t index,
                          const double* draw_inject, uint64_t draw_index, uint32_t sub, double* z,
                          int64_t z_chain_stride, int64_t z_elem_stride, double* lq_fwd, double* lq_rev);

/* ManifoldMALA.proposal / _proposal_params / _log_proposal_density (metropolis_hastings.py:301-373) when the Hessian
 * is DIAGONAL per chain (hdiag [C][kmax], e.g. the mixture-Normal prior of a variable-size coefficient vector):
 *   precision = h/step^2, L_jj = sqrt(h_j)/step, m = x + (1/2) precision^{-1} grad;
 *   propose != 0: x_other = m + z / L is WRITTEN (zeros beyond count[c]), lq = log q(x_other | x);
 *   propose == 0: x_other is READ, lq = log q(x_other | x) (the reverse move, with grad / hdiag of the proposed state
 *   passed as x's); log q = sum_j log L_jj - |L'(x_other - m)|^2 / 2.  z_inject [C][kmax] standard normals.      */
omc_status omc_mala_diag(omc_ctx* ctx, int64_t kmax, const double* x, const double* grad, const double* hdiag,
                         const double* count, double step, int32_t propose, const double* z_inject, uint64_t draw_index,
                         uint32_t sub, double* x_other, double* lq);

/* MetropolisHastings._accept_reject_proposal + accept_proposal (metropolis_hastings.py:127-173):
 *   log_alpha = lp_prop + lq_rev - (lp_cur + lq_fwd); accept[c] = log(u) < log_alpha (NaN rejects);
 *   counters (int64, per chain) incremented for participating chains; lq_* / log_alpha / counters may be NULL. */
omc_status omc_mh_accept(omc_ctx* ctx, const double* lp_cur, const double* lp_prop, const double* lq_fwd,
                         const double* lq_rev, const double* count, int64_t index, const double* u_inject,
                         uint64_t draw_index, uint32_t sub, int32_t* accept, double* log_alpha,
                         int64_t* accept_count, int64_t* proposal_count);

/* current_state = prop_state for accepted chains (metropolis_hastings.py:157-159):
 *   dst[c][0..width) = src[c][0..width) where accept[c] != 0.                                          */
omc_status omc_chain_select(omc_ctx* ctx, const int32_t* accept, int64_t width, const double* src, double* dst);
/* the same for n_items (<= OMC_SELECT_MAX) entries in one launch: entry e is widths[e] doubles per chain, chain-major,
 * copied from srcs[e] to dsts[e] on the accepted chains (host arrays of n_items elements).                      */
omc_status omc_chain_select_multi(omc_ctx* ctx, const int32_t* accept, int32_t n_items, const int64_t* widths,
                                  const double* const* srcs, double* const* dsts);

/* np.concatenate / np.delete along the ragged axis (reversible_jump.py:131, 175) for every chain:
 *   birth[c] != 0: dst = src with new_vals[c][0..rows) appended at position count[c] (NULL = zeros);
 *   birth[c] == 0: dst = src with entry del_index[c] removed, later entries shifted down.
 *   element (c, r, j) lives at base[c*chain_stride + r*row_stride + j*col_stride], j < kmax the ragged axis;
 *   dst is fully written (zeros beyond the new count); src and dst must not alias.                     */
omc_status omc_ragged_resize(omc_ctx* ctx, int64_t rows, int64_t kmax, const double* count, const int32_t* birth,
                             const int64_t* del_index, const double* new_vals, const double* src, double* dst,
                             int64_t chain_stride, int64_t row_stride, int64_t col_stride);

/* Gaussian-kernel basis on per-chain knots, the design matrix of the reference's reversible-jump example
 * (tests/test_reversible_jump.py:24-40: norm.pdf(X, loc=knot, scale=scale) per column):
 *   B[c][j][i] = exp(-t^2/2) / sqrt(2 pi) / s,  t = (X[i] - knots[c][j]) / s,  s = scales[c][j] or scale0 (scales NULL)
 *   for j < count[c], 0 beyond.  column >= 0 rewrites only that column (one knot moved), -1 all of them.
 *   prev_count (C, or NULL; needs count): B already holds zeros in the columns >= prev_count[c] of chain c (a buffer
 *   this function filled before with that count) -- only columns < max(count[c], prev_count[c]) are written.    */
omc_status omc_gaussian_basis(omc_ctx* ctx, int64_t n, int64_t kmax, const double* X, const double* knots,
                              const double* scales, double scale0, const double* count, const double* prev_count,
                              int64_t column, double* B);

/* RandomWalkLoop over the knots of that basis under a regression likelihood, every knot of every chain in one launch
 * (metropolis_hastings.py:276-289 -> :212-269 proposal, :127-173 accept/reject) for
 *   y ~ N(B_c beta_c + add_chain[c] + add_shared, (tau[c] diag(w))^-1),  knots a priori uniform on [lower, upper]:
 *   for k < count[c], in order: z ~ truncated N(theta[c][k], step^2) on [lower, upper]; accepted with probability
 *   min(1, exp(dloglik + log q(theta|z) - log q(z|theta))); an accepted move writes theta[c][k] = z and column k of B_c.
 *   The same draws as the launch-by-launch route: stream (draw_index, uniform), Philox block 2k for the proposal of knot k
 *   and 2k+1 for its accept uniform; inject_z / inject_u (kmax x C, [k][c]) replace them (tests).  w, tau, add_* may be
 *   NULL (ones / zeros).  accept_count / proposal_count (C) are incremented; accept_out / log_alpha_out (kmax x C,
 *   [k][c], entries k < count[c] written) may be NULL.  The chain's residual lives in registers: n <= 10 240.       */
omc_status omc_knot_loop(omc_ctx* ctx, int64_t n, int64_t kmax, const double* X, double scale, const double* y,
                         const double* add_shared, const double* add_chain, const double* w, const double* tau,
                         const double* beta, double* theta, const double* count, double* B, double step, double lower,
                         double upper, const double* inject_z, const double* inject_u, uint64_t draw_index,
                         int64_t* accept_count, int64_t* proposal_count, int32_t* accept_out, double* log_alpha_out);

/* Per-chain design matrices B_c (n x kmax; column j of chain c contiguous at B[(c*kmax + j)*n]):
 *   omc_design_predict_batched: out[c] = chain_scale[c] * (alpha * B_c coef_c + add_chain[c] + add_shared)
 *     (LinearCombination.predictor / predictor_conditional, parameter.py:162-197, for a basis that depends on
 *     per-chain knots; with alpha = -1, chain_scale = tau it is the per-chain part of b = A'W(y - d), sampler.py:192);
 *     add_chain [C][n], add_shared [n], chain_scale [C] may be NULL;
 *   omc_design_gram_batched:    gram[c] = B_c' diag(w) B_c (kmax x kmax, row-major),
 *     rhs[c] = B_c' diag(w) (resid_shared - resid_chain[c])   (location_scale.py:238-241, sampler.py:192);
 *     w [n] shared or NULL = ones; resid_* / rhs may be NULL; count [C] (NULL = kmax): only the leading count[c]
 *     columns are live, the rest of gram / rhs is written as 0.  kmax <= 36.
 *   omc_design_resid_sq_batched: out[c] = sum_i w[i] (y[i] - (B_c coef_c + add_chain[c] + add_shared)[i])^2, the
 *     quadratic form of Normal.log_p for a regression on a per-chain basis (gmrf.py:343-344 on the residual of
 *     parameter.py:162-197) without materialising the fitted values; y [n] shared; w, add_chain, add_shared may be
 *     NULL.  Rows are summed in a fixed order (parts, then an ordered sum of the parts).                        */
omc_status omc_design_predict_batched(omc_ctx* ctx, int64_t n, int64_t kmax, const double* B, const double* coef,
                                      const double* add_chain, const double* add_shared, double alpha,
                                      const double* chain_scale, double* out);
omc_status omc_design_resid_sq_batched(omc_ctx* ctx, int64_t n, int64_t kmax, const double* B, const double* coef,
                                       const double* add_chain, const double* add_shared, const double* y,
                                       const double* w, double* out);
omc_status omc_design_gram_batched(omc_ctx* ctx, int64_t n, int64_t kmax, const double* B, const double* w,
                                   const double* resid_shared, const double* resid_chain, const double* count,
                                   double* gram, double* rhs);
/* The same Gram matrix with the basis picked per chain: chain c uses (B_alt, count_alt) where select[c] != 0, (B, count)
 * otherwise -- the matched reversible-jump transition (reversible_jump.py:240-242, 290-292) needs X'X of the LARGER of
 * the current and the proposed basis only (proposed for a birth, current for a death).                          */
omc_status omc_design_gram_select(omc_ctx* ctx, int64_t n, int64_t kmax, const double* B, const double* count,
                                  const double* B_alt, const double* count_alt, const int32_t* select, const double* w,
                                  double* gram);

/* NormalNormal.sample (sampler.py:176-197 -> gmrf.py:167-198) for a small ragged parameter:
 *   Q_c = diag(prior_prec[c]) + lik_scale[c]*gram[c],  b_c = prior_prec[c]*prior_mean[c] + lik_scale[c]*gram_rhs[c]
 *   on the live count[c] x count[c] block; x = Q^{-1}b + L^{-T}z, natural-order Cholesky; entries beyond
 *   count[c] are written as 0.  kmax <= 64.  z_inject [C][kmax]; prior_mean / lik_scale / count / mu may be NULL. */
omc_status omc_small_sample_canonical(omc_ctx* ctx, int64_t kmax, const double* gram, const double* gram_rhs,
                                      const double* lik_scale, const double* prior_prec, const double* prior_mean,
                                      const double* count, const double* z_inject, uint64_t draw_index,
                                      double* x, double* mu);

/* Per-chain small SPD matrices A [C][k][k] (k <= 64): Av_out[c] = A_c v_c, quad_out[c] = v_c' A_c v_c, logdet_out[c] =
 * log det A_c by the natural-order Cholesky (a non-positive pivot latches the chain); any output may be NULL.  The pieces
 * of ManifoldMALA's proposal for a Hessian that depends on the parameter (metropolis_hastings.py:325-373): Lambda x,
 * (. - m)' Lambda (. - m) and sum log L_ii with Lambda_c = H_c / step^2; the solve itself is omc_small_sample_canonical. */
omc_status omc_small_spd_ops(omc_ctx* ctx, int64_t k, const double* A, const double* v, double* Av_out, double* quad_out,
                             double* logdet_out);

/* ReversibleJump.matched_birth_transition / matched_death_transition (reversible_jump.py:195-308):
 *   with X the larger of the two bases (the proposed one for a birth, the current one for a death),
 *   G = (X'X + 1e-10 I)^{-1} X'X_small by LU with partial pivoting (np.linalg.solve);
 *   birth: coef* = G coef, last entry ~ truncated N(coef*_last, scale) on [lim_lo, lim_hi] (has_limits) or
 *          N(coef*_last, scale^2); lq_fwd += its log-density; lq_rev += log det [G | e_last];
 *   death: F = G with e_idx inserted at column idx; coef_aug = F^{-1} coef; coef* = coef_aug without idx;
 *          lq_fwd += log det F; lq_rev += log-density of coef_aug[idx] under (truncated) N(0, scale).
 *   gram_cur / gram_prop [C][kmax][kmax]: Gram matrices of the current and proposed bases (the products
 *   X'X_small are sub-blocks of the larger one).  draw_inject [C]: uniform (limits) or normal draw.      */
omc_status omc_rj_matched_transition(omc_ctx* ctx, int64_t kmax, const double* gram_cur, const double* gram_prop,
                                     const double* count, const int32_t* birth, const int64_t* del_index,
                                     const double* coef_cur, double scale, int32_t has_limits, double lim_lo,
                                     double lim_hi, const double* draw_inject, uint64_t draw_index, uint32_t sub,
                                     double* coef_prop, double* lq_fwd, double* lq_rev);

/* Log-density pieces for ragged parameters (accumulate != 0 adds into out[c]):
 *   omc_diag_gauss_logpdf: Normal with diagonal precision (MixtureParameterMatrix, parameter.py:474-538):
 *     0.5*(sum_j log prec_j - k log 2pi - sum_j prec_j (x_j - mean_j)^2) over the k = count[c] live entries;
 *   omc_poisson_logpmf:    Poisson.log_p (distribution.py:490-508): k log(rate) - rate - lgamma(k+1), k = x[c];
 *   omc_count_logpdf:      out = per_element * count[c]  (Uniform.log_p, distribution.py:422-442: every live
 *     replicate contributes -sum log(upper - lower));
 *   omc_mixture_gather:    MixtureParameterVector.predictor (parameter.py:447): out[c][j] = param[alloc[c][j]]
 *     for live j, `fill` beyond (param [m] shared with param_stride 0, or per chain [C][m] with param_stride m;
 *     alloc holds integer values as float64).                                                              */
/* LogNormal.log_p (location_scale.py:279-300) = Normal.log_p at log(response) minus sum log(response):
 *   out[c][i] = log x[c][i], sumlog[c] = sum_i log x[c][i] (sumlog may be NULL).                              */
omc_status omc_log_transform(omc_ctx* ctx, int64_t n, const double* x, int64_t ld_x, double* out, int64_t ld_o,
                             double* sumlog);

/* out[c] = sum_i (a[c][i] - center_a[i]) * (b[c][i] - center_b[i]).  With b = M x from omc_design_predict (one GEMM over
 * all chains) and center_b = M m this is the quadratic form (x - m)' M (x - m) of a DENSE shared precision: the
 * sufficient statistic of NormalGamma.sample (sampler.py:276,284) and of Normal.log_p (gmrf.py:343-344).      */
omc_status omc_centered_rowdot(omc_ctx* ctx, int64_t n, const double* a, int64_t ld_a, const double* center_a,
                               const double* b, int64_t ld_b, const double* center_b, double* out);

/* Poisson.rvs for every chain (distribution.py:508-523: scipy.stats.poisson.rvs, i.e. NumPy's legacy generator -- multiplication
 * method below rate 10, PTRS transformed rejection from 10 on): out[c] = one Poisson(rate[c * rate_stride]) count as a double
 * (rate_stride 0: one shared rate).  u_inject [C][ld_u]: the uniforms the draw consumes, in order, NaN-padded (the parity tests
 * replay the reference's generator through it; running off the tape latches the chain in omc_ctx_status); NULL: the chain's
 * Philox stream at draw_index.  Used for the prior draw of a jump parameter without an initial value (mcmc.py:78-85).  */
omc_status omc_poisson_draw(omc_ctx* ctx, const double* rate, int64_t rate_stride, const double* u_inject, int64_t ld_u,
                            uint64_t draw_index, double* out);
/* Uniform.rvs (distribution.py:444-458): out[c][e] = lower[e] + range[e] * U(0,1], e < p (lower/range device [p]);
 * u_inject [C][p]; in-kernel uniforms come from Philox blocks sub, sub+1, ... (two per block).               */
omc_status omc_uniform_draw(omc_ctx* ctx, int64_t p, const double* lower, const double* range, const double* u_inject,
                            uint64_t draw_index, uint32_t sub, double* out);
omc_status omc_diag_gauss_logpdf(omc_ctx* ctx, int64_t kmax, const double* x, const double* mean, const double* prec,
                                 const double* count, double* out, int32_t accumulate);
/* Mixture models: a parameter vector whose prior mean / precision are picked per element
 * by a categorical allocation (parameter.py:376-538), the allocation's conditional draw and the per-component
 * precision update.
 *   omc_mixture_allocation: MixtureAllocation.sample (sampler.py:321-355): prob_k = prior[i or 0][k] *
 *     N(y[c][i]; mean[c][k], 1/prec[c][k]), normalised; alloc = #{k : U > cumsum_k}.  prior [prior_rows x K] shared
 *     (prior_rows 1 or p), mean / prec [C][K] (stride 0 = shared [K]), u_inject [C][p];
 *   omc_categorical_logpmf: Categorical.log_p (distribution.py:318-352, one trial per element):
 *     sum_i log prob[i or 0][alloc[c][i]];
 *   omc_mixture_normal_gamma: NormalGamma.sample for a MixtureParameterMatrix precision (sampler.py:276-287 with
 *     parameter.py:525-538): a_k = a0[k] + #{alloc == k}/2, b_k = b0[k] + sum_{alloc == k} resid^2 / 2,
 *     out[c][k] = Gamma(a_k, scale 1/b_k); a0 / b0 device [K], g_inject [C][K] standard gammas;
 *   omc_gamma_logpdf_vec: Gamma.log_p of a (K, 1) response with per-element shape / rate (device [K]).       */
omc_status omc_mixture_allocation(omc_ctx* ctx, int64_t p, int64_t K, const double* y, const double* prior,
                                  int64_t prior_rows, const double* mean, int64_t mean_stride, const double* prec,
                                  int64_t prec_stride, const double* u_inject, uint64_t draw_index, double* alloc);
omc_status omc_categorical_logpmf(omc_ctx* ctx, int64_t p, int64_t K, const double* alloc, const double* prob,
                                  int64_t prob_rows, double* out, int32_t accumulate);
omc_status omc_mixture_normal_gamma(omc_ctx* ctx, int64_t p, int64_t K, const double* resid, const double* alloc,
                                    const double* a0, const double* b0, const double* g_inject, uint64_t draw_index,
                                    double* out);
omc_status omc_gamma_logpdf_vec(omc_ctx* ctx, int64_t K, const double* x, const double* shape, const double* rate,
                                double* out, int32_t accumulate);
/* omc_gamma_logpdf_ragged: Gamma.log_p (distribution.py:241-261) of a ragged (1, k) response: the sum over the live
 *   entries, or with last_only != 0 the density of the LAST live entry (what ReversibleJump reads from
 *   log_p(..., by_observation=True)[-1], reversible_jump.py:132,143);
 * omc_diag_gauss_grad: gradient of the diagonal Gaussian log-density w.r.t. its response, -prec (x - mean) on the live
 *   entries, 0 beyond (location_scale.py:222-226).                                                              */
omc_status omc_gamma_logpdf_ragged(omc_ctx* ctx, int64_t kmax, const double* x, const double* count, double shape,
                                   double rate, int32_t last_only, double* out, int32_t accumulate);
omc_status omc_diag_gauss_grad(omc_ctx* ctx, int64_t kmax, const double* x, const double* mean, const double* prec,
                               const double* count, double* grad);
omc_status omc_poisson_logpmf(omc_ctx* ctx, const double* x, double rate, double* out, int32_t accumulate);
omc_status omc_count_logpdf(omc_ctx* ctx, const double* count, double per_element, double* out, int32_t accumulate);
omc_status omc_mixture_gather(omc_ctx* ctx, int64_t kmax, int64_t m, const double* param, int64_t param_stride,
                              const double* alloc, const double* count, double fill, double* out);
/* two tables gathered by one allocation in one launch (the mean and the precision of a mixture Normal) */
omc_status omc_mixture_gather2(omc_ctx* ctx, int64_t kmax, int64_t m, const double* alloc, const double* count,
                               const double* param_a, int64_t stride_a, double fill_a, double* out_a, const double* param_b,
                               int64_t stride_b, double fill_b, double* out_b);

/* ---- on-device posterior summaries of the device-resident store ----
 * store is [n_iter][C][size] (iteration-major, as MCMC writes it: mcmc.py:105-106 per chain).
 *   pooled == 0: mean_out / var_out [C][size]: per chain over its n_iter stored iterations;
 *   pooled != 0: mean_out / var_out [size]:    over all chains and iterations.
 * var is the unbiased sample variance (0 when there is a single sample); either output may be NULL.
 * Saves gathering the whole store (82 MB per iteration at cfg3) when only summaries are wanted.   */
omc_status omc_store_moments(omc_ctx* ctx, int64_t n_iter, int64_t size, const double* store,
                             int32_t pooled, double* mean_out, double* var_out);
/* Quantiles of the same store, np.quantile's default ("linear") method, what a user of the reference computes on
 * MCMC.store[param] (host arrays there: mcmc.py:105-111, sampler/sampler.py:89-118):
 *   q        [n_q]  (HOST) levels in [0, 1]
 *   pooled == 0: out [n_q][C][size]: per chain over its n_iter stored iterations (np.quantile(store_c, q, axis=-1));
 *   pooled != 0: out [n_q][size]:    over all chains and iterations.
 *   omit_nan != 0: NaN entries (the padding of variable-size parameters beyond the live length, sampler.py:81-87; iterations
 *   not yet written) are left out per element like np.nanquantile, an element with no valid value gives NaN;
 *   omit_nan == 0: an element with any NaN gives NaN (np.quantile).
 * Exact order statistics by radix refinement on the device (eight passes over the store per group of four levels; no sort,
 * no copy of the store), then numpy's interpolation operation by operation: results are bit-equal to np.quantile's.      */
omc_status omc_store_quantiles(omc_ctx* ctx, int64_t n_iter, int64_t size, const double* store, int32_t pooled, int32_t n_q,
                               const double* q, int32_t omit_nan, double* out);
/* Thinned copy of the store for a thinned gather: out[j] = store[first + j * every] (slabs of C * size doubles),
 * j = 0 .. ceil((n_iter - first) / every) - 1 (that count is left in *n_out when n_out is not NULL, host).           */
omc_status omc_store_thin(omc_ctx* ctx, int64_t n_iter, int64_t size, const double* store, int64_t first, int64_t every,
                          double* out, int64_t* n_out);

/* ---- the one collective of the path: gather of the per-rank stores on a root (RCCL over xGMI) ----
 * Nothing in the reference to replace (it has one chain and no collective; SURVEY section 2.2 COLL row, section 8b
 * export list, section 8e): with the chains of MCMC.run_mcmc (mcmc.py:97-115) sharded over GPUs, this is what puts
 * MCMC.store back together.  One process per GPU; the communicator is RCCL's, bootstrapped from a unique id that the
 * caller ships to every rank by whatever channel it has (the Python mirror uses torch.distributed's object
 * broadcast; a non-Python caller its own).
 *   omc_comm_unique_id     [host] on ONE rank: fills id_out (omc_comm_unique_id_bytes() bytes, 128);
 *   omc_comm_create        on every rank, collectively: ncclCommInitRank on ctx's device;
 *   omc_gather_samples     on every rank, collectively, on ctx's stream (returns without synchronising):
 *       send   [n_outer][counts[rank]][row]  this rank's block (device), chains in local order
 *       recv   [n_outer][sum counts][row]    on `root` (device; ignored elsewhere): chains in rank order
 *       counts [world]                       (host) chains held by every rank; shards may be uneven, also 0
 *     All peers send at once, point to point into the root (xGMI gives every peer its own link); the root stages at
 *     most staging_limit_bytes (0 = 1 GiB) per round and interleaves on the device.                              */
typedef struct omc_comm omc_comm;
int64_t omc_comm_unique_id_bytes(void);
omc_status omc_comm_unique_id(char* id_out, int64_t id_bytes);
omc_status omc_comm_create(omc_ctx* ctx, int32_t world, int32_t rank, const char* id, int64_t id_bytes, omc_comm** out);
omc_status omc_comm_destroy(omc_comm* comm);
omc_status omc_gather_samples(omc_ctx* ctx, omc_comm* comm, const double* send, int64_t n_outer, int64_t row,
                              const int64_t* counts, double* recv, int32_t root, int64_t staging_limit_bytes);
/* The same gather with the peers' blocks already on this GPU: sends[r] = device pointer of rank r's [n_outer][counts[r]][row]
 * block (r = 0..world-1; sends[root] is the root's own block), no communicator.  It runs the root's side of
 * omc_gather_samples unchanged -- slab loop under the staging budget, staging offsets, direct placement for n_outer == 1,
 * the interleave kernel -- with a device-to-device copy standing where ncclRecv stands: what a process that keeps several
 * contexts (shards) on one GPU uses to join their stores, and what tests the root-side arithmetic without peers
 * (tests/test_gather_gpu.py: world 2, 3, 8, even and uneven counts, several staging limits).                          */
omc_status omc_gather_samples_local(omc_ctx* ctx, int32_t world, const double* const* sends, int64_t n_outer, int64_t row,
                                    const int64_t* counts, double* recv, int32_t root, int64_t staging_limit_bytes);

/* ---- raw random streams (tests, prior draws for missing state: mcmc.py:78-80) -------------- */
omc_status omc_fill_normal(omc_ctx* ctx, int64_t n, uint64_t draw_index, double* out, int64_t ld);
omc_status omc_fill_philox_u32(omc_ctx* ctx, int64_t n_words, uint64_t draw_index, uint32_t* out,
                               int64_t ld); /* raw words: INT path, bit-exact vs the host model */

#ifdef __cplusplus
}
#endif
#endif /* OMCMC_HIP_H */
