// Context, status latch, options.
#include <string.h>

#include <string>

#include "omc_common.h"

static thread_local std::string g_last_error;

void omc_set_error(const char* what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}

void omc_set_error_text(const char* text) { g_last_error = text; }

extern "C" {

const char* omc_last_error(void) { return g_last_error.c_str(); }

int32_t omc_abi_version(void) { return 2; }

omc_status omc_ctx_create(int32_t device, int64_t n_chains, uint64_t seed, int64_t chain_id_offset,
                          void* stream, int32_t create_stream, omc_ctx** out) {
  if (!out || n_chains <= 0 || chain_id_offset < 0) return OMC_INVALID_ARG;
  *out = nullptr;
  int count = 0;
  OMC_HIP_CHECK(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(device));
  omc_ctx* c = new omc_ctx();
  c->device = device;
  c->n_chains = n_chains;
  c->seed = seed;
  c->chain_offset = chain_id_offset;
  c->own_stream = (create_stream != 0);
  c->workspace = nullptr;
  c->long_band = nullptr; c->long_band_bytes = 0; c->long_quad = nullptr; c->long_quad_bytes = 0;
  c->workspace_bytes = 0;
  c->blas = nullptr;
  c->store_ws = nullptr; c->store_ws_bytes = 0;
  c->dense_factor = nullptr; c->dense_factor_bytes = 0;
  c->dense_winv = nullptr; c->dense_winv_bytes = 0; c->dense_panel_old = 0;
  c->dense_info = nullptr; c->dense_info_bytes = 0;
  c->slice_buf = nullptr; c->slice_buf_bytes = 0;
  c->dense_tmp = nullptr; c->dense_tmp_bytes = 0;
  c->rj_tmp = nullptr; c->rj_tmp_bytes = 0;
  c->mh_work = nullptr; c->mh_work_bytes = 0;
  c->white_prep = nullptr; c->white_prep_bytes = 0; c->white_L = nullptr; c->white_mu = nullptr; c->white_d = 0;
  c->white_a = nullptr; c->white_a_bytes = 0; c->white_x = nullptr; c->white_ld = 0;
  c->white_traj = nullptr; c->white_traj_bytes = 0;
  for (int i = 0; i < 4; ++i) c->white_ev[i] = nullptr;
  c->rww_a = nullptr; c->rww_a_bytes = 0; c->rww_x = nullptr; c->rww_ld = 0; c->rww_mu = nullptr; c->rww_mu_neg = nullptr; c->rww_mu_bytes = 0;
  c->rw_prep = nullptr; c->rw_prep_bytes = 0; c->rw_LQ = nullptr; c->rw_d = 0;
  c->mala_prep = nullptr; c->mala_prep_bytes = 0; c->mala_Q = nullptr; c->mala_L = nullptr; c->mala_step = 0.0; c->mala_d = 0;
  c->tridiag_algo = 0;
  c->tridiag_seg = 0;
  c->tridiag_generic = 0;
  c->tridiag_newton_max = 4;
  c->tridiag_perturb_ppb = 0;
  c->tridiag_quad_skip = 0;
  c->stamps = nullptr;
  c->sweep_times = nullptr; c->sweep_times_cap = 0; c->sweep_times_pos = 0;
  c->launch_log_n = 0; c->launch_log_total = 0;
  c->blas_aux = nullptr;
  c->aux_stream = nullptr;
  c->ev_fork = c->ev_join = nullptr;
  c->dense_overlap = 1;
  c->dense_blocked_min = 144;  // measured crossover with rocSOLVER: even at 128, the blocked route 1.3-1.8x faster at 200
  c->debug_zero_z = 0;
  c->dense_use_rocsolver = 0;
  c->gram_use_rocblas = 0;
  c->mh_use_rocblas = 0;
  c->mh_gemm_ksplit = 4;
  c->band_algo = 0; c->band_seg_overlap = 192; c->band_seg_count = 0; c->band_blocked_threads = 0;
  if (c->own_stream) {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { omc_set_error("hipStreamCreate", e); delete c; return OMC_HIP_ERROR; }
  } else {
    c->stream = (hipStream_t)stream;
  }
  c->d_gamma_tab = nullptr;
  c->run_ev_begin = c->run_ev_end = nullptr;
  c->d_handoff = nullptr; c->run_epoch = 1; c->run_sweeps_per_launch = 32; c->run_reenter = 2; c->run_reenter_force = 0; c->run_block_sweeps = 0;
  hipError_t e = hipMalloc(&c->d_bad_chain, 8 * sizeof(long long));
  if (e != hipSuccess) { omc_set_error("hipMalloc", e); delete c; return OMC_HIP_ERROR; }
  c->d_fallbacks = (unsigned long long*)(c->d_bad_chain + 1);
  long long init[8] = {OMC_NO_BAD_CHAIN, 0, 0, 0, 0, 0, 0, 0};
  e = hipMemcpy(c->d_bad_chain, init, sizeof(init), hipMemcpyHostToDevice);
  if (e != hipSuccess) { omc_set_error("hipMemcpy", e); hipFree(c->d_bad_chain); delete c; return OMC_HIP_ERROR; }
  *out = c;
  return OMC_OK;
}

omc_status omc_ctx_destroy(omc_ctx* ctx) {
  if (!ctx) return OMC_INVALID_ARG;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->workspace) hipFree(ctx->workspace);
  if (ctx->store_ws) hipFree(ctx->store_ws);
  if (ctx->long_band) hipFree(ctx->long_band);
  if (ctx->long_quad) hipFree(ctx->long_quad);
  if (ctx->d_gamma_tab) hipFree(ctx->d_gamma_tab);
  if (ctx->d_handoff) hipFree(ctx->d_handoff);
  if (ctx->dense_factor) hipFree(ctx->dense_factor);
  if (ctx->dense_winv) hipFree(ctx->dense_winv);
  if (ctx->dense_info) hipFree(ctx->dense_info);
  if (ctx->slice_buf) hipFree(ctx->slice_buf);
  if (ctx->dense_tmp) hipFree(ctx->dense_tmp);
  if (ctx->rj_tmp) hipFree(ctx->rj_tmp);
  if (ctx->mh_work) hipFree(ctx->mh_work);
  if (ctx->mala_prep) hipFree(ctx->mala_prep);
  if (ctx->white_prep) hipFree(ctx->white_prep);
  if (ctx->white_a) hipFree(ctx->white_a);
  if (ctx->white_traj) hipFree(ctx->white_traj);
  if (ctx->rww_a) hipFree(ctx->rww_a);
  if (ctx->rww_mu_neg) hipFree(ctx->rww_mu_neg);
  if (ctx->rw_prep) hipFree(ctx->rw_prep);
  omc_dense_release(ctx);
  hipFree(ctx->d_bad_chain);
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return OMC_OK;
}

omc_status omc_ctx_synchronize(omc_ctx* ctx) {
  if (!ctx) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return OMC_OK;
}

omc_status omc_ctx_status(omc_ctx* ctx, int64_t* first_bad_chain) {
  if (!ctx || !first_bad_chain) return OMC_INVALID_ARG;
  long long w[3] = {OMC_NO_BAD_CHAIN, 0, 0};
  OMC_HIP_CHECK(hipMemcpyAsync(w, ctx->d_bad_chain, sizeof(w), hipMemcpyDeviceToHost, ctx->stream));
  OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  const long long v = w[0];
  if (w[2] != 0) {  // a several-sweeps launch whose hand-over never arrived: the results of that run are not to be used
    long long zero = 0;
    OMC_HIP_CHECK(hipMemcpyAsync(ctx->d_bad_chain + 2, &zero, sizeof(zero), hipMemcpyHostToDevice, ctx->stream));
    OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    omc_set_error_text("omc_gmrf_run: a sweep's hand-over to the next sweep of its chain did not arrive (set run_sweeps_per_launch = 1)");
    *first_bad_chain = -1;
    return OMC_HIP_ERROR;
  }
  if (v == OMC_NO_BAD_CHAIN) {
    *first_bad_chain = -1;
    return OMC_OK;
  }
  *first_bad_chain = (int64_t)v;
  long long init = OMC_NO_BAD_CHAIN;
  OMC_HIP_CHECK(hipMemcpyAsync(ctx->d_bad_chain, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return OMC_NOT_POSDEF;
}

omc_status omc_ctx_counter(omc_ctx* ctx, const char* name, int64_t* value) {
  if (!ctx || !name || !value) return OMC_INVALID_ARG;
  if (!strcmp(name, "run_handoff_timeouts")) {
    unsigned long long v = 0;
    OMC_HIP_CHECK(hipMemcpyAsync(&v, ctx->d_fallbacks + 1, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *value = (int64_t)v;
    return OMC_OK;
  }
  if (!strcmp(name, "band_join_retries")) {
    unsigned long long v = 0;
    OMC_HIP_CHECK(hipMemcpyAsync(&v, ctx->d_fallbacks + 3, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *value = (int64_t)v;
    return OMC_OK;
  }
  if (!strcmp(name, "band_join_fallbacks")) {
    unsigned long long v = 0;
    OMC_HIP_CHECK(hipMemcpyAsync(&v, ctx->d_fallbacks + 2, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *value = (int64_t)v;
    return OMC_OK;
  }
  if (!strcmp(name, "wall_clock_khz")) {  // rate of the s_memrealtime counter the sweep clock records (not a counter; no sync)
    int khz = 0;
    OMC_HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    *value = khz;
    return OMC_OK;
  }
  if (!strcmp(name, "reenter_abi_ok")) {  // probes the loaded kernel descriptors on first use (synchronises then)
    OMC_HIP_CHECK(hipSetDevice(ctx->device));
    *value = omc_reentry_probe_result(ctx);
    return OMC_OK;
  }
  if (!strcmp(name, "sweep_times_pos")) {  // ring index the next sweep of omc_gmrf_run will write (no sync)
    *value = ctx->sweep_times_pos;
    return OMC_OK;
  }
  if (!strcmp(name, "tridiag_join_fallbacks")) {
    unsigned long long v = 0;
    OMC_HIP_CHECK(hipMemcpyAsync(&v, ctx->d_fallbacks, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    OMC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *value = (int64_t)v;
    return OMC_OK;
  }
  return OMC_INVALID_ARG;
}

omc_status omc_ctx_launch_log(omc_ctx* ctx, double* out, int64_t cap, int64_t* n_launches) {
  if (!ctx || !n_launches || cap < 0 || (cap > 0 && !out)) return OMC_INVALID_ARG;
  *n_launches = ctx->launch_log_total;
  const int64_t m = ctx->launch_log_n < cap ? ctx->launch_log_n : cap;
  for (int64_t i = 0; i < m; ++i) {
    const omc_ctx::LaunchRec& r = ctx->launch_log[i];
    out[5 * i] = r.t_begin; out[5 * i + 1] = r.t_end; out[5 * i + 2] = (double)r.n_sweeps; out[5 * i + 3] = (double)r.form;
    out[5 * i + 4] = (double)r.ring_pos;
  }
  return OMC_OK;
}

omc_status omc_ctx_set_option(omc_ctx* ctx, const char* name, int64_t value) {
  if (!ctx || !name) return OMC_INVALID_ARG;
  if (!strcmp(name, "tridiag_algo")) {
    if (value < 0 || value > 2) return OMC_INVALID_ARG;
    ctx->tridiag_algo = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "tridiag_seg")) {
    if (value != 0 && value != 8 && value != 10 && value != 16 && value != 20 && value != 32) return OMC_INVALID_ARG;
    ctx->tridiag_seg = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "run_sweeps_per_launch")) {
    if (value < 1 || value > 32) return OMC_INVALID_ARG;
    ctx->run_sweeps_per_launch = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "run_block_sweeps")) {
    if (value < 0 || value > 32) return OMC_INVALID_ARG;
    ctx->run_block_sweeps = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "run_reenter")) {
    if (value < 0 || value > 2) return OMC_INVALID_ARG;
    ctx->run_reenter = (int)value;
    ctx->run_reenter_force = 1;  // an explicit choice holds for every chain count (tests, A/B runs)
    return OMC_OK;
  }
  if (!strcmp(name, "tridiag_quad_skip")) {
    if (value < 0 || value > 15) return OMC_INVALID_ARG;
    ctx->tridiag_quad_skip = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "tridiag_perturb_ppb")) {
    if (value < 0 || value > 1000000000) return OMC_INVALID_ARG;
    ctx->tridiag_perturb_ppb = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "tridiag_newton_max")) {
    if (value < 0 || value > 64) return OMC_INVALID_ARG;
    ctx->tridiag_newton_max = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "tridiag_generic")) {
    ctx->tridiag_generic = value != 0;
    return OMC_OK;
  }
  if (!strcmp(name, "band_algo")) {
    if (value < 0 || value > 3) return OMC_INVALID_ARG;
    ctx->band_algo = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "band_seg_overlap")) {
    if (value < 8 || value > 65536) return OMC_INVALID_ARG;
    ctx->band_seg_overlap = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "band_seg_count")) {
    if (value < 0 || value == 1 || value > 128) return OMC_INVALID_ARG;
    ctx->band_seg_count = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "band_blocked_threads")) {
    if (value != 0 && value != 512 && value != 4 && value != 8 && value != 16) return OMC_INVALID_ARG;
    ctx->band_blocked_threads = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "mh_gemm_ksplit")) {
    if (value != 1 && value != 2 && value != 4) return OMC_INVALID_ARG;
    ctx->mh_gemm_ksplit = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "mh_use_rocblas")) {
    ctx->mh_use_rocblas = value != 0;
    return OMC_OK;
  }
  if (!strcmp(name, "gram_use_rocblas")) {
    ctx->gram_use_rocblas = value != 0;
    return OMC_OK;
  }
  if (!strcmp(name, "dense_use_rocsolver")) {
    ctx->dense_use_rocsolver = value != 0;
    return OMC_OK;
  }
  if (!strcmp(name, "debug_zero_z")) {
    ctx->debug_zero_z = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "dense_blocked_min")) {
    if (value < 1 || value > 32768) return OMC_INVALID_ARG;
    ctx->dense_blocked_min = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "dense_panel_old")) {
    if (value != 0 && value != 1) return OMC_INVALID_ARG;
    ctx->dense_panel_old = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "dense_overlap")) {
    if (value != 0 && value != 1) return OMC_INVALID_ARG;
    ctx->dense_overlap = (int)value;
    return OMC_OK;
  }
  if (!strcmp(name, "sweep_times_ptr")) {  // diagnostic: device ring [cap][n_chains][2] of uint64, 0 = off (set the capacity first)
    if (value != 0 && ctx->sweep_times_cap < OMC_SWEEP_RING_MIN) return OMC_INVALID_ARG;
    ctx->sweep_times = (unsigned long long*)(uintptr_t)value;
    ctx->sweep_times_pos = 0;
    return OMC_OK;
  }
  if (!strcmp(name, "sweep_times_cap")) {
    if (value != 0 && value < OMC_SWEEP_RING_MIN) return OMC_INVALID_ARG;
    // a ring set earlier was sized for the OLD capacity: the clock stays off until a pointer is given for the new one
    // (a larger capacity on the old pointer would have omc_gmrf_run write records past the ring's end)
    if (value != ctx->sweep_times_cap) ctx->sweep_times = nullptr;
    ctx->sweep_times_cap = value;
    ctx->sweep_times_pos = 0;
    return OMC_OK;
  }
  if (!strcmp(name, "run_event_begin") || !strcmp(name, "run_event_end")) {
    // a caller-owned hipEvent_t (0 = none): omc_gmrf_run records it on the context's stream in front of its first / behind its
    // last launch -- the timing events of a caller whose own event calls cost more than the launch (an interpreter)
    (name[10] == 'b' ? ctx->run_ev_begin : ctx->run_ev_end) = (hipEvent_t)(uintptr_t)value;
    return OMC_OK;
  }
  if (!strcmp(name, "stamps_ptr")) {  // diagnostic: device buffer [n_chains][16][16] of uint64, 0 = off
    ctx->stamps = (unsigned long long*)(uintptr_t)value;
    return OMC_OK;
  }
  return OMC_INVALID_ARG;
}

}  // extern "C"
