"""ctypes binding of libomcmc_hip.so (include/omcmc_hip.h).

The library is the product: there is no CPU fallback.  Importing this module without the
built library raises ImportError; calling into it without a GPU raises RuntimeError from
omc_ctx_create.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OMC_HIP_LIB selects another build of the same library (A/B timing of kernel variants, benchmarks/ab_headline.py)
LIB_PATH = os.environ.get("OMC_HIP_LIB") or os.path.join(_HERE, "libomcmc_hip.so")

OMC_MAX_TERMS = 4
OMC_SELECT_MAX = 8
OK, INVALID_ARG, NOT_POSDEF, HIP_ERROR, UNSUPPORTED = range(5)

c_dp = C.c_void_p  # device pointer
i64, u64, i32, u32 = C.c_int64, C.c_uint64, C.c_int32, C.c_uint32


class TridiagTerms(C.Structure):
    """omc_tridiag_terms."""

    _fields_ = [
        ("n_terms", i32),
        ("diag", c_dp * OMC_MAX_TERMS),
        ("off", c_dp * OMC_MAX_TERMS),
        ("rhs", c_dp * OMC_MAX_TERMS),
        ("center", c_dp * OMC_MAX_TERMS),
        ("scale", c_dp * OMC_MAX_TERMS),
        ("center_chain", c_dp * OMC_MAX_TERMS),
        ("ld_center_chain", i64),
    ]


class LogpPiece(C.Structure):
    """omc_logp_piece."""

    _fields_ = [("kind", i32), ("n", C.c_double), ("scale", c_dp), ("logdet", c_dp), ("logdet_mult", C.c_double), ("quad", c_dp),
                ("x", c_dp), ("shape", C.c_double), ("rate", C.c_double)]


class DenseTerms(C.Structure):
    """omc_dense_terms."""

    _fields_ = [
        ("n_terms", i32),
        ("mat", c_dp * OMC_MAX_TERMS),
        ("rhs", c_dp * OMC_MAX_TERMS),
        ("scale", c_dp * OMC_MAX_TERMS),
        ("diag_chain", c_dp),
    ]


class BandTerms(C.Structure):
    """omc_band_terms."""

    _fields_ = [
        ("n_terms", i32),
        ("band", c_dp * OMC_MAX_TERMS),
        ("bw", i32 * OMC_MAX_TERMS),
        ("rhs", c_dp * OMC_MAX_TERMS),
        ("scale", c_dp * OMC_MAX_TERMS),
    ]


class GammaBlock(C.Structure):
    """omc_gamma_block."""

    _fields_ = [
        ("enabled", i32),
        ("a0", C.c_double),
        ("b0", C.c_double),
        ("n_pos", i64),
        ("g_inject", c_dp),
        ("draw_index", u64),
        ("store", c_dp),
        ("logdet_unscaled", c_dp),
    ]


# name -> (restype, argtypes); the single source for the symbol-export test
SIGNATURES = {
    "omc_ctx_create": (i32, [i32, i64, u64, i64, C.c_void_p, i32, C.POINTER(C.c_void_p)]),
    "omc_ctx_destroy": (i32, [C.c_void_p]),
    "omc_ctx_status": (i32, [C.c_void_p, C.POINTER(i64)]),
    "omc_ctx_synchronize": (i32, [C.c_void_p]),
    "omc_ctx_set_option": (i32, [C.c_void_p, C.c_char_p, i64]),
    "omc_ctx_counter": (i32, [C.c_void_p, C.c_char_p, C.POINTER(i64)]),
    "omc_poisson_draw": (i32, [C.c_void_p, c_dp, i64, c_dp, i64, u64, c_dp]),
    "omc_ctx_launch_log": (i32, [C.c_void_p, C.POINTER(C.c_double), i64, C.POINTER(i64)]),
    "omc_reentry_descriptor_ok": (i32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "omc_last_error": (C.c_char_p, []),
    "omc_abi_version": (i32, []),
    "omc_tridiag_sample_canonical": (
        i32,
        [C.c_void_p, i64, C.POINTER(TridiagTerms), c_dp, i64, c_dp, i64, u64, c_dp, i64, c_dp, i64, c_dp, c_dp],
    ),
    "omc_gmrf_sweep": (
        i32,
        [C.c_void_p, i64, C.POINTER(TridiagTerms), C.POINTER(GammaBlock), c_dp, i64, c_dp, i64, u64, c_dp, i64, c_dp],
    ),
    "omc_dense_sample_canonical": (
        i32,
        [C.c_void_p, i64, C.POINTER(DenseTerms), c_dp, i64, c_dp, i64, u64, c_dp, i64, c_dp, i64, c_dp],
    ),
    "omc_gram": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp]),
    "omc_design_rhs": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, c_dp]),
    "omc_design_predict": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, i64, c_dp, i64]),
    "omc_weighted_resid_sq": (i32, [C.c_void_p, i64, c_dp, c_dp, i64, c_dp, c_dp]),
    "omc_dense_cholesky": (i32, [C.c_void_p, i64, c_dp, C.c_double, c_dp, c_dp]),
    "omc_rw_step_white": (
        i32,
        [C.c_void_p, i64, c_dp, c_dp, c_dp, C.c_double, c_dp, i64, c_dp, u64, c_dp, i64, i32, c_dp, c_dp, c_dp],
    ),
    "omc_mala_step_white": (
        i32,
        [C.c_void_p, i64, c_dp, c_dp, c_dp, C.c_double, c_dp, i64, c_dp, u64, c_dp, i64, i32, c_dp, c_dp, c_dp],
    ),
    "omc_mala_run_white": (
        i32,
        [C.c_void_p, i64, c_dp, c_dp, c_dp, C.c_double, c_dp, i64, c_dp, u64, u64, i64, c_dp, i64, i32, c_dp, c_dp, c_dp, c_dp, c_dp],
    ),
    "omc_mala_step": (
        i32,
        [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp, C.c_double, c_dp, i64, c_dp, u64, c_dp, i64, c_dp, c_dp],
    ),
    "omc_rw_step": (
        i32,
        [C.c_void_p, i64, c_dp, c_dp, c_dp, C.c_double, c_dp, i64, c_dp, u64, c_dp, i64, c_dp, c_dp],
    ),
    "omc_gmrf_run": (
        i32,
        [C.c_void_p, i64, C.POINTER(TridiagTerms), C.POINTER(GammaBlock), i64, i64, i64, u64, u64, c_dp, i64, i64,
         i64, i64, c_dp, c_dp],
    ),
    "omc_tridiag_quadform": (i32, [C.c_void_p, i64, C.POINTER(TridiagTerms), c_dp, i64, c_dp]),
    "omc_tridiag_takes_center_chain": (i32, [C.c_void_p, i64]),
    "omc_log_post_sum": (i32, [C.c_void_p, i32, C.POINTER(LogpPiece), C.c_double, c_dp]),
    "omc_tridiag_matvec": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp]),
    "omc_tridiag_logdet": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp]),
    "omc_normal_gamma_update": (i32, [C.c_void_p, C.c_double, C.c_double, i64, c_dp, c_dp, u64, c_dp]),
    "omc_scaled_gauss_logpdf": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp, i32]),
    "omc_gamma_logpdf": (i32, [C.c_void_p, c_dp, C.c_double, C.c_double, c_dp, i32]),
    "omc_rj_move": (i32, [C.c_void_p, i64, C.c_double, c_dp, c_dp, c_dp, u64, c_dp, c_dp, c_dp, c_dp]),
    "omc_rj_move_densities": (i32, [C.c_void_p, i64, C.c_double, c_dp, c_dp, c_dp, u64, c_dp, C.c_double, c_dp, c_dp, c_dp, c_dp,
                                    c_dp]),
    "omc_store_ragged": (i32, [C.c_void_p, i64, c_dp, i64, c_dp, c_dp, i64]),
    "omc_store_moments": (i32, [C.c_void_p, i64, i64, c_dp, i32, c_dp, c_dp]),
    "omc_store_quantiles": (i32, [C.c_void_p, i64, i64, c_dp, i32, i32, C.POINTER(C.c_double), i32, c_dp]),
    "omc_store_thin": (i32, [C.c_void_p, i64, i64, c_dp, i64, i64, c_dp, C.POINTER(i64)]),
    "omc_band_sample_canonical": (
        i32, [C.c_void_p, i64, i64, C.POINTER(BandTerms), c_dp, i64, c_dp, i64, u64, c_dp, i64, c_dp, i64, c_dp]),
    "omc_band_quadform": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, i64, c_dp]),
    "omc_band_matvec_chain": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, i64, c_dp, c_dp, i64, i32]),
    "omc_tridiag_gibbs_truncated": (
        i32, [C.c_void_p, i64, C.POINTER(TridiagTerms), c_dp, i64, c_dp, c_dp, c_dp, i64, u64, c_dp, i64]),
    "omc_dense_gibbs_truncated": (
        i32, [C.c_void_p, i64, C.POINTER(DenseTerms), c_dp, i64, c_dp, c_dp, c_dp, i64, u64, c_dp, i64]),
    "omc_domain_penalty": (i32, [C.c_void_p, i64, c_dp, i64, c_dp, c_dp, c_dp]),
    "omc_rw_propose": (
        i32,
        [C.c_void_p, i64, c_dp, i64, i64, c_dp, i64, c_dp, c_dp, c_dp, i64, c_dp, u64, u32, c_dp, i64, i64, c_dp, c_dp],
    ),
    "omc_mala_diag": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp, C.c_double, i32, c_dp, u64, u32, c_dp, c_dp]),
    "omc_mh_accept": (i32, [C.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp, i64, c_dp, u64, u32, c_dp, c_dp, c_dp, c_dp]),
    "omc_chain_select": (i32, [C.c_void_p, c_dp, i64, c_dp, c_dp]),
    "omc_chain_select_multi": (i32, [C.c_void_p, c_dp, i32, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "omc_ragged_resize": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, i64, i64, i64]),
    "omc_gaussian_basis": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, C.c_double, c_dp, c_dp, i64, c_dp]),
    "omc_design_gram_select": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "omc_knot_loop": (i32, [C.c_void_p, i64, i64, c_dp, C.c_double, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                            C.c_double, C.c_double, C.c_double, c_dp, c_dp, u64, c_dp, c_dp, c_dp, c_dp]),
    "omc_design_predict_batched": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, c_dp, C.c_double, c_dp, c_dp]),
    "omc_design_resid_sq_batched": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "omc_design_gram_batched": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "omc_small_sample_canonical": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, u64, c_dp, c_dp]),
    "omc_rj_matched_transition": (
        i32,
        [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_double, i32, C.c_double, C.c_double, c_dp, u64, u32,
         c_dp, c_dp, c_dp],
    ),
    "omc_log_transform": (i32, [C.c_void_p, i64, c_dp, i64, c_dp, i64, c_dp]),
    "omc_centered_rowdot": (i32, [C.c_void_p, i64, c_dp, i64, c_dp, c_dp, i64, c_dp, c_dp]),
    "omc_uniform_draw": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, u64, u32, c_dp]),
    "omc_diag_gauss_logpdf": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp, c_dp, i32]),
    "omc_gamma_logpdf_ragged": (i32, [C.c_void_p, i64, c_dp, c_dp, C.c_double, C.c_double, i32, c_dp, i32]),
    "omc_diag_gauss_grad": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "omc_poisson_logpmf": (i32, [C.c_void_p, c_dp, C.c_double, c_dp, i32]),
    "omc_count_logpdf": (i32, [C.c_void_p, c_dp, C.c_double, c_dp, i32]),
    "omc_mixture_gather": (i32, [C.c_void_p, i64, i64, c_dp, i64, c_dp, c_dp, C.c_double, c_dp]),
    "omc_mixture_gather2": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, i64, C.c_double, c_dp, c_dp, i64, C.c_double, c_dp]),
    "omc_mixture_allocation": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, i64, c_dp, i64, c_dp, i64, c_dp, u64, c_dp]),
    "omc_categorical_logpmf": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, i64, c_dp, i32]),
    "omc_mixture_normal_gamma": (i32, [C.c_void_p, i64, i64, c_dp, c_dp, c_dp, c_dp, c_dp, u64, c_dp]),
    "omc_gamma_logpdf_vec": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp, i32]),
    "omc_mh_invalidate": (i32, [C.c_void_p]),
    "omc_small_spd_ops": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "omc_band_gibbs_truncated": (i32, [C.c_void_p, i64, i64, C.POINTER(BandTerms), c_dp, i64, c_dp, c_dp, c_dp, i64, u64, c_dp, i64]),
    "omc_tridiag_matvec_chain": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp, i64, c_dp, c_dp, i64, i32]),
    "omc_chain_lincomb": (i32, [C.c_void_p, i64, C.c_double, c_dp, i64, C.c_double, c_dp, i64, c_dp, i64]),
    "omc_chain_copy": (i32, [C.c_void_p, i64, c_dp, i64, c_dp, i64]),
    "omc_dense_spectral_prepare": (i32, [C.c_void_p, i64, c_dp, c_dp, c_dp]),
    "omc_dense_spectral_sample": (i32, [C.c_void_p, i64, C.POINTER(DenseTerms), i32, c_dp, c_dp, c_dp, i64, c_dp, i64, u64,
                                        c_dp, i64, c_dp, i64, c_dp]),
    "omc_comm_unique_id_bytes": (i64, []),
    "omc_comm_unique_id": (i32, [C.c_char_p, i64]),
    "omc_comm_create": (i32, [C.c_void_p, i32, i32, C.c_char_p, i64, C.POINTER(C.c_void_p)]),
    "omc_comm_destroy": (i32, [C.c_void_p]),
    "omc_gather_samples": (i32, [C.c_void_p, C.c_void_p, c_dp, i64, i64, C.POINTER(i64), c_dp, i32, i64]),
    "omc_gather_samples_local": (i32, [C.c_void_p, i32, C.POINTER(C.c_void_p), i64, i64, C.POINTER(i64), c_dp, i32, i64]),
    "omc_fill_normal": (i32, [C.c_void_p, i64, u64, c_dp, i64]),
    "omc_fill_philox_u32": (i32, [C.c_void_p, i64, u64, c_dp, i64]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C openmcmc_amd/csrc`).  openmcmc_amd has no CPU fallback."
        )
    # PyTorch-ROCm ships its own libamdhip64.so.7; load it first so that this library binds to
    # the SAME HIP runtime (two runtimes in one process cannot share the device or a stream).
    import torch  # noqa: F401

    # same for rocBLAS / rocSOLVER (dense path): bind to the copies PyTorch bundles, not to a
    # second set from /opt/rocm
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    for name in ("librocblas.so", "librocsolver.so"):
        path = os.path.join(tlib, name)
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("OMC_HIP_LIB") and not hasattr(lib, name):
            continue  # an A/B build of an older revision (benchmarks/build_rev.sh): entry points added since are simply absent
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(status, ctx=None):
    """Map omc_status to the reference's exception conventions."""
    if status == OK:
        return
    if status == INVALID_ARG:
        raise ValueError("libomcmc_hip: invalid argument")
    if status == NOT_POSDEF:
        raise np.linalg.LinAlgError("Matrix is not positive definite")
    if status == UNSUPPORTED:
        raise NotImplementedError("libomcmc_hip: unsupported configuration")
    raise RuntimeError("libomcmc_hip: " + (lib.omc_last_error() or b"HIP error").decode())
