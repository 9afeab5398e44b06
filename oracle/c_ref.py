"""ctypes access to the C oracle (oracle/c/tridiag_ref.c).  TEST INFRASTRUCTURE ONLY."""

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_DIR, "libtridiag_ref.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", _DIR], check=True, capture_output=True)
        _LIB = C.CDLL(path)
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        _LIB.omc_ref_tridiag_draw.restype = C.c_int
        _LIB.omc_ref_tridiag_draw.argtypes = [C.c_int64, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp, dp]
        _LIB.omc_ref_tridiag_logdet.restype = C.c_int
        _LIB.omc_ref_tridiag_logdet.argtypes = [C.c_int64, dp, dp, dp]
    return _LIB


def tridiag_draw(p_diag, p_off, lam, tau, y, mu, z):
    """x ~ N(Q^-1 b, Q^-1), Q = lam*P + tau*I, b = lam*P mu + tau*y; returns (x, quad[2], logdetQ)."""
    n = p_diag.size
    f = lambda v: np.ascontiguousarray(v, dtype=np.float64)  # noqa: E731
    off = f(p_off) if n > 1 else np.zeros(1)
    x, quad, logdet, work = np.empty(n), np.empty(2), np.empty(1), np.empty(n)
    rc = _lib().omc_ref_tridiag_draw(n, f(p_diag), off, float(lam), float(tau), f(y), f(mu), f(z), x, quad, logdet, work)
    if rc:
        raise np.linalg.LinAlgError(f"non-positive pivot at node {rc - 1}")
    return x, quad, float(logdet[0])


def tridiag_logdet(diag, off):
    n = diag.size
    out = np.empty(1)
    rc = _lib().omc_ref_tridiag_logdet(n, np.ascontiguousarray(diag, dtype=np.float64),
                                       np.ascontiguousarray(off, dtype=np.float64) if n > 1 else np.zeros(1), out)
    if rc:
        raise np.linalg.LinAlgError(f"non-positive pivot at node {rc - 1}")
    return float(out[0])
