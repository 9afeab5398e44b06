"""TEST INFRASTRUCTURE (CPU oracle) -- prior draws, restated with every random number passed in.

* truncated_normal_rejection: gmrf.sample_truncated_normal for one replicate (gmrf.py:64-110 -> the rejection sampler
  gmrf.py:113-164; every attempt is one gmrf.sample_normal, gmrf.py:29-61: x = mu + L^-T z with L the lower Cholesky factor
  of Q).  Called from Normal.rvs when the response has domain limits (location_scale.py:252-272).
* poisson_legacy: Poisson.rvs (distribution.py:508-523) is scipy.stats.poisson.rvs, i.e. NumPy's legacy
  RandomState.poisson -- third-party arithmetic that is not under /root/reference (numpy 2.2.6 in the build container,
  numpy/random/src/distributions/distributions.c: random_poisson_mult below rate 10, random_poisson_ptrs -- Hoermann's PTRS
  transformed rejection -- from 10 on, with NumPy's own random_loggam).  Restated here from the published algorithm and pinned,
  value by value and in the number of uniforms consumed, by tests/golden/prior_draws.npz (tests/test_oracle_golden.py).
"""

import numpy as np
from scipy import linalg, sparse


def truncated_normal_rejection(mu, Q, lower, upper, z_tape):
    """One draw.  z_tape (attempts, p): the standard normals of attempt 0, 1, ...  Returns (x (p,), attempts used)."""
    mu = np.asarray(mu, dtype=float).reshape(-1)
    p = mu.size
    lower = np.full(p, -np.inf) if lower is None else np.broadcast_to(np.asarray(lower, dtype=float).reshape(-1), (p,))
    upper = np.full(p, np.inf) if upper is None else np.broadcast_to(np.asarray(upper, dtype=float).reshape(-1), (p,))
    if np.any(lower >= upper):
        raise ValueError("Error lower bound must be strictly less than upper bound")  # gmrf.py:149-150
    Qd = Q.toarray() if sparse.issparse(Q) else np.asarray(Q, dtype=float)
    L = np.linalg.cholesky(Qd)
    for a in range(z_tape.shape[0]):
        x = mu + linalg.solve_triangular(L.T, np.asarray(z_tape[a], dtype=float).reshape(-1), lower=False)
        if not np.any((x < lower) | (x > upper)):
            return x, a + 1
    raise RuntimeError("draw tape exhausted")


_LOGGAM_A = [8.333333333333333e-02, -2.777777777777778e-03, 7.936507936507937e-04, -5.952380952380952e-04,
             8.417508417508418e-04, -1.917526917526918e-03, 6.410256410256410e-03, -2.955065359477124e-02,
             1.796443723688307e-01, -1.39243221690590e+00]


def _loggam(x):
    """NumPy's random_loggam (distributions.c): log Gamma(x) for the integer-valued arguments PTRS needs."""
    if x == 1.0 or x == 2.0:
        return 0.0
    n = int(7 - x) if x < 7.0 else 0
    x0 = x + n
    x2 = (1.0 / x0) * (1.0 / x0)
    gl0 = _LOGGAM_A[9]
    for k in range(8, -1, -1):
        gl0 = gl0 * x2 + _LOGGAM_A[k]
    gl = gl0 / x0 + 0.5 * 1.8378770664093453 + (x0 - 0.5) * np.log(x0) - x0
    if x < 7.0:
        for _ in range(n):
            gl -= np.log(x0 - 1.0)
            x0 -= 1.0
    return gl


def poisson_legacy(lam, uniforms):
    """One draw of NumPy's legacy Poisson generator from the uniforms it would have consumed.  Returns (k, uniforms used)."""
    u = iter(np.asarray(uniforms, dtype=float))
    used = 0
    if lam == 0:
        return 0, 0
    if lam < 10:  # random_poisson_mult
        enlam, X, prod = np.exp(-lam), 0, 1.0
        while True:
            prod *= next(u)
            used += 1
            if prod > enlam:
                X += 1
            else:
                return X, used
    slam, loglam = np.sqrt(lam), np.log(lam)  # random_poisson_ptrs
    b = 0.931 + 2.53 * slam
    a = -0.059 + 0.02483 * b
    invalpha = 1.1239 + 1.1328 / (b - 3.4)
    vr = 0.9277 - 3.6224 / (b - 2)
    while True:
        U = next(u) - 0.5
        V = next(u)
        used += 2
        us = 0.5 - abs(U)
        k = int(np.floor((2 * a / us + b) * U + lam + 0.43))
        if us >= 0.07 and V <= vr:
            return k, used
        if k < 0 or (us < 0.013 and V > us):
            continue
        with np.errstate(divide="ignore"):
            if (np.log(V) + np.log(invalpha) - np.log(a / (us * us) + b)) <= (-lam + k * loglam - _loggam(k + 1)):
                return k, used
