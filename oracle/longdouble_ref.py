"""Extended-precision restatement of the tridiagonal canonical draw (test infrastructure, never shipped).

The same recurrences as gmrf.sample_normal_canonical on a tridiagonal Q (factor gmrf.py:489-520, solves
gmrf.py:414-462, draw gmrf.py:29-61), carried in numpy.longdouble (x87 80-bit on the x86 hosts here and on the GPU
box: 64-bit mantissa, eps 1.1e-19).  It is the yardstick for weakly contractive chains (lambda/tau >> 1), where the
fp64 oracle itself is only good to eps * cond(Q): with it the tests can say how far the fp64 sequential algorithm
and the GPU kernel each are from the exact-arithmetic answer, instead of loosening a tolerance on faith.
"""

import numpy as np


def tridiag_draw(a, b, r, z):
    """a (n,) diagonal, b (n-1,) off-diagonal of Q; r (n,) right-hand side; z (n,) N(0,1) draws.
    Returns x = Q^-1 r + L^-T z, mu = Q^-1 r, log det Q as float64 values computed in longdouble."""
    ld = np.longdouble
    a, b, r, z = (np.asarray(v, dtype=ld) for v in (a, b, r, z))
    n = a.size
    D = np.empty(n, dtype=ld)
    l = np.zeros(n, dtype=ld)
    u = np.empty(n, dtype=ld)
    D[0], u[0] = a[0], r[0]
    for i in range(1, n):
        l[i - 1] = b[i - 1] / D[i - 1]
        D[i] = a[i] - l[i - 1] * b[i - 1]
        u[i] = r[i] - l[i - 1] * u[i - 1]
    if np.any(D <= 0):
        raise np.linalg.LinAlgError("not positive definite")
    g = u / D + z / np.sqrt(D)
    m = u / D
    x = np.empty(n, dtype=ld)
    mu = np.empty(n, dtype=ld)
    x[-1], mu[-1] = g[-1], m[-1]
    for i in range(n - 2, -1, -1):
        x[i] = g[i] - l[i] * x[i + 1]
        mu[i] = m[i] - l[i] * mu[i + 1]
    return x.astype(np.float64), mu.astype(np.float64), float(np.sum(np.log(D)))
