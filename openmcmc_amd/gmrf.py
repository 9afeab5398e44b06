"""GMRF helpers with the reference's names (gmrf.py).

Builders of the RW1 precision are setup-time host code (they run once per model); the numerical
hot path -- sample_normal_canonical, multivariate_normal_pdf on chain batches -- goes to the HIP
library through the Engine.
"""

import numpy as np
from scipy import sparse


def precision_irregular(s, is_sparse: bool = True):
    """First-order random-walk precision for irregular locations (Rue & Held 2005, pp. 97-99;
    reference gmrf.py:375-411): Q_ii = 1/d_{i-1} + 1/d_i, Q_{i,i+1} = -1/d_i, d = diff(s)."""
    s = np.asarray(s, dtype=np.float64)
    if s.ndim > 1:
        s = np.squeeze(s)
    if s.size <= 1:
        return np.array(1, ndmin=2)
    w = 1.0 / np.diff(s)
    main = np.zeros(s.size)
    main[:-1] += w
    main[1:] += w
    if is_sparse:
        return sparse.diags(diagonals=(-w, main, -w), offsets=[-1, 0, 1], format="csc")
    return np.diag(main) - np.diag(w, k=-1) - np.diag(w, k=1)


def precision_temporal(time, unit_length: float = 1.0, is_sparse: bool = True):
    """RW1 precision from a pandas DatetimeIndex/array (reference gmrf.py:351-372)."""
    seconds = (time - time.min()).total_seconds() / unit_length
    return precision_irregular(np.asarray(seconds, dtype=np.float64), is_sparse=is_sparse)


def sample_normal_canonical(engine, n, terms, x_out=None, z=None, draw_index=0):
    """x_c ~ N(Q_c^{-1} b_c, Q_c^{-1}) for every chain (reference gmrf.py:167-198) with Q, b given as
    tridiagonal terms (see Engine.tridiag_terms)."""
    x = engine.empty(engine.n_chains, n) if x_out is None else x_out
    engine.tridiag_sample_canonical(n, terms, x, z=z, draw_index=draw_index)
    return x
