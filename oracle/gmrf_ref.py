"""Oracle for the GMRF numerics layer (reference: gmrf.py).  TEST INFRASTRUCTURE ONLY.

All functions are pure: random draws (`z`) are arguments, never generated here.
"""

import numpy as np
from scipy import linalg as sla
from scipy import sparse
from scipy.sparse import linalg as spla


def rw1_precision(s, is_sparse=True):
    """First-order random-walk precision on irregular locations `s`  [gmrf.py:375-411].

    diag_i = 1/d_{i-1} + 1/d_i (one-sided at the ends), off_i = -1/d_i, d = diff(s).
    A single location gives the 1x1 matrix [[1]].
    """
    s = np.squeeze(np.asarray(s, dtype=float)) if np.ndim(s) > 1 else np.asarray(s, dtype=float)
    if s.size <= 1:
        return np.array(1, ndmin=2)
    inv_gap = 1.0 / np.diff(s)
    diag = np.concatenate(([inv_gap[0]], inv_gap[:-1] + inv_gap[1:], [inv_gap[-1]]))
    if is_sparse:
        return sparse.diags((-inv_gap, diag, -inv_gap), offsets=[-1, 0, 1], format="csc")
    return np.diag(diag) - np.diag(inv_gap, -1) - np.diag(inv_gap, 1)


def rw1_precision_temporal(seconds, unit_length=1.0, is_sparse=True):
    """RW1 precision from timestamps already converted to seconds  [gmrf.py:351-372]."""
    seconds = np.asarray(seconds, dtype=float)
    return rw1_precision((seconds - seconds.min()) / unit_length, is_sparse=is_sparse)


def factor_lower(Q):
    """Lower Cholesky factor, Q = L L'  [gmrf.py:465-520].

    Sparse Q: unpivoted, unpermuted SuperLU LU, L = L_lu * diag(sqrt(U_ii)); if any U_ii <= 0
    the dense factorisation of Q.toarray() is used instead (and raises LinAlgError when Q is
    not positive definite).  Dense Q: LAPACK potrf.  Non-square input is a ValueError.
    """
    if Q.shape[0] != Q.shape[1]:
        raise ValueError("Matrix is not square")
    if not sparse.issparse(Q):
        return np.linalg.cholesky(Q)
    lu = spla.splu(sparse.csc_matrix(Q), diag_pivot_thresh=0, options={"RowPerm": False, "ColPerm": False})
    pivots = lu.U.diagonal()
    if (pivots > 0).all():
        return lu.L.dot(sparse.diags(pivots**0.5))
    return np.linalg.cholesky(Q.toarray())


def solve_general(a, b):
    """x with a x = b: sparse -> SuperLU spsolve, dense -> LAPACK gesv  [gmrf.py:414-434]."""
    if sparse.issparse(a) or sparse.issparse(b):
        return spla.spsolve(a, b)
    return np.linalg.solve(a, b)


def solve_with_factor(L, b, lower=True):
    """(L L')^{-1} b from a Cholesky factor  [gmrf.py:437-462]."""
    if sparse.issparse(L) or sparse.issparse(b):
        lo, up = (L, L.T) if lower else (L.T, L)
        return spla.spsolve(up, spla.spsolve(lo, b))
    return sla.cho_solve((L, lower), b)


def draw_from_factor(mu, L, z):
    """Rue & Held Alg. 2.4 with the N(0,I) draw `z` (p x n) supplied: mu + L^{-T} z  [gmrf.py:29-61]."""
    z = np.asarray(z, dtype=float)
    return np.asarray(solve_general(L.T, z)).reshape(z.shape) + mu


def draw_canonical(b, Q, z):
    """Rue & Held Alg. 2.5: x ~ N(Q^{-1} b, Q^{-1}) with the draw `z` supplied  [gmrf.py:167-198].

    Returns (x, mu, L) so tests can also pin the intermediate mean and factor.
    """
    L = factor_lower(Q)
    mu = np.asarray(solve_with_factor(L, b, lower=True)).reshape(b.shape)
    return draw_from_factor(mu, L, np.asarray(z).reshape(b.shape)), mu, L


def gauss_logpdf(x, mu, Q, by_observation=False):
    """Gaussian log-density from the precision  [gmrf.py:321-348].

    0.5 * (2 sum log L_ii - d log 2pi - sum_j (L'(x-mu))_j^2) per column; summed over columns
    unless by_observation.
    """
    L = factor_lower(Q)
    d = L.shape[0]
    logdet = 2 * np.sum(np.log(L.diagonal()))
    w = L.T @ (x - mu)
    lp = 0.5 * (logdet - d * np.log(2 * np.pi) - np.sum(np.power(w, 2), axis=0))
    return lp if by_observation else np.sum(lp)


def gibbs_truncated_scan(b, Q, x, lower, upper, u):
    """One scan of single-site truncated-normal updates in index order  [gmrf.py:201-266], the uniforms `u` behind
    scipy's truncnorm.rvs supplied (one per coordinate).  Both limits infinite -> the caller uses draw_canonical
    (gmrf.py:231-232).  Returns the updated copy of x."""
    from oracle import truncnorm_ref

    x = np.array(x, dtype=float).reshape(-1, 1)
    b = np.asarray(b, dtype=float).reshape(-1, 1)
    p = x.size
    lower = np.broadcast_to(np.asarray(-np.inf if lower is None else lower, dtype=float).reshape(-1, 1), (p, 1))
    upper = np.broadcast_to(np.asarray(np.inf if upper is None else upper, dtype=float).reshape(-1, 1), (p, 1))
    u = np.asarray(u, dtype=float).reshape(-1)
    Qd = Q.toarray() if sparse.issparse(Q) else np.asarray(Q, dtype=float)
    if p == 1:  # gmrf.py:244-247
        return np.array(truncnorm_ref.truncated_normal_rv(b / Qd, 1 / np.sqrt(Qd), lower, upper, u[0]), ndmin=2)
    for i in range(p):  # gmrf.py:254-264
        Q_ii = Qd[i, i]
        v_i = 1 / Q_ii
        row = (Q.getrow(i) @ x) if sparse.issparse(Q) else (Qd[i, :] @ x)
        cond_mean = v_i * (b[i] - row + Q_ii * x[i])
        x[i] = truncnorm_ref.truncated_normal_rv(np.ravel(cond_mean)[0], np.sqrt(v_i), lower[i, 0], upper[i, 0], u[i])
    return x
