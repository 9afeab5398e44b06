"""Oracle for the conjugate-Gibbs samplers and the per-iteration bookkeeping
(reference: sampler/sampler.py, distribution/distribution.py, distribution/location_scale.py,
mcmc.py).  TEST INFRASTRUCTURE ONLY.

The reference walks a dict of Distribution objects; here the same arithmetic is written
against explicit arrays.  Random draws are arguments.
"""

import numpy as np
from scipy import sparse, stats

from oracle import gmrf_ref


# ----------------------------------------------------------------------------- Normal-Normal
def normal_conditional(n_param, prior, likelihoods):
    """Canonical parameters (Q, b) of the Gaussian full conditional  [sampler.py:176-192].

    prior       = (P, m): precision matrix and (n,1) mean of the parameter's own distribution
                  (sampler.py:181-183:  Q += P ; b += P m).
    likelihoods = list of dicts, one per response distribution that has the parameter in its mean:
        {"W": precision of the response, "y": (n_y, n_rep) response,
         "A": design matrix or None, "rest": predictor of the other mean terms or None,
         "dense_identity": bool}
      A is None  -> Identity mean (sampler.py:187-188): b += W sum_rep(y); the Hessian is
                    n_rep * G W G' with G = eye(n) DENSE when dense_identity (location_scale.py:
                    237-241, parameter.py:138) -- this is what turns example 4 dense.
      A given    -> LinearCombination mean (sampler.py:190-192): b += A' W (y - rest);
                    Hessian n_rep * A' W A (location_scale.py:238-241, parameter.py:228).
    """
    Q = sparse.csc_matrix((n_param, n_param))
    b = np.zeros((n_param, 1))
    P, m = prior
    Q = Q + P
    b = b + P @ m
    for lk in likelihoods:
        W, y = lk["W"], lk["y"]
        n_rep = y.shape[1]
        if lk.get("A") is None:
            G = np.eye(n_param) if lk.get("dense_identity", True) else sparse.identity(n_param, format="csc")
            GW = G @ W
            Q = Q + n_rep * GW @ G.T
            b = b + W @ np.sum(y, axis=1, keepdims=True)
        else:
            A = lk["A"]
            GW = A.T @ W
            Q = Q + n_rep * GW @ A
            rest = lk.get("rest")
            resid = y if rest is None else y - rest
            b = b + A.T @ W @ resid
    return Q, np.asarray(b)


def normal_normal_draw(n_param, prior, likelihoods, z):
    """NormalNormal.sample for an untruncated prior  [sampler.py:154-207 -> gmrf.py:167-198]."""
    Q, b = normal_conditional(n_param, prior, likelihoods)
    x, mu, L = gmrf_ref.draw_canonical(b, Q, z)
    return np.asarray(x).reshape(n_param, 1), np.asarray(mu).reshape(n_param, 1), Q


# ----------------------------------------------------------------------------- Normal-Gamma
def gamma_conditional(a0, b0, residual, P_unscaled):
    """Posterior shape/rate of a scalar precision  [sampler.py:276-284].

    a = a0 + #{diag(P) > 0}/2  (counts positive diagonal entries, NOT the rank),
    b = b0 + r' P r / 2.
    """
    diag = P_unscaled.diagonal()
    a = float(a0) + np.sum(diag > 0) / 2
    b = float(b0) + (residual.T @ P_unscaled @ residual).item() / 2
    return a, b


def gamma_draw_from_standard(a, b, g):
    """Gamma(a, rate=b) from a standard Gamma(a,1) draw g: g * (1/b); rate 0 -> scale inf
    [sampler.py:285-287; scipy gamma.rvs(a, scale=s) == standard_gamma(a)*s]."""
    scale = np.inf if b == 0 else 1.0 / b
    return g * scale


def gamma_logpdf(x, shape, rate):
    """Gamma log-density, shape/rate convention  [distribution.py:241-261]."""
    return float(np.sum(stats.gamma.logpdf(x, shape, scale=1.0 / rate)))


# ----------------------------------------------------------------------------- whole sweeps
def gmrf_smoother_chain(y, P, n_burn, n_iter, z, g, mu=None, lam0=100.0, tau0=1.0,
                        a_lam=10.0, b_lam=1.0, a_tau=1.0, b_tau=1.0, dense_identity=False):
    """MCMC.run_mcmc for the example-4 model with samplers [NormalNormal(b), NormalGamma(lambda),
    NormalGamma(tau)]  [mcmc.py:87-115; examples/4_GMRF_smoother.ipynb:163-164, 192-203, 236-238].

    z: (n_burn+n_iter, n) standard normals, g: (n_burn+n_iter, 2) standard gammas [lambda, tau].
    Returns the store dict {b (n,n_iter), lambda, tau (1,n_iter), log_post (n_iter,1)}.
    """
    n = y.size
    y = np.asarray(y, dtype=float).reshape(n, 1)
    mu = np.zeros((n, 1)) if mu is None else np.asarray(mu, dtype=float).reshape(n, 1)
    P = sparse.csc_matrix(P)
    # (sparse.identity, not csc_matrix(np.eye(n)): the dense detour is O(n^2) set-up -- 800 MB and 0.65 s at n = 10 000 --
    #  and a timed call would charge it to the sweeps; the matrix is the same, sorted CSC of float64 ones)
    I_n = sparse.identity(n, format="csc")
    A = None if dense_identity else sparse.identity(n, format="csc")
    lam, tau = float(lam0), float(tau0)
    store = {"b": np.full((n, n_iter), np.nan), "lambda": np.full((1, n_iter), np.nan),
             "tau": np.full((1, n_iter), np.nan), "log_post": np.full((n_iter, 1), np.nan)}
    for it in range(-n_burn, n_iter):
        k = it + n_burn
        like = {"W": tau * I_n, "y": y, "A": A, "rest": None if A is None else 0.0, "dense_identity": dense_identity}
        x, _, _ = normal_normal_draw(n, (lam * P, mu), [like], z[k])
        a, b = gamma_conditional(a_lam, b_lam, x - mu, P)
        lam = gamma_draw_from_standard(a, b, g[k, 0])
        a, b = gamma_conditional(a_tau, b_tau, y - x, I_n)
        tau = gamma_draw_from_standard(a, b, g[k, 1])
        if it < 0:
            continue
        store["b"][:, [it]] = x
        store["lambda"][0, it], store["tau"][0, it] = lam, tau
        store["log_post"][it] = gmrf_smoother_log_post(y, x, mu, P, I_n, lam, tau, a_lam, b_lam, a_tau, b_tau)
    return store


def gmrf_smoother_log_post(y, x, mu, P, I_n, lam, tau, a_lam, b_lam, a_tau, b_tau):
    """Model.log_p for the example-4 model  [model.py:57-70; location_scale.py:145-167;
    distribution.py:241-261]: two Gaussian terms (each re-factorises its precision) + two Gamma priors."""
    return (gmrf_ref.gauss_logpdf(y, x, tau * I_n) + gmrf_ref.gauss_logpdf(x, mu, lam * P)
            + gamma_logpdf(lam, a_lam, b_lam) + gamma_logpdf(tau, a_tau, b_tau))


def linreg_chain(X, y, n_burn, n_iter, z, g, lam0=0.01, tau0=1.0, a_tau=1e-3, b_tau=1e-3,
                 a_lam=1e-3, b_lam=1e-3):
    """MCMC.run_mcmc for the example-3 model with samplers [NormalNormal(beta), NormalGamma(tau),
    NormalGamma(lambda)]  [examples/3_linear_regression.ipynb:158-200; mcmc.py:87-115].

    z: (n_burn+n_iter, p) standard normals, g: (n_burn+n_iter, 2) standard gammas [tau, lambda].
    Store keys: beta (p,n_iter), tau, lambda (1,n_iter), log_post (n_iter,1), y = fitted mean (N,n_iter).
    """
    N, p = X.shape
    y = np.asarray(y, dtype=float).reshape(N, 1)
    P_tau, P_lam = sparse.identity(N, format="csc"), sparse.identity(p, format="csc")
    mu = np.zeros((p, 1))
    lam, tau = float(lam0), float(tau0)
    store = {"beta": np.full((p, n_iter), np.nan), "tau": np.full((1, n_iter), np.nan),
             "lambda": np.full((1, n_iter), np.nan), "log_post": np.full((n_iter, 1), np.nan),
             "y": np.full((N, n_iter), np.nan)}
    for it in range(-n_burn, n_iter):
        k = it + n_burn
        like = {"W": tau * P_tau, "y": y, "A": X, "rest": 0}
        beta, _, _ = normal_normal_draw(p, (lam * P_lam, mu), [like], z[k])
        fitted = X @ beta
        a, b = gamma_conditional(a_tau, b_tau, y - fitted, P_tau)
        tau = gamma_draw_from_standard(a, b, g[k, 0])
        a, b = gamma_conditional(a_lam, b_lam, beta - mu, P_lam)
        lam = gamma_draw_from_standard(a, b, g[k, 1])
        if it < 0:
            continue
        store["beta"][:, [it]] = beta
        store["tau"][0, it], store["lambda"][0, it] = tau, lam
        store["log_post"][it] = (gmrf_ref.gauss_logpdf(y, fitted, tau * P_tau)
                                 + gmrf_ref.gauss_logpdf(beta, mu, lam * P_lam)
                                 + gamma_logpdf(tau, a_tau, b_tau) + gamma_logpdf(lam, a_lam, b_lam))
        store["y"][:, [it]] = fitted
    return store
