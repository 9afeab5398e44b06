"""Oracle for the Metropolis-Hastings family on a Gaussian target
(reference: sampler/metropolis_hastings.py, distribution/location_scale.py).  TEST INFRASTRUCTURE ONLY.

Target: x ~ N(mu, Q^{-1}) given as Normal("x", mean="mu", precision="Q"), the cfg4 model of
SURVEY.md section 8d.  Draws (z ~ N(0,I), u ~ U(0,1)) are arguments.
"""

import numpy as np

from oracle import gmrf_ref


def gauss_grad_hess(x, mu, Q):
    """Gradient of +log p and Hessian of -log p w.r.t. the response  [location_scale.py:222-232]."""
    return -Q @ (x - mu), Q


def mala_proposal_params(x, grad, hess, step):
    """mMALA proposal mean and factor  [metropolis_hastings.py:325-348]:
    Lam = H/step^2, L = chol(Lam), m = x + 0.5 Lam^{-1} g."""
    L = gmrf_ref.factor_lower(hess / (step**2))
    m = x + 0.5 * np.asarray(gmrf_ref.solve_with_factor(L, grad, lower=True)).reshape(grad.shape)
    return m, L


def mala_log_q(x, m, L):
    """log proposal density without the 2 pi constant  [metropolis_hastings.py:350-373]."""
    w = L.T @ (x - m)
    return float(np.sum(np.log(L.diagonal())) - 0.5 * (w.T @ w).item())


def mala_step(x, mu, Q, step, z, u):
    """One ManifoldMALA.sample  [metropolis_hastings.py:102-125, 127-173, 301-323].
    Returns (x_next, accepted, internals dict)."""
    g, H = gauss_grad_hess(x, mu, Q)
    m_f, L_f = mala_proposal_params(x, g, H, step)
    prop = gmrf_ref.draw_from_factor(m_f, L_f, z.reshape(x.shape))
    lq_fwd = mala_log_q(prop, m_f, L_f)
    g2, H2 = gauss_grad_hess(prop, mu, Q)
    m_r, L_r = mala_proposal_params(prop, g2, H2, step)
    lq_rev = mala_log_q(x, m_r, L_r)
    lp_cur = gmrf_ref.gauss_logpdf(x, mu, Q)
    lp_prop = gmrf_ref.gauss_logpdf(prop, mu, Q)
    log_alpha = lp_prop + lq_rev - (lp_cur + lq_fwd)
    accepted = bool(np.log(u) < log_alpha)
    info = {"grad": g, "hess": H, "mu": m_f, "chol": L_f, "prop": prop, "lq_fwd": lq_fwd,
            "lq_rev": lq_rev, "lp_cur": lp_cur, "lp_prop": lp_prop, "log_alpha": log_alpha}
    return (prop if accepted else x), accepted, info


def rw_step(x, mu, Q, step, z, u):
    """One untruncated RandomWalk.sample  [metropolis_hastings.py:212-269, 127-173]:
    x' = x + step*z, symmetric proposal, accept iff log u < lp' - lp."""
    prop = x + step * z.reshape(x.shape)
    log_alpha = gmrf_ref.gauss_logpdf(prop, mu, Q) - gmrf_ref.gauss_logpdf(x, mu, Q)
    accepted = bool(np.log(u) < log_alpha)
    return (prop if accepted else x), accepted, {"prop": prop, "log_alpha": log_alpha}
