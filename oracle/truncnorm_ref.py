"""Oracle for the truncated-normal draws and densities the reference takes from SciPy
(reference: gmrf.py:269-318 -> scipy.stats.truncnorm, SciPy 1.15: `_continuous_distns.py`
`_log_gauss_mass`, `truncnorm_gen._ppf/_logpdf`; third-party arithmetic, not under /root/reference).
TEST INFRASTRUCTURE ONLY.

SciPy's algorithm, restated for scalars/arrays with the uniform draw `u` as an argument
(truncnorm.rvs(a, b, loc, scale) == truncnorm.ppf(U, a, b)*scale + loc):

  log mass(a, b):   b <= 0          log(Phi(b) - Phi(a))      in log space from log_ndtr
                    a  > 0          the same for (-b, -a)
                    a <= 0 < b      log1p(-Phi(a) - Phi(-b))
  ppf(u; a, b):     a < 0           ndtri_exp(logaddexp(log Phi(a),  log u      + log mass))
                    a >= 0         -ndtri_exp(logaddexp(log Phi(-b), log1p(-u)  + log mass))
  logpdf(x):        -x^2/2 - log sqrt(2 pi) - log mass - log scale on [a, b], -inf outside
"""

import numpy as np
from scipy import special

_LOG_SQRT_2PI = np.log(np.sqrt(2 * np.pi))


def _log_diff(log_p, log_q):
    """log(exp(log_p) - exp(log_q)) for log_p >= log_q (SciPy: logsumexp with a +pi*i phase)."""
    m = np.maximum(log_p, log_q)
    with np.errstate(divide="ignore", invalid="ignore"):
        return m + np.log(np.exp(log_p - m) - np.exp(log_q - m))


def _log_sum(log_p, log_q):
    """log(exp(log_p) + exp(log_q)) the way SciPy forms it (special.logsumexp over the pair)."""
    return special.logsumexp([log_p, log_q], axis=0)


def log_gauss_mass(a, b):
    a, b = np.broadcast_arrays(np.asarray(a, dtype=float), np.asarray(b, dtype=float))
    out = np.empty(a.shape)
    left, right = b <= 0, a > 0
    central = ~(left | right)
    out[left] = _log_diff(special.log_ndtr(b[left]), special.log_ndtr(a[left]))
    out[right] = _log_diff(special.log_ndtr(-a[right]), special.log_ndtr(-b[right]))
    out[central] = np.log1p(-special.ndtr(a[central]) - special.ndtr(-b[central]))
    return out


def truncnorm_ppf(u, a, b):
    u, a, b = np.broadcast_arrays(np.asarray(u, dtype=float), np.asarray(a, dtype=float), np.asarray(b, dtype=float))
    mass = log_gauss_mass(a, b)
    out = np.empty(u.shape)
    lft = a < 0
    with np.errstate(divide="ignore"):
        out[lft] = special.ndtri_exp(_log_sum(special.log_ndtr(a[lft]), np.log(u[lft]) + mass[lft]))
        rgt = ~lft
        out[rgt] = -special.ndtri_exp(_log_sum(special.log_ndtr(-b[rgt]), np.log1p(-u[rgt]) + mass[rgt]))
    return out


def truncated_normal_rv(mean, scale, lower, upper, u):
    """gmrf.truncated_normal_rv with the uniform supplied  [gmrf.py:269-292]."""
    lower = -np.inf if lower is None else lower
    upper = np.inf if upper is None else upper
    a, b = (lower - mean) / scale, (upper - mean) / scale
    return truncnorm_ppf(u, a, b) * scale + mean


def truncated_normal_log_pdf(x, mean, scale, lower, upper):
    """gmrf.truncated_normal_log_pdf  [gmrf.py:295-318]."""
    lower = -np.inf if lower is None else lower
    upper = np.inf if upper is None else upper
    a, b = (lower - mean) / scale, (upper - mean) / scale
    t = (np.asarray(x, dtype=float) - mean) / scale
    lp = -0.5 * t * t - _LOG_SQRT_2PI - log_gauss_mass(a, b) - np.log(scale)
    return np.where((t >= a) & (t <= b), lp, -np.inf)
