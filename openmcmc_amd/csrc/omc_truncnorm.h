// Truncated-normal inverse CDF and log-density on the device (fp64).  Not part of the ABI.
//
// The reference takes these from scipy.stats.truncnorm (gmrf.py:269-318).  SciPy's log-space
// formulas are restated here in a form that never subtracts nearly equal probabilities, so the
// result is the exactly rounded quantity to a few ulp wherever the problem is well conditioned:
//
//   window [a, b] in standard units, uniform u:
//     log Phi(x)      = logaddexp(log(1-u) + log Phi(a),  log u + log Phi(b))        (both terms > 0)
//     log(1 - Phi(x)) = logaddexp(log(1-u) + log Phi(-a), log u + log Phi(-b))
//   the smaller of the two tails is inverted (Newton on log Phi with the Mills ratio, started from
//   normcdfinv or the tail asymptote), which keeps the inversion away from Phi -> 1.
//
//   log mass(a, b) = log(Phi(b) - Phi(a)):
//     b <= 0:  log Phi(b) + log(-expm1(log Phi(a) - log Phi(b)))     (mirror for a >= 0)
//     a < 0 < b:  log(0.5 * (erf(b/sqrt2) + erf(-a/sqrt2)))           (a sum of positive terms)
#pragma once
#include <math.h>

#define OMC_RSQRT2 0.70710678118654752440
#define OMC_LOG_SQRT_2PI 0.91893853320467274178

__device__ inline double omc_log_ndtr(double t) {  // log Phi(t), any t including +-inf
  if (t > 0.0) return log1p(-0.5 * erfc(t * OMC_RSQRT2));
  if (t > -20.0) return log(0.5 * erfc(-t * OMC_RSQRT2));
  if (t == -INFINITY) return -INFINITY;
  return log(0.5 * erfcx(-t * OMC_RSQRT2)) - 0.5 * t * t;
}

__device__ inline double omc_logaddexp(double p, double q) {
  const double m = fmax(p, q), l = fmin(p, q);
  if (m == -INFINITY) return -INFINITY;
  return m + log1p(exp(l - m));
}

// x <= ~0 with log Phi(x) = y  (y <= log 0.5 up to rounding)
__device__ inline double omc_ndtri_exp_lower(double y) {
  if (y == -INFINITY) return -INFINITY;
  double x;
  int iters;
  if (y > -600.0) {
    x = normcdfinv(exp(y));
    iters = 1;  // normcdfinv is already good to a few ulp; one Newton step removes the rounding of exp(y) for large |y|
  } else {  // Phi(x) ~ phi(x)/|x|:  x^2 = -2y - log(2 pi) - log(x^2)
    const double r = -2.0 * y - 1.8378770664093453;
    x = -sqrt(r - log(r));
    iters = 4;
  }
  for (int i = 0; i < iters; ++i) {
    // Newton on f(x) = log Phi(x) - y, f' = phi/Phi = 1/Mills;  Mills(x) = sqrt(pi/2) erfcx(-x/sqrt2)
    const double mills = 1.2533141373155003 * erfcx(-x * OMC_RSQRT2);
    x -= (omc_log_ndtr(x) - y) * mills;
  }
  return x;
}

__device__ inline double omc_truncnorm_ppf(double u, double a, double b) {
  const double l1 = log1p(-u), l0 = log(u);
  // log Phi(x); the two tails satisfy exp(yp) + exp(yq) = 1, so "yp is the smaller one" is yp <= log(1/2) and the
  // upper tail is only worked out when it is the one to invert
  const double yp = omc_logaddexp(l1 + omc_log_ndtr(a), l0 + omc_log_ndtr(b));
  double x;
  if (yp <= -0.69314718055994530942) {
    x = omc_ndtri_exp_lower(yp);
  } else {
    const double yq = omc_logaddexp(l1 + omc_log_ndtr(-a), l0 + omc_log_ndtr(-b));
    x = -omc_ndtri_exp_lower(yq);
  }
  return fmin(fmax(x, a), b);
}

__device__ inline double omc_log_gauss_mass(double a, double b) {
  if (b <= 0.0) {
    const double lb = omc_log_ndtr(b);
    return lb + log(-expm1(omc_log_ndtr(a) - lb));
  }
  if (a >= 0.0) {
    const double la = omc_log_ndtr(-a);
    return la + log(-expm1(omc_log_ndtr(-b) - la));
  }
  return log(0.5 * (erf(b * OMC_RSQRT2) + erf(-a * OMC_RSQRT2)));
}

// gmrf.truncated_normal_rv with the uniform supplied (gmrf.py:269-292)
__device__ inline double omc_truncated_normal_rv(double mean, double scale, double lower, double upper, double u) {
  const double a = (lower - mean) / scale, b = (upper - mean) / scale;
  return omc_truncnorm_ppf(u, a, b) * scale + mean;
}

// gmrf.truncated_normal_log_pdf (gmrf.py:295-318): -inf outside [lower, upper]
__device__ inline double omc_truncated_normal_log_pdf(double x, double mean, double scale, double lower, double upper) {
  const double a = (lower - mean) / scale, b = (upper - mean) / scale;
  const double t = (x - mean) / scale;
  if (!(t >= a && t <= b)) return -INFINITY;
  return -0.5 * t * t - OMC_LOG_SQRT_2PI - omc_log_gauss_mass(a, b) - log(scale);
}
