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

// Phi^-1(p), 0 < p < 1: Wichura's PPND16 (Algorithm AS 241, Appl. Statist. 37 (1988) 477-484; the routine behind R's
// qnorm), rational approximations good to about 1e-16 relative.  Written out because the scan's cost per site is the
// length of ONE dependent instruction chain: this is ~35 instructions in the centre (85 % of the draws) and ~90 in the
// tails against the several hundred of the library's normcdfinv.  Host-checkable (tests/native).
__host__ __device__ inline double omc_ndtri_as241(double p) {
  const double q = p - 0.5;
  if (fabs(q) <= 0.425) {
    const double r = 0.180625 - q * q;
    const double num = (((((((2.5090809287301226727e+3 * r + 3.3430575583588128105e+4) * r + 6.7265770927008700853e+4) * r +
                            4.5921953931549871457e+4) * r + 1.3731693765509461125e+4) * r + 1.9715909503065514427e+3) * r +
                          1.3314166789178437745e+2) * r + 3.3871328727963666080e0);
    const double den = (((((((5.2264952788528545610e+3 * r + 2.8729085735721942674e+4) * r + 3.9307895800092710610e+4) * r +
                            2.1213794301586595867e+4) * r + 5.3941960214247511077e+3) * r + 6.8718700749205790830e+2) * r +
                          4.2313330701600911252e+1) * r + 1.0);
    return q * num * omc_rcp_nr(den);  // den > 1: the refined hardware reciprocal, not the ~35-instruction divide
  }
  double r = sqrt(-log(q < 0.0 ? p : 1.0 - p));
  double x;
  if (r <= 5.0) {
    r -= 1.6;
    const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r +
                            1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r +
                          4.63033784615654529590e0) * r + 1.42343711074968357734e0);
    const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r +
                            1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r +
                          2.05319162663775882187e0) * r + 1.0);
    x = num * omc_rcp_nr(den);
  } else {
    r -= 5.0;
    const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r +
                            2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r +
                          5.46378491116411436990e0) * r + 6.65790464350110377720e0);
    const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r +
                            7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                          5.99832206555887937690e-1) * r + 1.0);
    x = num * omc_rcp_nr(den);
  }
  return q < 0.0 ? -x : x;
}

// The same function with both of its common branches evaluated and one selected: a wave of 64 uniforms practically
// always holds a lane outside the central region, so the branches cost their sum anyway, and without them several
// quantiles can be worked out side by side in one basic block (the caller's way of hiding the latency of these long
// dependent chains).  Same rational approximations, same results.
__device__ inline double omc_ndtri_as241_nb(double p) {
  const double q = p - 0.5;
  const double rc = 0.180625 - q * q;
  const double numc = (((((((2.5090809287301226727e+3 * rc + 3.3430575583588128105e+4) * rc + 6.7265770927008700853e+4) * rc +
                           4.5921953931549871457e+4) * rc + 1.3731693765509461125e+4) * rc + 1.9715909503065514427e+3) * rc +
                         1.3314166789178437745e+2) * rc + 3.3871328727963666080e0);
  const double denc = (((((((5.2264952788528545610e+3 * rc + 2.8729085735721942674e+4) * rc + 3.9307895800092710610e+4) * rc +
                           2.1213794301586595867e+4) * rc + 5.3941960214247511077e+3) * rc + 6.8718700749205790830e+2) * rc +
                         4.2313330701600911252e+1) * rc + 1.0);
  const double xc = q * numc * omc_rcp_nr(denc);
  // p in (1e-15, 1 - 1e-15): the argument is a finite positive normal number <= 1/2 -- the library's branch-free
  // logarithm kernel and refined reciprocal square root apply (the libm routines carry special-case branches)
  const double rt = omc_sqrt_nr(-omc_log_unit(q < 0.0 ? p : 1.0 - p));
  double xm, xf;
  {
    const double r = rt - 1.6;
    const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r +
                            1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r +
                          4.63033784615654529590e0) * r + 1.42343711074968357734e0);
    const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r +
                            1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r +
                          2.05319162663775882187e0) * r + 1.0);
    xm = num * omc_rcp_nr(den);
  }
  {
    const double r = rt - 5.0;
    const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r +
                            2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r +
                          5.46378491116411436990e0) * r + 6.65790464350110377720e0);
    const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r +
                            7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                          5.99832206555887937690e-1) * r + 1.0);
    xf = num * omc_rcp_nr(den);
  }
  const double xt = (rt <= 5.0) ? xm : xf;
  return (fabs(q) <= 0.425) ? xc : (q < 0.0 ? -xt : xt);
}

// ---- four quantiles side by side --------------------------------------------------------------------------------
// omc_ndtri_as241_nb on a vector of four: every statement turns into four independent instructions back to back, which is
// what a wave that is alone on its SIMD needs to keep issuing while each of the four long dependent chains (two Horner
// chains, a logarithm, a square root, three refined reciprocals) waits for itself.  Same operations per element as the
// scalar function (up to where the compiler fuses a multiply into an add: <= 1 ulp in the logarithm).
template <int N> using omc_dv = double __attribute__((ext_vector_type(N)));
#define OMC_DV(x) (omc_dv<N>(x))   /* broadcast */
template <int N> __device__ __forceinline__ omc_dv<N> omc_fmav(omc_dv<N> a, omc_dv<N> b, omc_dv<N> c) { return __builtin_elementwise_fma(a, b, c); }
template <int N> __device__ __forceinline__ omc_dv<N> omc_rcp_nrv(omc_dv<N> d) {
  omc_dv<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_amdgcn_rcp(d[i]);
  const omc_dv<N> e = omc_fmav<N>(-d, r, OMC_DV(1.0));
  return omc_fmav<N>(r, omc_fmav<N>(e, e, e), r);
}
template <int N> __device__ __forceinline__ omc_dv<N> omc_sqrt_nrv(omc_dv<N> r) {
  omc_dv<N> g;
#pragma unroll
  for (int i = 0; i < N; ++i) g[i] = __builtin_amdgcn_rsq(r[i]);
  omc_dv<N> s = r * g;
  const omc_dv<N> h = OMC_DV(0.5) * g;
  omc_dv<N> e = omc_fmav<N>(-s, s, r);
  s = omc_fmav<N>(e, h, s);
  e = omc_fmav<N>(-s, s, r);
  return omc_fmav<N>(e, h, s);
}
template <int N> __device__ __forceinline__ omc_dv<N> omc_log_unitv(omc_dv<N> u) {  // omc_log_unit, element by element
  omc_dv<N> m, dk;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int k = __builtin_amdgcn_frexp_exp(u[i]);
    double mi = __builtin_amdgcn_frexp_mant(u[i]);
    const bool low = mi < 0.70710678118654752440;
    m[i] = low ? mi * 2.0 : mi;
    dk[i] = (double)(low ? k - 1 : k);
  }
  const omc_dv<N> f = m - OMC_DV(1.0);
  const omc_dv<N> s = f * omc_rcp_nrv<N>(OMC_DV(2.0) + f);
  const omc_dv<N> z = s * s, w = z * z;
  const omc_dv<N> t1 = w * omc_fmav<N>(w, omc_fmav<N>(w, OMC_DV(1.531383769920937332e-01), OMC_DV(2.222219843214978396e-01)),
                                 OMC_DV(3.999999999940941908e-01));
  const omc_dv<N> t2 = z * omc_fmav<N>(w, omc_fmav<N>(w, omc_fmav<N>(w, OMC_DV(1.479819860511658591e-01), OMC_DV(1.818357216161805012e-01)),
                                             OMC_DV(2.857142874366239149e-01)), OMC_DV(6.666666666666735130e-01));
  const omc_dv<N> R = t2 + t1, hfsq = OMC_DV(0.5) * f * f;
  return dk * OMC_DV(6.93147180369123816490e-01) -
         ((hfsq - omc_fmav<N>(s, hfsq + R, dk * OMC_DV(1.90821492927058770002e-10))) - f);
}
#define OMC_HORNER8(r, c7, c6, c5, c4, c3, c2, c1, c0)                                                                 \
  omc_fmav<N>(omc_fmav<N>(omc_fmav<N>(omc_fmav<N>(omc_fmav<N>(omc_fmav<N>(omc_fmav<N>(OMC_DV(c7), r, OMC_DV(c6)), r, OMC_DV(c5)), r, OMC_DV(c4)), r, \
                                      OMC_DV(c3)), r, OMC_DV(c2)), r, OMC_DV(c1)), r, OMC_DV(c0))
template <int N> __device__ __forceinline__ omc_dv<N> omc_ndtri_as241_nbv(omc_dv<N> p) {
  const omc_dv<N> q = p - OMC_DV(0.5);
  const omc_dv<N> rc = omc_fmav<N>(-q, q, OMC_DV(0.180625));
  const omc_dv<N> numc = OMC_HORNER8(rc, 2.5090809287301226727e+3, 3.3430575583588128105e+4, 6.7265770927008700853e+4,
                                  4.5921953931549871457e+4, 1.3731693765509461125e+4, 1.9715909503065514427e+3,
                                  1.3314166789178437745e+2, 3.3871328727963666080e0);
  const omc_dv<N> denc = OMC_HORNER8(rc, 5.2264952788528545610e+3, 2.8729085735721942674e+4, 3.9307895800092710610e+4,
                                  2.1213794301586595867e+4, 5.3941960214247511077e+3, 6.8718700749205790830e+2,
                                  4.2313330701600911252e+1, 1.0);
  const omc_dv<N> xc = q * numc * omc_rcp_nrv<N>(denc);
  omc_dv<N> tail;
#pragma unroll
  for (int i = 0; i < N; ++i) tail[i] = q[i] < 0.0 ? p[i] : 1.0 - p[i];
  const omc_dv<N> rt = omc_sqrt_nrv<N>(-omc_log_unitv<N>(tail));
  const omc_dv<N> rm = rt - OMC_DV(1.6), rf = rt - OMC_DV(5.0);
  const omc_dv<N> numm = OMC_HORNER8(rm, 7.74545014278341407640e-4, 2.27238449892691845833e-2, 2.41780725177450611770e-1,
                                  1.27045825245236838258e0, 3.64784832476320460504e0, 5.76949722146069140550e0,
                                  4.63033784615654529590e0, 1.42343711074968357734e0);
  const omc_dv<N> denm = OMC_HORNER8(rm, 1.05075007164441684324e-9, 5.47593808499534494600e-4, 1.51986665636164571966e-2,
                                  1.48103976427480074590e-1, 6.89767334985100004550e-1, 1.67638483018380384940e0,
                                  2.05319162663775882187e0, 1.0);
  const omc_dv<N> numf = OMC_HORNER8(rf, 2.01033439929228813265e-7, 2.71155556874348757815e-5, 1.24266094738807843860e-3,
                                  2.65321895265761230930e-2, 2.96560571828504891230e-1, 1.78482653991729133580e0,
                                  5.46378491116411436990e0, 6.65790464350110377720e0);
  const omc_dv<N> denf = OMC_HORNER8(rf, 2.04426310338993978564e-15, 1.42151175831644588870e-7, 1.84631831751005468180e-5,
                                  7.86869131145613259100e-4, 1.48753612908506148525e-2, 1.36929880922735805310e-1,
                                  5.99832206555887937690e-1, 1.0);
  const omc_dv<N> xm = numm * omc_rcp_nrv<N>(denm), xf = numf * omc_rcp_nrv<N>(denf);
  omc_dv<N> out;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double xt = (rt[i] <= 5.0) ? xm[i] : xf[i];
    out[i] = (fabs(q[i]) <= 0.425) ? xc[i] : (q[i] < 0.0 ? -xt : xt);
  }
  return out;
}

__device__ inline double omc_truncnorm_ppf(double u, double a, double b) {
  // Both bounds far out: Phi(x) = u + Phi(a)(1 - u) - u Phi(-b) with Phi(a), Phi(-b) < 1e-38, so for u in
  // [1e-15, 1 - 1e-15] (the 2^-53 grid of the in-kernel uniforms lies inside, all but its end points) the window
  // moves either tail probability by less than 1e-23 of itself: x = Phi^-1(u) to rounding.  This is the common case
  // of a scan whose sites sit many conditional standard deviations inside their limits (a smoother under a positivity
  // constraint), and it costs one rational approximation instead of ten transcendental evaluations.
  if (a < -13.0 && b > 13.0 && u > 1e-15 && u < 1.0 - 1e-15) return omc_ndtri_as241(u);
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
// the same with 1/scale supplied (no division on the caller's serial path); an infinite limit stays infinite
__device__ inline double omc_truncated_normal_rv_inv(double mean, double scale, double inv_scale, double lower, double upper, double u) {
  const double a = (lower - mean) * inv_scale, b = (upper - mean) * inv_scale;
  return omc_truncnorm_ppf(u, a, b) * scale + mean;
}

// gmrf.truncated_normal_log_pdf (gmrf.py:295-318): -inf outside [lower, upper]
__device__ inline double omc_truncated_normal_log_pdf(double x, double mean, double scale, double lower, double upper) {
  const double a = (lower - mean) / scale, b = (upper - mean) / scale;
  const double t = (x - mean) / scale;
  if (!(t >= a && t <= b)) return -INFINITY;
  return -0.5 * t * t - OMC_LOG_SQRT_2PI - omc_log_gauss_mass(a, b) - log(scale);
}
