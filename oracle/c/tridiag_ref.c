/* CPU oracle, plain C: TEST INFRASTRUCTURE ONLY (checker and cpu_baseline leg of bench.py).
 *
 * Sequential restatement of gmrf.sample_normal_canonical (gmrf.py:167-198) for a tridiagonal
 * precision Q = lam*P + tau*I with rhs b = lam*P*mu + tau*y, i.e. the NormalNormal update of the
 * example-4 model (sampler.py:176-197), followed by the two NormalGamma sufficient statistics
 * (sampler.py:282-284).  The factorisation is the unpivoted LU SuperLU performs for
 * gmrf.sparse_cholesky (gmrf.py:514-516): pivots D_i = U_ii, L = L_lu*diag(sqrt(D)).
 * Pinned against the reference's golden vectors by tests/test_oracle_golden.py::test_c_oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* returns 0, or 1+i for a non-positive pivot at node i (gmrf.py:515-518 would fall back / raise) */
int omc_ref_tridiag_draw(int64_t n, const double* p_diag, const double* p_off, double lam, double tau,
                         const double* y, const double* mu, const double* z, double* x, double* quad,
                         double* logdet, double* work /* n doubles */) {
  double lprev = 0.0, bprev = 0.0, u = 0.0, ld = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double a = lam * p_diag[i] + tau;
    const double b = (i < n - 1) ? lam * p_off[i] : 0.0;
    /* (P mu)_i */
    double pm = p_diag[i] * mu[i];
    if (i > 0) pm += p_off[i - 1] * mu[i - 1];
    if (i < n - 1) pm += p_off[i] * mu[i + 1];
    const double r = lam * pm + tau * y[i];
    const double D = a - lprev * bprev; /* U_ii of the LU, gmrf.py:514 */
    if (!(D > 0.0)) return (int)(1 + i);
    u = r - lprev * u;               /* forward solve L w = b (gmrf.py:459), w_i = u_i/sqrt(D_i) */
    const double d = sqrt(D);        /* L_ii, gmrf.py:516 */
    x[i] = (u / d + z[i]) / d;       /* (w + z)_i / L_ii : rhs of the backward solve (gmrf.py:460, 61) */
    ld += log(d);
    lprev = b / D;
    work[i] = lprev;
    bprev = b;
  }
  double xn = 0.0, q0 = 0.0, q1 = 0.0, r0n = 0.0;
  for (int64_t i = n - 1; i >= 0; --i) {
    xn = x[i] - work[i] * xn;        /* backward solve L' x = w + z */
    x[i] = xn;
    const double r0 = xn - mu[i], r1 = y[i] - xn;
    q0 += p_diag[i] * r0 * r0 + ((i < n - 1) ? 2.0 * p_off[i] * r0 * r0n : 0.0); /* r'Pr, sampler.py:284 */
    q1 += r1 * r1;
    r0n = r0;
  }
  quad[0] = q0;
  quad[1] = q1;
  *logdet = 2.0 * ld;                /* gmrf.py:342 */
  return 0;
}

/* log det of a tridiagonal SPD matrix by the same pivots (gmrf.py:342) */
int omc_ref_tridiag_logdet(int64_t n, const double* diag, const double* off, double* logdet) {
  double D = 0.0, bprev = 0.0, ld = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    D = (i == 0) ? diag[0] : diag[i] - bprev * bprev / D;
    if (!(D > 0.0)) return (int)(1 + i);
    ld += log(D);
    bprev = (i < n - 1) ? off[i] : 0.0;
  }
  *logdet = ld;
  return 0;
}
