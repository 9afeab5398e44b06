"""One tridiagonal draw with a per-chain prior mean at the headline size (n = 10 000, 1024 chains), timed on the device:
the product vector and the residual formed inside the launch (omc_tridiag_terms.center_chain) against the route it replaces
(omc_tridiag_matvec_chain -> rhs_chain, omc_chain_lincomb + omc_tridiag_quadform), and the plain generic draw for scale."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from openmcmc_amd.engine import Engine

    n, C, reps = int(os.environ.get("N", 10000)), int(os.environ.get("C", 1024)), 50
    rng = np.random.default_rng(0)
    eng = Engine(C, seed=1)
    d = np.full(n, 2.0); d[0] = d[-1] = 1.0; d[0] += 1e-3
    off = -np.ones(n - 1)
    y = rng.standard_normal(n) + 2
    d_d, d_o, d_y = eng.to_device(d), eng.to_device(off), eng.to_device(y)
    lam, tau = eng.full((C,), 50.0), eng.full((C,), 1.0)
    m = eng.to_device(1.0 + 0.1 * rng.standard_normal((C, n)))
    x, quad, q1 = eng.empty(C, n), eng.empty(2, C), eng.empty(1, C)
    lik = {"rhs": d_y, "center": d_y, "scale": tau}
    plain = eng.tridiag_terms([{"diag": d_d, "off": d_o, "scale": lam}, lik], n)
    incl = eng.tridiag_terms([{"diag": d_d, "off": d_o, "scale": lam, "center_chain": m}, lik], n)
    unit = eng.tridiag_terms([{"diag": d_d, "off": d_o}], n)
    eng.set_option("tridiag_generic", 1)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return 1e3 * a.elapsed_time(b) / reps

    def old_route():
        rc = eng.tridiag_matvec_chain(n, d_d, d_o, m, scale=lam)
        eng.tridiag_sample_canonical(n, plain, x, rhs_chain=rc, draw_index=1)
        r = eng.chain_lincomb(1.0, x, -1.0, m)
        eng.tridiag_quadform(n, unit, r, q1)

    print("plain generic draw, no quadratic forms     %7.1f us" % timed(lambda: eng.tridiag_sample_canonical(n, plain, x, draw_index=1)))
    print("plain generic draw + fused quadratic forms %7.1f us" % timed(lambda: eng.tridiag_sample_canonical(n, plain, x, draw_index=1, quad_out=quad)))
    print("per-chain centre inside the launch + quad  %7.1f us" % timed(lambda: eng.tridiag_sample_canonical(n, incl, x, draw_index=1, quad_out=quad)))
    print("per-chain centre inside the launch, no quad%7.1f us" % timed(lambda: eng.tridiag_sample_canonical(n, incl, x, draw_index=1)))
    print("the route it replaces (4 launches)         %7.1f us" % timed(old_route))
    eng.set_option("tridiag_generic", 0)  # the shifted smoother (SIG 3) takes this structure
    print("shifted smoother (SIG 3) + quad            %7.1f us" % timed(lambda: eng.tridiag_sample_canonical(n, incl, x, draw_index=1, quad_out=quad)))
    print("shifted smoother (SIG 3), no quad          %7.1f us" % timed(lambda: eng.tridiag_sample_canonical(n, incl, x, draw_index=1)))
    print("plain specialised draw (SIG 1) + quad      %7.1f us" % timed(lambda: eng.tridiag_sample_canonical(n, plain, x, draw_index=1, quad_out=quad)))
    eng.check_status()


if __name__ == "__main__":
    main()
