#!/usr/bin/env python3
"""Randomised cross-check of the shifted smoother (tridiagonal term centred at a per-chain vector + scaled identity, SIG 3)
against the generic instantiation, with the draws GENERATED in the kernel (the parked-slice paths are taken only then):
random chain lengths (both segment widths, partial last waves), chain counts, with and without a shared centre of the
identity term, with and without quadratic forms."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main(trials=60, seed=0):
    from openmcmc_amd.engine import Engine

    rng = np.random.default_rng(seed)
    worst = 0.0
    for t in range(trials):
        n = int(rng.integers(4097, 10241)) if t % 4 else 10000
        C = int(rng.integers(1, 9)) if t % 4 else 300
        shared_centre = bool(rng.integers(0, 2))
        want_quad = bool(rng.integers(0, 2))
        pd = np.full(n, 2.0); pd[0] = pd[-1] = 1.0; pd *= 1 + 0.1 * rng.random(n)
        po = -np.ones(n - 1)
        y = rng.standard_normal(n) + 2
        m = 0.3 * rng.standard_normal((C, n)) + 1.0
        lam, tau = 20 + 50 * rng.random(C), 0.5 + rng.random(C)
        out = []
        for generic in (0, 1):
            eng = Engine(C, seed=1234 + t)
            eng.set_option("tridiag_generic", generic)
            t_prior = {"diag": eng.to_device(pd), "off": eng.to_device(po), "scale": eng.to_device(lam), "center_chain": eng.to_device(m)}
            t_lik = {"scale": eng.to_device(tau)}
            if shared_centre:
                t_lik.update(rhs=eng.to_device(y), center=eng.to_device(y))
            x, quad = eng.empty(C, n), (eng.empty(2, C) if want_quad else None)
            for rep in range(3):
                eng.tridiag_sample_canonical(n, [t_prior, t_lik], x, draw_index=7 + rep, quad_out=quad)
            eng.check_status()
            out.append((x.cpu().numpy(), None if quad is None else quad.cpu().numpy()))
            eng.close()
        (x0, q0), (x1, q1) = out
        e = np.abs(x0 - x1).max() / np.abs(x1).max()
        if q0 is not None:
            e = max(e, (np.abs(q0 - q1) / np.abs(q1)).max())
        worst = max(worst, e)
        if not e < 1e-10:
            print(f"trial {t}: n={n} C={C} shared_centre={shared_centre} quad={want_quad}: relative difference {e:.3e}")
    print(f"{trials} trials, worst relative difference {worst:.2e}")
    return 0 if worst < 1e-10 else 1


if __name__ == "__main__":
    sys.exit(main(*(int(a) for a in sys.argv[1:])))
