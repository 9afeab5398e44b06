#!/usr/bin/env python3
"""Randomised cross-check of the structure-specialised sweep kernel against the generic instantiation: random
chain lengths (both segment widths, partial last waves), chain counts, term orders, offsets; two fused sweeps each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main(trials=60, seed=0):
    from openmcmc_amd.engine import Engine
    rng = np.random.default_rng(seed)
    worst = 0.0
    for t in range(trials):
        n = int(rng.integers(4097, 10241))
        C = int(rng.integers(1, 9))
        p_first = bool(rng.integers(0, 2))
        with_offsets = bool(rng.integers(0, 2))
        pd = np.full(n, 2.0); pd[0] = pd[-1] = 1.0; pd[0] += 1e-3; pd *= 1 + 0.1 * rng.random(n)
        po = -np.ones(n - 1)
        y = rng.standard_normal(n) + 2
        extra = 0.3 * rng.standard_normal((C, n))
        out = []
        for generic in (0, 1):
            eng = Engine(C, seed=1234 + t)
            eng.set_option("tridiag_generic", generic)
            d_pd, d_po, d_y = eng.to_device(pd), eng.to_device(po), eng.to_device(y)
            lam, tau = eng.to_device(80 + np.arange(C) * 0.5), eng.full((C,), 1.25)
            tp = {"diag": d_pd, "off": d_po, "scale": lam}
            ti = {"rhs": d_y, "center": d_y, "scale": tau}
            terms = eng.tridiag_terms([tp, ti] if p_first else [ti, tp], n)
            ldP, ldI = eng.tridiag_logdet(n, d_pd, d_po), eng.zeros(1)
            bp = {"a0": 10.0, "b0": 1.0, "n_pos": n, "logdet": ldP}
            bi = {"a0": 1.0, "b0": 1.0, "n_pos": n, "logdet": ldI}
            x, lp = eng.empty(C, n), eng.empty(C)
            for it in range(2):
                eng.gmrf_sweep(n, terms, [bp, bi] if p_first else [bi, bp], x, rhs_chain=eng.to_device(extra) if with_offsets else None,
                               draw_index=3 * it, log_post_out=lp, gamma_draw_base=3 * it + 1)
            eng.check_status()
            out.append([v.cpu().numpy().copy() for v in (x, lam, tau, lp)])
            eng.close()
        for a, b in zip(out[0], out[1]):
            assert np.all(np.isfinite(a)), (n, C, p_first, with_offsets)
            err = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
            worst = max(worst, err)
            assert err < 1e-10, (n, C, p_first, with_offsets, err)
    print(f"{trials} trials, worst relative difference {worst:.2e}")


if __name__ == "__main__":
    main()
