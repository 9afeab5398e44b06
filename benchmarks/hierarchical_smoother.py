"""The hierarchical version of the headline model at the headline size: y ~ N(b, (tau I)^-1), b ~ N(m, (lambda P)^-1) with a
first-order random-walk P, m ~ N(0, (kappa I)^-1); samplers [NormalNormal(b), NormalNormal(m), NormalGamma(lambda),
NormalGamma(tau)], n = 10 000, 1024 chains, through MCMC.run_mcmc (two Normal-Normal blocks: the sweep is issued sampler by
sampler; each block's tridiagonal draw is centred at the other block's state, omc_tridiag_terms.center_chain).  Prints the
time per sweep of the repeated run_mcmc calls (steady state) and of the first call (set-up included: what round 2 quoted)."""
import os
import sys
import time

import numpy as np
from scipy import sparse

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    n, C = int(os.environ.get("N", 10000)), int(os.environ.get("C", 1024))
    n_burn, n_iter = int(os.environ.get("BURN", 10)), int(os.environ.get("ITER", 20))
    rng = np.random.default_rng(2)
    t = np.arange(n) * 60.0 / n
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + 0.3 * rng.standard_normal(n)
    D = sparse.diags([-np.ones(n - 1), np.ones(n - 1)], offsets=[0, 1], shape=(n - 1, n))
    P = (D.T @ D + 1e-3 * sparse.identity(n)).tocsc()
    mdl = Model([
        Normal("y", mean="b", precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
        Normal("b", mean="m", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
        Normal("m", mean="m0", precision="P_m"),
        Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
    state = {"y": y, "b": y.copy(), "m": np.full(n, 1.0), "m0": np.zeros(n), "P_m": sparse.identity(n, format="csc") * 0.5,
             "lambda": 50.0, "P_lambda": P, "a_lam": 10.0, "b_lam": 1.0, "tau": 1.0, "P_tau": sparse.identity(n, format="csc"),
             "a_tau": 1.0, "b_tau": 1.0}
    samplers = [NormalNormal("b", mdl), NormalNormal("m", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
    M = MCMC(state, samplers, model=mdl, n_burn=n_burn, n_iter=n_iter, n_chains=C, seed=3)
    if os.environ.get("GENERIC"):  # A/B: the generic instantiation instead of the shifted smoother
        M.engine.set_option("tridiag_generic", 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    M.run_mcmc()  # the first call of a fresh MCMC object: plans, device caches, first touch of the stores, clocks
    torch.cuda.synchronize()
    first = (time.perf_counter() - t0) / (n_burn + n_iter)
    M.engine.check_status()
    best = float("inf")
    for _ in range(int(os.environ.get("REPEATS", 3))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        M.run_mcmc()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (n_burn + n_iter))
    dt = best
    M.engine.check_status()
    out = M.collect()
    print(f"n={n} C={C}: {1e3 * dt:.2f} ms per sweep = {C / dt:.0f} chain-updates/s (best of the repeated run_mcmc calls; the first call, "
          f"set-up included, {1e3 * first:.2f} ms per sweep over its {n_burn + n_iter} sweeps); all finite: "
          f"{bool(np.isfinite(out['b']).all() and np.isfinite(out['m']).all())}; tau mean {out['tau'].mean():.2f}, lambda mean {out['lambda'].mean():.1f}")


if __name__ == "__main__":
    main()
