"""A second-order random-walk smoother next to a regression block, y ~ N(b + X beta, (tau I)^-1): b (n = 10 000) on the band
route with beta's part of the mean as a per-chain offset, beta (p = 20) on the dense route with b as its offset, NormalGamma
on lambda and tau; 1024 chains through MCMC.run_mcmc.  Prints the time per sweep and the counters of the segmented band route (a throughput measurement: the chains are not run to convergence)."""
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
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    n, p, C = int(os.environ.get("N", 10000)), int(os.environ.get("P", 20)), int(os.environ.get("C", 1024))
    n_burn, n_iter = int(os.environ.get("BURN", 10)), int(os.environ.get("ITER", 20))
    rng = np.random.default_rng(1)
    t = np.linspace(0, 1, n)
    X = np.stack([np.cos(2 * np.pi * (k + 1) * t * 40) for k in range(p)], 1)  # fast oscillations: not in the smoother's reach
    beta = rng.standard_normal(p)
    smooth = np.sin(2 * np.pi * t) + 0.5 * t
    y = smooth + X @ beta + 0.1 * rng.standard_normal(n)
    D = sparse.diags([np.ones(n - 2), -2 * np.ones(n - 2), np.ones(n - 2)], offsets=[0, 1, 2], shape=(n - 2, n))
    P = (D.T @ D + 1e-4 * sparse.identity(n)).tocsc()
    mdl = Model([
        Normal("y", mean=LinearCombination(form={"b": "A", "beta": "X"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
        Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
        Normal("beta", mean="mu_beta", precision="P_beta"),
        Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
    state = {"y": y, "b": y.copy(), "mu": np.zeros(n), "lambda": float(os.environ.get("LAMBDA0", 100.0)), "P_lambda": P, "a_lam": 1.0, "b_lam": 1e-6, "tau": 1.0,
             "P_tau": sparse.identity(n, format="csc"), "a_tau": 1.0, "b_tau": 1.0, "A": sparse.identity(n, format="csc"), "X": X,
             "beta": np.zeros(p), "mu_beta": np.zeros(p), "P_beta": sparse.identity(p, format="csc") * 0.01}
    samplers = [NormalNormal("b", mdl), NormalNormal("beta", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
    M = MCMC(state, samplers, model=mdl, n_burn=n_burn, n_iter=n_iter, n_chains=C, seed=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    M.run_mcmc()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n_burn + n_iter)
    M.engine.check_status()
    out = M.collect()
    print(f"n={n} p={p} C={C}: {1e3 * dt:.2f} ms per sweep = {C / dt:.0f} chain-updates/s; kinds "
          f"{samplers[0].plan(M.state)['kind']}/{samplers[1].plan(M.state)['kind']}; "
          f"all finite: {bool(np.isfinite(out['b']).all() and np.isfinite(out['tau']).all())}; lambda / tau at the end {out['lambda'][:, 0, -1].mean() / out['tau'][:, 0, -1].mean():.3g}; "
          f"band joins retried {M.engine.counter('band_join_retries')}, in one piece {M.engine.counter('band_join_fallbacks')} (groups of 64 chains x draws)")


if __name__ == "__main__":
    main()
