"""BASELINE configs[1] (linear regression p = 1000, n = 10 000, 256 chains) through the MCMC object itself -- Model, samplers
[NormalNormal(beta), NormalGamma(tau), NormalGamma(lambda)], response store, log_post -- rather than through direct engine
calls (bench.py --config cfg2): what a user of the reference's API gets per sweep, host layer included."""
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

    n, p, C = int(os.environ.get("N", 10000)), int(os.environ.get("P", 1000)), int(os.environ.get("C", 256))
    n_burn, n_iter = int(os.environ.get("BURN", 50)), int(os.environ.get("ITER", 150))
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, p))
    beta = rng.standard_normal(p)
    y = X @ beta + 0.1 * rng.standard_normal(n)
    mdl = Model([Normal("y", mean=LinearCombination(form={"beta": "X"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
                 Normal("beta", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
                 Gamma("tau", shape="a_tau", rate="b_tau"), Gamma("lambda", shape="a_lambda", rate="b_lambda")], response={"y": "mean"})
    state = {"y": y, "X": X, "beta": np.zeros(p), "P_tau": sparse.identity(n, format="csc"), "tau": 1.0,
             "P_lambda": sparse.identity(p, format="csc"), "mu": np.zeros(p), "lambda": 0.01, "a_tau": 1e-3, "b_tau": 1e-3,
             "a_lambda": 1e-3, "b_lambda": 1e-3}
    samplers = [NormalNormal("beta", mdl), NormalGamma("tau", mdl), NormalGamma("lambda", mdl)]
    M = MCMC(state, samplers, model=mdl, n_burn=n_burn, n_iter=n_iter, n_chains=C, seed=3)
    M.n_burn, M.n_iter = 3, 0  # set-up outside the timing: Gram matrix, eigendecomposition, device copies
    M.run_mcmc()
    M.n_burn, M.n_iter = n_burn, n_iter
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    M.run_mcmc()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n_burn + n_iter)
    M.engine.check_status()
    mean_beta, _ = M.summary("beta")
    err = float(np.abs(np.asarray(mean_beta.cpu() if hasattr(mean_beta, "cpu") else mean_beta).reshape(-1) - beta).max())
    print(f"n={n} p={p} C={C}: {1e3 * dt:.3f} ms per sweep = {C / dt:.0f} chain-updates/s through MCMC.run_mcmc; posterior mean error of beta {err:.2e}")


if __name__ == "__main__":
    main()
