"""A regression mean with two sampled coefficient blocks at the cfg2 size: y ~ N(X beta + Z gamma, (tau I)^-1), n = 10 000,
beta (p = 1000) under lambda I, gamma (q = 200) under a dense prior precision, 256 chains, through MCMC.run_mcmc with the
reference's sampler list [NormalNormal(beta), NormalNormal(gamma), NormalGamma(tau), NormalGamma(lambda)].  Each conditional
sees the other block as a per-chain offset (sampler/sampler.py:185-192): beta on the spectral route with a per-chain right-hand
side, gamma on the factorisation route.  Prints the time per sweep and how far the posterior means are from the truth."""
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

    n, p, q, C = int(os.environ.get("N", 10000)), int(os.environ.get("P", 1000)), int(os.environ.get("Q", 200)), int(os.environ.get("C", 256))
    n_burn, n_iter = int(os.environ.get("BURN", 20)), int(os.environ.get("ITER", 40))
    rng = np.random.default_rng(0)
    X, Z = rng.standard_normal((n, p)), rng.standard_normal((n, q))
    beta, gamma = rng.standard_normal(p), rng.standard_normal(q)
    y = X @ beta + Z @ gamma + 0.1 * rng.standard_normal(n)
    A = rng.standard_normal((q, q)) * 0.1
    mdl = Model([
        Normal("y", mean=LinearCombination(form={"beta": "X", "gamma": "Z"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
        Normal("beta", mean="mu_b", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
        Normal("gamma", mean="mu_g", precision="P_g"),
        Gamma("tau", shape="a_tau", rate="b_tau"), Gamma("lambda", shape="a_lambda", rate="b_lambda")])
    state = {"y": y, "X": X, "Z": Z, "beta": np.zeros(p), "gamma": np.zeros(q), "mu_b": np.zeros(p), "mu_g": np.zeros(q),
             "P_tau": sparse.identity(n, format="csc"), "tau": 1.0, "P_lambda": sparse.identity(p, format="csc"), "lambda": 0.1,
             "P_g": A @ A.T + np.eye(q), "a_tau": 1e-2, "b_tau": 1e-2, "a_lambda": 1e-2, "b_lambda": 1e-2}
    samplers = [NormalNormal("beta", mdl), NormalNormal("gamma", mdl), NormalGamma("tau", mdl), NormalGamma("lambda", mdl)]
    M = MCMC(state, samplers, model=mdl, n_burn=n_burn, n_iter=n_iter, n_chains=C, seed=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    M.run_mcmc()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n_burn + n_iter)
    M.engine.check_status()
    out = M.collect()
    eb = np.abs(out["beta"].mean(axis=(0, 2)) - beta).max()
    eg = np.abs(out["gamma"].mean(axis=(0, 2)) - gamma).max()
    print(f"n={n} p={p} q={q} C={C}: {1e3 * dt:.2f} ms per sweep = {C / dt:.0f} chain-updates/s; kinds "
          f"{samplers[0].plan(M.state)['kind']}/{samplers[1].plan(M.state)['kind']}; posterior mean error beta {eb:.2e}, gamma {eg:.2e}; "
          f"tau mean {out['tau'].mean():.1f} (truth 100)")


if __name__ == "__main__":
    main()
