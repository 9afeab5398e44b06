"""BASELINE configs[3] (ManifoldMALA on a d = 500 Gaussian target, 512 chains) through the MCMC object: Model([Normal("x",
mean="mu", precision="Q")]), sampler list [ManifoldMALA("x")], store and log_post per step -- the host layer included
(bench.py --config cfg4 times the library call alone)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA, RandomWalk

    d, C = int(os.environ.get("D", 500)), int(os.environ.get("C", 512))
    n_burn, n_iter = int(os.environ.get("BURN", 200)), int(os.environ.get("ITER", 800))
    rng = np.random.default_rng(0)
    A = rng.standard_normal((d, 2 * d))
    Q = np.linalg.inv(A @ A.T / (2 * d))
    Q = (Q + Q.T) / 2
    for cls in (ManifoldMALA, RandomWalk):
        mdl = Model([Normal("x", mean="mu", precision="Q")])
        state = {"x": np.zeros(d), "mu": np.zeros((d, 1)), "Q": Q}
        smp = cls("x", mdl, step=np.array([[0.5 if cls is ManifoldMALA else 0.05]]))
        M = MCMC(state, [smp], model=mdl, n_burn=n_burn, n_iter=n_iter, n_chains=C, seed=3)
        M.n_burn, M.n_iter = 5, 0
        M.run_mcmc()
        M.n_burn, M.n_iter = n_burn, n_iter
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        M.run_mcmc()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (n_burn + n_iter)
        M.engine.check_status()
        print(f"{cls.__name__} d={d} C={C}: {1e3 * dt:.4f} ms per step = {C / dt:.0f} chain-updates/s through MCMC.run_mcmc; {smp.accept_rate.get_acceptance_rate()}")


if __name__ == "__main__":
    main()
