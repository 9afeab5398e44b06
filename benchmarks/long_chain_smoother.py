"""The headline model on LONG chains through the MCMC object: n beyond the 16 384 nodes one workgroup takes.  NormalNormal
then plans the band route (w = 1: the segmented lane-per-chain kernels), the Normal-Gamma updates and log_post run as their
own launches."""
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

    C = int(os.environ.get("C", 1024))
    for n in [int(v) for v in os.environ.get("NS", "20000,50000").split(",")]:
        rng = np.random.default_rng(2)
        t = np.arange(n) * 60.0 / 10000
        y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + 0.3 * rng.standard_normal(n)
        D = sparse.diags([-np.ones(n - 1), np.ones(n - 1)], offsets=[0, 1], shape=(n - 1, n))
        P = (D.T @ D + 1e-3 * sparse.identity(n)).tocsc()
        mdl = Model([Normal("y", mean=LinearCombination(form={"b": "A"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
                     Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
                     Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
        state = {"y": y, "b": y.copy(), "mu": np.zeros(n), "lambda": 50.0, "P_lambda": P, "a_lam": 10.0, "b_lam": 1.0, "tau": 1.0,
                 "P_tau": sparse.identity(n, format="csc"), "a_tau": 1.0, "b_tau": 1.0, "A": sparse.identity(n, format="csc")}
        samplers = [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
        M = MCMC(state, samplers, model=mdl, n_burn=5, n_iter=10, n_chains=C, seed=3)
        M.run_mcmc()   # (untimed: the model's one-off set-up -- plans, the log determinant of the prior precision, the store -- and warm-up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        M.run_mcmc()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 15
        M.engine.check_status()
        lam = M.store["lambda"].mean().item()
        print(f"n={n} C={C}: {1e3 * dt:.2f} ms per sweep = {C / dt:.0f} chain-updates/s; plan {samplers[0].plan(M.state)['kind']}; lambda mean {lam:.1f}; "
              f"band joins retried {M.engine.counter('band_join_retries')}, in one piece {M.engine.counter('band_join_fallbacks')}")
        del M


if __name__ == "__main__":
    main()
