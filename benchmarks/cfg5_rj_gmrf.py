"""Secondary measurement: BASELINE configs[4] / SURVEY.md section 8d cfg5 -- reversible jump over the number of
Gaussian-kernel basis functions next to a GMRF, n = 5000 nodes, n_max = 20, 512 chains per GPU (4096 over 8),
in-kernel random streams.  One sweep = NormalNormal(b), NormalNormal(beta), NormalGamma(lambda), NormalGamma(tau),
RandomWalkLoop(theta) over every knot, ReversibleJump(n_basis), then store + log_post + fitted values.

python benchmarks/cfg5_rj_gmrf.py [--n 5000 --n-max 20 --chains 512 --steps 20 --warmup 3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import argparse, json, time
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=5000); ap.add_argument("--n-max", type=int, default=20)
ap.add_argument("--chains", type=int, default=512); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3); ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()

from rj_problem import build, make_basis_host
from openmcmc_amd import gmrf
from openmcmc_amd.mcmc import MCMC

n, n_max, C = a.n, a.n_max, a.chains
rng = np.random.default_rng(0)
X = np.linspace(-10, 10, n)
theta_true = np.array([[-6.0, -1.0, 4.5]])
beta_true = np.array([[3.0], [-2.0], [4.0]])
b_true = 0.05 * np.cumsum(rng.standard_normal(n)) * np.sqrt(48.0 / n)
y = (make_basis_host(X.reshape(n, 1), theta_true) @ beta_true).ravel() + b_true + 0.1 * rng.standard_normal(n)
P = gmrf.precision_irregular(np.arange(float(n))).tolil()
P[0, 0] += 1e-3
k0 = np.clip(rng.poisson(5, size=C), 1, n_max)                       # n_basis ~ Poisson(5), theta ~ U(-10, 10)
init_theta = [rng.uniform(-10, 10, size=k) for k in k0]
init_beta = [rng.standard_normal(k) for k in k0]
from openmcmc_amd.engine import Engine

eng = Engine(C, seed=a.seed)
mdl, state, samplers = build(y, X, P.tocsc(), n_max, eng, init_theta, init_beta, k0.astype(float))
total = a.warmup + a.steps
M = MCMC(state, samplers, model=mdl, n_burn=a.warmup, n_iter=a.steps, n_chains=C, seed=a.seed, engine=eng)
# time the stored iterations only: run the burn-in, then the rest
M_n_iter = M.n_iter
M.n_iter = 0
t0 = time.perf_counter(); M.run_mcmc(); torch.cuda.synchronize(); t_warm = time.perf_counter() - t0
M.n_burn, M.n_iter = 0, M_n_iter
t0 = time.perf_counter(); M.run_mcmc(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
nb = M.store["n_basis"][:, :, 0]
print(json.dumps({
    "workload": f"cfg5 RJ+GMRF n={n} n_max={n_max} chains={C}", "ms_per_sweep": 1e3 * dt, "chain_updates_per_s": C / dt,
    "warmup_ms_per_sweep": 1e3 * t_warm / max(a.warmup, 1),
    "n_basis_mean": float(nb.mean().item()), "n_basis_max": float(nb.max().item()),
    "accept_theta": samplers[4].accept_rate.acceptance_rate, "accept_n_basis": samplers[5].accept_rate.acceptance_rate,
    "reference_chain_updates_per_s_this_container": 12.3,
}))
