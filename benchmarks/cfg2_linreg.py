"""Secondary measurement: BASELINE configs[1] (Bayesian linear regression p=1000, n=10000, 256 chains)
through the dense path.  python benchmarks/cfg2_linreg.py [--p 1000 --n 10000 --chains 256 --steps 5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json, time
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--p", type=int, default=1000); ap.add_argument("--n", type=int, default=10000)
ap.add_argument("--chains", type=int, default=256); ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
torch.cuda.set_stream(torch.cuda.Stream())
from openmcmc_amd.engine import Engine
rng = np.random.default_rng(0)
n, p, C = a.n, a.p, a.chains
X = rng.standard_normal((n, p)); beta = rng.standard_normal(p); y = X @ beta + 0.1 * rng.standard_normal(n)
eng = Engine(C, seed=1)
dX, dy = eng.to_device(X), eng.to_device(y)
torch.cuda.synchronize(); t0 = time.perf_counter()
Gram, Xty = eng.gram(dX), eng.design_rhs(dX, dy)
torch.cuda.synchronize(); t_gram = time.perf_counter() - t0
lam, tau = eng.full((C,), 0.01), eng.full((C,), 1.0)
terms = eng.dense_terms([{"mat": None, "scale": lam}, {"mat": Gram, "rhs": Xty, "scale": tau}], p)
ident = eng.tridiag_terms([{}], p)
b, fitted = eng.empty(C, p), eng.empty(C, n)
q_tau, q_lam, lp, zero = eng.empty(C), eng.empty(1, C), eng.empty(C), eng.zeros(1)
def sweep(it):
    eng.dense_sample_canonical(p, terms, b, draw_index=3 * it)
    eng.design_predict(dX, b, fitted); eng.weighted_resid_sq(dy, fitted, q_tau)
    eng.normal_gamma_update(1e-3, 1e-3, n, q_tau, tau, draw_index=3 * it + 1)
    eng.tridiag_quadform(p, ident, b, q_lam)
    eng.normal_gamma_update(1e-3, 1e-3, p, q_lam[0], lam, draw_index=3 * it + 2)
    eng.scaled_gauss_logpdf(n, tau, zero, q_tau, lp); eng.scaled_gauss_logpdf(p, lam, zero, q_lam[0], lp, accumulate=True)
    eng.gamma_logpdf(tau, 1e-3, 1e-3, lp, accumulate=True); eng.gamma_logpdf(lam, 1e-3, 1e-3, lp, accumulate=True)
sweep(0); sweep(1); eng.check_status()
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(a.steps): sweep(2 + it)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
flop = C * (p**3 / 3 + 2 * n * p + 4 * p * p)
print(json.dumps({"workload": f"linreg p={p} n={n} chains={C}", "ms_per_sweep": 1e3 * dt, "chain_updates_per_s": C / dt,
                  "gram_ms_one_off": 1e3 * t_gram, "gram_tflops": 2 * p * p * n / t_gram / 1e12,
                  "alg_flop_per_sweep": flop, "achieved_tflops": flop / dt / 1e12,
                  "tau_mean": tau.mean().item(), "beta_err": float(np.abs(b.mean(0).cpu().numpy() - beta).max())}))
