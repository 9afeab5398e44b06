"""One truncated-Gaussian Gibbs scan under the RW1 prior (for profiling): python benchmarks/trunc_profile.py [--n 10000 --chains 1024]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json, time
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10000); ap.add_argument("--chains", type=int, default=1024)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
from openmcmc_amd.engine import Engine
n, C = a.n, a.chains
eng = Engine(C, seed=2)
rng = np.random.default_rng(0)
t = np.arange(n) * 60.0 / n
y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)
pd = np.full(n, 2.0); pd[0] = pd[-1] = 1.0; pd[0] += 1e-3
tt = [{"diag": eng.to_device(pd), "off": eng.full((n - 1,), -1.0), "scale": eng.full((C,), 100.0)},
      {"rhs": eng.to_device(y), "scale": eng.full((C,), 1.0)}]
TT = eng.tridiag_terms(tt, n)
lower = eng.full((n,), 0.0)
xs = eng.full((C, n), 2.0)
for i in range(2):
    eng.tridiag_gibbs_truncated(n, TT, xs, lower=lower, draw_index=i)
eng.check_status(); torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(a.steps):
    eng.tridiag_gibbs_truncated(n, TT, xs, lower=lower, draw_index=2 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"workload": f"truncated Gibbs scan RW1 n={n} chains={C}", "ms_per_scan": 1e3 * dt, "chain_updates_per_s": C / dt,
                  "us_per_site": 1e6 * dt / n}))
