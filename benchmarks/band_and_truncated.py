"""Secondary measurements of the two rows next to the headline path at the
cfg3 size: (a) the conditional draw under a second-order random-walk prior (pentadiagonal precision, band route),
(b) one truncated-Gaussian Gibbs scan under the RW1 prior.  python benchmarks/band_and_truncated.py [--n 10000 --chains 1024]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json, time
import numpy as np
import torch
from scipy import sparse

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10000); ap.add_argument("--chains", type=int, default=1024)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--w", type=int, default=2)
a = ap.parse_args()
from openmcmc_amd.engine import Engine
n, C = a.n, a.chains
eng = Engine(C, seed=2)
rng = np.random.default_rng(0)
t = np.arange(n) * 60.0 / n
y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)

def timed(f, steps):
    f(); eng.check_status(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): f(i + 1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

# (a) band route: RW(w) precision = D'D + ridge with D the w-th difference operator
D = sparse.identity(n, format="csr")
for _ in range(a.w):
    D = D[1:] - D[:-1]
P = (D.T @ D + 1e-3 * sparse.identity(n)).tocsc()
band = np.zeros((a.w + 1, n))
for d in range(a.w + 1):
    band[d, : n - d] = P.diagonal(-d)
terms = [{"band": eng.to_device(band), "scale": eng.full((C,), 100.0)}, {"rhs": eng.to_device(y), "scale": eng.full((C,), 1.0)}]
T = eng.band_terms(terms, n)
x = eng.empty(C, n)
dt = timed(lambda i=0: eng.band_sample_canonical(n, T, x, draw_index=i), a.steps)
print(json.dumps({"workload": f"band draw RW{a.w} n={n} chains={C}", "ms_per_draw": 1e3 * dt, "chain_updates_per_s": C / dt}))

# (b) truncated scan under the RW1 prior
pd = np.full(n, 2.0); pd[0] = pd[-1] = 1.0; pd[0] += 1e-3
tt = [{"diag": eng.to_device(pd), "off": eng.full((n - 1,), -1.0), "scale": eng.full((C,), 100.0)},
      {"rhs": eng.to_device(y), "scale": eng.full((C,), 1.0)}]
TT = eng.tridiag_terms(tt, n)
lower = eng.full((n,), 0.0)
xs = eng.full((C, n), 2.0)
dt = timed(lambda i=0: eng.tridiag_gibbs_truncated(n, TT, xs, lower=lower, draw_index=i), a.steps)
print(json.dumps({"workload": f"truncated Gibbs scan RW1 n={n} chains={C}", "ms_per_scan": 1e3 * dt, "chain_updates_per_s": C / dt}))
