"""Repeatability stress of the blocked band kernel at full size: every shape drawn several times with the same injected z, all
results compared bit for bit, the first also against the column-at-a-time kernel (a few chains).  A hand-over between waves that
is not fenced shows here as a differing repeat (it did once: see tests/test_full_size_gpu.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openmcmc_amd.engine import Engine

shapes = [(100, 100, 1024), (312, 32, 2048), (156, 64, 1024), (1250, 8, 3000), (666, 15, 2100), (83, 120, 300), (200, 50, 1500), (2500, 4, 1300),
          (384, 26, 700)]
reps = int(os.environ.get("REPS", 6))
rng = np.random.default_rng(1)
for R, K, C in shapes:
    n, w = R * K, K
    band = np.zeros((w + 1, n))
    band[0] = 4.05
    band[1, : n - 1] = -1.0
    band[1, K - 1 :: K] = 0.0           # no link across a lattice row's end
    band[w, : n - w] = -1.0
    eng = Engine(C, seed=3)
    terms = [{"band": eng.to_device(band), "scale": eng.to_device(1.0 + rng.random(C))},
             {"rhs": eng.to_device(rng.standard_normal(n)), "scale": eng.to_device(0.5 + rng.random(C))}]
    z = eng.to_device(rng.standard_normal((C, n)))
    first = None
    bad = 0
    t0 = time.perf_counter()
    for r in range(reps):
        x, mu, ld = eng.empty(C, n), eng.empty(C, n), eng.empty(C)
        eng.band_sample_canonical(n, terms, x, z=z, mean_out=mu, logdet_out=ld)
        eng.check_status()
        cur = (x.cpu().numpy(), mu.cpu().numpy(), ld.cpu().numpy())
        if first is None:
            first = cur
        else:
            bad += sum(0 if np.array_equal(a, b) else 1 for a, b in zip(first, cur))
    dt = time.perf_counter() - t0
    # against the column-at-a-time kernel on the first chains
    Cs = min(C, 8)
    eng2 = Engine(Cs, seed=3)
    eng2.set_option("band_algo", 2)
    t2 = [{"band": terms[0]["band"], "scale": terms[0]["scale"][:Cs].contiguous()}, {"rhs": terms[1]["rhs"], "scale": terms[1]["scale"][:Cs].contiguous()}]
    xo, lo = eng2.empty(Cs, n), eng2.empty(Cs)
    eng2.band_sample_canonical(n, t2, xo, z=z[:Cs].contiguous(), logdet_out=lo)
    eng2.check_status()
    err = np.max(np.abs(first[0][:Cs] - xo.cpu().numpy())) / np.max(np.abs(xo.cpu().numpy()))
    print(f"w {w:4d} n {n} chains {C:5d}: {reps} draws, differing repeats {bad}, max rel diff to the column-at-a-time kernel {err:.1e}, {dt:.2f} s", flush=True)
    eng.close(); eng2.close()
