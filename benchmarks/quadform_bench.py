"""Bandwidth of the standalone quadratic form omc_tridiag_quadform (long-chain route, non-fused paths): one workgroup per
chain streams the chain's row.  2.3-2.5 TB/s on the row with one term, 1.6 with two (n = 10 000-20 000 x 1024 chains; 1.2-1.4
and 0.9-1.0 before the loop lost its null tests and its 64-bit address arithmetic; more threads per chain: +5 %); requesting four strides ahead by hand made it slower: still well below
the fused kernels, open.  python benchmarks/quadform_bench.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from openmcmc_amd.engine import Engine
for n, C in ((20000, 1024), (10000, 1024), (50000, 1024)):
    eng = Engine(C, seed=1)
    pd = np.full(n, 2.0); po = -np.ones(n - 1)
    y = np.random.default_rng(0).standard_normal(n)
    x = eng.to_device(np.random.default_rng(1).standard_normal((C, n)))
    for terms in ([{"diag": eng.to_device(pd), "off": eng.to_device(po)}], [{"diag": eng.to_device(pd), "off": eng.to_device(po)}, {"center": eng.to_device(y)}]):
        q = eng.empty(len(terms), C)
        T = eng.tridiag_terms(terms, n)
        for _ in range(3): eng.tridiag_quadform(n, T, x, q)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): eng.tridiag_quadform(n, T, x, q)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"n={n} C={C} terms={len(terms)}: {1e6*dt:.1f} us  ({C*n*8/dt/1e12:.2f} TB/s on x)  checksum {float(q.sum()):.10e}")
    eng.close()
