"""Bandwidth of the standalone quadratic form omc_tridiag_quadform (long-chain route, non-fused paths): one workgroup per
chain streams the chain's row, once for all terms.  n = 20 000 x 1024 chains: tridiagonal term 61 us (2.7 TB/s on the row;
119 us at the start of round 3), identity term around a shared centre 37 us (4.4 TB/s), both 82 us (166), with a per-chain
centre 130 us (two rows).  python benchmarks/quadform_bench.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from openmcmc_amd.engine import Engine
for n, C in ((20000, 1024), (10000, 1024), (50000, 1024)):
    eng = Engine(C, seed=1)
    pd = np.full(n, 2.0); po = -np.ones(n - 1)
    y = np.random.default_rng(0).standard_normal(n)
    x = eng.to_device(np.random.default_rng(1).standard_normal((C, n)))
    tri = {"diag": eng.to_device(pd), "off": eng.to_device(po)}
    for terms in ([tri], [{"center": eng.to_device(y)}], [tri, {"center": eng.to_device(y)}], [dict(tri, center_chain=x * 0.5), {"center": eng.to_device(y)}]):
        q = eng.empty(len(terms), C)
        T = eng.tridiag_terms(terms, n)
        for _ in range(3): eng.tridiag_quadform(n, T, x, q)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): eng.tridiag_quadform(n, T, x, q)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        shape = "+".join(("tri" if "off" in t else "ident") + ("(c)" if "center" in t else "") + ("(cc)" if "center_chain" in t else "") for t in terms)
        print(f"n={n} C={C} {shape}: {1e6*dt:.1f} us  ({C*n*8/dt/1e12:.2f} TB/s on x)  checksum {float(q.sum()):.10e}")
    eng.close()
