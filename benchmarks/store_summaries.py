#!/usr/bin/env python3
"""SURVEY 8f rank 3 at the headline size: on-device summaries of store["b"] (n_iter x 1024 chains x 10 000 nodes) and the ring store.

    python3 benchmarks/store_summaries.py [--iters 128] [--chains 1024] [--nodes 10000]

Prints one JSON line: time and HBM rate of the moments and of the quantile passes (three levels: 2.5 %, 50 %, 97.5 %), pooled and
per chain, against the bytes they must read (moments: the store once; quantiles: eight digit passes over the store), and the sweep
rate of a run whose store is a ring drained to pinned host memory while it samples.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=128)
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--ring", type=int, default=16)
    args = ap.parse_args()
    import torch

    from bench import GmrfSweep

    n, C, K = args.nodes, args.chains, args.iters
    sw = GmrfSweep(n, C, seed=7, chain_offset=0, device=0, n_store=K)
    eng = sw.eng
    sw.run_fused(K + 8)  # fill the store (the first 8 sweeps are overwritten: burn-in)
    torch.cuda.synchronize()
    store = sw.store_b
    nbytes = store.numel() * 8
    out = {"store": f"{K} iterations x {C} chains x {n} nodes", "store_GB": nbytes / 1e9}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, r

    for pooled in (True, False):
        tag = "pooled" if pooled else "per_chain"
        ms, _ = timed(lambda: eng.store_moments(store, pooled=pooled))
        out[f"moments_{tag}"] = {"ms": ms, "GBps_on_one_read": nbytes / ms / 1e6}
        ms, q = timed(lambda: eng.store_quantiles(store, [0.025, 0.5, 0.975], pooled=pooled), reps=1)
        out[f"quantiles_{tag}"] = {"ms": ms, "GBps_on_eight_reads": 8 * nbytes / ms / 1e6, "levels": 3}
        if pooled:  # spot check against numpy on a few elements
            idx = [0, n // 2, n - 1]
            host = store[:, :, idx].cpu().numpy().reshape(-1, len(idx))
            out["check_vs_numpy"] = bool(np.array_equal(q[:, idx].cpu().numpy(), np.quantile(host, [0.025, 0.5, 0.975], axis=0)))
    eng.check_status()
    del store, sw
    torch.cuda.empty_cache()

    # a run through MCMC with a ring store: sampling while the second stream drains to pinned host memory
    from scipy import sparse

    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal
    from bench import gmrf_problem

    y, d, off = gmrf_problem(n)
    P = sparse.diags((off, d, off), offsets=[-1, 0, 1], format="csc")
    mdl = Model([Normal("y", mean=LinearCombination(form={"b": "A"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
                 Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
                 Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
    state = {"y": y, "b": y, "mu": np.zeros(n), "lambda": 100, "P_lambda": P, "a_lam": 10, "b_lam": 1, "tau": 1,
             "P_tau": sparse.identity(n, format="csc"), "a_tau": 1, "b_tau": 1, "A": sparse.identity(n, format="csc")}
    n_it = 4 * args.ring
    for ring in (0, args.ring):
        smp = [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
        M = MCMC(state, smp, model=mdl, n_burn=8, n_iter=n_it, n_chains=C, seed=3, store_ring=ring)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        M.run_mcmc()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["run_ring_%d" % ring] = {"n_iter": n_it, "ms_per_sweep": 1e3 * dt / (n_it + 8),
                                     "device_slabs": int(M.store["b"].shape[0]),
                                     "host_GBps": (n_it * C * n * 8 / dt / 1e9) if ring else None}
        M.engine.close()
        del M
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
