#!/usr/bin/env python3
"""What does the sequential join fallback of the segmented tridiagonal kernel cost?  ("Pivot joins
that Newton does not settle".)  The fallback is forced for EVERY chain (start pivots spoiled by `tridiag_perturb_ppb`,
Newton switched off) on the headline model at several prior/likelihood ratios -- the weaker the ridge, the less
contractive the pivot recurrence and the more passes the fallback needs (at most one per segment) -- and timed per sweep
through omc_gmrf_run with the sweep clock on, so that the worst single sweep of a chain is read off the device.

    python3 benchmarks/join_fallback_cost.py [--chains 1024] [--nodes 10000] [--sweeps 8]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--sweeps", type=int, default=8)
    args = ap.parse_args()
    import torch

    import bench

    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    rows = []
    for tau0, ppb, newton_max, label in [(1.0, 0, 4, "normal path (no fallback)"),
                                         (1.0, 1000, 0, "fallback forced, lambda/tau = 1e2 (the headline's ratio)"),
                                         (1e-4, 1000, 0, "fallback forced, lambda/tau = 1e6"),
                                         (1e-8, 1000, 0, "fallback forced, lambda/tau = 1e10 (next to no ridge)"),
                                         (1e-8, 1000000, 0, "fallback forced, lambda/tau = 1e10, start pivots 1e-3 off")]:
        sw = bench.GmrfSweep(args.nodes, args.chains, seed=3, chain_offset=0, device=0, n_store=args.sweeps)
        eng = sw.eng
        sw.tau.fill_(tau0)
        # the Normal-Gamma blocks would redraw tau from the data; keep the ratio: switch the blocks' redraw off by huge priors
        sw.A_LAM, sw.B_LAM, sw.A_TAU, sw.B_TAU = 1e12, 1e10, 1e12 * tau0, 1e12
        eng.set_option("tridiag_perturb_ppb", ppb)
        eng.set_option("tridiag_newton_max", newton_max)
        ring = eng.sweep_clock(64)
        sw.run_fused(args.sweeps)
        torch.cuda.synchronize()
        fb0 = eng.counter("tridiag_join_fallbacks")
        pos0 = eng.counter("sweep_times_pos")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sw.run_fused(args.sweeps)
        e1.record()
        torch.cuda.synchronize()
        try:
            eng.check_status()
            status = "ok"
        except Exception as exc:  # noqa: BLE001
            status = repr(exc)
        khz = eng.counter("wall_clock_khz")
        tk = ring[[(pos0 + i) % 64 for i in range(args.sweeps)]].cpu().numpy()
        dur = (tk[:, :, 1] - tk[:, :, 0]) / khz * 1e3
        rows.append({"case": label, "status": status, "ms_per_sweep_all_chains": e0.elapsed_time(e1) / args.sweeps,
                     "fallbacks_per_sweep": (eng.counter("tridiag_join_fallbacks") - fb0) / args.sweeps,
                     "one_chain_sweep_us": {"median": float(np.median(dur)), "max": float(dur.max())},
                     "tau_after": float(sw.tau.mean().item())})
        print(json.dumps(rows[-1]), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
