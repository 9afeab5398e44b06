#!/usr/bin/env python3
"""Launch-level structure of one omc_gmrf_run launch from the sweep clock: when do the workgroups of a launch start and end,
how long is a chain's first sweep against its later ones, how even are the CUs' finishing times.

    python3 benchmarks/sweep_clock_dump.py [--steps 20] [--chains 1024]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--block-sweeps", type=int, default=0)
    args = ap.parse_args()
    import torch

    import bench

    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    K, C = args.steps, args.chains
    sw = bench.GmrfSweep(10000, C, seed=2025, chain_offset=0, device=0, n_store=K)
    eng = sw.eng
    eng.set_option("run_block_sweeps", args.block_sweeps)
    ring = eng.sweep_clock(64)
    for _ in range(40):
        sw.run_fused(64)
    torch.cuda.synchronize()
    for rep in range(3):
        pos0 = eng.counter("sweep_times_pos")
        sw.run_fused(K)
        torch.cuda.synchronize()
        khz = eng.counter("wall_clock_khz")
        tk = ring[[(pos0 + i) % 64 for i in range(K)]].cpu().numpy().astype(np.float64)
        st, en = tk[:, :, 0] * 1e3 / khz, tk[:, :, 1] * 1e3 / khz  # us
        t0 = st.min()
        st -= t0
        en -= t0
        dur = en - st
        print(f"--- run {rep}: span {en.max():.1f} us; sweep duration by sweep index (median over chains):")
        print("   ", " ".join(f"{np.median(dur[i]):.2f}" for i in range(K)))
        first = st[0]  # start of each chain's first sweep
        order = np.argsort(first)
        rounds = [order[i:i + 256] for i in range(0, C, 256)]
        for r, idx in enumerate(rounds):
            print(f"    round {r}: starts {first[idx].min():8.1f} .. {first[idx].max():8.1f}   ends {en[-1, idx].min():8.1f} .. {en[-1, idx].max():8.1f}"
                  f"   chain time mean {np.mean(en[-1, idx] - first[idx]):8.1f}  sd {np.std(en[-1, idx] - first[idx]):6.1f}")
        tot = en[-1] - st[0]
        print(f"    per-chain total: median {np.median(tot):.1f}  min {tot.min():.1f}  max {tot.max():.1f}; sum of medians x rounds = {np.median(tot) * len(rounds):.1f}")
        # the first 256 to start: a start stagger?
        f0 = np.sort(first)[:256]
        print(f"    first round start stagger: p50 {np.percentile(f0, 50):.2f} p90 {np.percentile(f0, 90):.2f} max {f0.max():.2f} us")
    eng.close()


if __name__ == "__main__":
    main()
