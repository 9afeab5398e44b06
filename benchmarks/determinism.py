#!/usr/bin/env python3
"""Run-to-run determinism of the fused smoother sweep: K sweeps from the same state and seed, R times; every
stored draw must agree bit for bit with the first run (the draws are pure functions of the Philox counter and no
reduction in the kernel depends on arrival order)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--repeats", type=int, default=10)
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--generic", type=int, default=0)
    ap.add_argument("--store", type=int, default=16, help="ring of stored iterations (the last ones are compared)")
    args = ap.parse_args()
    import torch
    from bench import GmrfSweep

    ref = None
    bad = 0
    for r in range(args.repeats):
        sw = GmrfSweep(args.nodes, args.chains, seed=2025, chain_offset=0, device=0, n_store=min(args.steps, args.store))
        sw.eng.set_option("tridiag_generic", args.generic)
        sw.run_fused(args.steps)
        sw.eng.check_status()
        out = (sw.store_lam.cpu().numpy().copy(), sw.store_tau.cpu().numpy().copy(), sw.store_b[-1].cpu().numpy().copy())
        if ref is None:
            ref = out
        else:
            same = all(np.array_equal(a, b) for a, b in zip(out, ref))
            if not same:
                bad += 1
                d = np.argwhere(out[0] != ref[0])
                first = d[0] if len(d) else None
                print(f"run {r}: differs from run 0; first differing (iteration, chain) of lambda: {first}; "
                      f"chains affected at the end: {int((out[0][-1] != ref[0][-1]).sum())}")
        del sw
        torch.cuda.synchronize()
    print(f"generic={args.generic}: {bad} of {args.repeats - 1} repeats differ from the first run")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
