"""The per-chain factorisation route of the dense conditional alone (omc_dense_sample_canonical at the cfg2 size: p = 1000,
256 chains), for a kernel trace:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dense -- python3 benchmarks/dense_factor_profile.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from openmcmc_amd.engine import Engine

    p, C, reps = int(os.environ.get("P", 1000)), int(os.environ.get("C", 256)), int(os.environ.get("REPS", 10))
    rng = np.random.default_rng(0)
    A = rng.standard_normal((2 * p, p))
    G = A.T @ A / p
    eng = Engine(C, seed=1)
    Gd, rhs = eng.to_device(G), eng.to_device(rng.standard_normal(p))
    lam, tau = eng.full((C,), 0.01) * (1 + eng.to_device(rng.random(C))), eng.full((C,), 1.0) * (1 + eng.to_device(rng.random(C)))
    terms = eng.dense_terms([{"mat": None, "scale": lam}, {"mat": Gd, "rhs": rhs, "scale": tau}], p)
    x = eng.empty(C, p)
    ref = None
    for overlap in ([int(os.environ["OVERLAP"])] if "OVERLAP" in os.environ else [0, 1]):
        eng.set_option("dense_overlap", overlap)  # 1: the two halves of the chains on two streams (panels under GEMMs)
        for _ in range(2):
            eng.dense_sample_canonical(p, terms, x, draw_index=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            eng.dense_sample_canonical(p, terms, x, draw_index=i + 1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        eng.check_status()
        same = "" if ref is None else f", draws identical to the one-batch run: {bool(torch.equal(ref, x))}"
        ref = x.clone()
        print(f"p={p} C={C} overlap={overlap}: {1e3 * dt:.3f} ms per draw of all chains, {C * p**3 / 3 / dt / 1e12:.1f} TFLOP/s on p^3/3{same}")


if __name__ == "__main__":
    main()
