"""Is the hierarchical smoother at size reproducible run to run?  Two MCMC objects, same seed, same process: the stores
must be bit-equal; if not, which parameter differs first, at which iteration, in which chains.
python benchmarks/determinism_hier.py [opt:name=value ...]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch

import test_hier_gpu as T

n = int(os.environ.get("N", "10000"))
C = int(os.environ.get("CHAINS", "1024"))
rng = np.random.default_rng(1)
G = T._synthetic(n, rng, n_burn=int(os.environ.get("BURN", "20")), n_iter=int(os.environ.get("ITER", "30")))
runs = []
for r in range(int(os.environ.get("RUNS", "2"))):
    M, _ = T.build(G, "s_", C, seed=3)
    for a in sys.argv[1:]:
        if a.startswith("opt:"):
            name, val = a[4:].split("=")
            M.engine.set_option(name, int(val))
    try:
        M.run_mcmc()
        st = "ok"
    except Exception as e:
        st = f"FAILED {e}"
    print("run", r, st, "join fallbacks", M.engine.counter("tridiag_join_fallbacks"), flush=True)
    runs.append({k: v.clone() for k, v in M.store.items() if isinstance(v, torch.Tensor)})
    lam = runs[-1]["lambda"].reshape(runs[-1]["lambda"].shape[0], -1)
    tau = runs[-1]["tau"].reshape(lam.shape[0], -1)
    print("   lambda range", float(lam.min()), float(lam.max()), " tau range", float(tau.min()), float(tau.max()))
    M.engine.close()
    del M
a = runs[0]
for r, b in enumerate(runs[1:], 1):
    for k in a:
        x, y = a[k].reshape(a[k].shape[0], C, -1), b[k].reshape(b[k].shape[0], C, -1)
        same = (x == y) | (torch.isnan(x) & torch.isnan(y))
        if bool(same.all()):
            print(f"run {r} vs 0: {k} bit-equal")
            continue
        it_bad = np.flatnonzero(~same.all(dim=2).all(dim=1).cpu().numpy())
        ch_bad = np.flatnonzero(~same[int(it_bad[0])].all(dim=1).cpu().numpy())
        d = (x[int(it_bad[0])] - y[int(it_bad[0])]).abs().max().item()
        print(f"run {r} vs 0: {k} DIFFERS first at stored iteration {it_bad[0]} ({it_bad.size} iterations), chains {ch_bad[:12]} ({ch_bad.size}), max abs diff there {d:.3e}")
