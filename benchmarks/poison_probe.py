"""Diagnostic: does what an earlier, unrelated workload leaves behind (freed device memory, LDS contents) change the
hierarchical smoother's run?  python benchmarks/poison_probe.py [band0] [band1]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch

import test_band_seg_gpu as B
import test_hier_gpu as T

if "nanempty" in sys.argv:  # every uninitialised float64 allocation comes back as NaN: a read before the write shows
    _empty, _empty_like = torch.empty, torch.empty_like

    def empty(*a, **k):
        t = _empty(*a, **k)
        if t.is_floating_point():
            t.fill_(float("nan"))
        return t

    def empty_like(*a, **k):
        t = _empty_like(*a, **k)
        if t.is_floating_point():
            t.fill_(float("nan"))
        return t

    torch.empty, torch.empty_like = empty, empty_like
if "bandtests" in sys.argv:
    import pytest
    pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_band_seg_gpu.py"),
                 os.path.join(root, "tests", "test_band_gpu.py"), os.path.join(root, "tests", "test_hier_gpu.py"), "-k", "not at_size"])
for a in sys.argv[1:]:
    if a.startswith("band") and a[4:].isdigit():
        out = B.draw(10000, 1024, 2, 100.0, 1.0, int(a[4:]), inject=False)
        print(a, "done, fallbacks", out[3], flush=True)
rng = np.random.default_rng(1)
G = T._synthetic(10000, rng, n_burn=20, n_iter=30)
M, _ = T.build(G, "s_", 1024, seed=3)
eng = M.engine
for a in sys.argv[1:]:
    if a.startswith("opt:"):
        name, val = a[4:].split("=")
        eng.set_option(name, int(val))
        print("option", name, val)
for rep in range(int(os.environ.get("RUNS", "4"))):
    try:
        M.run_mcmc()
        print("run", rep, "status ok", flush=True)
    except Exception as e:
        print("run", rep, "FAILED", type(e).__name__, e, flush=True)
        for key in ("lambda", "tau"):
            if key in M.state:
                v = M.state[key]
                v = v.data if hasattr(v, "data") and not isinstance(v, torch.Tensor) else v
                if isinstance(v, torch.Tensor):
                    w = v.reshape(-1).cpu().numpy()
                    print("   state", key, "chain 773:", w[773] if w.size > 773 else None, "non-finite:", np.flatnonzero(~np.isfinite(w))[:8],
                          "non-positive:", np.flatnonzero(w <= 0)[:8])
        break
for name in ("tridiag_join_fallbacks", "run_handoff_timeouts", "band_join_fallbacks", "band_join_retries"):
    try:
        print(name, eng.counter(name))
    except Exception as e:
        print(name, "?", e)
for k, v in M.store.items():
    if isinstance(v, torch.Tensor):
        fin = torch.isfinite(v.reshape(v.shape[0], -1)).all(dim=1).cpu().numpy()
        bad = np.flatnonzero(~fin)
        print(k, tuple(v.shape), "all finite" if bad.size == 0 else f"first non-finite iteration {bad[0]} ({bad.size} of {fin.size})")
        if bad.size:
            w = v[int(bad[0])]
            chains = np.flatnonzero(~torch.isfinite(w.reshape(w.shape[0], -1)).all(dim=1).cpu().numpy())
            print("   chains", chains[:10], "of", chains.size)
