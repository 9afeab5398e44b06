"""Warm timing of G = X' diag(w) X at BASELINE configs[1] (n = 10 000, p = 1 000): own fp64 MFMA kernel vs the rocBLAS
route.  python3 benchmarks/gram_bench.py [--n 10000 --p 1000 --reps 20]"""
import os, sys, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openmcmc_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10000); ap.add_argument("--p", type=int, default=1000); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
torch.cuda.set_stream(torch.cuda.Stream())
eng = Engine(1)
rng = np.random.default_rng(0)
X, w = eng.to_device(rng.standard_normal((a.n, a.p))), eng.to_device(0.5 + rng.random(a.n))
out = {}
for name, flag in (("mfma", 0), ("rocblas", 1)):
    eng.set_option("gram_use_rocblas", flag)
    for _ in range(3):
        eng.gram(X, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        eng.gram(X, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    out[name] = {"ms": ms, "tflops_on_2p2n": 2.0 * a.p * a.p * a.n / (ms * 1e-3) / 1e12}
print(json.dumps({"workload": f"gram n={a.n} p={a.p} weighted", **out}))
