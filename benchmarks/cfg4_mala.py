"""Secondary measurement: BASELINE configs[3] (ManifoldMALA on a 500-dim correlated Gaussian target),
512 chains per GPU (4096 over 8).  python benchmarks/cfg4_mala.py [--d 500 --chains 512 --steps 200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json, time
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=500); ap.add_argument("--chains", type=int, default=512)
ap.add_argument("--steps", type=int, default=200); ap.add_argument("--kind", default="mala")
ap.add_argument("--ksplit", type=int, default=0, help="omc_dgemm_small: wave groups cutting the contraction (0 = library default)")
ap.add_argument("--rocblas", action="store_true", help="products through rocBLAS instead of the own GEMM")
a = ap.parse_args()
torch.cuda.set_stream(torch.cuda.Stream())
from openmcmc_amd.engine import Engine
d, C = a.d, a.chains
rng = np.random.default_rng(0)
A = rng.standard_normal((d, 2 * d)); Sig = A @ A.T / (2 * d); Qh = np.linalg.inv(Sig); Qh = (Qh + Qh.T) / 2
eng = Engine(C, seed=3)
if a.ksplit: eng.set_option("mh_gemm_ksplit", a.ksplit)
if a.rocblas: eng.set_option("mh_use_rocblas", 1)
Q = eng.to_device(Qh)
step = 0.5 if a.kind == "mala" else 0.05
L, sl = eng.dense_cholesky(Q, 1.0 / step**2 if a.kind == "mala" else 1.0)
x = eng.to_device(np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T)
acc = torch.zeros(C, dtype=torch.int64, device="cuda"); prop = torch.zeros(C, dtype=torch.int64, device="cuda")
def one(it):
    if a.kind == "mala": eng.mala_step(Q, None, L, sl, step, x, draw_index=it, accept_count=acc, proposal_count=prop)
    else: eng.rw_step(None, L, sl, step, x, draw_index=it, accept_count=acc, proposal_count=prop)
for it in range(10): one(it)
eng.check_status(); torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(a.steps): one(10 + it)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
flop = 14 * d * d * C
print(json.dumps({"workload": f"{a.kind} d={d} chains={C}", "ms_per_step": 1e3 * dt, "chain_updates_per_s": C / dt,
                  "alg_flop_per_step": flop, "achieved_tflops": flop / dt / 1e12,
                  "acceptance": acc.sum().item() / prop.sum().item()}))
