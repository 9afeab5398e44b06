"""RW(w) band draws only (for rocprofv3): python benchmarks/band_profile.py [--n 10000 --chains 1024 --w 2 --steps 20]
--lattice K: a K x K first-order lattice GMRF instead (4-neighbour Laplacian + ridge, row-major order: n = K^2, bandwidth K --
SURVEY 8f rank 1's "2-D lattice GMRF with bandwidth sqrt(n)", gmrf.py:489-520 on a sparse precision of that shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json, time
import numpy as np
import torch
from scipy import sparse

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10000); ap.add_argument("--chains", type=int, default=1024)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--w", type=int, default=2)
ap.add_argument("--overlap", type=int, default=0); ap.add_argument("--algo", type=int, default=0)
ap.add_argument("--segments", type=int, default=0)
ap.add_argument("--lattice", type=int, default=0)
ap.add_argument("--rows", type=int, default=0, help="--lattice K --rows R: an R x K lattice (n = R K, bandwidth K)")
a = ap.parse_args()
if a.lattice:
    a.rows = a.rows or a.lattice
    a.n, a.w = a.rows * a.lattice, a.lattice
from openmcmc_amd.engine import Engine
n, C = a.n, a.chains
eng = Engine(C, seed=2)
eng.set_option("band_algo", a.algo)
if os.environ.get("OMC_BLOCKED_THREADS", "0") not in ("", "0"):
    eng.set_option("band_blocked_threads", int(os.environ["OMC_BLOCKED_THREADS"]))
if a.segments:
    eng.set_option("band_seg_count", a.segments)
if a.overlap:
    eng.set_option("band_seg_overlap", a.overlap)
rng = np.random.default_rng(0)
t = np.arange(n) * 60.0 / n
y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)
if a.lattice:
    K = a.lattice
    D1 = sparse.identity(K, format="csr")
    D1 = D1[1:] - D1[:-1]
    L1 = (D1.T @ D1)
    DR = sparse.identity(a.rows, format="csr")
    DR = DR[1:] - DR[:-1]
    LR = (DR.T @ DR)
    P = (sparse.kron(sparse.identity(a.rows), L1) + sparse.kron(LR, sparse.identity(K)) + 1e-3 * sparse.identity(n)).tocsc()
else:
    D = sparse.identity(n, format="csr")
    for _ in range(a.w):
        D = D[1:] - D[:-1]
    P = (D.T @ D + 1e-3 * sparse.identity(n)).tocsc()
band = np.zeros((a.w + 1, n))
for d in range(a.w + 1):
    band[d, : n - d] = P.diagonal(-d)
terms = [{"band": eng.to_device(band), "scale": eng.full((C,), 100.0)}, {"rhs": eng.to_device(y), "scale": eng.full((C,), 1.0)}]
T = eng.band_terms(terms, n)
x = eng.empty(C, n)
for i in range(3):
    eng.band_sample_canonical(n, T, x, draw_index=i)
eng.check_status(); torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(a.steps):
    eng.band_sample_canonical(n, T, x, draw_index=3 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
flop = C * (n * a.w * a.w + 6.0 * n * a.w)  # band Cholesky n w^2 + three band solves 2 n w each
if os.environ.get("OMC_WIDE_STAMPS"):
    st = torch.zeros(4096, dtype=torch.int64, device="cuda")
    eng.set_option("stamps_ptr", st.data_ptr())
    eng.band_sample_canonical(n, T, x, draw_index=99)
    torch.cuda.synchronize()
    v = st.cpu().numpy()[:8].astype(float)
    print("own work per wave in the look-ahead phase (cycles):", st.cpu().numpy()[8:16].tolist(), file=sys.stderr)
    print("own work per wave in the backward pass (cycles):", st.cpu().numpy()[16:24].tolist(), file=sys.stderr)
    names = ["prefetch", "factor (first block; late ones of narrow bands)", "first tile column + rhs", "factor ahead | store + tiles", "S4 refill", "u+z", "back: first far sums", "back: a block per barrier"]
    print("stamps (cycles, chain 0):", {k: int(a) for k, a in zip(names, v)}, "total", int(v.sum()), file=sys.stderr)
    eng.set_option("stamps_ptr", 0)
print(json.dumps({"workload": (f"band draw {a.rows} x {a.lattice} lattice (w = {a.w})" if a.lattice else f"band draw RW{a.w}") + f" n={n} chains={C}",
                  "tflops_on_n_w2": flop / dt / 1e12, "factor_workspace_GB": C * n * (a.w + 1) * 8 / 1e9, "ms_per_draw": 1e3 * dt, "chain_updates_per_s": C / dt,
                  "join_fallbacks": eng.counter("band_join_fallbacks"), "overlap": a.overlap or 192, "segments": a.segments or "auto"}))
