import sys, numpy as np
sys.path.insert(0, '.')
import torch
from openmcmc_amd.engine import Engine
for d in (500, 1000, 1024, 1026, 1100):
    C = 5
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, 2 * d)); Qh = np.linalg.inv(A @ A.T / (2 * d)); Qh = (Qh + Qh.T) / 2
    mu = rng.standard_normal(d)
    step = 0.3
    eng = Engine(C, seed=1)
    L, sl = eng.dense_cholesky(eng.to_device(Qh), 1.0 / step**2)
    Lh = L.cpu().numpy().T  # column-major on device -> numpy sees transpose
    Lh = np.tril(Lh)
    print(d, "chol err", np.abs(Lh @ Lh.T - Qh / step**2).max())
    x0 = mu + np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T
    x = eng.to_device(x0)
    xs = eng.empty(1, C, d)
    # zero steps with store? use 1 step with huge rejection: inject u = 1 (log u = 0 -> accept iff log_alpha > 0) -> use u ~ 1-1e-16
    z = eng.to_device(rng.standard_normal((1, C, d)))
    u = eng.to_device(np.full((1, C), 1.0))
    acc = torch.zeros(C, dtype=torch.int64, device="cuda")
    eng.mala_run_white(eng.to_device(mu), L, sl, step, x, 1, z=z, u=u, x_store=xs, accept_count=acc)
    got = xs.cpu().numpy()[0]
    err = np.abs(got - x0)
    print(d, "acc", acc.cpu().numpy(), "per-chain roundtrip err", err.max(axis=1), "bad elements of rejected chains", [np.nonzero(err[c] > 1e-8)[0][:6].tolist() + [int((err[c] > 1e-8).sum())] for c in range(C) if acc[c].item() == 0])
    eng.close()
