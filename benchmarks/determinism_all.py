#!/usr/bin/env python3
"""Run-to-run reproducibility of the other production-size workloads (the headline kernel: benchmarks/determinism.py, the
hierarchical smoother: determinism_hier.py): each is run twice from the same seed and state in one process and must leave
bit-equal results.  A race does not reproduce; a difference here is one.
python benchmarks/determinism_all.py [band] [cfg2] [cfg2chol] [cfg4] [cfg4prod] [cfg5] [trunc]   (default: all)"""
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from scipy import sparse

from openmcmc_amd.engine import Engine


def band(order=2, n=10000, C=1024, draws=6):
    eng = Engine(C, seed=2)
    rng = np.random.default_rng(0)
    D = sparse.identity(n, format="csr")
    for _ in range(order):
        D = D[1:] - D[:-1]
    P = (D.T @ D + 1e-3 * sparse.identity(n)).tocsc()
    bandm = np.zeros((order + 1, n))
    for d in range(order + 1):
        bandm[d, : n - d] = P.diagonal(-d)
    y = np.sin(np.arange(n) * 0.003) + rng.standard_normal(n)
    T = eng.band_terms([{"band": eng.to_device(bandm), "scale": eng.to_device(100.0 * (0.5 + rng.random(C)))},
                        {"rhs": eng.to_device(y), "scale": eng.to_device(0.5 + rng.random(C))}], n)
    out = []
    for i in range(draws):
        x, m, ld = eng.empty(C, n), eng.empty(C, n), eng.empty(C)
        eng.band_sample_canonical(n, T, x, draw_index=i, mean_out=m, logdet_out=ld)
        out += [x, m, ld]
    eng.check_status()
    res = {f"t{i}": t.clone() for i, t in enumerate(out)}
    res["fallbacks"] = torch.tensor([eng.counter("band_join_fallbacks"), eng.counter("band_join_retries")])
    eng.close()
    return res


def cfg2(spectral=True, steps=40):
    n, p, C = 10000, 1000, 256
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, p))
    y = X @ rng.standard_normal(p) + 0.1 * rng.standard_normal(n)
    eng = Engine(C, seed=1)
    dX, dy = eng.to_device(X), eng.to_device(y)
    Gram = eng.gram(dX)
    Xty = eng.design_rhs(dX, dy)
    lam, tau = eng.full((C,), 0.01), eng.full((C,), 1.0)
    terms = eng.dense_terms([{"mat": None, "scale": lam}, {"mat": Gram, "rhs": Xty, "scale": tau}], p)
    ident = eng.tridiag_terms([{}], p)
    b, fitted = eng.empty(C, p), eng.empty(C, n)
    q_tau, q_lam = eng.empty(C), eng.empty(1, C)
    V, ev = eng.dense_spectral_prepare(Gram)
    for it in range(steps if spectral else max(4, steps // 8)):
        if spectral:
            eng.dense_spectral_sample(p, terms, 1, V, ev, b, draw_index=3 * it)
        else:
            eng.dense_sample_canonical(p, terms, b, draw_index=3 * it)
        eng.design_predict(dX, b, fitted)
        eng.weighted_resid_sq(dy, fitted, q_tau)
        eng.normal_gamma_update(1e-3, 1e-3, n, q_tau, tau, draw_index=3 * it + 1)
        eng.tridiag_quadform(p, ident, b, q_lam)
        eng.normal_gamma_update(1e-3, 1e-3, p, q_lam[0], lam, draw_index=3 * it + 2)
    eng.check_status()
    res = {"gram": Gram.clone(), "b": b.clone(), "tau": tau.clone(), "lam": lam.clone(), "fitted": fitted.clone()}
    eng.close()
    return res


def cfg4(products=False, steps=300):
    d, C = 500, 512
    rng = np.random.default_rng(0)
    A = rng.standard_normal((d, 2 * d))
    Qh = np.linalg.inv(A @ A.T / (2 * d))
    Qh = (Qh + Qh.T) / 2
    eng = Engine(C, seed=3)
    Q = eng.to_device(Qh)
    step = 0.5
    L, sl = eng.dense_cholesky(Q, 1.0 / step**2)
    x = eng.to_device(np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T)
    acc = torch.zeros(C, dtype=torch.int64, device="cuda")
    prop = torch.zeros(C, dtype=torch.int64, device="cuda")
    for it in range(steps):
        if products:
            eng.mala_step(Q, None, L, sl, step, x, draw_index=it, accept_count=acc, proposal_count=prop)
        else:
            eng.mala_step_white(None, L, sl, step, x, state_is_current=it > 0, draw_index=it, accept_count=acc, proposal_count=prop)
    eng.check_status()
    res = {"x": x.clone(), "acc": acc.clone()}
    eng.close()
    return res


def cfg5(steps=30):
    from rj_problem import build, make_basis_host

    from openmcmc_amd import gmrf
    from openmcmc_amd.mcmc import MCMC

    n, n_max, C = 5000, 20, 512
    rng = np.random.default_rng(0)
    X = np.linspace(-10, 10, n)
    b_true = 0.05 * np.cumsum(rng.standard_normal(n)) * np.sqrt(48.0 / n)
    y = (make_basis_host(X.reshape(n, 1), np.array([[-6.0, -1.0, 4.5]])) @ np.array([[3.0], [-2.0], [4.0]])).ravel() + b_true \
        + 0.1 * rng.standard_normal(n)
    P = gmrf.precision_irregular(np.arange(float(n))).tolil()
    P[0, 0] += 1e-3
    k0 = np.clip(rng.poisson(5, size=C), 1, n_max)
    init_theta = [rng.uniform(-10, 10, size=k) for k in k0]
    init_beta = [rng.standard_normal(k) for k in k0]
    eng = Engine(C, seed=1)
    mdl, state, samplers = build(y, X, P.tocsc(), n_max, eng, init_theta, init_beta, k0.astype(float))
    M = MCMC(state, samplers, model=mdl, n_burn=5, n_iter=steps, n_chains=C, seed=1, engine=eng)
    with contextlib.redirect_stdout(sys.stderr):
        M.run_mcmc()
    res = {k: v.clone() for k, v in M.store.items() if isinstance(v, torch.Tensor)}
    eng.close()
    return res


def trunc(n=4000, C=256, scans=3):
    eng = Engine(C, seed=5)
    rng = np.random.default_rng(0)
    diag = np.full(n, 2.0); diag[0] = diag[-1] = 1.0
    T = eng.tridiag_terms([{"diag": eng.to_device(diag), "off": eng.to_device(np.full(n - 1, -1.0)), "scale": eng.to_device(50.0 * (0.5 + rng.random(C)))},
                           {"rhs": eng.to_device(rng.standard_normal(n) + 1.0), "scale": eng.to_device(0.5 + rng.random(C))}], n)
    x = eng.full((C, n), 0.5)
    lo = eng.to_device(np.zeros(n))
    for s in range(scans):
        eng.tridiag_gibbs_truncated(n, T, x, lower=lo, draw_index=s)
    eng.check_status()
    res = {"x": x.clone()}
    eng.close()
    return res


WORK = {"band": lambda: band(2), "band1": lambda: band(1), "band3": lambda: band(3, C=512), "cfg2": lambda: cfg2(True), "cfg2chol": lambda: cfg2(False),
        "cfg4": lambda: cfg4(False), "cfg4prod": lambda: cfg4(True, 60), "cfg5": cfg5, "trunc": trunc}

if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in WORK] or list(WORK)
    bad = 0
    for name in names:
        try:
            a, b = WORK[name](), WORK[name]()
        except Exception as e:  # (a failing workload is reported, the others still run)
            print(f"{name}: FAILED {type(e).__name__}: {e}", flush=True)
            bad += 1
            continue
        diff = []
        for k in a:
            x, y = a[k], b[k]
            same = (x == y) | (torch.isnan(x) & torch.isnan(y)) if x.is_floating_point() else (x == y)
            if not bool(same.all()):
                diff.append(f"{k} ({int((~same).sum())} of {same.numel()} entries)")
        print(f"{name}: " + ("bit-equal over " + ", ".join(a) if not diff else "DIFFERS in " + "; ".join(diff)), flush=True)
        bad += bool(diff)
    sys.exit(1 if bad else 0)
